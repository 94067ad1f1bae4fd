// TEST INFRASTRUCTURE.  minimap2_amd/csrc/hits.cpp -- the host's chains -> hits bookkeeping, filters, sorts, fragment rules and pairing -- against the
// reference's OWN functions (hit.c, pe.c, align.c's mm_update_dp_max; linked from oracle/_ref/libminimap2_ref.a) on random hit lists built to make the
// rules bite: equal scores and hashes (the unstable sorts), overlapping query intervals (parents), dead hits, ALT hits, hits with and without alignments,
// both reads' hits interleaved on the reference (pairing).  Every field of every record is compared, alignments (mm_extra_t) by content.
// Prints "OK <cases>" or fails with the first difference.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include <vector>
#include "../../minimap2_amd/csrc/hits.hpp"

using mm2amd::Anchor;
using mm2amd::Reg;
using mm2amd::RegVec;
namespace ref = mm2amd::ref;

struct SegRef { int n_u, n_a; uint64_t *u; Anchor *a; }; // mm_seg_t, mmpriv.h:53-57
extern "C" {
Reg *mm_gen_regs(void *km, uint32_t hash, int qlen, int n_u, uint64_t *u, Anchor *a, int is_qstrand);
void mm_sync_regs(void *km, int n_regs, Reg *regs);
int mm_squeeze_a(void *km, int n_regs, Reg *regs, Anchor *a);
int mm_set_sam_pri(int n, Reg *r);
void mm_set_parent(void *km, float mask_level, int mask_len, int n, Reg *r, int sub_diff, int hard_mask_level, float alt_diff_frac);
void mm_select_sub(void *km, float pri_ratio, int min_diff, int best_n, int check_strand, int min_strand_sc, int *n_, Reg *r);
void mm_select_sub_multi(void *km, float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs, const int *qlens, int *n_, Reg *r);
int mm_filter_strand_retained(int n_regs, Reg *r);
void mm_filter_regs(const ref::MapOpt *opt, int qlen, int *n_regs, Reg *regs);
void mm_hit_sort(void *km, int *n_regs, Reg *r, float alt_diff_frac);
void mm_set_mapq2(void *km, int n_regs, Reg *regs, int min_chain_sc, int match_sc, int rep_len, int is_sr, int is_splice);
void mm_update_dp_max(int qlen, int n_regs, Reg *regs, float frac, int a, int b);
SegRef *mm_seg_gen(void *km, uint32_t hash, int n_segs, const int *qlens, int n_regs0, const Reg *regs0, int *n_regs, Reg **regs, const Anchor *a);
void mm_seg_free(void *km, int n_segs, SegRef *segs);
void mm_pair(void *km, int max_gap_ref, int dp_bonus, int sub_diff, int match_sc, const int *qlens, int *n_regs, Reg **regs);
void mm_est_err(const ref::Idx *mi, int qlen, int n_regs, Reg *regs, const Anchor *a, int32_t n, const uint64_t *mini_pos);
}

static std::mt19937_64 rng(20260923);
static int rnd(int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); } // inclusive
static bool coin(int one_in) { return rng() % (uint64_t)one_in == 0; }
static long n_div_pos, n_div_skipped, n_paired, n_rescored, n_sorted_out, n_multi_dropped, n_secondary_dropped, n_mapq_nonzero; // did the cases reach the rules they are for?

static ref::Extra *make_extra(int n_cigar, int dp_max)
{
	ref::Extra *p = (ref::Extra *)calloc(1, sizeof(ref::Extra) + (size_t)(n_cigar > 0 ? n_cigar : 1) * 4);
	p->capacity = (uint32_t)(sizeof(ref::Extra) / 4 + (n_cigar > 0 ? n_cigar : 1));
	p->dp_max = p->dp_max0 = p->dp_score = dp_max, p->dp_max2 = dp_max / 2, p->n_cigar = (uint32_t)n_cigar, p->n_ambi = (uint32_t)rnd(0, 3);
	for (int k = 0; k < n_cigar; ++k) p->cigar[k] = (uint32_t)rnd(1, 60) << 4 | (uint32_t)(k % 2 == 0 ? 0 : rnd(0, 3) == 0 ? 0 : rnd(1, 2));
	return p;
}
static ref::Extra *clone_extra(const ref::Extra *p)
{
	if (!p) return nullptr;
	const size_t bytes = sizeof(ref::Extra) + (size_t)(p->n_cigar > 0 ? p->n_cigar : 1) * 4;
	ref::Extra *c = (ref::Extra *)malloc(bytes);
	memcpy(c, p, bytes);
	return c;
}
static RegVec clone(const RegVec &r) { RegVec c = r; for (Reg &h : c) h.p = clone_extra(h.p); return c; }
static void release(RegVec &r) { for (Reg &h : r) free(h.p), h.p = nullptr; }

static void fail(const char *what, int it, const char *detail) { fprintf(stderr, "hits_test: %s, case %d: %s\n", what, it, detail); exit(1); }
static void same(const char *what, int it, const Reg *a, int na, const Reg *b, int nb)
{
	if (na != nb) { char m[96]; snprintf(m, sizeof m, "%d hits against the reference's %d", na, nb); fail(what, it, m); }
	for (int i = 0; i < na; ++i) {
		Reg x = a[i], y = b[i];
		const ref::Extra *px = x.p, *py = y.p;
		x.p = y.p = nullptr;
		if (memcmp(&x, &y, sizeof(Reg)) != 0) { char m[160]; snprintf(m, sizeof m, "hit %d differs (id %d/%d parent %d/%d score %d/%d mapq %d/%d subsc %d/%d)", i, x.id, y.id, x.parent, y.parent, x.score, y.score, (int)x.mapq, (int)y.mapq, x.subsc, y.subsc); fail(what, it, m); }
		if (!px != !py) fail(what, it, "an alignment is kept on one side only");
		if (px && (px->n_cigar != py->n_cigar || px->dp_max != py->dp_max || px->dp_max2 != py->dp_max2 || px->dp_score != py->dp_score || memcmp(px->cigar, py->cigar, (size_t)px->n_cigar * 4) != 0)) fail(what, it, "alignments differ");
	}
}

// hits of one read, best chain score first, the way the chaining step and mm_gen_regs leave them (ids = positions), then parents as the reference sets them
static RegVec random_hits(int n, int qlen, bool aligned, int n_rid = 3)
{
	RegVec r((size_t)n);
	int score = rnd(50, 4000);
	for (int i = 0; i < n; ++i) {
		Reg &h = r[i];
		memset(&h, 0, sizeof h);
		h.id = h.parent = i;
		h.qs = rnd(0, qlen - 20), h.qe = h.qs + rnd(10, qlen - h.qs);
		if (coin(3) && i > 0) h.qs = r[i - 1].qs + rnd(-5, 5) < 0 ? 0 : r[i - 1].qs, h.qe = r[i - 1].qe; // nested in / equal to the one before: parents and sub-scores
		if (h.qe > qlen) h.qe = qlen;
		if (h.qe <= h.qs) h.qe = h.qs + 1;
		h.rid = rnd(0, n_rid - 1), h.rs = rnd(0, 5000), h.re = h.rs + (h.qe - h.qs) + rnd(-3, 3);
		if (h.re <= h.rs) h.re = h.rs + 1;
		h.rev = (uint32_t)rnd(0, 1);
		if (!coin(3)) score -= rnd(0, 40);
		if (score < 1) score = 1;
		h.score = h.score0 = score;
		h.cnt = coin(12) ? 0 : rnd(1, 40), h.as = 0;
		h.mlen = rnd(10, h.qe - h.qs > 10 ? h.qe - h.qs : 10), h.blen = h.mlen + rnd(0, 30);
		h.hash = coin(2) ? (uint32_t)rnd(0, 3) : (uint32_t)rng(); // few distinct values: equal sort keys
		h.div = (float)rnd(0, 200) / 1000.0f;
		h.is_alt = coin(6), h.inv = coin(15), h.strand_retained = coin(5), h.seg_split = coin(9), h.is_spliced = coin(10);
		h.subsc = 0, h.n_sub = 0;
		if (aligned) h.p = make_extra(rnd(1, 9), rnd(0, 3) == 0 ? score : score * 2 + rnd(-20, 20));
	}
	return r;
}

int main(int argc, char **argv)
{
	const int n_case = argc > 1 ? atoi(argv[1]) : 3000;
	for (int it = 0; it < n_case; ++it) {
		const int qlen = rnd(60, 3000), n = rnd(1, it % 10 == 0 ? 60 : 9);
		const float alt_frac = coin(2) ? 0.15f : 0.0f, mask_level = coin(3) ? 0.9f : 0.5f;
		const int mask_len = coin(2) ? 2147483647 : rnd(10, 500), sub_diff = rnd(0, 12);
		{ // parents -> secondaries -> renumbering -> SAM primary -> MAPQ (hit.c:125-186, :255-281, :231-253, :220-229, :432-485)
			RegVec a = random_hits(n, qlen, it % 2 == 0), b = clone(a);
			const bool hard = coin(4);
			mm2amd::set_parent(mask_level, mask_len, a, sub_diff, hard, alt_frac);
			mm_set_parent(nullptr, mask_level, mask_len, (int)b.size(), b.data(), sub_diff, hard, alt_frac);
			same("set_parent", it, a.data(), (int)a.size(), b.data(), (int)b.size());
			const float pri = coin(5) ? 0.0f : 0.8f;
			const int min_diff = rnd(0, 30), best_n = rnd(0, 6), chk = rnd(0, 1), min_strand = rnd(0, 3000);
			int nb = (int)b.size();
			const size_t before_sub = a.size();
			mm2amd::select_sub(pri, min_diff, best_n, chk, min_strand, a);
			n_secondary_dropped += a.size() != before_sub;
			mm_select_sub(nullptr, pri, min_diff, best_n, chk, min_strand, &nb, b.data());
			same("select_sub", it, a.data(), (int)a.size(), b.data(), nb);
			b.resize((size_t)nb);
			if (mm2amd::set_sam_pri(a) != mm_set_sam_pri(nb, b.data())) fail("set_sam_pri", it, "counts differ");
			same("set_sam_pri", it, a.data(), (int)a.size(), b.data(), nb);
			const int rep_len = rnd(0, qlen), sr = coin(4), spl = coin(4), min_sc = rnd(20, 60), match_sc = rnd(1, 4);
			mm2amd::set_mapq(a, min_sc, match_sc, rep_len, sr, spl);
			mm_set_mapq2(nullptr, nb, b.data(), min_sc, match_sc, rep_len, sr, spl);
			same("set_mapq", it, a.data(), (int)a.size(), b.data(), nb);
			for (const Reg &h : a) n_mapq_nonzero += h.mapq > 0 && h.mapq < 60;
			mm2amd::filter_strand_retained(a);
			nb = mm_filter_strand_retained(nb, b.data());
			same("filter_strand_retained", it, a.data(), (int)a.size(), b.data(), nb);
			b.resize((size_t)nb);
			release(a), release(b);
		}
		{ // filter -> rescoring -> sort (hit.c:301-320, align.c:1005-1046, hit.c:188-218)
			RegVec a = random_hits(n, qlen, it % 3 != 0), b = clone(a);
			ref::MapOpt opt;
			memset(&opt, 0, sizeof opt);
			opt.min_cnt = rnd(1, 5), opt.min_chain_score = rnd(10, 60), opt.min_dp_max = rnd(0, 400), opt.max_clip_ratio = coin(2) ? 1.0f : 0.3f;
			int nb = (int)b.size();
			mm2amd::filter_regs(opt, qlen, a);
			mm_filter_regs(&opt, qlen, &nb, b.data());
			same("filter_regs", it, a.data(), (int)a.size(), b.data(), nb);
			b.resize((size_t)nb);
			const float frac = coin(2) ? 0.8f : 0.2f;
			const int ma = rnd(1, 4), mb = rnd(2, 8);
			{ long sum0 = 0, sum1 = 0; for (const Reg &h : a) if (h.p) sum0 += h.p->dp_max;
			mm2amd::update_dp_max(qlen, a, frac, ma, mb);
			for (const Reg &h : a) if (h.p) sum1 += h.p->dp_max; n_rescored += sum0 != sum1; }
			mm_update_dp_max(qlen, nb, b.data(), frac, ma, mb);
			same("update_dp_max", it, a.data(), (int)a.size(), b.data(), nb);
			const size_t before_sort = a.size();
			mm2amd::hit_sort(a, alt_frac);
			n_sorted_out += a.size() != before_sort;
			mm_hit_sort(nullptr, &nb, b.data(), alt_frac);
			same("hit_sort", it, a.data(), (int)a.size(), b.data(), nb);
			b.resize((size_t)nb);
			release(a), release(b);
		}
		{ // chains -> hits, the anchor squeeze, a fragment's chains cut by segment (hit.c:52-88, :322-340, :342-396)
			const int n_u = rnd(1, 12), n_segs = 2, qlens[2] = { rnd(50, 250), rnd(50, 250) }, qsum = qlens[0] + qlens[1];
			std::vector<uint64_t> u((size_t)n_u);
			std::vector<Anchor> anchors;
			for (int c = 0; c < n_u; ++c) {
				const int cnt = rnd(1, 12), rev = rnd(0, 1), rid = rnd(0, 2);
				u[(size_t)c] = (uint64_t)rnd(40, 900) << 32 | (uint32_t)cnt;
				int rpos = rnd(100, 100000), qpos = rnd(20, qsum / 2);
				for (int k = 0; k < cnt; ++k) {
					rpos += rnd(1, 40), qpos += rnd(1, 25);
					if (qpos >= qsum) qpos = qsum - 1;
					const int seg = (rev ? qsum - 1 - qpos : qpos) < qlens[0] ? 0 : 1; // which read the position falls on
					Anchor x;
					x.x = (uint64_t)rev << 63 | (uint64_t)rid << 32 | (uint32_t)rpos;
					x.y = (uint64_t)15 << 32 | (uint64_t)seg << ref::SEED_SEG_SHIFT | (uint32_t)qpos;
					anchors.push_back(x);
				}
			}
			std::vector<uint64_t> u2 = u;
			std::vector<Anchor> a1 = anchors, a2 = anchors;
			const uint32_t hash = (uint32_t)rng();
			RegVec ours;
			mm2amd::gen_regs(hash, qsum, u.data(), n_u, a1.data(), false, ours);
			Reg *theirs = mm_gen_regs(nullptr, hash, qsum, n_u, u2.data(), a2.data(), 0);
			same("gen_regs", it, ours.data(), (int)ours.size(), theirs, n_u);
			RegVec seg_ours[2];
			std::vector<Anchor> seg_a[2];
			mm2amd::seg_gen(hash, n_segs, qlens, ours, a1.data(), seg_ours, seg_a);
			int n_seg_regs[2];
			Reg *seg_theirs[2];
			SegRef *sg = mm_seg_gen(nullptr, hash, n_segs, qlens, n_u, theirs, n_seg_regs, seg_theirs, a2.data());
			for (int s = 0; s < n_segs; ++s) {
				same("seg_gen", it, seg_ours[s].data(), (int)seg_ours[s].size(), seg_theirs[s], n_seg_regs[s]);
				if ((int)seg_a[s].size() != sg[s].n_a || (sg[s].n_a && memcmp(seg_a[s].data(), sg[s].a, (size_t)sg[s].n_a * sizeof(Anchor)) != 0)) fail("seg_gen", it, "a segment's anchors differ");
				free(seg_theirs[s]);
			}
			mm_seg_free(nullptr, n_segs, sg);
			// drop some hits, then squeeze
			RegVec kept;
			std::vector<Reg> kept2;
			for (int i = 0; i < n_u; ++i) if (!coin(3)) kept.push_back(ours[(size_t)i]), kept2.push_back(theirs[i]);
			const int na = mm2amd::squeeze_anchors(kept, a1.data()), nb2 = mm_squeeze_a(nullptr, (int)kept2.size(), kept2.data(), a2.data());
			if (na != nb2 || memcmp(a1.data(), a2.data(), (size_t)na * sizeof(Anchor)) != 0) fail("squeeze_anchors", it, "anchors differ");
			same("squeeze_anchors", it, kept.data(), (int)kept.size(), kept2.data(), (int)kept2.size());
			free(theirs);
		}
		{ // divergence from the minimizers a chain misses (esterr.c:30-64): chains over random subsets of the read's minimizers, both strands, spans of k or (homopolymer-compressed) of their own
			const int k = rnd(11, 19), n_mini = rnd(1, 400), n_seq = 3;
			const bool hpc = coin(3);
			std::vector<uint64_t> mini((size_t)n_mini);
			int pos = k;
			for (int i = 0; i < n_mini; ++i) { const int span = hpc ? k + rnd(0, 20) : k; pos += rnd(1, 12); mini[(size_t)i] = (uint64_t)span << 32 | (uint32_t)pos; }
			const int rlen = pos + rnd(1, 60);
			mm2amd::FlatIndex fi;
			fi.k = k, fi.w = 10, fi.flag = hpc ? ref::I_HPC : 0, fi.n_seq = n_seq;
			ref::IdxSeq seqs[3];
			memset(seqs, 0, sizeof seqs);
			for (int i = 0; i < n_seq; ++i) { const uint32_t l = (uint32_t)rnd(500, 200000); fi.seq_len.push_back(l), seqs[i].len = l; }
			ref::Idx mi;
			memset(&mi, 0, sizeof mi);
			mi.k = k, mi.w = 10, mi.flag = fi.flag, mi.n_seq = n_seq, mi.seq = seqs;
			const int n_hit = rnd(1, 8);
			RegVec a((size_t)n_hit);
			std::vector<Anchor> anchors;
			for (Reg &h : a) {
				memset(&h, 0, sizeof h);
				h.rev = (uint32_t)rnd(0, 1), h.rid = rnd(0, n_seq - 1), h.as = (int32_t)anchors.size();
				std::vector<int> pick;
				for (int i = rnd(0, n_mini - 1); i < n_mini && (int)pick.size() < 60; i += rnd(1, 4)) pick.push_back(i);
				if (coin(10)) pick.clear(); // a hit that lost its anchors
				h.cnt = (int32_t)pick.size();
				for (int c = 0; c < h.cnt; ++c) {
					const uint64_t m = mini[(size_t)pick[(size_t)(h.rev ? h.cnt - 1 - c : c)]];
					const int span = (int)(m >> 32 & 0xff), p = (int)(uint32_t)m;
					Anchor x;
					x.x = (uint64_t)h.rev << 63 | (uint64_t)h.rid << 32 | (uint32_t)(1000 + (int)anchors.size());
					x.y = (uint64_t)span << 32 | (uint32_t)(h.rev ? rlen - 1 - (p + 1 - span) : p);
					if (coin(40)) x.y += 1; // a position that is no minimizer's: the first one makes the reference give the hit up, a later one is a miss
					anchors.push_back(x);
				}
				h.qs = rnd(0, rlen - 1), h.qe = rlen - rnd(0, 30), h.rs = rnd(0, 40), h.re = (int32_t)fi.seq_len[(size_t)h.rid] - rnd(0, 40);
			}
			RegVec b = a;
			mm2amd::est_err(fi, rlen, a, anchors.data(), mini.data(), n_mini);
			mm_est_err(&mi, rlen, n_hit, b.data(), anchors.data(), n_mini, mini.data());
			same("est_err", it, a.data(), n_hit, b.data(), n_hit);
			for (const Reg &h : a) n_div_pos += h.div > 0.0f, n_div_skipped += h.div < 0.0f;
		}
		{ // a fragment's secondaries; the two reads' hits paired (pe.c:6-50, :81-182)
			const int qlens[2] = { rnd(80, 250), rnd(80, 250) };
			RegVec f = random_hits(n, qlens[0] + qlens[1], false);
			mm_set_parent(nullptr, mask_level, mask_len, (int)f.size(), f.data(), sub_diff, 0, alt_frac);
			RegVec g = clone(f);
			int ng = (int)g.size();
			const int gap_ref = rnd(100, 2000), best_n = rnd(0, 5), min_diff = rnd(0, 30);
			const size_t before_multi = f.size();
			mm2amd::select_sub_multi(0.8f, 0.2f, 0.7f, gap_ref, min_diff, best_n, 2, qlens, f);
			n_multi_dropped += f.size() != before_multi;
			mm_select_sub_multi(nullptr, 0.8f, 0.2f, 0.7f, gap_ref, min_diff, best_n, 2, qlens, &ng, g.data());
			same("select_sub_multi", it, f.data(), (int)f.size(), g.data(), ng);
			RegVec ends[2], ends2[2];
			for (int s = 0; s < 2; ++s) {
				ends[s] = random_hits(coin(8) ? 0 : rnd(1, it % 10 == 0 ? 25 : 6), qlens[s], true, 2);
				const bool level = coin(2); // few distinct alignment scores and hashes: pairs with EQUAL keys, the first one found must win
				for (Reg &h : ends[s]) {
					h.rs = rnd(0, 3000), h.re = h.rs + rnd(30, coin(4) ? 2500 : 250); // close together: many candidate pairs; some long hits, so that reach is not monotone
					if (level) h.p->dp_max = 100 + 10 * rnd(0, 2), h.hash = (uint32_t)rnd(0, 1);
				}
				mm_set_parent(nullptr, mask_level, mask_len, (int)ends[s].size(), ends[s].data(), sub_diff, 0, alt_frac);
				mm_set_sam_pri((int)ends[s].size(), ends[s].data());
				mm_set_mapq2(nullptr, (int)ends[s].size(), ends[s].data(), 40, 2, 0, 1, 0);
				ends2[s] = clone(ends[s]);
			}
			int n_ends[2] = { (int)ends2[0].size(), (int)ends2[1].size() };
			Reg *pp[2] = { ends2[0].data(), ends2[1].data() };
			const int pair_gap = rnd(200, 3000), bonus = rnd(0, 60), sd = rnd(0, 12), msc = rnd(1, 4);
			mm2amd::pair_hits(pair_gap, bonus, sd, msc, qlens, ends);
			for (const Reg &h : ends[0]) n_paired += h.proper_frag;
			mm_pair(nullptr, pair_gap, bonus, sd, msc, qlens, n_ends, pp);
			for (int s = 0; s < 2; ++s) same("pair_hits", it, ends[s].data(), (int)ends[s].size(), ends2[s].data(), n_ends[s]), release(ends[s]), release(ends2[s]);
		}
	}
	if (n_case >= 500 && (n_paired < n_case / 20 || n_rescored < n_case / 50 || n_sorted_out < n_case / 50 || n_multi_dropped < n_case / 50 || n_secondary_dropped < n_case / 50 || n_mapq_nonzero < n_case / 20 || n_div_pos < n_case || n_div_skipped < n_case / 20))
		fail("coverage", n_case, "the random cases no longer reach a rule they are for");
	printf("OK %d (pairs found %ld, rescored %ld, dead hits sorted out %ld, secondaries dropped %ld + %ld, MAPQ between 1 and 59: %ld, divergence estimated %ld / given up %ld)\n", n_case, n_paired, n_rescored, n_sorted_out, n_secondary_dropped, n_multi_dropped, n_mapq_nonzero, n_div_pos, n_div_skipped);
	return 0;
}

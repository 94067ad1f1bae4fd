#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "hits.hpp"
#include "chain_host.hpp"
#include "hit_rules.hpp"

namespace mm2amd {

namespace {

inline int span_of(const Anchor &a) { return (int)(a.y >> 32 & 0xff); }

inline int alt_score(int score, float alt_diff_frac) { return hr_alt_score(score, alt_diff_frac); }

} // namespace

// coordinates and fuzzy lengths of a hit from its anchors [as, as + cnt) (mm_reg_set_coor, hit.c:24-38): hit_rules.hpp's pieces, every other field kept
void reg_set_coor(Reg &r, int32_t qlen, const Anchor *a, bool is_qstrand)
{
	Reg t;
	hr_new_hit(t, r.id, 0, r.as, r.cnt, qlen, a, is_qstrand);
	r.rev = t.rev, r.rid = t.rid, r.rs = t.rs, r.re = t.re, r.qs = t.qs, r.qe = t.qe;
	int bl = span_of(a[r.as]), ml = bl;
	for (int i = r.as + 1; i < r.as + r.cnt; ++i) hr_fuzzy_step(a[i], a[i - 1], &bl, &ml);
	r.blen = bl, r.mlen = ml;
}

// the read's chains as hit records, best first (mm_gen_regs, hit.c:52-88).  Equal keys are ordered by the reference's unstable sort, replayed here
// (the device path hands such reads back for exactly that)
void gen_regs(uint32_t hash, int qlen, const uint64_t *u, int n_u, const Anchor *a, bool is_qstrand, RegVec &out)
{
	out.clear();
	if (n_u <= 0) return;
	std::vector<Anchor> z(n_u); // x: the sort key; y: first anchor << 32 | anchors
	for (int i = 0, k = 0; i < n_u; k += (int32_t)u[i], ++i) z[i].x = hr_chain_key(u[i], a[k], hash), z[i].y = (uint64_t)k << 32 | (uint32_t)u[i];
	sort_by_x(z.data(), z.data() + n_u);
	out.resize(n_u);
	for (int i = 0; i < n_u; ++i) { // descending
		const Anchor &c = z[n_u - 1 - i];
		hr_new_hit(out[i], i, c.x, (int32_t)(c.y >> 32), (int32_t)c.y, qlen, a, is_qstrand);
		reg_set_coor(out[i], qlen, a, is_qstrand);
	}
}

void split_reg(Reg &r, Reg &r2, int n, int qlen, const Anchor *a, bool is_qstrand)
{
	if (n <= 0 || n >= r.cnt) return;
	r2 = r;
	r2.id = -1, r2.sam_pri = 0, r2.p = nullptr, r2.split_inv = 0;
	r2.cnt = r.cnt - n;
	r2.score = (int32_t)(r.score * ((float)r2.cnt / r.cnt) + .499);
	r2.as = r.as + n;
	if (r.parent == r.id) r2.parent = ref::PARENT_TMP_PRI;
	reg_set_coor(r2, qlen, a, is_qstrand);
	r.cnt -= r2.cnt, r.score -= r2.score;
	reg_set_coor(r, qlen, a, is_qstrand);
	r.split |= 1, r2.split |= 2;
}

void set_parent(float mask_level, int mask_len, RegVec &r, int sub_diff, bool hard_mask_level, float alt_diff_frac)
{
	const int n = (int)r.size();
	if (n <= 0) return;
	thread_local std::vector<uint64_t> cov;
	thread_local std::vector<int32_t> prim;
	cov.resize(n), prim.resize(n);
	hr_mark_parents(r.data(), n, cov.data(), prim.data(), mask_level, mask_len, sub_diff, hard_mask_level, alt_diff_frac); // hit_rules.hpp: the device kernel's formulation
}

void hit_sort(RegVec &r, float alt_diff_frac)
{
	const int n = (int)r.size();
	if (n <= 1) return;
	std::vector<Anchor> aux;
	aux.reserve(n);
	int has_cigar = 0, no_cigar = 0;
	for (int i = 0; i < n; ++i) {
		if (r[i].inv || r[i].cnt > 0) { // cnt==0 marks a soft-deleted hit
			int score;
			if (r[i].p) score = r[i].p->dp_max, has_cigar = 1;
			else score = r[i].score, no_cigar = 1;
			if (r[i].is_alt) score = alt_score(score, alt_diff_frac);
			aux.push_back(Anchor{(uint64_t)score << 32 | r[i].hash, (uint64_t)i});
		} else if (r[i].p) {
			free(r[i].p);
			r[i].p = nullptr;
		}
	}
	assert(has_cigar + no_cigar == 1);
	(void)has_cigar; (void)no_cigar;
	sort_by_x(aux.data(), aux.data() + aux.size());
	RegVec t(aux.size());
	for (int i = (int)aux.size() - 1; i >= 0; --i) t[aux.size() - 1 - i] = r[aux[i].y];
	r.swap(t);
}

int set_sam_pri(RegVec &r) { return hr_mark_sam_primary(r.data(), (int)r.size()); }

void sync_regs(RegVec &r)
{
	const int n = (int)r.size();
	if (n <= 0) return;
	int max_id = -1;
	for (const Reg &x : r) max_id = max_id > x.id ? max_id : x.id;
	thread_local std::vector<int32_t> where;
	where.resize((size_t)max_id + 1);
	hr_renumber(r.data(), n, where.data(), max_id + 1);
}

void select_sub(float pri_ratio, int min_diff, int best_n, bool check_strand, int min_strand_sc, RegVec &r)
{
	if (!(pri_ratio > 0.0f) || r.empty()) return;
	const int n = (int)r.size();
	thread_local std::vector<uint8_t> keep;
	keep.resize(n);
	hr_select_secondaries(r.data(), n, keep.data(), pri_ratio, min_diff, best_n, check_strand, min_strand_sc);
	int k = 0;
	for (int i = 0; i < n; ++i) {
		if (keep[i]) { if (k < i) r[k] = r[i]; ++k; }
		else if (r[i].p) free(r[i].p);
	}
	r.resize(k);
	if (k != n) sync_regs(r);
}

void filter_strand_retained(RegVec &r)
{
	const int n = (int)r.size();
	std::vector<uint8_t> keep(n);
	for (int i = 0; i < n; ++i) {
		const int p = r[i].parent;
		keep[i] = (!r[i].strand_retained || r[i].div < r[p].div * 5.0f || r[i].div < 0.01f);
	}
	int k = 0;
	for (int i = 0; i < n; ++i)
		if (keep[i]) { if (k < i) r[k] = r[i]; ++k; }
	r.resize(k);
}

void filter_regs(const ref::MapOpt &opt, int qlen, RegVec &regs)
{
	int k = 0;
	for (size_t i = 0; i < regs.size(); ++i) {
		Reg &r = regs[i];
		bool flt = false;
		if (!r.inv && !r.seg_split && r.cnt < opt.min_cnt) flt = true;
		if (r.p) {
			if (r.mlen < opt.min_chain_score) flt = true;
			else if (r.p->dp_max < opt.min_dp_max) flt = true;
			else if (r.qs > qlen * opt.max_clip_ratio && qlen - r.qe > qlen * opt.max_clip_ratio) flt = true;
			if (flt) free(r.p);
		}
		if (!flt) { if (k < (int)i) regs[k] = regs[i]; ++k; }
	}
	regs.resize(k);
}

int squeeze_anchors(RegVec &regs, Anchor *a)
{
	const int n = (int)regs.size();
	int as = 0;
	std::vector<uint64_t> aux(n);
	for (int i = 0; i < n; ++i) aux[i] = (uint64_t)regs[i].as << 32 | (uint32_t)i;
	sort_u64(aux.data(), aux.data() + n);
	for (int i = 0; i < n; ++i) {
		Reg &r = regs[(int32_t)aux[i]];
		if (r.as != as) {
			memmove(&a[as], &a[r.as], (size_t)r.cnt * sizeof(Anchor));
			r.as = as;
		}
		as += r.cnt;
	}
	return as;
}

namespace {
// What mm_set_mapq2 (hit.c:432-485) gives one primary hit.  A product of damping factors times the log of the hit's strength, minus a term for the number of
// near-equal secondaries; single-precision throughout, in the reference's operation order (every intermediate rounds as its float expression does), libm's logf.
struct MapqContext { float share_unique; int floor_sub, match_sc; bool short_reads, spliced_short, lone_spliced; };

int mapq_of(const Reg &h, const MapqContext &c)
{
	const float weak_score = (h.score > 100 ? 1.0f : 0.01f * h.score) * c.share_unique; // few bases chained, or much of the read is repetitive
	const float few_seeds = h.cnt > 10 ? 1.0f : 0.1f * h.cnt;
	const float damp = weak_score < few_seeds ? weak_score : few_seeds;
	const int runner_up = h.subsc > c.floor_sub ? h.subsc : c.floor_sub; // the best secondary chain, at least the chaining threshold
	int q;
	if (h.p && h.p->dp_max2 > 0 && h.p->dp_max > 0) { // aligned, with an aligned competitor
		const float identity = (float)h.mlen / h.blen;
		const float rel = c.spliced_short ? (float)h.p->dp_max2 / h.p->dp_max : (float)h.p->dp_max2 * runner_up / h.p->dp_max / h.score0;
		q = (int)(identity * damp * 40.0f * (1.0f - rel * rel) * logf((float)h.p->dp_max / c.match_sc));
		if (!c.short_reads) { // long reads: never more than the score gap to the competitor allows
			const int by_gap = (int)(6.02f * identity * identity * (h.p->dp_max - h.p->dp_max2) / c.match_sc + .499f);
			q = q < by_gap ? q : by_gap;
		}
		if (c.spliced_short && h.is_spliced && c.lone_spliced) q += 10;
	} else {
		const float rel = (float)runner_up / h.score0;
		if (h.p) q = (int)((float)h.mlen / h.blen * damp * 40.0f * (1.0f - rel) * logf((float)h.p->dp_max / c.match_sc));
		else q = (int)(damp * 40.0f * (1.0f - rel) * logf(h.score));
	}
	q -= (int)(4.343f * logf(h.n_sub + 1) + .499f);
	q = q > 0 ? (q < 60 ? q : 60) : 0;
	return h.p && h.p->dp_max > h.p->dp_max2 && q == 0 ? 1 : q; // an alignment that beats its competitor is never reported as ambiguous
}
}

void set_mapq(RegVec &regs, int min_chain_sc, int match_sc, int rep_len, bool is_sr, bool is_splice)
{
	const int n = (int)regs.size();
	if (n == 0) return;
	int64_t primary_score = 0;
	int spliced_secondaries = 0;
	bool any_inv = false;
	for (const Reg &r : regs) {
		if (r.parent == r.id) primary_score += r.score;
		else if (r.is_spliced) ++spliced_secondaries;
		any_inv |= r.inv;
	}
	MapqContext c;
	c.share_unique = (float)primary_score / (primary_score + rep_len);
	c.floor_sub = min_chain_sc, c.match_sc = match_sc, c.short_reads = is_sr, c.spliced_short = is_sr && is_splice, c.lone_spliced = spliced_secondaries == 0;
	for (Reg &r : regs) r.mapq = r.inv || r.parent != r.id ? 0 : (uint32_t)mapq_of(r, c);
	// an inversion takes the lower MAPQ of its two neighbours along the reference (mm_set_inv_mapq, hit.c:406-430)
	if (n < 3 || !any_inv) return;
	std::vector<Anchor> by_pos;
	for (int i = 0; i < n; ++i)
		if (regs[i].parent == i || regs[i].parent < 0) by_pos.push_back(Anchor{(uint64_t)regs[i].rid << 32 | (uint32_t)regs[i].rs, (uint64_t)i});
	sort_by_x(by_pos.data(), by_pos.data() + by_pos.size());
	for (size_t i = 1; i + 1 < by_pos.size(); ++i) {
		Reg &mid = regs[by_pos[i].y];
		if (!mid.inv) continue;
		const uint32_t left = regs[by_pos[i - 1].y].mapq, right = regs[by_pos[i + 1].y].mapq;
		mid.mapq = left < right ? left : right;
	}
}

void est_err(const FlatIndex &fi, int qlen, RegVec &regs, const Anchor *a, const uint64_t *mini_pos, int32_t n_mini_pos)
{
	const int32_t n = n_mini_pos;
	if (n == 0) return;
	uint64_t sum_k = 0;
	if (!(fi.flag & ref::I_HPC) && (uint64_t)n * (uint64_t)fi.k < (1u << 24)) sum_k = (uint64_t)n * (uint64_t)fi.k; // every span is k: the float quotient below is exactly k
	else for (int32_t i = 0; i < n; ++i) sum_k += mini_pos[i] >> 32 & 0xff;
	const float avg_k = (float)sum_k / n;
	for (Reg &r : regs) { // per hit: the counts by hit_rules.hpp (the device kernel's formulation: one search per anchor), libm's pow here
		r.div = -1.0f;
		if (r.cnt == 0) continue;
		const int st = hr_first_minimizer(mini_pos, n, hr_chain_qpos(r, a, qlen, 0));
		if (st < 0) continue;
		int32_t n_match, n_tot;
		hr_est_err_totals(r, a, qlen, mini_pos, n, st, hr_first_miss(r, a, qlen, mini_pos, n, st, 1, 1), avg_k, (int32_t)fi.seq_len[r.rid], &n_match, &n_tot);
		r.div = n_match >= n_tot ? 0.0f : (float)(1.0 - pow((double)n_match / n_tot, 1.0 / avg_k));
	}
}

namespace {
void count_gaps(const Reg &r, int32_t &n_gap, int32_t &n_gapo) // align.c:983-995
{
	n_gap = n_gapo = 0;
	for (uint32_t i = 0; i < r.p->n_cigar; ++i) {
		const int32_t op = r.p->cigar[i] & 0xf, len = r.p->cigar[i] >> 4;
		if (op == 1 || op == 2) ++n_gapo, n_gap += len;
	}
}
}

void update_dp_max(int qlen, RegVec &regs, float frac, int a, int b)
{
	const int n = (int)regs.size();
	if (n < 2) return;
	int32_t max = -1, max2 = -1, max_i = -1;
	for (int i = 0; i < n; ++i) {
		const Reg &r = regs[i];
		if (!r.p) continue;
		if (r.p->dp_max > max) max2 = max, max = r.p->dp_max, max_i = i;
		else if (r.p->dp_max > max2) max2 = r.p->dp_max;
	}
	if (max_i < 0 || max < 0 || max2 < 0) return;
	if (regs[max_i].qe - regs[max_i].qs < (double)qlen * frac) return;
	if (max2 < (double)max * frac) return;
	int32_t n_gap, n_gapo;
	count_gaps(regs[max_i], n_gap, n_gapo);
	const Reg &rm = regs[max_i];
	double div = 1. - (double)rm.mlen / (rm.blen + rm.p->n_ambi - n_gap + n_gapo); // 1 - mm_event_identity (align.c:997-1003)
	if (div < 0.02) div = 0.02;
	double b2 = 0.5 / div;
	if (b2 * a < b) b2 = (double)a / b;
	for (Reg &r : regs) {
		if (!r.p) continue;
		// rescore the alignment with a divergence-adapted mismatch penalty and log gap cost (align.c:1005-1020)
		int32_t gaps = 0;
		double gap_cost = 0.0;
		for (uint32_t i = 0; i < r.p->n_cigar; ++i) {
			const int32_t op = r.p->cigar[i] & 0xf, len = r.p->cigar[i] >> 4;
			if (op == 1 || op == 2) gap_cost += b2 + (double)fast_log2(1.0 + len), gaps += len;
		}
		const int32_t n_mis = r.blen + r.p->n_ambi - r.mlen - gaps;
		r.p->dp_max = (int32_t)(a * (r.mlen - b2 * n_mis - gap_cost) + .499);
		if (r.p->dp_max < 0) r.p->dp_max = 0;
	}
}

} // namespace mm2amd

// ---------------------------------------------------------------------------------------------------------
// Two-segment fragments (paired-end reads): hit.c:342-396 and pe.c
// ---------------------------------------------------------------------------------------------------------
namespace mm2amd {

// mm_seg_gen (hit.c:342-396): the chains of a fragment, found on the concatenation of its segments, are cut into one set of
// chains per segment; anchors are copied with the query coordinate made relative to their own segment.
void seg_gen(uint32_t hash, int n_segs, const int *qlens, const RegVec &regs0, const Anchor *a, RegVec *regs, std::vector<Anchor> *seg_a)
{
	const int n_regs0 = (int)regs0.size();
	int acc_qlen[3] = {0, 0, 0}, qlen_sum;
	for (int s = 1; s < n_segs; ++s) acc_qlen[s] = acc_qlen[s - 1] + qlens[s - 1];
	qlen_sum = acc_qlen[n_segs - 1] + qlens[n_segs - 1];
	std::vector<uint64_t> u[2];
	size_t n_a[2] = {0, 0};
	for (int s = 0; s < n_segs; ++s) {
		u[s].resize(n_regs0);
		for (int i = 0; i < n_regs0; ++i) u[s][i] = (uint64_t)regs0[i].score << 32;
	}
	for (int i = 0; i < n_regs0; ++i)
		for (int j = 0; j < regs0[i].cnt; ++j) {
			const int sid = (int)((a[regs0[i].as + j].y & ref::SEED_SEG_MASK) >> ref::SEED_SEG_SHIFT);
			++u[sid][i], ++n_a[sid];
		}
	for (int s = 0; s < n_segs; ++s) {
		size_t k = 0;
		for (int i = 0; i < n_regs0; ++i) if ((int32_t)u[s][i] != 0) u[s][k++] = u[s][i]; // chains with no anchor on this segment vanish
		u[s].resize(k);
		seg_a[s].clear();
		seg_a[s].reserve(n_a[s]);
	}
	for (int i = 0; i < n_regs0; ++i)
		for (int j = 0; j < regs0[i].cnt; ++j) {
			Anchor a1 = a[regs0[i].as + j];
			const int sid = (int)((a1.y & ref::SEED_SEG_MASK) >> ref::SEED_SEG_SHIFT);
			a1.y -= a1.x >> 63 ? (uint64_t)(qlen_sum - (qlens[sid] + acc_qlen[sid])) : (uint64_t)acc_qlen[sid];
			seg_a[sid].push_back(a1);
		}
	for (int s = 0; s < n_segs; ++s) {
		gen_regs(hash, qlens[s], u[s].data(), (int)u[s].size(), seg_a[s].data(), false, regs[s]);
		for (Reg &r : regs[s]) r.seg_split = 1, r.seg_id = (uint32_t)s;
	}
}

// mm_select_sub_multi (pe.c:6-50)
void select_sub_multi(float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs, const int *qlens, RegVec &r)
{
	if (!(pri_ratio > 0.0f) || r.empty()) return;
	const int n = (int)r.size(), max_dist = n_segs == 2 ? qlens[0] + qlens[1] + max_gap_ref : 0;
	int n_2nd = 0, k = 0;
	std::vector<uint8_t> keep(n, 0);
	for (int i = 0; i < n; ++i) {
		int to_keep = 0;
		if (r[i].parent == i) to_keep = 1;
		else if (r[i].score + min_diff >= r[r[i].parent].score) to_keep = 1;
		else {
			const Reg &p = r[r[i].parent], &q = r[i];
			if (p.rev == q.rev && p.rid == q.rid && q.re - p.rs < max_dist && p.re - q.rs < max_dist) { // child and parent are close on the reference
				if (q.score >= p.score * pri1) to_keep = 1;
			} else {
				const int is_par_both = (n_segs == 2 && p.qs < qlens[0] && p.qe > qlens[0]);
				const int is_chi_both = (n_segs == 2 && q.qs < qlens[0] && q.qe > qlens[0]);
				if (is_chi_both || is_chi_both == is_par_both) {
					if (q.score >= p.score * pri_ratio) to_keep = 1;
				} else if (q.score >= p.score * pri2) to_keep = 1;
			}
		}
		if (to_keep && r[i].parent != i && n_2nd++ >= best_n) to_keep = 0;
		keep[i] = (uint8_t)to_keep;
	}
	for (int i = 0; i < n; ++i) {
		if (keep[i]) r[k++] = r[i];
		else if (r[i].p) free(r[i].p);
	}
	r.resize(k);
	if (k != n) sync_regs(r);
}

// mm_set_pe_thru (pe.c:52-71)
static void set_pe_thru(const int *qlens, RegVec *regs)
{
	int n_pri[2] = {0, 0}, pri[2] = {-1, -1};
	for (int s = 0; s < 2; ++s)
		for (int i = 0; i < (int)regs[s].size(); ++i)
			if (regs[s][i].id == regs[s][i].parent) ++n_pri[s], pri[s] = i;
	if (n_pri[0] == 1 && n_pri[1] == 1) {
		Reg &p = regs[0][pri[0]], &q = regs[1][pri[1]];
		if (p.rid == q.rid && p.rev == q.rev && abs(p.rs - q.rs) < 3 && abs(p.re - q.re) < 3
		    && ((p.qs == 0 && qlens[1] - q.qe == 0) || (q.qs == 0 && qlens[0] - p.qe == 0)))
			p.pe_thru = q.pe_thru = 1;
	}
}

namespace {
struct PairEnt { int s, rev; uint64_t key; Reg *r; };
struct PairKey { uint64_t operator()(const PairEnt &e) const { return e.key; } };
}

// mm_pair (pe.c:81-182): choose the best properly oriented pair of hits within max_gap_ref and adjust primaries / MAPQ
void pair_hits(int max_gap_ref, int pe_bonus, int sub_diff, int match_sc, const int *qlens, RegVec *regs)
{
	std::vector<PairEnt> a;
	int dp_thres = 0, segs = 0;
	for (int s = 0; s < 2; ++s) {
		int max = 0;
		for (Reg &r : regs[s]) {
			PairEnt e;
			e.s = s, e.r = &r, e.rev = r.rev;
			e.key = (uint64_t)r.rid << 32 | (uint64_t)(int64_t)(r.rs << 1 | (s ^ e.rev)); // pe.c:97: int operands widened into the key
			max = max > r.p->dp_max ? max : r.p->dp_max;
			a.push_back(e);
			segs |= 1 << s;
		}
		dp_thres += max;
	}
	if (segs != 3) return; // only one end is mapped
	dp_thres -= pe_bonus;
	if (dp_thres < 0) dp_thres = 0;
	const int n = (int)a.size();
	{ RsortScratch sc; exact_radix_sort(a.data(), a.data() + n, PairKey(), sc); } // radix_sort_pair: the unstable sort, replayed exactly
	int64_t max = -1;
	int max_idx[2] = {-1, -1}, last[2] = {-1, -1};
	std::vector<uint64_t> sc;
	for (int i = 0; i < n; ++i) {
		if (a[i].key & 1) { // reverse first read or forward second read
			if (last[a[i].rev] < 0) continue;
			Reg *r = a[i].r, *q = a[last[a[i].rev]].r;
			if (r->rid != q->rid || r->rs - q->re > max_gap_ref) continue;
			for (int j = last[a[i].rev]; j >= 0; --j) {
				if (a[j].rev != a[i].rev || a[j].s == a[i].s) continue;
				q = a[j].r;
				if (r->rid != q->rid || r->rs - q->re > max_gap_ref) break;
				if (r->p->dp_max + q->p->dp_max < dp_thres) continue;
				const int64_t score = (int64_t)(r->p->dp_max + q->p->dp_max) << 32 | (r->hash + q->hash);
				if (score > max) max = score, max_idx[a[j].s] = j, max_idx[a[i].s] = i;
				sc.push_back((uint64_t)score);
			}
		} else last[a[i].rev] = i; // forward first read or reverse second read
	}
	if (sc.size() > 1) sort_u64(sc.data(), sc.data() + sc.size());
	if (!sc.empty() && max > 0) { // found at least one pair
		Reg *r[2] = { a[max_idx[0]].r, a[max_idx[1]].r };
		r[0]->proper_frag = r[1]->proper_frag = 1;
		for (int s = 0; s < 2; ++s) {
			if (r[s]->id != r[s]->parent) { // lift to primary and update the parents
				Reg &p = regs[s][r[s]->parent];
				for (Reg &x : regs[s]) if (x.parent == p.id) x.parent = r[s]->id;
				p.mapq = 0;
			}
			if (!r[s]->sam_pri) {
				for (Reg &x : regs[s]) x.sam_pri = 0;
				r[s]->sam_pri = 1;
			}
		}
		int mapq_pe = r[0]->mapq > r[1]->mapq ? r[0]->mapq : r[1]->mapq, n_sub = 0;
		for (uint64_t v : sc) if ((v >> 32) + sub_diff >= (uint64_t)max >> 32) ++n_sub;
		if (sc.size() > 1) {
			const int mapq_pe_alt = (int)(6.02f * ((max >> 32) - (sc[sc.size() - 2] >> 32)) / match_sc - 4.343f * logf(n_sub));
			mapq_pe = mapq_pe < mapq_pe_alt ? mapq_pe : mapq_pe_alt;
		}
		if ((int)r[0]->mapq < mapq_pe) r[0]->mapq = (int)(.2f * r[0]->mapq + .8f * mapq_pe + .499f);
		if ((int)r[1]->mapq < mapq_pe) r[1]->mapq = (int)(.2f * r[1]->mapq + .8f * mapq_pe + .499f);
		if (sc.size() == 1) {
			if (r[0]->mapq < 2) r[0]->mapq = 2;
			if (r[1]->mapq < 2) r[1]->mapq = 2;
		} else if ((uint64_t)max >> 32 > sc[sc.size() - 2] >> 32) {
			if (r[0]->mapq < 1) r[0]->mapq = 1;
			if (r[1]->mapq < 1) r[1]->mapq = 1;
		}
	}
	set_pe_thru(qlens, regs);
}

} // namespace mm2amd

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "hits.hpp"
#include <stdexcept>
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
	thread_local std::vector<Anchor> z; // x: the sort key; y: first anchor << 32 | anchors  (kept per thread: a short read has a chain or two, and this ran three times per read pair)
	z.resize(n_u);
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

namespace {
// The hits a predicate keeps, in order; the alignments of the others are released.  Returns true if any went.
template <class Keep>
bool keep_hits(RegVec &r, Keep keep, bool release = true)
{
	size_t w = 0;
	for (size_t i = 0; i < r.size(); ++i) {
		if (keep(i)) { if (w != i) r[w] = r[i]; ++w; }
		else if (release && r[i].p) free(r[i].p), r[i].p = nullptr;
	}
	const bool any = w != r.size();
	r.resize(w);
	return any;
}
} // namespace

// mm_hit_sort (hit.c:188-218): hits from best to worst by (alignment score if aligned, else chain score; ALT hits marked down) with the hash as the
// tie-breaker; hits that were dropped earlier (no anchors left and not an inversion) go.  The order among equal keys is the unstable radix sort's,
// so the sort itself is exact_rsort.hpp's replay of it, on (key, position) pairs.
void hit_sort(RegVec &r, float alt_diff_frac)
{
	if (r.size() <= 1) return;
	thread_local std::vector<Anchor> ranked;
	ranked.clear();
	for (size_t i = 0; i < r.size(); ++i) {
		Reg &h = r[i];
		const bool live = h.inv || h.cnt > 0;
		if (!live) { free(h.p); h.p = nullptr; continue; }
		const int raw = h.p ? h.p->dp_max : h.score; // (all hits are aligned or none is)
		const int score = h.is_alt ? alt_score(raw, alt_diff_frac) : raw;
		ranked.push_back(Anchor{(uint64_t)score << 32 | h.hash, (uint64_t)i});
	}
	sort_by_x(ranked.data(), ranked.data() + ranked.size());
	RegVec best_first;
	best_first.reserve(ranked.size());
	for (size_t k = ranked.size(); k-- > 0;) best_first.push_back(r[ranked[k].y]);
	r.swap(best_first);
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
	thread_local std::vector<uint8_t> keep;
	keep.resize(r.size());
	hr_select_secondaries(r.data(), (int)r.size(), keep.data(), pri_ratio, min_diff, best_n, check_strand, min_strand_sc);
	if (keep_hits(r, [&](size_t i) { return keep[i] != 0; })) sync_regs(r);
}

// mm_filter_strand_retained (hit.c:283-299): a hit kept only for being on the other strand stays if it is not much more diverged than its parent
void filter_strand_retained(RegVec &r)
{
	thread_local std::vector<uint8_t> stays;
	stays.resize(r.size());
	for (size_t i = 0; i < r.size(); ++i) { // (decided for all hits before any moves: parents are looked up by position)
		const Reg &h = r[i];
		stays[i] = !h.strand_retained || h.div < r[h.parent].div * 5.0f || h.div < 0.01f;
	}
	keep_hits(r, [&](size_t i) { return stays[i] != 0; }, false);
}

// mm_filter_regs (hit.c:301-320): what a hit needs to be reported -- enough anchors (unless it is an inversion or one segment's share of a chain), and,
// once aligned, enough matching bases, a high enough alignment score, and not both ends of the read hanging off by more than max_clip_ratio
void filter_regs(const ref::MapOpt &opt, int qlen, RegVec &regs)
{
	auto reportable = [&](const Reg &h) {
		if (!h.inv && !h.seg_split && h.cnt < opt.min_cnt) return false;
		if (!h.p) return true;
		if (h.mlen < opt.min_chain_score || h.p->dp_max < opt.min_dp_max) return false;
		return !(h.qs > qlen * opt.max_clip_ratio && qlen - h.qe > qlen * opt.max_clip_ratio);
	};
	keep_hits(regs, [&](size_t i) { return reportable(regs[i]); });
}

// mm_squeeze_a (hit.c:322-340): the anchors the hits still own, moved to the front of the array in the order they lie in it; returns how many
int squeeze_anchors(RegVec &regs, Anchor *a)
{
	thread_local std::vector<uint64_t> by_start; // anchor offset << 32 | hit: offsets are distinct, any sort gives the one order
	by_start.resize(regs.size());
	for (size_t i = 0; i < regs.size(); ++i) by_start[i] = (uint64_t)regs[i].as << 32 | (uint32_t)i;
	std::sort(by_start.begin(), by_start.end());
	int32_t packed = 0;
	for (const uint64_t e : by_start) {
		Reg &h = regs[(uint32_t)e];
		if (h.as != packed) memmove(a + packed, a + h.as, (size_t)h.cnt * sizeof(Anchor)), h.as = packed;
		packed += h.cnt;
	}
	return packed;
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
	if (!mini_pos) throw std::logic_error("[mm2amd] est_err: the read's minimizer positions were left on the device"); // (a caller that forgot to fetch them must not read through null)
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
// the gap operations of a hit's CIGAR: how many, and how many bases they span (mm_count_gaps, align.c:983-995)
struct GapTally { int32_t opens = 0, bases = 0; };
template <class PerGap>
GapTally tally_gaps(const Reg &h, PerGap per_gap)
{
	GapTally g;
	for (uint32_t k = 0; k < h.p->n_cigar; ++k) {
		const uint32_t op = h.p->cigar[k] & 0xf;
		if (op != 1 && op != 2) continue;
		const int32_t len = (int32_t)(h.p->cigar[k] >> 4);
		++g.opens, g.bases += len;
		per_gap(len);
	}
	return g;
}
} // namespace

// mm_update_dp_max (align.c:1005-1046), for reads with several aligned hits whose two best scores are close and whose best hit covers most of the read:
// the hits are ranked again by a score recomputed from their own columns under a mismatch penalty fitted to the best hit's divergence (one event per gap,
// at least 2 %) and a logarithmic gap cost.  Double arithmetic in the reference's order.
void update_dp_max(int qlen, RegVec &regs, float frac, int a, int b)
{
	if (regs.size() < 2) return;
	int32_t top = -1, second = -1;
	const Reg *best = nullptr;
	for (const Reg &h : regs) {
		if (!h.p) continue;
		if (h.p->dp_max > top) second = top, top = h.p->dp_max, best = &h;
		else if (h.p->dp_max > second) second = h.p->dp_max;
	}
	if (!best || top < 0 || second < 0) return;
	if (best->qe - best->qs < (double)qlen * frac || second < (double)top * frac) return;
	const GapTally bg = tally_gaps(*best, [](int32_t) {});
	double div = 1. - (double)best->mlen / (best->blen + best->p->n_ambi - bg.bases + bg.opens); // 1 - mm_event_identity (align.c:997-1003)
	if (div < 0.02) div = 0.02;
	double mis = 0.5 / div; // the mismatch penalty in units of the match score
	if (mis * a < b) mis = (double)a / b;
	for (Reg &h : regs) {
		if (!h.p) continue;
		double gap_cost = 0.0;
		const GapTally g = tally_gaps(h, [&](int32_t len) { gap_cost += mis + (double)fast_log2(1.0 + len); });
		const int32_t n_mis = h.blen + h.p->n_ambi - h.mlen - g.bases;
		const int32_t rescored = (int32_t)(a * (h.mlen - mis * n_mis - gap_cost) + .499);
		h.p->dp_max = rescored < 0 ? 0 : rescored;
	}
}

} // namespace mm2amd

// ---------------------------------------------------------------------------------------------------------
// Two-segment fragments (paired-end reads): hit.c:342-396 and pe.c
// ---------------------------------------------------------------------------------------------------------
namespace mm2amd {

// mm_seg_gen (hit.c:342-396): the chains of a fragment were found on the concatenation of its segments; every segment gets the chains that have anchors
// on it -- with the fragment chain's score and its own anchor count -- and those anchors, their query coordinate counted from the segment's own start
// (from its end for reverse-strand anchors, whose coordinate runs backwards over the concatenation).
void seg_gen(uint32_t hash, int n_segs, const int *qlens, const RegVec &regs0, const Anchor *a, RegVec *regs, std::vector<Anchor> *seg_a)
{
	auto seg_of = [](const Anchor &x) { return hr_anchor_seg(x); };
	int before[2] = {0, 0}, total = 0; // bases of the fragment before a segment; all of them
	for (int s = 0; s < n_segs; ++s) before[s] = total, total += qlens[s];
	const size_t n_chain = regs0.size();
	thread_local std::vector<int32_t> on_seg[2]; // anchors of every fragment chain on the segment
	thread_local std::vector<uint64_t> u;        // score << 32 | anchors, as the chaining step hands chains over
	for (int s = 0; s < n_segs; ++s) on_seg[s].assign(n_chain, 0), seg_a[s].clear();
	for (size_t c = 0; c < n_chain; ++c)
		for (int j = 0; j < regs0[c].cnt; ++j) {
			const Anchor x = a[regs0[c].as + j];
			const int s = seg_of(x);
			++on_seg[s][c];
			seg_a[s].push_back(hr_seg_anchor(x, total, before[s], qlens[s])); // (chain by chain, so a segment's anchors of one chain stay together, in order)
		}
	for (int s = 0; s < n_segs; ++s) {
		u.clear();
		for (size_t c = 0; c < n_chain; ++c)
			if (on_seg[s][c]) u.push_back((uint64_t)regs0[c].score << 32 | (uint32_t)on_seg[s][c]);
		gen_regs(hash, qlens[s], u.data(), (int)u.size(), seg_a[s].data(), false, regs[s]);
		for (Reg &h : regs[s]) h.seg_split = 1, h.seg_id = (uint32_t)s;
	}
}

// mm_select_sub_multi (pe.c:6-50): which secondary chains of a fragment stay.  A secondary within min_diff of its parent always does; otherwise it has to
// reach a share of the parent's score that depends on where it lies: next to the parent on the reference (a pair's other placement) a small one, a chain
// confined to one read against a parent that spans both a large one, else the usual ratio.  At most best_n secondaries.
void select_sub_multi(float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs, const int *qlens, RegVec &r)
{
	if (!(pri_ratio > 0.0f) || r.empty()) return;
	thread_local std::vector<uint8_t> stays;
	stays.assign(r.size(), 0);
	hr_select_secondaries_multi(r.data(), (int)r.size(), stays.data(), pri_ratio, pri1, pri2, max_gap_ref, min_diff, best_n, n_segs, qlens[0], n_segs > 1 ? qlens[1] : 0); // hit_rules.hpp: the device kernel's formulation
	if (keep_hits(r, [&](size_t i) { return stays[i] != 0; })) sync_regs(r);
}

namespace {
// the position of a segment's only primary hit, -1 if it has none or several
int sole_primary(const RegVec &regs)
{
	int at = -1, n = 0;
	for (size_t i = 0; i < regs.size(); ++i)
		if (regs[i].id == regs[i].parent) at = (int)i, ++n;
	return n == 1 ? at : -1;
}

// mm_set_pe_thru (pe.c:52-71): both reads cover the same stretch of the reference end to end (the fragment is no longer than a read)
void mark_read_through(const int *qlens, RegVec *regs)
{
	const int i0 = sole_primary(regs[0]), i1 = sole_primary(regs[1]);
	if (i0 < 0 || i1 < 0) return;
	Reg &p = regs[0][i0], &q = regs[1][i1];
	const bool same_place = p.rid == q.rid && p.rev == q.rev && abs(p.rs - q.rs) < 3 && abs(p.re - q.re) < 3;
	const bool end_to_end = (p.qs == 0 && qlens[1] - q.qe == 0) || (q.qs == 0 && qlens[0] - p.qe == 0);
	if (same_place && end_to_end) p.pe_thru = q.pe_thru = 1;
}

// One hit of either read as mm_pair (pe.c:81-182) looks at it: ordered along the reference; the lowest key bit says whether the hit can CLOSE a pair
// (reverse hit of read 1 / forward hit of read 2) or OPEN one
struct End { int seg, rev; uint64_t key; Reg *hit; };
struct EndKey { uint64_t operator()(const End &e) const { return e.key; } };
inline int pair_dp(const End &x, const End &y) { return x.hit->p->dp_max + y.hit->p->dp_max; }
} // namespace

// mm_pair (pe.c:81-182): the best pair of hits, one per read, properly oriented and no further apart than max_gap_ref, becomes the fragment's placement;
// the MAPQ of its two hits is raised towards a pair-level MAPQ computed from how far the best pair is ahead of the next.
void pair_hits(int max_gap_ref, int pe_bonus, int sub_diff, int match_sc, const int *qlens, RegVec *regs)
{
	// the ends, sorted by (sequence, start, closes-a-pair); a pair must beat the sum of the reads' best scores minus the bonus
	std::vector<End> ends;
	int bar = 0;
	for (int s = 0; s < 2; ++s) {
		if (regs[s].empty()) return; // only one read is mapped
		int top = 0;
		for (Reg &h : regs[s]) {
			ends.push_back(End{s, (int)h.rev, (uint64_t)h.rid << 32 | (uint64_t)(int64_t)(h.rs << 1 | (s ^ (int)h.rev)), &h}); // pe.c:97: int operands widened into the key
			top = top > h.p->dp_max ? top : h.p->dp_max;
		}
		bar += top;
	}
	bar = bar > pe_bonus ? bar - pe_bonus : 0;
	const int n = (int)ends.size();
	{ RsortScratch scratch; exact_radix_sort(ends.data(), ends.data() + n, EndKey(), scratch); } // radix_sort_pair: the unstable sort, replayed exactly
	// every closing end looks back over the opening ends of its strand, nearest first, until one is too far away
	int64_t best = -1;
	int best_at[2] = {-1, -1}, last_open[2] = {-1, -1};
	std::vector<uint64_t> found; // (dp sum << 32) + (hash sum) of every pair over the bar
	for (int i = 0; i < n; ++i) {
		const End &c = ends[i];
		if (!(c.key & 1)) { last_open[c.rev] = i; continue; }
		const int from = last_open[c.rev];
		if (from < 0) continue;
		auto in_reach = [&](const End &o) { return c.hit->rid == o.hit->rid && c.hit->rs - o.hit->re <= max_gap_ref; };
		if (!in_reach(ends[from])) continue;
		for (int j = from; j >= 0; --j) {
			const End &o = ends[j];
			if (o.rev != c.rev || o.seg == c.seg) continue;
			if (!in_reach(o)) break;
			if (pair_dp(c, o) < bar) continue;
			const int64_t score = (int64_t)pair_dp(c, o) << 32 | (c.hit->hash + o.hit->hash);
			if (score > best) best = score, best_at[o.seg] = j, best_at[c.seg] = i;
			found.push_back((uint64_t)score);
		}
	}
	if (found.size() > 1) sort_u64(found.data(), found.data() + found.size());
	if (!found.empty() && best > 0) {
		Reg *mate[2] = { ends[best_at[0]].hit, ends[best_at[1]].hit };
		for (int s = 0; s < 2; ++s) { // the pair's hits become their reads' primaries
			Reg &h = *mate[s];
			h.proper_frag = 1;
			if (h.id != h.parent) {
				Reg &old_primary = regs[s][h.parent];
				for (Reg &x : regs[s]) if (x.parent == old_primary.id) x.parent = h.id;
				old_primary.mapq = 0;
			}
			if (!h.sam_pri) {
				for (Reg &x : regs[s]) x.sam_pri = 0;
				h.sam_pri = 1;
			}
		}
		// pair-level MAPQ: the better of the two hits', capped by the lead over the runner-up pair; single precision as in the reference
		const uint64_t best_dp = (uint64_t)best >> 32;
		const bool has_runner_up = found.size() > 1;
		const uint64_t runner_up_dp = has_runner_up ? found[found.size() - 2] >> 32 : 0;
		int n_near = 0;
		for (const uint64_t v : found) if ((v >> 32) + sub_diff >= best_dp) ++n_near;
		int mapq_pair = mate[0]->mapq > mate[1]->mapq ? mate[0]->mapq : mate[1]->mapq;
		if (has_runner_up) {
			const int by_lead = (int)(6.02f * ((best >> 32) - (found[found.size() - 2] >> 32)) / match_sc - 4.343f * logf(n_near));
			mapq_pair = mapq_pair < by_lead ? mapq_pair : by_lead;
		}
		const uint32_t floor_q = !has_runner_up ? 2 : best_dp > runner_up_dp ? 1 : 0;
		for (int s = 0; s < 2; ++s) {
			Reg &h = *mate[s];
			if ((int)h.mapq < mapq_pair) h.mapq = (int)(.2f * h.mapq + .8f * mapq_pair + .499f);
			if (h.mapq < floor_q) h.mapq = floor_q;
		}
	}
	mark_read_through(qlens, regs);
}

} // namespace mm2amd

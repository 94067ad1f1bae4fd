// The rules that turn a chain into DP windows (the reference's mm_align1 up to its first ksw call: align.c:435-561 seed clean-up, :695-767 extension limits),
// written ONCE for the device and the host: region_plan_kernel (region_dev.hip) and the host path's Aligner::plan (align.cpp) both call these.  Round 6: they
// replace the two statement-for-statement restatements of the reference's blocks that sat in those files -- the formulations are ours:
//   * what the reference computes for the LEFT and the RIGHT end of a chain as two mirrored code blocks is one routine here, run on coordinates that face the end
//     in question (distances from the sequence's far end on the right side), so that "how far may the extension reach" reads the same both ways;
//   * the two long-gap filters are a sweep over an explicit "open window" / a cluster grown by a predicate instead of the reference's index juggling;
//   * the end trimming is one walk per side over (earlier, later) anchor steps.
// tests/cpucheck/region_rules_test.cpp pins every routine to the reference's own static functions (compiled from the reference's align.c by
// oracle/ref_align_shim.c) on random and adversarial chains; the extension limits, which are not a function of their own in the reference, are pinned through
// mm_map on whole reads (tests/test_host_pipeline.py, tests/test_gpu_regions.py).
#pragma once
#include <cstdint>
#include "abi_ref.hpp"
#include "exact_rsort.hpp" // MM2_HD
#include "types.hpp"

namespace mm2amd {

MM2_HD inline int32_t rr_x(const Anchor &a) { return (int32_t)a.x; }                 // last base of the seed on the reference (strand-specific coordinate)
MM2_HD inline int32_t rr_y(const Anchor &a) { return (int32_t)a.y; }                 // ... on the query
MM2_HD inline int32_t rr_span(const Anchor &a) { return (int32_t)(a.y >> 32 & 0xff); }
MM2_HD inline bool rr_same_target(const Anchor &a, const Anchor &b) { return a.x >> 32 == b.x >> 32; } // reference sequence and strand
// query advance minus reference advance between anchors i - 1 and i: > 0 an insertion's worth, < 0 a deletion's
MM2_HD inline int32_t rr_gap(const Anchor *a, int i) { return (rr_y(a[i]) - rr_y(a[i - 1])) - (int32_t)(a[i].x - a[i - 1].x); }
MM2_HD inline bool rr_is_long_gap(const Anchor *a, int i, int min_gap) { const int32_t g = rr_gap(a, i); return g < -min_gap || g > min_gap; }

// ---------------------------------------------------------------------------------------------------------
// Long-gap filter 1 (mm_filter_bad_seeds, align.c:454-489): an insertion followed closely by a deletion of about the same size (or the other way round) is
// usually two wrong seeds rather than two events -- the seeds between such a pair are not aligned from.  sites[0..n): the chain's long-gap anchors (indices into
// a, ascending; collect_long_gaps, :435-452).
// A site's COMPENSATION is the largest amount 2 min(inserted, deleted) over the gaps from it to a later site within reach (max_sites further, max_len bases on
// either sequence), with the first site that reaches it.  The sweep keeps one window open -- (amount, first site, partner) -- and lets a site inside it take over
// only with a strictly larger amount; when the sweep arrives at the open window's partner, the seeds from its first site up to the partner are flagged.
// ---------------------------------------------------------------------------------------------------------
struct RrCompensation { int amount, partner; };
MM2_HD inline RrCompensation rr_compensation(const Anchor *a, const int32_t *sites, int n, int k, int max_len, int max_sites)
{
	const Anchor &before = a[sites[k] - 1];
	int ins = 0, del = 0;
	RrCompensation best = { 0, -1 };
	auto add = [&](int32_t g) { if (g > 0) ins += g; else del -= g; };
	add(rr_gap(a, sites[k]));
	for (int l = k + 1; l < n && l - k <= max_sites; ++l) {
		const Anchor &there = a[sites[l]];
		if (rr_y(there) - rr_y(before) > max_len || rr_x(there) - rr_x(before) > max_len) break;
		add(rr_gap(a, sites[l]));
		const int cancelled = 2 * (ins < del ? ins : del);
		if (cancelled > best.amount) best.amount = cancelled, best.partner = l;
	}
	return best;
}
MM2_HD inline void rr_drop_compensating_gaps(Anchor *a, const int32_t *sites, int n, int min_amount, int max_len, int max_sites)
{
	int open_amount = 0, open_first = -1, open_partner = -1;
	auto close = [&]() {
		if (open_partner >= 0) for (int32_t i = sites[open_first]; i < sites[open_partner]; ++i) a[i].y |= ref::SEED_IGNORE;
		open_amount = 0, open_first = open_partner = -1;
	};
	for (int k = 0; k < n; ++k) {
		if (open_partner >= 0 && k >= open_partner) close();
		const RrCompensation c = rr_compensation(a, sites, n, k, max_len, max_sites);
		if (c.amount > min_amount && c.amount > open_amount) open_amount = c.amount, open_first = k, open_partner = c.partner;
	}
	close();
}

// ---------------------------------------------------------------------------------------------------------
// Long-gap filter 2 (mm_filter_bad_seeds_alt, align.c:491-525): long gaps that follow each other with less aligned sequence between them than the gaps are long
// form ONE event; the seeds inside such a cluster are not aligned from and the cluster's last gap is bridged by one long window (MM_SEED_LONG_JOIN).
// A cluster grows from its last member `at` to the next site `to` while `to` lies within max_ext of `at` on both sequences and the room between them -- from `at`
// to the END of the seed before `to`, the shorter of the two sequences' -- is no larger than the two gaps together.
// ---------------------------------------------------------------------------------------------------------
MM2_HD inline bool rr_gap_joins(const Anchor *a, int32_t at, int32_t to, int max_ext)
{
	const Anchor &here = a[at], &there = a[to], &pre = a[to - 1];
	if (rr_y(there) - rr_y(here) > max_ext || rr_x(there) - rr_x(here) > max_ext) return false;
	const int32_t room_t = rr_x(pre) + rr_span(pre) - rr_x(here), room_q = rr_y(pre) + rr_span(pre) - rr_y(here);
	const int32_t g1 = rr_gap(a, at), g2 = rr_gap(a, to);
	return (room_t < room_q ? room_t : room_q) <= (g1 < 0 ? -g1 : g1) + (g2 < 0 ? -g2 : g2);
}
MM2_HD inline void rr_join_gap_clusters(Anchor *a, const int32_t *sites, int n, int max_ext)
{
	for (int first = 0; first < n;) {
		int last = first;
		while (last + 1 < n && rr_gap_joins(a, sites[last], sites[last + 1], max_ext)) ++last;
		if (last > first) {
			for (int32_t i = sites[first]; i < sites[last]; ++i) a[i].y |= ref::SEED_IGNORE;
			a[sites[last]].y |= ref::SEED_LONG_JOIN;
		}
		first = last + 1;
	}
}

// ---------------------------------------------------------------------------------------------------------
// End trimming (mm_fix_bad_ends, align.c:527-561): the first seeds of a chain are only trusted once they are followed by enough sequence on one diagonal.  Walking
// inward from an end over the steps between consecutive seeds, a step whose two advances differ by more than half of what has been aligned so far moves the end
// to the step's inner seed; the walk stops at a long-join step, or once the aligned length reaches 2 bw, or the matching bases reach max(min_match, bw) or half
// of the chain's.  The left walk runs first; the right one stops where the left one put the start.
// ---------------------------------------------------------------------------------------------------------
struct RrTrimWalk {
	int32_t aligned, matched; // bases aligned / matching so far, the end seed's own span included
	MM2_HD explicit RrTrimWalk(const Anchor &end_seed) : aligned(rr_span(end_seed)), matched(rr_span(end_seed)) {}
	// one step between consecutive seeds; *off_diagonal: the end moves inward past it.  false: the walk ends BEFORE this step
	MM2_HD bool step(const Anchor &earlier, const Anchor &later, bool *off_diagonal)
	{
		if (later.y & ref::SEED_LONG_JOIN) return false;
		const int32_t dt = rr_x(later) - rr_x(earlier), dq = rr_y(later) - rr_y(earlier), shorter = dt < dq ? dt : dq, longer = dt < dq ? dq : dt;
		*off_diagonal = longer - shorter > aligned >> 1;
		aligned += shorter;
		matched += shorter < rr_span(later) ? shorter : rr_span(later);
		return true;
	}
	MM2_HD bool enough(int bw, int min_match, int32_t chain_mlen) const { return aligned >= bw << 1 || (matched >= min_match && matched >= bw) || matched >= chain_mlen >> 1; }
};
MM2_HD inline void rr_trim_ends(const ref::Reg1 &r, const Anchor *a, int bw, int min_match, int32_t *as, int32_t *cnt)
{
	*as = r.as, *cnt = r.cnt;
	if (r.cnt < 3) return;
	const int32_t last = r.as + r.cnt - 1;
	RrTrimWalk left(a[r.as]);
	for (int32_t i = r.as + 1; i < last; ++i) { // steps (i - 1, i)
		bool cut;
		if (!left.step(a[i - 1], a[i], &cut)) break;
		if (cut) *as = i;
		if (left.enough(bw, min_match, r.mlen)) break;
	}
	int32_t end = last;
	RrTrimWalk right(a[last]);
	for (int32_t i = last - 1; i > *as; --i) { // steps (i, i + 1)
		bool cut;
		if (!right.step(a[i], a[i + 1], &cut)) break;
		if (cut) end = i;
		if (right.enough(bw, min_match, r.mlen)) break;
	}
	*cnt = end + 1 - *as;
}

// ---------------------------------------------------------------------------------------------------------
// What a short read is aligned from (mm_max_stretch, align.c:563-589): of the chain's maximal runs of consecutive seeds on ONE diagonal, the run whose seeds
// cover the most bases (a seed adds its step, at most its span); the first of equals.
// ---------------------------------------------------------------------------------------------------------
MM2_HD inline void rr_best_diagonal_run(const ref::Reg1 &r, const Anchor *a, int32_t *as, int32_t *cnt)
{
	*as = r.as, *cnt = r.cnt;
	if (r.cnt < 2) return;
	const int32_t end = r.as + r.cnt;
	int32_t best_cover = -1;
	for (int32_t first = r.as; first < end;) {
		int32_t cover = rr_span(a[first]), next = first + 1;
		for (; next < end && rr_x(a[next]) - rr_x(a[next - 1]) == rr_y(a[next]) - rr_y(a[next - 1]); ++next) {
			const int32_t step = rr_y(a[next]) - rr_y(a[next - 1]);
			cover += step < rr_span(a[next]) ? step : rr_span(a[next]);
		}
		if (cover > best_cover) best_cover = cover, *as = first, *cnt = next - first;
		first = next;
	}
}

// ---------------------------------------------------------------------------------------------------------
// Extension limits (align.c:706-767): how far beyond the chain's first / last seed the end extensions may look.  One routine for both ends, on coordinates that
// FACE the end: for the left end a position is its distance from the sequence start (the coordinate itself), for the right end its distance from the
// sequence END, on the reference and on the query alike.  In facing coordinates "further out" is "smaller" on both sides.
//   seed_t / seed_q   : the end seed's outer edge
//   bound_t / bound_q : where the first window starts (the seed's boundary as mm_adjust_minier put it, :418-433); an end with nothing beyond it (0) is not extended
//   next(k, &t, &q)   : the outer edge of the k-th seed beyond the end seed (k = 0, 1, ...) on the same reference sequence and strand, false when there is none
// The limit starts at the end seed's own edge; the (min_cnt + 1)-th seed lying strictly further out on BOTH sequences bounds it (the extension must not run into
// another chain's territory: as far as the larger of the two distances, on both sequences); then the query may reach max_gap bases beyond the boundary, the
// reference as far as a gapped alignment of that many query bases can (rr_ext_reach), both never past what the seeds beyond allow.  clamp_to_bound (left end
// only, :723): the reference limit is not left inside the boundary.
// ---------------------------------------------------------------------------------------------------------
struct RrExtScoring { int a, q, e, max_gap, min_cnt; };
MM2_HD inline int32_t rr_ext_reach(int32_t l, const RrExtScoring &S) // reference bases a gapped extension of l query bases can span (:716-718)
{
	if (l * S.a > S.q) l += (l * S.a - S.q) / S.e;
	return l < S.max_gap ? l : S.max_gap;
}
template <class Next>
MM2_HD inline void rr_extension_limit(int32_t seed_t, int32_t seed_q, int32_t bound_t, int32_t bound_q, Next next, const RrExtScoring &S, bool clamp_to_bound, int32_t *lim_t, int32_t *lim_q)
{
	int32_t t = seed_t < 0 ? 0 : seed_t, q = seed_q; // (a seed's span can exceed its position with a homopolymer-compressed index)
	int32_t far_t = 0, far_q = 0;                    // what the seeds beyond allow: the sequence's end unless enough of them lie further out
	int n_beyond = 0;
	int32_t nt, nq;
	for (int k = 0; next(k, &nt, &nq); ++k) {
		if (!(nt < t && nq < q)) continue;
		if (++n_beyond > S.min_cnt) {
			const int32_t d = t - nt > q - nq ? t - nt : q - nq;
			far_t = t - d, far_q = q - d;
			break;
		}
	}
	if (bound_t > 0 && bound_q > 0) {
		int32_t l = bound_q < S.max_gap ? bound_q : S.max_gap;
		if (far_q < bound_q - l) far_q = bound_q - l;
		if (q > far_q) q = far_q;
		l = rr_ext_reach(l, S);
		if (l > bound_t) l = bound_t;
		if (far_t < bound_t - l) far_t = bound_t - l;
		if (t > far_t) t = far_t;
		if (clamp_to_bound && t > bound_t) t = bound_t;
	} else t = bound_t, q = bound_q;
	*lim_t = t, *lim_q = q;
}
// a hit that overlaps itself (MM_SEED_SELF, :760-767) is not extended across its own diagonal: no further beyond the hit's edge than `room`, the offset between the
// hit's two coordinates at that end (lim and hit_edge face the end; the caller takes room from the coordinates as they are)
MM2_HD inline int32_t rr_self_limit(int32_t lim, int32_t hit_edge, int32_t room) { return hit_edge - lim > room ? hit_edge - room : lim; }
// short reads (:696-704): the whole read is aligned, and as much reference beyond the chain as its l unaligned end bases could span with gaps, the end bonus counted
MM2_HD inline int32_t rr_sr_reach(int32_t l, int a, int q, int e, int end_bonus) { return l * a + end_bonus > q ? l + (l * a + end_bonus - q) / e : l; }

} // namespace mm2amd

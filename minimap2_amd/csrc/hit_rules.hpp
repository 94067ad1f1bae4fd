// The hit-record rules between chaining and alignment and after it (the reference's hit.c / esterr.c), written once for the device and the host:
// chain_regs_kernel (region_dev.hip) runs them on records in LDS, hits.cpp on the host path's vectors.  Round 5: these replace round 1's
// function-by-function restatements in hits.cpp -- the formulations are the device kernel's (an owner search instead of the reference's goto
// ladder, keep flags then one compaction, the minimizer walk of mm_est_err as one binary search per anchor).
#pragma once
#include <cstdint>
#include "abi_ref.hpp"
#include "exact_rsort.hpp" // MM2_HD
#include "types.hpp"

namespace mm2amd {

MM2_HD inline int hr_span(const Anchor &a) { return (int)(a.y >> 32 & 0xff); }

MM2_HD inline uint64_t hr_mix64(uint64_t key) // hash64 of hit.c:40-50 (the invertible integer hash without a mask)
{
	key = (~key + (key << 21));
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8));
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4));
	key = key ^ key >> 28;
	key = (key + (key << 31));
	return key;
}
// sort key of a chain among its read's chains (mm_gen_regs, hit.c:60-66): the chain record (score << 32 | anchors) with its low word scrambled by
// a hash of the chain's first anchor and the read's own hash -- equal scores come in an order that differs from read to read
MM2_HD inline uint64_t hr_chain_key(uint64_t u, const Anchor &first, uint32_t read_hash) { return u ^ (uint32_t)hr_mix64((hr_mix64(first.x) + hr_mix64(first.y)) ^ read_hash); }

// a new hit record for the chain of `cnt` anchors that starts at anchor `as` (hit.c:68-86 with mm_reg_set_coor, :24-38); fuzzy lengths not yet set
MM2_HD inline void hr_new_hit(ref::Reg1 &r, int id, uint64_t key, int32_t as, int32_t cnt, int32_t qlen, const Anchor *a, bool is_qstrand)
{
	__builtin_memset(&r, 0, sizeof r);
	r.id = id, r.parent = ref::PARENT_UNSET;
	r.score = r.score0 = (int32_t)(key >> 32);
	r.hash = (uint32_t)key;
	r.cnt = cnt, r.as = as;
	r.div = -1.0f;
	const Anchor f = a[as], l = a[as + cnt - 1];
	const int span = hr_span(f);
	r.rev = f.x >> 63;
	r.rid = (int32_t)(f.x << 1 >> 33);
	r.rs = (int32_t)f.x + 1 > span ? (int32_t)f.x + 1 - span : 0;
	r.re = (int32_t)l.x + 1;
	if (!r.rev || is_qstrand) r.qs = (int32_t)f.y + 1 - span, r.qe = (int32_t)l.y + 1;
	else r.qs = qlen - ((int32_t)l.y + 1), r.qe = qlen - ((int32_t)f.y + 1 - span);
}
// what one pair of consecutive anchors adds to a chain's fuzzy block / match lengths (mm_cal_fuzzy_len, hit.c:5-22); the first anchor adds its span to both
MM2_HD inline void hr_fuzzy_step(const Anchor &cur, const Anchor &prev, int *bl, int *ml)
{
	const int span = hr_span(cur), tl = (int32_t)cur.x - (int32_t)prev.x, ql = (int32_t)cur.y - (int32_t)prev.y;
	*bl += tl > ql ? tl : ql;
	*ml += tl > span && ql > span ? span : tl < ql ? tl : ql;
}

// a hit on an ALT contig competes with a handicap (mm_alt_score, hit.c:99-104)
MM2_HD inline int hr_alt_score(int score, float alt_diff_frac)
{
	if (score < 0) return score;
	score = (int)(score * (1.0 - alt_diff_frac) + .499);
	return score > 0 ? score : 1;
}

// Which hits are secondary to which (mm_set_parent, hit.c:125-186).  Hit i (in score order) is tested against the hits that are primary so
// far: first the part of its query interval that no primary covers (the primaries' intervals, clipped to the hit, kept as a sorted list), then the
// first primary it overlaps by more than mask_level of the shorter of the two -- its owner.  Without an owner the hit is a new primary.
// cov, prim: scratch of n entries each.
MM2_HD inline void hr_mark_parents(ref::Reg1 *r, int n, uint64_t *cov, int32_t *prim, float mask_level, int mask_len, int sub_diff, bool hard_mask_level, float alt_diff_frac)
{
	if (n <= 0) return;
	for (int i = 0; i < n; ++i) r[i].id = i;
	int n_prim = 1;
	prim[0] = 0, r[0].parent = 0;
	for (int i = 1; i < n; ++i) {
		const int si = r[i].qs, ei = r[i].qe, len_i = ei - si;
		int uncov = 0;
		bool touches = hard_mask_level; // (with a hard mask level the uncovered length stays 0 and every primary is tested)
		if (!hard_mask_level) {
			int n_cov = 0;
			for (int j = 0; j < n_prim; ++j) {
				int sj = r[prim[j]].qs, ej = r[prim[j]].qe;
				if (ej <= si || sj >= ei) continue;
				sj = sj < si ? si : sj, ej = ej > ei ? ei : ej;
				const uint64_t v = (uint64_t)sj << 32 | (uint32_t)ej; // (the reference sorts these with radix_sort_64: plain integers, any sort gives the same list)
				int p = n_cov++;
				while (p > 0 && cov[p - 1] > v) cov[p] = cov[p - 1], --p;
				cov[p] = v;
			}
			if (n_cov > 0) {
				touches = true;
				int x = si;
				for (int c = 0; c < n_cov; ++c) {
					const int cs = (int)(cov[c] >> 32), ce = (int32_t)cov[c];
					if (cs > x) uncov += cs - x;
					x = ce > x ? ce : x;
				}
				if (ei > x) uncov += ei - x;
			}
		}
		int owner = -1, ol = 0, mn = 0;
		if (touches)
			for (int j = 0; j < n_prim && owner < 0; ++j) {
				const ref::Reg1 &rp = r[prim[j]];
				const int sj = rp.qs, ej = rp.qe, len_j = ej - sj;
				if (ej <= si || sj >= ei) continue;
				mn = len_j < len_i ? len_j : len_i;
				const int mx = len_j > len_i ? len_j : len_i, lo = si > sj ? si : sj, hi = ei < ej ? ei : ej;
				ol = hi > lo ? hi - lo : 0;
				if ((float)ol / mn - (float)uncov / mx > mask_level && uncov <= mask_len) owner = j;
			}
		if (owner < 0) { prim[n_prim++] = i, r[i].parent = i, r[i].n_sub = 0; continue; }
		ref::Reg1 &rp = r[prim[owner]], &ri = r[i];
		const bool handicap = !rp.is_alt && ri.is_alt;
		int sci = handicap ? hr_alt_score(ri.score, alt_diff_frac) : ri.score;
		bool counts = ri.cnt >= rp.cnt;
		ri.parent = rp.parent;
		rp.subsc = rp.subsc > sci ? rp.subsc : sci;
		if (rp.p && ri.p && (rp.rid != ri.rid || rp.rs != ri.rs || rp.re != ri.re || ol != mn)) { // both aligned, and not the same alignment found twice
			sci = handicap ? hr_alt_score(ri.p->dp_max, alt_diff_frac) : ri.p->dp_max;
			rp.p->dp_max2 = rp.p->dp_max2 > sci ? rp.p->dp_max2 : sci;
			if (rp.p->dp_max - ri.p->dp_max <= sub_diff) counts = true;
		}
		if (counts) ++rp.n_sub;
	}
}

// which hit is reported as the primary SAM record (mm_set_sam_pri, hit.c:220-229); returns the number of primaries
MM2_HD inline int hr_mark_sam_primary(ref::Reg1 *r, int n)
{
	int n_pri = 0;
	for (int i = 0; i < n; ++i) {
		if (r[i].id == r[i].parent) { ++n_pri; r[i].sam_pri = n_pri == 1; }
		else r[i].sam_pri = 0;
	}
	return n_pri;
}

// ids follow positions after a compaction; parents that were dropped leave orphans (mm_sync_regs, hit.c:231-253).  where: scratch of where_n
// entries, where_n > the largest id in use.
MM2_HD inline void hr_renumber(ref::Reg1 *r, int n, int32_t *where, int where_n)
{
	for (int i = 0; i < where_n; ++i) where[i] = -1;
	for (int i = 0; i < n; ++i) if (r[i].id >= 0) where[r[i].id] = i;
	for (int i = 0; i < n; ++i) {
		ref::Reg1 &x = r[i];
		x.id = i;
		if (x.parent == ref::PARENT_TMP_PRI) x.parent = i;
		else if (x.parent >= 0 && where[x.parent] >= 0) x.parent = where[x.parent];
		else x.parent = ref::PARENT_UNSET;
	}
	hr_mark_sam_primary(r, n);
}

// Which secondaries are kept (mm_select_sub, hit.c:255-281): keep[i] for every hit; the caller drops the others (freeing what they own),
// compacts and renumbers.  min_diff = 2 k, min_strand_sc = 0.8 max_gap at the call sites (map.c:210, :219).
MM2_HD inline void hr_select_secondaries(ref::Reg1 *r, int n, uint8_t *keep, float pri_ratio, int min_diff, int best_n, bool check_strand, int min_strand_sc)
{
	int n_2nd = 0;
	for (int i = 0; i < n; ++i) {
		const int p = r[i].parent;
		bool k = false;
		if (p == i || r[i].inv) k = true;
		else if ((r[i].score >= r[p].score * pri_ratio || r[i].score + min_diff >= r[p].score) && n_2nd < best_n) {
			const bool same = r[i].qs == r[p].qs && r[i].qe == r[p].qe && r[i].rid == r[p].rid && r[i].rs == r[p].rs && r[i].re == r[p].re;
			if (!same) k = true, ++n_2nd; // (an identical hit found twice is dropped)
		} else if (check_strand && n_2nd < best_n && r[i].score > min_strand_sc && r[i].rev != r[p].rev) {
			r[i].strand_retained = 1;
			k = true, ++n_2nd;
		}
		keep[i] = k ? 1 : 0;
	}
}

// Which secondary chains of a FRAGMENT are kept (mm_select_sub_multi, pe.c:6-50): keep[i] for every hit.  A secondary within min_diff of its parent always stays;
// otherwise it has to reach a share of the parent's score that depends on where it lies: next to the parent on the reference (a pair's other placement) pri1, a chain
// confined to one read against a parent that spans both pri2, else the usual ratio.  At most best_n secondaries (a good one beyond the cap still counts).
MM2_HD inline void hr_select_secondaries_multi(const ref::Reg1 *r, int n, uint8_t *keep, float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs, int qlen0, int qlen1)
{
	const int reach = n_segs == 2 ? qlen0 + qlen1 + max_gap_ref : 0;
	int n_secondary = 0;
	for (int i = 0; i < n; ++i) {
		const ref::Reg1 &h = r[i];
		if (h.parent == i) { keep[i] = 1; continue; }
		const ref::Reg1 &par = r[h.parent];
		const bool sec_both = n_segs == 2 && h.qs < qlen0 && h.qe > qlen0, par_both = n_segs == 2 && par.qs < qlen0 && par.qe > qlen0;
		const bool beside = par.rev == h.rev && par.rid == h.rid && h.re - par.rs < reach && par.re - h.rs < reach;
		const float share = beside ? pri1 : (sec_both || sec_both == par_both) ? pri_ratio : pri2;
		const bool good = h.score + min_diff >= par.score || h.score >= par.score * share;
		keep[i] = good && n_secondary++ < best_n ? 1 : 0;
	}
}

// Which segment of its fragment an anchor lies on, and the anchor as its segment sees it (mm_seg_gen, hit.c:342-396): the query coordinate counted from the segment's
// own start -- from its end for reverse-strand anchors, whose coordinate runs backwards over the concatenation.  before: bases of the fragment before the segment.
MM2_HD inline int hr_anchor_seg(const Anchor &x) { return (int)((x.y & ref::SEED_SEG_MASK) >> ref::SEED_SEG_SHIFT); }
MM2_HD inline Anchor hr_seg_anchor(Anchor x, int total, int before, int seg_len)
{
	x.y -= x.x >> 63 ? (uint64_t)(total - (seg_len + before)) : (uint64_t)before;
	return x;
}

// position of an anchor's k-mer on the read as given (get_for_qpos, esterr.c:7-14)
MM2_HD inline int32_t hr_fwd_qpos(int32_t qlen, const Anchor &a)
{
	int32_t x = (int32_t)a.y;
	if (a.x >> 63) x = qlen - 1 - (x + 1 - hr_span(a));
	return x;
}

// mm_est_err's counts for one hit (esterr.c:41-60): of the read's minimizers between the hit's first and last anchor, how many are anchors of the
// hit.  The reference walks the minimizer positions and the chain together; both ascend strictly, so "anchor k is found after anchor k - 1" is
// one search per anchor, and the walk ends at the first anchor that is not found.  *n_match, *n_tot; n_tot < 0: no estimate (div stays -1).
// k_from / k_step: the anchors this caller takes (a wavefront's lanes stride them and combine the first misses; the host takes all).
MM2_HD inline int hr_first_minimizer(const uint64_t *mp, int n_mp, int32_t x0) // the reference's own search (esterr.c:44-52)
{
	int32_t L = 0, R = n_mp - 1;
	while (L <= R) {
		const int32_t mid = (int32_t)(((uint64_t)L + R) >> 1), y = (int32_t)mp[mid];
		if (y < x0) L = mid + 1;
		else if (y > x0) R = mid - 1;
		else return mid;
	}
	return -1;
}
MM2_HD inline int hr_find_after(const uint64_t *mp, int n_mp, int st, int32_t x) // the first entry after st at position x; -1: none
{
	int lo = st + 1, hi = n_mp;
	while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int32_t)mp[mid] < x) lo = mid + 1; else hi = mid; }
	return lo < n_mp && (int32_t)mp[lo] == x ? lo : -1;
}
MM2_HD inline int32_t hr_chain_qpos(const ref::Reg1 &r, const Anchor *a, int32_t qlen, int k) { return hr_fwd_qpos(qlen, r.rev ? a[r.as + r.cnt - 1 - k] : a[r.as + k]); }
// the first anchor of [k_from, cnt) in steps of k_step that is not among the minimizers after st (cnt: all are)
MM2_HD inline int hr_first_miss(const ref::Reg1 &r, const Anchor *a, int32_t qlen, const uint64_t *mp, int n_mp, int st, int k_from, int k_step)
{
	for (int k = k_from; k < r.cnt; k += k_step) {
		const int32_t x = hr_chain_qpos(r, a, qlen, k);
		if (x <= hr_chain_qpos(r, a, qlen, k - 1) || hr_find_after(mp, n_mp, st, x) < 0) return k; // (found AFTER anchor k - 1: the positions ascend)
	}
	return r.cnt;
}
MM2_HD inline void hr_est_err_totals(const ref::Reg1 &r, const Anchor *a, int32_t qlen, const uint64_t *mp, int n_mp, int st, int kfail, float avg_k, int32_t l_ref, int32_t *n_match, int32_t *n_tot)
{
	*n_match = kfail; // the first anchor and anchors 1 .. kfail - 1
	const int en = kfail > 1 ? hr_find_after(mp, n_mp, st, hr_chain_qpos(r, a, qlen, kfail - 1)) : st;
	int32_t t = en - st + 1;
	if (r.qs > avg_k && r.rs > avg_k) ++t;
	if (qlen - r.qs > avg_k && l_ref - r.re > avg_k) ++t;
	*n_tot = t;
}

} // namespace mm2amd

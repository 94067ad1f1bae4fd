// chain_regs_kernel / region_plan_kernel / region_consume_kernel: see region_dev.hpp.
//
// These stages are short, branchy and integer: a 10 kb read has one or two chains of ~700 anchors and ~40 DP windows.  They are bound by the
// latency of dependent loads, not by bandwidth or issue -- what matters is that tens of thousands of reads / regions are in flight at once and
// that nothing crosses PCIe.  Algorithmic bytes: 16 B per chained anchor read twice (coordinates + fuzzy lengths, window walk) and written once
// (squeeze), 80 B per hit record, 48 + 24 B written per DP window, 68 B read per DP result.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "region_dev.hpp"
#include "region_rules.hpp"
#include "hit_rules.hpp"

namespace mm2amd {

namespace {

using ref::Reg1;

__device__ __forceinline__ int rg_span(const Anchor &a) { return (int)(a.y >> 32 & 0xff); }
__device__ __forceinline__ int32_t rg_x(const Anchor &a) { return (int32_t)a.x; }
__device__ __forceinline__ int32_t rg_y(const Anchor &a) { return (int32_t)a.y; }
__device__ __forceinline__ uint32_t rg_roundup32(uint32_t x) { --x; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; return ++x; }

#define RG_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int rg_wave_sum(int v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
	return v;
}
__device__ __forceinline__ int rg_wave_min(int v)
{
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
	return v;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------------------------
// chain_regs_kernel
// ---------------------------------------------------------------------------------------------------------------------------------------
// LDS per read (C = lds_chains): hit records 80 C, sort keys 8 C, interval list 8 C, five 32-bit arrays, the order and the keep flags: 119 C bytes;
// with two-segment fragments in the launch another 80 C for a segment's hit records
__global__ void __launch_bounds__(64) chain_regs_kernel(RgnBuffers B, RgnOpts O)
{
	MM2_DYN_LDS(uint64_t, s_raw);
	const int C = B.lds_chains, lane = threadIdx.x, rd_i = blockIdx.x;
	Reg1 *s_reg = (Reg1 *)s_raw;                       // C hit records (80 B each: a multiple of 8)
	uint64_t *s_key = s_raw + (size_t)C * 10;          // C sort keys, later (as << 32 | index) of the squeeze
	uint64_t *s_cov = s_key + C;                       // C clipped intervals
	int32_t *s_start = (int32_t *)(s_cov + C);         // C chain starts (in the read's chained anchors), later the squeezed starts
	int32_t *s_nm = s_start + C, *s_nt = s_nm + C;     // C + C: mm_est_err's counts
	int32_t *s_prim = s_nt + C, *s_where = s_prim + C; // C + C: the primaries so far; the id map of a compaction
	int16_t *s_ord = (int16_t *)(s_where + C);         // C: chain at each rank
	uint8_t *s_keep = (uint8_t *)(s_ord + C);          // C
	Reg1 *s_seg = (Reg1 *)(s_raw + ((size_t)C * 119 + 7) / 8); // C hit records of one segment (two-segment fragments only)
	__shared__ int32_t s_hdr[4];

	const RgnRead rd = B.reads[rd_i];
	RgnReadOut *const rout = B.rout + (size_t)rd_i * (size_t)B.rout_stride;
	RgnReadOut out;
	out.reg0 = 0, out.n_regs = 0, out.n_a_sq = 0, out.flags = 0, out.avg_k = 0.0f, out.pad = 0;
	const int n = rd.n_u;
	const bool is_sr = (O.flag & ref::F_SR) != 0, paired = rd.qlen2 > 0;
	if (paired && lane == 0) rout[0] = out, rout[1] = out;
	if ((rd.src & RGN_SRC_SKIP) || n > C || n == 0) {
		if (rd.src & RGN_SRC_SKIP) out.flags = RGN_F_SKIPPED;
		else if (n > C) out.flags = RGN_F_MANY_CHAINS;
		if (lane == 0) rout[0] = out;
		return;
	}
	const Anchor *a = B.a_src[rd.src & RGN_SRC_LJ] + rd.a_off;
	const uint64_t *u = B.u_src[rd.src & RGN_SRC_LJ] + rd.u_off;
	const int qlen = rd.qlen + rd.qlen2; // (a fragment's chains were found on the concatenation of its segments)

	// chain starts: the chains' anchors lie back to back in chain order
	if (lane == 0) { int acc = 0; for (int i = 0; i < n; ++i) { s_start[i] = acc; acc += (int32_t)u[i]; } }
	RG_SYNC();
	// sort key of mm_gen_regs (hit.c:60-66): the chain record with its low word scrambled by a hash of the first anchor and the read
	for (int i = lane; i < n; i += 64) {
		s_key[i] = hr_chain_key(u[i], a[s_start[i]], rd.hash);
	}
	RG_SYNC();
	// descending order by counting; equal keys (2^-32 per pair of equal scores) are ordered by the reference's unstable sort: the host replays it
	int tie = 0;
	for (int i = lane; i < n; i += 64) {
		const uint64_t me = s_key[i];
		int rank = 0;
		for (int j = 0; j < n; ++j) { const uint64_t o = s_key[j]; rank += o > me; tie |= (o == me && j != i); }
		s_ord[rank] = (int16_t)i;
	}
	if (__ballot(tie)) {
		out.flags = RGN_F_SORT_TIE;
		if (lane == 0) rout[0] = out;
		return;
	}
	RG_SYNC();
	// the hit records (hit.c:68-86 with mm_reg_set_coor, :24-38)
	for (int p = lane; p < n; p += 64) {
		const int c = s_ord[p];
		Reg1 r;
		hr_new_hit(r, p, s_key[c], s_start[c], (int32_t)u[c], qlen, a, false);
		s_reg[p] = r;
	}
	RG_SYNC();
	// fuzzy match / block lengths (mm_cal_fuzzy_len, hit.c:5-22): a sum over consecutive anchor pairs -- all lanes, hit after hit
	for (int p = 0; p < n; ++p) {
		const int st = s_reg[p].as, cnt = s_reg[p].cnt;
		int ml = 0, bl = 0;
		for (int i = 1 + lane; i < cnt; i += 64) hr_fuzzy_step(a[st + i], a[st + i - 1], &bl, &ml);
		ml = rg_wave_sum(ml), bl = rg_wave_sum(bl);
		if (lane == 0) { const int s0 = rg_span(a[st]); s_reg[p].mlen = ml + s0, s_reg[p].blen = bl + s0; }
	}
	RG_SYNC();
	// parents, secondaries (chain_post, map.c:206-213)
	if (lane == 0) {
		int m = n;
		if (!(O.flag & ref::F_ALL_CHAINS)) {
			hr_mark_parents(s_reg, n, s_cov, s_prim, O.mask_level, O.mask_len, O.sub_diff, (O.flag & ref::F_HARD_MLEVEL) != 0, 0.0f);
			if (O.pri_ratio > 0.0f) {
				if (paired) hr_select_secondaries_multi(s_reg, n, s_keep, O.pri_ratio, 0.2f, 0.7f, rd.gap_ref, O.k * 2, O.best_n, 2, rd.qlen, rd.qlen2); // map.c:211
				else hr_select_secondaries(s_reg, n, s_keep, O.pri_ratio, O.k * 2, O.best_n, true, O.min_strand_sc);
				m = 0;
				for (int i = 0; i < n; ++i) if (s_keep[i]) { if (m < i) s_reg[m] = s_reg[i]; ++m; }
				if (m != n) hr_renumber(s_reg, m, s_where, n);
			}
		}
		int retained = 0;
		for (int p = 0; p < m; ++p) retained |= s_reg[p].strand_retained;
		s_hdr[0] = m, s_hdr[1] = retained;
	}
	RG_SYNC();
	const int m = s_hdr[0];
	if (s_hdr[1] && !is_sr) { // mm_filter_strand_retained (hit.c:283-299) compares divergences: libm's pow decides, on the host (short reads skip the filter, map.c:333)
		out.flags = RGN_F_STRAND_RETAINED;
		if (lane == 0) rout[0] = out;
		return;
	}
	if (paired) {
		// ---- two segments (mm_seg_gen, hit.c:342-396): every segment gets the chains that have anchors on it -- with the fragment chain's score and its own anchor
		// count -- and those anchors in its own coordinates, chain by chain; its hit records are made and ranked like a read's, parents marked again (map.c:346) ----
		for (int sg = 0; sg < 2; ++sg) {
			const int before = sg ? rd.qlen : 0, seg_len = sg ? rd.qlen2 : rd.qlen;
			Anchor *sq = B.sq_a + rd.sq_off + (sg ? (uint64_t)rd.n_a : 0ull);
			int acc = 0;
			for (int c = 0; c < m; ++c) {
				const int st = s_reg[c].as, cnt = s_reg[c].cnt;
				int n_on = 0;
				for (int i0 = 0; i0 < cnt; i0 += 64) {
					const int i = i0 + lane;
					bool on = false;
					Anchor x; x.x = x.y = 0;
					if (i < cnt) { x = a[st + i]; on = hr_anchor_seg(x) == sg; }
					const unsigned long long mk = __ballot(on);
					if (on) sq[acc + n_on + __popcll(mk & ((1ull << lane) - 1ull))] = hr_seg_anchor(x, qlen, before, seg_len);
					n_on += __popcll(mk);
				}
				if (lane == 0) s_nm[c] = n_on, s_nt[c] = acc;
				acc += n_on;
			}
			RG_SYNC();
			if (lane == 0) { int ns = 0; for (int c = 0; c < m; ++c) if (s_nm[c] > 0) s_where[ns++] = c; s_hdr[0] = ns; }
			RG_SYNC();
			const int ns = s_hdr[0];
			for (int i = lane; i < ns; i += 64) {
				const int c = s_where[i];
				s_key[i] = hr_chain_key((uint64_t)s_reg[c].score << 32 | (uint32_t)s_nm[c], sq[s_nt[c]], rd.hash);
			}
			RG_SYNC();
			int tie2 = 0;
			for (int i = lane; i < ns; i += 64) {
				const uint64_t me = s_key[i];
				int rank = 0;
				for (int j = 0; j < ns; ++j) { const uint64_t o = s_key[j]; rank += o > me; tie2 |= (o == me && j != i); }
				s_ord[rank] = (int16_t)i;
			}
			if (__ballot(tie2)) { // (segment 0's records may be out already: the flag in the read's first entry makes the host take the whole fragment)
				if (lane == 0) atomicOr(&rout[0].flags, (unsigned)RGN_F_SORT_TIE);
				return;
			}
			RG_SYNC();
			for (int p = lane; p < ns; p += 64) {
				const int i = s_ord[p], c = s_where[i];
				Reg1 r;
				hr_new_hit(r, p, s_key[i], s_nt[c], s_nm[c], seg_len, sq, false);
				r.seg_split = 1, r.seg_id = (uint32_t)sg;
				s_seg[p] = r;
			}
			RG_SYNC();
			for (int p = 0; p < ns; ++p) {
				const int st = s_seg[p].as, cnt = s_seg[p].cnt;
				int ml = 0, bl = 0;
				for (int i = 1 + lane; i < cnt; i += 64) hr_fuzzy_step(sq[st + i], sq[st + i - 1], &bl, &ml);
				ml = rg_wave_sum(ml), bl = rg_wave_sum(bl);
				if (lane == 0) { const int s0 = rg_span(sq[st]); s_seg[p].mlen = ml + s0, s_seg[p].blen = bl + s0; }
			}
			RG_SYNC();
			if (lane == 0) {
				if (!(O.flag & ref::F_ALL_CHAINS)) hr_mark_parents(s_seg, ns, s_cov, s_prim, O.mask_level, O.mask_len, O.sub_diff, (O.flag & ref::F_HARD_MLEVEL) != 0, 0.0f);
				s_hdr[3] = (int32_t)atomicAdd(&B.cursors[RGN_CUR_REGS], (unsigned)ns);
			}
			RG_SYNC();
			const uint32_t reg0 = (uint32_t)s_hdr[3];
			RgnReadOut so = out;
			if (reg0 + (uint32_t)ns <= B.max_regs) {
				uint32_t *dst = (uint32_t *)(B.regs + reg0);
				const uint32_t *srcw = (const uint32_t *)s_seg;
				for (int w = lane; w < ns * 20; w += 64) dst[w] = srcw[w];
				for (int p = lane; p < ns; p += 64) {
					RgnAux x; x.n_match = 0, x.n_tot = -1;
					B.aux[reg0 + p] = x;
					RgnPlan pl;
					__builtin_memset(&pl, 0, sizeof pl);
					pl.read = (uint32_t)rd_i, pl.status = -1, pl.seg = sg;
					B.plan[reg0 + p] = pl;
				}
			} else if (lane == 0) atomicOr(&rout[0].flags, (unsigned)RGN_F_MANY_CHAINS);
			so.reg0 = reg0, so.n_regs = ns, so.n_a_sq = acc;
			if (lane == 0) rout[sg].reg0 = so.reg0, rout[sg].n_regs = so.n_regs, rout[sg].n_a_sq = so.n_a_sq; // (the flags of entry 0 may have been raised already: fields one by one)
			RG_SYNC();
		}
		return;
	}
	// mm_est_err's counts (esterr.c:30-64): how many of the minimizers between a hit's first and last anchor are anchors of the hit.  The
	// reference walks the read's minimizer positions and the chain together; both ascend strictly, so "anchor k is found after anchor k - 1"
	// is one binary search per anchor, and the walk stops at the first anchor that is not found.
	const uint64_t *mp = B.mini_pos + rd.mp_off;
	const int n_mp = is_sr ? 0 : rd.n_mp; // (short reads: no divergence estimate, map.c:333)
	// (esterr.c:37-40: the mean minimizer span -- k for every one of them without HPC; with it the spans are summed, by all lanes)
	uint64_t sum_k = (uint64_t)n_mp * (uint64_t)O.k;
	if (O.hpc) {
		int part = 0;
		for (int i = lane; i < n_mp; i += 64) part += (int)(mp[i] >> 32 & 0xff);
		sum_k = (uint64_t)(uint32_t)rg_wave_sum(part);
	}
	const float avg_k = n_mp > 0 ? (float)sum_k / n_mp : 0.0f;
	for (int p = 0; p < m; ++p) {
		const Reg1 r = s_reg[p];
		int n_match = 0, n_tot = -1;
		if (n_mp > 0 && r.cnt > 0) {
			const int st = hr_first_minimizer(mp, n_mp, hr_chain_qpos(r, a, qlen, 0));
			if (st >= 0) {
				const int kfail = rg_wave_min(hr_first_miss(r, a, qlen, mp, n_mp, st, 1 + lane, 64));
				hr_est_err_totals(r, a, qlen, mp, n_mp, st, kfail, avg_k, (int32_t)B.ref_len[r.rid], &n_match, &n_tot);
			}
		}
		if (lane == 0) s_nm[p] = n_match, s_nt[p] = n_tot;
	}
	RG_SYNC();
	// the surviving hits' anchors squeezed together, in anchor order (mm_squeeze_a, hit.c:322-340)
	if (lane == 0) {
		for (int p = 0; p < m; ++p) { // insertion sort of (as, index): distinct starts
			const uint64_t v = (uint64_t)s_reg[p].as << 32 | (uint32_t)p;
			int q = p;
			while (q > 0 && s_key[q - 1] > v) s_key[q] = s_key[q - 1], --q;
			s_key[q] = v;
		}
		int acc = 0;
		for (int q = 0; q < m; ++q) { const int p = (int32_t)(uint32_t)s_key[q]; s_start[p] = acc; acc += s_reg[p].cnt; }
		s_hdr[2] = acc;
		s_hdr[3] = (int32_t)atomicAdd(&B.cursors[RGN_CUR_REGS], (unsigned)m);
	}
	RG_SYNC();
	const uint32_t reg0 = (uint32_t)s_hdr[3];
	Anchor *sq = B.sq_a + rd.sq_off;
	for (int p = 0; p < m; ++p) {
		const int src = s_reg[p].as, dst = s_start[p], cnt = s_reg[p].cnt;
		for (int i = lane; i < cnt; i += 64) sq[dst + i] = a[src + i];
	}
	RG_SYNC();
	if (reg0 + (uint32_t)m <= B.max_regs) {
		if (lane == 0) for (int p = 0; p < m; ++p) s_reg[p].as = s_start[p];
		RG_SYNC();
		uint32_t *dst = (uint32_t *)(B.regs + reg0);
		const uint32_t *srcw = (const uint32_t *)s_reg;
		for (int w = lane; w < m * 20; w += 64) dst[w] = srcw[w];
		for (int p = lane; p < m; p += 64) {
			RgnAux x; x.n_match = s_nm[p], x.n_tot = s_nt[p];
			B.aux[reg0 + p] = x;
			RgnPlan pl;
			__builtin_memset(&pl, 0, sizeof pl);
			pl.read = (uint32_t)rd_i, pl.status = -1;
			B.plan[reg0 + p] = pl;
		}
	} else out.flags = RGN_F_MANY_CHAINS; // (cannot happen: the host sizes the arrays by the chain count)
	out.reg0 = reg0, out.n_regs = m, out.n_a_sq = s_hdr[2], out.avg_k = avg_k;
	if (lane == 0) rout[0] = out;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// region_plan_kernel: one region per thread
// ---------------------------------------------------------------------------------------------------------------------------------------
namespace {

// the anchors before which the two sequences drift apart by more than min_gap (collect_long_gaps, align.c:435-452); none when there is only one.
// All lanes: 64 anchors per step, the sites compacted in order by a ballot.  Every lane returns the count.
__device__ int rg_gap_sites(const Anchor *a, int cnt1, int min_gap, int32_t *K, int lane)
{
	int n = 0;
	for (int i0 = 1; i0 < cnt1; i0 += 64) {
		const int i = i0 + lane;
		bool site = false;
		if (i < cnt1) site = rr_is_long_gap(a, i, min_gap);
		const unsigned long long m = __ballot(site);
		if (site) K[n + __popcll(m & ((1ull << lane) - 1ull))] = i;
		n += __popcll(m);
	}
	return n <= 1 ? 0 : n;
}

// (the two long-gap filters, the end trimming and the extension limits: region_rules.hpp -- one definition for this kernel and for align.cpp)

struct RgnJobCtx { uint64_t q_fwd, q_rev, t_base; int rev, gen_flag; };

// a region's read as the window rules see it: the read itself, or one segment of a fragment (its own length, query block and squeezed anchors)
struct RgnUnit { int32_t qlen, n_a; uint64_t qpool_fwd, sq_off; };
__device__ __forceinline__ RgnUnit rg_unit(const RgnBuffers &B, const RgnRead &rd, uint32_t read, int seg)
{
	RgnUnit u;
	u.qlen = seg ? rd.qlen2 : rd.qlen;
	u.qpool_fwd = rd.qpool_fwd + (seg ? 2ull * (uint64_t)rd.qlen : 0ull);
	u.sq_off = rd.sq_off + (seg ? (uint64_t)rd.n_a : 0ull);
	u.n_a = B.rout[(size_t)read * (size_t)B.rout_stride + (size_t)seg].n_a_sq;
	return u;
}

// A short read's one gap window lies on one diagonal (align.c:823-833): the score of its ungapped alignment, and the largest drop of the running score below its
// running maximum -- what mm_test_zdrop (align.c:59-84) finds on a CIGAR of one M.  All lanes, 64 columns per step; every lane returns both.
__device__ void rg_ungapped(const RgnBuffers &B, const RgnOpts &O, const RgnJobCtx &X, int32_t qs, int32_t rs, int32_t len, int lane, int32_t *score, int32_t *max_drop)
{
	const uint8_t *q = B.qpool + (X.rev ? X.q_rev : X.q_fwd) + (uint64_t)qs;
	const int amb = O.sc_ambi > 0 ? -O.sc_ambi : O.sc_ambi;
	int32_t sum = 0, run = 0, run_max = INT32_MIN, drop = 0;
	for (int32_t j0 = 0; j0 < len; j0 += 64) {
		const int32_t j = j0 + lane;
		int s1 = 0, s2 = 0;
		if (j < len) {
			const int cq = q[j];
			const uint64_t o = X.t_base + (uint64_t)(rs + j);
			const int ct = (int)(B.S[o >> 3] >> ((o & 7) << 2) & 0xf);
			s1 = cq >= 4 || ct >= 4 ? amb : cq == ct ? O.a : -O.b;
			s2 = O.mat[(ct > 4 ? 4 : ct) * 5 + (cq > 4 ? 4 : cq)];
		}
		sum += s1;
		int pre = s2;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(pre, d, 64); if (lane >= d) pre += t; }
		const int sc = run + pre;
		int mx = j < len ? sc : INT32_MIN;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(mx, d, 64); if (lane >= d) mx = t > mx ? t : mx; }
		mx = mx > run_max ? mx : run_max;
		if (j < len && mx - sc > drop) drop = mx - sc;
		run = __shfl(sc, 63, 64), run_max = __shfl(mx, 63, 64);
	}
	sum = rg_wave_sum(sum);
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(drop, d, 64); drop = o > drop ? o : drop; }
	*score = sum, *max_drop = drop;
}

// where an anchor's window boundary sits (mm_adjust_minier, align.c:418-433): the middle of the k-mer -- or, with a homopolymer-compressed index, the start of the
// homopolymer run that holds the anchor's last base, on the query strand being aligned and on the reference
__device__ __forceinline__ void rg_boundary(const RgnBuffers &B, const RgnOpts &O, const RgnJobCtx &X, const Anchor &c, int32_t *r, int32_t *q)
{
	if (!O.hpc) { *r = rg_x(c) - (O.k >> 1), *q = rg_y(c) - (O.k >> 1); return; }
	const uint8_t *qs = B.qpool + (X.rev ? X.q_rev : X.q_fwd);
	int32_t i = rg_y(c);
	const int cq = qs[i];
	for (--i; i > 0; --i) if (qs[i] != cq) break;
	*q = i + 1;
	const int64_t x = rg_x(c);
	auto base = [&](int64_t p) -> int { const uint64_t o = X.t_base + (uint64_t)p; return (int)(B.S[o >> 3] >> ((o & 7) << 2) & 0xf); };
	const int cb = base(x);
	int64_t j = x - 1;
	for (; j >= 0; --j) if (base(j) != cb) break;
	*r = rg_x(c) + 1 - (int32_t)(x - j);
}

__device__ __forceinline__ void rg_emit(const RgnBuffers &B, const RgnOpts &O, const RgnJobCtx &X, uint32_t idx, int kind, int qs, int qe, int rs, int re, int bw, int anchor_i,
                                        int flag, int zdrop, int end_bonus)
{
	RgnWin w;
	w.qs = qs, w.qe = qe, w.rs = rs, w.re = re, w.anchor_i = anchor_i, w.kind = kind;
	B.win[idx] = w;
	const bool reversed = kind == 0; // the left extension runs backwards from the first anchor (align.c:787-788)
	KswJob j;
	j.qlen = qe - qs, j.tlen = re - rs;
	j.w = bw, j.zdrop = zdrop, j.end_bonus = end_bonus;
	flag |= X.gen_flag;
	if (O.max_sw_mat > 0 && (int64_t)j.tlen * j.qlen > O.max_sw_mat) flag |= KSWJ_SKIP; // align.c:349-351
	const uint64_t qbase = X.rev ? X.q_rev : X.q_fwd;
	j.q_off = reversed ? qbase + (uint64_t)qe - 1 : qbase + (uint64_t)qs;
	j.t_off = reversed ? X.t_base + (uint64_t)re - 1 : X.t_base + (uint64_t)rs;
	j.flag = flag | KSWJ_T_PACKED | (reversed ? (KSWJ_Q_REVERSED | KSWJ_T_REVERSED) : 0);
	j.tag = 0, j.reserved = 0;
	B.jobs[idx] = j;
}

} // namespace

// The window walk of mm_align1 (align.c:803-846) by the wave: the walk's state is where the last window ended; 64 anchors are tested against it at once, the first
// one that closes a window is found by a ballot, and the lanes after it are tested again against the new state.  emit == false only counts.
template <bool EMIT>
__device__ int rg_walk_windows(const RgnBuffers &B, const RgnOpts &O, const RgnJobCtx &X, const Anchor *a, int cnt1, int32_t rs, int32_t qs, uint32_t idx0, int lane)
{
	int32_t cs = rs, cq = qs;
	int n = 0;
	for (int i0 = 1; i0 < cnt1; i0 += 64) {
		const int i = i0 + lane;
		uint64_t y = 0;
		int32_t e_r = 0, e_q = 0;
		bool cand = false;
		if (i < cnt1) {
			const Anchor c = a[i];
			y = c.y;
			cand = !((y & (ref::SEED_IGNORE | ref::SEED_TANDEM)) && i != cnt1 - 1);
			if (cand) rg_boundary(B, O, X, c, &e_r, &e_q);
		}
		unsigned long long live = ~0ull; // lanes not yet passed by a closed window
		for (;;) {
			const bool close = cand && (i == cnt1 - 1 || (y & ref::SEED_LONG_JOIN) || (e_q - cq >= O.min_ksw_len && e_r - cs >= O.min_ksw_len));
			const unsigned long long m = __ballot(close) & live;
			if (!m) break;
			const int f = __builtin_ctzll(m);
			const int32_t n_r = __shfl(e_r, f, 64), n_q = __shfl(e_q, f, 64);
			if (EMIT && lane == f) {
				int bw = O.bw_gap;
				if (y & ref::SEED_LONG_JOIN) bw = e_q - cq > e_r - cs ? e_q - cq : e_r - cs;
				rg_emit(B, O, X, idx0 + (uint32_t)n, 1, cq, e_q, cs, e_r, bw, i, KSW_APPROX_MAX, O.zdrop, -1);
			}
			cs = n_r, cq = n_q, ++n;
			live = f == 63 ? 0ull : ~0ull << (f + 1);
		}
	}
	return n;
}

// one wavefront per region: the lanes share the passes over the region's anchors (gap sites, window walk); the short sequential rules (end trimming, the
// two long-gap filters over the site lists, the extension limits) are walked by lane 0
__global__ void __launch_bounds__(256) region_plan_kernel(RgnBuffers B, RgnOpts O)
{
	const int lane = threadIdx.x & 63;
	const uint32_t slot = blockIdx.x * 4u + (threadIdx.x >> 6);
	const uint32_t n_regs = B.cursors[RGN_CUR_REGS] < B.max_regs ? B.cursors[RGN_CUR_REGS] : B.max_regs;
	if (slot >= n_regs) return;
	const Reg1 r = B.regs[slot];
	RgnPlan pl = B.plan[slot];
	const RgnRead rd = B.reads[pl.read];
	const RgnUnit U = rg_unit(B, rd, pl.read, pl.seg);
	Anchor *a = B.sq_a + U.sq_off;                // the READ's squeezed anchors: the extension limits look at its other hits' anchors too
	const int qlen = U.qlen, n_a = U.n_a;
	pl.status = 0, pl.n_win = 0, pl.ug_len = 0, pl.ug_score = 0;
	if (r.cnt == 0) { pl.status = RGN_F_NO_CIGAR; if (lane == 0) B.plan[slot] = pl; return; }
	const int32_t rid = (int32_t)(a[r.as].x << 1 >> 33), rev = (int32_t)(a[r.as].x >> 63);
	const int32_t ref_len = (int32_t)B.ref_len[rid];
	int32_t as1 = r.as, cnt1 = r.cnt;
	if (O.flag & ref::F_SR) {
		// ---- a short read (align.c:664-669, :696-704, :803-833): aligned from its best run of seeds on one diagonal -- one gap window from the run's first seed to its
		// last, an M if the ungapped alignment beats any gapped one --, the extensions over the whole read and as much reference as its unaligned ends could span ----
		if (lane == 0) rr_best_diagonal_run(r, a, &as1, &cnt1); // mm_max_stretch, align.c:563-589
		as1 = __shfl(as1, 0, 64), cnt1 = __shfl(cnt1, 0, 64);
		const Anchor f = a[as1], l = a[as1 + cnt1 - 1];
		const int32_t rs = rg_x(f) + 1 - rg_span(f), qs = rg_y(f) + 1 - rg_span(f), re = rg_x(l) + 1, qe = rg_y(l) + 1;
		int32_t qs0 = 0, qe0 = qlen;
		int32_t rs0 = rs - rr_sr_reach(qs, O.a, O.q, O.e, O.end_bonus), re0 = re + rr_sr_reach(qlen - qe, O.a, O.q, O.e, O.end_bonus);
		rs0 = rs0 > 0 ? rs0 : 0, re0 = re0 < ref_len ? re0 : ref_len;
		if (a[r.as].y & ref::SEED_SELF) { // align.c:760-767
			const int32_t room_l = r.qs > r.rs ? r.qs - r.rs : r.rs - r.qs, room_r = r.qe > r.re ? r.qe - r.re : r.re - r.qe;
			rs0 = rr_self_limit(rs0, r.rs, room_l), qs0 = rr_self_limit(qs0, r.qs, room_l);
			re0 = ref_len - rr_self_limit(ref_len - re0, ref_len - r.re, room_r), qe0 = qlen - rr_self_limit(qlen - qe0, qlen - r.qe, room_r);
		}
		RgnJobCtx X;
		X.q_fwd = U.qpool_fwd, X.q_rev = U.qpool_fwd + (uint64_t)qlen, X.t_base = B.ref_off[rid], X.rev = rev;
		X.gen_flag = O.transition != 0 && O.b != O.transition ? KSW_GENERIC_SC : 0;
		const int32_t len = qe - qs;
		int32_t ug = 0, drop = 0;
		bool ungapped = false;
		if (len == re - rs && len > 0) {
			rg_ungapped(B, O, X, qs, rs, len, lane, &ug, &drop);
			ungapped = ug > (len - 2) * O.a - 2 * (O.q + O.e);
		}
		const bool has_left = qs > 0 && rs > 0, has_right = qe < qe0 && re < re0;
		const int n_win = (has_left ? 1 : 0) + (ungapped ? 0 : 1) + (has_right ? 1 : 0);
		uint32_t job0 = 0, piece0 = 0;
		if (lane == 0) job0 = atomicAdd(&B.cursors[RGN_CUR_JOBS], (unsigned)n_win), piece0 = atomicAdd(&B.cursors[RGN_CUR_PIECES], 3u);
		job0 = (uint32_t)__shfl((int)job0, 0, 64), piece0 = (uint32_t)__shfl((int)piece0, 0, 64);
		pl.job0 = job0, pl.piece0 = piece0, pl.n_win = n_win, pl.as1 = as1, pl.cnt1 = cnt1, pl.rid = rid, pl.rev = rev, pl.rs = rs, pl.qs = qs, pl.has_left = has_left, pl.has_right = has_right;
		if (ungapped) {
			pl.ug_len = len, pl.ug_score = ug;
			if (drop > O.zdrop) pl.status = RGN_F_MULTI_ROUND; // (the Z-drop test of the M trips: a second, exact pass -- the host's rounds; align.c:843-844)
		}
		if (job0 + (uint32_t)n_win > B.max_jobs || piece0 + 3u > B.max_jobs) pl.status = RGN_F_MULTI_ROUND, pl.n_win = 0;
		if (pl.status == 0 && lane == 0) {
			uint32_t idx = job0;
			if (has_left) rg_emit(B, O, X, idx++, 0, qs0, qs, rs0, rs, O.bw_ext, 0, KSW_EXTZ_ONLY | KSW_RIGHT | KSW_REV_CIGAR, r.split_inv ? O.zdrop_inv : O.zdrop, O.end_bonus);
			if (!ungapped) rg_emit(B, O, X, idx++, 1, qs, qe, rs, re, O.bw_gap, cnt1 - 1, KSW_APPROX_MAX, O.zdrop, -1);
			if (has_right) rg_emit(B, O, X, idx, 2, qe, qe0, re, re0, O.bw_ext, 0, KSW_EXTZ_ONLY, O.zdrop, O.end_bonus);
		}
		if (lane == 0) B.plan[slot] = pl;
		return;
	}
	if (!(O.flag & ref::F_NO_END_FLT)) {
		if (lane == 0) rr_trim_ends(r, a, O.bw, O.min_chain_score * 2, &as1, &cnt1); // mm_fix_bad_ends, align.c:527-561
		as1 = __shfl(as1, 0, 64), cnt1 = __shfl(cnt1, 0, 64);
	}
	int32_t *K = B.gap_sites + U.sq_off + as1;
	{
		const int n10 = rg_gap_sites(a + as1, cnt1, 10, K, lane);
		RG_SYNC();
		if (lane == 0) rr_drop_compensating_gaps(a + as1, K, n10, 40, O.max_gap >> 1, 10); // mm_filter_bad_seeds, align.c:454-489
		RG_SYNC();
		const int n30 = rg_gap_sites(a + as1, cnt1, 30, K, lane);
		RG_SYNC();
		if (lane == 0) rr_join_gap_clusters(a + as1, K, n30, O.max_gap >> 1);               // mm_filter_bad_seeds_alt, align.c:491-525
		RG_SYNC(); // (the flags lane 0 set in the anchors are read by all lanes below)
	}
	RgnJobCtx X;
	X.q_fwd = U.qpool_fwd, X.q_rev = U.qpool_fwd + (uint64_t)qlen, X.t_base = B.ref_off[rid], X.rev = rev;
	X.gen_flag = O.transition != 0 && O.b != O.transition ? KSW_GENERIC_SC : 0; // align.c:347-348
	int32_t rs, qs, re, qe; // the boundaries of the first and the last anchor (mm_adjust_minier, align.c:418-433)
	rg_boundary(B, O, X, a[as1], &rs, &qs);
	rg_boundary(B, O, X, a[as1 + cnt1 - 1], &re, &qe);

	// how far the two extensions may reach (align.c:706-767; region_rules.hpp: one routine for both ends, on coordinates that face the end): lane 0
	int32_t rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;
	if (lane == 0) {
		const RrExtScoring S = { O.a, O.q, O.e, O.max_gap, O.min_cnt };
		const Anchor first = a[r.as], last = a[r.as + r.cnt - 1];
		rr_extension_limit(rr_x(first) + 1 - rr_span(first), rr_y(first) + 1 - rr_span(first), rs, qs,
			[&](int k, int32_t *nt, int32_t *nq) { // the read's earlier seeds on the same sequence and strand
				const int32_t i = r.as - 1 - k;
				if (i < 0 || !rr_same_target(a[i], first)) return false;
				*nt = rr_x(a[i]) + 1 - rr_span(a[i]), *nq = rr_y(a[i]) + 1 - rr_span(a[i]);
				return true;
			}, S, true, &rs0, &qs0);
		int32_t far_t, far_q; // the right end, in distances from the sequences' ends
		rr_extension_limit(ref_len - (rr_x(last) + 1), qlen - (rr_y(last) + 1), ref_len - re, qlen - qe,
			[&](int k, int32_t *nt, int32_t *nq) { // its later ones
				const int32_t i = r.as + r.cnt + k;
				if (i >= n_a || !rr_same_target(a[i], first)) return false;
				*nt = ref_len - (rr_x(a[i]) + 1), *nq = qlen - (rr_y(a[i]) + 1);
				return true;
			}, S, false, &far_t, &far_q);
		re0 = ref_len - far_t, qe0 = qlen - far_q;
		if (first.y & ref::SEED_SELF) { // an overlap with itself must not extend across the diagonal (align.c:760-767)
			const int32_t room_l = r.qs > r.rs ? r.qs - r.rs : r.rs - r.qs, room_r = r.qe > r.re ? r.qe - r.re : r.re - r.qe;
			rs0 = rr_self_limit(rs0, r.rs, room_l), qs0 = rr_self_limit(qs0, r.qs, room_l);
			re0 = ref_len - rr_self_limit(ref_len - re0, ref_len - r.re, room_r), qe0 = qlen - rr_self_limit(qlen - qe0, qlen - r.qe, room_r);
		}
	}
	rs0 = __shfl(rs0, 0, 64), qs0 = __shfl(qs0, 0, 64), re0 = __shfl(re0, 0, 64), qe0 = __shfl(qe0, 0, 64);
	// the windows, in the order the reference aligns them (align.c:779-890): counted first, so that the region's jobs are one dense run
	const bool has_left = qs > 0 && rs > 0;
	const int n_gap_win = rg_walk_windows<false>(B, O, X, a + as1, cnt1, rs, qs, 0, lane);
	// (the right extension exists when the LAST gap window's end -- the last anchor's boundary, or the first one's when there is a single anchor -- lies inside the limits)
	const bool has_right = qe < qe0 && re < re0;
	const int n_win = (has_left ? 1 : 0) + n_gap_win + (has_right ? 1 : 0);
	uint32_t job0 = 0;
	if (lane == 0) job0 = atomicAdd(&B.cursors[RGN_CUR_JOBS], (unsigned)n_win);
	job0 = (uint32_t)__shfl((int)job0, 0, 64);
	pl.job0 = job0, pl.piece0 = job0, pl.n_win = n_win, pl.as1 = as1, pl.cnt1 = cnt1, pl.rid = rid, pl.rev = rev, pl.rs = rs, pl.qs = qs, pl.has_left = has_left, pl.has_right = has_right;
	if (job0 + (uint32_t)n_win > B.max_jobs) { pl.status = RGN_F_MULTI_ROUND, pl.n_win = 0; if (lane == 0) B.plan[slot] = pl; return; } // (cannot happen: sized by anchors + 2 per chain)
	uint32_t idx = job0;
	if (has_left) { if (lane == 0) rg_emit(B, O, X, idx, 0, qs0, qs, rs0, rs, O.bw_ext, 0, KSW_EXTZ_ONLY | KSW_RIGHT | KSW_REV_CIGAR, r.split_inv ? O.zdrop_inv : O.zdrop, O.end_bonus); ++idx; }
	idx += (uint32_t)rg_walk_windows<true>(B, O, X, a + as1, cnt1, rs, qs, idx, lane);
	if (has_right && lane == 0) rg_emit(B, O, X, idx, 2, qe, qe0, re, re0, O.bw_ext, 0, KSW_EXTZ_ONLY, O.zdrop, O.end_bonus);
	if (lane == 0) B.plan[slot] = pl;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// region_consume_kernel: one region per thread
// ---------------------------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) region_consume_kernel(RgnBuffers B, RgnOpts O, const uint32_t *cigar_pool)
{
	const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
	const uint32_t n_regs = B.cursors[RGN_CUR_REGS] < B.max_regs ? B.cursors[RGN_CUR_REGS] : B.max_regs;
	if (slot >= n_regs) return;
	RgnPlan pl = B.plan[slot];
	FinRegion fr;
	__builtin_memset(&fr, 0, sizeof fr);
	if (pl.status != 0 || (pl.n_win <= 0 && pl.ug_len <= 0)) {
		if (pl.status == 0) pl.status = RGN_F_NO_CIGAR;
		B.fin[slot] = fr, B.plan[slot] = pl;
		atomicOr(&B.rout[(size_t)pl.read * (size_t)B.rout_stride].flags, (unsigned)pl.status);
		return;
	}
	const RgnRead rd = B.reads[pl.read];
	const RgnUnit U = rg_unit(B, rd, pl.read, pl.seg);
	int32_t rs1 = pl.rs, qs1 = pl.qs, re1 = pl.rs, qe1 = pl.qs; // no left extension: the alignment starts at the first anchor (align.c:800-801)
	int32_t dp = 0;
	uint32_t cap = 0, n_ops = 0, last_op = 0, n_pieces = 0, sum_ops = 0;
	int status = 0;
	const bool inv_test_off = (O.flag & (ref::F_SPLICE | ref::F_SR | ref::F_FOR_ONLY | ref::F_REV_ONLY)) != 0;
	bool ug_todo = pl.ug_len > 0; // a short read's ungapped gap window: between the left extension and the right one, an M of its own (no DP job)
	for (int k = 0; k <= pl.n_win && status == 0; ++k) {
		const uint32_t job = pl.job0 + (uint32_t)k;
		RgnWin w;
		w.kind = 2;
		if (k < pl.n_win) w = B.win[job];
		if (ug_todo && w.kind == 2) {
			ug_todo = false;
			dp += pl.ug_score;
			re1 = pl.rs + pl.ug_len, qe1 = pl.qs + pl.ug_len;
			const uint32_t word = (uint32_t)pl.ug_len << 4; // MM_CIGAR_MATCH
			FinPiece pc; pc.off = word, pc.n = kFinLiteral;
			B.pieces[pl.piece0 + n_pieces++] = pc;
			sum_ops += 1;
			if (cap == 0) cap = rg_roundup32(1 + 7);
			else if (n_ops + 1 + 7 > cap) cap = rg_roundup32(n_ops + 1 + 7);
			n_ops += n_ops > 0 && last_op == 0 ? 0 : 1;
			last_op = 0;
		}
		if (k == pl.n_win) break;
		const KswRes ez = B.res[B.perm[job]];
		bool take = ez.n_cigar > 0;
		if (w.kind == 1) { // a gap fill: the approximate pass is tested first (align.c:843; mm_test_zdrop on the kernel's own scan of its alignment)
			if (ez.zd_max == KSW_ZD_NONE) { status = RGN_F_MULTI_ROUND; break; } // (a job no register-resident kernel took: the host scans its CIGAR)
			const bool maybe_inv = !inv_test_off && ez.zd_max > O.zdrop_inv && ez.zd_q1 - ez.zd_q0 < O.max_gap && ez.zd_t1 - ez.zd_t0 < O.max_gap;
			if (maybe_inv || ez.zd_max > O.zdrop) { status = RGN_F_MULTI_ROUND; break; } // second pass, maybe an inversion: the host's rounds
			if (ez.zdropped) { status = RGN_F_MULTI_ROUND; break; }                     // truncated: cut and split (align.c:848-868)
			dp += ez.score;
			re1 = w.re, qe1 = w.qe;
		} else if (w.kind == 0) {
			if (take) dp += ez.max;
			rs1 = w.re - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
			qs1 = w.qe - (ez.reach_end ? w.qe - w.qs : ez.max_q + 1);
		} else {
			if (take) dp += ez.max;
			re1 = w.rs + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
			qe1 = w.qs + (ez.reach_end ? w.qe - w.qs : ez.max_q + 1);
		}
		if (take) {
			const uint32_t n = (uint32_t)ez.n_cigar;
			FinPiece pc; pc.off = ez.cigar_off, pc.n = n;
			B.pieces[pl.piece0 + n_pieces++] = pc;
			sum_ops += n;
			// mm_extra_t's growth, as mm_append_cigar would have done it window by window (align.c:305-334): the hand-over carries `capacity`
			if (cap == 0) cap = rg_roundup32(n + 7);
			else if (n_ops + n + 7 > cap) cap = rg_roundup32(n_ops + n + 7);
			n_ops += n_ops > 0 && last_op == (cigar_pool[ez.cigar_off] & 0xf) ? n - 1 : n;
			last_op = cigar_pool[ez.cigar_off + n - 1] & 0xf;
		}
	}
	if (status == 0 && sum_ops == 0) status = RGN_F_NO_CIGAR;
	if (status == 0 && sum_ops > (uint32_t)kFinMaxOps) status = RGN_F_LONG_CIGAR;
	pl.status = status, pl.dp_score = dp, pl.capacity = cap;
	if (status == 0) {
		Reg1 r = B.regs[slot];
		r.rs = rs1, r.re = re1;
		if (!pl.rev) r.qs = qs1, r.qe = qe1; // align.c:894
		else r.qs = U.qlen - qe1, r.qe = U.qlen - qs1;
		B.regs[slot] = r;
		fr.q_pos = (pl.rev ? U.qpool_fwd + (uint64_t)U.qlen : U.qpool_fwd) + (uint64_t)qs1;
		fr.t_pos = B.ref_off[pl.rid] + (uint64_t)rs1;
		fr.piece0 = pl.piece0, fr.n_pieces = n_pieces;
		fr.out_off = atomicAdd(&B.cursors[RGN_CUR_OUT], sum_ops);
		fr.q_len = qe1 - qs1, fr.t_len = re1 - rs1;
		atomicMax(&B.cursors[RGN_CUR_MAX_OPS], sum_ops);
		atomicAdd(&B.cursors[RGN_CUR_N_FIN], 1u);
	} else atomicOr(&B.rout[(size_t)pl.read * (size_t)B.rout_stride].flags, (unsigned)status);
	B.fin[slot] = fr, B.plan[slot] = pl;
}

__global__ void __launch_bounds__(256) first_chain_span_kernel(int n_reads, const Anchor *a, const uint64_t *u, const uint64_t *a_off, const uint64_t *u_off, const int32_t *n_u, int32_t *span)
{
	const int r = (int)(blockIdx.x * 256u + threadIdx.x);
	if (r >= n_reads) return;
	int32_t y0 = 0, y1 = 0;
	if (n_u[r] > 0) {
		const int32_t cnt = (int32_t)u[u_off[r]];
		y0 = (int32_t)a[a_off[r]].y, y1 = (int32_t)a[a_off[r] + (uint64_t)(cnt - 1)].y;
	}
	span[2 * r] = y0, span[2 * r + 1] = y1;
}
void launch_first_chain_span(int n_reads, const Anchor *a, const uint64_t *u, const uint64_t *a_off, const uint64_t *u_off, const int32_t *n_u, int32_t *span, void *stream)
{
	if (n_reads <= 0) return;
	hipLaunchKernelGGL(first_chain_span_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_reads, a, u, a_off, u_off, n_u, span);
	HIP_CHECK(hipGetLastError());
}

__global__ void __launch_bounds__(256) gather_chains_kernel(const RgnGather *g, const Anchor *a0, const Anchor *a1, const uint64_t *mp, Anchor *a_out, uint64_t *mp_out)
{
	const RgnGather G = g[blockIdx.x];
	const Anchor *src = (G.src ? a1 : a0) + G.a_src;
	for (int i = threadIdx.x; i < G.n_a; i += 256) a_out[G.a_dst + i] = src[i];
	for (int i = threadIdx.x; i < G.n_mp; i += 256) mp_out[G.mp_dst + i] = mp[G.mp_src + i];
}
void launch_gather_chains(int n, const RgnGather *g, const Anchor *a0, const Anchor *a1, const uint64_t *mp, Anchor *a_out, uint64_t *mp_out, void *stream)
{
	if (n <= 0) return;
	hipLaunchKernelGGL(gather_chains_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, g, a0, a1, mp, a_out, mp_out);
	HIP_CHECK(hipGetLastError());
}

void launch_chain_regs(const RgnBuffers &B, const RgnOpts &O, void *stream)
{
	if (B.n_reads <= 0) return;
	const size_t C = (size_t)B.lds_chains;
	const size_t lds = C * (80 + 8 + 8 + 4 * 5 + 2 + 1) + 64 + (B.rout_stride == 2 ? C * 80 + 8 : 0);
	hipLaunchKernelGGL(chain_regs_kernel, dim3(B.n_reads), dim3(64), lds, (hipStream_t)stream, B, O);
	HIP_CHECK(hipGetLastError());
}
void launch_region_plan(const RgnBuffers &B, const RgnOpts &O, void *stream)
{
	if (B.max_regs == 0) return;
	hipLaunchKernelGGL(region_plan_kernel, dim3((B.max_regs + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, O);
	HIP_CHECK(hipGetLastError());
}
void launch_region_consume(const RgnBuffers &B, const RgnOpts &O, const uint32_t *cigar_pool, void *stream)
{
	if (B.max_regs == 0) return;
	hipLaunchKernelGGL(region_consume_kernel, dim3((B.max_regs + 63) / 64), dim3(64), 0, (hipStream_t)stream, B, O, cigar_pool);
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

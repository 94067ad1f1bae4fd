// Batched dual-affine extension DP for gfx950 -- the device counterpart of the reference's
// ksw_extd2_sse (/root/reference/ksw2_extd2_sse.c:34-401) and ksw_backtrack (ksw2.h:130-162).
//
// Mapping.  A TEAM of 1, 4 or 8 wavefronts owns one DP job at a time (persistent teams pull jobs from a queue head with
// one atomic per job; TEAM > 1: the team is the workgroup, round 4).  An anti-diagonal r is swept in chunks of 64 target positions t,
// one chunk per wave and round (highest chunks first: a round reads row r-1 at its positions and their left neighbours, meets at a
// barrier, then stores row r), the score fill and the exact row maximum are strided over all the team's lanes, a barrier separates
// the passes.  A band-751 anti-diagonal is twelve chunks: one wave took 3 passes x 12 chunks per row and a 500 x 1000 extension held its
// launch for 6 ms (rounds 1-3: 24 % of the kernel time for 3 % of the cells); eight waves take two rounds per pass.
// the per-position difference state (u,v,x,y | x2,y2,s) lives in LDS as two packed dwords per t, the
// target/reversed-query bytes next to it, so a row costs 4 ds_read_b32 + 2 ds_write_b32 per cell and no
// HBM traffic except the 1 B/cell direction byte.  No MFMA: this is int8 max/add with data-dependent
// control, not a contraction.
//
// Exactness.  The reference evaluates 16-lane blocks over the block-aligned interval [st,en] around the
// valid interval [st0,en0]; with a binding band valid cells read those out-of-range lanes (SURVEY.md
// section 7, hard part 1).  We therefore sweep exactly [st,en], keep the state bytes of every position a
// later row can still read, reproduce the 16-byte chunked score fill including its overshoot past en0 (and
// past the end of s[] into the target copy, ksw2_extd2_sse.c:166-180 with the layout of :107-110), and do
// all arithmetic mod 256 with signed 8-bit compares.  Row maxima use the reference's scan order (:326-358).
//
// State window.  Row r touches positions [st-1, en+15] only, st and en never decrease, and a position is first
// touched when it comes within 16 of en.  The per-position arrays are therefore rings of L.ring slots (a power of
// two >= the widest row + 64): slot t & (ring-1) belongs to position t from the moment the frontier reaches it
// (when it gets the values the reference's arrays are filled with up front) until position t + ring takes it over,
// by which time no row can read t any more.  The footprint depends on min(qlen, tlen, band), not on tlen: a
// 300 x 50000 splice gap fill needs the same 7 KB of LDS as a 300 x 300 one.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"

namespace mm2amd {

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__device__ __forceinline__ int sx8(int v) { return (int)(int8_t)v; }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d)
{
	return (uint32_t)(a & 0xff) | (uint32_t)(b & 0xff) << 8 | (uint32_t)(c & 0xff) << 16 | (uint32_t)(d & 0xff) << 24;
}

__device__ __forceinline__ int wave_bcast_i32(int v, int src_lane) { return __shfl(v, src_lane, 64); }

// max-reduce a 64-bit key over the wave; every lane returns the maximum
__device__ __forceinline__ long long wave_max_i64(long long k)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		long long o = __shfl_xor(k, off, 64);
		k = o > k ? o : k;
	}
	return k;
}

struct RowIv { int st0, en0, st, en; };

// valid and block-aligned interval of anti-diagonal r (ksw2_extd2_sse.c:137-147); st0>en0 means "band closed"
__device__ __forceinline__ RowIv row_interval(int r, int qlen, int tlen, int w)
{
	RowIv iv;
	int st = 0, en = tlen - 1;
	if (st < r - qlen + 1) st = r - qlen + 1;
	if (en > r) en = r;
	if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
	if (en > (r + w) >> 1) en = (r + w) >> 1;
	iv.st0 = st, iv.en0 = en;
	iv.st = st / 16 * 16, iv.en = (en + 16) / 16 * 16 - 1;
	return iv;
}

struct EzState { int max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end; };

// ksw_apply_zdrop with is_rot=1 (ksw2.h:171-187)
__device__ __forceinline__ bool zdrop_test(EzState &ez, int H, int r, int t, int zdrop, int e)
{
	if (H > ez.max) {
		ez.max = H, ez.max_t = t, ez.max_q = r - t;
	} else if (t >= ez.max_t && r - t >= ez.max_q) {
		int tl = t - ez.max_t, ql = (r - t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && ez.max - H > zdrop + l * e) { ez.zdropped = 1; return true; }
	}
	return false;
}

struct CigOut { uint32_t *c; int n; uint32_t last; };

// ksw_push_cigar (ksw2.h:114-124); the scratch buffer holds qlen+tlen entries, which a traceback cannot exceed
__device__ __forceinline__ void cig_push(CigOut &g, uint32_t op, int len)
{
	if (g.n == 0 || op != (g.last & 0xf)) {
		if (g.n > 0) g.c[g.n - 1] = g.last;
		g.last = (uint32_t)len << 4 | op;
		++g.n;
	} else g.last += (uint32_t)len << 4;
}

// Traceback over the direction matrix in HBM by the whole wave (ksw2.h:130-162, is_rot=1): lane k looks k cells ahead along the
// run of the current state (match diagonal, or one of the gap states) and a ballot tells how far the run goes, so a run costs one
// round of loads however long it is.  The cell that ends a run is handled exactly as the reference handles every cell, including
// the forced moves outside the band.  min_intron_len > 0 (splice mode): state 3 is an intron and becomes an N operation.
// off[r]/off_end[r] of the reference are st/en of row r, recomputed here.  All lanes keep identical copies of (i, j, state, g).
__device__ void traceback(const uint8_t *dir, size_t ncol, int qlen, int tlen, int w, int i0, int j0, int min_intron_len, CigOut &g, int lane)
{
	int i = i0, j = j0, state = 0;
	const uint32_t op3 = min_intron_len > 0 ? 3u : 2u;
	while (i >= 0 && j >= 0) {
		const int di = (state == 2 || state == 4) ? 0 : 1, dj = (state == 1 || state == 3) ? 0 : 1;
		const int ii = i - lane * di, jj = j - lane * dj;
		const bool valid = ii >= 0 && jj >= 0;
		int force = -1, tmp = 0;
		if (valid) {
			const RowIv iv = row_interval(ii + jj, qlen, tlen, w);
			if (ii < iv.st) force = 2;
			if (ii > iv.en) force = 1;
			if (force < 0) tmp = dir[(size_t)(ii + jj) * ncol + (size_t)(ii - iv.st)];
		}
		const bool cont = valid && force < 0 && (state == 0 ? (tmp & 7) == 0 : (tmp >> (state + 2) & 1) != 0);
		const unsigned long long stop = ~__ballot(cont);
		const int run = stop ? __builtin_ctzll(stop) : 64;
		if (run > 0) {
			cig_push(g, state == 0 ? 0u : state == 1 ? 2u : state == 3 ? op3 : 1u, run);
			i -= run * di, j -= run * dj;
			continue;
		}
		const int f0 = __builtin_amdgcn_readfirstlane(force), t0 = __builtin_amdgcn_readfirstlane(tmp); // the cell we stand on
		if (state == 0) state = t0 & 7;
		else if (!(t0 >> (state + 2) & 1)) state = 0;
		if (state == 0) state = t0 & 7;
		if (f0 >= 0) state = f0;
		if (state == 0) cig_push(g, 0, 1), --i, --j;
		else if (state == 1 || (state == 3 && min_intron_len <= 0)) cig_push(g, 2, 1), --i;
		else if (state == 3) cig_push(g, 3, 1), --i;
		else cig_push(g, 1, 1), --j;
	}
	if (i >= 0) cig_push(g, min_intron_len > 0 && i >= min_intron_len ? 3 : 2, i + 1);
	if (j >= 0) cig_push(g, 1, j + 1);
	if (g.n > 0) g.c[g.n - 1] = g.last; // flush the run being accumulated
}

// LDS_STATE: the per-position state lives in LDS (jobs up to ~11 k positions).  The other instantiation keeps it in a per-wave
// slab of HBM instead: slower, but it takes jobs of any length (very long gaps on real genomes), so that no input makes the
// library give up.
// ordering between the lanes of the wave: LDS needs a wavefront fence, the HBM-resident state a workgroup-scope one
#define STATE_SYNC() do { if (TEAM > 1) { __syncthreads(); } else if (LDS_STATE) { WAVE_SYNC(); } else { __threadfence_block(); __builtin_amdgcn_wave_barrier(); } } while (0)
// MODE 1 (SINGLE): the single-affine recurrences of ksw_extz2_sse (ksw2_extz2_sse.c:25-311) instead of the dual-affine ones:
// scores shifted by 2(q+e), unsigned second maximum and clamp (:49-50), zero-initialised state, two gap states only.
// MODE 2 (SPLICE): the splice-aware recurrences of ksw_exts2_sse (ksw2_exts2_sse.c:33-465): no band, the second gap state is an
// intron on the target (x2 only; opening costs q2, extension nothing) priced per position by donor/acceptor bytes that take the
// place of y2 and of the spare byte in the second state dword; no clamp; Z-drop without the diagonal term; N in the CIGAR.
template <bool LDS_STATE, int MODE, int TEAM>
__global__ void __launch_bounds__(TEAM > 4 ? 512 : 256) ksw_extd2_kernel(KswLaunch L)
{
	constexpr bool SINGLE = MODE == 1, SPLICE = MODE == 2;
	static_assert(TEAM == 1 || LDS_STATE, "a team shares its state through LDS");
	constexpr int NT = 64 * TEAM; // lanes working on one job
	MM2_DYN_LDS(uint8_t, lds_raw);
	__shared__ long long s_best[TEAM > 1 ? TEAM : 1];
	__shared__ int s_team_all[TEAM > 1 ? 4 : 16]; // per job in flight (TEAM == 1: one per wave of the block): [0] the job id, [1] H(en0), [2] H(st0) of the row just swept
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	int *const s_team = s_team_all + (TEAM > 1 ? 0 : 4 * wave_in_block);
	const int tid = TEAM > 1 ? (int)threadIdx.x : lane, twave = TEAM > 1 ? wave_in_block : 0; // within the team
	const int slot = TEAM > 1 ? (int)blockIdx.x : blockIdx.x * (blockDim.x >> 6) + wave_in_block;
	const size_t region = (ksw_lds_per_wave(L.ring, L.max_Q16) + 15) / 16 * 16;
	uint8_t *my;
	if (TEAM > 1) my = lds_raw;
	else if (LDS_STATE) my = lds_raw + (size_t)wave_in_block * region;
	else my = L.state_pool + (size_t)slot * region;
	uint8_t *dir = L.dir_pool + (size_t)slot * L.slot_bytes;
	const int m = L.sc.m;
	const int RS = L.ring, RM = RS - 1;

	for (;;) {
		int jid = 0;
		if (TEAM > 1) {
			if (threadIdx.x == 0) s_team[0] = atomicAdd(L.counter, 1);
			__syncthreads();
			jid = s_team[0];
			__syncthreads(); // (everyone has it before the next fetch overwrites it)
		} else {
			if (lane == 0) jid = atomicAdd(L.counter, 1);
			jid = __builtin_amdgcn_readfirstlane(jid);
		}
		if (jid >= L.n_jobs) break;
		const KswJob J = L.jobs[jid];
		const int qlen = J.qlen, tlen = J.tlen, flag = J.flag;
		EzState ez;
		ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1;
		ez.max = 0, ez.score = ez.mqe = ez.mte = KSW_NEG_INF, ez.zdropped = 0, ez.reach_end = 0;
		CigOut g = { L.cigar_tmp + (size_t)slot * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;

		int q = L.sc.q, e = L.sc.e, q2 = L.sc.q2, e2 = L.sc.e2;
		const int qe_in = q + e; // taken before the swap (ksw2_extd2_sse.c:68 vs :78); seeds H(0,0)
		if (MODE == 0 && q2 + e2 < q + e) { int t = q; q = q2, q2 = t; t = e, e = e2, e2 = t; }
		const int qe = q + e, qe2 = q2 + e2;
		int min_sc = L.sc.mat[1];
		for (int t = 1; t < m * m; ++t) min_sc = min_sc < L.sc.mat[t] ? min_sc : L.sc.mat[t];
		const bool degenerate = (SINGLE ? m <= 0 : m <= 1) || qlen <= 0 || tlen <= 0 || -min_sc > 2 * (q + e) || (SPLICE && (q2 <= q + e || e <= 0));
		int w = J.w;
		if (w < 0 || SPLICE) w = tlen > qlen ? tlen : qlen; // a band this wide never binds

		if (flag & KSWJ_SKIP) ez.zdropped = 1;
		else if (!degenerate) {
			const bool with_cigar = !(flag & KSW_SCORE_ONLY), approx_max = flag & KSW_APPROX_MAX, right = flag & KSW_RIGHT;
			const int sc_mch = L.sc.mat[0], sc_mis = L.sc.mat[1];
			const int sc_N = L.sc.mat[m * m - 1] == 0 ? sx8(MODE ? -e : -e2) : L.sc.mat[m * m - 1];
			const int qe2s = (q + e) * 2, max_scu = (L.sc.mat[0] + (q + e) * 2) & 0xff; // single-affine: score shift and unsigned clamp (:69,:79)
			const int T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
			size_t ncol = qlen < tlen ? qlen : tlen;
			ncol = (((ncol < (size_t)w + 1 ? ncol : (size_t)w + 1) + 15) / 16 + 1) * 16;
			int long_thres, long_diff;
			if (SPLICE) { // ksw2_exts2_sse.c:98-101
				long_thres = (q2 - q) / e - 1;
				if (q2 > q + e + long_thres * e) ++long_thres;
				long_diff = long_thres * e - (q2 - q);
			} else {
				long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
				if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
				long_diff = long_thres * (e - e2) - (q2 - q) - e2;
			}

			uint32_t *A = (uint32_t *)my;      // rings of RS slots: byte0 u, byte1 v, byte2 x, byte3 y
			uint32_t *B = A + RS;              // byte0 x2, byte1 y2 (splice: donor), byte2 s, (splice: byte3 acceptor)
			int32_t *H = (int32_t *)(B + RS);
			uint8_t *TG = (uint8_t *)(H + RS); // target copy (the reference's sf[])
			uint8_t *QR = TG + RS;             // [0, Q16+16): reversed query, zero padded (the reference's qr[], which follows sf[])
			const int nqe = sx8(-q - e), nqe2 = SPLICE ? sx8(-q2) : sx8(-q2 - e2);
			auto tfetch = [&](int t) -> int { // target base t straight from the pools
				const uint64_t pos = (flag & KSWJ_T_REVERSED) ? J.t_off - (uint64_t)t : J.t_off + (uint64_t)t;
				return (flag & KSWJ_T_PACKED) ? (int)(L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (int)L.tpool[pos];
			};
			const bool sp_strand = SPLICE && (flag & (KSW_SPLICE_FOR | KSW_SPLICE_REV));
			const bool sp_for = flag & KSW_SPLICE_FOR, sp_revc = flag & KSW_REV_CIGAR;
			int sp0 = 0, sp1 = 0, sp2 = 0, sp3 = 0;
			if (SPLICE) {
				if (flag & KSW_SPLICE_CMPLX) sp0 = 3, sp1 = 5, sp2 = 7, sp3 = 10; // (int)({8,15,21,30} / 3. + .499)
				else sp0 = (flag & KSW_SPLICE_FLANK) ? L.sc.noncan / 2 : 0, sp1 = sp2 = sp3 = L.sc.noncan;
			}
			auto sp_cost = [&](int z) { return z < 0 ? 0 : z == 0 ? -sp0 : z == 1 ? -sp1 : z == 2 ? -sp2 : -sp3; };
			// annotated splice sites of this window (ksw2_exts2_sse.c:201-217): junc[i] of the target as the job reads it
			const uint32_t nj = SPLICE && L.sc.juncs ? J.reserved : 0u;
			const uint32_t *jent = nj ? L.sc.juncs + J.tag : nullptr;
			auto jbits = [&](int i) -> uint32_t {
				if (i < 0 || i >= tlen) return 0u;
				const uint32_t pos = (flag & KSWJ_T_REVERSED) ? (uint32_t)(tlen - 1 - i) : (uint32_t)i;
				uint32_t lo = 0, hi = nj;
				while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (jent[mid] >> 4 < pos) lo = mid + 1; else hi = mid; }
				return lo < nj && jent[lo] >> 4 == pos ? jent[lo] & 15u : 0u;
			};
			auto jscore = [&](int i) -> int { // the score byte of position i, 0xff when the window has none there (entries  t << 8 | byte)
				const uint32_t pos = (flag & KSWJ_T_REVERSED) ? (uint32_t)(tlen - 1 - i) : (uint32_t)i;
				uint32_t lo = 0, hi = nj;
				while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (jent[mid] >> 8 < pos) lo = mid + 1; else hi = mid; }
				return lo < nj && jent[lo] >> 8 == pos ? (int)(jent[lo] & 0xffu) : 0xff;
			};
			const uint32_t jd_mask = !sp_revc ? (sp_for ? 1u : 0u) | ((flag & KSW_SPLICE_REV) ? 8u : 0u) : (sp_for ? 2u : 0u) | ((flag & KSW_SPLICE_REV) ? 4u : 0u);
			const uint32_t ja_mask = !sp_revc ? (sp_for ? 2u : 0u) | ((flag & KSW_SPLICE_REV) ? 4u : 0u) : (sp_for ? 1u : 0u) | ((flag & KSW_SPLICE_REV) ? 8u : 0u);
			// Take positions (frontier, upto] into the window with the values the reference's up-front fill gives them
			// (:107-128; splice: donor / acceptor costs from the neighbouring bases, ksw2_exts2_sse.c:120-194).
			int frontier = -1;
			auto admit = [&](int upto) {
				for (int t = frontier + 1 + tid; t <= upto; t += NT) {
					const int k = t & RM;
					uint32_t bv = SINGLE ? 0u : SPLICE ? (uint32_t)(nqe2 & 0xff) : pack4(nqe2, nqe2, 0, 0);
					if (sp_strand) {
						int zd = 3, za = 3;
						if (t < tlen - 4) {
							const int c1 = tfetch(t + 1), c2 = tfetch(t + 2), c3 = tfetch(t + 3);
							if (!sp_revc) {
								if (sp_for) {
									if (c1 == 2 && c2 == 3) zd = (c3 == 0 || c3 == 2) ? -1 : 0;
									else if (c1 == 2 && c2 == 1) zd = 1;
									else if (c1 == 0 && c2 == 3) zd = 2;
								} else {
									if (c1 == 1 && c2 == 3) zd = (c3 == 0 || c3 == 2) ? -1 : 0;
									else if (c1 == 2 && c2 == 3) zd = 2;
								}
							} else {
								if (sp_for) {
									if (c1 == 2 && c2 == 0) zd = (c3 == 1 || c3 == 3) ? -1 : 0;
									else if (c1 == 1 && c2 == 0) zd = 2;
								} else {
									if (c1 == 1 && c2 == 0) zd = (c3 == 1 || c3 == 3) ? -1 : 0;
									else if (c1 == 1 && c2 == 2) zd = 1;
									else if (c1 == 3 && c2 == 0) zd = 2;
								}
							}
						}
						if (t >= 2 && t < tlen) {
							const int c0 = tfetch(t), c1 = tfetch(t - 1), c2 = tfetch(t - 2);
							if (!sp_revc) {
								if (sp_for) {
									if (c1 == 0 && c0 == 2) za = (c2 == 1 || c2 == 3) ? -1 : 0;
									else if (c1 == 0 && c0 == 1) za = 2;
								} else {
									if (c1 == 0 && c0 == 1) za = (c2 == 1 || c2 == 3) ? -1 : 0;
									else if (c1 == 2 && c0 == 1) za = 1;
									else if (c1 == 0 && c0 == 3) za = 2;
								}
							} else {
								if (sp_for) {
									if (c1 == 3 && c0 == 2) za = (c2 == 0 || c2 == 2) ? -1 : 0;
									else if (c1 == 1 && c0 == 2) za = 1;
									else if (c1 == 3 && c0 == 0) za = 2;
								} else {
									if (c1 == 3 && c0 == 1) za = (c2 == 0 || c2 == 2) ? -1 : 0;
									else if (c1 == 3 && c0 == 2) za = 2;
								}
							}
						}
						int dcost = sp_cost(zd), acost = sp_cost(za);
						if (flag & KSW_SPLICE_SCORE) { // --spsc (ksw2_exts2_sse.c:196-200): every site is priced, by its score or by junc_pen
							if (t < tlen - 1) {
								const int b = jscore(t + 1), donor_val = (sp_for == !sp_revc) ? 0 : 1;
								dcost = sx8(dcost + ((b == 0xff || (b & 1) != donor_val) ? -L.sc.junc_pen : (b >> 1) - 64));
								acost = sx8(acost + ((b == 0xff || (b & 1) != !donor_val) ? -L.sc.junc_pen : (b >> 1) - 64));
							}
						} else if (nj) { // the bonus is added in the 8-bit lanes (int8 wrap) wherever the annotation has the site on the assumed strand
							if (t < tlen - 1 && (jbits(t + 1) & jd_mask)) dcost = sx8(dcost + L.sc.junc_bonus);
							if (t < tlen && (jbits(t) & ja_mask)) acost = sx8(acost + L.sc.junc_bonus);
						}
						bv |= (uint32_t)(dcost & 0xff) << 8 | (uint32_t)(acost & 0xff) << 24;
					}
					A[k] = SINGLE ? 0u : pack4(nqe, nqe, nqe, nqe);
					B[k] = bv;
					H[k] = KSW_NEG_INF;
					TG[k] = t < tlen ? (uint8_t)tfetch(t) : (uint8_t)0;
				}
				if (upto > frontier) frontier = upto;
			};

			// ---- per-job initialisation (ksw2_extd2_sse.c:107-128) ----
			for (int i = tid; i < Q16 + 16; i += NT) {
				uint8_t c = 0;
				if (i < qlen) { // qr[i] = query[qlen-1-i]
					int k = qlen - 1 - i;
					c = L.qpool[(flag & KSWJ_Q_REVERSED) ? J.q_off - (uint64_t)k : J.q_off + (uint64_t)k];
				}
				QR[i] = c;
			}
			admit(T16 - 1 < 31 ? T16 - 1 : 31);
			STATE_SYNC();

			int last_st = -1, last_en = -1, H0 = 0, last_H0_t = 0;
			const int n_rows = qlen + tlen - 1;
			for (int r = 0; r < n_rows; ++r) {
				const RowIv iv = row_interval(r, qlen, tlen, w);
				const int st0 = iv.st0, en0 = iv.en0, st = iv.st, en = iv.en;
				if (st0 > en0) { ez.zdropped = 1; break; }
				if (en + 16 > frontier && frontier < T16 - 1) { // the window moves on (a wave-uniform condition)
					admit(en + 16 < T16 - 1 ? en + 16 : T16 - 1);
					STATE_SYNC();
				}
				// boundary values (:148-163)
				const int bnd = SINGLE ? (r ? q : 0) : r == 0 ? nqe : r < long_thres ? sx8(-e) : r == long_thres ? sx8(long_diff) : SPLICE ? 0 : sx8(-e2);
				const int init1 = SINGLE ? 0 : nqe, init2 = SINGLE ? 0 : nqe2; // value of a state byte that was never computed
				int x1 = init1, x21 = init2, v1 = st > 0 ? init1 : bnd;
				if (st > 0 && st - 1 >= last_st && st - 1 <= last_en) {
					uint32_t a = A[(st - 1) & RM], b = B[(st - 1) & RM];
					x1 = sx8(a >> 16), v1 = sx8(a >> 8), x21 = sx8(b);
				}
				// score differences as they read back from a state dword: signed bytes (dual-affine) or unsigned bytes minus (q+e) (single, :236-262)
				auto du = [&](uint32_t w) { return SINGLE ? (int)(w & 0xff) - qe : sx8((int)w); };
				auto dv = [&](uint32_t w) { return SINGLE ? (int)(w >> 8 & 0xff) - qe : sx8((int)(w >> 8)); };
				const int zd_e = SINGLE ? e : SPLICE ? 0 : e2, h00 = SINGLE ? qe : qe_in;
				// Substitution scores (:165-184): the reference fills s[] in 16-byte chunks from st0 BEFORE the sweep -- total bytes, up to 15 past
				// en0; positions of [st, en] outside that stretch keep what an earlier row left there (they are lanes of the block-aligned
				// interval that hold no valid cell, but valid cells of later rows read what they compute).  Round 4: a lane of the sweep computes the score
				// of its own position when it lies in the row's stretch and carries it in the state dword it is about to store (no separate pass, no
				// barrier in between); positions of the stretch beyond en get theirs from the wave that sweeps the top chunk.  Past T16 the reference
				// reads the start of qr[] as target bytes and writes into the first bytes of the target copy (its arrays are contiguous: s | sf | qr;
				// such a write only matters while position idx-T16 still owns its slot): those rows -- a job's last few -- keep the separate pass, made
				// by the team's first wave in the one-wave order (chunks ascending, a step's loads before its stores).
				const int qoff = qlen - 1 - r;
				const bool generic_sc = (flag & KSW_GENERIC_SC) != 0;
				const int total = ((en0 - st0) / 16 + 1) * 16;
				const bool sep_fill = !generic_sc && st0 + total > T16;
				const int f1 = generic_sc ? en0 + 1 : st0 + total; // scores of [st0, f1) are fresh this row
				if (sep_fill) {
					if (twave == 0)
						for (int i0 = 0; i0 < total; i0 += 64) {
							const int i = i0 + lane, idx = st0 + i;
							int sc = 0;
							if (i < total) {
								const int a = idx < T16 ? TG[idx & RM] : QR[idx - T16], b = QR[qoff + idx];
								sc = (a == m - 1 || b == m - 1) ? sc_N : a == b ? sc_mch : sc_mis;
							}
							MM2_LOCKSTEP(); // the overshoot below writes into the target copy other lanes of this step have just read
							if (i < total) {
								if (idx < T16) ((uint8_t *)&B[idx & RM])[2] = (uint8_t)sc;
								else if (frontier < idx - T16 + RS) TG[(idx - T16) & RM] = (uint8_t)sc;
							}
						}
					STATE_SYNC();
				}
				auto fresh_score = [&](int t) -> int {
					const int a = TG[t & RM], b = QR[qoff + t];
					return generic_sc ? (int)L.sc.mat[a * m + b] : (a == m - 1 || b == m - 1) ? sc_N : a == b ? sc_mch : sc_mis;
				};
				// one sweep over [st,en], highest chunks first so that lane t still sees row r-1 at t-1: TEAM chunks per round, one per wave.  The exact
				// row maximum (:325-365) rides along: H(t) += v(t) for st0 <= t < en0 as the lane stores v, H(en0) from its left neighbour's old H and u,
				// every lane keeps its best (score, rank in the reference's scan order) key.
				uint8_t *pr = dir + (size_t)r * ncol;
				const int n_chunk = (en - st + 64) >> 6;
				const bool exact_max = !approx_max;
				const int en1 = st0 + (en0 - st0) / 4 * 4, nq = (en1 - st0) >> 2; // the reference's 4-lane strided scan of [st0,en1), then the tail [en1,en0); en0 itself first
				long long best = INT64_MIN; // below every key
				for (int c_hi = n_chunk - 1; c_hi >= 0; c_hi -= TEAM) {
					const int c = c_hi - twave;
					const int t = c >= 0 ? st + (c << 6) + lane : en + 1;
					const int tk = t & RM;
					uint32_t a_cur = 0, b_cur = 0;
					int xt1 = x1, vt1 = v1, x2t1 = x21, h_cur = 0, h_left = 0;
					if (t <= en) {
						a_cur = A[tk], b_cur = B[tk];
						if (t == r) { // the row's border column (:148-163): u[r], y[r] (and y2[r]) take their border values
							a_cur = (a_cur & 0x00ffff00u) | (uint32_t)(bnd & 0xff) | (uint32_t)(init1 & 0xff) << 24;
							if (!SPLICE) b_cur = (b_cur & 0xffff00ffu) | (uint32_t)(init2 & 0xff) << 8;
						}
						if (t > st) {
							const uint32_t a_prev = A[(t - 1) & RM], b_prev = B[(t - 1) & RM];
							xt1 = sx8(a_prev >> 16), vt1 = sx8(a_prev >> 8), x2t1 = sx8(b_prev);
						}
						if (!sep_fill && t >= st0 && t < f1) b_cur = (b_cur & 0xff00ffffu) | (uint32_t)(fresh_score(t) & 0xff) << 16;
						if (exact_max) { h_cur = H[tk]; if (t == en0 && en0 > 0) h_left = H[(t - 1) & RM]; }
					}
					if (TEAM > 1) __syncthreads(); // every lane of the round has read row r-1 (its own position and its left neighbour's) before any lane stores row r
					else MM2_LOCKSTEP();
					if (t <= en) {
						uint32_t a_new = 0;
						if (SINGLE) { // ksw2_extz2_sse.c:34-55 with the left/right variants at :186-204 / :213-231 (and :164-170 score-only)
							const int ut = sx8(a_cur), yt = sx8(a_cur >> 24);
							int z = sx8(sx8(b_cur >> 16) + qe2s);
							int a = sx8(xt1 + vt1), b = sx8(yt + ut), d = 0;
							if (!with_cigar) z = z > a ? z : a;
							else if (!right) { d = a > z ? 1 : 0; z = z > a ? z : a; d = b > z ? 2 : d; }
							else { d = z > a ? 0 : 1; z = z > a ? z : a; d = z > b ? d : 2; }
							int zu = (z & 0xff) > (b & 0xff) ? (z & 0xff) : (b & 0xff); // unsigned maximum, then unsigned clamp
							zu = zu < max_scu ? zu : max_scu;
							const int un = zu - vt1, vn = zu - ut;
							z = sx8(zu - q);
							a = sx8(a - z), b = sx8(b - z);
							int xn, yn;
							if (!with_cigar || !right) {
								xn = a > 0 ? a : 0; d |= a > 0 ? 0x08 : 0;
								yn = b > 0 ? b : 0; d |= b > 0 ? 0x10 : 0;
							} else {
								xn = 0 > a ? 0 : a; d |= 0 > a ? 0 : 0x08;
								yn = 0 > b ? 0 : b; d |= 0 > b ? 0 : 0x10;
							}
							a_new = pack4(un, vn, xn, yn); A[tk] = a_new;
							if (with_cigar) pr[t - st] = (uint8_t)d;
						} else if (SPLICE) { // ksw2_exts2_sse.c:37-64 with the variants at :283-285 (score only), :312-348 (left), :355-392 (right)
							const int ut = sx8(a_cur), yt = sx8(a_cur >> 24), dn = sx8(b_cur >> 8), ac = sx8(b_cur >> 24);
							int z = sx8(b_cur >> 16);
							int a = sx8(xt1 + vt1), b = sx8(yt + ut), a2 = sx8(x2t1 + vt1), d = 0;
							const int a2a = sx8(a2 + ac);
							if (!with_cigar) { z = z > a ? z : a; z = z > b ? z : b; z = z > a2a ? z : a2a; }
							else if (!right) {
								d = a > z ? 1 : 0;    z = z > a ? z : a;
								d = b > z ? 2 : d;    z = z > b ? z : b;
								d = a2a > z ? 3 : d;  z = z > a2a ? z : a2a;
							} else {
								d = z > a ? 0 : 1;    z = z > a ? z : a;
								d = z > b ? d : 2;    z = z > b ? z : b;
								d = z > a2a ? d : 3;  z = z > a2a ? z : a2a;
							}
							const int un = z - vt1, vn = z - ut;
							const int tmp = sx8(z - q);
							a = sx8(a - tmp), b = sx8(b - tmp), a2 = sx8(a2 - sx8(z - q2));
							int xn, yn, x2n;
							if (!with_cigar || !right) {
								xn = (a > 0 ? a : 0) - qe;      d |= a > 0 ? 0x08 : 0;
								yn = (b > 0 ? b : 0) - qe;      d |= b > 0 ? 0x10 : 0;
								x2n = (a2 > dn ? a2 : dn) - q2; d |= a2 > dn ? 0x20 : 0;
							} else {
								xn = (a > 0 ? a : 0) - qe;      d |= a >= 0 ? 0x08 : 0;
								yn = (b > 0 ? b : 0) - qe;      d |= b >= 0 ? 0x10 : 0;
								x2n = (a2 > dn ? a2 : dn) - q2; d |= a2 >= dn ? 0x20 : 0;
							}
							a_new = pack4(un, vn, xn, yn); A[tk] = a_new;
							B[tk] = (b_cur & 0xffffff00u) | (uint32_t)(x2n & 0xff);
							if (with_cigar) pr[t - st] = (uint8_t)d;
						} else {
							const int ut = sx8(a_cur), yt = sx8(a_cur >> 24), y2t = sx8(b_cur >> 8);
							int z = sx8(b_cur >> 16);
							int a = sx8(xt1 + vt1), b = sx8(yt + ut), a2 = sx8(x2t1 + vt1), b2 = sx8(y2t + ut), d;
							if (!right) { // strictly greater wins (:235-243)
								d = a > z ? 1 : 0;   z = z > a ? z : a;
								d = b > z ? 2 : d;   z = z > b ? z : b;
								d = a2 > z ? 3 : d;  z = z > a2 ? z : a2;
								d = b2 > z ? 4 : d;  z = z > b2 ? z : b2;
							} else {      // ties go to the gap state (:282-290)
								d = z > a ? 0 : 1;   z = z > a ? z : a;
								d = z > b ? d : 2;   z = z > b ? z : b;
								d = z > a2 ? d : 3;  z = z > a2 ? z : a2;
								d = z > b2 ? d : 4;  z = z > b2 ? z : b2;
							}
							z = z < sc_mch ? z : sc_mch;
							const int un = z - vt1, vn = z - ut;
							int tmp = sx8(z - q);   a = sx8(a - tmp),   b = sx8(b - tmp);
							tmp = sx8(z - q2);      a2 = sx8(a2 - tmp), b2 = sx8(b2 - tmp);
							int xn, yn, x2n, y2n;
							if (!right) {
								xn = (a > 0 ? a : 0) - qe;     d |= a > 0 ? 0x08 : 0;
								yn = (b > 0 ? b : 0) - qe;     d |= b > 0 ? 0x10 : 0;
								x2n = (a2 > 0 ? a2 : 0) - qe2; d |= a2 > 0 ? 0x20 : 0;
								y2n = (b2 > 0 ? b2 : 0) - qe2; d |= b2 > 0 ? 0x40 : 0;
							} else {
								xn = (a > 0 ? a : 0) - qe;     d |= a >= 0 ? 0x08 : 0;
								yn = (b > 0 ? b : 0) - qe;     d |= b >= 0 ? 0x10 : 0;
								x2n = (a2 > 0 ? a2 : 0) - qe2; d |= a2 >= 0 ? 0x20 : 0;
								y2n = (b2 > 0 ? b2 : 0) - qe2; d |= b2 >= 0 ? 0x40 : 0;
							}
							a_new = pack4(un, vn, xn, yn); A[tk] = a_new;
							B[tk] = (b_cur & 0xffff0000u) | (uint32_t)(x2n & 0xff) | (uint32_t)(y2n & 0xff) << 8;
							if (with_cigar) pr[t - st] = (uint8_t)d;
						}
						if (SINGLE && !sep_fill && t >= st0 && t < f1) ((uint8_t *)&B[tk])[2] = (uint8_t)(b_cur >> 16); // (the single-affine cell stores no second state dword)
						if (exact_max && t >= st0 && t <= en0) {
							int h, rank;
							if (t == en0) { // candidate 0 of the scan; for r == 0 the first cell's own score (:361-364)
								h = r == 0 ? dv(a_new) - h00 : en0 > 0 ? h_left + du(a_new) : h_cur + dv(a_new);
								rank = 0;
								s_team[1] = h; // H(en0) for the end-of-target / end-of-query bookkeeping below
								if (st0 == en0) s_team[2] = h;
							} else {
								h = h_cur + dv(a_new);
								const int k = t - st0;
								rank = t < en1 ? 1 + (k & 3) * (nq + 1) + (k >> 2) : 1 + 4 * (nq + 1) + (t - en1);
								if (t == st0) s_team[2] = h;
							}
							H[tk] = h;
							const long long key = (long long)h << 32 | (long long)(0x7fffffff - rank);
							best = key > best ? key : best;
						}
					}
					if (!sep_fill && c == n_chunk - 1) // the stretch's scores beyond en: read by later rows only
						for (int p2 = en + 1 + lane; p2 < f1; p2 += 64) ((uint8_t *)&B[p2 & RM])[2] = (uint8_t)fresh_score(p2);
				}
				if (exact_max) { // the lanes' keys -> the row's maximum and where the reference's scan finds it
					best = wave_max_i64(best);
					if (TEAM > 1 && lane == 0) s_best[twave] = best;
				}
				STATE_SYNC(); // row r is complete: state, H, the team's keys
				if (exact_max) {
					if (TEAM > 1) {
						best = s_best[0];
#pragma unroll
						for (int k = 1; k < TEAM; ++k) best = s_best[k] > best ? s_best[k] : best;
					}
					const int max_H = (int)(best >> 32);
					int max_t;
					{
						const int rank = 0x7fffffff - (int)(best & 0x7fffffffLL);
						if (rank == 0) max_t = en0;
						else if (rank < 1 + 4 * (nq + 1)) { const int k = rank - 1; max_t = st0 + (k % (nq + 1)) * 4 + k / (nq + 1); }
						else max_t = en1 + (rank - 1 - 4 * (nq + 1));
					}
					const int Hen0 = s_team[1], Hst0 = s_team[2];
					if (en0 == tlen - 1 && Hen0 > ez.mte) ez.mte = Hen0, ez.mte_q = r - en0;
					if (r - st0 == qlen - 1 && Hst0 > ez.mqe) ez.mqe = Hst0, ez.mqe_t = st0;
					if (zdrop_test(ez, max_H, r, max_t, J.zdrop, zd_e)) break;
					if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = Hen0;
				} else { // follow one cell (:366-383)
					if (r > 0) {
						if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
							const int d0 = dv(A[last_H0_t & RM]), d1 = du(A[(last_H0_t + 1) & RM]);
							if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
						} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += dv(A[last_H0_t & RM]);
						else ++last_H0_t, H0 += du(A[last_H0_t & RM]);
					} else H0 = dv(A[0]) - h00, last_H0_t = 0;
					// the single-affine code tests the drop only from the second anti-diagonal on (ksw2_extz2_sse.c:291 sits inside r > 0)
					if ((flag & KSW_APPROX_DROP) && (!SINGLE || r > 0) && zdrop_test(ez, H0, r, last_H0_t, J.zdrop, zd_e)) break;
					if (r == qlen + tlen - 2 && en0 == tlen - 1) ez.score = H0;
				}
				last_st = st, last_en = en;
			}
			// ---- traceback (:385-399) ----
			if (with_cigar) {
				__threadfence_block(); // the direction bytes were written by all lanes of this wave (team: of this workgroup)
				if (TEAM > 1) __syncthreads();
				if (!ez.zdropped && (flag & KSW_EXTZ_ONLY) && ez.mqe + J.end_bonus > ez.max) ez.reach_end = 1;
				const int min_intron = SPLICE ? long_thres : 0;
				if (twave == 0) { // the team's first wave walks back and packs; the others wait at the barrier below
					if (!ez.zdropped && !(flag & KSW_EXTZ_ONLY)) traceback(dir, ncol, qlen, tlen, w, tlen - 1, qlen - 1, min_intron, g, lane);
					else if (ez.reach_end) traceback(dir, ncol, qlen, tlen, w, ez.mqe_t, qlen - 1, min_intron, g, lane);
					else if (ez.max_t >= 0 && ez.max_q >= 0) traceback(dir, ncol, qlen, tlen, w, ez.max_t, ez.max_q, min_intron, g, lane);
					if (lane == 0 && g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
					// pack the CIGAR into the pool: forward order unless the caller asked for the traceback order (:153-155 of ksw2.h)
					const int n_cig = __builtin_amdgcn_readfirstlane(g.n);
					cig_off = __builtin_amdgcn_readfirstlane(cig_off);
					__threadfence_block();
					if (n_cig > 0) {
						if ((unsigned long long)cig_off + (unsigned)n_cig > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
						else {
							const bool keep_order = flag & KSW_REV_CIGAR;
							for (int k = lane; k < n_cig; k += 64) L.cigar_pool[cig_off + k] = g.c[keep_order ? k : n_cig - 1 - k];
						}
					}
				}
			}
			STATE_SYNC();
		}
		if (tid == 0) {
			KswRes R;
			R.max = ez.max, R.zdropped = ez.zdropped, R.max_q = ez.max_q, R.max_t = ez.max_t;
			R.mqe = ez.mqe, R.mqe_t = ez.mqe_t, R.mte = ez.mte, R.mte_q = ez.mte_q;
			R.score = ez.score, R.n_cigar = g.n, R.reach_end = ez.reach_end, R.cigar_off = cig_off;
			R.zd_max = KSW_ZD_NONE, R.zd_t0 = R.zd_t1 = R.zd_q0 = R.zd_q1 = -1;
			L.res[jid] = R;
		}
	}
}

#undef STATE_SYNC

namespace {
template <bool LDS_STATE, int MODE, int TEAM>
void launch_mode(const KswLaunch &L, int n_blocks, int waves_per_block, size_t lds, hipStream_t stream)
{
	if (lds > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void *)ksw_extd2_kernel<LDS_STATE, MODE, TEAM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipLaunchKernelGGL((ksw_extd2_kernel<LDS_STATE, MODE, TEAM>), dim3(n_blocks), dim3(64 * (TEAM > 1 ? TEAM : waves_per_block)), lds, stream, L);
	HIP_CHECK(hipGetLastError());
}
template <bool LDS_STATE, int TEAM>
void launch_any(const KswLaunch &L, int n_blocks, int waves_per_block, size_t lds, hipStream_t stream)
{
	if (L.splice) launch_mode<LDS_STATE, 2, TEAM>(L, n_blocks, waves_per_block, lds, stream);
	else if (L.single_affine) launch_mode<LDS_STATE, 1, TEAM>(L, n_blocks, waves_per_block, lds, stream);
	else launch_mode<LDS_STATE, 0, TEAM>(L, n_blocks, waves_per_block, lds, stream);
}
}

// team: wavefronts per job (1: n_slots waves, waves_per_block of them in a block, each with a job of its own; 4 / 8: n_slots workgroups of one job each)
void ksw_extd2_launch(const KswLaunch &L, int n_slots, int waves_per_block, int team, void *stream)
{
	if (L.n_jobs <= 0) return;
	const size_t region = (ksw_lds_per_wave(L.ring, L.max_Q16) + 15) / 16 * 16;
	if (L.state_pool) { // state in HBM: any job length, a wave per job
		launch_any<false, 1>(L, (n_slots + waves_per_block - 1) / waves_per_block, waves_per_block, 0, (hipStream_t)stream);
		return;
	}
	const size_t lds = team > 1 ? region : region * waves_per_block;
	const int n_blocks = team > 1 ? n_slots : (n_slots + waves_per_block - 1) / waves_per_block;
	if (lds > 160 * 1024) throw std::runtime_error("[mm2amd] ksw_extd2: job class does not fit LDS");
	if (team == 8) launch_any<true, 8>(L, n_blocks, 1, lds, (hipStream_t)stream);
	else if (team == 4) launch_any<true, 4>(L, n_blocks, 1, lds, (hipStream_t)stream);
	else launch_any<true, 1>(L, n_blocks, waves_per_block, lds, (hipStream_t)stream);
}

} // namespace mm2amd

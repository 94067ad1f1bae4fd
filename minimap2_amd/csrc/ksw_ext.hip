// Register-resident DP for the EXTENSION calls of long-read mapping on gfx950: ksw_extd2_sse (ksw2_extd2_sse.c:34-401) as mm_align1 calls it
// from a chain's first and last anchor outwards (align.c:791, :883: KSW_EZ_EXTZ_ONLY, left extensions also KSW_EZ_RIGHT | KSW_EZ_REV_CIGAR).
//
// Two extensions per read, small (a 10 kb ONT read's flanks: query ~50, target ~100) but with everything the gap fills do not need: the
// EXACT maximum of every anti-diagonal with the reference's tie order (:325-358), the best scores on the last query row and target column,
// the Z-drop test on every row (ksw2.h:171-187), the end bonus, tracebacks that start at the best cell, right-aligned gaps.  They used to
// take the lane-exact kernel (ksw_extd2.hip): 3 % of the DP cells, a quarter of the kernel time.  When the band cannot bind (w + 1 >=
// max(qlen, tlen): ksw_host.cpp, band_cannot_bind) only valid cells matter (ksw_gapfill.hip explains why), so they run here on the register-resident layout instead: lane = target
// column, four or eight register sets of 64 columns (targets up to 256 / 512), two jobs per wavefront in the halves of packed 16-bit registers, the same
// cell arithmetic (gf_cell; gf_cell_right for KSW_EZ_RIGHT) and direction dwords.  On top of it every column keeps its cell's score H as a
// 32-bit register per job: H += v down a column, H(left neighbour, row before) + u where a column starts (:329-357) -- mathematically the
// scores the reference recovers from its difference arrays; the row maximum is a DPP reduction, its position is picked among the lanes that
// hold it in the reference's scan order.  A job that Z-drops keeps computing with its results frozen (its partner may still run).
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"
#include "ksw_gapfill_dev.hpp"

namespace mm2amd {

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

constexpr int EX_QCAP = 512; // queries up to EX_QCAP; targets up to 64 * NC columns (NC register sets: 4 or 8)

// (gf_cell_right, the cell with right-aligned gaps: ksw_gapfill_dev.hpp -- shared with ksw_extq.hip)

struct ExtState { int max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, done; };

template <bool RIGHT, int NC>
__global__ void __launch_bounds__(256, NC > 4 ? 2 : 4) ksw_ext_kernel(KswLaunch L) // (eight sets: ~190 VGPRs, two workgroups per CU)
{
	__shared__ uint8_t s_q[4][2][EX_QCAP]; // query bytes of the pair
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	const int slot = blockIdx.x * 4 + wave_in_block;
	const int m = L.sc.m;
	int q = L.sc.q, e = L.sc.e, q2 = L.sc.q2, e2 = L.sc.e2;
	const int qe_in = q + e; // before the swap (ksw2_extd2_sse.c:68 vs :78): seeds H(0,0)
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t; t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2, nqe = -qe, nqe2 = -qe2;
	const int sc_mch = L.sc.mat[0], sc_mis = L.sc.mat[1];
	const int sc_N = L.sc.mat[m * m - 1] == 0 ? -e2 : L.sc.mat[m * m - 1];
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const uint32_t S_MISD = pk2(sc_mis - sc_mch), S_SCN = pk2(sc_N), S_Q = pk2(q), S_Q2 = pk2(q2), S_QE = pk2(qe), S_QE2 = pk2(qe2);
	const uint32_t S_NQE = pk2(nqe), S_NQE2 = pk2(nqe2);
	const uint32_t P_MCH = pk2v(sc_mch);
	const uint32_t lane4 = (uint32_t)lane * 4u;
	uint8_t *const qb = s_q[wave_in_block][0];
	const uint8_t *const s_qflat = &s_q[0][0][0];
	const int qb_addr = wave_in_block * 2 * EX_QCAP;
	const uint32_t qb_last = (uint32_t)(qb_addr + EX_QCAP - 1);

	for (;;) {
		int pid = 0;
		if (lane == 0) pid = atomicAdd(L.counter, 1);
		pid = __builtin_amdgcn_readfirstlane(pid);
		if (2 * pid >= L.n_jobs) break;
		const int jid[2] = { 2 * pid, 2 * pid + 1 };
		const bool hasB = jid[1] < L.n_jobs;
		const KswJob JA = L.jobs[jid[0]], JB = L.jobs[hasB ? jid[1] : jid[0]];
		const int qlen[2] = { JA.qlen, hasB ? JB.qlen : 0 }, tlen[2] = { JA.tlen, hasB ? JB.tlen : 0 };
		const int zdrop[2] = { JA.zdrop, JB.zdrop }, end_bonus[2] = { JA.end_bonus, JB.end_bonus };
		const int n_rows_h[2] = { qlen[0] + tlen[0] - 1, hasB ? qlen[1] + tlen[1] - 1 : 0 };
		const int tmax = tlen[0] > tlen[1] ? tlen[0] : tlen[1];
		const int ncol = (tmax + 63) & ~63; // columns of the shared direction matrix: dword (r >> 1) * ncol + t = [row r: A, B][row r + 1: A, B]
		uint8_t *const dir = L.dir_pool + (size_t)(2 * slot) * L.slot_bytes;
		for (int i = lane; i < qlen[0]; i += 64) qb[i] = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)i : JA.q_off + (uint64_t)i];
		for (int i = lane; i < qlen[1]; i += 64) qb[EX_QCAP + i] = L.qpool[(JB.flag & KSWJ_Q_REVERSED) ? JB.q_off - (uint64_t)i : JB.q_off + (uint64_t)i];
		WAVE_SYNC();

		uint32_t T[NC], U[NC], V[NC], X[NC], Y[NC], X2[NC], Y2[NC], DE[NC];
		int32_t H[2][NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int t = c * 64 + lane;
			uint32_t bA = 4, bB = 4;
			if (t < tlen[0]) {
				const uint64_t pos = (JA.flag & KSWJ_T_REVERSED) ? JA.t_off - (uint64_t)t : JA.t_off + (uint64_t)t;
				bA = (JA.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
			}
			if (t < tlen[1]) {
				const uint64_t pos = (JB.flag & KSWJ_T_REVERSED) ? JB.t_off - (uint64_t)t : JB.t_off + (uint64_t)t;
				bB = (JB.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
			}
			T[c] = bA | bB << 16;
			U[c] = V[c] = X[c] = Y[c] = X2[c] = Y2[c] = DE[c] = 0u;
			H[0][c] = H[1][c] = 0;
		}
		ExtState ez[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			ez[h].max = 0, ez[h].zdropped = 0, ez[h].max_q = ez[h].max_t = ez[h].mqe_t = ez[h].mte_q = -1;
			ez[h].mqe = ez[h].mte = ez[h].score = KSW_NEG_INF, ez[h].done = n_rows_h[h] <= 0;
		}
		const int n_rows = n_rows_h[0] > n_rows_h[1] ? n_rows_h[0] : n_rows_h[1];

		for (int r0 = 0; r0 < n_rows && !(ez[0].done && ez[1].done); r0 += 2) {
			int lo2, hi2; // union of the two rows' valid cells over both jobs
			{
				const int stA = r0 - qlen[0] + 1 > 0 ? r0 - qlen[0] + 1 : 0, enA = r0 + 1 < tlen[0] - 1 ? r0 + 1 : tlen[0] - 1;
				const int stB = r0 - qlen[1] + 1 > 0 ? r0 - qlen[1] + 1 : 0, enB = r0 + 1 < tlen[1] - 1 ? r0 + 1 : tlen[1] - 1;
				const bool okA = r0 < n_rows_h[0] && stA <= enA, okB = r0 < n_rows_h[1] && stB <= enB;
				lo2 = okA ? (okB && stB < stA ? stB : stA) : okB ? stB : 1;
				hi2 = okA ? (okB && enB > enA ? enB : enA) : okB ? enB : 0;
			}
			uint32_t *const prow = (uint32_t *)(dir + (size_t)(r0 >> 1) * (size_t)ncol * 4u);
#pragma unroll
			for (int par = 0; par < 2; ++par) {
				const int r = r0 + par;
				const int bnd = r == 0 ? nqe : r < long_thres ? -e : r == long_thres ? long_diff : -e2; // v[-1] / u[r] on the matrix border (:148-163)
				const uint32_t S_BND = pk2(bnd);
				const bool topA = r < tlen[0] && r < n_rows_h[0], topB = r < tlen[1] && r < n_rows_h[1]; // the anti-diagonal still starts a new column (t = r)
				const int edge_set = r >> 6, edge_lane = r & 63;
				const uint32_t edge_halves = (topA ? 0xffffu : 0u) | (topB ? 0xffff0000u : 0u);
#pragma unroll
				for (int c = NC - 1; c >= 0; --c) { // from the highest set down: set c still sees row r - 1 in set c - 1
					if (c * 64 > hi2 || c * 64 + 63 < lo2) continue;
					uint32_t cV = S_BND, cX = S_NQE, cX2 = S_NQE2;
					if (c > 0) cV = gf_ror1(V[c - 1]), cX = gf_ror1(X[c - 1]), cX2 = gf_ror1(X2[c - 1]);
					const uint32_t vp = dpp_shr1u(cV, V[c]), xp = dpp_shr1u(cX, X[c]), x2p = dpp_shr1u(cX2, X2[c]);
					if (edge_halves && edge_set == c) { // u[r], y[r], y2[r] take their border values on first use (:156-163)
						const uint32_t em = lane == edge_lane ? edge_halves : 0u;
						U[c] = bfi(em, S_BND, U[c]), Y[c] = bfi(em, S_NQE, Y[c]), Y2[c] = bfi(em, S_NQE2, Y2[c]);
					}
					uint32_t qa = (uint32_t)(qb_addr + r - c * 64) - (uint32_t)lane; // query position of this column's cell, as an LDS address (clamped: dead cells)
					qa = qa < qb_last ? qa : qb_last;
					const uint32_t qv = (uint32_t)s_qflat[qa] | (uint32_t)s_qflat[qa + EX_QCAP] << 16, tv = T[c];
					uint32_t d;
					if (RIGHT) gf_cell_right(tv ^ qv, tv | qv, xp, vp, x2p, U[c], V[c], X[c], Y[c], X2[c], Y2[c], d, P_MCH, S_MISD, S_SCN, S_Q, S_Q2, S_QE, S_QE2);
					else gf_cell(tv ^ qv, tv | qv, xp, vp, x2p, U[c], V[c], X[c], Y[c], X2[c], Y2[c], d, P_MCH, S_MISD, S_SCN, S_Q, S_Q2, S_QE, S_QE2);
					if (par == 0) DE[c] = d;
					else {
						const uint32_t t = (uint32_t)(c * 64 + lane);
						if (t - (uint32_t)lo2 <= (uint32_t)(hi2 - lo2))
							*(uint32_t *)((uint8_t *)prow + c * 256 + lane4) = __builtin_amdgcn_perm(d, DE[c], 0x06040200u); // [even A, even B, odd A, odd B]
					}
				}
				// ---- the row's scores, its exact maximum, the end scores and the Z-drop test, per job (ksw2_extd2_sse.c:325-365) ----
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					if (ez[h].done || r >= n_rows_h[h]) continue;
					const int st0 = r - qlen[h] + 1 > 0 ? r - qlen[h] + 1 : 0, en0 = r < tlen[h] - 1 ? r : tlen[h] - 1;
					const bool edge = r < tlen[h]; // column t = r has its first cell on this row
					int32_t hleft = 0;             // H of the cell left of it on the row before
					if (edge && r > 0) {
#pragma unroll
						for (int c = 0; c < NC; ++c) {
							if (edge_lane > 0 && c == edge_set) hleft = __builtin_amdgcn_readlane(H[h][c], edge_lane - 1);
							if (edge_lane == 0 && c + 1 == edge_set) hleft = __builtin_amdgcn_readlane(H[h][c], 63);
						}
					}
					int32_t best = INT32_MIN;
#pragma unroll
					for (int c = 0; c < NC; ++c) {
						if (c < (st0 >> 6) || c > (en0 >> 6)) continue;
						const int t = c * 64 + lane;
						const int32_t dv = (int32_t)(int16_t)(V[c] >> (16 * h)), du = (int32_t)(int16_t)(U[c] >> (16 * h));
						int32_t hv = H[h][c];
						if (edge && t == r) hv = r == 0 ? dv - qe_in : hleft + du;
						else if (t >= st0 && t <= en0) hv += dv;
						H[h][c] = hv;
						if (t >= st0 && t <= en0) best = hv > best ? hv : best;
					}
					const int32_t max_H = __builtin_amdgcn_readlane(wave_prefix_max_i32_ext(best), 63);
					// its position: among the lanes that hold it, the first in the reference's order -- en0, then the four interleaved streams
					// of the 4-lane scan over [st0, en1), then the tail [en1, en0)
					const int en1 = st0 + ((en0 - st0) & ~3), nq = (en1 - st0) >> 2;
					int best_rank = INT32_MAX, max_t = en0;
#pragma unroll
					for (int c = 0; c < NC; ++c) {
						if (c < (st0 >> 6) || c > (en0 >> 6)) continue;
						const int t = c * 64 + lane;
						unsigned long long cand = __ballot(t >= st0 && t <= en0 && H[h][c] == max_H);
						while (cand) {
							const int tt = c * 64 + (__ffsll((long long)cand) - 1);
							cand &= cand - 1;
							const int k = tt - st0;
							const int rank = tt == en0 ? 0 : tt < en1 ? 1 + (k & 3) * (nq + 1) + (k >> 2) : 1 + 4 * (nq + 1) + (tt - en1);
							if (rank < best_rank) best_rank = rank, max_t = tt;
						}
					}
					int32_t Hen = 0, Hst = 0;
#pragma unroll
					for (int c = 0; c < NC; ++c) {
						if (c == (en0 >> 6)) Hen = __builtin_amdgcn_readlane(H[h][c], en0 & 63);
						if (c == (st0 >> 6)) Hst = __builtin_amdgcn_readlane(H[h][c], st0 & 63);
					}
					ExtState &z = ez[h];
					if (en0 == tlen[h] - 1 && Hen > z.mte) z.mte = Hen, z.mte_q = r - en0;
					if (r - st0 == qlen[h] - 1 && Hst > z.mqe) z.mqe = Hst, z.mqe_t = st0;
					if (max_H > z.max) z.max = max_H, z.max_t = max_t, z.max_q = r - max_t; // ksw_apply_zdrop (ksw2.h:171-187)
					else if (max_t >= z.max_t && r - max_t >= z.max_q) {
						const int tl = max_t - z.max_t, ql = (r - max_t) - z.max_q, l = tl > ql ? tl - ql : ql - tl;
						if (zdrop[h] >= 0 && z.max - max_H > zdrop[h] + l * e2) z.zdropped = 1, z.done = 1;
					}
					if (!z.done && r == qlen[h] + tlen[h] - 2 && en0 == tlen[h] - 1) z.score = Hen;
					if (r == n_rows_h[h] - 1) z.done = 1;
				}
			}
		}
		// ---- tracebacks (ksw2_extd2_sse.c:385-399; ksw_backtrack, every cell inside the matrix): from the last query row when the end bonus
		//      makes reaching the end the better alignment, else from the best cell; lanes 0-31 serve job A, lanes 32-63 job B ----
		__threadfence_block();
		const bool isB = lane >= 32;
		const int hsel = isB ? 1 : 0;
		const ExtState zz = isB ? ez[1] : ez[0];
		const int my_flag = isB ? JB.flag : JA.flag, my_qlen = isB ? qlen[1] : qlen[0];
		const bool have_job = !isB || hasB;
		const int reach_end = have_job && !zz.zdropped && zz.mqe + (isB ? end_bonus[1] : end_bonus[0]) > zz.max ? 1 : 0;
		int i = -1, j = -1;
		if (have_job) {
			if (reach_end) i = zz.mqe_t, j = my_qlen - 1;
			else if (zz.max_t >= 0 && zz.max_q >= 0) i = zz.max_t, j = zz.max_q;
		}
		FastCig g = { L.cigar_tmp + (size_t)(2 * slot + hsel) * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;
		{ // each half-wave follows its job's path (gf_traceback, ksw_gapfill_dev.hpp)
			const size_t hoff = (size_t)hsel;
			gf_traceback(i >= 0 && j >= 0, i, j, [&](int ii, int jj) {
				const int rr = ii + jj;
				return (int)dir[((size_t)(rr >> 1) * (size_t)ncol + (size_t)ii) * 4u + (size_t)((rr & 1) << 1) + hoff];
			}, g);
		}
		if ((lane & 31) == 0 && have_job) {
			if (g.n > 0) g.c[g.n - 1] = g.last;
			if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
		}
		WAVE_SYNC();
		__threadfence_block();
#pragma unroll
		for (int which = 0; which < 2; ++which) { // the CIGARs into the pool: forward order unless the caller asked for the traceback's (KSW_EZ_REV_CIGAR, ksw2.h:153-155)
			if (which == 1 && !hasB) break;
			const int src = which * 32;
			const int n_cig = __builtin_amdgcn_readlane(g.n, src);
			const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)cig_off, src);
			const bool keep_order = ((which ? JB.flag : JA.flag) & KSW_REV_CIGAR) != 0;
			const uint32_t *tmpc = L.cigar_tmp + (size_t)(2 * slot + which) * L.cigar_tmp_cap;
			if (n_cig > 0) {
				if ((unsigned long long)off + (unsigned)n_cig > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
				else for (int k = lane; k < n_cig; k += 64) L.cigar_pool[off + k] = tmpc[keep_order ? k : n_cig - 1 - k];
			}
		}
		if ((lane & 31) == 0 && have_job) {
			KswRes R;
			R.max = zz.max, R.zdropped = zz.zdropped, R.max_q = zz.max_q, R.max_t = zz.max_t, R.mqe = zz.mqe, R.mqe_t = zz.mqe_t;
			R.mte = zz.mte, R.mte_q = zz.mte_q, R.score = zz.score, R.n_cigar = g.n, R.reach_end = reach_end, R.cigar_off = cig_off;
			R.zd_max = KSW_ZD_NONE, R.zd_t0 = R.zd_t1 = R.zd_q0 = R.zd_q1 = -1;
			L.res[isB ? jid[1] : jid[0]] = R;
		}
		(void)my_flag;
		WAVE_SYNC();
	}
}

void ksw_ext_launch(const KswLaunch &L, int n_slots, bool right, int n_sets, void *stream)
{
	if (L.n_jobs <= 0) return;
	const int n_blocks = (n_slots + 3) / 4;
	hipStream_t s = (hipStream_t)stream;
	if (n_sets <= 4) {
		if (right) hipLaunchKernelGGL((ksw_ext_kernel<true, 4>), dim3(n_blocks), dim3(256), 0, s, L);
		else hipLaunchKernelGGL((ksw_ext_kernel<false, 4>), dim3(n_blocks), dim3(256), 0, s, L);
	} else {
		if (right) hipLaunchKernelGGL((ksw_ext_kernel<true, 8>), dim3(n_blocks), dim3(256), 0, s, L);
		else hipLaunchKernelGGL((ksw_ext_kernel<false, 8>), dim3(n_blocks), dim3(256), 0, s, L);
	}
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

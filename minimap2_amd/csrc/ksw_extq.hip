// The extension DP with the QUERY across the lanes (round 6): ksw_ext.hip transposed.
//
// mm_align1 gives an end extension a target window about twice as long as the query it extends (align.c:716-718: l query bases may span l + (l a - q) / e
// reference bases), so the layout of ksw_ext.hip -- lane = target column, 64 columns per register set -- needs twice the register sets the query would: a
// 200 x 400 extension sweeps eight sets per anti-diagonal, and everything with a target beyond 512 (a query beyond ~256) went to the lane-exact kernel, whose
// launches of a few hundred long jobs were the largest un-overlapped kernel time of a step after the banded gap fill (124 ms, for 3 % of the cells).  Here
// lane = QUERY position, NC register sets of 64 (queries up to 128 / 256 / 512), the target streams through the lanes:
//
//   * cell (i, j) of anti-diagonal r = i + j sits in lane j.  Its left neighbour (i - 1, j) is the lane's OWN cell of the row before: (x, v, x2) stay where they
//     are; its upper neighbour (i, j - 1) is lane j - 1's: (u, y, y2) arrive by one DPP wave shift each (lane 0 of a set from lane 63 of the set below, lane 0 of
//     the first set the matrix border u[i] of :156-163).  A lane's target base is the base lane j - 1 had the row before: the fourth DPP shift; lane 0 takes
//     target[r] from LDS.  The query base of a lane never changes.
//   * the lane whose first cell lies on this row (j = r, i = 0) takes its (x, v, x2) from the matrix border (:148-155) -- one bit-field insert per register, in the
//     one register set the lane is in;
//   * every lane keeps its cell's score H (32 bits per job): H += u along a query row, H(lane j - 1, row before) + v where a lane starts -- the scores the
//     reference recovers from its difference arrays (:329-357).  The exact row maximum, its position in the reference's scan order (:325-358), the end scores, the
//     Z-drop test (ksw2.h:171-187), the end bonus and the traceback are ksw_ext.hip's, with t = r - j.
//
// Eligibility is ksw_ext.hip's (extension flags, default scores, dual affine costs, a band that cannot bind: only valid cells matter) with the query instead of
// the target bounded by the register sets and the target by the LDS array (EQ_TCAP).  tests/test_gpu_ksw.py runs every flag combination, every preset's scoring
// and every register-set boundary through it against the lane-exact oracle and the compiled reference; MM2AMD_EXT_BY_TARGET=1 sends the jobs through ksw_ext.hip
// instead (A/B).
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"
#include "ksw_gapfill_dev.hpp"

namespace mm2amd {

#define EQ_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

constexpr int EQ_TCAP = 2048; // targets up to EQ_TCAP (4 KB of LDS per wave: both jobs' bases, A | B << 8)

struct ExtqState { int max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, done; };

template <bool RIGHT, int NC>
__global__ void __launch_bounds__(256, NC > 4 ? 2 : 4) ksw_extq_kernel(KswLaunch L)
{
	__shared__ uint16_t s_t[4][EQ_TCAP]; // target bases of the pair by target index
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
	constexpr int ncol = NC * 64; // dword (r >> 1) * ncol + j = [row r: A, B][row r + 1: A, B]
	uint16_t *const tb = &s_t[wave_in_block][0];

	for (;;) {
		int pid = 0;
		if (lane == 0) pid = atomicAdd(L.counter, 1);
		pid = __builtin_amdgcn_readfirstlane(pid);
		if (2 * pid >= L.n_jobs) break;
		const int jid[2] = { 2 * pid, 2 * pid + 1 };
		const bool hasB = jid[1] < L.n_jobs;
		const KswJob JA = L.jobs[jid[0]], JB = L.jobs[hasB ? jid[1] : jid[0]];
		const int qlen[2] = { __builtin_amdgcn_readfirstlane(JA.qlen), hasB ? __builtin_amdgcn_readfirstlane(JB.qlen) : 0 };
		const int tlen[2] = { __builtin_amdgcn_readfirstlane(JA.tlen), hasB ? __builtin_amdgcn_readfirstlane(JB.tlen) : 0 };
		const int zdrop[2] = { JA.zdrop, JB.zdrop }, end_bonus[2] = { JA.end_bonus, JB.end_bonus };
		const int n_rows_h[2] = { qlen[0] + tlen[0] - 1, hasB ? qlen[1] + tlen[1] - 1 : 0 };
		uint8_t *const dir = L.dir_pool + (size_t)(2 * slot) * L.slot_bytes;
		{ // the pair's target bases into LDS
			const int tmax = tlen[0] > tlen[1] ? tlen[0] : tlen[1];
			for (int t = lane; t < tmax; t += 64) {
				uint32_t bA = 4, bB = 4;
				if (t < tlen[0]) {
					const uint64_t pos = (JA.flag & KSWJ_T_REVERSED) ? JA.t_off - (uint64_t)t : JA.t_off + (uint64_t)t;
					bA = (JA.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
				}
				if (t < tlen[1]) {
					const uint64_t pos = (JB.flag & KSWJ_T_REVERSED) ? JB.t_off - (uint64_t)t : JB.t_off + (uint64_t)t;
					bB = (JB.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
				}
				tb[t] = (uint16_t)(bA | bB << 8);
			}
			EQ_SYNC();
		}
		uint32_t T[NC], Q[NC], U[NC], V[NC], X[NC], Y[NC], X2[NC], Y2[NC], DE[NC];
		int32_t H[2][NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = c * 64 + lane;
			uint32_t bA = 4, bB = 4;
			if (j < qlen[0]) bA = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)j : JA.q_off + (uint64_t)j];
			if (j < qlen[1]) bB = L.qpool[(JB.flag & KSWJ_Q_REVERSED) ? JB.q_off - (uint64_t)j : JB.q_off + (uint64_t)j];
			Q[c] = bA | bB << 16;
			T[c] = 0x00040004u, U[c] = V[c] = X[c] = Y[c] = X2[c] = Y2[c] = DE[c] = 0u;
			H[0][c] = H[1][c] = 0;
		}
		ExtqState ez[2];
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			ez[h].max = 0, ez[h].zdropped = 0, ez[h].max_q = ez[h].max_t = ez[h].mqe_t = ez[h].mte_q = -1;
			ez[h].mqe = ez[h].mte = ez[h].score = KSW_NEG_INF, ez[h].done = n_rows_h[h] <= 0;
		}
		const int n_rows = n_rows_h[0] > n_rows_h[1] ? n_rows_h[0] : n_rows_h[1];

		for (int r0 = 0; r0 < n_rows && !(ez[0].done && ez[1].done); r0 += 2) {
			// query rows with a cell on either anti-diagonal of the pair, over both jobs: j in [max(0, r - tlen + 1), min(qlen - 1, r)]
			int lo2, hi2;
			{
				const int loA = r0 - tlen[0] + 1 > 0 ? r0 - tlen[0] + 1 : 0, hiA = r0 + 1 < qlen[0] - 1 ? r0 + 1 : qlen[0] - 1;
				const int loB = r0 - tlen[1] + 1 > 0 ? r0 - tlen[1] + 1 : 0, hiB = r0 + 1 < qlen[1] - 1 ? r0 + 1 : qlen[1] - 1;
				const bool okA = r0 < n_rows_h[0] && loA <= hiA, okB = r0 < n_rows_h[1] && loB <= hiB;
				lo2 = okA ? (okB && loB < loA ? loB : loA) : okB ? loB : 1;
				hi2 = okA ? (okB && hiB > hiA ? hiB : hiA) : okB ? hiB : 0;
			}
			uint32_t *const prow = (uint32_t *)(dir + (size_t)(r0 >> 1) * (size_t)ncol * 4u);
#pragma unroll
			for (int par = 0; par < 2; ++par) {
				const int r = r0 + par;
				const int bnd = r == 0 ? nqe : r < long_thres ? -e : r == long_thres ? long_diff : -e2; // u[r] above the first query row / v[-1] left of the first target column (:148-163): the same function of r
				const uint32_t S_BND = pk2(bnd);
				const bool topA = r < qlen[0] && r < n_rows_h[0], topB = r < qlen[1] && r < n_rows_h[1]; // the anti-diagonal still starts a new query row (j = r, i = 0)
				const int edge_set = r >> 6, edge_lane = r & 63;
				const uint32_t edge_halves = (topA ? 0xffffu : 0u) | (topB ? 0xffff0000u : 0u);
				const uint32_t t_in = r < EQ_TCAP ? __builtin_amdgcn_perm(0u, (uint32_t)tb[r], 0x0c010c00u) : 0x00040004u; // target[r] of both jobs: enters at lane 0
#pragma unroll
				for (int c = NC - 1; c >= 0; --c) { // from the highest set down: set c still sees row r - 1 in set c - 1
					if (c * 64 > hi2 || c * 64 + 63 < lo2) continue;
					uint32_t cU = S_BND, cY = S_NQE, cY2 = S_NQE2, cT = t_in;
					if (c > 0) cU = gf_ror1(U[c - 1]), cY = gf_ror1(Y[c - 1]), cY2 = gf_ror1(Y2[c - 1]), cT = gf_ror1(T[c - 1]);
					uint32_t uu = dpp_shr1u(cU, U[c]), yy = dpp_shr1u(cY, Y[c]), yy2 = dpp_shr1u(cY2, Y2[c]);
					T[c] = dpp_shr1u(cT, T[c]);
					if (edge_halves && edge_set == c) { // the lane's first cell: (x, v, x2) of the column left of the matrix (:148-155)
						const uint32_t em = lane == edge_lane ? edge_halves : 0u;
						V[c] = bfi(em, S_BND, V[c]), X[c] = bfi(em, S_NQE, X[c]), X2[c] = bfi(em, S_NQE2, X2[c]);
					}
					const uint32_t tv = T[c], qv = Q[c];
					uint32_t d;
					if (RIGHT) gf_cell_right(tv ^ qv, tv | qv, X[c], V[c], X2[c], uu, V[c], X[c], yy, X2[c], yy2, d, P_MCH, S_MISD, S_SCN, S_Q, S_Q2, S_QE, S_QE2);
					else gf_cell(tv ^ qv, tv | qv, X[c], V[c], X2[c], uu, V[c], X[c], yy, X2[c], yy2, d, P_MCH, S_MISD, S_SCN, S_Q, S_Q2, S_QE, S_QE2);
					U[c] = uu, Y[c] = yy, Y2[c] = yy2;
					if (par == 0) DE[c] = d;
					else {
						const uint32_t j = (uint32_t)(c * 64 + lane);
						if (j - (uint32_t)lo2 <= (uint32_t)(hi2 - lo2))
							*(uint32_t *)((uint8_t *)prow + c * 256 + lane4) = __builtin_amdgcn_perm(d, DE[c], 0x06040200u); // [even A, even B, odd A, odd B]
					}
				}
				// ---- the row's scores, its exact maximum, the end scores and the Z-drop test, per job (ksw2_extd2_sse.c:325-365); t = r - j ----
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					if (ez[h].done || r >= n_rows_h[h]) continue;
					const int st0 = r - qlen[h] + 1 > 0 ? r - qlen[h] + 1 : 0, en0 = r < tlen[h] - 1 ? r : tlen[h] - 1;
					const int j_lo = r - en0, j_hi = r - st0;   // the lanes that hold the row's cells
					const bool edge = r < qlen[h];              // query row j = r has its first cell (i = 0) on this row
					int32_t hup = 0;                            // H of the cell above it on the row before: lane r - 1
					if (edge && r > 0) {
#pragma unroll
						for (int c = 0; c < NC; ++c) {
							if (edge_lane > 0 && c == edge_set) hup = __builtin_amdgcn_readlane(H[h][c], edge_lane - 1);
							if (edge_lane == 0 && c + 1 == edge_set) hup = __builtin_amdgcn_readlane(H[h][c], 63);
						}
					}
					int32_t best = INT32_MIN;
#pragma unroll
					for (int c = 0; c < NC; ++c) {
						if (c < (j_lo >> 6) || c > (j_hi >> 6)) continue;
						const int j = c * 64 + lane;
						const int32_t dv = (int32_t)(int16_t)(V[c] >> (16 * h)), du = (int32_t)(int16_t)(U[c] >> (16 * h));
						int32_t hv = H[h][c];
						if (edge && j == r) hv = r == 0 ? dv - qe_in : hup + dv;
						else if (j >= j_lo && j <= j_hi) hv += du;
						H[h][c] = hv;
						if (j >= j_lo && j <= j_hi) best = hv > best ? hv : best;
					}
					const int32_t max_H = __builtin_amdgcn_readlane(wave_prefix_max_i32_ext(best), 63);
					// its position: among the cells that hold it, the first in the reference's order over t -- en0, then the four interleaved streams of the 4-lane scan
					// over [st0, en1), then the tail [en1, en0)
					const int en1 = st0 + ((en0 - st0) & ~3), nq = (en1 - st0) >> 2;
					int best_rank = INT32_MAX, max_t = en0;
#pragma unroll
					for (int c = 0; c < NC; ++c) {
						if (c < (j_lo >> 6) || c > (j_hi >> 6)) continue;
						const int j = c * 64 + lane;
						unsigned long long cand = __ballot(j >= j_lo && j <= j_hi && H[h][c] == max_H);
						while (cand) {
							const int tt = r - (c * 64 + (__ffsll((long long)cand) - 1));
							cand &= cand - 1;
							const int k = tt - st0;
							const int rank = tt == en0 ? 0 : tt < en1 ? 1 + (k & 3) * (nq + 1) + (k >> 2) : 1 + 4 * (nq + 1) + (tt - en1);
							if (rank < best_rank) best_rank = rank, max_t = tt;
						}
					}
					int32_t Hen = 0, Hst = 0;
#pragma unroll
					for (int c = 0; c < NC; ++c) {
						if (c == (j_lo >> 6)) Hen = __builtin_amdgcn_readlane(H[h][c], j_lo & 63);
						if (c == (j_hi >> 6)) Hst = __builtin_amdgcn_readlane(H[h][c], j_hi & 63);
					}
					ExtqState &z = ez[h];
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
		// ---- tracebacks (ksw2_extd2_sse.c:385-399): from the last query row when the end bonus makes reaching the end the better alignment, else from the best cell;
		//      lanes 0-31 serve job A, lanes 32-63 job B ----
		__threadfence_block();
		const bool isB = lane >= 32;
		const int hsel = isB ? 1 : 0;
		const ExtqState zz = isB ? ez[1] : ez[0];
		const int my_qlen = isB ? qlen[1] : qlen[0];
		const bool have_job = !isB || hasB;
		const int reach_end = have_job && !zz.zdropped && zz.mqe + (isB ? end_bonus[1] : end_bonus[0]) > zz.max ? 1 : 0;
		int i = -1, j = -1;
		if (have_job) {
			if (reach_end) i = zz.mqe_t, j = my_qlen - 1;
			else if (zz.max_t >= 0 && zz.max_q >= 0) i = zz.max_t, j = zz.max_q;
		}
		FastCig g = { L.cigar_tmp + (size_t)(2 * slot + hsel) * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;
		{
			const uint32_t hoff = (uint32_t)hsel;
			gf_traceback(i >= 0 && j >= 0, i, j, [&](int ii, int jj) {
				const int rr = ii + jj;
				return (int)dir[((size_t)(rr >> 1) * (size_t)ncol + (size_t)jj) * 4u + (size_t)((rr & 1) << 1) + hoff];
			}, g);
		}
		if ((lane & 31) == 0 && have_job) {
			if (g.n > 0) g.c[g.n - 1] = g.last;
			if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
		}
		EQ_SYNC();
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
		EQ_SYNC();
	}
}

void ksw_extq_launch(const KswLaunch &L, int n_slots, bool right, int n_sets, void *stream)
{
	if (L.n_jobs <= 0) return;
	const int n_blocks = (n_slots + 3) / 4;
	hipStream_t s = (hipStream_t)stream;
#define EQ_GO(R_, N_) hipLaunchKernelGGL((ksw_extq_kernel<R_, N_>), dim3(n_blocks), dim3(256), 0, s, L)
	if (n_sets <= 2) { if (right) EQ_GO(true, 2); else EQ_GO(false, 2); }
	else if (n_sets <= 4) { if (right) EQ_GO(true, 4); else EQ_GO(false, 4); }
	else { if (right) EQ_GO(true, 8); else EQ_GO(false, 8); }
#undef EQ_GO
	HIP_CHECK(hipGetLastError());
}

int ksw_extq_max_t() { return EQ_TCAP; }

} // namespace mm2amd

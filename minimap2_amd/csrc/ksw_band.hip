// Banded gap-fill DP for gfx950 (round 6): the register-resident gap-fill cell (gf_cell_k) with the BAND across the lanes instead of the target.
//
// The streaming and the strip kernel (ksw_stream.hip, ksw_gapfill.hip) give every target column a lane and compute the whole qlen x tlen rectangle the
// reference computes (align.c:810-844 fills the gap between two anchors with a band that cannot bind).  Three quarters of those cells cannot lie on an
// optimal alignment (profiles/r05_band_bound_cpu.txt), and the kernels are bound by VALU issue: only fewer cells make them faster.  Here a wavefront
// computes W = 128 * NB diagonals d = i - j (i target index, j query index) around the diagonals 0 and D = tlen - qlen of the matrix' two corners:
//
//   * anti-diagonal r holds the cells with i + j = r, whose diagonals have r's parity: lane l' (= 64 * register set + lane) owns the band's diagonal pair
//     k = 2 l' (even rows) and k = 2 l' + 1 (odd rows), d = k - 2 c.  On an even row its cell is (r/2 - c + l', r/2 + c - l') and takes (x, v, x2) of the left
//     neighbour (i - 1, j) -- diagonal k - 1, the odd-row cell of lane l' - 1: one DPP wave shift right -- and (u, y, y2) of the upper neighbour (i, j - 1) --
//     diagonal k + 1, its OWN odd-row cell.  On an odd row it is the other way round: (x, v, x2) are its own, (u, y, y2) come from lane l' + 1 by a DPP wave
//     shift left.  Three DPP moves per row as in the column layout, all six states stay in VGPRs, 64 lanes x 2 jobs (packed 16-bit halves) = 128 cells per
//     register set and row at full lane occupancy inside the matrix;
//   * a lane's target index advances on odd rows, its query index on even rows: one 16-bit LDS load per row (both halves' bases), from arrays laid out so
//     that the slot is the same function of (row, lane) for both jobs of the pair although each has its own band offset c;
//   * the matrix' first row and column (u / v on the border, ksw2_extd2_sse.c:148-163) meet the band only during its first ~64 NB rows: one lane per half
//     whose cell has i = 0, one whose cell has j = 0, patched by bit-field inserts in a uniform branch; the band's own edges take the reference's "neighbour not
//     computed" constants (:111-116, :152-154) through the DPP moves' carry-in;
//   * the direction bytes of a row pair form one dword per lane ([even A, even B, odd A, odd B]) in HBM: 64 NB bytes per row and job instead of tlen.
//
// After its last row a pair is traced back and scanned exactly as in the other two kernels (gf_traceback, gf_zdrop_scan); the scan's score under the DP's own
// costs IS the corner score, and band_outside_bound (ksw_band.hpp, where the argument is) tells whether that score proves the band sufficient.  If so the
// result is the rectangle's and is written; if not, nothing is written and the job's index goes onto a list: `widen` (a launch of this kernel with a wider
// band, when the score found says that one would do) or `retry` / `big` (the full rectangle: the streaming kernel's up to 512 x 512, the strip kernel's beyond).  The lists are consumed by launches whose job count is
// read on the device -- no host round trip.  tests/test_gpu_ksw.py: every accepted result against the reference's unbanded ksw_extd2_sse, with the acceptance
// forced to fail, with repeats that put equally good alignments far from the diagonal, and through the list-driven launches.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"
#include "ksw_gapfill_dev.hpp"
#include "ksw_band.hpp"
#include <type_traits>

namespace mm2amd {

// bases kept per pair and sequence: (qlen + tlen) / 2 + 64 NB + 2 slots are in use (query, target <= HALF; 4 B per slot, two arrays: 4.6 / 5.1 KB of LDS per wave with
// one / two register sets and windows up to 512 x 512, 10.3 KB with four sets and windows up to 1024 x 1024)
constexpr int bd_slots(int n_sets, int half) { return half + 64 * n_sets + 8; }

__device__ __forceinline__ uint32_t bd_shl1(uint32_t carry_in, uint32_t v) // lane i <- v[i + 1], lane 63 <- carry_in
{
	return (uint32_t)__builtin_amdgcn_update_dpp((int)carry_in, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t bd_rol1(uint32_t v) // lane i <- v[(i + 1) & 63]
{
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ int bd_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t bd_uni64(uint64_t v)
{
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}

template <int NB, int WAVES, int HALF>
__global__ void __launch_bounds__(256, WAVES) ksw_band_kernel(KswLaunch L)
{
	__shared__ uint32_t s_q[4][bd_slots(NB, HALF)];     // slot s: query base j = s - QOFF + c of each half (A | B << 16: as the cell wants it -- one LDS load per row, nothing to unpack); 4 where there is none
	__shared__ uint32_t s_t[4][bd_slots(NB, HALF)];     // slot s: target base i = s - c
	__shared__ int8_t s_mat[32];
	constexpr int W = 128 * NB, NL = 64 * NB, QOFF = NL;
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	const int slot = blockIdx.x * 4 + wave_in_block;
	if (threadIdx.x < 25) s_mat[threadIdx.x] = L.sc.mat[threadIdx.x];
	__syncthreads();
	const int m = L.sc.m;
	int q = L.sc.q, e = L.sc.e, q2 = L.sc.q2, e2 = L.sc.e2;
	const int qe_in = q + e; // before the swap (ksw2_extd2_sse.c:68 vs :78)
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t; t = e, e = e2, e2 = t; }
	const int qe = q + e, nqe = -qe;
	const int sc_mch = L.sc.mat[0], sc_mis = L.sc.mat[1];
	const int sc_N = L.sc.mat[m * m - 1] == 0 ? -e2 : L.sc.mat[m * m - 1];
	int sc_max = 0;
	for (int k = 0; k < m * m; ++k) sc_max = L.sc.mat[k] > sc_max ? L.sc.mat[k] : sc_max;
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const GfK K = gf_k_consts(sc_mch, sc_mis, sc_N, q, e, q2, e2);
	const uint32_t P_MCH = pk2v(8 * sc_mch + GF_K_TS);
	const uint32_t S_OPEN = pk2(8 * nqe); // u / v next to a cell that was not computed (:111-112)
	const uint32_t lane4 = (uint32_t)lane * 4u;
	uint8_t *const dir = L.dir_pool + (size_t)(2 * slot) * L.slot_bytes; // ((rows + 1) / 2) x NL dwords
	const uint8_t *const qbytes = (const uint8_t *)&s_q[wave_in_block][0], *const tbytes = (const uint8_t *)&s_t[wave_in_block][0];
	auto border = [&](int i) { return 8 * (i == 0 ? nqe : i < long_thres ? -e : i == long_thres ? long_diff : -e2); }; // v[-1] / u[i] on the matrix border (:148-163), times 8
	const int n_avail = L.n_list ? bd_uni(*L.n_list) : L.n_jobs;
	unsigned long long acc_got = 0, acc_best = 0; // (lanes 0 and 32: their halves' windows)

	for (;;) {
		int pid = 0;
		if (lane == 0) pid = atomicAdd(L.counter, 1);
		pid = __builtin_amdgcn_readfirstlane(pid);
		if (2 * pid >= n_avail) break;
		const bool hasB = 2 * pid + 1 < n_avail;
		const int jidA = L.list ? bd_uni((int)L.list[2 * pid]) : 2 * pid, jidB = !hasB ? jidA : L.list ? bd_uni((int)L.list[2 * pid + 1]) : 2 * pid + 1;
		KswJob JA = L.jobs[jidA], JB = L.jobs[jidB];
		JA.q_off = bd_uni64(JA.q_off), JA.t_off = bd_uni64(JA.t_off), JA.qlen = bd_uni(JA.qlen), JA.tlen = bd_uni(JA.tlen), JA.flag = bd_uni(JA.flag);
		JB.q_off = bd_uni64(JB.q_off), JB.t_off = bd_uni64(JB.t_off), JB.qlen = bd_uni(JB.qlen), JB.tlen = bd_uni(JB.tlen), JB.flag = bd_uni(JB.flag);
		const int qlenA = JA.qlen, tlenA = JA.tlen, qlenB = hasB ? JB.qlen : 0, tlenB = hasB ? JB.tlen : 0;
		const int cA = band_c(qlenA, tlenA, W), cB = band_c(qlenB, tlenB, W);
		const int n_rowsA = qlenA + tlenA - 1, n_rowsB = hasB ? qlenB + tlenB - 1 : 0;
		// a window whose corners the band does not hold, or that is longer than the LDS arrays, is not computed at all: it fails the acceptance below
		const bool fitA = band_holds_corners(qlenA, tlenA, W) && (n_rowsA + 1) / 2 + NL + 2 <= bd_slots(NB, HALF);
		const bool fitB = hasB && band_holds_corners(qlenB, tlenB, W) && (n_rowsB + 1) / 2 + NL + 2 <= bd_slots(NB, HALF);
		const int rows_run = (fitA ? n_rowsA : 0) > (fitB ? n_rowsB : 0) ? (fitA ? n_rowsA : 0) : (fitB ? n_rowsB : 0);

		// ---- the pair's bases into LDS ----
		{
			const int smax = (rows_run + 1) / 2 + NL + 2;
			for (int s = lane; s < smax; s += 64) {
				uint32_t bqA = 4, bqB = 4, btA = 4, btB = 4;
				const int jA = s - QOFF + cA, jB = s - QOFF + cB, iA = s - cA, iB = s - cB;
				if (fitA && (unsigned)jA < (unsigned)qlenA) bqA = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)jA : JA.q_off + (uint64_t)jA];
				if (fitB && (unsigned)jB < (unsigned)qlenB) bqB = L.qpool[(JB.flag & KSWJ_Q_REVERSED) ? JB.q_off - (uint64_t)jB : JB.q_off + (uint64_t)jB];
				if (fitA && (unsigned)iA < (unsigned)tlenA) {
					const uint64_t pos = (JA.flag & KSWJ_T_REVERSED) ? JA.t_off - (uint64_t)iA : JA.t_off + (uint64_t)iA;
					btA = (JA.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
				}
				if (fitB && (unsigned)iB < (unsigned)tlenB) {
					const uint64_t pos = (JB.flag & KSWJ_T_REVERSED) ? JB.t_off - (uint64_t)iB : JB.t_off + (uint64_t)iB;
					btB = (JB.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
				}
				s_q[wave_in_block][s] = bqA | bqB << 16, s_t[wave_in_block][s] = btA | btB << 16;
			}
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}

		// ---- the rows ----
		// lanes whose cell lies on the matrix' first column (i = 0: l' = c - ceil(r / 2)) or first row (j = 0: l' = c + floor(r / 2)) exist up to this row
		const int cmax = cA > cB ? cA : cB, cmin = cA < cB ? cA : cB;
		const int r_border_end = 2 * cmax > 2 * (NL - 1 - cmin) + 1 ? 2 * cmax : 2 * (NL - 1 - cmin) + 1;
		uint32_t T[NB], Q[NB], U[NB], V[NB], X[NB], Y[NB], X2[NB], Y2[NB], DE[NB];
#pragma unroll
		for (int c = 0; c < NB; ++c) T[c] = Q[c] = 0x00040004u, U[c] = V[c] = X[c] = Y[c] = X2[c] = Y2[c] = DE[c] = 0u;
		// What the band's outermost lanes see beyond the band (diagonal -1 on even rows, diagonal W on odd rows: not computed, :111-116) arrives through the DPP moves'
		// untouched lane: wave_shr:1 never writes lane 0 of its destination, wave_shl:1 never lane 63 -- so the six destinations are set to the constants ONCE and
		// passed on from row to row as the moves' `old` operand (no per-row copy of a constant into a VGPR); the cell reads them, it never writes them (gf_cell_k2)
		uint32_t eV = S_OPEN, eX = K.nqe_x, eX2 = K.nqe2_x, oU = S_OPEN, oY = K.nqe_y, oY2 = K.nqe2_y;
		// one pair of rows; BORDER: some lane's cell may lie on the matrix' first row or column (the first r_border_end rows only -- a loop of their own, so that
		// the other rows carry neither the masks nor the branch)
		auto row_pair = [&](const int r0, auto border_tag) {
			constexpr bool BORDER = decltype(border_tag)::value;
			uint32_t need = (1u << NB) - 1u;
			if (NB > 1) { // register sets with a cell inside either matrix on either row of the pair (uniform)
				need = 0;
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					const int ch = h ? cB : cA, qh = h ? qlenB : qlenA, th = h ? tlenB : tlenA;
					if (!(h ? fitB : fitA) || r0 >= (h ? n_rowsB : n_rowsA)) continue;
					// row r: l' >= c - ceil(r/2), l' >= floor(r/2) + c - q + 1, l' <= t - 1 + c - ceil(r/2), l' <= floor(r/2) + c; the union over r0 and r0 + 1
					const int hr = r0 >> 1;
					int lo = ch - hr - 1 > hr + ch - qh + 1 ? ch - hr - 1 : hr + ch - qh + 1, hi = th - 1 + ch - hr < hr + ch ? th - 1 + ch - hr : hr + ch;
					lo = lo < 0 ? 0 : lo, hi = hi > NL - 1 ? NL - 1 : hi;
					if (lo <= hi) need |= ((2u << (hi >> 6)) - 1u) & ~((1u << (lo >> 6)) - 1u);
				}
			}
			uint32_t *const prow = (uint32_t *)(dir + (size_t)(r0 >> 1) * (size_t)(NL * 4));
			const int hr = r0 >> 1;
			{ // ---- the even row r0: (x, v, x2) from lane l' - 1, (u, y, y2) the lane's own; the query index moved on ----
				const uint32_t S_BND = pk2(border(r0));
#pragma unroll
				for (int c = NB - 1; c >= 0; --c) { // from the highest set down: set c still sees row r - 1 in set c - 1
					if (!(need >> c & 1u)) continue;
					const int lp = c * 64 + lane;
					Q[c] = *(const uint32_t *)(qbytes + ((uint32_t)(4 * (hr + QOFF - 64 * c)) - lane4));
					if (r0 == 0) T[c] = *(const uint32_t *)(tbytes + ((uint32_t)(256 * c) + lane4));
					uint32_t vp, xp, x2p;
					if (c > 0) vp = dpp_shr1u(gf_ror1(V[c - 1]), V[c]), xp = dpp_shr1u(gf_ror1(X[c - 1]), X[c]), x2p = dpp_shr1u(gf_ror1(X2[c - 1]), X2[c]);
					else vp = eV = dpp_shr1u(eV, V[0]), xp = eX = dpp_shr1u(eX, X[0]), x2p = eX2 = dpp_shr1u(eX2, X2[0]);
					uint32_t uu = U[c], yy = Y[c], yy2 = Y2[c];
					if (BORDER) {
						const uint32_t emI = (lp == cA - hr ? 0xffffu : 0u) | (lp == cB - hr ? 0xffff0000u : 0u); // i = 0: the left neighbour is the border (:148-155)
						const uint32_t emJ = (lp == cA + hr ? 0xffffu : 0u) | (lp == cB + hr ? 0xffff0000u : 0u); // j = 0: the upper one is (:156-163)
						vp = bfi(emI, S_BND, vp), xp = bfi(emI, K.nqe_x, xp), x2p = bfi(emI, K.nqe2_x, x2p);
						uu = bfi(emJ, S_BND, uu), yy = bfi(emJ, K.nqe_y, yy), yy2 = bfi(emJ, K.nqe2_y, yy2);
					}
					uint32_t d;
					gf_cell_k2(T[c] ^ Q[c], T[c] | Q[c], xp, vp, x2p, uu, yy, yy2, U[c], V[c], X[c], Y[c], X2[c], Y2[c], d, P_MCH, K);
					DE[c] = d;
				}
			}
			{ // ---- the odd row r0 + 1: (x, v, x2) the lane's own, (u, y, y2) from lane l' + 1; the target index moved on ----
				const uint32_t S_BND = pk2(border(r0 + 1));
#pragma unroll
				for (int c = 0; c < NB; ++c) { // from the lowest set up: set c still sees row r - 1 in set c + 1
					if (!(need >> c & 1u)) continue;
					const int lp = c * 64 + lane;
					T[c] = *(const uint32_t *)(tbytes + ((uint32_t)(4 * (hr + 1 + 64 * c)) + lane4));
					uint32_t uu, yy, yy2;
					if (c + 1 < NB) uu = bd_shl1(bd_rol1(U[c + 1]), U[c]), yy = bd_shl1(bd_rol1(Y[c + 1]), Y[c]), yy2 = bd_shl1(bd_rol1(Y2[c + 1]), Y2[c]);
					else uu = oU = bd_shl1(oU, U[c]), yy = oY = bd_shl1(oY, Y[c]), yy2 = oY2 = bd_shl1(oY2, Y2[c]);
					uint32_t vp = V[c], xp = X[c], x2p = X2[c];
					if (BORDER) {
						const uint32_t emI = (lp == cA - hr - 1 ? 0xffffu : 0u) | (lp == cB - hr - 1 ? 0xffff0000u : 0u);
						const uint32_t emJ = (lp == cA + hr ? 0xffffu : 0u) | (lp == cB + hr ? 0xffff0000u : 0u);
						vp = bfi(emI, S_BND, vp), xp = bfi(emI, K.nqe_x, xp), x2p = bfi(emI, K.nqe2_x, x2p);
						uu = bfi(emJ, S_BND, uu), yy = bfi(emJ, K.nqe_y, yy), yy2 = bfi(emJ, K.nqe2_y, yy2);
					}
					uint32_t d;
					gf_cell_k2(T[c] ^ Q[c], T[c] | Q[c], xp, vp, x2p, uu, yy, yy2, U[c], V[c], X[c], Y[c], X2[c], Y2[c], d, P_MCH, K);
					*(uint32_t *)((uint8_t *)prow + c * 256 + lane4) = __builtin_amdgcn_perm(d, DE[c], 0x06040200u); // [even A, even B, odd A, odd B]
				}
			}
		};
		int r0 = 0;
		for (; r0 < rows_run && r0 <= r_border_end; r0 += 2) row_pair(r0, std::true_type());
		for (; r0 < rows_run; r0 += 2) row_pair(r0, std::false_type());

		// ---- traceback from the corner (ksw2_extd2_sse.c:389-391), mm_test_zdrop's scan (align.c:61-84) and the score, one half-wave per job ----
		__threadfence_block();
		const bool isB = lane >= 32;
		const int h = isB ? 1 : 0;
		const bool have = isB ? fitB : fitA;
		const int my_q = isB ? qlenB : qlenA, my_t = isB ? tlenB : tlenA, my_c = isB ? cB : cA, my_id = isB ? jidB : jidA;
		FastCig g = { L.cigar_tmp + (size_t)(2 * slot + h) * L.cigar_tmp_cap, 0, 0u };
		gf_traceback(have, my_t - 1, my_q - 1, [&](int ii, int jj) {
			const int k = ii - jj + 2 * my_c, rr = ii + jj;
			if ((unsigned)k >= (unsigned)W) return 0; // outside the band: never on an accepted path; "diagonal" keeps the walk moving
			return gf_k_decode(dir[((uint32_t)(rr >> 1) * (uint32_t)NL + (uint32_t)(k >> 1)) * 4u + (uint32_t)((rr & 1) << 1) + (uint32_t)h], K.bias);
		}, g);
		if ((lane & 31) == 0 && have && g.n > 0) g.c[g.n - 1] = g.last;
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		int32_t dp_sum = 0;
		GfZdrop z;
		{
			const uint32_t third = L.cigar_tmp_cap / 3u; // a job's scratch: its operations (last first), then two prefix arrays
			z = gf_zdrop_scan(have, g.n, g.c, g.c + third, g.c + 2u * third, [&](int i) { return (int)tbytes[(i + my_c) * 4 + 2 * h]; },
			                  [&](int j) { return (int)qbytes[(j - my_c + QOFF) * 4 + 2 * h]; }, s_mat, L.sc.q, L.sc.e, L.sc.q2, L.sc.e2, sc_N);
			dp_sum = z.dp_sum;
		}
		// ---- is the band proven sufficient?  (ksw_band.hpp)  MM2AMD_BAND_REJECT (tests): every first attempt fails, so that the lists and their launches run
		const int bound = band_outside_bound(my_q, my_t, W, sc_max, q, e, q2, e2);
		const bool accept = have && dp_sum > bound && !L.band_reject;
		if (have) { // what the launch classes' expectation is fitted to (ksw_host.cpp: band_rho)
			const int D = my_t - my_q, got = dp_sum + band_gap_cost(D < 0 ? -D : D, q, e, q2, e2);
			acc_got += (unsigned long long)(got > 0 ? got : 0), acc_best += (unsigned long long)(sc_max * (my_q < my_t ? my_q : my_t));
		}
		__threadfence_block();
		uint32_t cig_off = 0;
		if ((lane & 31) == 0 && (isB ? hasB : true)) {
			if (accept) { if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n); }
			else {
				// a band twice as wide is worth a launch when the score found here would be accepted there (it can only be higher there)
				const bool widen = L.widen_list && band_holds_corners(my_q, my_t, L.widen_W) && (my_q + my_t) / 2 + L.widen_W / 2 + 2 <= L.widen_slots &&
				                   (have ? dp_sum : INT32_MIN) > band_outside_bound(my_q, my_t, L.widen_W, sc_max, q, e, q2, e2) && L.band_reject < 2;
				if (widen) L.widen_list[atomicAdd(L.widen_count, 1)] = L.list_base + (uint32_t)my_id;
				else if (my_q > L.retry_max || my_t > L.retry_max) L.big_list[atomicAdd(L.big_count, 1)] = L.list_base + (uint32_t)my_id;
				else L.retry_list[atomicAdd(L.retry_count, 1)] = L.list_base + (uint32_t)my_id;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int which = 0; which < 2; ++which) { // the CIGARs into the pool in forward order, all lanes copying
			const int src = which * 32;
			if (!__builtin_amdgcn_readlane((int)accept, src)) continue;
			const int n_cig = __builtin_amdgcn_readlane(g.n, src);
			const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)cig_off, src);
			const uint32_t *tmpc = L.cigar_tmp + (size_t)(2 * slot + which) * L.cigar_tmp_cap;
			if (n_cig > 0) {
				if ((unsigned long long)off + (unsigned)n_cig > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
				else for (int k = lane; k < n_cig; k += 64) L.cigar_pool[off + k] = tmpc[n_cig - 1 - k];
			}
		}
		if ((lane & 31) == 0 && accept) {
			KswRes R;
			R.max = 0, R.zdropped = 0, R.max_q = R.max_t = -1, R.mqe = R.mte = KSW_NEG_INF, R.mqe_t = R.mte_q = -1;
			R.score = dp_sum + qe - qe_in, R.n_cigar = g.n, R.reach_end = 0, R.cigar_off = cig_off; // (the reference's score offset when the second cost pair is the cheaper one, :68 vs :78)
			R.zd_max = z.zd_max, R.zd_t0 = z.t0, R.zd_t1 = z.t1, R.zd_q0 = z.q0, R.zd_q1 = z.q1;
			L.res[my_id] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
	if (L.band_acc && (lane & 31) == 0 && acc_best) atomicAdd(&L.band_acc[0], acc_got), atomicAdd(&L.band_acc[1], acc_best);
}

void ksw_band_launch(const KswLaunch &L, int n_slots, int n_sets, void *stream)
{
	if (L.n_jobs <= 0 && !L.n_list) return;
	const int n_blocks = (n_slots + 3) / 4;
	hipStream_t s = (hipStream_t)stream;
	if (n_sets == 1) hipLaunchKernelGGL((ksw_band_kernel<1, 8, 512>), dim3(n_blocks), dim3(256), 0, s, L);
	else if (n_sets == 2) hipLaunchKernelGGL((ksw_band_kernel<2, 6, 512>), dim3(n_blocks), dim3(256), 0, s, L);
	else if (n_sets == 4) hipLaunchKernelGGL((ksw_band_kernel<4, 3, 1024>), dim3(n_blocks), dim3(256), 0, s, L);
	else throw std::runtime_error("[mm2amd] ksw_band_launch: unsupported register-set count");
	HIP_CHECK(hipGetLastError());
}

int ksw_band_waves(int n_sets) { return n_sets == 1 ? 8 : n_sets == 2 ? 6 : 3; } // blocks of four waves per CU the instantiation is compiled for (four sets: 41 KB of LDS per block)
int ksw_band_slots(int n_sets) { return bd_slots(n_sets, n_sets == 4 ? 1024 : 512); } // base slots per sequence a wave holds: a window needs (qlen + tlen) / 2 + 64 n_sets + 2
size_t ksw_band_slot_bytes(int n_sets, int max_rows) { return (size_t)((max_rows + 3) / 2) * (size_t)(n_sets * 64) * 4 / 2; } // per job slot; a wave's matrix is two of them

} // namespace mm2amd

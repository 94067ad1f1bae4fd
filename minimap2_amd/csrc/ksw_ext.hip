// Register-resident extension DP for gfx950: ksw_extd2_sse (ksw2_extd2_sse.c:34-401) + ksw_backtrack for the two extension
// calls of mm_align1 -- to the right of the last anchor (align.c:883: KSW_EZ_EXTZ_ONLY) and to the left of the first one on the
// reversed sequences (align.c:791: KSW_EZ_EXTZ_ONLY | KSW_EZ_RIGHT | KSW_EZ_REV_CIGAR) -- when the band cannot bind
// (qlen <= w and tlen <= w + 1: then no row interval is clipped by the band, see ksw_host.cpp), which is the normal case: a
// read's ends beyond its outermost anchors are a few dozen to a few hundred bases, the band is 751.
//
// The layout is ksw_fast.hip's (lane = target column, difference states in VGPRs, two jobs per wave in packed 16-bit halves);
// what an extension needs on top of a gap fill is the exact score of every cell's row maximum: H(r, t) per column in a 32-bit
// register per job, the reference's row-maximum search order (ksw2_extd2_sse.c:325-358: last cell first, then four strided
// scans, then the tail) as a rank that breaks ties in a two-step wave reduction, the best-extension bookkeeping (max, mqe, mte)
// and the Z-drop test as scalars, and the traceback from the cell they choose.  RIGHT selects the "ties go to the gap state"
// flavour of the recurrence (:282-320).  The lane-exact kernel remains the reference implementation these are tested against.
#include <hip/hip_runtime.h>
#include <climits>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"

namespace mm2amd {

constexpr int EXT_QCAP = 1024; // query bytes kept in LDS per job

namespace {
// full-wave maximum by DPP row shifts and row broadcasts; every lane returns it
__device__ __forceinline__ int wave_max_i32(int v)
{
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112 /* row_shr:2 */, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114 /* row_shr:4 */, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118 /* row_shr:8 */, 0xf, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false));
	v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false));
	return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v) { return -wave_max_i32(-v); } // callers keep v > INT_MIN

struct ExtEz { int max, max_t, max_q, mqe, mqe_t, mte, mte_q, score, zdropped; };

// ksw_apply_zdrop with is_rot = 1 (ksw2.h:171-187)
__device__ __forceinline__ bool ext_zdrop(ExtEz &ez, int H, int r, int t, int zdrop, int e)
{
	if (H > ez.max) ez.max = H, ez.max_t = t, ez.max_q = r - t;
	else if (t >= ez.max_t && r - t >= ez.max_q) {
		const int tl = t - ez.max_t, ql = (r - t) - ez.max_q, l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && ez.max - H > zdrop + l * e) { ez.zdropped = 1; return true; }
	}
	return false;
}
}

template <int NC, bool RIGHT>
__global__ void __launch_bounds__(256, (NC <= 4 ? 3 : 2)) ksw_ext_kernel(KswLaunch L)
{
	__shared__ uint8_t s_q[4][2][EXT_QCAP];
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	const int slot = blockIdx.x * 4 + wave_in_block;
	const int m = L.sc.m;
	int q = L.sc.q, e = L.sc.e, q2 = L.sc.q2, e2 = L.sc.e2;
	const int qe_in = q + e; // before the swap (ksw2_extd2_sse.c:68 vs :78)
	if (q2 + e2 < q + e) { int t = q; q = q2, q2 = t; t = e, e = e2, e2 = t; }
	const int qe = q + e, qe2 = q2 + e2, nqe = -qe, nqe2 = -qe2;
	const int sc_mch = L.sc.mat[0], sc_mis = L.sc.mat[1];
	const int sc_N = L.sc.mat[m * m - 1] == 0 ? -e2 : L.sc.mat[m * m - 1];
	int long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	const uint32_t P_ONE = pk2v(1), P_ZERO = pk2v(0), P_MCH = pk2v(sc_mch), P_MISD = pk2v(sc_mis - sc_mch), P_SCN = pk2v(sc_N);
	const uint32_t P_Q = pk2v(q), P_Q2 = pk2v(q2), P_QE = pk2v(qe), P_QE2 = pk2v(qe2), P_NQE = pk2(nqe), P_NQE2 = pk2(nqe2);
	const uint32_t P_8 = pk2v(8), P_16 = pk2v(16), P_32 = pk2v(32), P_64 = pk2v(64);
	const uint32_t P_TWO = pk2v(2), P_THREE = pk2v(3), P_FOUR = pk2v(4);

	for (;;) {
		int pid = 0;
		if (lane == 0) pid = atomicAdd(L.counter, 1);
		pid = __builtin_amdgcn_readfirstlane(pid);
		if (2 * pid >= L.n_jobs) break;
		const int jidA = 2 * pid, jidB = 2 * pid + 1;
		const bool hasB = jidB < L.n_jobs;
		const KswJob JA = L.jobs[jidA], JB = L.jobs[hasB ? jidB : jidA];
		const int qlenA = JA.qlen, tlenA = JA.tlen, qlenB = hasB ? JB.qlen : 0, tlenB = hasB ? JB.tlen : 0;
		const int ncolA = (tlenA + 63) & ~63, ncolB = (tlenB + 63) & ~63;
		uint8_t *dirA = L.dir_pool + (size_t)(2 * slot) * L.slot_bytes, *dirB = dirA + L.slot_bytes;
		uint8_t *qbA = s_q[wave_in_block][0]; // job B's bytes follow at + EXT_QCAP
		for (int i = lane; i < qlenA; i += 64) qbA[i] = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)i : JA.q_off + (uint64_t)i];
		for (int i = lane; i < qlenB; i += 64) qbA[i + EXT_QCAP] = L.qpool[(JB.flag & KSWJ_Q_REVERSED) ? JB.q_off - (uint64_t)i : JB.q_off + (uint64_t)i];
		uint32_t T[NC], U[NC], V[NC], X[NC], Y[NC], X2[NC], Y2[NC];
		int HA[NC], HB[NC]; // H(r, t) of the column's latest cell, per job (ksw2_extd2_sse.c:325-358)
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int t = c * 64 + lane;
			uint32_t bA = 4, bB = 4;
			if (t < tlenA) {
				const uint64_t pos = (JA.flag & KSWJ_T_REVERSED) ? JA.t_off - (uint64_t)t : JA.t_off + (uint64_t)t;
				bA = (JA.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
			}
			if (t < tlenB) {
				const uint64_t pos = (JB.flag & KSWJ_T_REVERSED) ? JB.t_off - (uint64_t)t : JB.t_off + (uint64_t)t;
				bB = (JB.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
			}
			T[c] = bA | bB << 16;
			U[c] = V[c] = X[c] = Y[c] = P_NQE, X2[c] = Y2[c] = P_NQE2;
			HA[c] = HB[c] = KSW_NEG_INF;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();

		ExtEz ezA = { 0, -1, -1, KSW_NEG_INF, -1, KSW_NEG_INF, -1, KSW_NEG_INF, 0 }, ezB = ezA;
		bool aliveA = qlenA > 0 && tlenA > 0, aliveB = hasB && qlenB > 0 && tlenB > 0; // still extending (not Z-dropped, rows left)
		const int n_rowsA = qlenA + tlenA - 1, n_rowsB = hasB ? qlenB + tlenB - 1 : 0, n_rows = n_rowsA > n_rowsB ? n_rowsA : n_rowsB;
		for (int r = 0; r < n_rows && (aliveA || aliveB); ++r) {
			int st0A = r - qlenA + 1 > 0 ? r - qlenA + 1 : 0, en0A = r < tlenA - 1 ? r : tlenA - 1;
			int st0B = r - qlenB + 1 > 0 ? r - qlenB + 1 : 0, en0B = r < tlenB - 1 ? r : tlenB - 1;
			if (!aliveA) st0A = 1, en0A = 0;
			if (!aliveB) st0B = 1, en0B = 0;
			const uint32_t wA = (uint32_t)(en0A - st0A + 1), wB = (uint32_t)(en0B - st0B + 1);
			const int lo = st0A <= en0A ? (st0B <= en0B && st0B < st0A ? st0B : st0A) : st0B, hi = en0A > en0B ? en0A : en0B;
			const int bnd = r == 0 ? nqe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
			const uint32_t P_BND = pk2(bnd);
			const bool topA = aliveA && r < tlenA, topB = aliveB && r < tlenB;
			const int edge_set = r >> 6, edge_lane = r & 63;
			const uint32_t edge_halves = (topA ? 0xffffu : 0u) | (topB ? 0xffff0000u : 0u);
			uint8_t *prA = dirA + (size_t)r * ncolA, *prB = dirB + (size_t)r * ncolB;
			// the row maximum: per lane the best (H, search rank) over its columns, reduced over the wave after the sweep
			const int en1A = st0A + (en0A - st0A) / 4 * 4, nqA = (en1A - st0A) >> 2, en1B = st0B + (en0B - st0B) / 4 * 4, nqB = (en1B - st0B) >> 2;
			int bestHA = INT_MIN, bestRA = INT_MAX, bestHB = INT_MIN, bestRB = INT_MAX;
			int HenA = KSW_NEG_INF, HstA = KSW_NEG_INF, HenB = KSW_NEG_INF, HstB = KSW_NEG_INF;
#pragma unroll
			for (int c = NC - 1; c >= 0; --c) {
				if (c * 64 > hi || c * 64 + 63 < lo) continue;
				const int t = c * 64 + lane;
				const bool actA = (uint32_t)(t - st0A) < wA, actB = (uint32_t)(t - st0B) < wB;
				uint32_t cV = P_BND, cX = P_NQE, cX2 = P_NQE2;
				int cHA = 0, cHB = 0;
				if (c > 0) {
					cV = __builtin_amdgcn_readlane(V[c - 1], 63), cX = __builtin_amdgcn_readlane(X[c - 1], 63), cX2 = __builtin_amdgcn_readlane(X2[c - 1], 63);
					cHA = __builtin_amdgcn_readlane(HA[c - 1], 63), cHB = __builtin_amdgcn_readlane(HB[c - 1], 63);
				}
				const uint32_t vp = dpp_shr1u(cV, V[c]), xp = dpp_shr1u(cX, X[c]), x2p = dpp_shr1u(cX2, X2[c]);
				const int hpA = dpp_shr1(cHA, HA[c]), hpB = dpp_shr1(cHB, HB[c]); // H(r-1, t-1)
				if (edge_halves && edge_set == c) {
					const uint32_t em = lane == edge_lane ? edge_halves : 0u;
					U[c] = bfi(em, P_BND, U[c]), Y[c] = bfi(em, P_NQE, Y[c]), Y2[c] = bfi(em, P_NQE2, Y2[c]);
				}
				int rt = r - t;
				rt = rt < 0 ? 0 : rt > EXT_QCAP - 1 ? EXT_QCAP - 1 : rt;
				const uint32_t qv = (uint32_t)qbA[rt] | (uint32_t)qbA[rt + EXT_QCAP] << 16, tv = T[c];
				uint32_t z = pk_mad(pk_minu(tv ^ qv, P_ONE), P_MISD, P_MCH);
				z = pk_mad(pk_shr2(tv | qv), pk_sub(P_SCN, z), z);
				const uint32_t ut = U[c];
				uint32_t a = pk_add(xp, vp), b = pk_add(Y[c], ut), a2 = pk_add(x2p, vp), b2 = pk_add(Y2[c], ut);
				const uint32_t z1 = pk_max(z, a), z2 = pk_max(z1, b), z3 = pk_max(z2, a2), z4 = pk_max(z3, b2);
				const uint32_t ne_a = pk_minu(pk_sub(z4, a), P_ONE), ne_b = pk_minu(pk_sub(z4, b), P_ONE);
				const uint32_t ne_a2 = pk_minu(pk_sub(z4, a2), P_ONE), ne_b2 = pk_minu(pk_sub(z4, b2), P_ONE);
				uint32_t d;
				if (!RIGHT) { // first of (s, a, b, a2, b2) equal to the maximum (:235-243)
					const uint32_t ne_s = pk_minu(pk_sub(z4, z), P_ONE);
					d = pk_mul(ne_s, pk_mad(ne_a, pk_mad(ne_b, pk_add(ne_a2, P_ONE), P_ONE), P_ONE));
				} else {      // last of them: ties go to the gap state (:282-290)
					uint32_t i = pk_sub(P_ONE, ne_a);
					i = pk_mad(ne_b, pk_sub(i, P_TWO), P_TWO);
					i = pk_mad(ne_a2, pk_sub(i, P_THREE), P_THREE);
					d = pk_mad(ne_b2, pk_sub(i, P_FOUR), P_FOUR);
				}
				z = pk_min(z4, P_MCH);
				U[c] = pk_sub(z, vp), V[c] = pk_sub(z, ut);
				uint32_t tmp = pk_sub(z, P_Q);
				a = pk_sub(a, tmp), b = pk_sub(b, tmp);
				tmp = pk_sub(z, P_Q2);
				a2 = pk_sub(a2, tmp), b2 = pk_sub(b2, tmp);
				const uint32_t ma = pk_max(a, P_ZERO), mb = pk_max(b, P_ZERO), ma2 = pk_max(a2, P_ZERO), mb2 = pk_max(b2, P_ZERO);
				if (!RIGHT) { // the gap can be extended when a > 0 (:261-272)
					d = pk_mad(pk_minu(ma, P_ONE), P_8, d);
					d = pk_mad(pk_minu(mb, P_ONE), P_16, d);
					d = pk_mad(pk_minu(ma2, P_ONE), P_32, d);
					d = pk_mad(pk_minu(mb2, P_ONE), P_64, d);
				} else {      // ... when a >= 0 (:308-320)
					d = pk_mad(pk_minu(pk_max(pk_add(a, P_ONE), P_ZERO), P_ONE), P_8, d);
					d = pk_mad(pk_minu(pk_max(pk_add(b, P_ONE), P_ZERO), P_ONE), P_16, d);
					d = pk_mad(pk_minu(pk_max(pk_add(a2, P_ONE), P_ZERO), P_ONE), P_32, d);
					d = pk_mad(pk_minu(pk_max(pk_add(b2, P_ONE), P_ZERO), P_ONE), P_64, d);
				}
				X[c] = pk_sub(ma, P_QE), Y[c] = pk_sub(mb, P_QE), X2[c] = pk_sub(ma2, P_QE2), Y2[c] = pk_sub(mb2, P_QE2);
				if (actA) prA[(uint32_t)t] = (uint8_t)d;
				if (actB) prB[(uint32_t)t] = (uint8_t)(d >> 16);
				// ---- exact scores of this row's cells (:325-358): H(r,t) = H(r-1,t) + v, except the row's last cell, which comes
				//      from its left neighbour: H(r-1,t-1) + u; cell (0,0) is v - (q+e) ----
				const int uA = (int)(U[c] << 16) >> 16, uB = (int)U[c] >> 16, vA = (int)(V[c] << 16) >> 16, vB = (int)V[c] >> 16;
				if (actA) {
					int h = HA[c] + vA;
					if (t == en0A) h = r == 0 ? vA - qe_in : en0A > 0 ? hpA + uA : h;
					HA[c] = h;
					const int k = t - st0A;
					const int rank = t == en0A ? 0 : t < en1A ? 1 + (k & 3) * (nqA + 1) + (k >> 2) : 1 + 4 * (nqA + 1) + (t - en1A);
					if (h > bestHA || (h == bestHA && rank < bestRA)) bestHA = h, bestRA = rank;
				}
				if (actB) {
					int h = HB[c] + vB;
					if (t == en0B) h = r == 0 ? vB - qe_in : en0B > 0 ? hpB + uB : h;
					HB[c] = h;
					const int k = t - st0B;
					const int rank = t == en0B ? 0 : t < en1B ? 1 + (k & 3) * (nqB + 1) + (k >> 2) : 1 + 4 * (nqB + 1) + (t - en1B);
					if (h > bestHB || (h == bestHB && rank < bestRB)) bestHB = h, bestRB = rank;
				}
				if (aliveA) {
					if ((en0A >> 6) == c) HenA = __builtin_amdgcn_readlane(HA[c], en0A & 63);
					if ((st0A >> 6) == c) HstA = __builtin_amdgcn_readlane(HA[c], st0A & 63);
				}
				if (aliveB) {
					if ((en0B >> 6) == c) HenB = __builtin_amdgcn_readlane(HB[c], en0B & 63);
					if ((st0B >> 6) == c) HstB = __builtin_amdgcn_readlane(HB[c], st0B & 63);
				}
			}
			// ---- per job: row maximum in the reference's search order, best-extension bookkeeping, Z-drop (:340-365) ----
			if (aliveA) {
				const int max_H = wave_max_i32(bestHA), rank = wave_min_i32(bestHA == max_H ? bestRA : INT_MAX);
				int max_t;
				if (rank == 0) max_t = en0A;
				else if (rank < 1 + 4 * (nqA + 1)) { const int k = rank - 1; max_t = st0A + (k % (nqA + 1)) * 4 + k / (nqA + 1); }
				else max_t = en1A + (rank - 1 - 4 * (nqA + 1));
				if (en0A == tlenA - 1 && HenA > ezA.mte) ezA.mte = HenA, ezA.mte_q = r - en0A;
				if (r - st0A == qlenA - 1 && HstA > ezA.mqe) ezA.mqe = HstA, ezA.mqe_t = st0A;
				if (ext_zdrop(ezA, max_H, r, max_t, JA.zdrop, e2)) aliveA = false;
				else if (r == n_rowsA - 1) { if (en0A == tlenA - 1) ezA.score = HenA; aliveA = false; }
			}
			if (aliveB) {
				const int max_H = wave_max_i32(bestHB), rank = wave_min_i32(bestHB == max_H ? bestRB : INT_MAX);
				int max_t;
				if (rank == 0) max_t = en0B;
				else if (rank < 1 + 4 * (nqB + 1)) { const int k = rank - 1; max_t = st0B + (k % (nqB + 1)) * 4 + k / (nqB + 1); }
				else max_t = en1B + (rank - 1 - 4 * (nqB + 1));
				if (en0B == tlenB - 1 && HenB > ezB.mte) ezB.mte = HenB, ezB.mte_q = r - en0B;
				if (r - st0B == qlenB - 1 && HstB > ezB.mqe) ezB.mqe = HstB, ezB.mqe_t = st0B;
				if (ext_zdrop(ezB, max_H, r, max_t, JB.zdrop, e2)) aliveB = false;
				else if (r == n_rowsB - 1) { if (en0B == tlenB - 1) ezB.score = HenB; aliveB = false; }
			}
		}
		// ---- where the alignments end (:385-399) and the tracebacks from there, the two jobs in the two half-waves ----
		__threadfence_block();
		const bool isB = lane >= 32;
		const ExtEz &ez = isB ? ezB : ezA;
		const KswJob &J = isB ? JB : JA;
		const int my_qlen = isB ? qlenB : qlenA, my_ncol = isB ? ncolB : ncolA;
		const uint8_t *my_dir = isB ? dirB : dirA;
		const bool present = !isB || hasB;
		const int reach_end = present && !ez.zdropped && ez.mqe + J.end_bonus > ez.max ? 1 : 0;
		FastCig g = { L.cigar_tmp + (size_t)(2 * slot + (isB ? 1 : 0)) * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;
		{
			const bool have = present && (reach_end || (ez.max_t >= 0 && ez.max_q >= 0));
			const int hl = lane & 31;
			int i = reach_end ? ez.mqe_t : ez.max_t, j = reach_end ? my_qlen - 1 : ez.max_q, state = 0;
			bool live = have && i >= 0 && j >= 0;
			while (__ballot(live) != 0ull) {
				const int di = (state == 2 || state == 4) ? 0 : 1, dj = (state == 1 || state == 3) ? 0 : 1;
				const int ii = i - hl * di, jj = j - hl * dj;
				const bool valid = live && ii >= 0 && jj >= 0;
				const int tmp = valid ? my_dir[(size_t)(ii + jj) * my_ncol + ii] : 0;
				const bool cont = valid && (state == 0 ? (tmp & 7) == 0 : (tmp >> (state + 2) & 1) != 0);
				const unsigned long long bal = __ballot(cont);
				const uint32_t mine = isB ? (uint32_t)(bal >> 32) : (uint32_t)bal;
				const int run = mine == 0xffffffffu ? 32 : __builtin_ctz(~mine);
				const int head = __shfl(tmp, lane & 32, 64);
				if (live) {
					if (run > 0) {
						fast_cig_push(g, state == 0 ? 0u : (state == 1 || state == 3) ? 2u : 1u, run);
						i -= run * di, j -= run * dj;
					} else {
						state = head & 7;
						if (state == 0) fast_cig_push(g, 0, 1), --i, --j;
						else if (state == 1 || state == 3) fast_cig_push(g, 2, 1), --i;
						else fast_cig_push(g, 1, 1), --j;
					}
					live = i >= 0 && j >= 0;
				}
			}
			if (have) {
				if (i >= 0) fast_cig_push(g, 2, i + 1);
				if (j >= 0) fast_cig_push(g, 1, j + 1);
			}
		}
		if (lane == 0 || (lane == 32 && hasB)) {
			if (g.n > 0) g.c[g.n - 1] = g.last;
			if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
		}
		__threadfence_block();
#pragma unroll
		for (int which = 0; which < 2; ++which) { // into the pool: traceback order for the left extension (REV_CIGAR), forward otherwise
			if (which == 1 && !hasB) break;
			const int src = which * 32;
			const int n_cig = __builtin_amdgcn_readlane(g.n, src);
			const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)cig_off, src);
			const uint32_t *tmpc = L.cigar_tmp + (size_t)(2 * slot + which) * L.cigar_tmp_cap;
			const bool keep_order = (which ? JB.flag : JA.flag) & KSW_REV_CIGAR;
			if (n_cig > 0) {
				if ((unsigned long long)off + (unsigned)n_cig > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
				else for (int k = lane; k < n_cig; k += 64) L.cigar_pool[off + k] = tmpc[keep_order ? k : n_cig - 1 - k];
			}
		}
		if (lane == 0 || (lane == 32 && hasB)) {
			KswRes R;
			R.max = ez.max, R.zdropped = ez.zdropped, R.max_q = ez.max_q, R.max_t = ez.max_t;
			R.mqe = ez.mqe, R.mqe_t = ez.mqe_t, R.mte = ez.mte, R.mte_q = ez.mte_q;
			R.score = ez.score, R.n_cigar = g.n, R.reach_end = reach_end, R.cigar_off = cig_off;
			R.zd_max = KSW_ZD_NONE, R.zd_t0 = R.zd_t1 = R.zd_q0 = R.zd_q1 = -1;
			L.res[isB ? jidB : jidA] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}

// n_sets register sets of 64 target columns; right: the KSW_EZ_RIGHT flavour (left extensions)
void ksw_ext_launch(const KswLaunch &L, int n_slots, int n_sets, bool right, void *stream)
{
	if (L.n_jobs <= 0) return;
	const int n_blocks = (n_slots + 3) / 4;
	hipStream_t s = (hipStream_t)stream;
	const bool r = right;
	switch (n_sets) {
	case 2: if (r) hipLaunchKernelGGL((ksw_ext_kernel<2, true>), dim3(n_blocks), dim3(256), 0, s, L); else hipLaunchKernelGGL((ksw_ext_kernel<2, false>), dim3(n_blocks), dim3(256), 0, s, L); break;
	case 4: if (r) hipLaunchKernelGGL((ksw_ext_kernel<4, true>), dim3(n_blocks), dim3(256), 0, s, L); else hipLaunchKernelGGL((ksw_ext_kernel<4, false>), dim3(n_blocks), dim3(256), 0, s, L); break;
	case 8: if (r) hipLaunchKernelGGL((ksw_ext_kernel<8, true>), dim3(n_blocks), dim3(256), 0, s, L); else hipLaunchKernelGGL((ksw_ext_kernel<8, false>), dim3(n_blocks), dim3(256), 0, s, L); break;
	default: throw std::runtime_error("[mm2amd] ksw_ext_launch: unsupported register-set count");
	}
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

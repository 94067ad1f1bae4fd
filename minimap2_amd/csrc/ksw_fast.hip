// Register-resident gap-fill DP for gfx950: the device counterpart of ksw_extd2_sse (ksw2_extd2_sse.c:34-401) +
// ksw_backtrack (ksw2.h:130-162) for the calls that make up >95 % of the DP cells of long-read mapping -- the global
// alignments between adjacent anchors (align.c:810-842: flag KSW_EZ_APPROX_MAX, band 1.5*bw_long+1, i.e. never binding).
//
// Why a second kernel.  ksw_extd2.hip reproduces the reference lane by lane (16-aligned row blocks, stale lanes, mod-256
// wrap) because a *binding* band lets valid cells read out-of-band garbage (hard part 1).  When the band
// cannot bind (w >= qlen + tlen) the valid cells of anti-diagonal r are exactly t in [max(0,r-qlen+1), min(tlen-1,r)], their
// neighbours are valid cells or the documented boundary values, and no 8-bit overflow occurs in valid cells (the scoring
// constraints mm_check_opt enforces, options.c:246-255, exist to guarantee that).  Only valid cells matter, so the layout is
// ours to choose:
//
//   * lane = target position: column t lives in lane t%64 of register set t/64 for the whole job, so the six difference
//     states (u,v,x,y,x2,y2) never leave VGPRs; the t-1 neighbour comes from a DPP wave shift (lane 0 takes the last lane of
//     the previous register set), the query base from a byte in LDS.  No LDS round trip, no barrier per row.
//   * one wavefront per job, persistent waves pulling jobs from a queue; the only HBM traffic is the 1 B/cell direction
//     byte (row-major, 64 B coalesced per active register set) and the sequential traceback over it.
//
//   * two jobs per wavefront in the halves of packed 16-bit registers (see below).
//
// tests/test_gpu_ksw.py checks this kernel against the lane-exact oracle on every preset's scoring; jobs that are not
// eligible (binding band, extension flags, exact-max mode, generic matrices, long sequences) take the exact kernel.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"

namespace mm2amd {

constexpr int FAST_QCAP = 1024; // query bytes kept in LDS per wave

// ---------------------------------------------------------------------------------------------------------
// Two jobs per wavefront, packed 16-bit arithmetic.
//
// The row loop is bound by VALU issue (about 70 instructions per 64 cells), and every quantity in it fits in 8 bits.  gfx950
// executes 2 x 16-bit packed integer ops (v_pk_add/sub/max/min/mad_u16) at the rate of one 32-bit op, so each lane carries the
// same column of TWO jobs: job A in the low halves of the state registers, job B in the high halves.  The jobs of a launch are
// ordered by cost, so the two members of a pair have nearly the same shape and advance in lockstep; DPP shifts, carries and
// border values act on both halves at once, and only the activity masks, the query bytes and the direction-byte stores are
// per job.  The direction index d ("first candidate that reaches the maximum", = the reference's strictly-greater update
// chain, ksw2_extd2_sse.c:235-243) and the continuation flags are computed arithmetically, because packed compares do not
// exist.  The packed instructions are issued through inline asm: written as C++ vector code the optimiser rewrites the
// min/mul idioms back into compares and de-vectorises them.
// ---------------------------------------------------------------------------------------------------------
// second launch bound = waves per SIMD to compile for: 4 (<= 128 VGPRs) up to 4 register sets, 3 (<= 168) for 5-6, 2 for 8
template <int NC>
__global__ void __launch_bounds__(256, (NC <= 4 ? 4 : NC <= 6 ? 3 : 2)) ksw_fast_kernel(KswLaunch L)
{
	__shared__ uint8_t s_q[4][2][FAST_QCAP];
	__shared__ uint8_t s_t[4][2][512];
	__shared__ int8_t s_mat[32];
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	const int slot = blockIdx.x * 4 + wave_in_block;
	if (threadIdx.x < 25) s_mat[threadIdx.x] = L.sc.mat[threadIdx.x];
	__syncthreads();
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
		uint8_t *qbA = s_q[wave_in_block][0], *qbB = s_q[wave_in_block][1], /* qbB == qbA + FAST_QCAP */ *tbA = s_t[wave_in_block][0], *tbB = s_t[wave_in_block][1];
		// ---- operands: query bytes to LDS, one packed target base pair per (register set, lane) ----
		for (int i = lane; i < qlenA; i += 64) qbA[i] = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)i : JA.q_off + (uint64_t)i];
		for (int i = lane; i < qlenB; i += 64) qbB[i] = L.qpool[(JB.flag & KSWJ_Q_REVERSED) ? JB.q_off - (uint64_t)i : JB.q_off + (uint64_t)i];
		uint32_t T[NC], U[NC], V[NC], X[NC], Y[NC], X2[NC], Y2[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int t = c * 64 + lane;
			uint32_t bA = 4, bB = 4;
			if (t < tlenA) {
				const uint64_t pos = (JA.flag & KSWJ_T_REVERSED) ? JA.t_off - (uint64_t)t : JA.t_off + (uint64_t)t;
				bA = (JA.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
				tbA[t] = (uint8_t)bA;
			}
			if (t < tlenB) {
				const uint64_t pos = (JB.flag & KSWJ_T_REVERSED) ? JB.t_off - (uint64_t)t : JB.t_off + (uint64_t)t;
				bB = (JB.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
				tbB[t] = (uint8_t)bB;
			}
			T[c] = bA | bB << 16;
			U[c] = V[c] = X[c] = Y[c] = P_NQE, X2[c] = Y2[c] = P_NQE2; // ksw2_extd2_sse.c:111-116
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();

		// Score of each job's corner cell H(tlen-1, qlen-1): the reference's approximate-score walk (ksw2_extd2_sse.c:366-383) sums
		// exact score differences along one monotone path and, without KSW_EZ_APPROX_DROP, reports only its path-independent end
		// point.  We sum along the matrix border: u of the first cell of every column along the top row (H(0,0) = u - (q+e) by the
		// border convention, :358), then v down the last column -- one scalar add per job and anti-diagonal.
		int H0A = -qe_in, H0B = -qe_in;
		const int n_rowsA = qlenA + tlenA - 1, n_rowsB = hasB ? qlenB + tlenB - 1 : 0, n_rows = n_rowsA > n_rowsB ? n_rowsA : n_rowsB;
		const int last_setA = (tlenA - 1) >> 6, last_laneA = (tlenA - 1) & 63, last_setB = (tlenB - 1) >> 6, last_laneB = (tlenB - 1) & 63;
		for (int r = 0; r < n_rows; ++r) {
			// valid cells of this anti-diagonal, per job (empty once a job has run out of rows)
			int st0A = r - qlenA + 1 > 0 ? r - qlenA + 1 : 0, en0A = r < tlenA - 1 ? r : tlenA - 1;
			int st0B = r - qlenB + 1 > 0 ? r - qlenB + 1 : 0, en0B = r < tlenB - 1 ? r : tlenB - 1;
			if (r >= n_rowsA) st0A = 1, en0A = 0;
			if (r >= n_rowsB) st0B = 1, en0B = 0;
			const uint32_t wA = (uint32_t)(en0A - st0A + 1), wB = (uint32_t)(en0B - st0B + 1); // widths (0 when empty)
			const int lo = st0A <= en0A ? (st0B <= en0B && st0B < st0A ? st0B : st0A) : st0B, hi = en0A > en0B ? en0A : en0B; // union (hi < lo if both empty)
			// value of v[-1] / u[r] on the matrix border (ksw2_extd2_sse.c:148-163): depends on r only, so it is shared
			const int bnd = r == 0 ? nqe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
			const uint32_t P_BND = pk2(bnd);
			const bool topA = r < tlenA && r < n_rowsA, topB = r < tlenB && r < n_rowsB; // the anti-diagonal still starts a new column (t = r)
			const int edge_set = r >> 6, edge_lane = r & 63;
			const uint32_t edge_halves = (topA ? 0xffffu : 0u) | (topB ? 0xffff0000u : 0u);
			uint8_t *prA = dirA + (size_t)r * ncolA, *prB = dirB + (size_t)r * ncolB;
			// register sets from the highest down, so that set c still sees row r-1 in set c-1 when it fetches its carry-ins
#pragma unroll
			for (int c = NC - 1; c >= 0; --c) {
				if (c * 64 > hi || c * 64 + 63 < lo) continue; // register set outside both anti-diagonals (uniform)
				const int t = c * 64 + lane;
				// one unsigned compare per job: st0 <= t <= en0 (an empty interval has en0 - st0 = -1 -> no lane passes... as unsigned it
				// would pass everything, so empty intervals are encoded with wA/wB = 0 and an impossible start)
				const bool actA = (uint32_t)(t - st0A) < wA, actB = (uint32_t)(t - st0B) < wB;
				uint32_t cV = P_BND, cX = P_NQE, cX2 = P_NQE2; // column -1: the matrix border
				if (c > 0) cV = __builtin_amdgcn_readlane(V[c - 1], 63), cX = __builtin_amdgcn_readlane(X[c - 1], 63), cX2 = __builtin_amdgcn_readlane(X2[c - 1], 63);
				const uint32_t vp = dpp_shr1u(cV, V[c]), xp = dpp_shr1u(cX, X[c]), x2p = dpp_shr1u(cX2, X2[c]);
				if (edge_halves && edge_set == c) { // u[r], y[r], y2[r] take their border values on first use (:156-163)
					const uint32_t em = lane == edge_lane ? edge_halves : 0u;
					U[c] = bfi(em, P_BND, U[c]), Y[c] = bfi(em, P_NQE, Y[c]), Y2[c] = bfi(em, P_NQE2, Y2[c]);
				}
				{
					// Every lane of the set computes, active or not: a column's registers are only ever read while the column (or its
					// right neighbour's next cell) is valid -- a column that has not started yet gets u,y,y2 from the border and x,v from
					// its left neighbour on its first cell, a finished one is never looked at again -- so whatever idle lanes leave in
					// their registers is harmless, and no per-lane masking of the state update is needed.  Only the stores are guarded.
					int rt = r - t;
					rt = rt < 0 ? 0 : rt > FAST_QCAP - 1 ? FAST_QCAP - 1 : rt;
					const uint32_t qv = (uint32_t)qbA[rt] | (uint32_t)qbA[rt + FAST_QCAP] << 16, tv = T[c]; // qbB = qbA + FAST_QCAP
					// substitution score: match / mismatch, overridden by sc_N when either base is ambiguous (code 4: bit 2)
					uint32_t z = pk_mad(pk_minu(tv ^ qv, P_ONE), P_MISD, P_MCH);
					z = pk_mad(pk_shr2(tv | qv), pk_sub(P_SCN, z), z);
					const uint32_t ut = U[c];
					uint32_t a = pk_add(xp, vp), b = pk_add(Y[c], ut), a2 = pk_add(x2p, vp), b2 = pk_add(Y2[c], ut);
					const uint32_t z1 = pk_max(z, a), z2 = pk_max(z1, b), z3 = pk_max(z2, a2), z4 = pk_max(z3, b2);
					// d = index of the first of (s, a, b, a2, b2) equal to the maximum
					const uint32_t ne_s = pk_minu(pk_sub(z4, z), P_ONE), ne_a = pk_minu(pk_sub(z4, a), P_ONE);
					const uint32_t ne_b = pk_minu(pk_sub(z4, b), P_ONE), ne_a2 = pk_minu(pk_sub(z4, a2), P_ONE);
					uint32_t d = pk_mul(ne_s, pk_mad(ne_a, pk_mad(ne_b, pk_add(ne_a2, P_ONE), P_ONE), P_ONE));
					z = pk_min(z4, P_MCH);
					U[c] = pk_sub(z, vp), V[c] = pk_sub(z, ut);
					uint32_t tmp = pk_sub(z, P_Q);
					a = pk_sub(a, tmp), b = pk_sub(b, tmp);
					tmp = pk_sub(z, P_Q2);
					a2 = pk_sub(a2, tmp), b2 = pk_sub(b2, tmp);
					const uint32_t ma = pk_max(a, P_ZERO), mb = pk_max(b, P_ZERO), ma2 = pk_max(a2, P_ZERO), mb2 = pk_max(b2, P_ZERO);
					d = pk_mad(pk_minu(ma, P_ONE), P_8, d);   // a > 0: the gap can be extended (continuation bits 0x08..0x40, :261-272)
					d = pk_mad(pk_minu(mb, P_ONE), P_16, d);
					d = pk_mad(pk_minu(ma2, P_ONE), P_32, d);
					d = pk_mad(pk_minu(mb2, P_ONE), P_64, d);
					X[c] = pk_sub(ma, P_QE), Y[c] = pk_sub(mb, P_QE), X2[c] = pk_sub(ma2, P_QE2), Y2[c] = pk_sub(mb2, P_QE2);
					if (actA) prA[(uint32_t)t] = (uint8_t)d;
					if (actB) prB[(uint32_t)t] = (uint8_t)(d >> 16);
				}
				if (topA) { if (edge_set == c) H0A += (int16_t)__builtin_amdgcn_readlane(U[c], edge_lane); }
				else if (r < n_rowsA && last_setA == c) H0A += (int16_t)__builtin_amdgcn_readlane(V[c], last_laneA);
				if (topB) { if (edge_set == c) H0B += (int16_t)(__builtin_amdgcn_readlane(U[c], edge_lane) >> 16); }
				else if (r < n_rowsB && last_setB == c) H0B += (int16_t)(__builtin_amdgcn_readlane(V[c], last_laneB) >> 16);
			}
		}
		// ---- tracebacks from (tlen-1, qlen-1) (ksw2_extd2_sse.c:389-391; ksw_backtrack with every cell inside the matrix) and
		//      mm_test_zdrop's scan of the result (align.c:61-84): lanes 0-31 serve job A, lanes 32-63 job B, concurrently ----
		__threadfence_block();
		const bool isB = lane >= 32;
		const int my_qlen = isB ? qlenB : qlenA, my_tlen = isB ? tlenB : tlenA, my_ncol = isB ? ncolB : ncolA;
		const uint8_t *my_dir = isB ? dirB : dirA, *my_qb = isB ? qbB : qbA, *my_tb = isB ? tbB : tbA;
		FastCig g = { L.cigar_tmp + (size_t)(2 * slot + (isB ? 1 : 0)) * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;
		int32_t zd_max = 0, zd_t0 = -1, zd_t1 = -1, zd_q0 = -1, zd_q1 = -1;
		{ // Each half-wave follows its job's path: lane k of the half looks k cells ahead along the current run (match diagonal or
		  // gap) and one ballot tells how far the run goes, so a read's typical 8-base match runs cost one load round, not eight.
			const bool have = !isB || hasB;
			const int hl = lane & 31;
			int i = my_tlen - 1, j = my_qlen - 1, state = 0;
			bool live = have && i >= 0 && j >= 0; // uniform within a half
			while (__ballot(live) != 0ull) {
				const int di = (state == 2 || state == 4) ? 0 : 1, dj = (state == 1 || state == 3) ? 0 : 1;
				const int ii = i - hl * di, jj = j - hl * dj;
				const bool valid = live && ii >= 0 && jj >= 0;
				const int tmp = valid ? my_dir[(size_t)(ii + jj) * my_ncol + ii] : 0;
				const bool cont = valid && (state == 0 ? (tmp & 7) == 0 : (tmp >> (state + 2) & 1) != 0);
				const unsigned long long bal = __ballot(cont);
				const uint32_t mine = isB ? (uint32_t)(bal >> 32) : (uint32_t)bal;
				const int run = mine == 0xffffffffu ? 32 : __builtin_ctz(~mine);
				const int head = __shfl(tmp, lane & 32, 64); // the cell the half stands on
				if (live) {
					if (run > 0) {
						fast_cig_push(g, state == 0 ? 0u : (state == 1 || state == 3) ? 2u : 1u, run);
						i -= run * di, j -= run * dj;
					} else { // the run ends on this cell: it names the next state (ksw2.h:141-144)
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
		if ((lane == 0 || (lane == 32 && hasB))) {
			if (g.n > 0) g.c[g.n - 1] = g.last;
			if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
			// update_max_zdrop (align.c:46-59) over the alignment, start to end (g.c holds the operations last first)
			int32_t score = 0, mx = INT32_MIN, mx_i = -1, mx_j = -1, ci = 0, cj = 0;
			const int gq = L.sc.q, ge = L.sc.e;
			auto track = [&](int32_t sc, int pi, int pj) {
				if (sc < mx) {
					const int li = pi - mx_i, lj = pj - mx_j, diff = li > lj ? li - lj : lj - li, zz = mx - sc - diff * ge;
					if (zz > zd_max) zd_max = zz, zd_t0 = mx_i, zd_t1 = pi, zd_q0 = mx_j, zd_q1 = pj;
				} else mx = sc, mx_i = pi, mx_j = pj;
			};
			for (int k = g.n - 1; k >= 0; --k) {
				const uint32_t op = g.c[k] & 0xf, len = g.c[k] >> 4;
				if (op == 0) {
					for (uint32_t l = 0; l < len; ++l) { score += s_mat[my_tb[ci + l] * 5 + my_qb[cj + l]]; track(score, ci + (int)l, cj + (int)l); }
					ci += len, cj += len;
				} else {
					score -= gq + ge * (int)len;
					if (op == 1) cj += len; else ci += len;
					track(score, ci, cj);
				}
			}
		}
		__threadfence_block();
		// pack the CIGARs into the pool in forward order, job A then job B, all lanes copying
#pragma unroll
		for (int which = 0; which < 2; ++which) {
			if (which == 1 && !hasB) break;
			const int src = which * 32;
			const int n_cig = __builtin_amdgcn_readlane(g.n, src);
			const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)cig_off, src);
			const uint32_t *tmpc = L.cigar_tmp + (size_t)(2 * slot + which) * L.cigar_tmp_cap;
			if (n_cig > 0) {
				if ((unsigned long long)off + (unsigned)n_cig > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
				else for (int k = lane; k < n_cig; k += 64) L.cigar_pool[off + k] = tmpc[n_cig - 1 - k];
			}
		}
		if (lane == 0 || (lane == 32 && hasB)) {
			KswRes R;
			R.max = 0, R.zdropped = 0, R.max_q = R.max_t = -1, R.mqe = R.mte = KSW_NEG_INF, R.mqe_t = R.mte_q = -1;
			R.score = isB ? H0B : H0A, R.n_cigar = g.n, R.reach_end = 0, R.cigar_off = cig_off;
			R.zd_max = zd_max, R.zd_t0 = zd_t0, R.zd_t1 = zd_t1, R.zd_q0 = zd_q0, R.zd_q1 = zd_q1;
			L.res[isB ? jidB : jidA] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}

void ksw_fast_launch(const KswLaunch &L, int n_slots, int n_sets, void *stream)
{
	if (L.n_jobs <= 0) return;
	const int n_blocks = (n_slots + 3) / 4;
	hipStream_t s = (hipStream_t)stream;
	switch (n_sets) {
	case 2: hipLaunchKernelGGL((ksw_fast_kernel<2>), dim3(n_blocks), dim3(256), 0, s, L); break;
	case 3: hipLaunchKernelGGL((ksw_fast_kernel<3>), dim3(n_blocks), dim3(256), 0, s, L); break;
	case 4: hipLaunchKernelGGL((ksw_fast_kernel<4>), dim3(n_blocks), dim3(256), 0, s, L); break;
	case 5: hipLaunchKernelGGL((ksw_fast_kernel<5>), dim3(n_blocks), dim3(256), 0, s, L); break;
	case 6: hipLaunchKernelGGL((ksw_fast_kernel<6>), dim3(n_blocks), dim3(256), 0, s, L); break;
	case 8: hipLaunchKernelGGL((ksw_fast_kernel<8>), dim3(n_blocks), dim3(256), 0, s, L); break;
	default: throw std::runtime_error("[mm2amd] ksw_fast_launch: unsupported register-set count");
	}
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

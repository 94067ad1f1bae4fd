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
//   * lane = target position: column t lives in lane t%64 of register set (t%256)/64 for a whole strip of 256 columns, so the
//     six difference states (u,v,x,y,x2,y2) never leave VGPRs; the t-1 neighbour comes from a DPP wave shift (lane 0 takes the
//     last lane of the previous register set through a DPP wave rotate -- no v_readlane, no SGPR round trip), the query base
//     from a byte in LDS.  No LDS round trip and no barrier per row.
//   * two jobs per wavefront in the halves of packed 16-bit registers (below), persistent waves pulling job pairs from a queue.
//   * targets wider than 256 columns are swept in strips of 256: a strip leaves (v, x, x2) of its last column, one entry per
//     row, in LDS for the next one.  Every job therefore runs in the same 4-register-set kernel at 80 VGPRs: six waves per
//     SIMD whatever the target length (round 1 compiled one kernel per width, the wide ones at two and three waves per SIMD).
//     Six and eight waves per SIMD (a 64-VGPR build) measured the same; DESIGN.md section 7 has the issue-rate measurements.
//   * the only HBM traffic is the 1 B/cell direction matrix, written as DWORDS: a lane owns one column, so the bytes of two
//     consecutive anti-diagonals of both jobs of the pair ([row r: A, B][row r+1: A, B]) form one dword per lane -- 256 B
//     contiguous per register set and store instruction, a quarter of the store instructions of byte stores and full lines
//     (round 1 stored 64-byte row segments: 1.9x write amplification in the PMC counters).
//
// Since ksw_stream.hip the gap fills with query <= 512 and target <= 512 (97 % of the DP cells of the map-ont benchmark) take the streaming
// kernel, which runs the same cell body (gf_cell) with the jobs of a wave back to back through the lanes; this kernel keeps the
// longer ones (queries <= 1024, targets <= 3072) and is the A/B partner of the streaming kernel (MM2AMD_NO_STREAM=1).  The Z-drop
// walk over a finished alignment is gf_zdrop_scan (ksw_gapfill_dev.hpp), shared by both.
//
// tests/test_gpu_ksw.py checks this kernel against the lane-exact oracle on every preset's scoring; jobs that are not
// eligible (binding band, extension flags, exact-max mode, generic matrices, very long sequences) take the exact kernel.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"
#include "ksw_gapfill_dev.hpp"

namespace mm2amd {

constexpr int GF_NC = 4;          // register sets of 64 columns
constexpr int GF_STRIP = 64 * GF_NC;

// ---------------------------------------------------------------------------------------------------------
// Two jobs per wavefront, packed 16-bit arithmetic.
//
// Every quantity of the row loop fits in 8 bits, and gfx950 executes 2 x 16-bit packed integer ops per lane and instruction, so
// each lane carries the same column of TWO jobs: job A in the low halves of the state registers, job B in the high halves.  The
// jobs of a launch are ordered by cost, so the two members of a pair have nearly the same shape and advance in lockstep; DPP
// shifts, carries and border values act on both halves at once, and only the query bytes are per job.  The direction index d
// ("first candidate that reaches the maximum", = the reference's strictly-greater update chain, ksw2_extd2_sse.c:235-243) and
// the continuation flags are computed arithmetically, because packed compares do not exist.  The packed instructions are
// issued through inline asm: written as C++ vector code the optimiser rewrites the min/mul idioms back into compares and
// de-vectorises them.
//
// QCAP = longest query of the launch class (512: 16 KB of LDS per block, six blocks per CU; 1024: four).
// ---------------------------------------------------------------------------------------------------------
template <int QCAP, int WAVES>
__global__ void __launch_bounds__(256, WAVES) ksw_gapfill_kernel(KswLaunch L)
{
	__shared__ uint8_t s_q[4][2][QCAP];       // query bytes of the pair
	__shared__ uint16_t s_b[4][3][QCAP];      // strip boundary: (v, x, x2) of a strip's last column per query position, A | B << 8; the target bytes for the Z-drop scan afterwards
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
	// the keyed cell (gf_cell_k, ksw_gapfill_dev.hpp): every score difference times 8, the gap states carry their candidate's tag in the low bits
	const GfK K = gf_k_consts(sc_mch, sc_mis, sc_N, q, e, q2, e2);
	const uint32_t S_NQE_X = K.nqe_x, S_NQE_Y = K.nqe_y, S_NQE2_X = K.nqe2_x, S_NQE2_Y = K.nqe2_y;
	const uint32_t P_MCH = pk2v(8 * sc_mch + GF_K_TS);
	const uint32_t lane4 = (uint32_t)lane * 4u;
	uint8_t *const qb = s_q[wave_in_block][0];             // qbB = qb + QCAP
	const uint8_t *const s_qflat = &s_q[0][0][0];
	const int qb_addr = wave_in_block * 2 * QCAP;
	const uint32_t qb_last = (uint32_t)(qb_addr + QCAP - 1);
	uint16_t *const bV = s_b[wave_in_block][0], *const bX = s_b[wave_in_block][1], *const bX2 = s_b[wave_in_block][2];
	uint8_t *const tbA = (uint8_t *)bV, *const tbB = tbA + 3 * QCAP; // after the DP (each job's target is at most 3 * QCAP bases)

	const int n_avail = L.n_list ? __builtin_amdgcn_readfirstlane(*L.n_list) : L.n_jobs;
	for (;;) {
		int pid = 0;
		if (lane == 0) pid = atomicAdd(L.counter, 1);
		pid = __builtin_amdgcn_readfirstlane(pid);
		if (2 * pid >= n_avail) break;
		const bool hasB = 2 * pid + 1 < n_avail;
		// (a launch that works off a list -- the banded kernel's rejects, ksw_dev.hpp -- finds its jobs through it)
		const int jidA = L.list ? __builtin_amdgcn_readfirstlane((int)L.list[2 * pid]) : 2 * pid;
		const int jidB = !hasB ? jidA : L.list ? __builtin_amdgcn_readfirstlane((int)L.list[2 * pid + 1]) : 2 * pid + 1;
		const KswJob JA = L.jobs[jidA], JB = L.jobs[jidB];
		const int qlenA = JA.qlen, tlenA = JA.tlen, qlenB = hasB ? JB.qlen : 0, tlenB = hasB ? JB.tlen : 0;
		const int tmax = tlenA > tlenB ? tlenA : tlenB, qmax = qlenA > qlenB ? qlenA : qlenB;
		const int ncol = (tmax + 63) & ~63;  // columns of the shared direction matrix: dword (r >> 1) * ncol + t = [row r: A, B][row r + 1: A, B]
		uint8_t *const dir = L.dir_pool + (size_t)(2 * slot) * L.slot_bytes;
		for (int i = lane; i < qlenA; i += 64) qb[i] = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)i : JA.q_off + (uint64_t)i];
		for (int i = lane; i < qlenB; i += 64) qb[QCAP + i] = L.qpool[(JB.flag & KSWJ_Q_REVERSED) ? JB.q_off - (uint64_t)i : JB.q_off + (uint64_t)i];
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();

		// Score of each job's corner cell H(tlen-1, qlen-1): the reference's approximate-score walk (ksw2_extd2_sse.c:366-383) sums
		// exact score differences along one monotone path and, without KSW_EZ_APPROX_DROP, reports only its path-independent end
		// point, i.e. the optimal global score -- which is the score of the path the traceback follows.  The scan after the
		// traceback sums it (substitution scores and the cheaper of the two gap cost models per gap run).  The reference starts its
		// sum from v[0] - (q + e) with the gap costs as PASSED, while the border holds the swapped ones (:68 vs :78): when the caller's
		// second cost pair is the cheaper one its score is short by the difference, and so is ours.
		const int n_rowsA = qlenA + tlenA - 1, n_rowsB = hasB ? qlenB + tlenB - 1 : 0, n_rows = n_rowsA > n_rowsB ? n_rowsA : n_rowsB;
		const int n_strips = (tmax + GF_STRIP - 1) / GF_STRIP;

		for (int s = 0; s < n_strips; ++s) {
			const int cb = s * GF_STRIP;              // first column of the strip
			const bool more = s + 1 < n_strips;       // a later strip reads this one's last column
			uint32_t T[GF_NC], U[GF_NC], V[GF_NC], X[GF_NC], Y[GF_NC], X2[GF_NC], Y2[GF_NC], DE[GF_NC];
#pragma unroll
			for (int c = 0; c < GF_NC; ++c) {
				const int t = cb + c * 64 + lane;
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
				// whatever a column's registers hold before its first cell is dead: u, y, y2 take their border values on first use, x and v
				// are written by the column's first cell before its right neighbour reads them (ksw2_extd2_sse.c:111-116 fills them all)
				U[c] = V[c] = X[c] = Y[c] = X2[c] = Y2[c] = DE[c] = 0u;
			}
			// rows that touch the strip: column cb starts at row cb, column cb + 255 ends at row cb + 255 + qlen - 1; rows go in pairs
			// (even, odd) because two rows share a direction dword.  Rows before a column's first or after its last cell compute dead values.
			const int r_begin = cb, r_last = cb + GF_STRIP - 1 + qmax - 1 < n_rows - 1 ? cb + GF_STRIP - 1 + qmax - 1 : n_rows - 1;
			for (int r0 = r_begin; r0 <= r_last; r0 += 2) {
				// union of the two rows' valid cells over both jobs (st/en never decrease with r): the register sets that meet it compute
				// both rows; its lanes store the dword
				int lo2, hi2;
				{
					const int stA = r0 - qlenA + 1 > 0 ? r0 - qlenA + 1 : 0, enA = r0 + 1 < tlenA - 1 ? r0 + 1 : tlenA - 1;
					const int stB = r0 - qlenB + 1 > 0 ? r0 - qlenB + 1 : 0, enB = r0 + 1 < tlenB - 1 ? r0 + 1 : tlenB - 1;
					const bool okA = r0 < n_rowsA && stA <= enA, okB = r0 < n_rowsB && stB <= enB;
					lo2 = okA ? (okB && stB < stA ? stB : stA) : okB ? stB : 1;
					hi2 = okA ? (okB && enB > enA ? enB : enA) : okB ? enB : 0;
				}
				uint32_t *const prow = (uint32_t *)(dir + ((size_t)(r0 >> 1) * (size_t)ncol + (size_t)cb) * 4u);
#pragma unroll
				for (int par = 0; par < 2; ++par) {
					const int r = r0 + par;
					// value of v[-1] / u[r] on the matrix border (ksw2_extd2_sse.c:148-163): depends on r only, so it is shared
					const int bnd = 8 * (r == 0 ? nqe : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
					const uint32_t S_BND = pk2(bnd);
					const bool topA = r < tlenA && r < n_rowsA, topB = r < tlenB && r < n_rowsB; // the anti-diagonal still starts a new column (t = r)
					const int edge_set = (r - cb) >> 6, edge_lane = r & 63;                      // (r - cb) >> 6 is outside 0..3 when column r is not in this strip
					const uint32_t edge_halves = (topA ? 0xffffu : 0u) | (topB ? 0xffff0000u : 0u);
					// column cb - 1 (the matrix border, or the previous strip's last column at the same query position)
					uint32_t c0V = S_BND, c0X = S_NQE_X, c0X2 = S_NQE2_X;
					if (s > 0) {
						const uint32_t bi = (uint32_t)(r - cb) < (uint32_t)QCAP ? (uint32_t)(r - cb) : (uint32_t)(QCAP - 1);
						const uint32_t pv = bV[bi], px = bX[bi], px2 = bX2[bi];
						// (the boundary holds the differences themselves, one byte per job: times 8 and tagged again here)
						c0V = gf_mad8(gf_sext8(__builtin_amdgcn_perm(0u, pv, 0x0c010c00u)), 0u), c0X = gf_mad8(gf_sext8(__builtin_amdgcn_perm(0u, px, 0x0c010c00u)), pk2(GF_K_TA));
						c0X2 = gf_mad8(gf_sext8(__builtin_amdgcn_perm(0u, px2, 0x0c010c00u)), pk2(GF_K_TA2));
					}
					// register sets from the highest down, so that set c still sees row r-1 in set c-1 when it fetches its carry-ins
#pragma unroll
					for (int c = GF_NC - 1; c >= 0; --c) {
						if (cb + c * 64 > hi2 || cb + c * 64 + 63 < lo2) continue; // register set outside the anti-diagonals of this row pair (uniform)
						uint32_t cV = c0V, cX = c0X, cX2 = c0X2;
						if (c > 0) cV = gf_ror1(V[c - 1]), cX = gf_ror1(X[c - 1]), cX2 = gf_ror1(X2[c - 1]); // lane 0 <- lane 63 of the previous set
						const uint32_t vp = dpp_shr1u(cV, V[c]), xp = dpp_shr1u(cX, X[c]), x2p = dpp_shr1u(cX2, X2[c]);
						if (edge_halves && edge_set == c) { // u[r], y[r], y2[r] take their border values on first use (:156-163)
							const uint32_t em = lane == edge_lane ? edge_halves : 0u;
							U[c] = bfi(em, S_BND, U[c]), Y[c] = bfi(em, S_NQE_Y, Y[c]), Y2[c] = bfi(em, S_NQE2_Y, Y2[c]);
						}
						// Every lane of the set computes, active or not: a column's registers are only ever read while the column (or its
						// right neighbour's next cell) is valid -- a column that has not started yet gets u,y,y2 from the border and x,v from
						// its left neighbour on its first cell, a finished one is never looked at again -- so whatever idle lanes leave in
						// their registers is harmless, and no per-lane masking of the state update is needed.  Only the stores are guarded.
						// query position of this column's cell, as an LDS address.  Positions outside the query belong to dead cells: beyond the
						// end they are clamped, before the start they read the bytes in front of this wave's buffer (a wave's buffers are
						// 2 * QCAP >= 1024 bytes apart, the first one wraps around to the clamp)
						uint32_t qa = (uint32_t)(qb_addr + r - cb - c * 64) - (uint32_t)lane;
						qa = qa < qb_last ? qa : qb_last;
						const uint32_t qv = (uint32_t)s_qflat[qa] | (uint32_t)s_qflat[qa + QCAP] << 16, tv = T[c];
						// substitution score (match / mismatch, sc_N when either base is ambiguous: code 4 = bit 2), the five candidates as keys, the
						// direction = the tag of the maximum = first of (s, a, b, a2, b2) equal to it, the continuation bits (:235-272; gf_cell_k)
						uint32_t d;
						gf_cell_k(tv ^ qv, tv | qv, xp, vp, x2p, U[c], V[c], X[c], Y[c], X2[c], Y2[c], d, P_MCH, K);
						if (par == 0) DE[c] = d;
						else {
							const uint32_t t = (uint32_t)(cb + c * 64 + lane);
							if (t - (uint32_t)lo2 <= (uint32_t)(hi2 - lo2))
								*(uint32_t *)((uint8_t *)prow + c * 256 + lane4) = __builtin_amdgcn_perm(d, DE[c], 0x06040200u); // [even A, even B, odd A, odd B]
						}
					}
					if (more) { // leave the strip's last column for the next strip: entry r - (cb + 255) = the query position of that cell
						const uint32_t bi = (uint32_t)(r - (cb + GF_STRIP - 1));
						if (bi < (uint32_t)QCAP && lane == 63) {
							bV[bi] = (uint16_t)__builtin_amdgcn_perm(0u, gf_asr3(V[GF_NC - 1]), 0x0c0c0200u); // key >> 3 (arithmetic) = the difference: the tag falls off
							bX[bi] = (uint16_t)__builtin_amdgcn_perm(0u, gf_asr3(X[GF_NC - 1]), 0x0c0c0200u);
							bX2[bi] = (uint16_t)__builtin_amdgcn_perm(0u, gf_asr3(X2[GF_NC - 1]), 0x0c0c0200u);
						}
					}
				}
			}
			if (more) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
		}
		// ---- the targets as bytes for the Z-drop scan (the boundary entries are dead now) ----
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		for (int t = lane; t < tlenA; t += 64) {
			const uint64_t pos = (JA.flag & KSWJ_T_REVERSED) ? JA.t_off - (uint64_t)t : JA.t_off + (uint64_t)t;
			tbA[t] = (uint8_t)((JA.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos]);
		}
		for (int t = lane; t < tlenB; t += 64) {
			const uint64_t pos = (JB.flag & KSWJ_T_REVERSED) ? JB.t_off - (uint64_t)t : JB.t_off + (uint64_t)t;
			tbB[t] = (uint8_t)((JB.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos]);
		}
		// ---- tracebacks from (tlen-1, qlen-1) (ksw2_extd2_sse.c:389-391; ksw_backtrack with every cell inside the matrix) and
		//      mm_test_zdrop's scan of the result (align.c:61-84): lanes 0-31 serve job A, lanes 32-63 job B, concurrently ----
		__threadfence_block();
		const bool isB = lane >= 32;
		const int my_qlen = isB ? qlenB : qlenA, my_tlen = isB ? tlenB : tlenA;
		const uint8_t *my_qb = isB ? qb + QCAP : qb, *my_tb = isB ? tbB : tbA;
		FastCig g = { L.cigar_tmp + (size_t)(2 * slot + (isB ? 1 : 0)) * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;
		int32_t zd_max = 0, zd_t0 = -1, zd_t1 = -1, zd_q0 = -1, zd_q1 = -1, dp_score = qe - qe_in;
		{ // each half-wave follows its job's path (gf_traceback, ksw_gapfill_dev.hpp)
			const size_t hoff = isB ? 1 : 0;
			gf_traceback(!isB || hasB, my_tlen - 1, my_qlen - 1, [&](int ii, int jj) {
				const int rr = ii + jj;
				return gf_k_decode(dir[((size_t)(rr >> 1) * (size_t)ncol + (size_t)ii) * 4u + (size_t)((rr & 1) << 1) + hoff], K.bias);
			}, g);
		}
		if ((lane == 0 || (lane == 32 && hasB))) {
			if (g.n > 0) g.c[g.n - 1] = g.last;
			if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		{ // mm_test_zdrop's walk over both alignments (align.c:46-84) and their scores under the DP's own costs, 32 lanes each
			const uint32_t third = L.cigar_tmp_cap / 3u; // a job's scratch: its operations (last first), then two prefix arrays
			const GfZdrop z = gf_zdrop_scan(!isB || hasB, g.n, g.c, g.c + third, g.c + 2u * third, [&](int i) { return (int)my_tb[i]; }, [&](int j) { return (int)my_qb[j]; },
			                                s_mat, L.sc.q, L.sc.e, L.sc.q2, L.sc.e2, sc_N);
			zd_max = z.zd_max, zd_t0 = z.t0, zd_t1 = z.t1, zd_q0 = z.q0, zd_q1 = z.q1, dp_score += z.dp_sum;
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
			R.score = dp_score, R.n_cigar = g.n, R.reach_end = 0, R.cigar_off = cig_off;
			R.zd_max = zd_max, R.zd_t0 = zd_t0, R.zd_t1 = zd_t1, R.zd_q0 = zd_q0, R.zd_q1 = zd_q1;
			L.res[isB ? jidB : jidA] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}

void ksw_gapfill_launch(const KswLaunch &L, int n_slots, int qcap, void *stream)
{
	if (L.n_jobs <= 0 && !L.n_list) return;
	const int n_blocks = (n_slots + 3) / 4;
	hipStream_t s = (hipStream_t)stream;
	if (qcap <= 512) hipLaunchKernelGGL((ksw_gapfill_kernel<512, 6>), dim3(n_blocks), dim3(256), 0, s, L);
	else if (qcap <= 1024) hipLaunchKernelGGL((ksw_gapfill_kernel<1024, 4>), dim3(n_blocks), dim3(256), 0, s, L);
	else throw std::runtime_error("[mm2amd] ksw_gapfill_launch: unsupported query capacity");
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

// Streaming gap-fill DP for gfx950: ksw_gapfill.hip's register-resident kernel with the jobs of a wavefront run BACK TO BACK through
// the lanes instead of one after the other.
//
// An anti-diagonal sweep of a q x t matrix with lane = target column keeps lane t busy for q consecutive rows (t .. t + q - 1) of the
// q + t - 1 the job takes: the ramps at both ends leave register sets partly empty, and over the gap fills of a read batch only
// 73 % of the computed lanes hold a cell.  Here lane t moves on to column t of the NEXT job in the row after
// its last cell of the current one.  A wave works on PAIRS of jobs (one per half of the packed 16-bit registers, consecutive in the
// launch order, started on the same row -- so that both finish together and are traced back by the two half-waves at once): pair
// k+1 starts at row R[k+1] = R[k] + max(q, t over pair k), its first anti-diagonals fill the lanes pair k's last ones have left,
// and a lane idles only while the jobs it is on are narrower or shorter than its column and the pair's longest side (counted:
// 0.73 -> 0.86 / 0.87 of the computed lanes hold a cell).  Everything a lane needs changes hands when the next pair's first
// anti-diagonal reaches it -- the "edge": its target bases (T <- s_tn), and u / y / y2, which take their border values there
// exactly as on the first use of a column in the one-job kernel (ksw2_extd2_sse.c:156-163); x / v / x2 arrive from the left
// neighbour as always, lane 0 takes the matrix border of the pair it is on.  The query bytes of the pairs in flight sit in an LDS
// ring addressed by row - column (both halves in one 16-bit load), the direction bytes ([row r: half A, half B][row r + 1: A, B]
// per lane) in a ring of the last 1024 or 2048 rows in HBM; a pair is traced back (and walked for mm_test_zdrop, gf_zdrop_scan)
// right after its last row, while its successor is already under way.  The arithmetic of a cell is gf_cell_k: the strip kernel's cell with keyed
// candidates (differences times 8, the direction read off the maximum's low bits: ksw_gapfill_dev.hpp), 38 packed operations instead of 50.
//
// At most two pairs are in flight (cur, nxt): nxt is promoted once all of its columns have seen its edge and cur has been traced
// back; only then is the pair after it fetched (its targets need the s_tn buffer).  Per row pair a bit mask names the register sets
// that hold a valid cell of either pair; the others are skipped.  Eligibility and exactness are those of ksw_gapfill.hip; this
// kernel takes the classes with query <= 512 and target <= 64 * NC, the strip kernel the rest.  tests/test_gpu_ksw.py streams
// hundreds of jobs of every shape through each half-wave (MM2AMD_KSW_MAX_SLOTS) against the lane-exact oracle.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"
#include "ksw_gapfill_dev.hpp"

namespace mm2amd {

constexpr int ST_QRING = 2048;    // query positions kept per wave: two pairs in flight plus the bubble between them stay below 1100 rows
constexpr int st_rows(int n_sets) { return n_sets <= 4 ? 1024 : 2048; } // direction rows kept per wave: a job spans query + target - 1 <= 767 (4 sets) or 1023 (8) rows and is traced back within two rows of its last
#ifndef ST_W4
#define ST_W4 4
#endif
constexpr int ST_TCAP = 512;      // target bytes of the jobs being walked for the Z-drop

// A job record read through a vector load sits in VGPRs, and everything computed from it (row counters, branch conditions) would
// follow it there: the fields are wave-uniform, say so.
__device__ __forceinline__ int st_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t st_uni64(uint64_t v)
{
	return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32;
}
__device__ __forceinline__ KswJob uniform_job(const KswJob &j)
{
	KswJob u = j;
	u.q_off = st_uni64(j.q_off), u.t_off = st_uni64(j.t_off), u.qlen = st_uni(j.qlen), u.tlen = st_uni(j.tlen), u.flag = st_uni(j.flag);
	return u;
}

template <int NC, int WAVES>
__global__ void __launch_bounds__(256, WAVES) ksw_stream_kernel(KswLaunch L)
{
	__shared__ uint16_t s_qr[4][ST_QRING];        // query bytes of the jobs in flight by row - column: half A | half B << 8
	__shared__ uint8_t s_tn[4][2][NC * 64];       // target bases of nxt, picked up by each lane as nxt's edge passes
	__shared__ uint8_t s_tb[4][2][ST_TCAP];
	__shared__ int8_t s_mat[32];
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	const int slot = blockIdx.x * 4 + wave_in_block;
	if (threadIdx.x < 25) s_mat[threadIdx.x] = L.sc.mat[threadIdx.x];
	__syncthreads();
	// (a value, not a kernel-argument load the compiler may repeat inside a divergent branch); a launch fed by a device-made list (the banded kernel's
	// rejects, ksw_band.hip) reads its job count here and maps queue positions through the list
	const int m = L.sc.m, n_jobs = L.n_list ? st_uni(*L.n_list) : st_uni(L.n_jobs);
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
	constexpr int ncol = NC * 64, ST_ROWS = st_rows(NC);
	uint8_t *const dir = L.dir_pool + (size_t)(2 * slot) * L.slot_bytes; // (ST_ROWS / 2) x ncol dwords
	uint8_t *const qring = (uint8_t *)&s_qr[wave_in_block][0];
	const uint32_t qring_off = (uint32_t)wave_in_block * (2u * ST_QRING), lane2 = (uint32_t)lane * 2u; // rings are 4 KB apart: offset | position
	auto border = [&](int i) { return 8 * (i == 0 ? nqe : i < long_thres ? -e : i == long_thres ? long_diff : -e2); }; // v[-1] / u[i] on the matrix border (:148-163), times 8

	// the two PAIRS of jobs in flight (a pair = one job per half, consecutive in the launch order, started on the same row): cur (all
	// its columns started) and nxt (its edge is sweeping the lanes, or it has not started); a half without a job has q = t = 0
	bool cv = false, nv = false;
	int cR = 0, nR = 0, cq[2] = { 0, 0 }, ct[2] = { 0, 0 }, cj[2] = { 0, 0 }, nq[2] = { 0, 0 }, nt[2] = { 0, 0 }, nj[2] = { 0, 0 };
	uint32_t T[NC], U[NC], V[NC], X[NC], Y[NC], X2[NC], Y2[NC], DE[NC];
#pragma unroll
	for (int c = 0; c < NC; ++c) T[c] = 0x00040004u, U[c] = V[c] = X[c] = Y[c] = X2[c] = Y2[c] = DE[c] = 0u;

	// the next pair, to start at row R: the targets into s_tn, the queries into the ring
	auto fetch = [&](int R) {
		int id = 0;
		if (lane == 0) id = atomicAdd(L.counter, 2);
		id = __builtin_amdgcn_readfirstlane(id);
		nv = id < n_jobs;
		if (!nv) return;
		nR = R;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			nj[h] = id + h, nq[h] = nt[h] = 0;
			if (id + h >= n_jobs) continue; // the last pair of an odd launch
			if (L.list) nj[h] = st_uni((int)L.list[id + h]);
			const KswJob J = uniform_job(L.jobs[nj[h]]);
			nq[h] = J.qlen, nt[h] = J.tlen;
#pragma unroll
			for (int c = 0; c < NC; ++c) {
				const int t = c * 64 + lane;
				uint32_t b = 4;
				if (t < J.tlen) {
					const uint64_t pos = (J.flag & KSWJ_T_REVERSED) ? J.t_off - (uint64_t)t : J.t_off + (uint64_t)t;
					b = (J.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos];
				}
				s_tn[wave_in_block][h][t] = (uint8_t)b;
			}
#pragma unroll 2
			for (int i = lane; i < J.qlen; i += 64) qring[((R + i) & (ST_QRING - 1)) * 2 + h] = L.qpool[(J.flag & KSWJ_Q_REVERSED) ? J.q_off - (uint64_t)i : J.q_off + (uint64_t)i];
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	};

	// traceback from (t - 1, q - 1) (ksw2_extd2_sse.c:389-391; ksw_backtrack with every cell inside the matrix) and mm_test_zdrop's scan
	// (align.c:61-84) of the pair that has just seen its last row; lanes 0-31 serve half A, lanes 32-63 half B, concurrently
	auto finish = [&]() {
		const bool doA = ct[0] > 0, doB = ct[1] > 0;
		__threadfence_block();
		const bool isB = lane >= 32;
		const int h = isB ? 1 : 0;
		const bool have = isB ? doB : doA;
		const int my_q = isB ? cq[1] : cq[0], my_t = isB ? ct[1] : ct[0], my_R = cR, my_id = isB ? cj[1] : cj[0];
#pragma unroll
		for (int hh = 0; hh < 2; ++hh) { // the target bytes for the scan
			if (!(hh ? doB : doA)) continue;
			const KswJob J = uniform_job(L.jobs[cj[hh]]);
			for (int t = lane; t < J.tlen; t += 64) {
				const uint64_t pos = (J.flag & KSWJ_T_REVERSED) ? J.t_off - (uint64_t)t : J.t_off + (uint64_t)t;
				s_tb[wave_in_block][hh][t] = (uint8_t)((J.flag & KSWJ_T_PACKED) ? (L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (uint32_t)L.tpool[pos]);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		const uint8_t *my_ring = qring + h, *my_tb = &s_tb[wave_in_block][h][0];
		FastCig g = { L.cigar_tmp + (size_t)(2 * slot + h) * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;
		int32_t zd_max = 0, zd_t0 = -1, zd_t1 = -1, zd_q0 = -1, zd_q1 = -1, dp_score = qe - qe_in; // the reference's score offset when the second cost pair is the cheaper one (:68 vs :78)
		{ // ring offsets fit 32 bits; the half's byte (A: 0, B: 1) is part of the offset, so the base stays wave-uniform
			const uint32_t hoff = (uint32_t)h;
			gf_traceback(have, my_t - 1, my_q - 1, [&](int ii, int jj) {
				const uint32_t rr = (uint32_t)(my_R + ii + jj);
				return gf_k_decode(dir[(((rr >> 1) & (uint32_t)(ST_ROWS / 2 - 1)) * (uint32_t)ncol + (uint32_t)ii) * 4u + ((rr & 1u) << 1) + hoff], K.bias);
			}, g);
		}
		if ((lane & 31) == 0 && have) {
			if (g.n > 0) g.c[g.n - 1] = g.last;
			if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		{ // mm_test_zdrop's walk over the alignment (align.c:46-84) and its score under the DP's own costs, by the 32 lanes of the half
			const uint32_t third = L.cigar_tmp_cap / 3u; // a job's scratch: its operations (last first), then two prefix arrays
			const GfZdrop z = gf_zdrop_scan(have, g.n, g.c, g.c + third, g.c + 2u * third, [&](int i) { return (int)my_tb[i]; },
			                                [&](int j) { return (int)my_ring[((my_R + j) & (ST_QRING - 1)) * 2]; }, s_mat, L.sc.q, L.sc.e, L.sc.q2, L.sc.e2, sc_N);
			zd_max = z.zd_max, zd_t0 = z.t0, zd_t1 = z.t1, zd_q0 = z.q0, zd_q1 = z.q1, dp_score += z.dp_sum;
		}
		__threadfence_block();
#pragma unroll
		for (int which = 0; which < 2; ++which) { // the CIGARs into the pool in forward order, all lanes copying
			if (!(which ? doB : doA)) continue;
			const int src = which * 32;
			const int n_cig = __builtin_amdgcn_readlane(g.n, src);
			const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)cig_off, src);
			const uint32_t *tmpc = L.cigar_tmp + (size_t)(2 * slot + which) * L.cigar_tmp_cap;
			if (n_cig > 0) {
				if ((unsigned long long)off + (unsigned)n_cig > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
				else for (int k = lane; k < n_cig; k += 64) L.cigar_pool[off + k] = tmpc[n_cig - 1 - k];
			}
		}
		if ((lane & 31) == 0 && have) {
			KswRes R;
			R.max = 0, R.zdropped = 0, R.max_q = R.max_t = -1, R.mqe = R.mte = KSW_NEG_INF, R.mqe_t = R.mte_q = -1;
			R.score = dp_score, R.n_cigar = g.n, R.reach_end = 0, R.cigar_off = cig_off;
			R.zd_max = zd_max, R.zd_t0 = zd_t0, R.zd_t1 = zd_t1, R.zd_q0 = zd_q0, R.zd_q1 = zd_q1;
			L.res[my_id] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	};

#ifdef MM2AMD_GF_COUNT // measurement build (tools/build_variant.sh, tools/lane_utilisation_emu.sh): executed register-set rows against cells, printed by two sample waves (=2: by every wave)
	long long n_setrows = 0, n_cells = 0;
#endif
	fetch(0);
	for (int r0 = 0; cv || nv; r0 += 2) {
		uint32_t *const prow = (uint32_t *)(dir + (size_t)((r0 >> 1) & (ST_ROWS / 2 - 1)) * (size_t)ncol * 4u);
		// ---- promotion: nxt becomes cur once every column of it has started and cur has been traced back; then the pair after it is
		//      fetched, to start when the lanes are through with the longer of the pair's jobs ----
		if (nv && !cv && r0 >= nR + (nt[0] > nt[1] ? nt[0] : nt[1])) {
			cv = true, cR = nR;
			int step = 0;
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				cq[h] = nq[h], ct[h] = nt[h], cj[h] = nj[h];
				step = cq[h] > step ? cq[h] : step, step = ct[h] > step ? ct[h] : step;
			}
			fetch(cR + step > r0 ? cR + step : r0);
		}
		// register sets with a valid cell on either row of the row pair (uniform): job (R, q, t) has its cells of row r in columns
		// max(0, r - R - q + 1) .. min(t - 1, r - R); both bounds only grow with r
		uint32_t need = 0;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			if (cv) {
				const int lo = r0 - cR - cq[h] + 1 > 0 ? r0 - cR - cq[h] + 1 : 0, hi = ct[h] - 1; // (cur: every column has started)
				if (lo <= hi) need |= ((2u << (hi >> 6)) - 1u) & ~((1u << (lo >> 6)) - 1u);
			}
			if (nv && r0 + 1 >= nR) {
				const int lo = r0 - nR - nq[h] + 1 > 0 ? r0 - nR - nq[h] + 1 : 0, hi = r0 + 1 - nR < nt[h] - 1 ? r0 + 1 - nR : nt[h] - 1;
				if (lo <= hi) need |= ((2u << (hi >> 6)) - 1u) & ~((1u << (lo >> 6)) - 1u);
			}
		}
#pragma unroll
		for (int par = 0; par < 2; ++par) {
			const int r = r0 + par;
			// ---- this row: where nxt's edge is (the column whose first cell of nxt lies on this anti-diagonal), which pair lane 0 is on ----
			const bool started = nv && r >= nR;
			const int edge = started ? r - nR : -1;
			const uint32_t edge_halves = (edge >= 0 && edge < nt[0] ? 0xffffu : 0u) | (edge >= 0 && edge < nt[1] ? 0xffff0000u : 0u);
			const uint32_t S_BND = pk2(border(r - (started ? nR : cR)));
			// register sets from the highest down, so that set c still sees row r-1 in set c-1 when it fetches its carry-ins
#pragma unroll
			for (int c = NC - 1; c >= 0; --c) {
				if (!(need >> c & 1u)) continue; // no valid cell on either row: whatever the set's registers hold stays dead until an edge re-enters it
#ifdef MM2AMD_GF_COUNT
				++n_setrows;
#endif
				uint32_t cV = S_BND, cX = S_NQE_X, cX2 = S_NQE2_X; // column -1: the matrix border
				if (c > 0) cV = gf_ror1(V[c - 1]), cX = gf_ror1(X[c - 1]), cX2 = gf_ror1(X2[c - 1]); // lane 0 <- lane 63 of the previous set
				const uint32_t vp = dpp_shr1u(cV, V[c]), xp = dpp_shr1u(cX, X[c]), x2p = dpp_shr1u(cX2, X2[c]);
				if (edge_halves && (edge >> 6) == c) { // the lane takes up the next pair: target bases, and u / y / y2 from the border (:156-163)
					const int t = c * 64 + lane;
					const uint32_t em = t == edge ? edge_halves : 0u;
					const uint32_t tn = (uint32_t)s_tn[wave_in_block][0][t] | (uint32_t)s_tn[wave_in_block][1][t] << 16;
					U[c] = bfi(em, S_BND, U[c]), Y[c] = bfi(em, S_NQE_Y, Y[c]), Y2[c] = bfi(em, S_NQE2_Y, Y2[c]), T[c] = bfi(em, tn, T[c]);
				}
				// query position of this column's cell = row - column, in the ring (byte offset 2 * position; both halves in one 16-bit load)
				const uint32_t qa = (((uint32_t)(2 * r - 128 * c) - lane2) & (uint32_t)(2 * ST_QRING - 2)) | qring_off;
				const uint32_t qv = __builtin_amdgcn_perm(0u, (uint32_t)*(const uint16_t *)((const uint8_t *)&s_qr[0][0] + qa), 0x0c010c00u), tv = T[c];
				uint32_t d;
				gf_cell_k(tv ^ qv, tv | qv, xp, vp, x2p, U[c], V[c], X[c], Y[c], X2[c], Y2[c], d, P_MCH, K);
				if (par == 0) DE[c] = d;
				else // (idle lanes store too: their direction bytes are never read)
					*(uint32_t *)((uint8_t *)prow + c * 256 + lane4) = __builtin_amdgcn_perm(d, DE[c], 0x06040200u); // [even A, even B, odd A, odd B]
			}
		}
		// ---- the pair's last anti-diagonal (row R + q + t - 2 of its longer job) has been stored ----
		if (cv) {
			const int rowsA = cq[0] + ct[0], rowsB = cq[1] + ct[1];
			if (r0 + 1 >= cR + (rowsA > rowsB ? rowsA : rowsB) - 2) {
				finish();
				cv = false;
#ifdef MM2AMD_GF_COUNT
				n_cells += (long long)cq[0] * ct[0] + (long long)cq[1] * ct[1];
#endif
			}
		}
	}
#ifdef MM2AMD_GF_COUNT
	if (lane == 0 && (slot == 0 || slot == 1001 || MM2AMD_GF_COUNT + 0 == 2)) printf("GFCOUNT stream<%d> slot %d: %lld register-set rows, %lld cells, lane utilisation %.3f\n", NC, slot, n_setrows, n_cells, (double)n_cells / (128.0 * (double)n_setrows));
#endif
}

void ksw_stream_launch(const KswLaunch &L, int n_slots, int n_sets, void *stream)
{
	if (L.n_jobs <= 0 && !L.n_list) return;
	const int n_blocks = (n_slots + 3) / 4;
	hipStream_t s = (hipStream_t)stream;
	if (n_sets == 4) hipLaunchKernelGGL((ksw_stream_kernel<4, ST_W4>), dim3(n_blocks), dim3(256), 0, s, L);
	else if (n_sets == 8) hipLaunchKernelGGL((ksw_stream_kernel<8, 4>), dim3(n_blocks), dim3(256), 0, s, L);
	else throw std::runtime_error("[mm2amd] ksw_stream_launch: unsupported register-set count");
	HIP_CHECK(hipGetLastError());
}

int ksw_stream_waves(int n_sets) { return n_sets == 4 ? ST_W4 : 4; } // blocks of four waves per CU the instantiation is compiled for

size_t ksw_stream_slot_bytes(int n_sets) { return (size_t)(st_rows(n_sets) / 2) * (size_t)(n_sets * 64) * 4 / 2; } // per job slot; a wave's ring is two of them

} // namespace mm2amd

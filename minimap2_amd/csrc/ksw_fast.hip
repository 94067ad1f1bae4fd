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
// tests/test_gpu_ksw.py checks this kernel against the lane-exact oracle on every preset's scoring; jobs that are not
// eligible (binding band, extension flags, exact-max mode, generic matrices, long sequences) take the exact kernel.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"

namespace mm2amd {

constexpr int FAST_QCAP = 1024; // query bytes kept in LDS per wave

__device__ __forceinline__ int dpp_shr1(int carry_in, int v) // lane i <- v[i-1], lane 0 <- carry_in
{
	return __builtin_amdgcn_update_dpp(carry_in, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

struct FastCig { uint32_t *c; int n; uint32_t last; };
__device__ __forceinline__ void fast_cig_push(FastCig &g, uint32_t op, int len) // ksw_push_cigar (ksw2.h:114-124)
{
	if (g.n == 0 || op != (g.last & 0xf)) {
		if (g.n > 0) g.c[g.n - 1] = g.last;
		g.last = (uint32_t)len << 4 | op;
		++g.n;
	} else g.last += (uint32_t)len << 4;
}

// second launch bound = waves per SIMD to compile for: 5 (<= 96 VGPRs) up to 4 register sets, 4 (<= 128) for 5-6, 3 for 8
template <int NC>
__global__ void __launch_bounds__(256, (NC <= 4 ? 5 : NC <= 6 ? 4 : 3)) ksw_fast_kernel(KswLaunch L)
{
	__shared__ uint8_t s_q[4][FAST_QCAP];
	__shared__ uint8_t s_t[4][512];
	__shared__ int8_t s_mat[32];
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	const int slot = blockIdx.x * 4 + wave_in_block;
	uint8_t *dir = L.dir_pool + (size_t)slot * L.slot_bytes;
	uint8_t *qb = s_q[wave_in_block], *tb = s_t[wave_in_block];
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

	for (;;) {
		int jid = 0;
		if (lane == 0) jid = atomicAdd(L.counter, 1);
		jid = __builtin_amdgcn_readfirstlane(jid);
		if (jid >= L.n_jobs) break;
		const KswJob J = L.jobs[jid];
		const int qlen = J.qlen, tlen = J.tlen, flag = J.flag;
		const int ncol = (tlen + 63) & ~63;
		// ---- operands: query bytes to LDS, one target base per (register set, lane) ----
		for (int i = lane; i < qlen; i += 64) qb[i] = L.qpool[(flag & KSWJ_Q_REVERSED) ? J.q_off - (uint64_t)i : J.q_off + (uint64_t)i];
		int T[NC], U[NC], V[NC], X[NC], Y[NC], X2[NC], Y2[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int t = c * 64 + lane;
			int b = 4;
			if (t < tlen) {
				const uint64_t pos = (flag & KSWJ_T_REVERSED) ? J.t_off - (uint64_t)t : J.t_off + (uint64_t)t;
				b = (flag & KSWJ_T_PACKED) ? (int)(L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (int)L.tpool[pos];
			}
			T[c] = b;
			if (t < tlen) tb[t] = (uint8_t)b;
			U[c] = V[c] = X[c] = Y[c] = nqe, X2[c] = Y2[c] = nqe2; // ksw2_extd2_sse.c:111-116
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();

		// Score of the corner cell H(tlen-1, qlen-1).  The reference's approximate-score walk (ksw2_extd2_sse.c:366-383) sums exact
		// score differences along one particular monotone path and, without KSW_EZ_APPROX_DROP, reports only its end point, which
		// is path-independent.  We sum along the matrix border instead: u of the first cell of every column along the top row
		// (H(0,0) = u - (q+e) by the border convention, :358), then v down the last column -- one scalar add per anti-diagonal.
		int H0 = -qe_in;
		const int n_rows = qlen + tlen - 1, last_set = (tlen - 1) >> 6, last_lane = (tlen - 1) & 63;
		for (int r = 0; r < n_rows; ++r) {
			const int st0 = r - qlen + 1 > 0 ? r - qlen + 1 : 0, en0 = r < tlen - 1 ? r : tlen - 1;
			// value of v[-1] / u[r] on the matrix border (ksw2_extd2_sse.c:148-163)
			const int bnd = r == 0 ? nqe : r < long_thres ? -e : r == long_thres ? long_diff : -e2;
			const bool top = r < tlen; // this anti-diagonal still starts a new column (t = r)
			const int edge_set = r >> 6, edge_lane = r & 63;
			uint8_t *pr = dir + (size_t)r * ncol;
			// register sets from the highest down, so that set c still sees row r-1 in set c-1 when it fetches its carry-ins
#pragma unroll
			for (int c = NC - 1; c >= 0; --c) {
				if (c * 64 > en0 || c * 64 + 63 < st0) continue; // register set outside the anti-diagonal (uniform)
				const int t = c * 64 + lane;
				const bool act = t >= st0 && t <= en0;
				int cV = bnd, cX = nqe, cX2 = nqe2; // column -1: the matrix border
				if (c > 0) cV = __builtin_amdgcn_readlane(V[c - 1], 63), cX = __builtin_amdgcn_readlane(X[c - 1], 63), cX2 = __builtin_amdgcn_readlane(X2[c - 1], 63);
				const int vp = dpp_shr1(cV, V[c]), xp = dpp_shr1(cX, X[c]), x2p = dpp_shr1(cX2, X2[c]);
				if (top && edge_set == c) { // u[r], y[r], y2[r] take their border values on first use (:156-163)
					const bool edge = lane == edge_lane;
					U[c] = edge ? bnd : U[c], Y[c] = edge ? nqe : Y[c], Y2[c] = edge ? nqe2 : Y2[c];
				}
				if (act) {
					const int ut = U[c], qv = qb[r - t], tv = T[c];
					int z = (tv == m - 1 || qv == m - 1) ? sc_N : tv == qv ? sc_mch : sc_mis;
					int a = xp + vp, b = Y[c] + ut, a2 = x2p + vp, b2 = Y2[c] + ut, d;
					d = a > z ? 1 : 0;   z = z > a ? z : a;    // strictly greater wins (:235-243)
					d = b > z ? 2 : d;   z = z > b ? z : b;
					d = a2 > z ? 3 : d;  z = z > a2 ? z : a2;
					d = b2 > z ? 4 : d;  z = z > b2 ? z : b2;
					z = z < sc_mch ? z : sc_mch;
					U[c] = z - vp, V[c] = z - ut;
					int tmp = z - q;  a -= tmp, b -= tmp;
					tmp = z - q2;     a2 -= tmp, b2 -= tmp;
					X[c] = (a > 0 ? a : 0) - qe;      d |= a > 0 ? 0x08 : 0;
					Y[c] = (b > 0 ? b : 0) - qe;      d |= b > 0 ? 0x10 : 0;
					X2[c] = (a2 > 0 ? a2 : 0) - qe2;  d |= a2 > 0 ? 0x20 : 0;
					Y2[c] = (b2 > 0 ? b2 : 0) - qe2;  d |= b2 > 0 ? 0x40 : 0;
					pr[t] = (uint8_t)d;
				}
				if (top) { if (edge_set == c) H0 += __builtin_amdgcn_readlane(U[c], edge_lane); }
				else if (last_set == c) H0 += __builtin_amdgcn_readlane(V[c], last_lane);
			}
		}
		// ---- traceback from (tlen-1, qlen-1) (ksw2_extd2_sse.c:389-391; ksw_backtrack with every cell inside the matrix) ----
		__threadfence_block();
		FastCig g = { L.cigar_tmp + (size_t)slot * L.cigar_tmp_cap, 0, 0u };
		uint32_t cig_off = 0;
		int32_t zd_max = 0, zd_t0 = -1, zd_t1 = -1, zd_q0 = -1, zd_q1 = -1;
		if (lane == 0) {
			int i = tlen - 1, j = qlen - 1, state = 0;
			while (i >= 0 && j >= 0) {
				const int tmp = dir[(size_t)(i + j) * ncol + i];
				if (state == 0) state = tmp & 7;
				else if (!(tmp >> (state + 2) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (state == 0) fast_cig_push(g, 0, 1), --i, --j;
				else if (state == 1 || state == 3) fast_cig_push(g, 2, 1), --i;
				else fast_cig_push(g, 1, 1), --j;
			}
			if (i >= 0) fast_cig_push(g, 2, i + 1);
			if (j >= 0) fast_cig_push(g, 1, j + 1);
			if (g.n > 0) g.c[g.n - 1] = g.last;
			if (g.n > 0) cig_off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
			// mm_test_zdrop's scan (align.c:61-84 with update_max_zdrop :46-59) over the alignment just produced, start to end
			// (g.c holds the operations in traceback order, i.e. last first)
			int32_t score = 0, mx = INT32_MIN, mx_i = -1, mx_j = -1, ci = 0, cj = 0;
			const int gq = L.sc.q, ge = L.sc.e;
			auto track = [&](int32_t sc, int pi, int pj) {
				if (sc < mx) {
					const int li = pi - mx_i, lj = pj - mx_j, diff = li > lj ? li - lj : lj - li, z = mx - sc - diff * ge;
					if (z > zd_max) zd_max = z, zd_t0 = mx_i, zd_t1 = pi, zd_q0 = mx_j, zd_q1 = pj;
				} else mx = sc, mx_i = pi, mx_j = pj;
			};
			for (int k = g.n - 1; k >= 0; --k) {
				const uint32_t op = g.c[k] & 0xf, len = g.c[k] >> 4;
				if (op == 0) {
					for (uint32_t l = 0; l < len; ++l) { score += s_mat[tb[ci + l] * 5 + qb[cj + l]]; track(score, ci + (int)l, cj + (int)l); }
					ci += len, cj += len;
				} else {
					score -= gq + ge * (int)len;
					if (op == 1) cj += len; else ci += len;
					track(score, ci, cj);
				}
			}
		}
		const int n_cig = __builtin_amdgcn_readfirstlane(g.n);
		cig_off = __builtin_amdgcn_readfirstlane(cig_off);
		__threadfence_block();
		if (n_cig > 0) {
			if ((unsigned long long)cig_off + (unsigned)n_cig > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
			else for (int k = lane; k < n_cig; k += 64) L.cigar_pool[cig_off + k] = g.c[n_cig - 1 - k]; // forward order
		}
		if (lane == 0) {
			KswRes R;
			R.max = 0, R.zdropped = 0, R.max_q = R.max_t = -1, R.mqe = R.mte = KSW_NEG_INF, R.mqe_t = R.mte_q = -1;
			R.score = H0, R.n_cigar = n_cig, R.reach_end = 0, R.cigar_off = cig_off;
			R.zd_max = zd_max, R.zd_t0 = zd_t0, R.zd_t1 = zd_t1, R.zd_q0 = zd_q0, R.zd_q1 = zd_q1;
			L.res[jid] = R;
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

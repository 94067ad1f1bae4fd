// region_finish_kernel: see region_finish.hpp.  One wavefront finishes two regions, one per half-wave, with the stitched CIGAR in LDS:
//
//   A  the windows' CIGARs are copied into LDS one after the other by the 32 lanes (a window's first operation joins an equal one before it);
//   B  mm_fix_cigar is a walk in which every step depends on the one before (a left-alignment changes the lengths the next one sees): ONE
//      lane per half does it, on LDS, fetching the few bases it compares through 8-byte windows;
//   C  mm_update_extra's walk -- s += score, clamp at zero, keep the maximum -- is a maximum-subarray problem: the CIGAR is cut into 32
//      stretches of operations, every lane walks its own (~300 columns of a 10 kb read) and keeps (sum, best suffix, best prefix, best
//      stretch); the 32 summaries combine associatively.  The walk's values are integers minus gap costs that are multiples of 2^-23
//      (mg_log2 returns a float >= 1), far below 2^29: the reference's doubles are exact, and 64-bit integers in units of 2^-23 hold the same
//      numbers, so cutting the walk changes nothing;
//   D  the CIGAR goes back to the output pool with coalesced stores.
//
// (A thread per region was tried first: 127-142 ms per 1-Gbase step -- the eight or sixty-four walks of a wave diverge at every operation.)
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "region_finish.hpp"

namespace mm2amd {

#define FIN_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

namespace {
constexpr int FIN_WIN = 1024, FIN_BACK = 128; // stage B: codes staged per round, of which this many lie behind the walking lane
// mg_log2 (mmpriv.h:139-147), float arithmetic as written there (compiled with -ffp-contract=off)
__device__ __forceinline__ float fin_log2(float x)
{
	uint32_t i = __float_as_uint(x);
	float l = (float)((int)((i >> 23) & 255) - 128);
	i &= ~(255u << 23);
	i += 127u << 23;
	const float f = __uint_as_float(i);
	l += (-0.34484843f * f + 2.02466578f) * f - 0.67487759f;
	return l;
}
struct ByteWindow { int64_t blk = -4; uint64_t w = 0; };  // the aligned 8-byte block of the query pool read last
struct CodeWindow { int64_t blk = -4; uint32_t w = 0; };  // the packed-reference word (8 bases) read last
} // namespace

__global__ void __launch_bounds__(64) region_finish_kernel(FinParams P)
{
	MM2_DYN_LDS(uint32_t, s_cg_all); // 2 x P.cap_ops: the launch's longest stitched CIGAR decides how many waves a CU holds, not the cap
	__shared__ long long s_tup[2][32][4];
	__shared__ int32_t s_cnt[2][32][8];
	__shared__ int32_t s_hdr[2][8];
	__shared__ int8_t s_mat[32];
	__shared__ __attribute__((aligned(8))) uint8_t s_qw[2][FIN_WIN], s_tw[2][FIN_WIN];
	const int lane = threadIdx.x & 63, h = lane >> 5, hl = lane & 31;
	const int id = blockIdx.x * 2 + h;
	const bool have = id < P.n_regions;
	FinRegion R;
	R.q_pos = R.t_pos = 0, R.piece0 = R.n_pieces = R.out_off = 0, R.q_len = R.t_len = 0;
	if (have) R = P.regions[id];
	uint32_t *const cg = s_cg_all + (size_t)h * (size_t)P.cap_ops;
	if (lane < 25) s_mat[lane] = P.mat[lane];
	auto qb = [&](ByteWindow &c, int32_t i) -> int {
		const uint64_t a = R.q_pos + (uint64_t)(int64_t)i;
		if ((int64_t)(a >> 3) != c.blk) c.blk = (int64_t)(a >> 3), c.w = *(const uint64_t *)(P.qpool + (a & ~7ull));
		return (int)(c.w >> ((a & 7) << 3) & 0xff);
	};
	auto tb = [&](CodeWindow &c, int32_t i) -> int {
		const uint64_t o = R.t_pos + (uint64_t)(int64_t)i;
		if ((int64_t)(o >> 3) != c.blk) c.blk = (int64_t)(o >> 3), c.w = P.S[o >> 3];
		return (int)(c.w >> ((o & 7) << 2) & 0xf);
	};

	// ---- A: the windows' CIGARs one after the other; the first operation of a window joins an equal one before it (align.c:320-334) ----
	uint32_t n = 0; // the same in every lane of the half
	{
		if (hl == 0) s_hdr[h][0] = (int32_t)R.n_pieces;
		FIN_SYNC();
		const uint32_t np_max = (uint32_t)(s_hdr[0][0] > s_hdr[1][0] ? s_hdr[0][0] : s_hdr[1][0]); // both halves take the same number of steps (they share the barriers)
		for (uint32_t pk = 0; pk < np_max; ++pk) {
			if (pk < R.n_pieces) {
				const FinPiece pc = P.pieces[R.piece0 + pk];
				const uint32_t *src = P.cigar_pool + pc.off;
				if (pc.n > 0) {
					const uint32_t first = src[0];
					const bool join = n > 0 && (cg[n - 1] & 0xf) == (first & 0xf);
					FIN_SYNC(); // (every lane has read cg[n - 1] before lane 0 changes it)
					if (join) {
						if (hl == 0) cg[n - 1] += first >> 4 << 4;
						for (uint32_t j = 1 + (uint32_t)hl; j < pc.n; j += 32) cg[n + j - 1] = src[j];
						n += pc.n - 1;
					} else {
						for (uint32_t j = (uint32_t)hl; j < pc.n; j += 32) cg[n + j] = src[j];
						n += pc.n;
					}
				} else FIN_SYNC();
			} else FIN_SYNC();
			FIN_SYNC();
		}
	}

	// ---- B: mm_fix_cigar (align.c:105-181), one lane per half.  Its first pass compares a few bases at every indel; fetched one by one they
	//      are ~1400 dependent global loads per region and were four fifths of this kernel's time.  The pass therefore runs in rounds: the 32
	//      lanes of the half copy the next FIN_WIN query and reference codes into LDS (coalesced), the walking lane goes on until an indel
	//      reaches past them.  A comparison that falls outside the window (a left-alignment longer than FIN_BACK bases) reads global memory. ----
	{
		int32_t toff = 0, qoff = 0; // the walking lane's position; the other lanes only see them through s_hdr
		bool shrink = false;
		uint32_t k = 0, prev = 0, cur = 0, next = 0; // cg[k - 1], cg[k], cg[k + 1] as this pass sees them (it rewrites the neighbours)
		ByteWindow qfar;
		CodeWindow tfar;
		if (hl == 0 && n > 1) cur = cg[0], next = cg[1];
		const bool walks = n > 1; // uniform in the half
		for (;;) {
			// where the next round starts (uniform in the wave: both halves take part in the barriers)
			if (hl == 0) s_hdr[h][4] = walks && k < n ? 1 : 0, s_hdr[h][5] = qoff, s_hdr[h][6] = toff;
			FIN_SYNC();
			if (!s_hdr[0][4] && !s_hdr[1][4]) break;
			const int32_t wq0 = s_hdr[h][5] > FIN_BACK ? s_hdr[h][5] - FIN_BACK : 0, wt0 = s_hdr[h][6] > FIN_BACK ? s_hdr[h][6] - FIN_BACK : 0;
			if (s_hdr[h][4]) { // eight consecutive codes per lane and block of 256: four wide loads, all in flight together
#pragma unroll
				for (int b = 0; b < FIN_WIN / 256; ++b) {
					const int i0 = b * 256 + hl * 8;
					uint64_t qv = 0, tv = 0;
					if (wq0 + i0 < R.q_len) {
						const uint64_t a = R.q_pos + (uint64_t)(wq0 + i0);
						const uint64_t lo = *(const uint64_t *)(P.qpool + (a & ~7ull)), hi = *(const uint64_t *)(P.qpool + (a & ~7ull) + 8);
						const int sh = (int)(a & 7) * 8;
						qv = sh ? lo >> sh | hi << (64 - sh) : lo;
					}
					if (wt0 + i0 < R.t_len) {
						const uint64_t o = R.t_pos + (uint64_t)(wt0 + i0);
						const uint64_t two = (uint64_t)P.S[o >> 3] | (uint64_t)P.S[(o >> 3) + 1] << 32;
						uint64_t x = (uint32_t)(two >> ((o & 7) << 2)); // eight 4-bit codes -> eight bytes
						x = (x | x << 16) & 0x0000FFFF0000FFFFull;
						x = (x | x << 8) & 0x00FF00FF00FF00FFull;
						tv = (x | x << 4) & 0x0F0F0F0F0F0F0F0Full;
					}
					*(uint64_t *)&s_qw[h][i0] = qv, *(uint64_t *)&s_tw[h][i0] = tv;
				}
			}
			FIN_SYNC();
			if (hl == 0 && s_hdr[h][4]) {
				auto qat = [&](int32_t i) -> int { return i >= wq0 && i < wq0 + FIN_WIN ? (int)s_qw[h][i - wq0] : qb(qfar, i); };
				auto tat = [&](int32_t i) -> int { return i >= wt0 && i < wt0 + FIN_WIN ? (int)s_tw[h][i - wt0] : tb(tfar, i); };
				for (; k < n; ++k) {
					const uint32_t op = cur & 0xf, len = cur >> 4;
					if ((op == 1 && qoff + (int32_t)len > wq0 + FIN_WIN && qoff > wq0 + FIN_BACK) || (op == 2 && toff + (int32_t)len > wt0 + FIN_WIN && toff > wt0 + FIN_BACK))
						break; // the indel reaches past the window, and a new window would start further on: next round
					if (len == 0) shrink = true;
					if (op == 0) toff += len, qoff += len;
					else if (op == 1 || op == 2) {
						if (k > 0 && k < n - 1 && (prev & 0xf) == 0 && (next & 0xf) == 0) {
							int l;
							const int prev_len = (int)(prev >> 4);
							if (op == 1) { for (l = 0; l < prev_len; ++l) if (qat(qoff - 1 - l) != qat(qoff + (int32_t)len - 1 - l)) break; }
							else { for (l = 0; l < prev_len; ++l) if (tat(toff - 1 - l) != tat(toff + (int32_t)len - 1 - l)) break; }
							if (l > 0) {
								prev -= (uint32_t)l << 4, next += (uint32_t)l << 4, qoff -= l, toff -= l;
								cg[k - 1] = prev, cg[k + 1] = next;
							}
							if (l == prev_len) shrink = true;
						}
						if (op == 1) qoff += len; else toff += len;
					} else if (op == 3) toff += len;
					prev = cur, cur = next, next = k + 2 < n ? cg[k + 2] : 0u;
				}
			}
			FIN_SYNC(); // (the window is rewritten next round)
		}
		if (hl == 0) {
			int32_t qshift = 0, tshift = 0;
			bool bad = false;
			if (n > 1) {
				bad = qoff != R.q_len || toff != R.t_len;
				for (uint32_t k2 = 0; k2 + 2 < n; ++k2) { // runs like 5I6D7I become one I and one D
					const uint32_t c0 = cg[k2];
					if ((c0 & 0xf) == 0) continue;
					if ((c0 & 0xf) + (cg[k2 + 1] & 0xf) == 3) {
						uint32_t l, sum[3] = {0, 0, 0};
						for (l = k2; l < n; ++l) {
							const uint32_t c = cg[l], op = c & 0xf;
							if (op == 1 || op == 2 || c >> 4 == 0) sum[op] += c >> 4;
							else break;
						}
						if (sum[1] > 0 && sum[2] > 0 && l - k2 > 2) {
							cg[k2] = sum[1] << 4 | 1;
							cg[k2 + 1] = sum[2] << 4 | 2;
							for (k2 += 2; k2 < l; ++k2) cg[k2] &= 0xf;
							shrink = true;
						}
						k2 = l;
					}
				}
				if (shrink) {
					uint32_t l = 0;
					for (uint32_t k2 = 0; k2 < n; ++k2) { const uint32_t c = cg[k2]; if (c >> 4 != 0) cg[l++] = c; }
					n = l;
					l = 0;
					if (n > 0) {
						uint32_t c = cg[0]; // the running operation: equal neighbours add up
						for (uint32_t k2 = 0; k2 < n; ++k2) {
							if (k2 == n - 1) { cg[l++] = c; break; }
							const uint32_t nx = cg[k2 + 1];
							if ((c & 0xf) != (nx & 0xf)) cg[l++] = c, c = nx;
							else c = nx + (c >> 4 << 4);
						}
					}
					n = l;
				}
				if (n > 0 && ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2)) { // an alignment never starts with a gap
					const int32_t l = (int32_t)(cg[0] >> 4);
					if ((cg[0] & 0xf) == 1) qshift = l; else tshift = l;
					--n;
					for (uint32_t k2 = 0; k2 < n; ++k2) cg[k2] = cg[k2 + 1];
				}
			}
			s_hdr[h][0] = (int32_t)n, s_hdr[h][1] = qshift, s_hdr[h][2] = tshift, s_hdr[h][3] = bad ? 1 : 0;
		}
	}
	FIN_SYNC();
	n = (uint32_t)s_hdr[h][0];
	const int32_t qshift = s_hdr[h][1], tshift = s_hdr[h][2];

	// ---- C: mm_update_extra (align.c:254-303).  Lane hl takes operations [k0, k1); first what its stretch consumes, so that every lane knows
	//      where its own starts ----
	const uint32_t k0 = (uint32_t)((uint64_t)n * (uint32_t)hl / 32), k1 = (uint32_t)((uint64_t)n * ((uint32_t)hl + 1) / 32);
	{
		int32_t dq = 0, dt = 0;
		for (uint32_t k = k0; k < k1; ++k) {
			const uint32_t c = cg[k], op = c & 0xf, len = c >> 4;
			if (op == 0) dq += len, dt += len;
			else if (op == 1) dq += len;
			else if (op == 2 || op == 3) dt += len;
		}
		s_cnt[h][hl][0] = dq, s_cnt[h][hl][1] = dt;
	}
	FIN_SYNC();
	int32_t qoff = qshift, toff = tshift;
	for (int l = 0; l < hl; ++l) qoff += s_cnt[h][l][0], toff += s_cnt[h][l][1];
	const int32_t q_end_all = qshift, t_end_all = tshift; // (lane 31 adds its own stretch below for the coverage check)
	FIN_SYNC(); // s_cnt is written again below
	int32_t n_ambi_all = 0, n_ambi_m = 0, n_diff = 0, len_m = 0, len_gap = 0, spliced = 0;
	long long A = 0, B = 0, Pm = 0, Q = 0; // sum; best suffix sum (>= 0); best prefix sum; best stretch sum (>= 0): units of 2^-23
	bool any_event = false;
	{
		ByteWindow qw;
		CodeWindow tw;
		long long pre_max = 0; // best prefix so far (valid once any_event)
		auto event = [&](long long d) {
			A += d;
			if (!any_event || A > pre_max) pre_max = A;
			any_event = true;
			B = B + d > 0 ? B + d : 0;
			Q = Q > B ? Q : B;
		};
		uint32_t k = k0, rem = 0;
		for (;;) { // flat: one column or one operation per step (the lanes' run lengths differ)
			if (rem == 0) {
				if (k >= k1) break;
				const uint32_t c = cg[k++], op = c & 0xf, len = c >> 4;
				if (op == 0) rem = len, len_m += (int32_t)len;
				else if (op == 1 || op == 2) {
					int n_ambi = 0;
					for (uint32_t l = 0; l < len; ++l) n_ambi += (op == 1 ? qb(qw, qoff + (int32_t)l) : tb(tw, toff + (int32_t)l)) > 3;
					len_gap += (int32_t)len, n_ambi_all += n_ambi;
					const double cost = P.log_gap ? P.q + (double)P.e * fin_log2((float)(1.0 + len)) : (double)(P.q + P.e);
					event(-(long long)(cost * 8388608.0));
					if (op == 1) qoff += len; else toff += len;
				} else if (op == 3) spliced = 1, toff += len;
				continue;
			}
			const int cq = qb(qw, qoff), ct = tb(tw, toff);
			const int amb = (ct > 3) | (cq > 3);
			n_ambi_all += amb, n_ambi_m += amb, n_diff += (ct != cq) & (amb ^ 1);
			event((long long)s_mat[ct * 5 + cq] * 8388608ll);
			++qoff, ++toff, --rem;
		}
		Pm = pre_max;
	}
	(void)q_end_all, (void)t_end_all;
	s_tup[h][hl][0] = A, s_tup[h][hl][1] = B, s_tup[h][hl][2] = any_event ? Pm : 0, s_tup[h][hl][3] = Q;
	s_cnt[h][hl][0] = n_ambi_all, s_cnt[h][hl][1] = n_ambi_m, s_cnt[h][hl][2] = n_diff, s_cnt[h][hl][3] = len_m, s_cnt[h][hl][4] = len_gap, s_cnt[h][hl][5] = spliced;
	s_cnt[h][hl][6] = any_event ? 1 : 0, s_cnt[h][hl][7] = hl == 31 ? (qoff == R.q_len && toff == R.t_len ? 0 : 1) : 0;
	FIN_SYNC();
	if (hl == 0 && have) {
		// the stretches in order: (A, B, P, Q) of "1 then 2" = (A1 + A2, max(B2, B1 + A2), max(P1, A1 + P2), max(Q1, Q2, B1 + P2)); a stretch without
		// events is the identity
		long long tA = 0, tB = 0, tQ = 0;
		int32_t c[6] = {0, 0, 0, 0, 0, 0};
		for (int l = 0; l < 32; ++l) {
			for (int j = 0; j < 5; ++j) c[j] += s_cnt[h][l][j];
			c[5] |= s_cnt[h][l][5];
			if (!s_cnt[h][l][6]) continue;
			const long long a2 = s_tup[h][l][0], b2 = s_tup[h][l][1], p2 = s_tup[h][l][2], q2 = s_tup[h][l][3];
			const long long cross = tB + p2;
			tQ = tQ > q2 ? tQ : q2;
			tQ = tQ > cross ? tQ : cross;
			tB = b2 > tB + a2 ? b2 : tB + a2;
			tA += a2;
		}
		(void)tA;
		bool bad = s_hdr[h][3] != 0;
		if (n >= 1) bad = bad || s_cnt[h][31][7] != 0; // the operations must add up to the windows
		FinResult out;
		out.n_cigar = bad ? -1 : (int32_t)n;
		out.blen = c[3] + c[4] - c[0], out.mlen = c[3] - c[1] - c[2], out.n_ambi = c[0];
		out.dp_max = (int32_t)((double)tQ / 8388608.0 + .499);
		out.qshift = qshift, out.tshift = tshift, out.is_spliced = c[5];
		P.results[id] = out;
	}
	// ---- D: the CIGAR back to the pool ----
	if (have) for (uint32_t k = (uint32_t)hl; k < n; k += 32) P.out_pool[R.out_off + k] = cg[k];
}

void region_finish_launch(const FinParams &P, void *stream)
{
	if (P.n_regions <= 0) return;
	hipLaunchKernelGGL(region_finish_kernel, dim3((P.n_regions + 1) / 2), dim3(64), (size_t)2 * (size_t)P.cap_ops * sizeof(uint32_t), (hipStream_t)stream, P);
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

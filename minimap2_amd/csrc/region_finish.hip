// region_finish_kernel: see region_finish.hpp.  One wavefront finishes one region, with the stitched CIGAR in LDS and every step spread over
// the 64 lanes (round 4; round 3: two regions per wave, mm_fix_cigar walked by ONE lane per region -- 1-2 ms of dependent LDS round trips per
// wave, 56 ms per 1-Gbase step):
//
//   A  the windows' CIGARs are copied into LDS by all lanes: a lane per window works out where the window starts (a window whose first operation
//      equals the last one before it joins it: its first length is added there), then the lanes copy window after window, coalesced;
//   B  mm_fix_cigar (align.c:105-181).  Its first pass looks sequential -- left-aligning indel k changes the length of the match before indel
//      k + 2 -- but the shift is  l_k = min(m_k, L_{k-1} + l_{k-2})  with m_k (how far the indel CAN move: a property of the sequence alone) and
//      L_{k-1} (the original length of the match before it) known up front; functions x -> min(a, b + x) are closed under composition, so the
//      l_k of a whole CIGAR are one prefix scan: every lane computes the m_k of its stretch of operations and the stretch's composed function,
//      a wave scan hands every lane its carry-in, a second walk applies it.  The I/D-cluster merge (second pass) almost never has work: lanes
//      look for a candidate, only then one lane walks.  Dropping empty operations and merging equal neighbours (third pass) are two compactions;
//   C  mm_update_extra's walk (align.c:254-303) -- s += score, clamp at zero, keep the maximum -- is a maximum-subarray problem: the CIGAR is cut
//      into 64 stretches, every lane walks its own and keeps (sum, best suffix, best prefix, best stretch); the summaries combine associatively.
//      The walk's values are integers minus gap costs that are multiples of 2^-23 (mg_log2 returns a float >= 1), far below 2^29: the
//      reference's doubles are exact, and 64-bit integers in units of 2^-23 hold the same numbers, so cutting the walk changes nothing;
//   D  the CIGAR goes back to the output pool with coalesced stores.
//
// (Rounds of measurements on the way: a thread per region 127-142 ms per 1-Gbase step -- the walks of a wave diverge at every operation --, two
// regions per wave with one lane walking each 56-100 ms.)
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "region_finish.hpp"

namespace mm2amd {

#define FIN_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

namespace {
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

constexpr int32_t FIN_INF = 1 << 30;
struct MinPlus { int32_t a, b; }; // x -> min(a, b + x) for x >= 0; a, b >= 0
__device__ __forceinline__ MinPlus mp_then(MinPlus f, MinPlus g) // first f, then g:  min(g.a, g.b + min(f.a, f.b + x))
{
	const int64_t via = (int64_t)g.b + f.a, sum = (int64_t)f.b + g.b;
	MinPlus r;
	r.a = (int64_t)g.a < via ? g.a : (int32_t)(via < FIN_INF ? via : FIN_INF);
	r.b = (int32_t)(sum < FIN_INF ? sum : FIN_INF);
	return r;
}
__device__ __forceinline__ int32_t mp_apply(MinPlus f, int32_t x) { const int64_t v = (int64_t)f.b + x; return (int64_t)f.a < v ? f.a : (int32_t)v; }
__device__ __forceinline__ uint32_t fin_scan_add(uint32_t v, int lane) // inclusive prefix sum over the wave
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(v, (unsigned)d, 64); if (lane >= d) v += t; }
	return v;
}
__device__ __forceinline__ MinPlus fin_scan_mp(MinPlus f, int lane) // inclusive: lane i gets f_0 then f_1 ... then f_i
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		MinPlus lo;
		lo.a = __shfl_up(f.a, (unsigned)d, 64), lo.b = __shfl_up(f.b, (unsigned)d, 64);
		if (lane >= d) f = mp_then(lo, f);
	}
	return f;
}
} // namespace

__global__ void __launch_bounds__(64) region_finish_kernel(FinParams P)
{
	MM2_DYN_LDS(uint32_t, s_all); // 2 x P.cap_ops words: the CIGAR, and a second array of the same size (the shifts; the compaction's target)
	__shared__ long long s_tup[64][4];
	__shared__ int32_t s_cnt[64][8];
	__shared__ uint32_t s_pc[64][4];
	__shared__ int32_t s_hdr[8];
	__shared__ int8_t s_mat[32];
	const int lane = (int)threadIdx.x;
	const int id = (int)blockIdx.x;
	const FinRegion R = P.regions[id];
	if (R.n_pieces == 0) { // nothing to stitch (an empty region; a slot region_consume_kernel handed back to the host): an empty result
		if (lane == 0) { FinResult z; __builtin_memset(&z, 0, sizeof z); P.results[id] = z; }
		return;
	}
	uint32_t *cg = s_all, *aux = s_all + (size_t)P.cap_ops;
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
	uint32_t n = 0; // the same in every lane
	{
		int32_t last_op = -1; // of the last window that had operations
		for (uint32_t p0 = 0; p0 < R.n_pieces; p0 += 64) {
			const uint32_t pi = p0 + (uint32_t)lane;
			FinPiece pc;
			pc.off = 0, pc.n = 0;
			uint32_t first = 0, lastw = 0;
			if (pi < R.n_pieces) {
				pc = P.pieces[R.piece0 + pi];
				if (pc.n == kFinLiteral) first = lastw = pc.off;
				else if (pc.n) first = P.cigar_pool[pc.off], lastw = P.cigar_pool[pc.off + pc.n - 1];
			}
			const bool literal = pc.n == kFinLiteral;
			if (literal) pc.n = 1;
			const bool nonempty = pc.n > 0;
			const unsigned long long ne = __ballot(nonempty), below = ne & ((1ull << lane) - 1ull);
			const int prev_lane = below ? 63 - __builtin_clzll(below) : -1;
			const int my_last = (int)(lastw & 0xf);
			int prev_last = __shfl(my_last, prev_lane < 0 ? 0 : prev_lane, 64);
			if (prev_lane < 0) prev_last = last_op;
			const uint32_t join = nonempty && prev_last == (int)(first & 0xf) ? 1u : 0u;
			const uint32_t contrib = nonempty ? pc.n - join : 0u;
			const uint32_t incl = fin_scan_add(contrib, lane), start = n + incl - contrib;
			s_pc[lane][0] = pc.off, s_pc[lane][1] = literal ? kFinLiteral : pc.n, s_pc[lane][2] = start, s_pc[lane][3] = join;
			FIN_SYNC();
			const uint32_t cnt = R.n_pieces - p0 < 64u ? R.n_pieces - p0 : 64u;
			for (uint32_t q = 0; q < cnt; ++q) {
				const uint32_t off = s_pc[q][0], nn = s_pc[q][1], st = s_pc[q][2], jn = s_pc[q][3];
				if (nn == kFinLiteral) { if (lane == 0 && !jn) cg[st] = off; continue; }
				const uint32_t *src = P.cigar_pool + off;
				for (uint32_t j = (uint32_t)lane + jn; j < nn; j += 64) cg[st + j - jn] = src[j];
			}
			FIN_SYNC();
			if (join) atomicAdd(&cg[start - 1], first >> 4 << 4);
			FIN_SYNC();
			n += (uint32_t)__shfl((int)incl, 63, 64);
			const int hl = ne ? 63 - __builtin_clzll(ne) : 0;
			const int hl_last = __shfl(my_last, hl, 64);
			if (ne) last_op = hl_last;
		}
	}

	// ---- B: mm_fix_cigar (align.c:105-181) ----
	int32_t qshift = 0, tshift = 0;
	bool bad = false;
	uint32_t base = 0; // the CIGAR is cg[base .. base + n) from here on (a dropped leading gap)
	if (n > 1) {
		// B1: left alignment.  Lane takes operations [k0, k1).
		const uint32_t per = (n + 63u) / 64u;
		const uint32_t k0 = (uint32_t)lane * per < n ? (uint32_t)lane * per : n, k1 = k0 + per < n ? k0 + per : n;
		uint32_t dq = 0, dt = 0;
		for (uint32_t k = k0; k < k1; ++k) {
			const uint32_t c = cg[k], op = c & 0xf, len = c >> 4;
			if (op == 0) dq += len, dt += len;
			else if (op == 1) dq += len;
			else if (op == 2 || op == 3) dt += len;
		}
		const uint32_t iq = fin_scan_add(dq, lane), it = fin_scan_add(dt, lane);
		const uint32_t tot_q = (uint32_t)__shfl((int)iq, 63, 64), tot_t = (uint32_t)__shfl((int)it, 63, 64);
		bad = (int32_t)tot_q != R.q_len || (int32_t)tot_t != R.t_len;
		auto eligible = [&](uint32_t k, uint32_t cp, uint32_t cc, uint32_t cn) { // indel k between two matches (align.c:116)
			const uint32_t op = cc & 0xf;
			return (op == 1 || op == 2) && k > 0 && k + 1 < n && (cp & 0xf) == 0 && (cn & 0xf) == 0;
		};
		const uint32_t NONE = 0xffffffffu; // (operation 15: nothing the tests below accept)
		MinPlus F;
		F.a = FIN_INF, F.b = 0;
		bool elig_before_k0 = false; // was operation k0 - 1 a left-alignable indel?
		if (k0 < k1 && k0 > 0) elig_before_k0 = eligible(k0 - 1, k0 > 1 ? cg[k0 - 2] : NONE, cg[k0 - 1], cg[k0]);
		{
			ByteWindow qw;
			CodeWindow tw;
			int32_t qoff = (int32_t)(iq - dq), toff = (int32_t)(it - dt);
			uint32_t cp = k0 > 0 && k0 < k1 ? cg[k0 - 1] : NONE, cc = k0 < k1 ? cg[k0] : NONE;
			bool ep = elig_before_k0;
			for (uint32_t k = k0; k < k1; ++k) {
				const uint32_t cn = k + 1 < n ? cg[k + 1] : NONE;
				const uint32_t op = cc & 0xf, len = cc >> 4;
				MinPlus f;
				f.a = 0, f.b = 0; // x -> 0
				const bool el = eligible(k, cp, cc, cn);
				if (el) {
					// how far the indel can move left: the stretch before it repeats with the indel's length as period.  The reference stops at the
					// length of the match before it, which never exceeds the bases consumed so far
					const int32_t cap = qoff < toff ? qoff : toff;
					int32_t m = 0;
					if (op == 1) { for (; m < cap; ++m) if (qb(qw, qoff - 1 - m) != qb(qw, qoff + (int32_t)len - 1 - m)) break; }
					else { for (; m < cap; ++m) if (tb(tw, toff - 1 - m) != tb(tw, toff + (int32_t)len - 1 - m)) break; }
					aux[k] = (uint32_t)m;
					f.a = m, f.b = (int32_t)(cp >> 4);
				} else {
					aux[k] = 0;
					if (op == 0 && ep) f.a = FIN_INF; // a match right after a left-alignable indel carries that indel's shift on to the next one
				}
				F = mp_then(F, f);
				if (op == 0) qoff += len, toff += len;
				else if (op == 1) qoff += len;
				else if (op == 2 || op == 3) toff += len;
				cp = cc, cc = cn, ep = el;
			}
		}
		{
			const MinPlus G = fin_scan_mp(F, lane); // everything up to and including this lane's stretch
			int32_t x = mp_apply(G, 0);            // the carry OUT of this lane's stretch ...
			x = __shfl_up(x, 1u, 64);              // ... is the carry INTO the next lane's
			if (lane == 0) x = 0;
			uint32_t cp = k0 > 0 && k0 < k1 ? cg[k0 - 1] : NONE, cc = k0 < k1 ? cg[k0] : NONE;
			bool ep = elig_before_k0;
			for (uint32_t k = k0; k < k1; ++k) {
				const uint32_t cn = k + 1 < n ? cg[k + 1] : NONE;
				const bool el = eligible(k, cp, cc, cn);
				if (el) {
					const int32_t m = (int32_t)aux[k], room = (int32_t)(cp >> 4) + x;
					x = m < room ? m : room;
					aux[k] = (uint32_t)x;
				} else if (!((cc & 0xf) == 0 && ep)) x = 0;
				cp = cc, cc = cn, ep = el;
			}
		}
		FIN_SYNC();
		bool zero = false; // an operation of length zero: the compaction below has work (align.c:113, :125)
		{
			// (every lane rewrites only its own operations, from the neighbours' shifts in aux: the reads of cg above are done -- barrier)
			for (uint32_t k = k0; k < k1; ++k) {
				uint32_t c = cg[k];
				if ((c & 0xf) == 0) {
					const uint32_t add = k > 0 ? aux[k - 1] : 0u, sub = k + 1 < n ? aux[k + 1] : 0u;
					c += add << 4, c -= sub << 4;
					cg[k] = c;
				}
				zero |= c >> 4 == 0;
			}
		}
		FIN_SYNC();
		// B2: runs like 5I6D7I become one I and one D (align.c:131-150).  A run starts where an I meets a D; DP windows do not produce that
		// (a mismatch is cheaper), stitched windows rarely do: look first, walk only then.
		bool cand = false;
		for (uint32_t k = k0; k < k1; ++k) {
			const uint32_t o0 = cg[k] & 0xf, o1 = k + 1 < n ? cg[k + 1] & 0xf : 15u;
			cand |= k + 2 < n && (o0 == 1 || o0 == 2) && o0 + o1 == 3;
		}
		bool shrink = __ballot(zero) != 0ull;
		const bool cand_any = __ballot(cand) != 0ull;
		if (cand_any) {
			if (lane == 0) {
				bool sh = false;
				for (uint32_t k2 = 0; k2 + 2 < n; ++k2) {
					const uint32_t c0 = cg[k2];
					if ((c0 & 0xf) == 0) continue;
					if ((c0 & 0xf) + (cg[k2 + 1] & 0xf) == 3) {
						uint32_t l, sum[4] = {0, 0, 0, 0};
						for (l = k2; l < n; ++l) {
							const uint32_t c = cg[l], op = c & 0xf;
							if (op == 1 || op == 2 || c >> 4 == 0) sum[op & 3] += c >> 4;
							else break;
						}
						if (sum[1] > 0 && sum[2] > 0 && l - k2 > 2) {
							cg[k2] = sum[1] << 4 | 1;
							cg[k2 + 1] = sum[2] << 4 | 2;
							for (k2 += 2; k2 < l; ++k2) cg[k2] &= 0xf;
							sh = true;
						}
						k2 = l;
					}
				}
				s_hdr[0] = sh ? 1 : 0;
			}
			FIN_SYNC();
			shrink = shrink || s_hdr[0] != 0;
		}
		// B3: empty operations go, equal neighbours add up (align.c:151-161): two compactions, cg -> aux -> cg
		if (shrink) {
			uint32_t keep = 0;
			for (uint32_t k = k0; k < k1; ++k) keep += cg[k] >> 4 != 0;
			const uint32_t ik = fin_scan_add(keep, lane);
			const uint32_t n1 = (uint32_t)__shfl((int)ik, 63, 64);
			uint32_t w = ik - keep;
			for (uint32_t k = k0; k < k1; ++k) { const uint32_t c = cg[k]; if (c >> 4 != 0) aux[w++] = c; }
			FIN_SYNC();
			const uint32_t per1 = (n1 + 63u) / 64u;
			const uint32_t j0 = (uint32_t)lane * per1 < n1 ? (uint32_t)lane * per1 : n1, j1 = j0 + per1 < n1 ? j0 + per1 : n1;
			uint32_t ends = 0;
			for (uint32_t k = j0; k < j1; ++k) ends += k + 1 == n1 || (aux[k] & 0xf) != (aux[k + 1] & 0xf);
			const uint32_t ie = fin_scan_add(ends, lane);
			const uint32_t n2 = (uint32_t)__shfl((int)ie, 63, 64);
			w = ie - ends;
			for (uint32_t k = j0; k < j1; ++k) {
				const uint32_t c = aux[k];
				if (k + 1 == n1 || (c & 0xf) != (aux[k + 1] & 0xf)) { // the last of a run of equal operations carries the run's sum
					uint32_t len = c >> 4;
					for (uint32_t j = k; j > 0 && (aux[j - 1] & 0xf) == (c & 0xf); --j) len += aux[j - 1] >> 4;
					cg[w++] = len << 4 | (c & 0xf);
				}
			}
			FIN_SYNC();
			n = n2;
		}
#ifdef MM2AMD_WAVE_EMU
		if (lane == 0 && getenv("MM2AMD_FIN_TRACE")) fprintf(stderr, "[mm2amd] FINPATH region %d: n %u shrink %d cand %d lead %d\n", id, n, (int)shrink, (int)cand_any, (int)(n > 0 && ((cg[0] & 0xf) == 1 || (cg[0] & 0xf) == 2)));
#endif
		// B4: an alignment never starts with a gap (align.c:162-180)
		if (n > 0) {
			const uint32_t c = cg[0];
			if ((c & 0xf) == 1 || (c & 0xf) == 2) {
				if ((c & 0xf) == 1) qshift = (int32_t)(c >> 4); else tshift = (int32_t)(c >> 4);
				base = 1, --n;
			}
		}
	}
	const uint32_t *const cf = cg + base;

	// ---- C: mm_update_extra (align.c:254-303).  Lane takes operations [k0, k1); first what its stretch consumes, so that every lane knows
	//      where its own starts ----
	const uint32_t perc = (n + 63u) / 64u;
	const uint32_t k0 = (uint32_t)lane * perc < n ? (uint32_t)lane * perc : n, k1 = k0 + perc < n ? k0 + perc : n;
	int32_t qoff, toff;
	{
		uint32_t dq = 0, dt = 0;
		for (uint32_t k = k0; k < k1; ++k) {
			const uint32_t c = cf[k], op = c & 0xf, len = c >> 4;
			if (op == 0) dq += len, dt += len;
			else if (op == 1) dq += len;
			else if (op == 2 || op == 3) dt += len;
		}
		const uint32_t iq = fin_scan_add(dq, lane), it = fin_scan_add(dt, lane);
		qoff = qshift + (int32_t)(iq - dq), toff = tshift + (int32_t)(it - dt);
		const int32_t q_end = qshift + __shfl((int)iq, 63, 64), t_end = tshift + __shfl((int)it, 63, 64);
		if (n >= 1) bad = bad || q_end != R.q_len || t_end != R.t_len; // the operations must add up to the windows
	}
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
				const uint32_t c = cf[k++], op = c & 0xf, len = c >> 4;
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
	s_tup[lane][0] = A, s_tup[lane][1] = B, s_tup[lane][2] = any_event ? Pm : 0, s_tup[lane][3] = Q;
	s_cnt[lane][0] = n_ambi_all, s_cnt[lane][1] = n_ambi_m, s_cnt[lane][2] = n_diff, s_cnt[lane][3] = len_m, s_cnt[lane][4] = len_gap, s_cnt[lane][5] = spliced;
	s_cnt[lane][6] = any_event ? 1 : 0;
	FIN_SYNC();
	if (lane == 0) {
		// the stretches in order: (A, B, P, Q) of "1 then 2" = (A1 + A2, max(B2, B1 + A2), max(P1, A1 + P2), max(Q1, Q2, B1 + P2)); a stretch without
		// events is the identity
		long long tB = 0, tQ = 0;
		int32_t c[6] = {0, 0, 0, 0, 0, 0};
		for (int l = 0; l < 64; ++l) {
			for (int j = 0; j < 5; ++j) c[j] += s_cnt[l][j];
			c[5] |= s_cnt[l][5];
			if (!s_cnt[l][6]) continue;
			const long long a2 = s_tup[l][0], b2 = s_tup[l][1], p2 = s_tup[l][2], q2 = s_tup[l][3];
			const long long cross = tB + p2;
			tQ = tQ > q2 ? tQ : q2;
			tQ = tQ > cross ? tQ : cross;
			tB = b2 > tB + a2 ? b2 : tB + a2;
		}
		FinResult out;
		out.n_cigar = bad ? -1 : (int32_t)n;
		out.blen = c[3] + c[4] - c[0], out.mlen = c[3] - c[1] - c[2], out.n_ambi = c[0];
		out.dp_max = (int32_t)((double)tQ / 8388608.0 + .499);
		out.qshift = qshift, out.tshift = tshift, out.is_spliced = c[5];
		P.results[id] = out;
	}
	// ---- D: the CIGAR back to the pool ----
	for (uint32_t k = (uint32_t)lane; k < n; k += 64) P.out_pool[R.out_off + k] = cf[k];
}

void region_finish_launch(const FinParams &P, void *stream)
{
	if (P.n_regions <= 0) return;
	hipLaunchKernelGGL(region_finish_kernel, dim3(P.n_regions), dim3(64), (size_t)2 * (size_t)P.cap_ops * sizeof(uint32_t), (hipStream_t)stream, P);
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

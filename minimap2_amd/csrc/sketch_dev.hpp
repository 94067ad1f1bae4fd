// The (w,k)-minimizer window automaton of mm_sketch (sketch.c:77-143) as a device routine that can start in the
// middle of a sequence.
//
// The reference walks a sequence once, carrying (k-mer registers, ring buffer of the last w slots, current minimum, run
// length l).  Its state after any stretch of w+k consecutive slot-consuming bases is a function of that stretch alone: the
// ring buffer then holds only slots from the stretch (their age order, not their ring position, is what the scans use), the
// tracked minimum is always the right-most minimum of the last w slots, and l is only ever compared with k, w+k-1 and w+k.
// So a lane that owns positions [cs,ce) starts `warm` bases earlier with exact k-mer registers, runs the unmodified automaton,
// reports only minimizers whose position it owns, and restarts further back in the rare case the warm-up did not reach a
// synchronised state before cs (N runs, strand-symmetric k-mers).  The union over chunks is the reference's output, in order.
//
// HPC (homopolymer-compressed k-mers, sketch.c:95-105): a run of equal bases is one symbol whose position is the run's last
// base; the k-mer's span is the total length of its k runs.  The same argument holds on the run-compressed sequence: a chunk
// starts its warm-up on a run boundary, takes its k-mer registers from the k runs before it, owns the runs that END inside it,
// and walks a run that straddles its end to completion like the reference does.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mm2amd {

template <typename KT>
__device__ __forceinline__ KT mm_hash_t(KT key, KT mask) // hash64, sketch.c:28-38.  KT = uint32_t when the mask has at most 32 bits: every step only
{                                                        // needs the masked low bits of its operands, so the narrower arithmetic gives the same value
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}
__device__ __forceinline__ uint64_t mm_hash64(uint64_t key, uint64_t mask) { return mm_hash_t<uint64_t>(key, mask); }

template <bool K32> struct SketchKmerType { typedef uint64_t type; };
template <> struct SketchKmerType<true> { typedef uint32_t type; };

// Runs the automaton for the chunk [cs,ce) of a sequence of len bases and calls emit(x, y) for every minimizer the reference would
// report at a position in [cs,ce), in the reference's order.  x = hash<<8|span, y = rid<<32 | pos<<1 | strand.
//   base_at(i)  the nt4 code (4 = ambiguous) of base i, 0 <= i < len
//   bx / by     caller-provided ring storage of at least w entries, accessed as bx[slot*stride]; YT = uint32_t keeps only the low half of y
//               in the ring (rid == 0 and len < 2^31: reads)
//   K32         2k <= 32: k-mer registers and hash in 32-bit arithmetic
//   warm0       bases of warm-up before cs on the first attempt (at least w + k; more = fewer restarts where symmetric k-mers or Ns
//               delay the synchronised state)
//   W           the window size when it is known at compile time (0: use w): the ring scans then unroll into W loads issued together and a
//               compare chain over registers -- with a runtime trip count every slot costs a dependent LDS round trip, and since some lane of
//               a wavefront loses its minimum at almost every base, the whole wave pays for that scan at almost every base
template <bool HPC, bool K32, typename YT, int W = 0, typename BaseAt, typename Emit>
__device__ __forceinline__ void sketch_chunk_core(BaseAt base_at, int64_t len, int64_t cs, int64_t ce, int w, int k, uint32_t rid,
                                                  uint64_t *bx, YT *by, int stride, Emit emit, int64_t warm0)
{
	typedef typename SketchKmerType<K32>::type KT;
	const int shift1 = 2 * (k - 1);
	const KT mask = (KT)((k < 32 ? 1ULL << 2 * k : 0ULL) - 1ULL);
	int64_t warm = warm0;
	for (;;) {
		int64_t ws = cs - warm;
		if (ws < 0) ws = 0;
		if (HPC) while (ws > 0 && base_at(ws) < 4 && base_at(ws) == base_at(ws - 1)) --ws; // start on a run boundary
		// exact k-mer registers at ws: the last k unambiguous bases before it (ambiguous bases do not shift them, sketch.c:96-116)
		KT kmer0 = 0, kmer1 = 0;
		if (ws > 0) {
			int got = 0;
			uint64_t packed = 0; // digit d = the valid base at distance d+1 before ws
			for (int64_t j = ws - 1; j >= 0 && got < k; --j) {
				const int c = base_at(j);
				if (c >= 4) continue;
				packed |= (uint64_t)c << (2 * got), ++got;
				if (HPC) while (j - 1 >= 0 && base_at(j - 1) == c) --j; // the whole run is one symbol
			}
			for (int d = got - 1; d >= 0; --d) { // replay the reference's shifts, oldest base first
				const KT c = (KT)(packed >> (2 * d) & 3ULL);
				kmer0 = (kmer0 << 2 | c) & mask;
				kmer1 = (kmer1 >> 2) | ((KT)3 ^ c) << shift1;
			}
		}
		uint64_t min_x = UINT64_MAX, min_y = UINT64_MAX;
		int l = 0, buf_pos = 0, min_pos = 0;
		uint16_t tq[32];                       // HPC: lengths of the last <= k runs (tiny_queue_t, sketch.c:42-60), and their sum
		int tq_front = 0, tq_count = 0, hpc_span = 0;
		bool synced = ws == 0; // at the sequence start the automaton is in its true initial state
		bool restart = false;
		for (int j = 0; j < w; ++j) bx[j * stride] = UINT64_MAX, by[j * stride] = (YT)UINT64_MAX;
#define MM2_EMIT(X, Y) do { const int64_t pp_ = (int64_t)((uint32_t)(Y) >> 1); if (pp_ >= cs && pp_ < ce) emit((X), (Y)); } while (0)
		for (int64_t i = ws; i < len; ++i) {
			if (i >= cs && !synced) { restart = true; break; } // not enough clean history: start further back
			// everything owned has been emitted: the window holds no valid k-mer, or its minimum lies past the chunk and the
			// first-full-window rule (which may still emit an older equal-hash slot) can no longer fire on owned slots
			if (i >= ce && (min_x == UINT64_MAX || ((int64_t)((uint32_t)min_y >> 1) >= ce && l >= w + k - 1))) break;
			const int c = base_at(i);
			uint64_t ix = UINT64_MAX, iy = UINT64_MAX;
			if (c < 4) {
				int kmer_span = l + 1 < k ? l + 1 : k;
				if (HPC) {
					int skip_len = 1;
					if (i + 1 < len && base_at(i + 1) == c) {
						for (skip_len = 2; i + skip_len < len; ++skip_len) if (base_at(i + skip_len) != c) break;
						i += skip_len - 1; // i = the last base of the run
					}
					const int sl = skip_len < 256 ? skip_len : 256; // only "span < 256" is ever asked of the sum
					tq[(tq_count++ + tq_front) & 0x1f] = (uint16_t)sl, hpc_span += sl;
					if (tq_count > k) hpc_span -= tq[tq_front], tq_front = (tq_front + 1) & 0x1f, --tq_count;
					kmer_span = hpc_span;
				}
				kmer0 = (kmer0 << 2 | (KT)c) & mask;
				kmer1 = (kmer1 >> 2) | ((KT)3 ^ (KT)c) << shift1;
				if (kmer0 == kmer1) continue; // strand-symmetric k-mer: no slot is consumed (sketch.c:108)
				const int z = kmer0 < kmer1 ? 0 : 1;
				++l;
				if (l >= k && (!HPC || kmer_span < 256)) {
					ix = (uint64_t)mm_hash_t<KT>(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)kmer_span;
					iy = (uint64_t)rid << 32 | (uint64_t)(uint32_t)i << 1 | (uint64_t)z;
				}
				if (l >= w + k) synced = true; // state now depends only on the last w+k slots
			} else l = 0, tq_front = tq_count = 0, hpc_span = 0;
			bx[buf_pos * stride] = ix, by[buf_pos * stride] = (YT)iy;
			if (W > 0 && l == w + k - 1 && min_x != UINT64_MAX) { // first full window (:117-122), oldest slot first; the newest (buf_pos) is not looked at
#pragma unroll
				for (int a = 0; a < (W > 0 ? W - 1 : 0); ++a) {
					int j = buf_pos + 1 + a;
					if (j >= W) j -= W;
					const uint64_t xj = bx[j * stride];
					const YT yj = by[j * stride];
					if (min_x == xj && yj != (YT)min_y) MM2_EMIT(xj, (uint64_t)yj);
				}
			} else if (W == 0 && l == w + k - 1 && min_x != UINT64_MAX) {
				for (int j = buf_pos + 1; j < w; ++j) if (min_x == bx[j * stride] && by[j * stride] != (YT)min_y) MM2_EMIT(bx[j * stride], (uint64_t)by[j * stride]);
				for (int j = 0; j < buf_pos; ++j)     if (min_x == bx[j * stride] && by[j * stride] != (YT)min_y) MM2_EMIT(bx[j * stride], (uint64_t)by[j * stride]);
			}
			if (ix <= min_x) {
				if (l >= w + k && min_x != UINT64_MAX) MM2_EMIT(min_x, min_y);
				min_x = ix, min_y = iy, min_pos = buf_pos;
			} else if (buf_pos == min_pos) {
				if (l >= w + k - 1 && min_x != UINT64_MAX) MM2_EMIT(min_x, min_y);
				min_x = UINT64_MAX;
				if (W > 0) { // the ring in age order (oldest first, the slot just written last) into registers, then the same two scans
					uint64_t xs[W > 0 ? W : 1];
					YT ys[W > 0 ? W : 1];
#pragma unroll
					for (int a = 0; a < W; ++a) {
						int j = buf_pos + 1 + a;
						if (j >= W) j -= W;
						xs[a] = bx[j * stride], ys[a] = by[j * stride];
					}
					int min_a = 0, n_eq = 0;
#pragma unroll
					for (int a = 0; a < W; ++a) {
						n_eq = min_x == xs[a] ? n_eq + 1 : min_x > xs[a] ? 1 : n_eq;
						if (min_x >= xs[a]) min_x = xs[a], min_y = (uint64_t)ys[a], min_a = a;
					}
					min_pos = buf_pos + 1 + min_a;
					if (min_pos >= W) min_pos -= W;
					if (n_eq > 1 && l >= w + k - 1 && min_x != UINT64_MAX) { // other slots with the minimum's hash (rare)
#pragma unroll
						for (int a = 0; a < W; ++a) if (min_x == xs[a] && (YT)min_y != ys[a]) MM2_EMIT(xs[a], (uint64_t)ys[a]);
					}
				} else {
				for (int j = buf_pos + 1; j < w; ++j) if (min_x >= bx[j * stride]) min_x = bx[j * stride], min_y = (uint64_t)by[j * stride], min_pos = j;
				for (int j = 0; j <= buf_pos; ++j)    if (min_x >= bx[j * stride]) min_x = bx[j * stride], min_y = (uint64_t)by[j * stride], min_pos = j;
				if (l >= w + k - 1 && min_x != UINT64_MAX) {
					for (int j = buf_pos + 1; j < w; ++j) if (min_x == bx[j * stride] && (YT)min_y != by[j * stride]) MM2_EMIT(bx[j * stride], (uint64_t)by[j * stride]);
					for (int j = 0; j <= buf_pos; ++j)    if (min_x == bx[j * stride] && (YT)min_y != by[j * stride]) MM2_EMIT(bx[j * stride], (uint64_t)by[j * stride]);
				}
				}
			}
			if (++buf_pos == (W > 0 ? W : w)) buf_pos = 0;
		}
		if (restart) { warm = warm * 4 + 1024; continue; }
		if (min_x != UINT64_MAX) MM2_EMIT(min_x, min_y); // end of the sequence, or an early exit with nothing owned pending
#undef MM2_EMIT
		return;
	}
}

// The same over nt4 bytes in memory (index build, the long-read fall-back, host tests): bases are fetched eight at a time (one aligned
// 64-bit load per eight positions instead of a byte load per position: lanes walk different chunks, so every byte load is its own memory
// transaction).  The buffers this runs on are 256-byte aligned and padded, which makes the aligned-down / aligned-up accesses safe.
template <bool HPC, typename Emit>
__device__ __forceinline__ void sketch_chunk(const uint8_t *seq, int64_t len, int64_t cs, int64_t ce, int w, int k, uint32_t rid,
                                             uint64_t *bx, uint64_t *by, int stride, Emit emit)
{
	uint64_t wbuf = 0;
	uintptr_t wcur = ~(uintptr_t)0;
	auto base_at = [&](int64_t i) -> int {
		const uintptr_t adr = (uintptr_t)(seq + i), al = adr & ~(uintptr_t)7;
		if (al != wcur) { wbuf = *(const uint64_t *)al; wcur = al; }
		return (int)(wbuf >> ((adr & 7) << 3) & 0xff);
	};
	sketch_chunk_core<HPC, false, uint64_t>(base_at, len, cs, ce, w, k, rid, bx, by, stride, emit, (int64_t)(2 * (w + k) + 32));
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 2-bit packed bases (sketch_wave_kernel stages a read in LDS this way): base r of a stretch sits in dword r >> 4 at bits
// 30 - 2 * (r & 15) -- big-endian within the stream, so that the k bases ending at r, read as one number, ARE the forward k-mer register
// (kmer0 = (kmer0 << 2 | c) & mask: oldest base in the highest digit).  Ambiguous bases are packed as 0 and flagged in a bit mask.
// ---------------------------------------------------------------------------------------------------------------------------------
// 16 nt4 codes (0..4), four per dword in memory order -> (packed dword, 16 flag bits: bit m = base m is ambiguous)
__device__ __forceinline__ void sk_pack16(const uint32_t q[4], uint32_t *packed, uint32_t *amb)
{
	uint32_t pk = 0, nb = 0;
#pragma unroll
	for (int d = 0; d < 4; ++d) {
		// the four 2-bit digits of a dword gathered into its top byte, first byte highest: the partial products land in disjoint 2-bit fields
		pk |= ((q[d] & 0x03030303u) * 0x40100401u >> 24) << (24 - 8 * d);
		nb |= ((q[d] >> 2 & 0x01010101u) * 0x01020408u >> 24 & 0xfu) << (4 * d);
	}
	*packed = pk, *amb = nb;
}
__device__ __forceinline__ int sk_base_at(const uint32_t *pk, const uint16_t *amb, int64_t r) // nt4 code of base r
{
	return (amb[r >> 4] >> (r & 15) & 1) ? 4 : (int)(pk[r >> 4] >> (30 - 2 * (int)(r & 15)) & 3u);
}
// forward k-mer register after base r (the k bases ending at r; pk[(r >> 4) - 2] must be readable)
__device__ __forceinline__ uint64_t sk_kmer_at(const uint32_t *pk, int64_t r, int k)
{
	const int64_t j = r >> 4;
	const int sh = 30 - 2 * (int)(r & 15);
	const uint64_t hi = (uint64_t)pk[j - 1] << 32 | pk[j], hi2 = (uint64_t)pk[j - 2] << 32 | pk[j - 1];
	const uint64_t v = (uint64_t)(uint32_t)(hi >> sh) | (uint64_t)(uint32_t)(hi2 >> sh) << 32;
	return v & ((k < 32 ? 1ULL << 2 * k : 0ULL) - 1ULL);
}
// the reverse-strand register of the same k bases: kmer1 = (kmer1 >> 2) | (3 ^ c) << 2(k-1), i.e. the digits of kmer0 reversed and complemented
__device__ __forceinline__ uint64_t sk_revcomp(uint64_t kmer0, int k)
{
#if defined(__has_builtin) && __has_builtin(__builtin_bitreverse64)
	uint64_t y = __builtin_bitreverse64(kmer0); // (v_bfrev_b32 twice) digits reversed, the two bits of a digit swapped
	y = (y >> 1 & 0x5555555555555555ULL) | (y & 0x5555555555555555ULL) << 1;
#else // host builds with a compiler that lacks the builtin: digits, nibbles, bytes
	uint64_t y = (kmer0 >> 2 & 0x3333333333333333ULL) | (kmer0 & 0x3333333333333333ULL) << 2;
	y = (y >> 4 & 0x0f0f0f0f0f0f0f0fULL) | (y & 0x0f0f0f0f0f0f0f0fULL) << 4;
	y = __builtin_bswap64(y);
#endif
	return (y >> (64 - 2 * k)) ^ ((k < 32 ? 1ULL << 2 * k : 0ULL) - 1ULL);
}
// the minimizer record of position p (rid 0, no HPC): what the automaton held when it put p's k-mer into the ring (sketch.c:108-113)
__device__ __forceinline__ void sk_minimizer_at(const uint32_t *pk, int64_t r, int64_t pos, int k, uint64_t *x, uint64_t *y)
{
	const uint64_t mask = (k < 32 ? 1ULL << 2 * k : 0ULL) - 1ULL;
	const uint64_t k0 = sk_kmer_at(pk, r, k), k1 = sk_revcomp(k0, k);
	const int z = k0 < k1 ? 0 : 1;
	const uint64_t h = 2 * k <= 32 ? (uint64_t)mm_hash_t<uint32_t>((uint32_t)(z ? k1 : k0), (uint32_t)mask) : mm_hash_t<uint64_t>(z ? k1 : k0, mask);
	*x = h << 8 | (uint64_t)k;
	*y = (uint64_t)(uint32_t)pos << 1 | (uint64_t)z;
}

} // namespace mm2amd

// Seeding and chaining kernels for gfx950: the device counterparts of mm_sketch (sketch.c:77-143),
// mm_seed_mz_flt / mm_collect_matches (seed.c:5-132), collect_seed_hits (map.c:168-204) and the fill loop of
// mg_lchain_dp (lchain.c:169-207).
//
// These stages are integer, branchy and latency-bound (random index probes, sequential state machines), not
// bandwidth- or FLOP-bound, so the mapping favours many independent reads in flight over intra-read tricks:
//   * sketch        : one lane per read runs the window state machine; the ring buffer lives in private memory.
//   * seed collect  : one wavefront per read; lanes stride over the read's minimizers for the index probes
//                     (2-3 dependent 32-64 B sector reads each), ballots/prefix sums do the order-preserving compaction.
//   * anchor sort   : one lane per read replays the reference's unstable in-place radix sort (exact_rsort.hpp).
//   * chain fill    : one wavefront per read; 64 predecessors are scored per step, then the reference's sequential
//                     rules (running maximum, skip counter, early exit) are resolved with ballots and a prefix max.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "seed_chain_dev.hpp"
#include "index_build.hpp"
#include "heap_order.hpp"
#include "sdust_core.hpp"
#include "sketch_dev.hpp"
#include "kernel_prof.hpp"

namespace mm2amd {

#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

__constant__ uint8_t c_nt4[256];
extern const uint8_t kNt4Table[256];
static bool g_nt4_uploaded = false;

static void upload_tables(hipStream_t s)
{
	if (g_nt4_uploaded) return;
	HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_nt4), kNt4Table, 256, 0, hipMemcpyHostToDevice, s));
	g_nt4_uploaded = true;
}

// ---------------------------------------------------------------------------------------------------------
// ASCII -> nt4 forward and reverse complement (align.c:1056-1061)
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) encode_kernel(SeedChainBuffers B)
{
	// B.ascii is the batch's pinned HOST buffer (read over PCIe, once): 16 bytes per lane and load, 1 KiB per wavefront request
	const int r = blockIdx.x;
	const uint64_t o = B.seq_off[r];
	const int64_t len = (int64_t)(B.seq_off[r + 1] - o);
	const char *src = B.ascii + o;
	uint8_t *f = B.qpool + 2 * o;
	const int64_t mis = (int64_t)((uintptr_t)src & 15u); // the read starts this far into its first aligned 16-byte word
	const uint4 *w = (const uint4 *)(src - mis);         // (the buffer is page-aligned and padded: whole words are readable)
	const int64_t n_words = (mis + len + 15) >> 4;
	for (int64_t c = threadIdx.x; c < n_words; c += blockDim.x) {
		const uint4 v = w[c];
		const uint32_t q[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
		for (int b = 0; b < 16; ++b) {
			const int64_t j = (c << 4) + b - mis;
			if (j >= 0 && j < len) {
				const uint8_t code = c_nt4[q[b >> 2] >> ((b & 3) * 8) & 0xffu];
				f[j] = code;
				f[2 * len - 1 - j] = code < 4 ? 3 - code : 4;
			}
		}
	}
}

void launch_encode(const SeedChainBuffers &B, void *stream)
{
	upload_tables((hipStream_t)stream);
	hipLaunchKernelGGL(encode_kernel, dim3(B.n_reads), dim3(256), 0, (hipStream_t)stream, B);
	HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// (w,k)-minimizers (sketch.c:77-143).  Minimizers of read r are written from slot seq_off[r] of the minimizer arrays (a read
// of L bases has at most L minimizers: every slot is reported at most once), so no count/scan pass is needed.
//   * sketch_wave_kernel : one wavefront per read, each lane owns len/64 consecutive positions (sketch_dev.hpp)
//   * sketch_kernel      : one lane per read, the plain sequential automaton (HPC minimizers, w > 32)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t key, uint64_t mask) { return mm_hash64(key, mask); }

template <int WMAX>
__global__ void __launch_bounds__(64) sketch_kernel(SeedChainBuffers B, int w, int k, int is_hpc)
{
	const int r = blockIdx.x * blockDim.x + threadIdx.x;
	if (r >= B.n_reads) return;
	const uint64_t o = B.seq_off[r];
	const int len = (int)(B.seq_off[r + 1] - o);
	const uint8_t *seq = B.qpool + 2 * o; // forward nt4 codes
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
	uint64_t kmer0 = 0, kmer1 = 0, min_x = UINT64_MAX, min_y = UINT64_MAX;
	uint64_t bx[WMAX], by[WMAX];
	int hq[32], hq_front = 0, hq_count = 0;
	int l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0;
	uint32_t n_out = 0;
	uint64_t *ox = B.mz_x + B.mz_off[r], *oy = B.mz_y + B.mz_off[r];
#define EMIT_MZ(X, Y) do { ox[n_out] = (X); oy[n_out] = (Y); ++n_out; } while (0)
	for (int j = 0; j < w; ++j) bx[j] = by[j] = UINT64_MAX;
	for (int i = 0; i < len; ++i) {
		const int c = seq[i];
		uint64_t ix = UINT64_MAX, iy = UINT64_MAX;
		if (c < 4) {
			if (is_hpc) {
				int run = 1;
				if (i + 1 < len && seq[i + 1] == c) {
					for (run = 2; i + run < len; ++run) if (seq[i + run] != c) break;
					i += run - 1;
				}
				hq[(hq_count++ + hq_front) & 0x1f] = run;
				kmer_span += run;
				if (hq_count > k) { kmer_span -= hq[hq_front++]; hq_front &= 0x1f; --hq_count; }
			} else kmer_span = l + 1 < k ? l + 1 : k;
			kmer0 = (kmer0 << 2 | (uint64_t)c) & mask;
			kmer1 = (kmer1 >> 2) | (3ULL ^ (uint64_t)c) << shift1;
			if (kmer0 == kmer1) continue; // strand-symmetric k-mer: no slot is consumed (sketch.c:108)
			const int z = kmer0 < kmer1 ? 0 : 1;
			++l;
			if (l >= k && kmer_span < 256) {
				ix = mix64(z ? kmer1 : kmer0, mask) << 8 | (uint64_t)kmer_span;
				iy = (uint64_t)(uint32_t)i << 1 | (uint64_t)z; // rid is 0 for reads
			}
		} else l = 0, hq_count = hq_front = 0, kmer_span = 0;
		bx[buf_pos] = ix, by[buf_pos] = iy;
		if (l == w + k - 1 && min_x != UINT64_MAX) { // first full window (:117-122)
			for (int j = buf_pos + 1; j < w; ++j) if (min_x == bx[j] && by[j] != min_y) EMIT_MZ(bx[j], by[j]);
			for (int j = 0; j < buf_pos; ++j)     if (min_x == bx[j] && by[j] != min_y) EMIT_MZ(bx[j], by[j]);
		}
		if (ix <= min_x) {
			if (l >= w + k && min_x != UINT64_MAX) EMIT_MZ(min_x, min_y);
			min_x = ix, min_y = iy, min_pos = buf_pos;
		} else if (buf_pos == min_pos) {
			if (l >= w + k - 1 && min_x != UINT64_MAX) EMIT_MZ(min_x, min_y);
			min_x = UINT64_MAX;
			for (int j = buf_pos + 1; j < w; ++j) if (min_x >= bx[j]) min_x = bx[j], min_y = by[j], min_pos = j;
			for (int j = 0; j <= buf_pos; ++j)    if (min_x >= bx[j]) min_x = bx[j], min_y = by[j], min_pos = j;
			if (l >= w + k - 1 && min_x != UINT64_MAX) {
				for (int j = buf_pos + 1; j < w; ++j) if (min_x == bx[j] && min_y != by[j]) EMIT_MZ(bx[j], by[j]);
				for (int j = 0; j <= buf_pos; ++j)    if (min_x == bx[j] && min_y != by[j]) EMIT_MZ(bx[j], by[j]);
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (min_x != UINT64_MAX) EMIT_MZ(min_x, min_y);
#undef EMIT_MZ
	B.mz_cnt[r] = n_out;
}

// One wavefront per read, everything between the read's bases and its minimizer list in LDS (round 4; the round-2 kernel let every lane
// stream its own stretch of bases from HBM and stage its minimizers through scattered 4-byte stores: with 4096 waves x 64 lanes each
// holding lines of their own the L2 cannot merge anything, and the kernel moved 16x its algorithmic bytes):
//   load   the tile's bases (a whole read up to tile_cap bases; longer reads in equal tiles) with coalesced 16-byte loads, packed to 2 bits
//          per base + an ambiguity mask (sketch_dev.hpp: sk_pack16)
//   sketch every lane runs the window automaton over its stretch of the tile (sketch_chunk_core: warm-up of w + k + 8 bases before it, ring
//          of the last w slots lane-interleaved in LDS) and marks the positions it reports in a bit mask: the reference reports a position
//          at most once and in increasing order (tests/test_oracle_seedchain.py), so the marks ARE the list
//   emit   the marks are compacted 64 mask words at a time into a position list; 64 lanes at a time rebuild a position's record from the
//          packed bases (sk_minimizer_at: the k-mer is a funnel shift, its reverse complement a bit reversal) and write mz_x / mz_y once,
//          coalesced, at the read's running offset
// LDS per wave: ring 768 w + packed bases and masks 0.5 B per base of tile_cap (w = 10, tile_cap 12 K: 14 KB; eleven reads per CU).
__device__ __forceinline__ uint32_t wave_prefix_add_u32(uint32_t v); // (below, with the other cross-lane helpers)
struct SketchLds {
	int ring_bytes, pk_words, tile_cap, per_wave; // bytes of the ring (also the emit phase's position list), dwords of packed bases (+ 2 in front), bases per tile, bytes per wave
};
__host__ __device__ inline SketchLds sketch_lds_layout(int w, int tile_cap)
{
	SketchLds L;
	L.tile_cap = tile_cap;
	L.ring_bytes = 64 * w * 12;
	if (L.ring_bytes < 1024) L.ring_bytes = 1024;
	L.pk_words = (tile_cap + 256) / 16;
	L.per_wave = L.ring_bytes + (L.pk_words + 2) * 4 + L.pk_words * 2 + tile_cap / 8; // ring | 2 pad dwords + packed bases | ambiguity halves | marks
	L.per_wave = (L.per_wave + 15) & ~15;
	return L;
}
constexpr int SK_WARM_LDS = 128, SK_TAIL_LDS = 128; // bases resident before / after a tile (warm-up; the automaton's run past the stretch's end)

template <bool K32, int W>
__global__ void __launch_bounds__(64) sketch_wave_kernel(SeedChainBuffers B, int w, int k, int tile_cap)
{
	MM2_DYN_LDS(uint64_t, lds);
	const int lane = threadIdx.x;
	const int r = blockIdx.x;
	const SketchLds L = sketch_lds_layout(w, tile_cap);
	uint64_t *const bx = lds + lane;                                   // bx[slot * 64 + lane]
	uint32_t *const by = (uint32_t *)(lds + (size_t)w * 64) + lane;    // by[slot * 64 + lane]
	uint32_t *const pk = (uint32_t *)((uint8_t *)lds + L.ring_bytes) + 2;
	uint16_t *const amb = (uint16_t *)(pk + L.pk_words);
	uint32_t *const marks = (uint32_t *)(amb + L.pk_words);
	uint16_t *const list = (uint16_t *)lds;                            // emit phase: the ring is dead by then
	const int list_cap = L.ring_bytes / 2;
	const uint64_t o = B.seq_off[r];
	const int64_t len = (int64_t)(B.seq_off[r + 1] - o);
	const uint8_t *seq = B.qpool + 2 * o;
	uint64_t *const ox = B.mz_x + B.mz_off[r], *const oy = B.mz_y + B.mz_off[r];
	if (lane == 0) pk[-1] = pk[-2] = 0;
	const int64_t n_tiles = len > tile_cap ? (len + tile_cap - 1) / tile_cap : 1;
	const int64_t tile = n_tiles > 1 ? ((len + n_tiles - 1) / n_tiles + 63) & ~(int64_t)63 : len;
	uint32_t n_out = 0;
	for (int64_t t0 = 0; t0 < len; t0 += tile) {
		const int64_t t1 = t0 + tile < len ? t0 + tile : len;
		const int64_t lo = t0 >= SK_WARM_LDS ? t0 - SK_WARM_LDS : 0;                       // (t0 is a multiple of 64, so lo is one of 16)
		const int64_t hi = t1 + SK_TAIL_LDS < len ? t1 + SK_TAIL_LDS : len;
		// ---- load: 16 bases per lane and step
		for (int64_t c = lo + 16 * lane; c < hi; c += 16 * 64) {
			struct __attribute__((packed, aligned(1))) Q16 { uint32_t v[4]; };
			const Q16 q = *(const Q16 *)(seq + c); // (up to 15 bytes past the read's last base: its reverse-complement block follows)
			const uint32_t qv[4] = { q.v[0], q.v[1], q.v[2], q.v[3] };
			uint32_t packed, flags;
			sk_pack16(qv, &packed, &flags);
			pk[(c - lo) >> 4] = packed, amb[(c - lo) >> 4] = (uint16_t)flags;
		}
		for (int64_t i = lane; i < (t1 - t0 + 31) >> 5; i += 64) marks[i] = 0;
		WAVE_SYNC();
		// ---- sketch
		int64_t chunk = (t1 - t0 + 63) / 64;
		if (chunk < 32) chunk = 32;
		const int64_t cs = t0 + (int64_t)lane * chunk, ce = cs + chunk < t1 ? cs + chunk : t1;
		if (cs < t1) {
			uint32_t wpk = 0, wamb = 0;
			int64_t wcur = -1;
			auto base_at = [&](int64_t i) -> int {
				if (i >= lo && i < hi) {
					const int64_t rel = i - lo;
					if ((rel >> 4) != wcur) wcur = rel >> 4, wpk = pk[wcur], wamb = amb[wcur];
					return (wamb >> (rel & 15) & 1u) ? 4 : (int)(wpk >> (30 - 2 * (int)(rel & 15)) & 3u);
				}
				return (int)seq[i]; // a restart that reaches far back, or a stretch of symmetric k-mers past the tail: rare
			};
			sketch_chunk_core<false, K32, uint32_t, W>(base_at, len, cs, ce, w, k, 0u, bx, by, 64, [&](uint64_t, uint64_t y) {
				const uint32_t rel = (uint32_t)((int64_t)((uint32_t)y >> 1) - t0);
				atomicOr(&marks[rel >> 5], 1u << (rel & 31));
			}, (int64_t)(w + k + 8));
		}
		WAVE_SYNC();
		// ---- emit: rounds of up to 64 mask words (2048 positions), bounded by the list's capacity
		const int n_words = (int)((t1 - t0 + 31) >> 5);
		int words_per_round = list_cap / 32;
		if (words_per_round > 64) words_per_round = 64;
		for (int w0 = 0; w0 < n_words; w0 += words_per_round) {
			uint32_t m = lane < words_per_round && w0 + lane < n_words ? marks[w0 + lane] : 0u;
			const uint32_t cnt = (uint32_t)__popc(m);
			const uint32_t incl = wave_prefix_add_u32(cnt);
			const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int32_t)incl, 63);
			uint32_t at = incl - cnt;
			while (m) { // (positions relative to the tile: < 2^16 by tile_cap)
				list[at++] = (uint16_t)(((w0 + lane) << 5) + __builtin_ctz(m));
				m &= m - 1;
			}
			WAVE_SYNC();
			for (uint32_t e = lane; e < total; e += 64) {
				const int64_t pos = t0 + (int64_t)list[e];
				uint64_t x, y;
				sk_minimizer_at(pk, pos - lo, pos, k, &x, &y);
				ox[n_out + e] = x, oy[n_out + e] = y;
			}
			n_out += total;
			WAVE_SYNC();
		}
	}
	if (lane == 0) B.mz_cnt[r] = n_out;
}

void launch_sketch(const SeedChainBuffers &B, const SeedChainParams &P, int max_len, void *stream)
{
	hipStream_t s = (hipStream_t)stream;
	if (!P.is_hpc && P.w <= 32) {
		// a tile holds the sub-batch's longest read if that takes no more than 16 K bases (positions within a tile are 16-bit in the emit phase's list;
		// LDS: 0.5 B per base): 10 kb ONT reads are one tile, longer reads are cut into equal tiles
		int tile_cap = (std::max(max_len, 1024) + 1023) & ~1023;
		if (tile_cap > 16384) tile_cap = 16384;
		static const int force_tile = getenv("MM2AMD_SKETCH_TILE") ? atoi(getenv("MM2AMD_SKETCH_TILE")) : 0; // tests: several tiles per read
		if (force_tile >= 64) tile_cap = force_tile & ~63;
		const SketchLds L = sketch_lds_layout(P.w, tile_cap);
		// the presets' window sizes are compiled in (the ring scans unroll: sketch_dev.hpp), any other w runs the generic instantiation
#define MM2_SKETCH_LAUNCH(K32_, W_) hipLaunchKernelGGL((sketch_wave_kernel<K32_, W_>), dim3(B.n_reads), dim3(64), (size_t)L.per_wave, s, B, P.w, P.k, tile_cap)
		static const bool generic_only = getenv("MM2AMD_SKETCH_GENERIC") != nullptr; // tests: the runtime-w instantiation on the presets too
		const bool k32 = 2 * P.k <= 32;
		if (generic_only) { if (k32) MM2_SKETCH_LAUNCH(true, 0); else MM2_SKETCH_LAUNCH(false, 0); }
		else if (k32 && P.w == 10) MM2_SKETCH_LAUNCH(true, 10);      // map-ont, map-pb, ava-*
		else if (k32 && P.w == 5) MM2_SKETCH_LAUNCH(true, 5);        // splice
		else if (!k32 && P.w == 19) MM2_SKETCH_LAUNCH(false, 19);    // map-hifi, lr:hq, asm*
		else if (!k32 && P.w == 10) MM2_SKETCH_LAUNCH(false, 10);    // asm20
		else if (k32) MM2_SKETCH_LAUNCH(true, 0);
		else MM2_SKETCH_LAUNCH(false, 0);
#undef MM2_SKETCH_LAUNCH
	} else {
		const dim3 grid((B.n_reads + 63) / 64), block(64);
		if (P.w <= 32) hipLaunchKernelGGL((sketch_kernel<32>), grid, block, 0, s, B, P.w, P.k, P.is_hpc);
		else hipLaunchKernelGGL((sketch_kernel<256>), grid, block, 0, s, B, P.w, P.k, P.is_hpc);
	}
	HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// SDUST (-T): the masked regions of every read are found on host threads (sdust_core.hpp: a sequential automaton with a sorted
// interval list that is rewritten at every base -- the wrong shape for a lane) while the sketch kernel runs, and uploaded into the
// read's still-unused seed slots: sd_n[o] = number of regions, sd_off[o + u] / sd_aoff[o + u] = start / end of region u.  This kernel
// only drops the minimizers that lie mostly inside masked regions (mm_dust_minier, map.c:34-57), one thread per read.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) dust_filter_kernel(SeedChainBuffers B)
{
	const int r = blockIdx.x * 64 + threadIdx.x;
	if (r >= B.n_reads) return;
	const int32_t u0 = B.unit_first ? B.unit_first[r] : 0, nu = B.unit_first ? B.unit_first[r + 1] - u0 : 1;
	for (int32_t k = 0; k < nu; ++k) {
		const uint64_t o = B.unit_first ? B.unit_off[u0 + k] : B.seq_off[r];
		uint32_t *cnt = B.unit_first ? const_cast<uint32_t *>(B.unit_cnt) + (u0 + k) : B.mz_cnt + r;
		const uint32_t *rs = B.sd_off + o, *re = B.sd_aoff + o; // the masked regions of this read
		const int n_reg = (int)B.sd_n[o];
		if (n_reg == 0) continue;
		const int pos_off = B.unit_first ? (int)(o - B.unit_off[u0]) : 0;
		*cnt = (uint32_t)dust_filter_minimizers((int)*cnt, B.mz_x + o, B.mz_y + o, n_reg,
			[&](int u, int32_t *st, int32_t *en) { *st = (int32_t)rs[u], *en = (int32_t)re[u]; }, pos_off);
	}
}

void launch_dust_filter(const SeedChainBuffers &B, void *stream)
{
	hipLaunchKernelGGL(dust_filter_kernel, dim3((B.n_reads + 63) / 64), dim3(64), 0, (hipStream_t)stream, B);
	HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// Seed collection, one wavefront per read
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t idx_lookup(const DevIndex &I, uint64_t hash, uint32_t *off) // mm_idx_get, index.c:93-110
{
	const uint64_t b = hash >> I.key_shift;
	if (b >= (1ull << I.bucket_bits)) return 0;
	uint32_t skip = 0;
	if (I.first) { // (round 6) the bucket's own record: empty, a single other key, or the key itself -- one sector read; only a bucket with more keys goes on
		const IdxSlot f = I.first[b];
		if (f.key == hash) { *off = f.off; return f.cnt & ~kIdxMoreKeys; }
		if (!(f.cnt & kIdxMoreKeys)) return 0;
		skip = 1; // (its first key has been looked at)
	}
	const uint32_t s = I.bucket_start[b] + skip, e = I.bucket_start[b + 1];
	// (round 5) one 16-byte record per key -- key, first position, count -- so the bucket's key scan and the answer come out of the same sector:
	// two dependent sector reads per probe (bucket_start, slots) instead of three to four (bucket_start, keys, val_off[i], val_off[i + 1])
	for (uint32_t i = s; i < e; ++i) {
		const IdxSlot k = I.slots[i];
		if (k.key == hash) { *off = k.off; return k.cnt; }
	}
	return 0;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int popc_below(unsigned long long m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

// Cross-lane moves as DPP / v_readlane instead of ds_bpermute (what __shfl compiles to: an LDS-crossbar round trip per call, and the
// chaining loops make dozens of dependent ones per step).  gfx9 DPP controls: row_shr:n 0x110+n, wave_shr:1 0x138, row_bcast:15 / 31 0x142 / 0x143.
__device__ __forceinline__ int32_t wave_prefix_max_i32(int32_t v) // inclusive prefix maximum over the lanes, lane 0 first
{
	int32_t o;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v; // lane 15 of rows 0 / 2 into rows 1 / 3
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v; // lane 31 into the upper half
	return v;
}
__device__ __forceinline__ uint32_t wave_prefix_add_u32(uint32_t v) // inclusive prefix sum over the lanes
{
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x111, 0xf, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x112, 0xf, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x114, 0xf, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x118, 0xf, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x142, 0xa, 0xf, false);
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)v, 0x143, 0xc, 0xf, false);
	return v;
}
__device__ __forceinline__ int32_t wave_shr1_i32(int32_t first, int32_t v) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false); } // lane i <- v[i-1], lane 0 <- first
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t first, uint64_t v)
{
	return (uint64_t)(uint32_t)wave_shr1_i32((int32_t)(uint32_t)first, (int32_t)(uint32_t)v) | (uint64_t)(uint32_t)wave_shr1_i32((int32_t)(uint32_t)(first >> 32), (int32_t)(uint32_t)(v >> 32)) << 32;
}
__device__ __forceinline__ int32_t lane_get_i32(int32_t v, int l) { return __builtin_amdgcn_readlane(v, l); } // l wave-uniform
__device__ __forceinline__ uint64_t lane_get_u64(uint64_t v, int l)
{
	return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)v, l) | (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)(v >> 32), l) << 32;
}
constexpr int HIST_N = 2048;
constexpr uint32_t SD_TANDEM = 1u << 8, SD_FLT = 1u << 9, SD_SEG1 = 1u << 31; // SD_SEG1: the seed comes from the second read of a pair

// max-heap sift-down on (n<<32 | index) keys, used by the rare high-occurrence thinning (seed.c:56-96)
__device__ void heap_down(uint64_t *h, int i, int n)
{
	int k = i;
	const uint64_t tmp = h[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && h[k] < h[k + 1]) ++k;
		if (h[k] < tmp) break;
		h[i] = h[k]; i = k;
	}
	h[i] = tmp;
}

// skip_seed (map.c:78-100) for one index hit rr of a query minimizer at qp (pos<<1 | strand).  The reference compares the read's
// name with the target's (strcmp); here both are ranks among the distinct sorted reference names: cmp > 0 <=> rank < nm_lb,
// cmp == 0 <=> rank == nm_eq.  *is_self: same-strand hit of a read on its own copy in the index (MM_SEED_SELF).
__device__ __forceinline__ bool skip_hit(int64_t flag, uint64_t rr, uint32_t qp, int qlen, int32_t nm_lb, int32_t nm_eq, const DevIndex &I, bool *is_self)
{
	*is_self = false;
	const bool fwd = (rr & 1) == (qp & 1);
	if (nm_lb >= 0 && (flag & (ref::F_NO_DIAG | ref::F_NO_DUAL))) {
		const uint32_t rid = (uint32_t)(rr >> 32);
		const int32_t rank = I.name_rank[rid];
		if ((flag & ref::F_NO_DIAG) && rank == nm_eq && (int)I.seq_len[rid] == qlen) {
			if ((uint32_t)rr >> 1 == qp >> 1) return true; // the diagonal itself
			if (fwd) *is_self = true;
		}
		if ((flag & ref::F_NO_DUAL) && rank < nm_lb) return true; // each pair once: the read's name sorts after the target's
	}
	if (fwd ? (flag & ref::F_REV_ONLY) != 0 : (flag & ref::F_FOR_ONLY) != 0) return true;
	return false;
}

// the per-read inputs of skip_hit: nm_lb < 0 switches the name rules off (no read names in this batch, or no reference names)
#define HIT_RULES_SETUP() \
	const bool name_rules = (P.flag & (ref::F_NO_DIAG | ref::F_NO_DUAL)) && B.name_lb && I.name_rank; \
	const int32_t nm_lb = name_rules ? B.name_lb[r] : -1, nm_eq = name_rules ? B.name_eq[r] : -1; \
	const bool hit_rules = (P.flag & (ref::F_FOR_ONLY | ref::F_REV_ONLY)) || nm_lb >= 0

__global__ void __launch_bounds__(256) seed_collect_kernel(SeedChainBuffers B, DevIndex I, SeedChainParams P)
{
	__shared__ __attribute__((aligned(8))) uint32_t s_hist[4][HIST_N];
	const int wave = threadIdx.x >> 6, lane = lane_id();
	const int r = blockIdx.x * 4 + wave;
	if (r >= B.n_reads) return;
	const uint64_t mo = B.mz_off[r];
	uint64_t *mx = B.mz_x + mo, *my = B.mz_y + mo;
	uint32_t *sd_n = B.sd_n + mo, *sd_off = B.sd_off + mo, *sd_aoff = B.sd_aoff + mo, *sd_qpos = B.sd_qpos + mo, *sd_info = B.sd_info + mo;
	int n;
	const int qlen = (int)(B.seq_off[r + 1] - B.seq_off[r]);
	uint32_t *hist = s_hist[wave];
	HIT_RULES_SETUP();
	if (B.unit_first) { // a fragment of one or two units; a pair's second list joins the first (collect_minimizers, map.c:59-72)
		const int32_t u0 = B.unit_first[r];
		n = (int)B.unit_cnt[u0];
		if (B.unit_first[r + 1] - u0 == 2) {
			const int n1 = (int)B.unit_cnt[u0 + 1];
			const uint64_t len0 = B.unit_off[u0 + 1] - B.unit_off[u0];
			const uint64_t *sx = mx + len0, *sy = my + len0; // the second unit's minimizers sit at its own slots, len0 further on
			for (int base = 0; base < n1; base += 64) {
				const int i = base + lane;
				uint64_t x = 0, y = 0;
				if (i < n1) x = sx[i], y = sy[i];
				WAVE_SYNC();
				if (i < n1) mx[n + i] = x, my[n + i] = y + (len0 << 1) + (1ULL << 32); // position within the fragment, segment id 1
				WAVE_SYNC();
			}
			n += n1;
		}
	} else n = (int)B.mz_cnt[r];

	// ---- query-side filter of over-represented minimizers (mm_seed_mz_flt, seed.c:5-28) ----
	if (P.q_occ_frac > 0.0f && n > P.q_mid_occ && P.q_mid_occ > 0) {
		for (int i = lane; i < HIST_N; i += 64) hist[i] = 0;
		WAVE_SYNC();
		for (int i = lane; i < n; i += 64) atomicAdd(&hist[(uint32_t)((mx[i] * 0x9E3779B97F4A7C15ull) >> 53)], 1u);
		WAVE_SYNC();
		const float thr = (float)n * P.q_occ_frac;
		bool hot = false;
		for (int i = lane; i < HIST_N; i += 64) { const uint32_t c = hist[i]; hot |= ((int)c > P.q_mid_occ && (float)c > thr); }
		if (__ballot(hot)) { // exact counts, only for minimizers that fall into a crowded bucket
			// pass 1: decide on the untouched list
			for (int base = 0; base < n; base += 64) {
				const int i = base + lane;
				if (i < n) {
					const uint64_t x = mx[i];
					bool keep = true;
					const uint32_t c = hist[(uint32_t)((x * 0x9E3779B97F4A7C15ull) >> 53)];
					if ((int)c > P.q_mid_occ && (float)c > thr) {
						int cnt = 0;
						for (int j = 0; j < n; ++j) cnt += (mx[j] == x);
						keep = !(cnt > P.q_mid_occ && (float)cnt > thr);
					}
					sd_info[i] = keep ? 1u : 0u;
				}
			}
			WAVE_SYNC();
			// pass 2: order-preserving compaction (destination never passes the chunk being read)
			int dst = 0;
			for (int base = 0; base < n; base += 64) {
				const int i = base + lane;
				uint64_t x = 0, y = 0;
				bool keep = false;
				if (i < n) { x = mx[i], y = my[i]; keep = sd_info[i] != 0; }
				const unsigned long long km = __ballot(keep);
				WAVE_SYNC();
				if (keep) { const int d = dst + popc_below(km, lane); mx[d] = x, my[d] = y; }
				dst += __popcll(km);
				WAVE_SYNC();
			}
			n = dst;
		}
	}
	// ---- index probes (mm_seed_collect_all, seed.c:30-52); entries without a hit get n = 0 ----
	for (int i = lane; i < n; i += 64) {
		const uint64_t x = mx[i];
		uint32_t off = 0;
		const uint32_t cnt = idx_lookup(I, x >> 8, &off);
		uint32_t info = (uint32_t)(x & 0xff);
		if (my[i] >> 32) info |= SD_SEG1;
		if (i > 0 && x >> 8 == mx[i - 1] >> 8) info |= SD_TANDEM;
		if (i < n - 1 && x >> 8 == mx[i + 1] >> 8) info |= SD_TANDEM;
		sd_n[i] = cnt, sd_off[i] = off, sd_info[i] = info, sd_qpos[i] = (uint32_t)my[i];
	}
	WAVE_SYNC();
	// ---- compact to the seeds that hit (the reference's m[] array), preserving order ----
	int n_m0 = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		uint32_t cnt = 0, off = 0, info = 0, qp = 0;
		if (i < n) cnt = sd_n[i], off = sd_off[i], info = sd_info[i], qp = sd_qpos[i];
		const unsigned long long hm = __ballot(cnt > 0);
		WAVE_SYNC();
		if (cnt > 0) { const int d = n_m0 + popc_below(hm, lane); sd_n[d] = cnt, sd_off[d] = off, sd_info[d] = info, sd_qpos[d] = qp; }
		n_m0 += __popcll(hm);
		WAVE_SYNC();
	}
	// ---- occurrence filter (seed.c:106-112) ----
	int n_high = 0;
	for (int i = lane; i < n_m0; i += 64) n_high += sd_n[i] > (uint32_t)P.mid_occ;
	for (int o = 32; o > 0; o >>= 1) n_high += __shfl_xor(n_high, o, 64);
	if (n_high > 0) {
		if (P.occ_dist > 0 && P.max_max_occ > P.mid_occ) {
			if (n_m0 > 1 && lane == 0) { // rare, sequential: keep ~1 low-occurrence seed per occ_dist bases in each high-occurrence streak
				uint64_t *const hb = (uint64_t *)hist; // 128 heap entries in the wave's histogram row, which the query-side filter above is done with (a private array was 1 KB of scratch per lane)
				int last0 = -1;
				for (int i = 0; i <= n_m0; ++i) {
					if (i == n_m0 || sd_n[i] <= (uint32_t)P.mid_occ) {
						if (i - last0 > 1) {
							const int ps = last0 < 0 ? 0 : (int)(sd_qpos[last0] >> 1), pe = i == n_m0 ? qlen : (int)(sd_qpos[i] >> 1);
							const int st = last0 + 1, en = i;
							int keep = (int)((double)(pe - ps) / P.occ_dist + .499), j, kk;
							if (keep > 0) {
								if (keep > 128) keep = 128;
#ifdef MM2AMD_WAVE_EMU
								if (getenv("MM2AMD_OCC_HEAP_TRACE")) fprintf(stderr, "[mm2amd] occurrence-filter heap: read %d keeps %d of %d seeds\n", r, keep, en - st);
#endif
								for (j = st, kk = 0; j < en && kk < keep; ++j, ++kk) hb[kk] = (uint64_t)sd_n[j] << 32 | (uint32_t)j;
								for (int q = kk >> 1; q-- > 0;) heap_down(hb, q, kk);
								for (; j < en; ++j)
									if ((int32_t)sd_n[j] < (int32_t)(hb[0] >> 32)) { hb[0] = (uint64_t)sd_n[j] << 32 | (uint32_t)j; heap_down(hb, 0, kk); }
								for (j = 0; j < kk; ++j) sd_info[(uint32_t)hb[j]] |= SD_FLT;
							}
							for (j = st; j < en; ++j) sd_info[j] ^= SD_FLT;
							for (j = st; j < en; ++j) if (sd_n[j] > (uint32_t)P.max_max_occ) sd_info[j] |= SD_FLT;
						}
						last0 = i;
					}
				}
			}
		} else {
			for (int i = lane; i < n_m0; i += 64) if (sd_n[i] > (uint32_t)P.mid_occ) sd_info[i] |= SD_FLT;
		}
		WAVE_SYNC();
	}
	// ---- repetitive length (seed.c:117-123,129): only filtered seeds contribute ----
	int rep_len = 0;
	if (n_high > 0 && lane == 0) {
		int rep_st = 0, rep_en = 0;
		for (int i = 0; i < n_m0; ++i) {
			if (!(sd_info[i] & SD_FLT)) continue;
			const int en = (int)(sd_qpos[i] >> 1) + 1, st = en - (int)(sd_info[i] & 0xff);
			if (st > rep_en) rep_len += rep_en - rep_st, rep_st = st, rep_en = en;
			else rep_en = en;
		}
		rep_len += rep_en - rep_st;
	}
	rep_len = __shfl(rep_len, 0, 64);
	// ---- kept seeds: anchor offsets (exclusive prefix of n) and their rank for mini_pos ----
	uint32_t n_a = 0, n_kept = 0;
	for (int base = 0; base < n_m0; base += 64) {
		const int i = base + lane;
		const bool kept = i < n_m0 && !(sd_info[i] & SD_FLT);
		uint32_t c = kept ? sd_n[i] : 0;
		if (kept && hit_rules) { // hits that skip_seed (map.c:78-100) drops are not counted
			const uint64_t *cr = I.pos + sd_off[i];
			const uint32_t qp = sd_qpos[i];
			uint32_t pass = 0;
			for (uint32_t h = 0; h < c; ++h) {
				bool is_self;
				if (!skip_hit(P.flag, cr[h], qp, qlen, nm_lb, nm_eq, I, &is_self)) ++pass;
			}
			c = pass;
		}
		uint32_t incl = c;
		for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
		const unsigned long long km = __ballot(kept);
		if (i < n_m0) sd_aoff[i] = kept ? n_a + incl - c : 0xffffffffu;
		// rank among kept seeds goes into bits 10..30 of info: up to 2M seeds per read
		if (kept) sd_info[i] = (sd_info[i] & (0x3ffu | SD_SEG1)) | ((n_kept + (uint32_t)popc_below(km, lane)) << 10);
		n_a += __shfl(incl, 63, 64);
		n_kept += (uint32_t)__popcll(km);
	}
	if (lane == 0) {
		B.n_anchor[r] = n_a, B.n_minipos[r] = n_kept, B.n_seedhit[r] = (uint32_t)n_m0, B.rep_len[r] = rep_len;
	}
}

void launch_seed_collect(const SeedChainBuffers &B, const DevIndex &I, const SeedChainParams &P, void *stream)
{
	hipLaunchKernelGGL(seed_collect_kernel, dim3((B.n_reads + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, I, P);
	HIP_CHECK(hipGetLastError());
}

// anchors and mini_pos from the kept seeds (map.c:176-200, seed.c:124); one wavefront per read.
// Round 4: the wave expands 64 seeds at a time COOPERATIVELY -- the seeds' hit counts are prefix-summed over the lanes, every lane then takes
// one (seed, hit) candidate of the flattened list (which seed: the seeds mark their first candidate in a 64-entry LDS row, a prefix maximum
// spreads the marks), candidates that pass skip_seed are compacted by ballot, and consecutive lanes write consecutive anchors: 64 x 8 B
// per store instruction.  (Lane-per-seed, each lane walking its own seed's hits, wrote every anchor as two scattered 8-byte stores: 10x the
// algorithmic bytes, PMC round 3.)  The position lists themselves stay what they are: one 64-byte sector per seed for the typical one or
// two hits.
__global__ void __launch_bounds__(256) seed_expand_kernel(SeedChainBuffers B, DevIndex I, SeedChainParams P)
{
	__shared__ uint32_t s_head[4][64], s_start[4][64], s_off[4][64], s_info[4][64], s_qp[4][64];
	const int wave = threadIdx.x >> 6, lane = lane_id();
	const int r = blockIdx.x * 4 + wave;
	if (r >= B.n_reads) return;
	const uint64_t mo = B.mz_off[r];
	const uint32_t *sd_n = B.sd_n + mo, *sd_off = B.sd_off + mo, *sd_aoff = B.sd_aoff + mo, *sd_qpos = B.sd_qpos + mo, *sd_info = B.sd_info + mo;
	const int n_m0 = (int)B.n_seedhit[r];
	const int qlen = (int)(B.seq_off[r + 1] - B.seq_off[r]);
	HIT_RULES_SETUP();
	uint64_t *akey = B.sort_key_in + B.a_off[r], *aval = B.sort_val_in + B.a_off[r];
	uint64_t *mp = B.mini_pos + B.mp_off[r];
	uint32_t *head = s_head[wave], *c_start = s_start[wave], *c_off = s_off[wave], *c_info = s_info[wave], *c_qp = s_qp[wave];
	uint32_t n_out = 0; // anchors of the read written so far == sd_aoff of the next kept seed (seed_collect_kernel counted with the same rules)
	for (int base = 0; base < n_m0; base += 64) {
		const int i = base + lane;
		const bool kept = i < n_m0 && sd_aoff[i] != 0xffffffffu;
		uint32_t cnt = 0, info = 0, qp = 0, off = 0;
		if (kept) {
			cnt = sd_n[i], info = sd_info[i], qp = sd_qpos[i], off = sd_off[i];
			mp[info << 1 >> 11] = (uint64_t)(info & 0xff) << 32 | (uint64_t)(qp >> 1);
		}
		const uint32_t incl = wave_prefix_add_u32(cnt), start = incl - cnt;
		const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int32_t)incl, 63);
		c_start[lane] = start, c_off[lane] = off, c_info[lane] = info, c_qp[lane] = qp;
		int32_t carry = 0; // (seed lane + 1) that owns the candidate before the group's first
		for (uint32_t e0 = 0; e0 < total; e0 += 64) {
			head[lane] = 0;
			WAVE_SYNC();
			if (cnt > 0 && start - e0 < 64u) head[start - e0] = (uint32_t)lane + 1u; // (start >= e0 by the unsigned compare)
			WAVE_SYNC();
			int32_t v = (int32_t)head[lane];
			if (lane == 0 && v == 0) v = carry;
			v = wave_prefix_max_i32(v);
			carry = __builtin_amdgcn_readlane(v, 63);
			const uint32_t e = e0 + (uint32_t)lane;
			bool pass = e < total;
			uint64_t kx = 0, ky = 0;
			if (pass) {
				const int sl = v - 1;
				const uint32_t s_inf = c_info[sl], s_q = c_qp[sl], span = s_inf & 0xff;
				const uint64_t rr = I.pos[(uint64_t)c_off[sl] + (e - c_start[sl])];
				const uint32_t rpos = (uint32_t)rr >> 1;
				bool is_self = false;
				if (hit_rules && skip_hit(P.flag, rr, s_q, qlen, nm_lb, nm_eq, I, &is_self)) pass = false;
				uint64_t px, py;
				if ((rr & 1) == (s_q & 1)) { // same strand
					px = (rr & 0xffffffff00000000ULL) | rpos;
					py = (uint64_t)span << 32 | (uint64_t)(s_q >> 1);
				} else if (!(P.flag & ref::F_QSTRAND)) {
					px = 1ULL << 63 | (rr & 0xffffffff00000000ULL) | rpos;
					py = (uint64_t)span << 32 | (uint64_t)(uint32_t)(qlen - ((int)(s_q >> 1) + 1 - (int)span) - 1);
				} else { // --qstrand (map.c:192-196): the reference coordinate is flipped, the query's is kept
					px = 1ULL << 63 | (rr & 0xffffffff00000000ULL) | (uint32_t)((int)I.seq_len[rr >> 32] - ((int)rpos + 1 - (int)span) - 1);
					py = (uint64_t)span << 32 | (uint64_t)(s_q >> 1);
				}
				if (s_inf & SD_SEG1) py |= 1ULL << ref::SEED_SEG_SHIFT;
				if (s_inf & SD_TANDEM) py |= ref::SEED_TANDEM;
				if (is_self) py |= ref::SEED_SELF;
				kx = (px >> 63) << (32 + B.rid_bits) | (px & 0x7fffffffffffffffULL); // compact sort key: strand | rid | rpos (anchor_sort_kernel)
				ky = py;
			}
			const unsigned long long pm = __ballot(pass);
			if (pass) { const uint32_t d = n_out + (uint32_t)popc_below(pm, lane); akey[d] = kx, aval[d] = ky; }
			n_out += (uint32_t)__popcll(pm);
		}
		WAVE_SYNC(); // (the next chunk rewrites the seeds' LDS rows)
	}
}

void launch_seed_expand(const SeedChainBuffers &B, const DevIndex &I, const SeedChainParams &P, void *stream)
{
	hipLaunchKernelGGL(seed_expand_kernel, dim3((B.n_reads + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, I, P);
	HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// Anchor sort (map.c:202: radix_sort_128x by x, an UNSTABLE in-place sort whose tie order is observable), one workgroup per read.
//
// The anchors of a read are contiguous, so the read is the unit: its (x key, original index) pairs go into LDS -- one 64-bit word per
// anchor, compact key (strand | rid | rpos: 33 + rid_bits bits) above a 14-bit index -- and are sorted there by a stable LSD radix sort
// (lds_radix_sort: five 8-bit passes for a 38-bit key, elements held in registers between a pass's read and write, so one LDS buffer
// suffices).  One trip through LDS instead of seven radix passes over all anchors through HBM.
// Where no two anchors of the read share x, the sorted order is unique and therefore the reference's.  Where they do (the same
// reference position hit from two query positions: tandem repeats inside the read), the order of the equal anchors is whatever the
// reference's in-place MSD radix sort (ksort.h:101-151) leaves, which depends on the whole array: the same workgroup then reloads the
// read in its ORIGINAL order and replays that sort permutation-exactly (tie_exact_replay), restricted to the chain of buckets that
// lead to duplicated keys, and rewrites the anchors that carry one.  Reads with more anchors than the LDS classes hold (whole contigs
// as queries; rare) are sorted on global scratch by a comparison network (bitonic_sort) and replayed there.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t key_to_x(uint64_t key, int rid_bits)
{
	const uint64_t low = key & ((1ULL << (32 + rid_bits)) - 1ULL);
	return (key >> (32 + rid_bits) & 1ULL) << 63 | low;
}

constexpr int AS_IDX_BITS = 14;                 // index bits of a packed element
constexpr int AS_LDS_MAX = 10240;               // anchors of the largest LDS class (80 KB of elements: one read per CU at a time, sorted by 1024 threads)
constexpr int TIE_MAX_KEYS = 64;                // distinct duplicated keys tracked per read; more => every bucket is replayed
constexpr int AS_STACK = (1 << AS_IDX_BITS) / 65 + 2; // the frames of a replay are disjoint ranges of more than 64 elements

struct TieFrame { int32_t b, e, shift; };

// What the sort and the replay move: the x of an anchor and the anchor's index in the read's original order.
struct PackedStore { // one LDS word per anchor: compact key << 13 | index
	uint64_t *E;
	int rid_bits;
	typedef uint64_t Elem;
	__device__ __forceinline__ uint64_t xk(Elem e) const { return key_to_x(e >> AS_IDX_BITS, rid_bits); }
	__device__ __forceinline__ uint64_t xkey(int32_t i) const { return xk(E[i]); }
	__device__ __forceinline__ uint32_t index(int32_t i) const { return (uint32_t)E[i] & ((1u << AS_IDX_BITS) - 1u); }
	__device__ __forceinline__ void set(int32_t i, uint64_t composite, uint32_t idx) { E[i] = composite << AS_IDX_BITS | idx; }
	__device__ __forceinline__ Elem get(int32_t i) const { return E[i]; }
	__device__ __forceinline__ void put(int32_t i, Elem e) { E[i] = e; }
	__device__ __forceinline__ void order(int32_t i, int32_t l) { const uint64_t a = E[i], b = E[l]; if (a > b) E[i] = b, E[l] = a; } // ascending (key, index)
};
struct SplitElem { uint64_t k; uint32_t i; };
struct SplitStore { // key and index in arrays of their own: global scratch, or the chain-end sort's LDS arrays
	uint64_t *K;
	uint32_t *I;
	int rid_bits; // < 0: set() takes x itself
	typedef SplitElem Elem;
	__device__ __forceinline__ uint64_t xk(const Elem &e) const { return e.k; }
	__device__ __forceinline__ uint64_t xkey(int32_t i) const { return K[i]; }
	__device__ __forceinline__ uint32_t index(int32_t i) const { return I[i]; }
	__device__ __forceinline__ void set(int32_t i, uint64_t composite, uint32_t idx) { K[i] = rid_bits < 0 ? composite : key_to_x(composite, rid_bits), I[i] = idx; }
	__device__ __forceinline__ Elem get(int32_t i) const { return Elem{K[i], I[i]}; }
	__device__ __forceinline__ void put(int32_t i, const Elem &e) { K[i] = e.k, I[i] = e.i; }
	__device__ __forceinline__ void order(int32_t i, int32_t l)
	{
		const uint64_t a = K[i], b = K[l];
		const uint32_t ai = I[i], bi = I[l];
		if (a > b || (a == b && ai > bi)) K[i] = b, K[l] = a, I[i] = bi, I[l] = ai;
	}
};

// Bitonic sorting network over s[0..n), ascending, by the whole workgroup.  Every stage first mirrors each block of k elements onto
// itself (i <-> block end - i) and then halves distances k/4 .. 1, and every compare-exchange puts the smaller element at the lower
// position -- so virtual elements at positions >= n, being +infinity, never move and their pairs are simply skipped.
template <class S>
__device__ void bitonic_sort(S s, int32_t n)
{
	if (n < 2) return;
	const int32_t tid = (int32_t)threadIdx.x, nt = (int32_t)blockDim.x;
	int lp = 1;
	while ((1 << lp) < n) ++lp;
	const int32_t n_pairs = 1 << (lp - 1); // pairs of the padded network; a pair whose upper element lies past the end is skipped
	for (int lk = 1; lk <= lp; ++lk) {
		const int32_t k = 1 << lk, hk = k >> 1;
#pragma unroll 4
		for (int32_t t = tid; t < n_pairs; t += nt) { // mirror: pair t of block b = t / (k/2)
			const int32_t blk = t >> (lk - 1), q = t & (hk - 1), i = blk * k + q, l = blk * k + k - 1 - q;
			if (l < n) s.order(i, l);
		}
		__syncthreads();
		for (int32_t j = k >> 2; j > 0; j >>= 1) {
#pragma unroll 4
			for (int32_t t = tid; t < n_pairs; t += nt) {
				const int32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
				if (l < n) s.order(i, l);
			}
			__syncthreads();
		}
	}
}

// LSD radix sort of the packed elements E[0..n) in LDS by the key above the index bits, 8 bits per pass, stable, by a workgroup of T
// threads; ONE buffer: a pass reads every element into registers (R per lane), ranks it, and writes it back to its new place after a
// barrier.  The array is cut into one contiguous segment per wavefront and a segment is walked in rounds of 64 consecutive elements
// (lane = position), so "earlier in the array" is (wave, round, lane) order and an element's rank among those with its digit is
//   base[digit][wave]  (digits below it, and its digit in the waves before: a scan over the per-wave counters)
// + count of its digit in the wave's earlier rounds  (the wave's counter when the round reaches it)
// + lanes below it in its round with the same digit  (the lanes holding a digit are found with eight ballots).
// Five passes for a 38-bit key: ~3 % of the LDS traffic of a comparison network over the same array.
template <int T, int R>
__device__ void lds_radix_sort(uint64_t *E, int32_t n, int key_bits, uint32_t *tab /* [T / 64][256] */, uint32_t *dig_tot /* [256] */)
{
	static_assert(T >= 256 && T % 64 == 0, "the 256 digits are scanned by the first 256 threads");
	constexpr int NW = T / 64;
	const int tid = (int)threadIdx.x, lane = tid & 63, w = tid >> 6;
	const int32_t seg = (n + NW - 1) / NW, w0 = w * seg, w1 = w0 + seg < n ? w0 + seg : n; // seg <= R * 64 (the launch classes see to it)
	uint32_t *const mine = tab + w * 256;
	const unsigned long long below = (1ull << lane) - 1ull;
	for (int shift = AS_IDX_BITS; shift < AS_IDX_BITS + key_bits; shift += 8) {
		uint64_t x[R];
		uint32_t rk[R];
		for (int k = lane; k < 256; k += 64) mine[k] = 0;
		WAVE_SYNC();
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const int32_t i = w0 + r * 64 + lane;
			const bool ok = i < w1;
			x[r] = ok ? E[i] : 0ull;
			const uint32_t d = (uint32_t)(x[r] >> shift) & 255u;
			unsigned long long same = __ballot(ok); // the valid lanes of this round that hold the same digit
#pragma unroll
			for (int b = 0; b < 8; ++b) {
				const unsigned long long bal = __ballot((d >> b & 1u) != 0);
				same &= (d >> b & 1u) ? bal : ~bal;
			}
			const uint32_t before = (uint32_t)__popcll(same & below);
			uint32_t old = 0;
			if (ok) old = mine[d];
			MM2_LOCKSTEP(); // every lane has read its digit's counter before the digit's first lane moves it on
			if (ok && before == 0) mine[d] = old + (uint32_t)__popcll(same);
			WAVE_SYNC();
			rk[r] = old + before;
		}
		__syncthreads();
		if (tid < 256) { // a digit's counts over the waves become the waves' offsets inside the digit; its total goes to the scan
			uint32_t acc = 0;
#pragma unroll
			for (int ww = 0; ww < NW; ++ww) { const uint32_t c = tab[ww * 256 + tid]; tab[ww * 256 + tid] = acc; acc += c; }
			dig_tot[tid] = acc;
		}
		__syncthreads();
		if (w == 0) { // exclusive scan of the 256 totals: four per lane
			const uint32_t a0 = dig_tot[lane * 4], a1 = dig_tot[lane * 4 + 1], a2 = dig_tot[lane * 4 + 2], a3 = dig_tot[lane * 4 + 3];
			const uint32_t sum = a0 + a1 + a2 + a3, excl = wave_prefix_add_u32(sum) - sum;
			MM2_LOCKSTEP();
			dig_tot[lane * 4] = excl, dig_tot[lane * 4 + 1] = excl + a0, dig_tot[lane * 4 + 2] = excl + a0 + a1, dig_tot[lane * 4 + 3] = excl + a0 + a1 + a2;
		}
		__syncthreads();
#pragma unroll
		for (int r = 0; r < R; ++r) {
			const int32_t i = w0 + r * 64 + lane;
			if (i < w1) { const uint32_t d = (uint32_t)(x[r] >> shift) & 255u; E[dig_tot[d] + mine[d] + rk[r]] = x[r]; }
		}
		__syncthreads();
	}
}

// LDS of the replay's tape walk (anchor_sort_ties_kernel): see "the walk over tapes" in tie_exact_replay
struct TapeScratch { // (passed by value: the kernel's LDS pointers stay LDS pointers -- ds_read / ds_write, not flat accesses -- after inlining)
	uint16_t *pos = nullptr;  // [cap] a tape entry's position in the range being partitioned
	uint16_t *via = nullptr;  // [cap] the entry whose slot the entry's element ends up taking
	uint8_t *dig = nullptr;   // [cap] the bucket the entry's element belongs to
	uint32_t *tab = nullptr;  // [3 * 256 + 32]: first entry / end / entries taken before the bucket's own turn, per bucket; a control word; one word per wave; the buckets with a tape, one bit each
	int cap = 0;              // 0: no tapes, partitions are walked by one thread (tapes need a workgroup of at least 256 threads)
};
constexpr int TAPE_MIN_LEN = 96; // shorter ranges: the one-thread walk costs less than the tape's passes

// Workgroup-cooperative, permutation-exact replay of radix_sort_128x (ksort.h:101-151) restricted to what can matter.
//
// s[0..n) holds the keys in the ORIGINAL (pre-sort) order with their original indices.  The reference's MSD radix sort partitions a
// range by one key byte with an in-place cycle-leader walk (sequential, replayed here by thread 0; the byte histogram before it is
// computed by all threads), then recurses into buckets of more than 64 elements and insertion-sorts (stably) the smaller ones.
// Sibling buckets are independent, and a bucket that contains no duplicated key ends in the unique sorted order whatever its
// incoming order was -- which the network above already produced.  So only the chain of buckets leading to duplicated keys is
// replayed (child_mask: the children of the current range that hold one); everything else is skipped.  Afterwards the elements with
// duplicated keys sit at their final positions.  Control flow is uniform over the workgroup: every decision is read from shared
// memory after a barrier.
template <class S, int TWO_PER = 1>
__device__ __forceinline__ void tie_exact_replay(S s, int32_t n, const uint64_t *tied, int n_tied, bool replay_all, uint32_t *cnt, uint32_t *head, uint32_t *start, uint32_t *child_mask,
                                 TieFrame *stack, int stack_cap, uint32_t *two_scratch = nullptr, int two_cap = 0, const TapeScratch tape = TapeScratch())
{
	// two_scratch / two_cap: LDS words for the two-bucket closed form below (20 control words, then two_cap positions as 16-bit entries); none: always walk
	// TWO_PER: elements a thread holds in registers in the closed form (the caller's class: anchors per read / threads)
	const int32_t tid = (int32_t)threadIdx.x, nt = (int32_t)blockDim.x;
	auto insertion = [&](int32_t b, int32_t e) { // rs_insertsort (ksort.h:105-115): stable
		for (int32_t i = b + 1; i < e; ++i)
			if (s.xkey(i) < s.xkey(i - 1)) {
				const typename S::Elem te = s.get(i);
				const uint64_t tk = s.xk(te);
				int32_t j;
				for (j = i; j > b && tk < s.xkey(j - 1); --j) s.put(j, s.get(j - 1));
				s.put(j, te);
			}
	};
	if (n <= 64) { // radix_sort top level (ksort.h:149)
		if (tid == 0) insertion(0, n);
		__syncthreads();
		return;
	}
	int sp = 0;
	if (tid == 0) stack[0] = TieFrame{0, n, 56};
	sp = 1;
	__syncthreads();
	while (sp > 0) {
		const TieFrame fr = stack[--sp];
		const int32_t len = fr.e - fr.b;
		for (int k = tid; k < 256; k += nt) cnt[k] = 0;
		if (tid < 8) child_mask[tid] = 0;
		__syncthreads();
		for (int32_t i = fr.b + tid; i < fr.e; i += nt) atomicAdd(&cnt[s.xkey(i) >> fr.shift & 255], 1u);
		if (!replay_all && fr.shift > 0) { // duplicated keys inside this range share its bits above the byte being partitioned
			const uint64_t prefix = fr.shift >= 56 ? 0 : s.xkey(fr.b) >> (fr.shift + 8);
			for (int t = tid; t < n_tied; t += nt)
				if ((fr.shift >= 56 ? 0 : tied[t] >> (fr.shift + 8)) == prefix) { const uint32_t d = (uint32_t)(tied[t] >> fr.shift & 255); atomicOr(&child_mask[d >> 5], 1u << (d & 31)); }
		}
		__syncthreads();
		if (tid < 64) { // the first wavefront: bucket starts and ends from the 256 counts, four per lane (one thread's loop over them cost 70 k cycles a range)
			const int l4 = tid * 4;
			const uint32_t a0 = cnt[l4], a1 = cnt[l4 + 1], a2 = cnt[l4 + 2], a3 = cnt[l4 + 3];
			const uint32_t sum = a0 + a1 + a2 + a3, excl = wave_prefix_add_u32(sum) - sum;
			const uint32_t m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3;
			const uint32_t mx = (uint32_t)lane_get_i32(wave_prefix_max_i32((int32_t)(m01 > m23 ? m01 : m23)), 63);
			const uint32_t n_nonempty = (uint32_t)lane_get_i32((int32_t)wave_prefix_add_u32((a0 != 0) + (a1 != 0) + (a2 != 0) + (a3 != 0)), 63);
			const unsigned long long nz = __ballot(sum != 0); // (not 0: the range is not empty)
			const int first_lane = __builtin_ffsll((long long)nz) - 1, last_lane = 63 - __builtin_clzll(nz);
			const uint32_t dA = (uint32_t)lane_get_i32(l4 + (a0 ? 0 : a1 ? 1 : a2 ? 2 : 3), first_lane); // the first and the last bucket in use
			const uint32_t dB = (uint32_t)lane_get_i32(l4 + (a3 ? 3 : a2 ? 2 : a1 ? 1 : 0), last_lane);
			MM2_LOCKSTEP();
			start[l4] = head[l4] = excl, cnt[l4] = excl + a0; // cnt becomes the bucket end
			start[l4 + 1] = head[l4 + 1] = excl + a0, cnt[l4 + 1] = excl + a0 + a1;
			start[l4 + 2] = head[l4 + 2] = excl + a0 + a1, cnt[l4 + 2] = excl + a0 + a1 + a2;
			start[l4 + 3] = head[l4 + 3] = excl + a0 + a1 + a2, cnt[l4 + 3] = excl + sum;
			WAVE_SYNC();
		  if (tid == 0) {
			// (measured: the same walk as wave-scalar code with the bucket heads in registers -- v_readlane / v_writelane instead of the LDS
			// tables -- was 35 % slower: the chain is one LDS round trip per element either way, and the scalar round trips cost more than
			// the table reads they replace)
			// exactly two buckets (always so at the top: the strand bit): the walk's result has a closed form, computed by all threads below
			const bool two = two_scratch && n_nonempty == 2 && len <= nt * TWO_PER && (int32_t)(cnt[dA] - start[dA] < cnt[dB] - start[dB] ? cnt[dA] - start[dA] : cnt[dB] - start[dB]) <= two_cap;
			if (two_scratch) two_scratch[0] = two ? 1u : 0u, two_scratch[1] = dA, two_scratch[2] = dB, two_scratch[3] = cnt[dA] - start[dA];
			const bool by_tape = tape.cap > 0 && !two && (int32_t)mx != len && len >= TAPE_MIN_LEN && len <= nt * TWO_PER && len <= tape.cap;
			if (tape.cap > 0) tape.tab[768] = by_tape ? 1u : 0u;
			if (!two && !by_tape && (int32_t)mx != len) { // not all in one bucket: the cycle-leader walk (ksort.h:126-138)
				for (int k = 0; k < 256;) {
					if (head[k] != cnt[k]) {
						int l = (int)(s.xkey(fr.b + (int32_t)head[k]) >> fr.shift & 255);
						if (l != k) {
							typename S::Elem te = s.get(fr.b + (int32_t)head[k]), se;
							do {
								se = te;
								te = s.get(fr.b + (int32_t)head[l]);
								s.put(fr.b + (int32_t)head[l], se);
								++head[l];
								l = (int)(s.xk(te) >> fr.shift & 255);
							} while (l != k);
							s.put(fr.b + (int32_t)head[k], te);
							++head[k];
						} else ++head[k];
					} else ++k;
				}
			}
		  }
		}
		__syncthreads();
		if (two_scratch && two_scratch[0]) {
#ifdef MM2AMD_WAVE_EMU
			if (tid == 0 && getenv("MM2AMD_TWO_BUCKET_TRACE")) fprintf(stderr, "[mm2amd] two-bucket closed form: %d elements at shift %d\n", (int)len, (int)fr.shift);
#endif
			// Two buckets A < B, regions [b, b + cA) and [b + cA, e).  The walk (ksort.h:126-138) handles A first: an element of B found in A's region
			// ("x_j", the j-th such) starts a cycle that puts it at B's head; B's own elements met there move one slot on, one after the other,
			// until an element of A ("z_j") turns up in B's region, which takes x_j's slot.  So: A's region keeps its own elements and z_j replaces
			// x_j; B's region becomes  x_1 B_0 x_2 B_1 ... x_m B_{m-1} B_m  (B_i: B's elements between z_i and z_{i+1}): x_1 at offset 0, x_j right
			// after z_{j-1}'s old slot, B's elements before z_m shifted by one, those after it where they were.
			const uint32_t dA = two_scratch[1], dB = two_scratch[2];
			const int32_t cA = (int32_t)two_scratch[3], regB = fr.b + cA;
			uint16_t *const posx = (uint16_t *)(two_scratch + 20);
			uint32_t *const wave_tot = two_scratch + 4; // words 4..19: one per wave (a workgroup has at most 16)
			const int32_t per = (len + nt - 1) / nt, p0 = fr.b + tid * per, p1 = p0 + per < fr.e ? p0 + per : fr.e;
			typename S::Elem el[TWO_PER], prev_el = typename S::Elem();
			uint32_t fmask = 0, mine = 0; // foreign flags of the thread's own positions
			bool prev_foreign = false;
#pragma unroll
			for (int k = 0; k < TWO_PER; ++k) {
				const int32_t p = p0 + k;
				el[k] = typename S::Elem();
				if (p < p1) {
					el[k] = s.get(p);
					const uint32_t d = (uint32_t)(s.xk(el[k]) >> fr.shift & 255);
					if (d == (p < regB ? dB : dA)) fmask |= 1u << k, ++mine;
				}
			}
			if (p0 < p1 && p0 > regB) { prev_el = s.get(p0 - 1); prev_foreign = (uint32_t)(s.xk(prev_el) >> fr.shift & 255) == dA; }
			// exclusive rank of the thread's first position among all foreign elements of the frame
			const uint32_t incl = wave_prefix_add_u32(mine);
			const int wv = tid >> 6, n_wv = (nt + 63) >> 6;
			if ((tid & 63) == 63) wave_tot[wv] = incl;
			__syncthreads();
			uint32_t base = incl - mine, total = 0;
			for (int w2 = 0; w2 < n_wv; ++w2) { const uint32_t t = wave_tot[w2]; if (w2 < wv) base += t; total += t; }
			const uint32_t m = total >> 1; // as many x's as z's
			{
				uint32_t r = base;
#pragma unroll
				for (int k = 0; k < TWO_PER; ++k) if (fmask >> k & 1u) { const int32_t p = p0 + k; if (p < regB) posx[r] = (uint16_t)(p - fr.b); ++r; }
			}
			__syncthreads();
			{ // B's region (every element was read above; the x's still sit in A's region: the z's move there after the next barrier)
				uint32_t r = base;
#pragma unroll
				for (int k = 0; k < TWO_PER; ++k) {
					const int32_t p = p0 + k;
					if (p < p1 && p >= regB) {
						const uint32_t cz = r - m; // z's strictly before p
						if (cz < m) {
							const bool pf = k > 0 ? (fmask >> (k > 0 ? k - 1 : 0) & 1u) && p - 1 >= regB : prev_foreign; // is the slot before a z?
							if (p == regB) s.put(p, s.get(fr.b + (int32_t)posx[0]));
							else if (pf) s.put(p, s.get(fr.b + (int32_t)posx[cz]));
							else s.put(p, k > 0 ? el[k > 0 ? k - 1 : 0] : prev_el);
						}
					}
					if (fmask >> k & 1u) ++r;
				}
			}
			__syncthreads();
			{ // z_{cz + 1} takes x_{cz + 1}'s slot
				uint32_t r = base;
#pragma unroll
				for (int k = 0; k < TWO_PER; ++k) {
					const int32_t p = p0 + k;
					if (fmask >> k & 1u) {
						if (p >= regB) s.put(fr.b + (int32_t)posx[r - m], el[k]);
						++r;
					}
				}
			}
			__syncthreads();
		}
		if (tape.cap > 0 && tape.tab[768]) {
			// The walk over tapes.  What the cycle-leader walk (ksort.h:126-138) does, bucket by bucket: in bucket l's part of the range call the
			// elements that belong elsewhere foreign; their (position, bucket) pairs, in position order, are l's TAPE.  The walk is then: at the
			// base bucket k (0, 1, 2, ... in turn) take the next entry of k's tape, follow the element to its bucket d, where it takes the place
			// of the next entry of d's tape, whose element travels on the same way, until one that belongs to k turns up and closes the cycle
			// in the slot the cycle started from.  In a bucket that is not the base an arriving element is put at the bucket's head and the
			// bucket's own elements between the head and the displaced foreign one each move one slot on (x_1 N_0 x_2 N_1 ...: the j-th arrival
			// lands right after the (j-1)-th foreign slot, the first at the bucket's start); in the base bucket own elements stay and an arrival
			// takes the foreign slot itself.  So the only sequential part is which entry's slot each travelling element consumes -- one LDS round
			// trip per element over byte-sized tapes (thread 0) -- and the tapes before it and all the moves after it are done by the whole
			// workgroup.
#ifdef MM2AMD_WAVE_EMU
			if (tid == 0 && getenv("MM2AMD_TWO_BUCKET_TRACE")) fprintf(stderr, "[mm2amd] tape walk: %d elements at shift %d\n", (int)len, (int)fr.shift);
#endif
			uint16_t *const tpos = tape.pos, *const via = tape.via;
			uint8_t *const tdig = tape.dig;
			uint32_t *const ent = head; // per bucket: next entry of its tape << 8 | that entry's bucket
			uint32_t *const Fs = tape.tab, *const Fe = tape.tab + 256, *const m1 = tape.tab + 512, *const wave_tot = tape.tab + 769, *const has_tape = tape.tab + 785;
			const int32_t per = (len + nt - 1) / nt, p0 = fr.b + tid * per, p1 = p0 + per < fr.e ? p0 + per : fr.e;
			typename S::Elem el[TWO_PER];
			uint32_t rg[TWO_PER]; // the bucket whose part of the range the position lies in
			uint32_t fmask = 0, mine = 0;
			if (tid < 256) Fs[tid] = 0, Fe[tid] = 0;
			if (tid < 8) has_tape[tid] = 0;
			{
				int32_t reg = 0;
				if (p0 < p1) { // first bucket whose end lies beyond the position (empty buckets end where they start)
					const uint32_t q0 = (uint32_t)(p0 - fr.b);
					int lo = 0, hi = 255;
					while (lo < hi) { const int mid = (lo + hi) >> 1; if (cnt[mid] > q0) hi = mid; else lo = mid + 1; }
					reg = lo;
				}
#pragma unroll
				for (int k = 0; k < TWO_PER; ++k) {
					const int32_t p = p0 + k;
					el[k] = typename S::Elem();
					rg[k] = 0;
					if (p < p1) {
						while (cnt[reg] <= (uint32_t)(p - fr.b)) ++reg;
						rg[k] = (uint32_t)reg;
						el[k] = s.get(p);
						if ((uint32_t)(s.xk(el[k]) >> fr.shift & 255) != (uint32_t)reg) fmask |= 1u << k, ++mine;
					}
				}
			}
			const uint32_t incl = wave_prefix_add_u32(mine);
			const int wv = tid >> 6, n_wv = (nt + 63) >> 6;
			if ((tid & 63) == 63) wave_tot[wv] = incl;
			__syncthreads();
			uint32_t base = incl - mine, total = 0;
			for (int w2 = 0; w2 < n_wv; ++w2) { const uint32_t t = wave_tot[w2]; if (w2 < wv) base += t; total += t; }
			{
				uint32_t r = base;
#pragma unroll
				for (int k = 0; k < TWO_PER; ++k) {
					const int32_t p = p0 + k;
					if (p < p1) {
						const uint32_t q = (uint32_t)(p - fr.b);
						if (q == start[rg[k]]) Fs[rg[k]] = r;
						if (fmask >> k & 1u) { tpos[r] = (uint16_t)q, tdig[r] = (uint8_t)(s.xk(el[k]) >> fr.shift & 255); ++r; }
						if (q + 1 == cnt[rg[k]]) Fe[rg[k]] = r;
					}
				}
			}
			__syncthreads();
			if (tid < 256) {
				ent[tid] = Fs[tid] << 8 | (Fs[tid] < Fe[tid] ? (uint32_t)tdig[Fs[tid]] : 0u);
				m1[tid] = Fs[tid]; // (stays so for a bucket without foreign elements: nothing arrives there)
				if (Fs[tid] < Fe[tid]) atomicOr(&has_tape[tid >> 5], 1u << (tid & 31));
			}
			__syncthreads();
			if (tid == 0) {
				const uint32_t last = total ? total - 1 : 0;
				for (uint32_t kw = 0; kw < 8; ++kw)
				for (uint32_t km = has_tape[kw]; km; km &= km - 1) { // the base bucket, in bucket order
					const uint32_t k = kw * 32 + (uint32_t)__builtin_ctz(km);
					uint32_t wk = ent[k];
					const uint32_t fe = Fe[k];
					m1[k] = wk >> 8; // entries before this one were taken by arrivals while an earlier bucket was the base
					while ((wk >> 8) < fe) {
						const uint32_t e0 = wk >> 8, nk = e0 + 1;
						uint32_t dst = wk & 255u, src = e0;
						uint32_t w2 = ent[dst]; // (dst != k: the entry is foreign to k)
						wk = nk << 8 | (nk < fe ? (uint32_t)tdig[nk] : 0u);
						for (;;) {
							const uint32_t t = w2 >> 8, nxt = w2 & 255u; // the element of entry src takes entry t's slot; t's element belongs to nxt (!= dst)
							uint32_t w3 = 0;
							if (nxt != k) w3 = ent[nxt]; // the next step waits for this one only (issued before the stores below): one LDS round trip per element
							const uint32_t nd = tdig[t < last ? t + 1 : t];
							via[src] = (uint16_t)t;
							ent[dst] = (t + 1) << 8 | nd;
							src = t;
							if (nxt == k) break;
							dst = nxt, w2 = w3;
						}
						via[src] = (uint16_t)e0; // the cycle closes in the slot it started from
					}
				}
			}
			__syncthreads();
			{
				uint32_t r = base;
				int32_t dpos[TWO_PER];
#pragma unroll
				for (int k = 0; k < TWO_PER; ++k) {
					const int32_t p = p0 + k;
					dpos[k] = -1;
					if (p < p1) {
						const uint32_t q = (uint32_t)(p - fr.b);
						if (fmask >> k & 1u) {
							const uint32_t d = (uint32_t)(s.xk(el[k]) >> fr.shift & 255), t = via[r];
							if (t < m1[d]) dpos[k] = t == Fs[d] ? (int32_t)start[d] : (int32_t)tpos[t - 1] + 1; // arrived before d was the base
							else dpos[k] = (int32_t)tpos[t];
							++r;
						} else dpos[k] = (int32_t)q + (r < m1[rg[k]] ? 1 : 0); // an own element moves one slot on if a foreign slot after it was taken early
					}
				}
#pragma unroll
				for (int k = 0; k < TWO_PER; ++k) if (dpos[k] >= 0 && dpos[k] != p0 + k - fr.b) s.put(fr.b + dpos[k], el[k]);
			}
			__syncthreads();
		}
		if (fr.shift == 0) continue;
		const int ns = fr.shift > 8 ? fr.shift - 8 : 0;
		// children: a thread per bucket -- the small ones are insertion-sorted side by side, the others join the stack (in any order: the ranges
		// are disjoint and what happens in one does not depend on the others)
		if (tid == 0) head[0] = (uint32_t)sp; // (the heads are dead after the walk)
		__syncthreads();
		for (int k = tid; k < 256; k += nt) {
			const int32_t cb = fr.b + (int32_t)start[k], ce = fr.b + (int32_t)cnt[k];
			if (ce - cb <= 1) continue;
			if (!replay_all && !(child_mask[k >> 5] >> (k & 31) & 1u)) continue;
			if (ce - cb > 64) {
				const uint32_t slot = atomicAdd(&head[0], 1u);
				if ((int)slot < stack_cap) stack[slot] = TieFrame{cb, ce, ns}; // cannot overflow: the frames are disjoint ranges of more than 64 elements
			} else insertion(cb, ce);
		}
		__syncthreads();
		sp = (int)head[0] < stack_cap ? (int)head[0] : stack_cap; // (two barriers away from the next range's scan, which writes the heads again)
	}
}

template <int THREADS, int ROUNDS, bool IN_LDS>
__global__ void __launch_bounds__(THREADS, (THREADS == 1024 && IN_LDS && ROUNDS <= 7) ? 8 : 4) /* (waves per SIMD) the 7 k class: two workgroups of 1024 per CU, 64 VGPRs */ anchor_sort_kernel(SeedChainBuffers B, const uint32_t *list, int heap_sort)
{
	MM2_DYN_LDS(uint64_t, as_lds); // IN_LDS: the read's packed elements
	__shared__ uint32_t tab[(THREADS / 64) * 256]; // the radix passes' per-wave digit counters; afterwards the replay's three 256-entry tables
	__shared__ uint32_t dig_tot[256], child_mask[8];
	__shared__ uint64_t tied[TIE_MAX_KEYS];
	__shared__ TieFrame lstack[AS_STACK];
	__shared__ uint32_t n_tied_s;
	uint32_t *const cnt = tab, *const head = tab + 256, *const start = tab + 512;
	const int32_t tid = (int32_t)threadIdx.x;
	const int r = (int)list[blockIdx.x];
	const uint64_t ao = B.a_off[r];
	const int32_t n = (int32_t)(B.a_off[r + 1] - ao);
	const uint64_t *kin = B.sort_key_in + ao, *vin = B.sort_val_in + ao;
	Anchor *out = B.anchors + ao;
	if (n == 0) return;
	auto run = [&](auto s, TieFrame *stack, int stack_cap) {
		// 1. (key, original index) pairs, sorted by (key, index)
		for (int32_t i = tid; i < n; i += THREADS) s.set(i, kin[i], (uint32_t)i);
		if (tid == 0) n_tied_s = 0;
		__syncthreads();
		if constexpr (IN_LDS) lds_radix_sort<THREADS, ROUNDS>(as_lds, n, 33 + B.rid_bits, tab, dig_tot); // stable: equal keys stay in index order
		else bitonic_sort(s, n);
		// 2. the anchors in sorted order; which keys occur more than once
		for (int32_t i = tid; i < n; i += THREADS) {
			const uint64_t x = s.xkey(i);
			Anchor a;
			a.x = x, a.y = vin[s.index(i)];
			out[i] = a;
			if (i + 1 < n && s.xkey(i + 1) == x && (i == 0 || s.xkey(i - 1) != x)) {
				const uint32_t slot = atomicAdd(&n_tied_s, 1u);
				if (slot < (uint32_t)TIE_MAX_KEYS) tied[slot] = x;
			}
		}
		__syncthreads();
		const uint32_t n_tied_all = n_tied_s;
#ifdef MM2AMD_WAVE_EMU
		if (tid == 0 && getenv("MM2AMD_TIE_TRACE")) fprintf(stderr, "[mm2amd] anchor sort: read %d, %d anchors, %u duplicated keys\n", r, (int)n, n_tied_all);
#endif
		if (n_tied_all == 0) return;
		if (tid == 0) B.tie_flag[r] = 1u;
		if (heap_sort & 1) return; // MM_F_HEAP_SORT: anchor_heap_order_kernel lays down the heap merge's order instead
		if (IN_LDS && (heap_sort & 4)) { // the few reads with duplicated keys are replayed together afterwards (anchor_sort_ties_kernel): no read holds this launch up
			if (tid == 0) B.tie_list[atomicAdd(B.tie_count, 1u)] = (uint32_t)r;
			return;
		}
		// 3. the reference's own permutation of the duplicated keys, from the original order
		const bool replay_all = n_tied_all > (uint32_t)TIE_MAX_KEYS;
		const int n_tied = replay_all ? TIE_MAX_KEYS : (int)n_tied_all;
		for (int32_t i = tid; i < n; i += THREADS) s.set(i, kin[i], (uint32_t)i);
		__syncthreads();
		tie_exact_replay<decltype(s), ROUNDS>(s, n, tied, n_tied, replay_all, cnt, head, start, child_mask, stack, stack_cap, IN_LDS && !(heap_sort & 2) ? tab + 768 : nullptr, IN_LDS ? ((THREADS / 64) * 256 - 768 - 20) * 2 : 0);
		for (int32_t i = tid; i < n; i += THREADS) {
			const uint64_t x = s.xkey(i);
			bool dup = replay_all;
			for (int t = 0; t < n_tied && !dup; ++t) dup = tied[t] == x;
			if (dup) { Anchor a; a.x = x, a.y = vin[s.index(i)]; out[i] = a; }
		}
	};
	if (IN_LDS) run(PackedStore{as_lds, B.rid_bits}, lstack, AS_STACK);
	else { // the sorted-pair arrays are free until the backtrack: keys, then indices and the frame stack in the value array
		uint32_t *I = (uint32_t *)(B.sort_val_out + ao);
		run(SplitStore{B.sort_key_out + ao, I, B.rid_bits}, (TieFrame *)(I + n), (int)((size_t)n * 4 / sizeof(TieFrame)));
	}
}

// The reads anchor_sort_kernel found duplicated keys in (B.tie_list; about one in 200 ONT reads against a random 3 Gb reference), replayed
// together: a workgroup per read with the tape scratch beside the read's elements, so a replay costs one LDS round trip per element of its
// sequential walk and the sorting launch does not wait for it.  (Measured before the split: the replays, sitting at random places in the
// sorting launches, made them 29 + 22 ms per step un-overlapped against 12 + 6 ms for the sorting alone.)
constexpr int AST_THREADS = 1024, AST_PER = AS_LDS_MAX / AST_THREADS;
constexpr size_t AST_LDS_BYTES = (size_t)AS_LDS_MAX * (8 + 2 + 2 + 1);
__global__ void __launch_bounds__(AST_THREADS, 4) anchor_sort_ties_kernel(SeedChainBuffers B, int mode)
{
	MM2_DYN_LDS(uint64_t, ast_lds); // the read's packed elements, then the tape arrays
	__shared__ uint32_t tab[768], tape_tab[768 + 32], child_mask[8];
	__shared__ uint64_t tied[TIE_MAX_KEYS];
	__shared__ TieFrame lstack[AS_STACK];
	__shared__ uint32_t n_tied_s;
	const int32_t tid = (int32_t)threadIdx.x;
	TapeScratch tape;
	tape.pos = (uint16_t *)(ast_lds + AS_LDS_MAX), tape.via = tape.pos + AS_LDS_MAX, tape.dig = (uint8_t *)(tape.via + AS_LDS_MAX), tape.tab = tape_tab, tape.cap = AS_LDS_MAX;
	// two workgroups per read, one per strand: the partition at the top is by the strand bit, what follows in one half does not depend on the
	// other, and a workgroup given only its strand's duplicated keys replays only that half (mode & 16: one workgroup does both)
	const bool split = !(mode & 16);
	const uint32_t n_list = *B.tie_count * (split ? 2u : 1u);
	for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
		const int r = (int)B.tie_list[split ? li >> 1 : li];
		const uint64_t strand = li & 1u;
		const uint64_t ao = B.a_off[r];
		const int32_t n = (int32_t)(B.a_off[r + 1] - ao);
		const uint64_t *kin = B.sort_key_in + ao, *vin = B.sort_val_in + ao;
		Anchor *out = B.anchors + ao;
		PackedStore s{ast_lds, B.rid_bits};
		__syncthreads(); // (the previous read's last pass over the shared arrays)
		if (tid == 0) n_tied_s = 0;
		__syncthreads();
		for (int32_t i = tid; i < n; i += AST_THREADS) { // the duplicated keys, from the sorted anchors
			const uint64_t x = out[i].x;
			if (i + 1 < n && out[i + 1].x == x && (i == 0 || out[i - 1].x != x) && (!split || x >> 63 == strand)) { // (the other workgroup rewrites y's only)
				const uint32_t slot = atomicAdd(&n_tied_s, 1u);
				if (slot < (uint32_t)TIE_MAX_KEYS) tied[slot] = x;
			}
			s.set(i, kin[i], (uint32_t)i); // the read in its original order
		}
		__syncthreads();
		const uint32_t n_tied_all = n_tied_s;
		if (n_tied_all == 0) continue; // nothing duplicated on this strand
		const bool replay_all = n_tied_all > (uint32_t)TIE_MAX_KEYS; // (then every bucket of both strands is replayed and rewritten: the same values the other workgroup writes)
		const int n_tied = replay_all ? TIE_MAX_KEYS : (int)n_tied_all;
		tie_exact_replay<PackedStore, AST_PER>(s, n, tied, n_tied, replay_all, tab, tab + 256, tab + 512, child_mask, lstack, AS_STACK,
		                                       (mode & 2) ? nullptr : (uint32_t *)tape.via, AS_LDS_MAX - 40, (mode & 8) ? TapeScratch() : tape);
		for (int32_t i = tid; i < n; i += AST_THREADS) {
			const uint64_t x = s.xkey(i);
			bool dup = replay_all;
			for (int t = 0; t < n_tied && !dup; ++t) dup = tied[t] == x;
			if (dup) { Anchor a; a.x = x, a.y = vin[s.index(i)]; out[i] = a; }
		}
	}
}

// MM_F_HEAP_SORT (collect_seed_hits_heap, map.c:102-166): reads that have two anchors with equal x get the order the reference's
// heap merge gives them (heap_order.hpp); one thread per such read, the minimizer arrays (dead after seed_collect) are the heap.
__global__ void __launch_bounds__(64) anchor_heap_order_kernel(SeedChainBuffers B, DevIndex I, SeedChainParams P)
{
	for (int r = blockIdx.x * 64 + threadIdx.x; r < B.n_reads; r += gridDim.x * 64) {
		if (!B.tie_flag[r]) continue;
		const uint64_t mo = B.mz_off[r], ao = B.a_off[r];
		const uint32_t *sd_n = B.sd_n + mo, *sd_off = B.sd_off + mo, *sd_aoff = B.sd_aoff + mo, *sd_qpos = B.sd_qpos + mo, *sd_info = B.sd_info + mo;
		const uint32_t n_m0 = B.n_seedhit[r];
		const int qlen = (int)(B.seq_off[r + 1] - B.seq_off[r]);
		const uint32_t n = (uint32_t)(B.a_off[r + 1] - ao);
		HIT_RULES_SETUP();
		Anchor *out = B.anchors + ao;
		uint32_t n_for = 0, n_rev = 0;
		heap_merge_order(n_m0, B.mz_x + mo, B.mz_y + mo,
			[&](uint32_t i, uint32_t *cnt) { *cnt = sd_aoff[i] == 0xffffffffu ? 0u : sd_n[i]; return I.pos + sd_off[i]; },
			[&](uint32_t i, uint64_t rr) {
				const uint32_t info = sd_info[i], span = info & 0xff, qp = sd_qpos[i], rpos = (uint32_t)rr >> 1;
				bool is_self = false;
				if (hit_rules && skip_hit(P.flag, rr, qp, qlen, nm_lb, nm_eq, I, &is_self)) return;
				Anchor p;
				if ((rr & 1) == (qp & 1)) {
					p.x = (rr & 0xffffffff00000000ULL) | rpos;
					p.y = (uint64_t)span << 32 | (uint64_t)(qp >> 1);
				} else {
					p.x = 1ULL << 63 | (rr & 0xffffffff00000000ULL) | rpos;
					p.y = (uint64_t)span << 32 | (uint64_t)(uint32_t)(qlen - ((int)(qp >> 1) + 1 - (int)span) - 1);
				}
				if (info & SD_SEG1) p.y |= 1ULL << ref::SEED_SEG_SHIFT;
				if (info & SD_TANDEM) p.y |= ref::SEED_TANDEM;
				if (is_self) p.y |= ref::SEED_SELF;
				if (p.x >> 63) { if (n_for + n_rev < n) out[n - (++n_rev)] = p; }
				else if (n_for + n_rev < n) out[n_for++] = p;
			});
		for (uint32_t j = 0; j < n_rev >> 1; ++j) { // the other-strand hits were laid down back to front (map.c:155-160)
			const Anchor t = out[n - 1 - j];
			out[n - 1 - j] = out[n - n_rev + j];
			out[n - n_rev + j] = t;
		}
	}
}

// the launch classes: reads in `list` are grouped by class, class c holds n_class[c] of them (backend: anchor_sort_class)
// anchors per read an LDS class holds (the last class sorts on global scratch with the comparison network) and the workgroup that sorts
// it (a 10 kb ONT read has ~7 k anchors against a 3 Gb reference: 56 KB of elements + 20 KB of tables, two workgroups of 1024 threads per CU)
const int kAnchorSortCap[kAnchorSortClasses] = { 1024, 2048, 4096, 7168, AS_LDS_MAX, 0 };
// workgroups: 256, 256, 512, 1024, 1024 threads (4, 8, 8, 7, 10 rounds of 64 elements per wave); 1024 for the global class

int anchor_sort_class(uint64_t n_anchors, int rid_bits)
{
	static const int force = getenv("MM2AMD_SORT_MIN_CLASS") ? atoi(getenv("MM2AMD_SORT_MIN_CLASS")) : 0; // tests: small reads through the large classes' workgroups
	if (33 + rid_bits + AS_IDX_BITS > 64) return kAnchorSortClasses - 1; // (more than 2^17 reference sequences: the key does not pack)
	if (force > 0) { for (int c = std::min(force, kAnchorSortClasses - 1); c + 1 < kAnchorSortClasses; ++c) if (n_anchors <= (uint64_t)kAnchorSortCap[c]) return c; return kAnchorSortClasses - 1; }
	for (int c = 0; c + 1 < kAnchorSortClasses; ++c) if (n_anchors <= (uint64_t)kAnchorSortCap[c]) return c;
	return kAnchorSortClasses - 1;
}

void launch_anchor_sort(const SeedChainBuffers &B, const DevIndex &I, const SeedChainParams &P, const uint32_t *d_list, const int *n_class, const double *anchors_in_class, void *stream,
                        KernelProfiler *kp)
{
	hipStream_t s = (hipStream_t)stream;
	KernelProfiler none;
	if (!kp) kp = &none;
	HIP_CHECK(hipMemsetAsync(B.tie_flag, 0, (size_t)B.n_reads * 4, s));
	static const char *kNames[kAnchorSortClasses] = { "anchor_sort_kernel[n1k]", "anchor_sort_kernel[n2k]", "anchor_sort_kernel[n4k]", "anchor_sort_kernel[n7k]", "anchor_sort_kernel[n10k]", "anchor_sort_kernel[global]" };
	// (function attributes are per device and lane drivers of several replicas call this concurrently: set it before every launch that needs it,
	// as ksw_extd2.hip does -- a table write in the runtime, microseconds)
	if (n_class[4] > 0) HIP_CHECK(hipFuncSetAttribute((const void *)anchor_sort_kernel<1024, 10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, AS_LDS_MAX * 8));
	static const bool no_replay = getenv("MM2AMD_SORT_NO_REPLAY") != nullptr; // TIMING ONLY (tools/r03_call5.sh): reads with duplicated keys keep the sorted order -- not the reference's
	const bool walk_only = getenv("MM2AMD_NO_TWO_BUCKET") != nullptr; // A/B checks: every partition of the replay by the sequential walk (read per launch)
	static const bool replay_inline = getenv("MM2AMD_TIE_REPLAY_INLINE") != nullptr; // A/B checks: the replay inside the sorting launch, as before round 5
	const bool no_tape = getenv("MM2AMD_NO_TAPE_WALK") != nullptr;                   // A/B checks: the replay's partitions by the one-thread walk
	const bool no_split = getenv("MM2AMD_TIE_NO_STRAND_SPLIT") != nullptr;           // A/B checks: one workgroup per read replays both strands' halves
	const int heap = ((P.flag & ref::F_HEAP_SORT) || no_replay ? 1 : 0) | (walk_only ? 2 : 0) | (replay_inline ? 0 : 4) | (no_tape ? 8 : 0) | (no_split ? 16 : 0);
	if (heap & 4) HIP_CHECK(hipMemsetAsync(B.tie_count, 0, 4, s));
	int first = 0;
	for (int c = 0; c < kAnchorSortClasses; first += n_class[c], ++c) {
		if (n_class[c] == 0) continue;
		kp->begin(s);
		const size_t lds = (size_t)kAnchorSortCap[c] * 8;
		const dim3 grid(n_class[c]);
		if (c == 0) hipLaunchKernelGGL((anchor_sort_kernel<256, 4, true>), grid, dim3(256), lds, s, B, d_list + first, heap);
		else if (c == 1) hipLaunchKernelGGL((anchor_sort_kernel<256, 8, true>), grid, dim3(256), lds, s, B, d_list + first, heap);
		else if (c == 2) hipLaunchKernelGGL((anchor_sort_kernel<512, 8, true>), grid, dim3(512), lds, s, B, d_list + first, heap);
		else if (c == 3) hipLaunchKernelGGL((anchor_sort_kernel<1024, 7, true>), grid, dim3(1024), lds, s, B, d_list + first, heap);
		else if (c == 4) hipLaunchKernelGGL((anchor_sort_kernel<1024, 10, true>), grid, dim3(1024), lds, s, B, d_list + first, heap);
		else hipLaunchKernelGGL((anchor_sort_kernel<1024, 1, false>), grid, dim3(1024), 0, s, B, d_list + first, heap);
		kp->end(s, kNames[c], 32.0 * anchors_in_class[c]); // 16 B per anchor in, 16 B out (nothing else leaves LDS)
		HIP_CHECK(hipGetLastError());
	}
	const int n_lds = n_class[0] + n_class[1] + n_class[2] + n_class[3] + n_class[4];
	if ((heap & 4) && !(heap & 1) && n_lds > 0) {
		kp->begin(s);
		HIP_CHECK(hipFuncSetAttribute((const void *)anchor_sort_ties_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AST_LDS_BYTES));
		hipLaunchKernelGGL(anchor_sort_ties_kernel, dim3(std::min(2 * n_lds, 512)), dim3(AST_THREADS), AST_LDS_BYTES, s, B, heap);
		kp->end(s, "anchor_sort_kernel[ties]", 0.0);
		HIP_CHECK(hipGetLastError());
	}
	if (P.flag & ref::F_HEAP_SORT) { // equal-x order of the heap merge instead of the radix sort's
		kp->begin(s);
		hipLaunchKernelGGL(anchor_heap_order_kernel, dim3(std::min((B.n_reads + 63) / 64, 4096)), dim3(64), 0, s, B, I, P);
		kp->end(s, "anchor_heap_order_kernel", 0.0);
		HIP_CHECK(hipGetLastError());
	}
}

// ---------------------------------------------------------------------------------------------------------
// Chaining DP fill (lchain.c:169-207), one wavefront per read
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_log2_dev(float x) // mg_log2, mmpriv.h:139-147
{
	uint32_t zi = __float_as_uint(x);
	float l = (float)((int)((zi >> 23) & 255) - 128);
	zi &= ~(255u << 23);
	zi += 127u << 23;
	const float zf = __uint_as_float(zi);
	l += (-0.34484843f * zf + 2.02466578f) * zf - 0.67487759f;
	return l;
}

// comput_sc (lchain.c:113-138).  n_seg = 2 for a read pair: its anchors carry the segment (read) they come from, the distance
// and bandwidth limits only hold between anchors of one read, and a jump from one read to its mate is charged like a deletion.
template <bool PAIRS>
__device__ __forceinline__ int32_t link_score(uint64_t ix, uint64_t iy, uint64_t jx, uint64_t jy, int32_t max_dist_x, int32_t max_dist_y, int32_t bw,
                                              float pen_gap, float pen_skip, int is_cdna, int n_seg)
{
	const int32_t dq = (int32_t)iy - (int32_t)jy;
	const bool same = PAIRS ? ((iy ^ jy) & ref::SEED_SEG_MASK) == 0 : true; // batches without pairs compile to the single-segment rules
	if (dq <= 0 || dq > max_dist_x) return INT32_MIN;
	const int32_t dr = (int32_t)(ix - jx);
	if (same && (dr == 0 || dq > max_dist_y)) return INT32_MIN;
	const int32_t dd = dr > dq ? dr - dq : dq - dr;
	if (same && dd > bw) return INT32_MIN;
	if (n_seg > 1 && !is_cdna && same && dr > max_dist_y) return INT32_MIN;
	const int32_t dg = dr < dq ? dr : dq, span = (int32_t)(jy >> 32 & 0xff);
	int32_t sc = span < dg ? span : dg;
	if (dd || dg > span) {
		const float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		const float lg = dd >= 1 ? fast_log2_dev((float)(dd + 1)) : 0.0f;
		if (is_cdna || !same) {
			if (!same && dr == 0) ++sc; // overlapping mates
			else if (dr > dq || !same) sc -= (int)(lin < lg ? lin : lg);
			else sc -= (int)(lin + .5f * lg);
		} else sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

// The DP is sequential in i, but only inside a CLUSTER of anchors.  Anchor i is "isolated" when its predecessor a[i-1] is
// already outside its look-back window (another target/strand, or more than max_dist_x upstream): the anchors are sorted, so
// then the whole window is empty, f[i] = q_span, p[i] = -1, and the loop state after i (window start st = i, best-scoring
// anchor max_ii = i) does not depend on anything before it.  With a 3 Gb reference ~85 % of a read's anchors are such random
// isolated hits.  Isolation is a purely local test, so each block of 64 anchors settles its isolated members in parallel and
// only the members of real clusters (the true chains) go through the sequential rules, restarting from that known state at
// every cluster head.
// The look-back window lives in LDS: per wavefront a ring of the last CF_RING anchors -- x, y, chain score f, predecessor p and the
// "already reached through a better predecessor" mark t of lchain.c:186 -- filled as the blocks of 64 anchors go by.  An anchor
// older than the ring (a look-back of more than ~200 anchors: tandem repeats, very dense windows) is read from the global arrays
// as before; where an index lives is a function of the index and the current block only, so marks and scores are never split
// between the two.  RING = false keeps everything in global memory (A/B checks: MM2AMD_CHAIN_FILL_GLOBAL=1).
constexpr int CF_RING = 256;

// A read with very many anchors (one that crosses a multi-copy element of the reference brings a hundred times the usual number) is cut into PIECES
// worked on by wavefronts of their own.  The cut points are isolated anchors -- cluster heads, after which the sequential rules restart from a known
// state (above for mg_lchain_dp; mg_lchain_rmq's trees are empty there) -- so every piece computes exactly what the one walk over the read would:
// piece k of a read covers [first isolated anchor at or after k * len, first isolated anchor at or after (k + 1) * len), piece 0 starts at 0, the
// last one ends at n.  A cluster longer than a piece leaves the pieces it spans empty.  The test is the kernels' own (the look-back window of
// anchor g is empty when a[g-1] is on another target/strand or more than max_dist upstream).
__device__ __forceinline__ int64_t chain_first_head(const Anchor *a, int64_t n, int64_t from, int32_t max_dist, int lane)
{
	for (int64_t base = from; base < n; base += 64) {
		const int64_t g = base + lane;
		bool head = false;
		if (g < n) { const uint64_t gx = a[g].x, px = a[g - 1].x; head = gx >> 32 != px >> 32 || gx > px + (uint64_t)(int64_t)max_dist; } // (from >= 1)
		const unsigned long long m = __ballot(head);
		if (m) return base + (__ffsll((long long)m) - 1);
	}
	return n;
}

__device__ __forceinline__ void chain_piece_bounds(const Anchor *a, int64_t n, int64_t k, int64_t len, int32_t max_dist, int lane, int64_t *lo, int64_t *hi)
{
	*lo = k == 0 ? 0 : (k * len < n ? chain_first_head(a, n, k * len, max_dist, lane) : n);
	*hi = (k + 1) * len < n ? chain_first_head(a, n, (k + 1) * len, max_dist, lane) : n;
}

template <bool PAIRS, int RING> // RING: entries of the LDS window per wavefront (CF_RING; 128: 14 KB per workgroup, eight workgroups per CU instead of five -- measured), 0 = none
__global__ void __launch_bounds__(256) chain_fill_kernel(SeedChainBuffers B, SeedChainParams P)
{
	constexpr int RN = RING ? RING : 1, RM = RN - 1;
	__shared__ uint64_t s_x[4][RN], s_y[4][RN];
	__shared__ int32_t s_f[4][RN], s_p[4][RN], s_t[4][RN];
	const int wave = threadIdx.x >> 6, lane = lane_id();
	const int w = blockIdx.x * 4 + wave;
	// a wavefront's work: a read -- or, when the host listed pieces (a read with very many anchors: its clusters are independent, see chain_piece_bounds), one piece of a read
	if (w >= (B.pieces ? B.n_pieces : B.n_reads)) return;
	const int r = B.pieces ? (int)B.pieces[2 * w] : w;
	uint64_t *const rx = s_x[wave], *const ry = s_y[wave];
	int32_t *const rf = s_f[wave], *const rp = s_p[wave], *const rt = s_t[wave];
	const Anchor *a = B.anchors + B.a_off[r];
	const int64_t n_all = (int64_t)(B.a_off[r + 1] - B.a_off[r]);
	int32_t *f = B.f + B.a_off[r], *p = B.p + B.a_off[r], *t = B.t + B.a_off[r];
	int32_t max_dist_x, max_dist_y;
	chain_gaps(P, (int)(B.seq_off[r + 1] - B.seq_off[r]), &max_dist_x, &max_dist_y);
	max_dist_x = __builtin_amdgcn_readfirstlane(max_dist_x), max_dist_y = __builtin_amdgcn_readfirstlane(max_dist_y); // one read per wavefront: keep the limits in scalar registers, as when they were launch constants
	const int n_seg = PAIRS ? B.unit_first[r + 1] - B.unit_first[r] : 1;
	const int32_t bw = P.bw;
	if (max_dist_x < bw) max_dist_x = bw;
	if (max_dist_y < bw && !P.is_cdna) max_dist_y = bw;
	int64_t lo = 0, n = n_all; // the anchors [lo, n) of the read are this wavefront's
	if (B.pieces) chain_piece_bounds(a, n_all, (int64_t)B.pieces[2 * w + 1], (int64_t)B.piece_len, max_dist_x, lane, &lo, &n);
	if (lo >= n) return;
	for (int64_t i = lo + lane; i < n; i += 64) t[i] = 0;
	__threadfence_block();

	int64_t st = lo, max_ii = -1;
	uint64_t mii_x = 0, mii_y = 0;   // a[max_ii]
	int32_t mii_f = 0;               // f[max_ii]
	uint64_t last_x = 0, last_y = 0; // the anchor just before the current block
	bool last_iso = false;           // ... and whether it was isolated
	for (int64_t blk = lo; blk < n; blk += 64) {
		const int64_t g = blk + lane;
		// indices >= ring_lo are in the ring while this block is worked on (the block itself has just entered it)
		const int64_t ring_lo = RING ? (blk + 64 - lo > RN ? blk + 64 - RN : lo) : INT64_MAX;
		auto ax = [&](int64_t j) { return j >= ring_lo ? rx[j & RM] : a[j].x; };
		auto ay = [&](int64_t j) { return j >= ring_lo ? ry[j & RM] : a[j].y; };
		auto af = [&](int64_t j) { return j >= ring_lo ? rf[j & RM] : f[j]; };
		auto ap = [&](int64_t j) { return j >= ring_lo ? rp[j & RM] : p[j]; };
		uint64_t bx = 0, by = 0;
		if (g < n) { const Anchor v = a[g]; bx = v.x, by = v.y; }
		const uint64_t px = wave_shr1_u64(last_x, bx), py = wave_shr1_u64(last_y, by); // a[g-1]
		const bool iso = g < n && (g == lo || (bx >> 32 != px >> 32 || bx > px + (uint64_t)(int64_t)max_dist_x)); // (a piece starts at an isolated anchor)
		if (iso) f[g] = (int32_t)(by >> 32 & 0xff), p[g] = -1;
		if (RING && g < n) {
			rx[g & RM] = bx, ry[g & RM] = by, rt[g & RM] = -1;
			if (iso) rf[g & RM] = (int32_t)(by >> 32 & 0xff), rp[g & RM] = -1;
		}
		__threadfence_block(); // cluster members read their head's f through memory
		unsigned long long todo = __ballot(g < n && !iso);
		const unsigned long long iso_mask = __ballot(iso);
		while (todo) {
			const int bl = __ffsll((long long)todo) - 1;
			todo &= todo - 1;
			const int64_t i = blk + bl;
			const uint64_t ix = lane_get_u64(bx, bl), iy = lane_get_u64(by, bl);
			const bool head_before = bl == 0 ? last_iso : (iso_mask >> (bl - 1) & 1) != 0; // a[i-1] isolated: a cluster starts here
			if (head_before) { // the state the sequential loop is in right after an isolated anchor (see above)
				const uint64_t hx = lane_get_u64(px, bl), hy = lane_get_u64(py, bl);
				st = i - 1, max_ii = i - 1, mii_x = hx, mii_y = hy, mii_f = (int32_t)(hy >> 32 & 0xff);
			}
			// advance the window start (lchain.c:172): first st in [st,i) on the same target/strand within max_dist_x
			while (st < i) {
				const int64_t c = st + lane;
				bool stop = true; // lanes past i stop the scan
				if (c < i) { const uint64_t cx = ax(c); stop = !(ix >> 32 != cx >> 32 || ix > cx + (uint64_t)(int64_t)max_dist_x); }
				const unsigned long long m = __ballot(stop);
				if (m) { st += __ffsll((long long)m) - 1; break; }
				st += 64;
			}
			if (st > i) st = i;
			if (i - st > P.max_chain_iter) st = i - P.max_chain_iter;

			int32_t max_f = (int32_t)(iy >> 32 & 0xff), n_skip = 0;
			int64_t max_j = -1, end_j = st - 1;
			bool broke = false;
			for (int64_t base = i - 1; base >= st && !broke; base -= 64) {
				const int64_t j = base - lane;
				int32_t sc = INT32_MIN, pj = -1;
				if (j >= st) {
					sc = link_score<PAIRS>(ix, iy, ax(j), ay(j), max_dist_x, max_dist_y, bw, P.chn_pen_gap, P.chn_pen_skip, P.is_cdna, n_seg);
					if (sc != INT32_MIN) sc += af(j), pj = ap(j);
				}
				const bool has = sc != INT32_MIN;
				// exclusive prefix maximum in processing order (lane 0 first)
				int32_t excl = wave_shr1_i32(INT32_MIN, wave_prefix_max_i32(has ? sc : INT32_MIN));
				excl = excl > max_f ? excl : max_f;
				const bool improve = has && sc > excl;
				// marks left by predecessors examined earlier in this iteration (lchain.c:186)
				if (has && pj >= 0) { if (pj >= ring_lo) rt[pj & RM] = (int32_t)i; else t[pj] = (int32_t)i; }
				__threadfence_block();
				const int64_t jm = j >= st ? j : st;
				const bool marked = has && !improve && (jm >= ring_lo ? rt[jm & RM] : t[jm]) == (int32_t)i;
				unsigned long long imp = __ballot(improve);
				const unsigned long long mk = __ballot(marked);
				// The skip counter (lchain.c:181-185) walks the candidates in order: +1 for one already reached through a better predecessor,
				// -1 (not below 0) for an improvement, stop when it exceeds max_skip.  Improvements are few, so the walk goes from one
				// improvement to the next and counts the marked candidates between them with a population count.
				int stop_lane = 64;
				for (int pos = 0; pos < 64;) {
					const unsigned long long later = imp >> pos;
					const int ni = later ? pos + (__ffsll((long long)later) - 1) : 64;               // the next improvement at or after pos
					unsigned long long seg = (mk >> pos) << pos;                                       // marked candidates in [pos, ni)
					if (ni < 64) seg &= (1ull << ni) - 1ull;
					const int c = __popcll(seg);
					if (n_skip + c > P.max_chain_skip) { // the counter overflows inside this stretch: at its (max_skip - n_skip + 1)-th marked candidate
						for (int k = P.max_chain_skip - n_skip; k > 0; --k) seg &= seg - 1;
						stop_lane = __ffsll((long long)seg) - 1;
						break;
					}
					n_skip += c;
					if (ni == 64) break;
					if (n_skip > 0) --n_skip;
					pos = ni + 1;
				}
				if (stop_lane < 64) {
					broke = true;
					end_j = base - stop_lane;
					imp &= (1ull << stop_lane) - 1ull;
				}
				if (imp) { // improvements are increasing, so the last one before the stop holds the running maximum
					const int last = 63 - __clzll((long long)imp);
					max_f = lane_get_i32(sc, last);
					max_j = base - last;
				}
			}
			// the best-scoring anchor in range may lie beyond the early exit (lchain.c:189-200)
			bool recompute = max_ii < 0;
			if (!recompute) recompute = ix - mii_x > (uint64_t)(int64_t)max_dist_x;
			if (recompute) {
				long long best = INT64_MIN; // (f, j): larger f first, then larger j
				for (int64_t base = i - 1; base >= st; base -= 64) {
					const int64_t j = base - lane;
					if (j >= st) { const long long key = (long long)af(j) << 32 | (long long)(uint32_t)j; best = key > best ? key : best; }
				}
				for (int o = 32; o > 0; o >>= 1) { const long long v = __shfl_xor(best, o, 64); best = v > best ? v : best; }
				max_ii = best == INT64_MIN ? -1 : (int64_t)(uint32_t)(best & 0xffffffffLL);
				if (max_ii >= 0) mii_x = ax(max_ii), mii_y = ay(max_ii), mii_f = (int32_t)(best >> 32);
			}
			if (max_ii >= 0 && max_ii < end_j) {
				const int32_t tmp = link_score<PAIRS>(ix, iy, mii_x, mii_y, max_dist_x, max_dist_y, bw, P.chn_pen_gap, P.chn_pen_skip, P.is_cdna, n_seg);
				if (tmp != INT32_MIN && max_f < tmp + mii_f) max_f = tmp + mii_f, max_j = max_ii;
			}
			// The member's result: with the ring it stays in LDS until the block is done (a global store here would have to be waited
			// for before the next member reads it back through the fence: microseconds per anchor of a chain)
			if (RING) { if (lane == 0) rf[i & RM] = max_f, rp[i & RM] = (int32_t)max_j; WAVE_SYNC(); }
			else { if (lane == 0) f[i] = max_f, p[i] = (int32_t)max_j; __threadfence_block(); }
			if (max_ii < 0 || (ix - mii_x <= (uint64_t)(int64_t)max_dist_x && mii_f < max_f)) max_ii = i, mii_x = ix, mii_y = iy, mii_f = max_f;
		}
		if (RING) { // the block's chain scores and predecessors go to the global arrays in one coalesced store (isolated anchors have theirs already)
			if (g < n && !iso) f[g] = rf[g & RM], p[g] = rp[g & RM];
			__threadfence_block(); // a later block reads them from there once they have left the ring
		}
		// carry the block's last anchor (and whether it was isolated) into the next block
		const int last_lane = (int)((n - blk < 64 ? n - blk : 64) - 1);
		last_x = lane_get_u64(bx, last_lane), last_y = lane_get_u64(by, last_lane);
		last_iso = (iso_mask >> last_lane & 1) != 0;
	}
}

void launch_chain_fill(const SeedChainBuffers &B, const SeedChainParams &P, void *stream)
{
	const char *ring_env = getenv("MM2AMD_CHAIN_RING"); // A/B: 0 = no LDS window (MM2AMD_CHAIN_FILL_GLOBAL=1 is the older name), 128 = the smaller one
	const int ring = getenv("MM2AMD_CHAIN_FILL_GLOBAL") ? 0 : ring_env ? atoi(ring_env) : CF_RING;
	const dim3 grid(((B.pieces ? B.n_pieces : B.n_reads) + 3) / 4), block(256);
	hipStream_t s = (hipStream_t)stream;
	if (B.unit_first) {
		if (ring == 0) hipLaunchKernelGGL((chain_fill_kernel<true, 0>), grid, block, 0, s, B, P);
		else if (ring == 128) hipLaunchKernelGGL((chain_fill_kernel<true, 128>), grid, block, 0, s, B, P);
		else hipLaunchKernelGGL((chain_fill_kernel<true, CF_RING>), grid, block, 0, s, B, P);
	} else {
		if (ring == 0) hipLaunchKernelGGL((chain_fill_kernel<false, 0>), grid, block, 0, s, B, P);
		else if (ring == 128) hipLaunchKernelGGL((chain_fill_kernel<false, 128>), grid, block, 0, s, B, P);
		else hipLaunchKernelGGL((chain_fill_kernel<false, CF_RING>), grid, block, 0, s, B, P);
	}
	HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// RMQ chaining fill (mg_lchain_rmq, lchain.c:250-368): the primary chainer of the MM_F_RMQ presets (asm5/10/20, lr:hqae), one
// wavefront per read.
//
// The reference scores anchor i against candidates it keeps in two balanced trees over the query coordinate.  Both trees hold
// exactly the anchors of a sliding INDEX window [st, i0) of the target-sorted array (insertion at i0, removal at st, lchain.c:285-
// 318), so no tree is needed to reproduce what they answer:
//   * the range-minimum look-up (:320-322) = the smallest priority among the window's anchors with query coordinate in
//     (y_i - max_dist, y_i): all 64 lanes scan the window and reduce.  The priority of an anchor, -(f + 0.5 * pen_gap * (x + y)) in
//     double precision, is fixed once its score is, so it is computed when the anchor enters the window.  If TWO anchors share the
//     smallest priority the reference's answer depends on the history of its tree (rmq_chain.cpp): the kernel then gives the read up
//     and flags it, and the host chains that read with the tie-exact tree (rare: equal chain scores on the same anti-diagonal);
//   * the scan of the close neighbourhood (:328-354) visits the narrow window's anchors in descending (query coordinate, index)
//     order: they are gathered into LDS, sorted there by the wave (bitonic, padded to a power of two), and scored 64 at a time with
//     the skip rule resolved from ballots exactly as in chain_fill_kernel (an anchor is only ever marked by candidates visited
//     before it: its chain successors have larger query coordinates).
// ---------------------------------------------------------------------------------------------------------
constexpr int RMQ_NEAR_CAP = 4096; // anchors of the narrow window scored per anchor; more: the read goes to the host
constexpr int RMQ_RANK_MAX = 256;   // neighbourhoods up to this many anchors are sorted by counting (SeedChainBuffers::rmq_rank_max <= this: 0 = always the bitonic network, A/B)
constexpr int RMQ_NEAR_SMALL = 1024; // ... of the first launch; what does not fit goes to a second one (MM2AMD_RMQ_NEAR_TINY=1: 64, for tests of that hand-over)

__device__ __forceinline__ int32_t simple_score_dev(uint64_t ix, uint64_t iy, uint64_t jx, uint64_t jy, float pen_gap, float pen_skip, bool *exact, int32_t *width) // comput_sc_simple, lchain.c:229-248
{
	const int32_t dq = (int32_t)iy - (int32_t)jy, dr = (int32_t)(ix - jx);
	const int32_t dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq, span = (int32_t)(jy >> 32 & 0xff);
	int32_t sc = span < dg ? span : dg;
	*width = dd;
	*exact = dd == 0 && dg <= span;
	if (dd || dq > span) {
		const float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		const float lg = dd >= 1 ? fast_log2_dev((float)(dd + 1)) : 0.0f;
		sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

// Launched twice: NEAR = RMQ_NEAR_SMALL for every read (8 KB of LDS per wavefront instead of 32: four times the wavefronts per CU, and the kernel is a chain of
// dependent look-ups), then NEAR = RMQ_NEAR_CAP for the reads whose neighbourhood did not fit (flag bit 1).  tie_flag[r], zeroed by the launcher: bit 0 = the
// host chains this read, bit 1 = the second launch does; chain_backtrack_kernel leaves bit 0 only.  With pieces (chain_piece_bounds) a read's wavefronts
// share its flag; the second launch redoes every piece of a flagged read (pieces are independent, their results do not depend on who computes them).
template <int NEAR>
__global__ void __launch_bounds__(64) chain_rmq_kernel(SeedChainBuffers B, SeedChainParams P, uint64_t *timing)
{
	__shared__ uint64_t s_near[NEAR];
	__shared__ uint64_t s_rank[RMQ_RANK_MAX];
	const int lane = threadIdx.x, w = blockIdx.x;
	const uint64_t t_begin = timing ? wall_clock64() : 0; // (MM2AMD_RMQ_TIMING=1: what each wavefront of a launch spent, 100 MHz ticks)
	const int r = B.pieces ? (int)B.pieces[2 * w] : w;
	const Anchor *a = B.anchors + B.a_off[r];
	const int64_t n_all = (int64_t)(B.a_off[r + 1] - B.a_off[r]);
	if (NEAR == RMQ_NEAR_CAP) { if ((atomicOr(&B.tie_flag[r], 0u) & 3u) != 2u) return; } // the second launch: only what the first one left to it (a sibling piece may have given the read up since)
	int32_t *f = B.f + B.a_off[r], *p = B.p + B.a_off[r], *t = B.t + B.a_off[r];
	double *pri = (double *)(B.sort_key_out + B.a_off[r]); // dead since the anchor sort; the backtrack reuses it afterwards
	int32_t max_dist = P.max_gap, max_dist_inner = P.rmq_inner_dist;
	const int32_t bw = P.bw, cap = P.rmq_size_cap;
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner < 0) max_dist_inner = 0;
	if (max_dist_inner > max_dist) max_dist_inner = max_dist;
	int64_t lo = 0, n = n_all; // the anchors [lo, n) of the read are this wavefront's
	if (B.pieces) chain_piece_bounds(a, n_all, (int64_t)B.pieces[2 * w + 1], (int64_t)B.piece_len, max_dist, lane, &lo, &n);
	if (lo >= n) return;
	if (B.pieces && B.piece_dense > 0 && n - lo >= (int64_t)B.piece_dense) return; // a long cluster: chain_rmq_wide_kernel's
	for (int64_t i = lo + lane; i < n; i += 64) t[i] = 0;
	__threadfence_block();
	int64_t i0 = lo, st = lo, st_in = lo;
	bool give_up = n_all > (int64_t)P.rmq_dev_max_anchors; // (whole contigs: backend.hpp)
	bool overflow = false;
	for (int64_t i = lo; i < n && !give_up; ++i) {
		const Anchor ai = a[i];
		const uint64_t ix = ai.x, iy = ai.y;
		const int32_t y_i = (int32_t)iy;
		if (i0 < i && a[i0].x != ix) { // the anchors with a smaller target coordinate become candidates (:285-298)
			for (int64_t j = i0 + lane; j < i; j += 64) pri[j] = -((double)f[j] + 0.5 * (double)P.chn_pen_gap * (double)((int32_t)a[j].x + (int32_t)a[j].y));
			i0 = i;
			__threadfence_block();
		}
		// candidates out of reach leave in index order (:300-318): "out of reach" is monotone in the index, the size cap a plain bound
		auto advance = [&](int64_t s0, int32_t dist) {
			int64_t s = s0;
			while (s < i) {
				const int64_t c = s + lane;
				bool stop = true;
				if (c < i) { const uint64_t cx = a[c].x; stop = !(ix >> 32 != cx >> 32 || ix > cx + (uint64_t)(int64_t)dist); }
				const unsigned long long m = __ballot(stop);
				if (m) { s += __ffsll((long long)m) - 1; break; }
				s += 64;
			}
			if (s > i) s = i;
			if (i0 - s > (int64_t)cap) s = i0 - cap;
			return s;
		};
		st = advance(st, max_dist);
		if (max_dist_inner > 0) st_in = advance(st_in, max_dist_inner);
		// ---- range minimum over [st, i0) ----
		double best = 0;
		int64_t best_j = -1;
		int n_best = 0;
		for (int64_t base = st; base < i0; base += 64) {
			const int64_t j = base + lane;
			if (j < i0) {
				const int32_t y_j = (int32_t)a[j].y;
				if ((y_j > y_i - max_dist && y_j < y_i) || (y_j == y_i && j == 0)) { // keys (y_i - max_dist, INT32_MAX) .. (y_i, 0), closed (:320-321)
					const double v = pri[j];
					if (best_j < 0 || v < best) best = v, best_j = j, n_best = 1;
					else if (v == best) ++n_best;
				}
			}
		}
		for (int o = 32; o > 0; o >>= 1) { // combine (value, count): equal minima add up
			const double ov = __shfl_xor(best, o, 64);
			const int64_t oj = __shfl_xor(best_j, o, 64);
			const int on = __shfl_xor(n_best, o, 64);
			if (oj >= 0) {
				if (best_j < 0 || ov < best) best = ov, best_j = oj, n_best = on;
				else if (ov == best) n_best += on;
			}
		}
		int32_t max_f = (int32_t)(iy >> 32 & 0xff);
		int64_t max_j = -1;
		if (best_j >= 0) {
			if (n_best > 1) { give_up = true; break; } // the reference's pick depends on its tree's history: the host replays it
			bool exact;
			int32_t width;
			const Anchor aj = a[best_j];
			int32_t sc = f[best_j] + simple_score_dev(ix, iy, aj.x, aj.y, P.chn_pen_gap, P.chn_pen_skip, &exact, &width);
			if (width <= bw && sc > max_f) max_f = sc, max_j = best_j;
			if (!exact && max_dist_inner > 0 && st_in < i0 && y_i > 0) {
				// ---- the close neighbourhood, nearest query coordinate first ----
				int n_c = 0;
				for (int64_t base = st_in; base < i0; base += 64) {
					const int64_t j = base + lane;
					bool in = false;
					int32_t y_j = 0;
					if (j < i0) { y_j = (int32_t)a[j].y; in = y_j <= y_i - 1 && y_j >= y_i - max_dist_inner; }
					const unsigned long long m = __ballot(in);
					if (in) { const int d = n_c + popc_below(m, lane); if (d < NEAR) s_near[d] = (uint64_t)(uint32_t)y_j << 32 | (uint64_t)(uint32_t)j; }
					n_c += __popcll(m);
				}
				if (n_c > NEAR) { if (NEAR == RMQ_NEAR_CAP) give_up = true; else overflow = true; break; }
				const uint64_t *sorted = s_near;
				if (n_c <= B.rmq_rank_max) { // a small neighbourhood (the usual one): every key's place is the number of larger keys (they are distinct) -- one pass over LDS instead of the bitonic network's dozens
					__syncthreads();
					for (int e = lane; e < n_c; e += 64) {
						const uint64_t key = s_near[e];
						int rank = 0;
						for (int k = 0; k < n_c; ++k) rank += s_near[k] > key;
						s_rank[rank] = key;
					}
					sorted = s_rank;
					__syncthreads();
				} else {
				int n_pad = 64;
				while (n_pad < n_c) n_pad <<= 1;
				for (int k = n_c + lane; k < n_pad; k += 64) s_near[k] = 0; // pads sort last (descending order; y >= 0 and every real key is > 0 unless (0, 0), which ties harmlessly)
				__syncthreads();
				for (int k2 = 2; k2 <= n_pad; k2 <<= 1)
					for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
						for (int e = lane; e < n_pad; e += 64) {
							const int partner = e ^ j2;
							if (partner > e) {
								const uint64_t u0 = s_near[e], u1 = s_near[partner];
								const bool desc = (e & k2) == 0; // descending runs first
								if (desc ? u0 < u1 : u0 > u1) s_near[e] = u1, s_near[partner] = u0;
							}
						}
						__syncthreads();
					}
				}
				int32_t n_skip = 0;
				bool broke = false;
				for (int base = 0; base < n_c && !broke; base += 64) {
					const int c = base + lane;
					int32_t scc = INT32_MIN, pj = -1;
					int64_t j = -1;
					if (c < n_c) {
						j = (int64_t)(uint32_t)sorted[c];
						bool ex;
						int32_t wd;
						const Anchor cj = a[j];
						const int32_t v = simple_score_dev(ix, iy, cj.x, cj.y, P.chn_pen_gap, P.chn_pen_skip, &ex, &wd);
						if (wd <= bw) scc = f[j] + v, pj = p[j];
					}
					const bool has = scc != INT32_MIN;
					int32_t pm = has ? scc : INT32_MIN; // exclusive prefix maximum in visiting order
					for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(pm, o, 64); if (lane >= o) pm = v > pm ? v : pm; }
					int32_t excl = __shfl_up(pm, 1, 64);
					if (lane == 0) excl = INT32_MIN;
					excl = excl > max_f ? excl : max_f;
					const bool improve = has && scc > excl;
					if (has && pj >= 0) t[pj] = (int32_t)i; // marks left by candidates visited earlier (:349)
					__threadfence_block();
					const bool marked = has && !improve && t[j] == (int32_t)i;
					unsigned long long imp = __ballot(improve), mk = __ballot(marked), ev = imp | mk;
					int stop_lane = 64;
					while (ev) {
						const int b = __ffsll((long long)ev) - 1;
						ev &= ev - 1;
						if (imp >> b & 1) { if (n_skip > 0) --n_skip; }
						else if (++n_skip > P.max_chain_skip) { stop_lane = b; break; }
					}
					if (stop_lane < 64) broke = true, imp &= (1ull << stop_lane) - 1ull;
					if (imp) {
						const int last = 63 - __clzll((long long)imp);
						max_f = __shfl(scc, last, 64);
						max_j = __shfl(j, last, 64);
					}
				}
				__syncthreads();
			}
		}
		if (lane == 0) f[i] = max_f, p[i] = (int32_t)max_j;
		__threadfence_block();
	}
	if (timing && lane == 0 && (int)w < (1 << 20)) timing[2 * w] = wall_clock64() - t_begin, timing[2 * w + 1] = (uint64_t)(n - lo) << 32 | (uint32_t)r;
	if (lane == 0 && (give_up || overflow)) atomicOr(&B.tie_flag[r], give_up ? 1u : 2u); // reused: bit 0 = the host chains this read (rmq_chain.cpp); the backtrack leaves such a read empty
}

// The same rules for a LONG CLUSTER, by a whole workgroup.  A piece that spans many anchors without a cluster head in it (the read's main chain: a thousand anchors; a tandem
// array: thousands at one locus) is where a wavefront's time goes: every anchor scans a window of hundreds to thousands of candidates for the range minimum and sorts
// a neighbourhood of up to RMQ_NEAR_CAP -- tens of microseconds per anchor, and the launch waited 100-280 ms for one such wavefront (MM2AMD_RMQ_TIMING).
// The walk over the anchors stays sequential; what is parallel inside it is spread over THREADS lanes: the window scan (per-wavefront minima combined through LDS, equal
// minima adding up as in the one-wavefront form), the gathering of the neighbourhood (slots handed out by an LDS counter: the sort that follows makes the order of arrival
// irrelevant, the keys are distinct), the bitonic sort.  The scoring of the sorted candidates with its skip rule runs on the first wavefront alone, as in chain_rmq_kernel --
// it ends after a few dozen candidates.  Every branch around a barrier is taken by the whole workgroup: the values deciding it are read from memory all wavefronts see alike
// (written before the last barrier) or combined through LDS.  Pieces shorter than len_lo or from len_hi on are other launches'.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) chain_rmq_wide_kernel(SeedChainBuffers B, SeedChainParams P, int len_lo, int len_hi)
{
	constexpr int NW = THREADS / 64;
	__shared__ uint64_t s_near[RMQ_NEAR_CAP];
	__shared__ uint64_t s_rank[RMQ_RANK_MAX];
	__shared__ double s_best[NW];
	__shared__ int64_t s_bj[NW];
	__shared__ int s_nb[NW];
	__shared__ int s_cnt;
	const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, w = blockIdx.x;
	const int r = (int)B.pieces[2 * w];
	const Anchor *a = B.anchors + B.a_off[r];
	const int64_t n_all = (int64_t)(B.a_off[r + 1] - B.a_off[r]);
	int32_t *f = B.f + B.a_off[r], *p = B.p + B.a_off[r], *t = B.t + B.a_off[r];
	double *pri = (double *)(B.sort_key_out + B.a_off[r]);
	int32_t max_dist = P.max_gap, max_dist_inner = P.rmq_inner_dist;
	const int32_t bw = P.bw, cap = P.rmq_size_cap;
	if (max_dist < bw) max_dist = bw;
	if (max_dist_inner < 0) max_dist_inner = 0;
	if (max_dist_inner > max_dist) max_dist_inner = max_dist;
	int64_t lo = 0, n = n_all;
	chain_piece_bounds(a, n_all, (int64_t)B.pieces[2 * w + 1], (int64_t)B.piece_len, max_dist, lane, &lo, &n); // (every wavefront for itself: the same answer)
	if (lo >= n || n - lo < (int64_t)len_lo || n - lo >= (int64_t)len_hi) return;
#ifdef MM2AMD_WAVE_EMU // (the emulator's cases check that this path is the one they ran)
	if (tid == 0 && getenv("MM2AMD_PIECE_DEBUG")) fprintf(stderr, "[mm2amd] chain_rmq_wide_kernel<%d>: read %d, anchors %lld..%lld\n", THREADS, r, (long long)lo, (long long)n);
#endif
	for (int64_t i = lo + tid; i < n; i += THREADS) t[i] = 0;
	__threadfence_block();
	__syncthreads();
	int64_t i0 = lo, st = lo, st_in = lo;
	bool give_up = n_all > (int64_t)P.rmq_dev_max_anchors;
	for (int64_t i = lo; i < n && !give_up; ++i) {
		const Anchor ai = a[i];
		const uint64_t ix = ai.x, iy = ai.y;
		const int32_t y_i = (int32_t)iy;
		if (i0 < i && a[i0].x != ix) {
			for (int64_t j = i0 + tid; j < i; j += THREADS) pri[j] = -((double)f[j] + 0.5 * (double)P.chn_pen_gap * (double)((int32_t)a[j].x + (int32_t)a[j].y));
			i0 = i;
			__threadfence_block();
			__syncthreads();
		}
		auto advance = [&](int64_t s0, int32_t dist) { // (each wavefront walks for itself)
			int64_t s = s0;
			while (s < i) {
				const int64_t c = s + lane;
				bool stop = true;
				if (c < i) { const uint64_t cx = a[c].x; stop = !(ix >> 32 != cx >> 32 || ix > cx + (uint64_t)(int64_t)dist); }
				const unsigned long long m = __ballot(stop);
				if (m) { s += __ffsll((long long)m) - 1; break; }
				s += 64;
			}
			if (s > i) s = i;
			if (i0 - s > (int64_t)cap) s = i0 - cap;
			return s;
		};
		st = advance(st, max_dist);
		if (max_dist_inner > 0) st_in = advance(st_in, max_dist_inner);
		// ---- range minimum over [st, i0): per lane, per wavefront, then over the wavefronts ----
		double best = 0;
		int64_t best_j = -1;
		int n_best = 0;
		for (int64_t base = st; base < i0; base += THREADS) {
			const int64_t j = base + tid;
			if (j < i0) {
				const int32_t y_j = (int32_t)a[j].y;
				if ((y_j > y_i - max_dist && y_j < y_i) || (y_j == y_i && j == 0)) {
					const double v = pri[j];
					if (best_j < 0 || v < best) best = v, best_j = j, n_best = 1;
					else if (v == best) ++n_best;
				}
			}
		}
		for (int o = 32; o > 0; o >>= 1) {
			const double ov = __shfl_xor(best, o, 64);
			const int64_t oj = __shfl_xor(best_j, o, 64);
			const int on = __shfl_xor(n_best, o, 64);
			if (oj >= 0) {
				if (best_j < 0 || ov < best) best = ov, best_j = oj, n_best = on;
				else if (ov == best) n_best += on;
			}
		}
		if (lane == 0) s_best[wv] = best, s_bj[wv] = best_j, s_nb[wv] = n_best;
		__syncthreads();
		best = 0, best_j = -1, n_best = 0;
		for (int k = 0; k < NW; ++k) {
			const double ov = s_best[k];
			const int64_t oj = s_bj[k];
			const int on = s_nb[k];
			if (oj >= 0) {
				if (best_j < 0 || ov < best) best = ov, best_j = oj, n_best = on;
				else if (ov == best) n_best += on;
			}
		}
		int32_t max_f = (int32_t)(iy >> 32 & 0xff);
		int64_t max_j = -1;
		if (best_j >= 0) {
			if (n_best > 1) { give_up = true; break; }
			bool exact;
			int32_t width;
			const Anchor aj = a[best_j];
			int32_t sc = f[best_j] + simple_score_dev(ix, iy, aj.x, aj.y, P.chn_pen_gap, P.chn_pen_skip, &exact, &width);
			if (width <= bw && sc > max_f) max_f = sc, max_j = best_j;
			if (!exact && max_dist_inner > 0 && st_in < i0 && y_i > 0) {
				// ---- the close neighbourhood ----
				if (tid == 0) s_cnt = 0;
				__syncthreads();
				for (int64_t base = st_in; base < i0; base += THREADS) {
					const int64_t j = base + tid;
					bool in = false;
					int32_t y_j = 0;
					if (j < i0) { y_j = (int32_t)a[j].y; in = y_j <= y_i - 1 && y_j >= y_i - max_dist_inner; }
					const unsigned long long m = __ballot(in);
					int first = 0;
					if (lane == 0 && m) first = atomicAdd(&s_cnt, (int)__popcll(m));
					first = __shfl(first, 0, 64);
					if (in) { const int d = first + popc_below(m, lane); if (d < RMQ_NEAR_CAP) s_near[d] = (uint64_t)(uint32_t)y_j << 32 | (uint64_t)(uint32_t)j; }
				}
				__syncthreads();
				const int n_c = s_cnt;
				if (n_c > RMQ_NEAR_CAP) { give_up = true; break; }
				const uint64_t *sorted = s_near;
				if (n_c <= B.rmq_rank_max) { // (as in chain_rmq_kernel: a key's place is the number of larger keys; one barrier instead of the network's dozens)
					for (int e = tid; e < n_c; e += THREADS) {
						const uint64_t key = s_near[e];
						int rank = 0;
						for (int k = 0; k < n_c; ++k) rank += s_near[k] > key;
						s_rank[rank] = key;
					}
					sorted = s_rank;
					__syncthreads();
				} else {
				int n_pad = 64;
				while (n_pad < n_c) n_pad <<= 1;
				for (int k = n_c + tid; k < n_pad; k += THREADS) s_near[k] = 0;
				__syncthreads();
				for (int k2 = 2; k2 <= n_pad; k2 <<= 1)
					for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
						for (int e = tid; e < n_pad; e += THREADS) {
							const int partner = e ^ j2;
							if (partner > e) {
								const uint64_t u0 = s_near[e], u1 = s_near[partner];
								const bool desc = (e & k2) == 0;
								if (desc ? u0 < u1 : u0 > u1) s_near[e] = u1, s_near[partner] = u0;
							}
						}
						__syncthreads();
					}
				}
				if (wv == 0) { // the sorted candidates, 64 at a time, first wavefront only (no barrier in here)
					int32_t n_skip = 0;
					bool broke = false;
					for (int base = 0; base < n_c && !broke; base += 64) {
						const int c = base + lane;
						int32_t scc = INT32_MIN, pj = -1;
						int64_t j = -1;
						if (c < n_c) {
							j = (int64_t)(uint32_t)sorted[c];
							bool ex;
							int32_t wd;
							const Anchor cj = a[j];
							const int32_t v = simple_score_dev(ix, iy, cj.x, cj.y, P.chn_pen_gap, P.chn_pen_skip, &ex, &wd);
							if (wd <= bw) scc = f[j] + v, pj = p[j];
						}
						const bool has = scc != INT32_MIN;
						int32_t pm = has ? scc : INT32_MIN;
						for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(pm, o, 64); if (lane >= o) pm = v > pm ? v : pm; }
						int32_t excl = __shfl_up(pm, 1, 64);
						if (lane == 0) excl = INT32_MIN;
						excl = excl > max_f ? excl : max_f;
						const bool improve = has && scc > excl;
						if (has && pj >= 0) t[pj] = (int32_t)i;
						__threadfence_block();
						const bool marked = has && !improve && t[j] == (int32_t)i;
						unsigned long long imp = __ballot(improve), mk = __ballot(marked), ev = imp | mk;
						int stop_lane = 64;
						while (ev) {
							const int b = __ffsll((long long)ev) - 1;
							ev &= ev - 1;
							if (imp >> b & 1) { if (n_skip > 0) --n_skip; }
							else if (++n_skip > P.max_chain_skip) { stop_lane = b; break; }
						}
						if (stop_lane < 64) broke = true, imp &= (1ull << stop_lane) - 1ull;
						if (imp) {
							const int last = 63 - __clzll((long long)imp);
							max_f = __shfl(scc, last, 64);
							max_j = __shfl(j, last, 64);
						}
					}
				}
			}
		}
		if (tid == 0) f[i] = max_f, p[i] = (int32_t)max_j; // (the first wavefront holds the scored neighbourhood's result)
		__threadfence_block();
		__syncthreads();
	}
	if (tid == 0 && give_up) atomicOr(&B.tie_flag[r], 1u);
}

void launch_chain_rmq(const SeedChainBuffers &B, const SeedChainParams &P, void *stream)
{
	hipStream_t s = (hipStream_t)stream;
	const dim3 grid(B.pieces ? B.n_pieces : B.n_reads);
	HIP_CHECK(hipMemsetAsync(B.tie_flag, 0, (size_t)B.n_reads * 4, s));
	uint64_t *timing = nullptr;
	const int n_w = (int)grid.x < (1 << 20) ? (int)grid.x : 1 << 20;
	if (getenv("MM2AMD_RMQ_TIMING")) { HIP_CHECK(hipMalloc((void **)&timing, (size_t)n_w * 16)); HIP_CHECK(hipMemsetAsync(timing, 0, (size_t)n_w * 16, s)); }
	if (getenv("MM2AMD_RMQ_NEAR_TINY")) hipLaunchKernelGGL(chain_rmq_kernel<64>, grid, dim3(64), 0, s, B, P, timing); // tests: most reads go on to the second launch
	else hipLaunchKernelGGL(chain_rmq_kernel<RMQ_NEAR_SMALL>, grid, dim3(64), 0, s, B, P, timing);
	if (timing) { // diagnostics: the launch's wavefronts by time spent -- is it the heaviest one or their sum that the launch waits for?
		std::vector<uint64_t> h((size_t)n_w * 2);
		HIP_CHECK(hipStreamSynchronize(s));
		HIP_CHECK(hipMemcpy(h.data(), timing, h.size() * 8, hipMemcpyDeviceToHost));
		HIP_CHECK(hipFree(timing));
		std::vector<int> order(n_w);
		for (int i = 0; i < n_w; ++i) order[i] = i;
		std::sort(order.begin(), order.end(), [&](int x, int y) { return h[2 * x] > h[2 * y]; });
		double sum = 0, anchors = 0;
		for (int i = 0; i < n_w; ++i) sum += (double)h[2 * i], anchors += (double)(h[2 * i + 1] >> 32);
		fprintf(stderr, "[mm2amd] chain_rmq_kernel: %d wavefronts (%d reads), %.0f anchors, %.1f ms of wavefront time in all; the longest:", n_w, B.n_reads, anchors, sum / 1e5);
		for (int i = 0; i < 6 && i < n_w; ++i) fprintf(stderr, " %.2f ms / %llu anchors (read %u)", (double)h[2 * order[i]] / 1e5, (unsigned long long)(h[2 * order[i] + 1] >> 32), (unsigned)h[2 * order[i] + 1]);
		fprintf(stderr, "\n");
	}
	if (B.pieces && B.piece_dense > 0) { // the long clusters, by workgroups of 4 and (four times the length on) 16 wavefronts
		const char *e = getenv("MM2AMD_RMQ_DENSE_BIG"); // (A/B: from how many times piece_dense on a workgroup has 16 wavefronts)
		const int big = B.piece_dense * (e && atoi(e) > 0 ? atoi(e) : 4);
		hipLaunchKernelGGL(chain_rmq_wide_kernel<256>, grid, dim3(256), 0, s, B, P, B.piece_dense, big);
		hipLaunchKernelGGL(chain_rmq_wide_kernel<1024>, grid, dim3(1024), 0, s, B, P, big, INT32_MAX);
	}
	hipLaunchKernelGGL(chain_rmq_kernel<RMQ_NEAR_CAP>, grid, dim3(64), 0, s, B, P, (uint64_t *)nullptr);
	HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// Chain backtrack + compaction (mg_chain_backtrack lchain.c:27-76, mg_chain_bk_end :9-25, compact_a :78-111), one wavefront
// per read.  Chain ends are visited in the order the reference's unstable radix sort leaves them in (equal scores are common),
// so that sort is replayed exactly (tie_exact_replay, all buckets); the walk itself is sequential (lane 0) and latency-bound,
// which is why the kernel keeps its footprint small: thousands of reads are in flight per GPU.
// Per-read scratch lives in arrays that are dead by now (the pre/post-sort key/value arrays and t[]); results go to dense
// output arrays through two atomic cursors.
// ---------------------------------------------------------------------------------------------------------
#ifndef MM2AMD_BT_LDS_CAP
#define MM2AMD_BT_LDS_CAP 512
#endif
constexpr int BT_LDS_CAP = MM2AMD_BT_LDS_CAP; // chain ends (or chains) whose sort runs in LDS (tools/sanitize_emu.sh builds with a tiny one, so that every read takes the global-scratch path)
constexpr int BT_STACK = BT_LDS_CAP / 65 + 2; // frames are disjoint ranges of more than 64 elements
constexpr int BT_PATH_CAP = 2048;             // anchors of a walk recorded in LDS (longer chains chase on through global memory)

__device__ void bt_sort(uint64_t *K, uint32_t *I, int32_t n, uint32_t *cnt, uint32_t *head, uint32_t *start, uint32_t *child_mask, TieFrame *stack, int stack_cap)
{
	tie_exact_replay(SplitStore{K, I, -1}, n, nullptr, 0, true, cnt, head, start, child_mask, stack, stack_cap);
}

__global__ void __launch_bounds__(64) chain_backtrack_kernel(SeedChainBuffers B, int min_cnt, int min_sc, int max_drop, int rmq)
{
	__shared__ uint64_t lK[BT_LDS_CAP];
	__shared__ uint32_t lI[BT_LDS_CAP];
	__shared__ uint32_t cnt[256], head[256], start[256], child_mask[8];
	__shared__ TieFrame stack[BT_STACK];
	__shared__ uint16_t path[BT_PATH_CAP]; // anchor indices within the read, 16 bits each (reads with more than 65536 anchors record nothing: pcap below) -- 13 KB of LDS per wave instead of 17.5: eleven reads per CU instead of nine
	__shared__ int32_t s_nu, s_nv;
	__shared__ uint64_t s_aoff, s_uoff;
	const int lane = threadIdx.x;
	const int r = blockIdx.x;
	const uint64_t ao = B.a_off[r];
	const int32_t n = (int32_t)(B.a_off[r + 1] - ao);
	const Anchor *a = B.anchors + ao;
	const int32_t *f = B.f + ao, *p = B.p + ao;
	int32_t *t = B.t + ao;
	if (rmq) { // chain_rmq_kernel's flag: a read it gave up (bit 0) stays empty here and is chained by the host; bit 1 was between its two launches
		const uint32_t fl = B.tie_flag[r];
		if (lane == 0 && fl != (fl & 1u)) B.tie_flag[r] = fl & 1u;
		if (fl & 1u) { if (lane == 0) B.bt_nu[r] = 0, B.bt_nv[r] = 0, B.bt_aoff[r] = 0, B.bt_uoff[r] = 0; return; }
	}
	if (n == 0) { if (lane == 0) B.bt_nu[r] = 0, B.bt_nv[r] = 0, B.bt_aoff[r] = 0, B.bt_uoff[r] = 0; return; }
	// ---- chain ends z = {(f[i], i) : f[i] >= min_sc} in index order (lchain.c:35-40) ----
	int32_t n_z = 0;
	for (int32_t i = lane; i < n; i += 64) { t[i] = 0; n_z += f[i] >= min_sc; }
	for (int o = 32; o > 0; o >>= 1) n_z += __shfl_xor(n_z, o, 64);
	if (n_z == 0) { if (lane == 0) B.bt_nu[r] = 0, B.bt_nv[r] = 0, B.bt_aoff[r] = 0, B.bt_uoff[r] = 0; return; }
	uint64_t *K = n_z <= BT_LDS_CAP ? lK : B.sort_key_out + ao;
	uint32_t *I = n_z <= BT_LDS_CAP ? lI : (uint32_t *)(B.sort_val_out + ao);
	int32_t *v = (int32_t *)(B.sort_val_in + ao);
	uint64_t *u = B.sort_key_in + ao;
	{
		int32_t dst = 0;
		for (int32_t base = 0; base < n; base += 64) {
			const int32_t i = base + lane;
			const bool keep = i < n && f[i] >= min_sc;
			const unsigned long long m = __ballot(keep);
			if (keep) { const int32_t d = dst + popc_below(m, lane); K[d] = (uint64_t)(uint32_t)f[i], I[d] = (uint32_t)i; }
			dst += __popcll(m);
		}
	}
	__threadfence_block();
	__syncthreads();
	// frames hold buckets of more than 64 elements, so at most n/65 per level; large inputs keep the stack in the upper half of v[]'s
	// scratch region (v holds at most n 32-bit entries in a region of n 64-bit ones; u[] can need all n entries when min_cnt < 2)
	TieFrame *big_stack = (TieFrame *)(B.sort_val_in + ao + ((size_t)n + 1) / 2);
	const int big_cap = (int)(((size_t)n - ((size_t)n + 1) / 2) * 8 / sizeof(TieFrame));
	bt_sort(K, I, n_z, cnt, head, start, child_mask, n_z <= BT_LDS_CAP ? stack : big_stack, n_z <= BT_LDS_CAP ? BT_STACK : big_cap); // radix_sort_128x(z, z + n_z), lchain.c:41
	__threadfence_block();
	__syncthreads();
	// ---- walk the ends best-first, claim anchors (lchain.c:57-72 with mg_chain_bk_end inlined) ----
	// Most ends lie on a chain that a better end has already claimed, so the ends are screened 64 at a time (one load of t[] per lane
	// instead of one dependent load after the other); only an end that is still free is walked.  The walk is ONE pointer chase by lane 0
	// -- p, f and t of the next anchor are three independent loads, one round trip per step -- that records the path in LDS; where the
	// chain is cut is known when it ends (the reference marks the path, finds the cut, unmarks, and walks again to collect), and the
	// anchors up to the cut are then claimed by all lanes at once.  p[i] < i, so a walk never meets its own anchors: no marks needed.
	int32_t n_v = 0, n_u_acc = 0; // identical in all lanes
	const int32_t pcap = n <= 65536 ? BT_PATH_CAP : 0; // how much of a walk is recorded
	for (int32_t k0 = n_z - 1; k0 >= 0; k0 -= 64) {
		const int32_t k = k0 - lane;
		int32_t zi = 0, zx = 0;
		if (k >= 0) zi = (int32_t)I[k], zx = (int32_t)K[k];
		unsigned long long todo = __ballot(k >= 0 && t[zi] == 0);
		while (todo) {
			const int src = __ffsll((long long)todo) - 1; // lowest lane = largest k: the best-scoring end first
			const int32_t czi = __shfl(zi, src, 64), czx = __shfl(zx, src, 64);
			int32_t keep = 0, max_s = 0;
			if (lane == 0) {
				int32_t i = czi, pi = p[czi], len = 0;
				for (;;) {
					if (len < pcap) path[len] = (uint16_t)i;
					++len;
					const int32_t nxt = pi;
					int32_t fn = 0, tn = 0, pn = -1;
					if (nxt >= 0) fn = f[nxt], tn = t[nxt], pn = p[nxt];
					const int32_t sc = nxt < 0 ? czx : czx - fn;
					if (sc > max_s) max_s = sc, keep = len; // the cut moves to just before nxt: the len anchors walked so far are kept
					else if (max_s - sc > max_drop) break;
					if (nxt < 0 || tn != 0) break;          // the chain's start, or an anchor another chain has claimed
					i = nxt, pi = pn;
				}
			}
			keep = __builtin_amdgcn_readfirstlane(keep), max_s = __builtin_amdgcn_readfirstlane(max_s);
			WAVE_SYNC();
			const int32_t n_rec = keep < pcap ? keep : pcap;
			for (int32_t q = lane; q < n_rec; q += 64) { const int32_t idx = path[q]; v[n_v + q] = idx, t[idx] = 1; }
			if (keep > pcap && lane == 0) { // longer than the recorded part (whole contigs as queries): chase on from its last anchor
				int32_t i = pcap > 0 ? p[path[pcap - 1]] : czi;
				for (int32_t q = pcap; q < keep; ++q) { v[n_v + q] = i, t[i] = 1; i = p[i]; }
			}
			if (max_s >= min_sc && keep > 0 && keep >= min_cnt) { // (a rejected chain keeps its anchors claimed, lchain.c:66-67)
				if (lane == 0) u[n_u_acc] = (uint64_t)(uint32_t)max_s << 32 | (uint64_t)(uint32_t)keep;
				++n_u_acc, n_v += keep;
			}
			__threadfence_block();
			WAVE_SYNC();
			todo &= ~((2ull << src) - 1ull);
			todo = __ballot((todo >> lane & 1ull) != 0 && t[zi] == 0); // which of the block's remaining ends are still free
		}
	}
	if (lane == 0) {
		s_nu = n_u_acc, s_nv = n_v;
		s_aoff = n_v ? atomicAdd((unsigned long long *)&B.bt_cursor[0], (unsigned long long)n_v) : 0;
		s_uoff = n_u_acc ? atomicAdd((unsigned long long *)&B.bt_cursor[1], (unsigned long long)n_u_acc) : 0;
		B.bt_nu[r] = n_u_acc, B.bt_nv[r] = n_v, B.bt_aoff[r] = s_aoff, B.bt_uoff[r] = s_uoff;
	}
	__threadfence_block();
	__syncthreads();
	const int32_t n_u = s_nu;
	if (n_u == 0) return;
	// ---- compact_a: chains in ascending anchor order, chains ordered by the reference position of their first anchor ----
	uint32_t *cst = (uint32_t *)t; // start of each chain inside v[] (t[] is dead now)
	if (lane == 0) { uint32_t k = 0; for (int32_t i = 0; i < n_u; ++i) { cst[i] = k; k += (uint32_t)u[i]; } }
	__threadfence_block();
	__syncthreads();
	uint64_t *K2 = n_u <= BT_LDS_CAP ? lK : B.sort_key_out + ao;
	uint32_t *I2 = n_u <= BT_LDS_CAP ? lI : (uint32_t *)(B.sort_val_out + ao);
	for (int32_t i = lane; i < n_u; i += 64) { // w[i].x = b[k].x : first anchor of chain i after reversal = last entry of its v segment
		const uint32_t ni = (uint32_t)u[i];
		K2[i] = a[v[cst[i] + ni - 1]].x, I2[i] = (uint32_t)i;
	}
	__threadfence_block();
	__syncthreads();
	bt_sort(K2, I2, n_u, cnt, head, start, child_mask, n_u <= BT_LDS_CAP ? stack : big_stack, n_u <= BT_LDS_CAP ? BT_STACK : big_cap); // radix_sort_128x(w, w + n_u), lchain.c:99
	__threadfence_block();
	__syncthreads();
	Anchor *oa = B.bt_out_a + s_aoff;
	uint64_t *ou = B.bt_out_u + s_uoff;
	uint32_t k = 0;
	for (int32_t i = 0; i < n_u; ++i) {
		const uint32_t j = I2[i], c = (uint32_t)u[j], s0 = cst[j];
		if (lane == 0) ou[i] = u[j];
		for (uint32_t q = lane; q < c; q += 64) oa[k + q] = a[v[s0 + (c - 1 - q)]];
		k += c;
	}
}

// Long-join re-chaining (map.c:283-292) starts from the read's CHAINED anchors (the backtrack's compacted output, chain by chain) sorted by
// reference position again: read r of the re-chain list takes src_off[r] .. of `src` to slot a_off[r] .. of the sort's (key, value) input, in
// that order -- the order the reference's unstable radix_sort_128x starts from, which the per-read sort replays where keys are equal.
__global__ void __launch_bounds__(256) rechain_gather_kernel(SeedChainBuffers B, const Anchor *src, const uint64_t *src_off)
{
	const int r = blockIdx.x;
	const uint64_t ao = B.a_off[r], so = src_off[r];
	const int64_t n = (int64_t)(B.a_off[r + 1] - ao);
	const uint64_t low_mask = (1ULL << (32 + B.rid_bits)) - 1ULL;
	for (int64_t i = threadIdx.x; i < n; i += 256) {
		const Anchor a = src[so + (uint64_t)i];
		B.sort_key_in[ao + (uint64_t)i] = (a.x >> 63) << (32 + B.rid_bits) | (a.x & low_mask); // the compact key seed_expand_kernel writes: strand | rid | rpos
		B.sort_val_in[ao + (uint64_t)i] = a.y;
	}
}

void launch_rechain_gather(const SeedChainBuffers &B, const Anchor *src, const uint64_t *src_off, void *stream)
{
	if (B.n_reads <= 0) return;
	hipLaunchKernelGGL(rechain_gather_kernel, dim3(B.n_reads), dim3(256), 0, (hipStream_t)stream, B, src, src_off);
	HIP_CHECK(hipGetLastError());
}

void launch_chain_backtrack(const SeedChainBuffers &B, const SeedChainParams &P, void *stream)
{
	const int max_drop = P.is_cdna ? INT32_MAX : P.bw;
	HIP_CHECK(hipMemsetAsync(B.bt_cursor, 0, 16, (hipStream_t)stream));
	hipLaunchKernelGGL(chain_backtrack_kernel, dim3(B.n_reads), dim3(64), 0, (hipStream_t)stream, B, P.min_cnt, P.min_chain_score, max_drop, P.rmq);
	HIP_CHECK(hipGetLastError());
}


} // namespace mm2amd

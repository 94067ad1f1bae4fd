// Device-wide stable radix sort of (u64 key, u64 value) pairs and exclusive prefix sum of u32, hand-written for gfx950 (no rocPRIM).
//
// The index build sorts ~190 minimizers per kilobase of reference -- 560 M (hash, position) pairs for a 3 Gb genome -- by hash
// (index.c:236 sorts each bucket with radix_sort_128x; here one device-wide sort replaces the 2^14 per-bucket sorts).  HBM-bound:
// a pass reads 8 B per pair for its histogram and moves 32 B per pair in its scatter.
//
// One LSD pass over an 8-bit digit is four launches, none of which waits for another workgroup (no decoupled look-back, so no
// forward-progress assumption between workgroups):
//   rs_hist_kernel          tile (4096 pairs) -> table[tile][digit] counts, and per-chunk-of-128-tiles column sums by atomics
//   rs_chunk_scan_kernel    one workgroup: column-wise exclusive scan over the chunks, and the digits' global bases
//   rs_block_offsets_kernel one workgroup per chunk: counts -> global offsets, walking the chunk's 128 rows (coalesced: lane = digit)
//   rs_scatter_kernel       tile -> ranks its pairs (stable), reorders the tile by digit in LDS, writes runs of equal digits
// Stability inside a tile: wave w owns the tile's pairs [1024 w, 1024 (w+1)) and visits them 64 at a time in order; the rank of a
// pair among its wave's pairs of the same digit is (the wave's running counter of the digit) + (matching lanes below it).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdexcept>
#include "hip_util.hpp"
#include "device_sort.hpp"
#include "device_sort_dev.hpp"

namespace mm2amd {

namespace {

__global__ void __launch_bounds__(kSortThreads) rs_hist_kernel(const uint64_t *keys, uint64_t n, int shift, uint32_t mask, uint32_t *table, uint32_t *chunk_sums)
{
	__shared__ uint32_t cnt[kRadix];
	const int tid = threadIdx.x;
	cnt[tid] = 0;
	__syncthreads();
	const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
	uint64_t key[kSortItems];
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint64_t i = base + (uint64_t)(r * kSortThreads + tid);
		key[r] = i < n ? keys[i] : 0;
	}
#pragma unroll
	for (int r = 0; r < kSortItems; ++r)
		if (base + (uint64_t)(r * kSortThreads + tid) < n) atomicAdd(&cnt[(uint32_t)(key[r] >> shift) & mask], 1u);
	__syncthreads();
	const uint32_t c = cnt[tid];
	table[(uint64_t)blockIdx.x * kRadix + tid] = c;
	if (c) atomicAdd(&chunk_sums[(uint64_t)(blockIdx.x / kChunkTiles) * kRadix + tid], c);
}

__global__ void __launch_bounds__(kSortThreads) rs_chunk_scan_kernel(uint32_t *chunk_sums, uint32_t n_chunks, uint32_t *digit_base)
{
	__shared__ uint32_t sh[4];
	const int d = threadIdx.x;
	uint32_t run = 0;
#pragma unroll 8
	for (uint32_t c = 0; c < n_chunks; ++c) {
		const uint32_t t = chunk_sums[(uint64_t)c * kRadix + d];
		chunk_sums[(uint64_t)c * kRadix + d] = run;
		run += t;
	}
	uint32_t total;
	digit_base[d] = block_exclusive_sum(run, sh, total); // pairs whose digit is smaller
}

__global__ void __launch_bounds__(kSortThreads) rs_block_offsets_kernel(uint32_t *table, uint32_t n_tiles, const uint32_t *chunk_sums, const uint32_t *digit_base)
{
	const int d = threadIdx.x;
	const uint32_t c = blockIdx.x, b0 = c * kChunkTiles, b1 = n_tiles < b0 + kChunkTiles ? n_tiles : b0 + kChunkTiles;
	uint32_t run = chunk_sums[(uint64_t)c * kRadix + d] + digit_base[d];
#pragma unroll 8
	for (uint32_t b = b0; b < b1; ++b) {
		const uint32_t t = table[(uint64_t)b * kRadix + d];
		table[(uint64_t)b * kRadix + d] = run;
		run += t;
	}
}

__global__ void __launch_bounds__(kSortThreads) rs_scatter_kernel(const uint64_t *kin, const uint64_t *vin, uint64_t *kout, uint64_t *vout, uint64_t n,
                                                                  int shift, uint32_t mask, const uint32_t *table)
{
	__shared__ uint32_t wcnt[kSortThreads / 64][kRadix]; // a wave's running digit counters; then the tile-local slot where the wave's run of a digit starts
	__shared__ uint32_t goff[kRadix];                    // (global offset of the tile's run of a digit) - (its tile-local start)
	__shared__ uint32_t sh[4];
	__shared__ uint64_t buf[kSortTile];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
	const uint32_t tile_n = n - base < (uint64_t)kSortTile ? (uint32_t)(n - base) : (uint32_t)kSortTile;
#pragma unroll
	for (int i = 0; i < kSortThreads / 64; ++i) wcnt[i][tid] = 0;
	__syncthreads();
	const uint32_t e0 = (uint32_t)(w * (kSortItems * 64) + lane);
	uint64_t key[kSortItems], val[kSortItems];
	uint32_t slot[kSortItems];
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint32_t e = e0 + (uint32_t)r * 64;
		key[r] = e < tile_n ? kin[base + e] : 0;
	}
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint32_t e = e0 + (uint32_t)r * 64;
		val[r] = e < tile_n ? vin[base + e] : 0;
	}
	const uint64_t below = (1ull << lane) - 1;
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const bool valid = e0 + (uint32_t)r * 64 < tile_n;
		const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
		uint64_t peers = __ballot(valid); // lanes holding the same digit
#pragma unroll
		for (int b = 0; b < 8; ++b) {
			const bool bit = d >> b & 1;
			const uint64_t bal = __ballot(bit);
			peers &= bit ? bal : ~bal;
		}
		const int leader = valid ? __ffsll((long long)peers) - 1 : lane;
		uint32_t old = 0;
		if (valid && lane == leader) old = atomicAdd(&wcnt[w][d], (uint32_t)__popcll(peers)); // (one lane per digit and round; atomic so that the counter is never held in a register)
		old = __shfl(old, leader);
		slot[r] = old + (uint32_t)__popcll(peers & below);
	}
	__syncthreads();
	{
		const uint32_t c0 = wcnt[0][tid], c1 = wcnt[1][tid], c2 = wcnt[2][tid], c3 = wcnt[3][tid];
		uint32_t total;
		const uint32_t ds = block_exclusive_sum(c0 + c1 + c2 + c3, sh, total); // tile-local start of digit tid's run
		wcnt[0][tid] = ds, wcnt[1][tid] = ds + c0, wcnt[2][tid] = ds + c0 + c1, wcnt[3][tid] = ds + c0 + c1 + c2;
		goff[tid] = table[(uint64_t)blockIdx.x * kRadix + tid] - ds;
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < kSortItems; ++r)
		if (e0 + (uint32_t)r * 64 < tile_n) {
			slot[r] += wcnt[w][(uint32_t)(key[r] >> shift) & mask];
			buf[slot[r]] = key[r];
		}
	__syncthreads();
	uint32_t gpos[kSortItems];
#pragma unroll
	for (int j = 0; j < kSortItems; ++j) {
		const uint32_t s = (uint32_t)(j * kSortThreads + tid);
		gpos[j] = 0;
		if (s < tile_n) {
			const uint64_t k = buf[s];
			gpos[j] = goff[(uint32_t)(k >> shift) & mask] + s;
			kout[gpos[j]] = k;
		}
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < kSortItems; ++r)
		if (e0 + (uint32_t)r * 64 < tile_n) buf[slot[r]] = val[r];
	__syncthreads();
#pragma unroll
	for (int j = 0; j < kSortItems; ++j) {
		const uint32_t s = (uint32_t)(j * kSortThreads + tid);
		if (s < tile_n) vout[gpos[j]] = buf[s];
	}
}

// ---- exclusive sum ----
__global__ void __launch_bounds__(kSortThreads) xs_reduce_kernel(const uint32_t *in, uint64_t n, uint32_t *sums)
{
	__shared__ uint32_t sh[4];
	const int tid = threadIdx.x;
	const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
	uint32_t s = 0;
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint64_t i = base + (uint64_t)(r * kSortThreads + tid);
		if (i < n) s += in[i];
	}
	uint32_t total;
	(void)block_exclusive_sum(s, sh, total);
	if (tid == 0) sums[blockIdx.x] = total;
}

// tile b of `in` scanned into `out`, starting from carry[b] (nullptr: 0); the last tile also writes out[n] = the total.
// `in` may BE `out` (the recursion over the tile sums scans them in place): every item of the tile is loaded into registers before the first store of the
// tile, and tiles do not overlap -- the loads-then-stores order below is what makes the aliasing safe; keep it.
__global__ void __launch_bounds__(kSortThreads) xs_apply_kernel(const uint32_t *in, uint32_t *out, uint64_t n, const uint32_t *carry)
{
	__shared__ uint32_t sh[4];
	const int tid = threadIdx.x;
	const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
	uint32_t run = carry ? carry[blockIdx.x] : 0;
	uint32_t v[kSortItems];
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint64_t i = base + (uint64_t)(r * kSortThreads + tid);
		v[r] = i < n ? in[i] : 0;
	}
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint64_t i = base + (uint64_t)(r * kSortThreads + tid);
		uint32_t total;
		const uint32_t ex = block_exclusive_sum(v[r], sh, total);
		if (i < n) out[i] = run + ex;
		run += total;
	}
	if (tid == 0 && base + kSortTile >= n) out[n] = run;
}

} // namespace

void device_exclusive_sum_u32(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t stream)
{
	if (n >= (1ull << 32)) throw std::invalid_argument("[mm2amd] device_exclusive_sum_u32: n must be below 2^32");
	if (n == 0) { HIP_CHECK(hipMemsetAsync(out, 0, 4, stream)); return; }
	const uint64_t n_tiles = (n + kSortTile - 1) / kSortTile;
	if (n_tiles == 1) {
		hipLaunchKernelGGL(xs_apply_kernel, dim3(1), dim3(kSortThreads), 0, stream, in, out, n, (const uint32_t *)nullptr);
		HIP_CHECK(hipGetLastError());
		return;
	}
	DevBuf<uint32_t> sums;
	sums.ensure(n_tiles + 1, 1.0);
	hipLaunchKernelGGL(xs_reduce_kernel, dim3((unsigned)n_tiles), dim3(kSortThreads), 0, stream, in, n, sums.p);
	HIP_CHECK(hipGetLastError());
	device_exclusive_sum_u32(sums.p, sums.p, n_tiles, stream);
	hipLaunchKernelGGL(xs_apply_kernel, dim3((unsigned)n_tiles), dim3(kSortThreads), 0, stream, in, out, n, (const uint32_t *)sums.p);
	HIP_CHECK(hipGetLastError());
	stream_wait(stream); // `sums` is freed on return (stream_wait sleeps between polls: hipStreamSynchronize spins a core, hip_util.hpp)
}

int device_sort_pairs_u64(uint64_t *k0, uint64_t *v0, uint64_t *k1, uint64_t *v1, uint64_t n, int bits, hipStream_t stream)
{
	if (n >= (1ull << 32)) throw std::invalid_argument("[mm2amd] device_sort_pairs_u64: n must be below 2^32");
	if (bits > 64) bits = 64;
	if (n < 2 || bits <= 0) return 0;
	const uint64_t n_tiles = (n + kSortTile - 1) / kSortTile, n_chunks = (n_tiles + kChunkTiles - 1) / kChunkTiles;
	DevBuf<uint32_t> table, chunk_sums, digit_base;
	table.ensure(n_tiles * kRadix, 1.0), chunk_sums.ensure(n_chunks * kRadix, 1.0), digit_base.ensure(kRadix, 1.0);
	uint64_t *kb[2] = { k0, k1 }, *vb[2] = { v0, v1 };
	int cur = 0;
	for (int shift = 0; shift < bits; shift += 8) {
		const int pass_bits = std::min(8, bits - shift);
		const uint32_t mask = (1u << pass_bits) - 1;
		HIP_CHECK(hipMemsetAsync(chunk_sums.p, 0, n_chunks * kRadix * 4, stream));
		hipLaunchKernelGGL(rs_hist_kernel, dim3((unsigned)n_tiles), dim3(kSortThreads), 0, stream, (const uint64_t *)kb[cur], n, shift, mask, table.p, chunk_sums.p);
		hipLaunchKernelGGL(rs_chunk_scan_kernel, dim3(1), dim3(kSortThreads), 0, stream, chunk_sums.p, (uint32_t)n_chunks, digit_base.p);
		hipLaunchKernelGGL(rs_block_offsets_kernel, dim3((unsigned)n_chunks), dim3(kSortThreads), 0, stream, table.p, (uint32_t)n_tiles, (const uint32_t *)chunk_sums.p,
		                   (const uint32_t *)digit_base.p);
		hipLaunchKernelGGL(rs_scatter_kernel, dim3((unsigned)n_tiles), dim3(kSortThreads), 0, stream, (const uint64_t *)kb[cur], (const uint64_t *)vb[cur], kb[cur ^ 1], vb[cur ^ 1], n,
		                   shift, mask, (const uint32_t *)table.p);
		HIP_CHECK(hipGetLastError());
		cur ^= 1;
	}
	stream_wait(stream); // the tables are freed on return
	return cur;
}

} // namespace mm2amd

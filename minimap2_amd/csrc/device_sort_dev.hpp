// Device-side helpers shared by device_sort.hip and index_build.hip: the tile shape of the device-wide passes and the workgroup prefix sum.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace mm2amd {

constexpr int kSortThreads = 256, kSortItems = 16, kSortTile = kSortThreads * kSortItems, kRadix = 256, kChunkTiles = 128;

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, int lane)
{
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) {
		const uint32_t u = __shfl_up(v, (unsigned)d);
		if (lane >= d) v += u;
	}
	return v;
}

// exclusive sum over the 256 threads of a workgroup (thread order); `total` receives the workgroup's sum.  sh: 4 words of LDS, free again on return.
__device__ __forceinline__ uint32_t block_exclusive_sum(uint32_t v, uint32_t *sh, uint32_t &total)
{
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	const uint32_t incl = wave_inclusive_sum(v, lane);
	if (lane == 63) sh[w] = incl;
	__syncthreads();
	uint32_t off = 0, tot = 0;
#pragma unroll
	for (int i = 0; i < kSortThreads / 64; ++i) {
		const uint32_t t = sh[i];
		if (i < w) off += t;
		tot += t;
	}
	__syncthreads();
	total = tot;
	return off + incl - v;
}

} // namespace mm2amd

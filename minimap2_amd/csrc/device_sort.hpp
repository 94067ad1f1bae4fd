// Device-wide primitives of the index build (device_sort.hip): a stable LSD radix sort of (key, value) pairs and the exclusive
// prefix sum of a u32 array -- the GPU counterparts of radix_sort_128x (ksort.h:101-151 as instantiated at sketch.c:13 / index.c:236)
// and of the running offsets mm_idx_gen's worker_post keeps (index.c:249-268).  Hand-written for gfx950; no rocPRIM / hipCUB.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace mm2amd {

// Sorts n pairs by key bits [0, bits), stably (equal keys keep their input order), ping-ponging between (k0, v0) and (k1, v1).
// Returns 0 when the sorted pairs end up in (k0, v0), 1 when in (k1, v1).  n < 2^32.  Work is queued on `stream`; temporary tables
// are allocated and freed inside (the call synchronises the stream before returning).
int device_sort_pairs_u64(uint64_t *k0, uint64_t *v0, uint64_t *k1, uint64_t *v1, uint64_t n, int bits, hipStream_t stream);

// out[i] = in[0] + ... + in[i-1] (mod 2^32), out[n] = the total; `out` holds n + 1 entries and may alias `in`.  n < 2^32.
void device_exclusive_sum_u32(const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t stream);

} // namespace mm2amd

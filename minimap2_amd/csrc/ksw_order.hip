// The DP batch's launch order made on the device (ksw_classify.hpp): when the job records are born there (region_plan_kernel), nothing of them
// crosses PCIe -- the jobs are classed and counted (one pass: class x cost-bucket histogram by atomics, per-class sizing figures reduced in LDS
// first), the histogram is scanned by the pass's last workgroup, and a second pass scatters the records into launch order.  The order inside a bucket is
// whatever the atomics give: the launch order is a scheduling hint (longest jobs first), results do not depend on it.
// Algorithmic bytes: 48 B read twice + 48 B written + 4 B per job.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_classify.hpp"

namespace mm2amd {

namespace {
constexpr int kBins = kNTiers * kOrderBuckets;

__global__ void __launch_bounds__(256) ksw_order_count_kernel(const KswJob *jobs, uint32_t n, KswClassCtx C, uint32_t *hist, uint32_t *bucket, KswOrderResult *out)
{
	__shared__ unsigned long long s_sum[kNTiers][3];  // alg_bytes, cells, sum_len
	__shared__ unsigned long long s_max64[kNTiers][2]; // slot_bytes, tmp_cap
	__shared__ unsigned int s_max[kNTiers][4];         // max_ring, max_Q16, max_rows, max_ncol
	__shared__ unsigned int s_n[kNTiers];
	for (int t = threadIdx.x; t < kNTiers; t += 256) {
		s_sum[t][0] = s_sum[t][1] = s_sum[t][2] = 0, s_max64[t][0] = s_max64[t][1] = 0, s_max[t][0] = s_max[t][1] = s_max[t][2] = s_max[t][3] = 0, s_n[t] = 0;
	}
	__syncthreads();
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const KswJob j = jobs[i];
		KswClassOut o;
		ksw_classify(j, C, o);
		const uint32_t bk = (uint32_t)(o.tier * kOrderBuckets + (kOrderBuckets - 1 - o.cb));
		bucket[i] = bk;
		atomicAdd(&hist[bk], 1u);
		const int t = o.tier;
		atomicAdd(&s_n[t], 1u);
		unsigned long long alg = sizeof(KswJob) + sizeof(KswRes);
		if (o.live) {
			alg += (unsigned long long)j.qlen + ((j.flag & KSWJ_T_PACKED) ? (unsigned long long)(j.tlen + 1) / 2 : (unsigned long long)j.tlen);
			atomicAdd(&s_sum[t][1], (unsigned long long)j.qlen * (unsigned long long)j.tlen);
			atomicMax(&s_max[t][0], (unsigned int)o.ring_need);
			atomicMax(&s_max[t][1], (unsigned int)((j.qlen + 15) / 16 * 16));
			if (!(j.flag & KSW_SCORE_ONLY)) {
				if (o.db > 160 * 1024 || ksw_band_sets(o.tier)) alg += (unsigned long long)o.db; // (the banded kernel's direction bytes always travel: ksw_host.cpp)
				atomicMax(&s_max64[t][0], (unsigned long long)o.db);
				atomicMax(&s_max64[t][1], (unsigned long long)j.qlen + (unsigned long long)j.tlen);
				if (o.fast || o.xfast) { atomicMax(&s_max[t][2], (unsigned int)(j.qlen + j.tlen - 1)); atomicMax(&s_max[t][3], (unsigned int)((j.tlen + 63) & ~63)); }
				atomicAdd(&s_sum[t][2], (unsigned long long)j.qlen + (unsigned long long)j.tlen);
			}
		}
		atomicAdd(&s_sum[t][0], alg);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < kNTiers; t += 256) {
		if (s_n[t] == 0) continue;
		KswClassStat &c = out->cls[t];
		atomicAdd(&c.n_jobs, s_n[t]);
		atomicAdd(&c.alg_bytes, s_sum[t][0]), atomicAdd(&c.cells, s_sum[t][1]), atomicAdd(&c.sum_len, s_sum[t][2]);
		atomicMax(&c.slot_bytes, s_max64[t][0]), atomicMax(&c.tmp_cap, s_max64[t][1]);
		atomicMax(&c.max_ring, s_max[t][0]), atomicMax(&c.max_Q16, s_max[t][1]), atomicMax(&c.max_rows, s_max[t][2]), atomicMax(&c.max_ncol, s_max[t][3]);
	}
	// The workgroup that finishes LAST turns the histogram into launch positions (exclusive scan; hist[b] becomes the first position of bin b,
	// tier_beg[t] that of class t).  A kernel of its own for this one-workgroup job waited milliseconds for a CU behind the persistent DP kernels of the
	// other lanes (rocprofv3, call 8: 2.4 ms on average, 23 ms at worst, eight times per step).
	__shared__ uint32_t s_part[256];
	__shared__ int s_last;
	__syncthreads();
	__threadfence(); // (this workgroup's counts and figures are out before its ticket is)
	if (threadIdx.x == 0) s_last = atomicAdd(&out->ticket, 1u) == gridDim.x - 1;
	__syncthreads();
	if (!s_last) return;
	__threadfence();
	constexpr int PER = (kBins + 255) / 256;
	const int tid = threadIdx.x, b0 = tid * PER;
	uint32_t sum = 0;
	for (int k = 0; k < PER; ++k) if (b0 + k < kBins) sum += atomicAdd(&hist[b0 + k], 0u); // (the counts were made by atomics: read them where they live)
	s_part[tid] = sum;
	__syncthreads();
	for (int d = 1; d < 256; d <<= 1) { // Hillis-Steele inclusive scan of the 256 partial sums
		const uint32_t v = tid >= d ? s_part[tid - d] : 0;
		__syncthreads();
		s_part[tid] += v;
		__syncthreads();
	}
	uint32_t acc = s_part[tid] - sum;
	for (int k = 0; k < PER; ++k) {
		const int b = b0 + k;
		if (b >= kBins) break;
		const uint32_t v = atomicExch(&hist[b], acc);
		if (b % kOrderBuckets == 0) out->tier_beg[b / kOrderBuckets] = acc;
		acc += v;
	}
	if (tid == 255) out->tier_beg[kNTiers] = s_part[255];
}

__global__ void __launch_bounds__(256) ksw_order_scatter_kernel(const KswJob *jobs, uint32_t n, uint32_t *hist, const uint32_t *bucket, KswJob *sorted, uint32_t *perm)
{
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		const uint32_t pos = atomicAdd(&hist[bucket[i]], 1u);
		perm[i] = pos;
		sorted[pos] = jobs[i];
	}
}
} // namespace

size_t ksw_order_work_words(size_t n) { return (size_t)kBins + n + 64; }

void ksw_order_device(const KswJob *d_jobs, size_t n, const KswClassCtx &C, KswJob *d_sorted, uint32_t *d_perm, uint32_t *d_work, KswOrderResult *d_out, void *stream)
{
	hipStream_t st = (hipStream_t)stream;
	uint32_t *hist = d_work, *bucket = d_work + kBins;
	HIP_CHECK(hipMemsetAsync(hist, 0, (size_t)kBins * sizeof(uint32_t), st));
	HIP_CHECK(hipMemsetAsync(d_out, 0, sizeof(KswOrderResult), st));
	if (n == 0) return;
	const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
	hipLaunchKernelGGL(ksw_order_count_kernel, dim3(grid), dim3(256), 0, st, d_jobs, (uint32_t)n, C, hist, bucket, d_out);
	hipLaunchKernelGGL(ksw_order_scatter_kernel, dim3(grid), dim3(256), 0, st, d_jobs, (uint32_t)n, hist, bucket, d_sorted, d_perm);
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

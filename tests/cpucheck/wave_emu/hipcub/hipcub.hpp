/* TEST INFRASTRUCTURE: the two device-wide primitives index_build.hip (set-up code, not the per-batch path) takes from hipcub, on the host. */
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <numeric>
#include <vector>
namespace hipcub {
template <typename T> struct DoubleBuffer {
	T *d_buffers[2];
	int selector = 0;
	DoubleBuffer(T *a, T *b) { d_buffers[0] = a, d_buffers[1] = b; }
	T *Current() { return d_buffers[selector]; }
};
struct DeviceRadixSort {
	template <typename K, typename V>
	static hipError_t SortPairs(void *tmp, size_t &tmp_bytes, DoubleBuffer<K> &k, DoubleBuffer<V> &v, int64_t n, int begin_bit, int end_bit, hipStream_t = nullptr)
	{
		if (!tmp) { tmp_bytes = 16; return hipSuccess; }
		std::vector<int64_t> perm((size_t)n);
		std::iota(perm.begin(), perm.end(), 0);
		const K *kk = k.Current();
		const V *vv = v.Current();
		const K mask = end_bit >= (int)sizeof(K) * 8 ? ~(K)0 : (((K)1 << end_bit) - 1);
		std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return ((kk[a] & mask) >> begin_bit) < ((kk[b] & mask) >> begin_bit); });
		K *ko = k.d_buffers[1 - k.selector];
		V *vo = v.d_buffers[1 - v.selector];
		for (int64_t i = 0; i < n; ++i) ko[i] = kk[perm[i]], vo[i] = vv[perm[i]];
		k.selector ^= 1, v.selector ^= 1;
		return hipSuccess;
	}
};
struct DeviceScan {
	template <typename I, typename O>
	static hipError_t ExclusiveSum(void *tmp, size_t &tmp_bytes, const I *in, O *out, int64_t n, hipStream_t = nullptr)
	{
		if (!tmp) { tmp_bytes = 16; return hipSuccess; }
		O acc = 0;
		for (int64_t i = 0; i < n; ++i) { const O v = (O)in[i]; out[i] = acc; acc += v; }
		return hipSuccess;
	}
};
} // namespace hipcub

// The order in which collect_seed_hits_heap (map.c:102-166, MM_F_HEAP_SORT) emits a read's index hits.
//
// The reference merges the position lists of the kept seeds with a binary min-heap keyed on the index entry r = rid<<32 | pos<<1 |
// strand alone (heap_lt, map.c:75), appending same-strand hits to the front part of the anchor array and opposite-strand hits to
// the back part.  Both parts come out ascending in r, i.e. in the order of the anchor key x -- the order the device-wide sort
// produces -- EXCEPT among hits with equal r, whose relative order is whatever the sift-down (ksort.h:43-53) made of it.  Equal r
// means two seeds with the same minimizer (a k-mer repeated in the read); only reads that have such seeds need this replay.
//
// One thread replays the heap for one read.  Shared by the device kernel and by a host unit test (tests/cpucheck/heap_order_test.cpp).
#pragma once
#include <cstdint>
#include "backend.hpp"

namespace mm2amd {

MM2_HD inline void heap_sift_down(uint64_t *hx, uint64_t *hy, uint32_t i, uint32_t n) // ks_heapdown with heap_lt(a, b) = a.x > b.x
{
	uint32_t k = i;
	const uint64_t tx = hx[i], ty = hy[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && hx[k] > hx[k + 1]) ++k;
		if (hx[k] > tx) break;
		hx[i] = hx[k], hy[i] = hy[k]; i = k;
	}
	hx[i] = tx, hy[i] = ty;
}

// n_seed seeds; list(i, &cnt) returns seed i's ascending position list.  hx/hy: scratch for n_seed heap entries.
// emit(i, r) is called once per hit, in pop order.
template <class ListOf, class Emit>
MM2_HD inline void heap_merge_order(uint32_t n_seed, uint64_t *hx, uint64_t *hy, ListOf list, Emit emit)
{
	uint32_t hs = 0;
	for (uint32_t i = 0; i < n_seed; ++i) {
		uint32_t cnt;
		const uint64_t *cr = list(i, &cnt);
		if (cnt > 0) hx[hs] = cr[0], hy[hs] = (uint64_t)i << 32, ++hs;
	}
	if (hs > 1) for (uint32_t i = (hs >> 1) - 1;; --i) { heap_sift_down(hx, hy, i, hs); if (i == 0) break; }
	while (hs > 0) {
		const uint32_t i = (uint32_t)(hy[0] >> 32), k = (uint32_t)hy[0];
		uint32_t cnt;
		const uint64_t *cr = list(i, &cnt);
		emit(i, hx[0]);
		if (k < cnt - 1) ++hy[0], hx[0] = cr[k + 1];
		else hx[0] = hx[hs - 1], hy[0] = hy[hs - 1], --hs;
		if (hs > 0) heap_sift_down(hx, hy, 0, hs);
	}
}

} // namespace mm2amd

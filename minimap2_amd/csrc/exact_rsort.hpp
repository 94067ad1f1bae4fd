// Permutation-exact re-implementation of the reference's in-place unstable radix sort
// (radix_sort_128x / radix_sort_64: ksort.h:101-151, misc.c:155-159).
//
// minimap2 sorts anchors, chain ends, chains and hits by a 64-bit key with an American-flag MSD radix sort that
// is NOT stable; which of two equal keys comes first decides, e.g., which chain claims a shared anchor
// (hard part 2).  To stay byte-identical we replay the same cycle-leader walk.  The walk is
// written iteratively with one 256-entry bucket table and a small explicit stack so the same code runs as a
// per-thread device routine (no recursion, bounded private memory) and on the host.
#pragma once
#include <cstdint>
#include <cstddef>

#ifdef __HIPCC__
#define MM2_HD __host__ __device__
#else
#define MM2_HD
#endif

namespace mm2amd {

struct RsortScratch {
	uint32_t head[256], tail[256];   // offsets relative to the range being partitioned
};

template <typename T, typename KeyFn>
MM2_HD inline void exact_insertion_sort(T *beg, T *end, KeyFn key)
{
	for (T *i = beg + 1; i < end; ++i)
		if (key(*i) < key(*(i - 1))) {
			T tmp = *i, *j;
			for (j = i; j > beg && key(tmp) < key(*(j - 1)); --j) *j = *(j - 1);
			*j = tmp;
		}
}

// One American-flag pass on [beg,end) by byte (key >> shift) & 255.
template <typename T, typename KeyFn>
MM2_HD inline void exact_flag_pass(T *beg, T *end, int shift, KeyFn key, RsortScratch &sc)
{
	for (int k = 0; k < 256; ++k) sc.tail[k] = 0;
	for (T *i = beg; i != end; ++i) ++sc.tail[key(*i) >> shift & 255];
	uint32_t acc = 0;
	for (int k = 0; k < 256; ++k) { sc.head[k] = acc; acc += sc.tail[k]; sc.tail[k] = acc; }
	for (int k = 0; k < 256;) {
		if (sc.head[k] != sc.tail[k]) {
			int l = (int)(key(beg[sc.head[k]]) >> shift & 255);
			if (l != k) {
				T tmp = beg[sc.head[k]], swap;
				do {
					swap = tmp; tmp = beg[sc.head[l]]; beg[sc.head[l]++] = swap;
					l = (int)(key(tmp) >> shift & 255);
				} while (l != k);
				beg[sc.head[k]++] = tmp;
			} else ++sc.head[k];
		} else ++k;
	}
}

// Sort [beg,end) by key(), 64-bit keys.  Sibling buckets are independent, so visiting them from an explicit
// stack instead of the reference's recursion yields the same final permutation.
template <typename T, typename KeyFn>
MM2_HD inline void exact_radix_sort(T *beg, T *end, KeyFn key, RsortScratch &sc)
{
	constexpr int SMALL = 64; // RS_MIN_SIZE
	if (end - beg <= SMALL) { exact_insertion_sort(beg, end, key); return; }
	struct Frame { T *b, *e; int shift; bool partitioned; };
	Frame stack[20]; // per level at most one continuation frame plus one child frame; 8 levels
	int sp = 0;
	stack[sp++] = Frame{beg, end, 56, false};
	while (sp > 0) {
		Frame fr = stack[--sp];
		if (!fr.partitioned) exact_flag_pass(fr.b, fr.e, fr.shift, key, sc);
		if (fr.shift == 0) continue;
		const int ns = fr.shift > 8 ? fr.shift - 8 : 0;
		// children = maximal runs of equal byte at this level
		T *rb = fr.b;
		while (rb < fr.e) {
			const uint64_t byte = key(*rb) >> fr.shift & 255;
			T *re = rb + 1;
			while (re < fr.e && (key(*re) >> fr.shift & 255) == byte) ++re;
			if (re - rb > SMALL) {
				if (re < fr.e) stack[sp++] = Frame{re, fr.e, fr.shift, true}; // rest of this level, scanned later
				stack[sp++] = Frame{rb, re, ns, false};
				break;
			}
			if (re - rb > 1) exact_insertion_sort(rb, re, key);
			rb = re;
		}
	}
}

} // namespace mm2amd

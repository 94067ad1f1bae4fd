/* oracle/rsort.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restatement of the reference's in-place unstable radix sort (ksort.h:101-151, instantiated at misc.c:155-159).
 * The order in which equal keys end up is part of minimap2's observable behaviour (hard part 2),
 * so the cycle-leader permutation is followed step by step rather than replaced by a stable sort. */
#include "oracle.h"

#define SMALL 64 /* RS_MIN_SIZE */

#define DEFINE_RSORT(NAME, T, KEY) \
static void ins_##NAME(T *beg, T *end) \
{ \
	T *i; \
	for (i = beg + 1; i < end; ++i) \
		if (KEY(*i) < KEY(*(i - 1))) { \
			T *j, tmp = *i; \
			for (j = i; j > beg && KEY(tmp) < KEY(*(j - 1)); --j) *j = *(j - 1); \
			*j = tmp; \
		} \
} \
static void msd_##NAME(T *beg, T *end, int shift) \
{ \
	T *head[256], *tail[256], *i; \
	int k; \
	size_t cnt[256] = { 0 }; \
	for (i = beg; i != end; ++i) ++cnt[KEY(*i) >> shift & 255]; \
	for (k = 0, i = beg; k < 256; ++k) head[k] = i, i += cnt[k], tail[k] = i; \
	for (k = 0; k < 256;) { /* American-flag pass: place the element at the head of bucket k, chasing displaced ones */ \
		if (head[k] != tail[k]) { \
			int l = (int)(KEY(*head[k]) >> shift & 255); \
			if (l != k) { \
				T tmp = *head[k], swap; \
				do { \
					swap = tmp; tmp = *head[l]; *head[l]++ = swap; \
					l = (int)(KEY(tmp) >> shift & 255); \
				} while (l != k); \
				*head[k]++ = tmp; \
			} else ++head[k]; \
		} else ++k; \
	} \
	if (shift) { \
		T *b = beg; \
		shift = shift > 8 ? shift - 8 : 0; \
		for (k = 0; k < 256; ++k) { \
			if (tail[k] - b > SMALL) msd_##NAME(b, tail[k], shift); \
			else if (tail[k] - b > 1) ins_##NAME(b, tail[k]); \
			b = tail[k]; \
		} \
	} \
} \
void ora_radix_sort_##NAME(T *beg, T *end) \
{ \
	if (end - beg <= SMALL) ins_##NAME(beg, end); \
	else msd_##NAME(beg, end, 56); \
}

#define KEY128(a) ((a).x)
#define KEY64(a) (a)
DEFINE_RSORT(128x, ora128_t, KEY128)
DEFINE_RSORT(64, uint64_t, KEY64)

/* oracle/sketch.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restatement of mm_sketch (/root/reference/sketch.c:77-143) and hash64 (:28-38). */
#include <string.h>
#include "oracle.h"

static const uint8_t NT4[256] = { /* seq_nt4_table, sketch.c:9-26: ACGT (any case) -> 0..3, U as T, everything else 4 */
#define R16 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4
	0,1,2,3, 4,4,4,4, 4,4,4,4, 4,4,4,4, R16, R16, R16,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,3,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,3,4,4, 4,4,4,4, 4,4,4,4,
	R16, R16, R16, R16, R16, R16, R16, R16
#undef R16
};

const uint8_t *ora_nt4_table(void) { return NT4; }

static inline uint64_t mix64(uint64_t key, uint64_t mask) /* invertible integer hash on 2k bits */
{
	key = (~key + (key << 21)) & mask;
	key = key ^ key >> 24;
	key = ((key + (key << 3)) + (key << 8)) & mask;
	key = key ^ key >> 14;
	key = ((key + (key << 2)) + (key << 4)) & mask;
	key = key ^ key >> 28;
	key = (key + (key << 31)) & mask;
	return key;
}

typedef struct { ora128_t *out; int64_t n, cap; } sink_t;
static inline void emit(sink_t *s, ora128_t v) { if (s->n < s->cap) s->out[s->n] = v; ++s->n; }

int64_t ora_sketch(const char *seq, int len, int w, int k, uint32_t rid, int is_hpc, ora128_t *out, int64_t cap)
{
	const uint64_t shift1 = 2 * (k - 1), mask = (1ULL << 2 * k) - 1;
	uint64_t kmer[2] = { 0, 0 };
	int i, j, l = 0, buf_pos = 0, min_pos = 0, kmer_span = 0;
	int hq[32], hq_front = 0, hq_count = 0; /* run lengths of the last k homopolymer runs (HPC only) */
	ora128_t buf[256], min = { UINT64_MAX, UINT64_MAX };
	sink_t s = { out, 0, cap };

	if (len <= 0 || w <= 0 || w >= 256 || k <= 0 || k > 28) return 0;
	memset(buf, 0xff, sizeof(ora128_t) * w);
	for (i = 0; i < len; ++i) {
		int c = NT4[(uint8_t)seq[i]];
		ora128_t info = { UINT64_MAX, UINT64_MAX };
		if (c < 4) {
			int z;
			if (is_hpc) {
				int run = 1;
				if (i + 1 < len && NT4[(uint8_t)seq[i + 1]] == c) {
					for (run = 2; i + run < len; ++run)
						if (NT4[(uint8_t)seq[i + run]] != c) break;
					i += run - 1;
				}
				hq[(hq_count++ + hq_front) & 0x1f] = run;
				kmer_span += run;
				if (hq_count > k) { kmer_span -= hq[hq_front++]; hq_front &= 0x1f; --hq_count; }
			} else kmer_span = l + 1 < k ? l + 1 : k;
			kmer[0] = (kmer[0] << 2 | c) & mask;
			kmer[1] = (kmer[1] >> 2) | (3ULL ^ c) << shift1;
			if (kmer[0] == kmer[1]) continue; /* strand-symmetric k-mer: no slot, no l increment (:108) */
			z = kmer[0] < kmer[1] ? 0 : 1;
			++l;
			if (l >= k && kmer_span < 256) {
				info.x = mix64(kmer[z], mask) << 8 | kmer_span;
				info.y = (uint64_t)rid << 32 | (uint32_t)i << 1 | z;
			}
		} else l = 0, hq_count = hq_front = 0, kmer_span = 0;
		buf[buf_pos] = info;
		if (l == w + k - 1 && min.x != UINT64_MAX) { /* first full window: flush earlier copies of the minimum (:117-122) */
			for (j = buf_pos + 1; j < w; ++j) if (min.x == buf[j].x && buf[j].y != min.y) emit(&s, buf[j]);
			for (j = 0; j < buf_pos; ++j)     if (min.x == buf[j].x && buf[j].y != min.y) emit(&s, buf[j]);
		}
		if (info.x <= min.x) { /* new (right-most) minimum */
			if (l >= w + k && min.x != UINT64_MAX) emit(&s, min);
			min = info, min_pos = buf_pos;
		} else if (buf_pos == min_pos) { /* the minimum left the window: rescan oldest -> newest */
			if (l >= w + k - 1 && min.x != UINT64_MAX) emit(&s, min);
			for (j = buf_pos + 1, min.x = UINT64_MAX; j < w; ++j) if (min.x >= buf[j].x) min = buf[j], min_pos = j;
			for (j = 0; j <= buf_pos; ++j)                        if (min.x >= buf[j].x) min = buf[j], min_pos = j;
			if (l >= w + k - 1 && min.x != UINT64_MAX) {
				for (j = buf_pos + 1; j < w; ++j) if (min.x == buf[j].x && min.y != buf[j].y) emit(&s, buf[j]);
				for (j = 0; j <= buf_pos; ++j)    if (min.x == buf[j].x && min.y != buf[j].y) emit(&s, buf[j]);
			}
		}
		if (++buf_pos == w) buf_pos = 0;
	}
	if (min.x != UINT64_MAX) emit(&s, min);
	return s.n;
}

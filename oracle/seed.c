/* oracle/seed.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restatement of the per-read seeding stage: /root/reference/seed.c:5-132 and map.c:168-204. */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define F_NO_DIAG  0x001LL
#define F_NO_DUAL  0x002LL
#define F_FOR_ONLY 0x100000LL
#define F_HEAP_SORT 0x400000LL
#define F_QSTRAND   0x100000000LL
#define F_REV_ONLY 0x200000LL
#define SEED_TANDEM (1ULL << 42)
#define SEED_SELF   (1ULL << 43)
#define MAX_HIGH_OCC 128

typedef struct { uint32_t n, q_pos, q_span, flt, seg_id, is_tandem; const uint64_t *cr; } seed_t;

static void heap_down(uint64_t *h, size_t i, size_t n) /* max-heap sift-down */
{
	size_t k = i;
	uint64_t tmp = h[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && h[k] < h[k + 1]) ++k;
		if (h[k] < tmp) break;
		h[i] = h[k]; i = k;
	}
	h[i] = tmp;
}

/* keep, in every run of over-frequent minimizers, only the rarest ~1 per occ_dist bases (mm_seed_select, seed.c:56-96) */
static void thin_high_occ(int32_t n, seed_t *a, int len, int max_occ, int max_max_occ, int dist)
{
	int32_t i, last0, m;
	uint64_t b[MAX_HIGH_OCC];
	if (n == 0 || n == 1) return;
	for (i = m = 0; i < n; ++i) if (a[i].n > (uint32_t)max_occ) ++m;
	if (m == 0) return;
	for (i = 0, last0 = -1; i <= n; ++i) {
		if (i == n || a[i].n <= (uint32_t)max_occ) {
			if (i - last0 > 1) {
				int32_t ps = last0 < 0 ? 0 : (int32_t)(a[last0].q_pos >> 1);
				int32_t pe = i == n ? len : (int32_t)(a[i].q_pos >> 1);
				int32_t j, k, st = last0 + 1, en = i;
				int32_t keep = (int32_t)((double)(pe - ps) / dist + .499);
				if (keep > 0) {
					if (keep > MAX_HIGH_OCC) keep = MAX_HIGH_OCC;
					for (j = st, k = 0; j < en && k < keep; ++j, ++k) b[k] = (uint64_t)a[j].n << 32 | (uint32_t)j;
					{ size_t q; for (q = (size_t)k >> 1; q-- > 0;) heap_down(b, q, k); }
					for (; j < en; ++j)
						if ((int32_t)a[j].n < (int32_t)(b[0] >> 32)) { b[0] = (uint64_t)a[j].n << 32 | (uint32_t)j; heap_down(b, 0, k); }
					for (j = 0; j < k; ++j) a[(uint32_t)b[j]].flt = 1;
				}
				for (j = st; j < en; ++j) a[j].flt ^= 1;
				for (j = st; j < en; ++j) if (a[j].n > (uint32_t)max_max_occ) a[j].flt = 1;
			}
			last0 = i;
		}
	}
}

/* one index hit r of seed q as an anchor; returns 0 when skip_seed (map.c:78-100) drops it */
static int make_anchor(ora128_t *p, uint64_t r, const seed_t *q, const char *qname, ora_seq_name_f seq_name, const void *idx, int64_t opt_flag, int qlen, int heap)
{
	const int32_t rpos = (int32_t)((uint32_t)r >> 1);
	const int fwd = (r & 1) == (q->q_pos & 1);
	int is_self = 0;
	if (qname && seq_name && (opt_flag & (F_NO_DIAG | F_NO_DUAL))) { /* all-vs-all rules (map.c:81-91) */
		uint32_t tl = 0;
		const char *tn = seq_name(idx, (uint32_t)(r >> 32), &tl);
		const int cmp = strcmp(qname, tn);
		if ((opt_flag & F_NO_DIAG) && cmp == 0 && (int)tl == qlen) {
			if ((uint32_t)r >> 1 == q->q_pos >> 1) return 0; /* the diagonal itself */
			if (fwd) is_self = 1;
		}
		if ((opt_flag & F_NO_DUAL) && cmp > 0) return 0; /* each pair once */
	}
	if (opt_flag & (F_FOR_ONLY | F_REV_ONLY)) {
		if (fwd && (opt_flag & F_REV_ONLY)) return 0;
		if (!fwd && (opt_flag & F_FOR_ONLY)) return 0;
	}
	if (fwd) {
		p->x = (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
		p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
	} else if (!(opt_flag & F_QSTRAND) || heap) {
		p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)rpos;
		p->y = (uint64_t)q->q_span << 32 | (uint32_t)(qlen - ((int32_t)(q->q_pos >> 1) + 1 - (int32_t)q->q_span) - 1);
	} else { /* query-strand mode (map.c:192-196): the reference coordinate is flipped instead of the query's */
		uint32_t tl = 0;
		seq_name(idx, (uint32_t)(r >> 32), &tl);
		p->x = 1ULL << 63 | (r & 0xffffffff00000000ULL) | (uint32_t)((int32_t)tl - (rpos + 1 - (int32_t)q->q_span) - 1);
		p->y = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
	}
	p->y |= (uint64_t)q->seg_id << 48;
	if (q->is_tandem) p->y |= SEED_TANDEM;
	if (is_self) p->y |= SEED_SELF;
	return 1;
}

/* min-heap on x only (heap_lt, map.c:75; ks_heapdown, ksort.h:43-53): the order of equal keys is whatever sifting leaves */
typedef struct { uint64_t x, y; } hp_t;
static void hp_down(hp_t *l, int64_t i, int64_t n)
{
	int64_t k = i;
	hp_t tmp = l[i];
	while ((k = (k << 1) + 1) < n) {
		if (k != n - 1 && l[k].x > l[k + 1].x) ++k;
		if (l[k].x > tmp.x) break;
		l[i] = l[k]; i = k;
	}
	l[i] = tmp;
}

int64_t ora_collect_seed_hits(const void *idx, ora_idx_get_f get, int64_t opt_flag, int qlen, int mid_occ, int max_max_occ, int occ_dist,
                              float q_occ_frac, ora128_t *mv, int64_t n_mv, ora128_t **anchors, int64_t *n_a_out,
                              uint64_t **mini_pos_out, int *n_mini_pos_out, int *rep_len_out)
{
	return ora_collect_seed_hits_named(idx, get, 0, 0, opt_flag, qlen, mid_occ, mid_occ, max_max_occ, occ_dist, q_occ_frac, mv, n_mv, anchors, n_a_out,
	                                   mini_pos_out, n_mini_pos_out, rep_len_out);
}

int64_t ora_collect_seed_hits_named(const void *idx, ora_idx_get_f get, const char *qname, ora_seq_name_f seq_name, int64_t opt_flag, int qlen,
                                    int q_mid_occ, int mid_occ, int max_max_occ, int occ_dist, float q_occ_frac, ora128_t *mv, int64_t n_mv,
                                    ora128_t **anchors, int64_t *n_a_out, uint64_t **mini_pos_out, int *n_mini_pos_out, int *rep_len_out)
{
	int64_t i, j, n_a = 0, k;
	int32_t n_m0 = 0, n_m = 0, rep_st = 0, rep_en = 0, rep_len = 0, n_mini_pos = 0;
	seed_t *m;
	uint64_t *mini_pos;
	ora128_t *a;

	/* query-side filter of over-represented minimizers (mm_seed_mz_flt, seed.c:5-28) */
	if (q_occ_frac > 0.0f && n_mv > q_mid_occ && q_mid_occ > 0) {
		ora128_t *s = (ora128_t*)malloc(n_mv * sizeof(ora128_t));
		int64_t st;
		for (i = 0; i < n_mv; ++i) s[i].x = mv[i].x, s[i].y = (uint64_t)i;
		ora_radix_sort_128x(s, s + n_mv);
		for (st = 0, i = 1; i <= n_mv; ++i) {
			if (i == n_mv || s[i].x != s[st].x) {
				int32_t cnt = (int32_t)(i - st);
				if (cnt > q_mid_occ && cnt > n_mv * q_occ_frac)
					for (j = st; j < i; ++j) mv[s[j].y].x = 0;
				st = i;
			}
		}
		free(s);
		for (i = j = 0; i < n_mv; ++i) if (mv[i].x != 0) mv[j++] = mv[i];
		n_mv = j;
	}
	/* look every minimizer up (mm_seed_collect_all, seed.c:30-52) */
	m = (seed_t*)malloc((n_mv ? n_mv : 1) * sizeof(seed_t));
	mini_pos = (uint64_t*)malloc((n_mv ? n_mv : 1) * sizeof(uint64_t));
	for (i = 0; i < n_mv; ++i) {
		int t;
		const uint64_t *cr = get(idx, mv[i].x >> 8, &t);
		seed_t *q;
		if (t == 0) continue;
		q = &m[n_m0++];
		q->q_pos = (uint32_t)mv[i].y, q->q_span = mv[i].x & 0xff, q->cr = cr, q->n = (uint32_t)t, q->seg_id = (uint32_t)(mv[i].y >> 32);
		q->is_tandem = q->flt = 0;
		if (i > 0 && mv[i].x >> 8 == mv[i - 1].x >> 8) q->is_tandem = 1;
		if (i < n_mv - 1 && mv[i].x >> 8 == mv[i + 1].x >> 8) q->is_tandem = 1;
	}
	/* occurrence filter (seed.c:106-112) */
	if (occ_dist > 0 && max_max_occ > mid_occ) thin_high_occ(n_m0, m, qlen, mid_occ, max_max_occ, occ_dist);
	else for (i = 0; i < n_m0; ++i) if (m[i].n > (uint32_t)mid_occ) m[i].flt = 1;
	/* repetitive length and surviving seeds (seed.c:113-131) */
	for (i = 0; i < n_m0; ++i) {
		seed_t *q = &m[i];
		if (q->flt) {
			int en = (int)(q->q_pos >> 1) + 1, st = en - (int)q->q_span;
			if (st > rep_en) rep_len += rep_en - rep_st, rep_st = st, rep_en = en;
			else rep_en = en;
		} else {
			n_a += q->n;
			mini_pos[n_mini_pos++] = (uint64_t)q->q_span << 32 | q->q_pos >> 1;
			m[n_m++] = *q;
		}
	}
	rep_len += rep_en - rep_st;
	/* expand to anchors and order them by target position: collect_seed_hits (map.c:168-204) or, with MM_F_HEAP_SORT,
	 * collect_seed_hits_heap (map.c:102-166) */
	a = (ora128_t*)malloc((n_a ? n_a : 1) * sizeof(ora128_t));
	if (opt_flag & F_HEAP_SORT) {
		hp_t *heap = (hp_t*)malloc((n_m ? n_m : 1) * sizeof(hp_t));
		int64_t n_for = 0, n_rev = 0, hs = 0;
		for (i = 0; i < n_m; ++i)
			if (m[i].n > 0) heap[hs].x = m[i].cr[0], heap[hs].y = (uint64_t)i << 32, ++hs;
		if (hs > 1) for (i = (hs >> 1) - 1; i >= 0; --i) hp_down(heap, i, hs);
		while (hs > 0) {
			const seed_t *q = &m[heap[0].y >> 32];
			ora128_t t;
			if (make_anchor(&t, heap[0].x, q, qname, seq_name, idx, opt_flag, qlen, 1) /* the heap path has no query-strand branch (map.c:129-137) */) {
				if (t.x >> 63) a[n_a - (++n_rev)] = t; /* the other strand is laid down back to front ... */
				else a[n_for++] = t;
			}
			if ((uint32_t)heap[0].y < q->n - 1) ++heap[0].y, heap[0].x = q->cr[(uint32_t)heap[0].y];
			else heap[0] = heap[hs - 1], --hs;
			hp_down(heap, 0, hs);
		}
		free(heap);
		for (j = 0; j < n_rev >> 1; ++j) { /* ... and turned around afterwards (map.c:155-160) */
			ora128_t t = a[n_a - 1 - j];
			a[n_a - 1 - j] = a[n_a - (n_rev - j)];
			a[n_a - (n_rev - j)] = t;
		}
		if (n_a > n_for + n_rev) memmove(a + n_for, a + n_a - n_rev, n_rev * sizeof(ora128_t));
		k = n_for + n_rev;
	} else {
		for (i = 0, k = 0; i < n_m; ++i) {
			const seed_t *q = &m[i];
			uint32_t c;
			for (c = 0; c < q->n; ++c)
				if (make_anchor(&a[k], q->cr[c], q, qname, seq_name, idx, opt_flag, qlen, 0)) ++k;
		}
		ora_radix_sort_128x(a, a + k);
	}
	free(m);
	*anchors = a, *n_a_out = k, *mini_pos_out = mini_pos, *n_mini_pos_out = n_mini_pos, *rep_len_out = rep_len;
	return n_mv;
}

/* oracle/refdrv.c -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/librefdrv.so, never linked into the product).
 *
 * Two helpers around the UNMODIFIED reference library (oracle/_ref/libminimap2_ref.a), compiled against the reference's own
 * headers where they lie under $(REF):
 *
 *   refdrv_idx_from_flat : wraps an already computed (minimizer -> ascending positions) table into a genuine mm_idx_t, using
 *       the reference's own khash instantiation for the buckets, so that the reference's mm_map() can be timed and compared on
 *       large references without spending minutes in its (largely single-threaded) index construction.  The table contents are
 *       what mm_idx_gen would have produced (tests/test_gpu_aligner.py checks our device-built tables against mm_idx_str), and
 *       mm_idx_get (index.c:93-110) returns the same (n, list) whatever the internal slot order is.
 *   refdrv_map          : kt_for over reads calling the reference's mm_map() (map.c:380-392) with one mm_tbuf_t per thread --
 *       the work worker_for does for single-segment reads (map.c:425-474) -- and reports the wall time of that region.
 *
 * index.c keeps mm_idx_bucket_t and its khash instantiation private (index.c:19-33); the five declarations below repeat them,
 * as SURVEY.md section 8(b) anticipates for any code that has to look inside an mm_idx_t. */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <sys/time.h>
#include "minimap.h"
#include "mmpriv.h"
#include "kalloc.h"
#include "khash.h"
#include "kthread.h"

#define idx_hash(a) ((a)>>1)
#define idx_eq(a, b) ((a)>>1 == (b)>>1)
KHASH_INIT(idx, uint64_t, uint64_t, 1, idx_hash, idx_eq)
typedef khash_t(idx) idxhash_t;
typedef struct mm_idx_bucket_s { mm128_v a; int32_t n; uint64_t *p; void *h; } mm_idx_bucket_t;
KHASH_MAP_INIT_STR(str, uint32_t)

extern mm_idx_t *mm_idx_init(int w, int k, int b, int flag); /* index.c:53 */

static double now_s(void) { struct timeval tv; gettimeofday(&tv, 0); return tv.tv_sec + tv.tv_usec * 1e-6; }

typedef struct {
	mm_idx_t *mi;
	const uint64_t *keys, *pos;
	const uint32_t *val_off;
	uint64_t n_keys;
	uint64_t *bkt_start; /* (1<<b)+1 */
	uint32_t *bkt_item;  /* key indices grouped by bucket */
} build_t;

static void build_bucket(void *data, long i, int tid)
{
	build_t *d = (build_t*)data;
	mm_idx_t *mi = d->mi;
	mm_idx_bucket_t *b = &mi->B[i];
	uint64_t s = d->bkt_start[i], e = d->bkt_start[i + 1], j, n_p = 0, off = 0;
	idxhash_t *h;
	(void)tid;
	if (e == s) return;
	for (j = s; j < e; ++j) {
		uint32_t ki = d->bkt_item[j], n = d->val_off[ki + 1] - d->val_off[ki];
		if (n > 1) n_p += n;
	}
	h = kh_init(idx);
	kh_resize(idx, h, (khint_t)(e - s));
	b->h = h, b->n = (int32_t)n_p;
	b->p = n_p ? (uint64_t*)calloc(n_p, 8) : 0;
	for (j = s; j < e; ++j) {
		uint32_t ki = d->bkt_item[j], n = d->val_off[ki + 1] - d->val_off[ki];
		int absent;
		khint_t itr = kh_put(idx, h, d->keys[ki] >> mi->b << 1, &absent);
		if (n == 1) {
			kh_key(h, itr) |= 1;
			kh_val(h, itr) = d->pos[d->val_off[ki]];
		} else {
			memcpy(&b->p[off], &d->pos[d->val_off[ki]], (size_t)n * 8);
			kh_val(h, itr) = off << 32 | n;
			off += n;
		}
	}
}

mm_idx_t *refdrv_idx_from_flat(int w, int k, int flag, int bucket_bits, uint32_t n_seq, const char **names, const uint32_t *lens, const uint32_t *S,
                               uint64_t n_keys, const uint64_t *keys, const uint32_t *val_off, const uint64_t *pos, int n_threads)
{
	mm_idx_t *mi = mm_idx_init(w, k, bucket_bits < 0 ? 14 : bucket_bits, flag);
	uint64_t sum_len = 0, i, nb, mask;
	khash_t(str) *h;
	build_t d;
	mi->n_seq = n_seq;
	mi->seq = (mm_idx_seq_t*)kcalloc(mi->km, n_seq, sizeof(mm_idx_seq_t));
	mi->h = h = kh_init(str);
	for (i = 0; i < n_seq; ++i) {
		mm_idx_seq_t *p = &mi->seq[i];
		if (names && names[i]) {
			int absent;
			khint_t itr;
			p->name = (char*)kmalloc(mi->km, strlen(names[i]) + 1);
			strcpy(p->name, names[i]);
			itr = kh_put(str, h, p->name, &absent);
			kh_val(h, itr) = (uint32_t)i;
		}
		p->offset = sum_len, p->len = lens[i], p->is_alt = 0;
		sum_len += lens[i];
	}
	mi->S = (uint32_t*)calloc((sum_len + 7) / 8, 4);
	memcpy(mi->S, S, (sum_len + 7) / 8 * 4);
	nb = 1ULL << mi->b, mask = nb - 1;
	d.mi = mi, d.keys = keys, d.pos = pos, d.val_off = val_off, d.n_keys = n_keys;
	d.bkt_start = (uint64_t*)calloc(nb + 1, 8);
	d.bkt_item = (uint32_t*)malloc((n_keys ? n_keys : 1) * 4);
	for (i = 0; i < n_keys; ++i) ++d.bkt_start[(keys[i] & mask) + 1];
	for (i = 0; i < nb; ++i) d.bkt_start[i + 1] += d.bkt_start[i];
	{
		uint64_t *cur = (uint64_t*)malloc(nb * 8);
		memcpy(cur, d.bkt_start, nb * 8);
		for (i = 0; i < n_keys; ++i) d.bkt_item[cur[keys[i] & mask]++] = (uint32_t)i;
		free(cur);
	}
	kt_for(n_threads > 0 ? n_threads : 1, build_bucket, &d, (long)nb);
	free(d.bkt_start); free(d.bkt_item);
	return mi;
}

typedef struct {
	const mm_idx_t *mi;
	const mm_mapopt_t *opt;
	const char **seqs, **names;
	const int *lens;
	int *n_reg;
	mm_reg1_t **reg;
	mm_tbuf_t **tbuf;
} map_t;

static void map_one(void *data, long i, int tid)
{
	map_t *m = (map_t*)data;
	m->reg[i] = mm_map(m->mi, m->lens[i], m->seqs[i], &m->n_reg[i], m->tbuf[tid], m->opt, m->names ? m->names[i] : 0);
}

/* returns the wall-clock seconds of the mapping loop; reg[i] / reg[i][j].p are libc blocks owned by the caller */
double refdrv_map(const mm_idx_t *mi, const mm_mapopt_t *opt, int n_reads, const char **seqs, const int *lens, const char **names,
                  int n_threads, int *n_reg, mm_reg1_t **reg)
{
	map_t m;
	int t;
	double t0;
	if (n_threads < 1) n_threads = 1;
	m.mi = mi, m.opt = opt, m.seqs = seqs, m.names = names, m.lens = lens, m.n_reg = n_reg, m.reg = reg;
	m.tbuf = (mm_tbuf_t**)calloc(n_threads, sizeof(mm_tbuf_t*));
	for (t = 0; t < n_threads; ++t) m.tbuf[t] = mm_tbuf_init();
	t0 = now_s();
	kt_for(n_threads, map_one, &m, n_reads);
	t0 = now_s() - t0;
	for (t = 0; t < n_threads; ++t) mm_tbuf_destroy(m.tbuf[t]);
	free(m.tbuf);
	return t0;
}

/* Read pairs: what worker_for (map.c:425-474) does for a two-segment fragment -- reverse-complement a mate as pe_ori says,
 * mm_map_frag() on the two reads, turn the hits of a flipped mate back -- on copies of the reads.
 * seqs / lens / n_reg / reg hold 2 * n_pairs entries (mates adjacent), names n_pairs. */
static void map_pair(void *data, long i, int tid)
{
	map_t *m = (map_t*)data;
	const int pe_ori = m->opt->pe_ori;
	int j, qlens[2];
	const char *qseqs[2];
	mm_bseq1_t t[2];
	for (j = 0; j < 2; ++j) {
		memset(&t[j], 0, sizeof(mm_bseq1_t));
		t[j].l_seq = m->lens[2 * i + j];
		t[j].seq = (char*)malloc(t[j].l_seq + 1);
		memcpy(t[j].seq, m->seqs[2 * i + j], t[j].l_seq);
		t[j].seq[t[j].l_seq] = 0;
		if ((j == 0 && (pe_ori >> 1 & 1)) || (j == 1 && (pe_ori & 1))) mm_revcomp_bseq(&t[j]);
		qlens[j] = t[j].l_seq, qseqs[j] = t[j].seq;
	}
	mm_map_frag(m->mi, 2, qlens, qseqs, &m->n_reg[2 * i], &m->reg[2 * i], m->tbuf[tid], m->opt, m->names ? m->names[i] : 0);
	for (j = 0; j < 2; ++j) {
		if ((j == 0 && (pe_ori >> 1 & 1)) || (j == 1 && (pe_ori & 1))) {
			int k;
			for (k = 0; k < m->n_reg[2 * i + j]; ++k) {
				mm_reg1_t *r = &m->reg[2 * i + j][k];
				int s = r->qs;
				r->qs = qlens[j] - r->qe, r->qe = qlens[j] - s, r->rev = !r->rev;
				if (r->p) {
					if (r->p->trans_strand == 1) r->p->trans_strand = 2;
					else if (r->p->trans_strand == 2) r->p->trans_strand = 1;
				}
			}
		}
		free(t[j].seq);
	}
}

double refdrv_map_pairs(const mm_idx_t *mi, const mm_mapopt_t *opt, int n_pairs, const char **seqs, const int *lens, const char **names,
                        int n_threads, int *n_reg, mm_reg1_t **reg)
{
	map_t m;
	int t;
	double t0;
	if (n_threads < 1) n_threads = 1;
	m.mi = mi, m.opt = opt, m.seqs = seqs, m.names = names, m.lens = lens, m.n_reg = n_reg, m.reg = reg;
	m.tbuf = (mm_tbuf_t**)calloc(n_threads, sizeof(mm_tbuf_t*));
	for (t = 0; t < n_threads; ++t) m.tbuf[t] = mm_tbuf_init();
	t0 = now_s();
	kt_for(n_threads, map_pair, &m, n_pairs);
	t0 = now_s() - t0;
	for (t = 0; t < n_threads; ++t) mm_tbuf_destroy(m.tbuf[t]);
	free(m.tbuf);
	return t0;
}

/* ---- order-independent digests of a minimizer index, to compare a 3 Gb index built by the reference's own mm_idx_gen with the
 * tables the device builder produced without holding both in Python: sum over the distinct minimizers of a hash of
 * (minimizer, its positions in ascending order), plus the counts.  refdrv_idx_digest walks the reference's private buckets
 * (index.c:19-33, :93-110); refdrv_flat_digest does the same over flat (keys, val_off, pos) tables. ---- */
static inline uint64_t dg_mix(uint64_t h) { h ^= h >> 33; h *= 0xff51afd7ed558ccdULL; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ULL; h ^= h >> 33; return h; }

typedef struct { const mm_idx_t *mi; uint64_t *sum, *n_keys, *n_pos; } dg_ref_t;

static void dg_ref_bucket(void *data, long i, int tid)
{
	dg_ref_t *d = (dg_ref_t*)data;
	const mm_idx_bucket_t *b = &d->mi->B[i];
	idxhash_t *h = (idxhash_t*)b->h;
	khint_t k;
	uint64_t sum = 0, nk = 0, np = 0;
	(void)tid;
	if (h == 0) return;
	for (k = 0; k < kh_end(h); ++k) {
		uint64_t key, acc;
		if (!kh_exist(h, k)) continue;
		key = (kh_key(h, k) >> 1) << d->mi->b | (uint64_t)i; /* the minimizer hash: bucket index in the low bits (index.c:98) */
		acc = dg_mix(key);
		if (kh_key(h, k) & 1) acc = dg_mix(acc ^ kh_val(h, k)), ++np;
		else {
			const uint64_t *p = &b->p[kh_val(h, k) >> 32];
			uint32_t n = (uint32_t)kh_val(h, k), j;
			for (j = 0; j < n; ++j) acc = dg_mix(acc ^ p[j]);
			np += n;
		}
		sum += acc, ++nk;
	}
	__sync_fetch_and_add(d->sum, sum), __sync_fetch_and_add(d->n_keys, nk), __sync_fetch_and_add(d->n_pos, np);
}

void refdrv_idx_digest(const mm_idx_t *mi, int n_threads, uint64_t out[3])
{
	dg_ref_t d;
	out[0] = out[1] = out[2] = 0;
	d.mi = mi, d.sum = &out[0], d.n_keys = &out[1], d.n_pos = &out[2];
	kt_for(n_threads, dg_ref_bucket, &d, 1L << mi->b);
}

typedef struct { const uint64_t *keys, *pos; const uint32_t *val_off; uint64_t n_keys, *sum; } dg_flat_t;

static void dg_flat_chunk(void *data, long c, int tid)
{
	dg_flat_t *d = (dg_flat_t*)data;
	uint64_t i, e = ((uint64_t)c + 1) << 20, sum = 0;
	(void)tid;
	if (e > d->n_keys) e = d->n_keys;
	for (i = (uint64_t)c << 20; i < e; ++i) {
		uint64_t acc = dg_mix(d->keys[i]);
		uint32_t j;
		for (j = d->val_off[i]; j < d->val_off[i + 1]; ++j) acc = dg_mix(acc ^ d->pos[j]);
		sum += acc;
	}
	__sync_fetch_and_add(d->sum, sum);
}

void refdrv_flat_digest(uint64_t n_keys, const uint64_t *keys, const uint32_t *val_off, const uint64_t *pos, int n_threads, uint64_t out[3])
{
	dg_flat_t d;
	out[0] = 0, out[1] = n_keys, out[2] = n_keys ? val_off[n_keys] : 0;
	d.keys = keys, d.pos = pos, d.val_off = val_off, d.n_keys = n_keys, d.sum = &out[0];
	kt_for(n_threads, dg_flat_chunk, &d, (long)((n_keys + (1 << 20) - 1) >> 20));
}

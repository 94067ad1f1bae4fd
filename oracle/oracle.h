/* oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, scalar restatement of the reference's hot-path kernels (lh3/minimap2 v2.30). Nothing in the
 * product library (minimap2_amd/) may include, link or call this; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do, and only as the checker.
 *
 * Parity status: PINNED. Every function here is fuzz-checked against the compiled, unmodified reference
 * (oracle/_ref/libminimap2_ref.so, built by oracle/Makefile from /root/reference) in tests/test_oracle_*.py;
 * the reference ships no golden vectors of its own for this path.
 */
#ifndef MM2AMD_ORACLE_H
#define MM2AMD_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* flag bits, same numeric values as the reference's KSW_EZ_* (ksw2.h:8-19) */
#define ORA_EZ_SCORE_ONLY   0x01
#define ORA_EZ_RIGHT        0x02
#define ORA_EZ_GENERIC_SC   0x04
#define ORA_EZ_APPROX_MAX   0x08
#define ORA_EZ_APPROX_DROP  0x10
#define ORA_EZ_EXTZ_ONLY    0x40
#define ORA_EZ_REV_CIGAR    0x80
#define ORA_EZ_SPLICE_FOR   0x100
#define ORA_EZ_SPLICE_REV   0x200
#define ORA_EZ_SPLICE_FLANK 0x400
#define ORA_EZ_SPLICE_CMPLX 0x800
#define ORA_EZ_SPLICE_SCORE 0x1000
#define ORA_NEG_INF         (-0x40000000)

/* result of one extension/global DP; field meaning as ksw_extz_t (ksw2.h:34-43) */
typedef struct {
	int32_t max, zdropped;
	int32_t max_q, max_t;
	int32_t mqe, mqe_t;
	int32_t mte, mte_q;
	int32_t score;
	int32_t n_cigar;
	int32_t reach_end;
	int32_t cigar_overflow;   /* set if cigar_cap was too small (n_cigar then counts what would be needed) */
} ora_ez_t;

typedef struct { uint64_t x, y; } ora128_t;

/* ksw2_extd2_sse.c:34-401, lane-exact (16-lane block garbage included) */
void ora_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                   ora_ez_t *ez, uint32_t *cigar, int cigar_cap);

/* ksw2_extz2_sse.c:25-311 (single-affine), lane-exact, SSE4.1 code path */
void ora_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ora_ez_t *ez, uint32_t *cigar, int cigar_cap);

/* ksw2_exts2_sse.c:33-465 (splice-aware), lane-exact, SSE4.1 code path; junc may be NULL */
void ora_ksw_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int end_bonus, int8_t junc_bonus, int8_t junc_pen, int flag,
                   const uint8_t *junc, ora_ez_t *ez, uint32_t *cigar, int cigar_cap);

/* mm_sketch, sketch.c:77-143 (non-HPC and HPC).  Appends to out[*n_out..cap); returns the number of
 * minimizers the sequence has (which may exceed cap - then only the first cap were stored). */
int64_t ora_sketch(const char *seq, int len, int w, int k, uint32_t rid, int is_hpc, ora128_t *out, int64_t cap);

/* radix_sort_128x / radix_sort_64, ksort.h:101-151 + misc.c:155-159: in-place UNSTABLE MSD radix sort with
 * insertion sort below 65 elements; the permutation of equal keys is reproduced exactly. */
void ora_radix_sort_128x(ora128_t *beg, ora128_t *end);
void ora_radix_sort_64(uint64_t *beg, uint64_t *end);

/* mg_lchain_dp fill only, lchain.c:169-207: f,p,v for sorted anchors a[0..n) (t is scratch, zeroed here). */
void ora_lchain_fill(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, float chn_pen_gap, float chn_pen_skip,
                     int is_cdna, int n_seg, int64_t n, const ora128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t);

/* mg_lchain_dp, lchain.c:148-217 (fill + mg_chain_backtrack :27-76 + compact_a :78-111).
 * a[] is overwritten with the compacted anchors; u_out (capacity n) receives score<<32|cnt per chain.
 * Returns n_u; *n_a_out = number of anchors kept. */
int ora_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                  float chn_pen_gap, float chn_pen_skip, int is_cdna, int n_seg, int64_t n, ora128_t *a,
                  uint64_t *u_out, int64_t *n_a_out);

/* index lookup callback with mm_idx_get semantics (index.c:93-110): returns the ascending position list and its length */
typedef const uint64_t *(*ora_idx_get_f)(const void *idx, uint64_t minier, int *n);

/* Seeding of one read: mm_seed_mz_flt (seed.c:5-28) + mm_collect_matches (seed.c:98-132, incl. mm_seed_collect_all
 * :30-52 and mm_seed_select :56-96) + the anchor expansion and sort of collect_seed_hits (map.c:168-204; skip_seed
 * :78-100; the qname-dependent NO_DIAG/NO_DUAL rules need the _named variant).
 * mv[0..n_mv) are the read's minimizers (modified in place by the query-occurrence filter).
 * Outputs: anchors (malloc'd, *n_a entries, sorted with the reference's unstable sort), mini_pos (malloc'd,
 * *n_mini_pos entries), *rep_len.  Returns the number of minimizers left after the filter. */
int64_t ora_collect_seed_hits(const void *idx, ora_idx_get_f get, int64_t opt_flag, int qlen, int mid_occ, int max_max_occ, int occ_dist,
                              float q_occ_frac, ora128_t *mv, int64_t n_mv, ora128_t **anchors, int64_t *n_a,
                              uint64_t **mini_pos, int *n_mini_pos, int *rep_len);
/* the same with the all-vs-all rules of skip_seed (map.c:81-91): qname is the read's name, seq_name returns the name and length
 * of reference sequence rid.  Either may be null (then the rules are off, as in the reference when qname is null).
 * q_mid_occ is the threshold of the query-side filter, mid_occ that of the index-side one: they differ in the second seeding
 * pass of map.c:311, which raises the latter to max_occ but works on the minimizers the first pass already filtered (:251). */
typedef const char *(*ora_seq_name_f)(const void *idx, uint32_t rid, uint32_t *len);
int64_t ora_collect_seed_hits_named(const void *idx, ora_idx_get_f get, const char *qname, ora_seq_name_f seq_name, int64_t opt_flag, int qlen,
                                    int q_mid_occ, int mid_occ, int max_max_occ, int occ_dist, float q_occ_frac, ora128_t *mv, int64_t n_mv,
                                    ora128_t **anchors, int64_t *n_a, uint64_t **mini_pos, int *n_mini_pos, int *rep_len);

#ifdef __cplusplus
}
#endif
#endif

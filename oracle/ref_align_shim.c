/* TEST INFRASTRUCTURE ONLY (oracle/Makefile -> oracle/_ref/librefalign.so; never linked into the product).
 *
 * The reference's mm_append_cigar / mm_fix_cigar / mm_update_extra (align.c:320-334, :105-181, :254-303) are `static`: to pin the
 * device's region_finish_kernel against the UNMODIFIED reference on CIGARs no DP would emit (empty operations, I/D clusters, leading
 * gaps, indels that left-align through whole matches), this translation unit compiles the reference's align.c where it lies
 * (-I$(REF); nothing is copied) and adds one entry point that calls those functions.  The two external functions align.c defines are
 * renamed so that the shim can sit beside libminimap2_ref.a. */
#define mm_enlarge_cigar refshim_mm_enlarge_cigar
#define mm_align_skeleton refshim_mm_align_skeleton
#include "align.c"

/* pieces: a region's window CIGARs in alignment order.  Returns the number of CIGAR operations left; res8 = { blen, mlen, n_ambi, dp_max,
 * qshift, tshift, is_spliced, capacity }. */
int refshim_update_extra(int n_pieces, const uint32_t *const *pieces, const int32_t *piece_len, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq,
                         const int8_t *mat, int gq, int ge, int log_gap, uint32_t *cigar_out, int32_t *res8)
{
	mm_reg1_t r;
	int i, n;
	memset(&r, 0, sizeof r);
	r.qs = 0, r.qe = qlen, r.rs = 0, r.re = tlen, r.rev = 0;
	for (i = 0; i < n_pieces; ++i) mm_append_cigar(&r, (uint32_t)piece_len[i], pieces[i]);
	memset(res8, 0, 8 * sizeof(int32_t));
	if (r.p == 0) return 0;
	mm_update_extra(&r, qseq, tseq, mat, (int8_t)gq, (int8_t)ge, 0, log_gap);
	n = (int)r.p->n_cigar;
	memcpy(cigar_out, r.p->cigar, (size_t)n * 4);
	res8[0] = r.blen, res8[1] = r.mlen, res8[2] = (int32_t)r.p->n_ambi, res8[3] = r.p->dp_max, res8[4] = r.qs, res8[5] = r.rs, res8[6] = r.is_spliced, res8[7] = (int32_t)r.p->capacity;
	free(r.p);
	return n;
}

/* The same with MM_F_EQX's mm_update_cigar_eqx (align.c:183-252, also static) applied: matches cut into = and X stretches, which can leave MORE operations
 * than came in -- cigar_out holds cigar_cap words; returns -(operations) if that is too few. */
int refshim_update_extra_eqx(int n_pieces, const uint32_t *const *pieces, const int32_t *piece_len, int qlen, const uint8_t *qseq, int tlen, const uint8_t *tseq,
                             const int8_t *mat, int gq, int ge, int log_gap, uint32_t *cigar_out, int cigar_cap, int32_t *res8)
{
	mm_reg1_t r;
	int i, n;
	memset(&r, 0, sizeof r);
	r.qs = 0, r.qe = qlen, r.rs = 0, r.re = tlen, r.rev = 0;
	for (i = 0; i < n_pieces; ++i) mm_append_cigar(&r, (uint32_t)piece_len[i], pieces[i]);
	memset(res8, 0, 8 * sizeof(int32_t));
	if (r.p == 0) return 0;
	mm_update_extra(&r, qseq, tseq, mat, (int8_t)gq, (int8_t)ge, 1, log_gap);
	n = (int)r.p->n_cigar;
	if (n > cigar_cap) { free(r.p); return -n; }
	memcpy(cigar_out, r.p->cigar, (size_t)n * 4);
	res8[0] = r.blen, res8[1] = r.mlen, res8[2] = (int32_t)r.p->n_ambi, res8[3] = r.p->dp_max, res8[4] = r.qs, res8[5] = r.rs, res8[6] = r.is_spliced, res8[7] = (int32_t)r.p->capacity;
	free(r.p);
	return n;
}

/* Round 6: the chain -> window rules of mm_align1 that are static functions of their own (align.c:454-561), for tests/cpucheck/region_rules_test.cpp --
 * minimap2_amd/csrc/region_rules.hpp is pinned to them routine by routine.  a: the read's anchors (mm128_t), modified in place like the reference does. */
void refshim_filter_bad_seeds(int as1, int cnt1, void *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt)
{
	mm_filter_bad_seeds(0, as1, cnt1, (mm128_t*)a, min_gap, diff_thres, max_ext_len, max_ext_cnt);
}
void refshim_filter_bad_seeds_alt(int as1, int cnt1, void *a, int min_gap, int max_ext)
{
	mm_filter_bad_seeds_alt(0, as1, cnt1, (mm128_t*)a, min_gap, max_ext);
}
void refshim_fix_bad_ends(int as, int cnt, int mlen, const void *a, int bw, int min_match, int32_t *as1, int32_t *cnt1)
{
	mm_reg1_t r;
	memset(&r, 0, sizeof r);
	r.as = as, r.cnt = cnt, r.mlen = mlen;
	mm_fix_bad_ends(&r, (const mm128_t*)a, bw, min_match, as1, cnt1);
}
/* mm_append_cigar on its own (align.c:320-334): head then tail; returns the operations of the result */
int refshim_append_cigar(int n_head, const uint32_t *head, int n_tail, const uint32_t *tail, uint32_t *out)
{
	mm_reg1_t r;
	int n;
	memset(&r, 0, sizeof r);
	mm_append_cigar(&r, (uint32_t)n_head, head);
	mm_append_cigar(&r, (uint32_t)n_tail, tail);
	if (r.p == 0) return 0;
	n = (int)r.p->n_cigar;
	memcpy(out, r.p->cigar, (size_t)n * 4);
	free(r.p);
	return n;
}
void refshim_max_stretch(int as, int cnt, const void *a, int32_t *as1, int32_t *cnt1)
{
	mm_reg1_t r;
	memset(&r, 0, sizeof r);
	r.as = as, r.cnt = cnt;
	mm_max_stretch(&r, (const mm128_t*)a, as1, cnt1);
}

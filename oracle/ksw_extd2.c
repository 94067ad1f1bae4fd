/* oracle/ksw_extd2.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar, lane-exact restatement of the reference's dual-affine extension DP
 * (ksw_extd2_sse, /root/reference/ksw2_extd2_sse.c:34-401) and of its traceback
 * (ksw_backtrack, ksw2.h:130-162; ksw_apply_zdrop, ksw2.h:171-187).
 *
 * "Lane-exact" means: the reference evaluates each anti-diagonal in 16-lane blocks over the
 * block-aligned interval [st,en] that encloses the valid interval [st0,en0]; lanes outside
 * [st0,en0] are computed from stale score bytes and old state, and with a binding band those
 * values are read by valid cells (hard part 1). We reproduce the same byte
 * arrays (u,v,x,y,x2,y2 | s | sf | qr contiguous as at ksw2_extd2_sse.c:107-110), the same
 * 16-byte chunked score fill (:166-180) with its overshoot, and the same mod-256 arithmetic.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static inline int8_t w8(int v) { return (int8_t)(uint8_t)v; } /* mod-256 wrap, like _mm_add/sub_epi8 */

typedef struct {
	int qlen, tlen, T16, ncol;      /* T16 = 16*ceil(tlen/16); ncol = bytes per row of the direction matrix */
	int8_t *u, *v, *x, *y, *x2, *y2, *s;
	uint8_t *sf, *qr;               /* target copy, reversed query; contiguous after s (overshoot relies on it) */
	uint8_t *dir;                   /* direction matrix, row r at dir + r*ncol, lane t at column t - off[r] */
	int *off, *off_end;
} dp_t;

static void ez_reset(ora_ez_t *ez) /* ksw_reset_extz, ksw2.h:164-169 */
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = ORA_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0, ez->cigar_overflow = 0;
}

static int zdrop_test(ora_ez_t *ez, int32_t H, int r, int t, int zdrop, int8_t e) /* ksw_apply_zdrop is_rot=1 */
{
	if (H > ez->max) {
		ez->max = H, ez->max_t = t, ez->max_q = r - t;
	} else if (t >= ez->max_t && r - t >= ez->max_q) {
		int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
	}
	return 0;
}

typedef struct { uint32_t *c; int n, cap, ovf; uint32_t last_op; } cig_t;

static void cig_push(cig_t *g, uint32_t op, int len) /* ksw_push_cigar, ksw2.h:114-124 (caller-provided buffer) */
{
	if (g->n == 0 || op != g->last_op) {
		if (g->n < g->cap) g->c[g->n] = (uint32_t)len << 4 | op; else g->ovf = 1;
		++g->n, g->last_op = op;
	} else if (g->n <= g->cap) g->c[g->n - 1] += (uint32_t)len << 4;
}

/* traceback over the rotated direction matrix; ksw2.h:130-162 with is_rot=1 */
static void traceback(const dp_t *d, int is_rev, int min_intron_len, int i0, int j0, cig_t *g)
{
	int i = i0, j = j0, state = 0;
	while (i >= 0 && j >= 0) {
		int r = i + j, force = -1, tmp;
		if (i < d->off[r]) force = 2;
		if (i > d->off_end[r]) force = 1;
		tmp = force < 0 ? d->dir[(size_t)r * d->ncol + (i - d->off[r])] : 0;
		if (state == 0) state = tmp & 7;
		else if (!(tmp >> (state + 2) & 1)) state = 0;
		if (state == 0) state = tmp & 7;
		if (force >= 0) state = force;
		if (state == 0) cig_push(g, 0, 1), --i, --j;
		else if (state == 1 || (state == 3 && min_intron_len <= 0)) cig_push(g, 2, 1), --i;
		else if (state == 3 && min_intron_len > 0) cig_push(g, 3, 1), --i;
		else cig_push(g, 1, 1), --j;
	}
	if (i >= 0) cig_push(g, min_intron_len > 0 && i >= min_intron_len ? 3 : 2, i + 1);
	if (j >= 0) cig_push(g, 1, j + 1);
	if (!is_rev) {
		int k, n = g->n < g->cap ? g->n : g->cap;
		if (!g->ovf) for (k = 0; k < n >> 1; ++k) { uint32_t t = g->c[k]; g->c[k] = g->c[n - 1 - k]; g->c[n - 1 - k] = t; }
	}
}

void ora_ksw_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                   ora_ez_t *ez, uint32_t *cigar, int cigar_cap)
{
	const int with_cigar = !(flag & ORA_EZ_SCORE_ONLY), approx_max = !!(flag & ORA_EZ_APPROX_MAX);
	const int right = !!(flag & ORA_EZ_RIGHT);
	int r, t, qe, qe2, qe_in = q + e /* :68 - taken BEFORE the swap at :78; seeds H(0,0) */, Q16, long_thres, long_diff, last_st = -1, last_en = -1, min_sc;
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int8_t sc_mch, sc_mis, sc_N;
	uint8_t *mem;
	dp_t d;

	ez_reset(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0) return;
	if (q2 + e2 < q + e) { int8_t z; z = q, q = q2, q2 = z; z = e, e = e2, e2 = z; } /* :78 */
	qe = q + e, qe2 = q2 + e2;
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? w8(-e2) : mat[m * m - 1];                              /* :87 */
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	d.qlen = qlen, d.tlen = tlen;
	d.T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
	d.ncol = qlen < tlen ? qlen : tlen;
	d.ncol = (((d.ncol < w + 1 ? d.ncol : w + 1) + 15) / 16 + 1) * 16;                 /* :94-95 (n_col_*16) */
	for (t = 1, min_sc = mat[1]; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
	if (-min_sc > 2 * (q + e)) return;                                                  /* :101 */

	long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;                                 /* :103-106 */
	if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * (e - e2) - (q2 - q) - e2;

	/* one zero-filled block: u v x y x2 y2 s | sf | qr   (:107-110; kcalloc zero-fills) */
	mem = (uint8_t*)calloc((size_t)d.T16 * 8 + Q16 + 16, 1);
	d.u = (int8_t*)mem, d.v = d.u + d.T16, d.x = d.v + d.T16, d.y = d.x + d.T16, d.x2 = d.y + d.T16, d.y2 = d.x2 + d.T16;
	d.s = d.y2 + d.T16, d.sf = (uint8_t*)(d.s + d.T16), d.qr = d.sf + d.T16;
	memset(d.u, w8(-q - e), d.T16); memset(d.v, w8(-q - e), d.T16);
	memset(d.x, w8(-q - e), d.T16); memset(d.y, w8(-q - e), d.T16);
	memset(d.x2, w8(-q2 - e2), d.T16); memset(d.y2, w8(-q2 - e2), d.T16);
	if (!approx_max) {
		H = (int32_t*)malloc(sizeof(int32_t) * d.T16);
		for (t = 0; t < d.T16; ++t) H[t] = ORA_NEG_INF;
	}
	d.dir = 0, d.off = d.off_end = 0;
	if (with_cigar) {
		d.dir = (uint8_t*)malloc((size_t)(qlen + tlen - 1) * d.ncol + 16);
		d.off = (int*)malloc(sizeof(int) * 2 * (qlen + tlen - 1));
		d.off_end = d.off + (qlen + tlen - 1);
	}
	for (t = 0; t < qlen; ++t) d.qr[t] = query[qlen - 1 - t];
	memcpy(d.sf, target, tlen);

	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1, bnd;
		const uint8_t *qrr = d.qr + (qlen - 1 - r);
		/* valid interval on this anti-diagonal (:137-146) */
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez->zdropped = 1; break; }
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		/* first-column / diagonal boundary values (:148-163) */
		bnd = r == 0 ? w8(-q - e) : r < long_thres ? w8(-e) : r == long_thres ? w8(long_diff) : w8(-e2);
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = d.x[st - 1], x21 = d.x2[st - 1], v1 = d.v[st - 1];
			else x1 = w8(-q - e), x21 = w8(-q2 - e2), v1 = w8(-q - e);
		} else x1 = w8(-q - e), x21 = w8(-q2 - e2), v1 = bnd;
		if (en >= r) d.y[r] = w8(-q - e), d.y2[r] = w8(-q2 - e2), d.u[r] = bnd;
		/* substitution scores: 16-byte chunks from st0, overshooting past en0 (:165-184) */
		if (!(flag & ORA_EZ_GENERIC_SC)) {
			for (t = st0; t <= en0; t += 16) {
				int k;
				int8_t tmp[16];
				for (k = 0; k < 16; ++k) {
					uint8_t a = d.sf[t + k], b = qrr[t + k];
					tmp[k] = (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) ? sc_N : a == b ? sc_mch : sc_mis;
				}
				memcpy(d.s + t, tmp, 16); /* store after all loads of the chunk, like the SSE code */
			}
		} else {
			for (t = st0; t <= en0; ++t) d.s[t] = mat[d.sf[t] * m + qrr[t]];
		}
		/* one sweep over the aligned interval; every lane reads the previous row's state (:186-323) */
		{
			int8_t cx = x1, cv = v1, cx2 = x21; /* carry = old value of lane t-1 */
			uint8_t *pr = with_cigar ? d.dir + (size_t)r * d.ncol - st : 0;
			if (with_cigar) d.off[r] = st, d.off_end[r] = en;
			for (t = st; t <= en; ++t) {
				int8_t z = d.s[t], xt1 = cx, vt1 = cv, x2t1 = cx2, ut = d.u[t], a, b, a2, b2, tmp;
				uint8_t dd;
				cx = d.x[t], cv = d.v[t], cx2 = d.x2[t];
				a = w8(xt1 + vt1), b = w8(d.y[t] + ut), a2 = w8(x2t1 + vt1), b2 = w8(d.y2[t] + ut);
				if (!right) { /* left-aligned gaps: strictly greater wins (:235-243) */
					dd = a > z ? 1 : 0;  z = z > a ? z : a;
					dd = b > z ? 2 : dd; z = z > b ? z : b;
					dd = a2 > z ? 3 : dd; z = z > a2 ? z : a2;
					dd = b2 > z ? 4 : dd; z = z > b2 ? z : b2;
				} else {      /* right-aligned gaps: ties go to the gap state (:282-290) */
					dd = z > a ? 0 : 1;  z = z > a ? z : a;
					dd = z > b ? dd : 2; z = z > b ? z : b;
					dd = z > a2 ? dd : 3; z = z > a2 ? z : a2;
					dd = z > b2 ? dd : 4; z = z > b2 ? z : b2;
				}
				z = z < sc_mch ? z : sc_mch;
				d.u[t] = w8(z - vt1), d.v[t] = w8(z - ut);                                /* :59-60 */
				tmp = w8(z - q);  a = w8(a - tmp),  b = w8(b - tmp);
				tmp = w8(z - q2); a2 = w8(a2 - tmp), b2 = w8(b2 - tmp);
				if (!right) {
					d.x[t]  = w8((a  > 0 ? a  : 0) - qe);  if (a  > 0) dd |= 0x08;
					d.y[t]  = w8((b  > 0 ? b  : 0) - qe);  if (b  > 0) dd |= 0x10;
					d.x2[t] = w8((a2 > 0 ? a2 : 0) - qe2); if (a2 > 0) dd |= 0x20;
					d.y2[t] = w8((b2 > 0 ? b2 : 0) - qe2); if (b2 > 0) dd |= 0x40;
				} else {
					d.x[t]  = w8((0 > a  ? 0 : a)  - qe);  if (!(0 > a))  dd |= 0x08;
					d.y[t]  = w8((0 > b  ? 0 : b)  - qe);  if (!(0 > b))  dd |= 0x10;
					d.x2[t] = w8((0 > a2 ? 0 : a2) - qe2); if (!(0 > a2)) dd |= 0x20;
					d.y2[t] = w8((0 > b2 ? 0 : b2) - qe2); if (!(0 > b2)) dd |= 0x40;
				}
				if (with_cigar) pr[t] = dd;
			}
		}
		if (!approx_max) { /* exact row maximum with the reference's scan order (:325-365) */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + d.u[en0] : H[en0] + d.v[en0];
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += d.v[t + i];
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += d.v[t];
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = d.v[0] - qe_in, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en0;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (zdrop_test(ez, max_H, r, max_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else { /* approximate: follow one cell downwards/diagonally (:366-383) */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = d.v[last_H0_t], d1 = d.u[last_H0_t + 1];
					if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += d.v[last_H0_t];
				else ++last_H0_t, H0 += d.u[last_H0_t];
			} else H0 = d.v[0] - qe_in, last_H0_t = 0;
			if ((flag & ORA_EZ_APPROX_DROP) && zdrop_test(ez, H0, r, last_H0_t, zdrop, e2)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem); free(H);
	if (with_cigar) { /* which cell to trace back from (:387-399) */
		cig_t g = { cigar, 0, cigar_cap, 0, 0xf };
		int rev = !!(flag & ORA_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & ORA_EZ_EXTZ_ONLY)) traceback(&d, rev, 0, tlen - 1, qlen - 1, &g);
		else if (!ez->zdropped && (flag & ORA_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > ez->max) {
			ez->reach_end = 1;
			traceback(&d, rev, 0, ez->mqe_t, qlen - 1, &g);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) traceback(&d, rev, 0, ez->max_t, ez->max_q, &g);
		ez->n_cigar = g.n, ez->cigar_overflow = g.ovf;
		free(d.dir); free(d.off);
	}
}

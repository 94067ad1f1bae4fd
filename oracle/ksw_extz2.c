/* oracle/ksw_extz2.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar, lane-exact restatement of the reference's single-affine extension DP (ksw_extz2_sse,
 * /root/reference/ksw2_extz2_sse.c:25-311, the SSE4.1 code path that ksw2_dispatch.c:59-61 selects on this hardware)
 * with its traceback (ksw_backtrack, ksw2.h:130-162).  Same "lane-exact" contract as ksw_extd2.c: 16-aligned row blocks,
 * stale lanes, the chunked score fill with its overshoot (arrays u v x y s | sf | qr contiguous and zero-initialised,
 * ksw2_extz2_sse.c:91-93), mod-256 arithmetic.  What differs from the dual-affine kernel: scores are shifted by 2(q+e) so
 * that the running value is non-negative, the second maximum and the clamp are UNSIGNED byte operations (:49-50), the
 * state starts at zero, and the score recurrences read u/v as unsigned bytes minus (q+e) (:236-262).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static inline int8_t w8(int v) { return (int8_t)(uint8_t)v; } /* mod-256 wrap, like _mm_add/sub_epi8 */
static inline uint8_t u8max(uint8_t a, uint8_t b) { return a > b ? a : b; }
static inline uint8_t u8min(uint8_t a, uint8_t b) { return a < b ? a : b; }

typedef struct {
	int qlen, tlen, T16, ncol;
	int8_t *u, *v, *x, *y, *s;
	uint8_t *sf, *qr;
	uint8_t *dir;
	int *off, *off_end;
} dpz_t;

static void ez_reset(ora_ez_t *ez)
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0, ez->score = ez->mqe = ez->mte = ORA_NEG_INF;
	ez->n_cigar = 0, ez->zdropped = 0, ez->reach_end = 0, ez->cigar_overflow = 0;
}

static int zdrop_test(ora_ez_t *ez, int32_t H, int r, int t, int zdrop, int8_t e) /* ksw_apply_zdrop is_rot=1, ksw2.h:171-187 */
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

static void cig_push(cig_t *g, uint32_t op, int len)
{
	if (g->n == 0 || op != g->last_op) {
		if (g->n < g->cap) g->c[g->n] = (uint32_t)len << 4 | op; else g->ovf = 1;
		++g->n, g->last_op = op;
	} else if (g->n <= g->cap) g->c[g->n - 1] += (uint32_t)len << 4;
}

static void traceback(const dpz_t *d, int is_rev, int i0, int j0, cig_t *g) /* ksw2.h:130-162, is_rot=1, min_intron_len=0 */
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
		else if (state == 1 || state == 3) cig_push(g, 2, 1), --i;
		else cig_push(g, 1, 1), --j;
	}
	if (i >= 0) cig_push(g, 2, i + 1);
	if (j >= 0) cig_push(g, 1, j + 1);
	if (!is_rev) {
		int k, n = g->n < g->cap ? g->n : g->cap;
		if (!g->ovf) for (k = 0; k < n >> 1; ++k) { uint32_t t = g->c[k]; g->c[k] = g->c[n - 1 - k]; g->c[n - 1 - k] = t; }
	}
}

void ora_ksw_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ora_ez_t *ez, uint32_t *cigar, int cigar_cap)
{
	const int with_cigar = !(flag & ORA_EZ_SCORE_ONLY), approx_max = !!(flag & ORA_EZ_APPROX_MAX), right = !!(flag & ORA_EZ_RIGHT);
	const int qe = q + e;
	int r, t, Q16, last_st = -1, last_en = -1, min_sc;
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int8_t sc_mch, sc_mis, sc_N, qe2b, max_scb;
	uint8_t *mem;
	dpz_t d;

	ez_reset(ez);
	if (m <= 0 || qlen <= 0 || tlen <= 0) return;                                       /* :66 */
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? w8(-e) : mat[m * m - 1];                               /* :77 */
	qe2b = w8((q + e) * 2), max_scb = w8(mat[0] + (q + e) * 2);                         /* :69, :79 */
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	d.qlen = qlen, d.tlen = tlen;
	d.T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
	d.ncol = qlen < tlen ? qlen : tlen;
	d.ncol = (((d.ncol < w + 1 ? d.ncol : w + 1) + 15) / 16 + 1) * 16;                  /* :84-85 */
	for (t = 1, min_sc = mat[1]; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
	if (-min_sc > 2 * (q + e)) return;                                                  /* :90 */

	mem = (uint8_t*)calloc((size_t)d.T16 * 6 + Q16 + 16, 1);                            /* u v x y s | sf | qr, zero-filled (:92-93) */
	d.u = (int8_t*)mem, d.v = d.u + d.T16, d.x = d.v + d.T16, d.y = d.x + d.T16, d.s = d.y + d.T16;
	d.sf = (uint8_t*)(d.s + d.T16), d.qr = d.sf + d.T16;
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
		int8_t x1, v1;
		const uint8_t *qrr = d.qr + (qlen - 1 - r), *u8 = (const uint8_t*)d.u, *v8 = (const uint8_t*)d.v;
		if (st < r - qlen + 1) st = r - qlen + 1;                                       /* :117-126 */
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez->zdropped = 1; break; }
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		if (st > 0) {                                                                   /* :128-134 */
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = d.x[st - 1], v1 = d.v[st - 1];
			else x1 = v1 = 0;
		} else x1 = 0, v1 = r ? q : 0;
		if (en >= r) d.y[r] = 0, d.u[r] = r ? q : 0;
		if (!(flag & ORA_EZ_GENERIC_SC)) {                                              /* :136-149 */
			for (t = st0; t <= en0; t += 16) {
				int k;
				int8_t tmp[16];
				for (k = 0; k < 16; ++k) {
					uint8_t a = d.sf[t + k], b = qrr[t + k];
					tmp[k] = (a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) ? sc_N : a == b ? sc_mch : sc_mis;
				}
				memcpy(d.s + t, tmp, 16);
			}
		} else {
			for (t = st0; t <= en0; ++t) d.s[t] = mat[d.sf[t] * m + qrr[t]];
		}
		{
			int8_t cx = x1, cv = v1;
			uint8_t *pr = with_cigar ? d.dir + (size_t)r * d.ncol - st : 0;
			if (with_cigar) d.off[r] = st, d.off_end[r] = en;
			for (t = st; t <= en; ++t) {
				int8_t z = w8(d.s[t] + qe2b), xt1 = cx, vt1 = cv, ut = d.u[t], a, b;   /* __dp_code_block1, :34-46 */
				uint8_t dd = 0, zu;
				cx = d.x[t], cv = d.v[t];
				a = w8(xt1 + vt1), b = w8(d.y[t] + ut);
				if (!with_cigar) z = z > a ? z : a;                                     /* :164 */
				else if (!right) { dd = a > z ? 1 : 0; z = z > a ? z : a; dd = b > z ? 2 : dd; }   /* :186-190 */
				else { dd = z > a ? 0 : 1; z = z > a ? z : a; dd = z > b ? dd : 2; }               /* :213-217 */
				zu = u8max((uint8_t)z, (uint8_t)b);                                     /* __dp_code_block2, :49-55: unsigned */
				zu = u8min(zu, (uint8_t)max_scb);
				d.u[t] = w8(zu - (uint8_t)vt1), d.v[t] = w8(zu - (uint8_t)ut);
				z = w8(zu - (uint8_t)q);
				a = w8(a - z), b = w8(b - z);
				if (!with_cigar) { d.x[t] = a > 0 ? a : 0; d.y[t] = b > 0 ? b : 0; }    /* :169-170 */
				else if (!right) {
					d.x[t] = a > 0 ? a : 0; if (a > 0) dd |= 0x08;                      /* :199-204 */
					d.y[t] = b > 0 ? b : 0; if (b > 0) dd |= 0x10;
				} else {
					d.x[t] = 0 > a ? 0 : a; if (!(0 > a)) dd |= 0x08;                   /* :226-231 */
					d.y[t] = 0 > b ? 0 : b; if (!(0 > b)) dd |= 0x10;
				}
				if (with_cigar) pr[t] = dd;
			}
		}
		if (!approx_max) {                                                              /* :236-278 */
			int32_t max_H, max_t;
			if (r > 0) {
				int32_t HH[4], tt[4], en1 = st0 + (en0 - st0) / 4 * 4, i;
				max_H = H[en0] = en0 > 0 ? H[en0 - 1] + u8[en0] - qe : H[en0] + v8[en0] - qe;
				max_t = en0;
				for (i = 0; i < 4; ++i) HH[i] = max_H, tt[i] = max_t;
				for (t = st0; t < en1; t += 4)
					for (i = 0; i < 4; ++i) {
						H[t + i] += (int32_t)v8[t + i] - qe;
						if (H[t + i] > HH[i]) HH[i] = H[t + i], tt[i] = t;
					}
				for (i = 0; i < 4; ++i)
					if (max_H < HH[i]) max_H = HH[i], max_t = tt[i] + i;
				for (; t < en0; ++t) {
					H[t] += (int32_t)v8[t] - qe;
					if (H[t] > max_H) max_H = H[t], max_t = t;
				}
			} else H[0] = v8[0] - qe - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en0;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (zdrop_test(ez, max_H, r, max_t, zdrop, e)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {                                                                        /* :279-296 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = v8[last_H0_t] - qe, d1 = u8[last_H0_t + 1] - qe;
					if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += v8[last_H0_t] - qe;
				else ++last_H0_t, H0 += u8[last_H0_t] - qe;
				if ((flag & ORA_EZ_APPROX_DROP) && zdrop_test(ez, H0, r, last_H0_t, zdrop, e)) break;
			} else H0 = v8[0] - qe - qe, last_H0_t = 0;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem); free(H);
	if (with_cigar) {                                                                   /* :301-311 */
		cig_t g = { cigar, 0, cigar_cap, 0, 0xf };
		int rev = !!(flag & ORA_EZ_REV_CIGAR);
		if (!ez->zdropped && !(flag & ORA_EZ_EXTZ_ONLY)) traceback(&d, rev, tlen - 1, qlen - 1, &g);
		else if (!ez->zdropped && (flag & ORA_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > ez->max) {
			ez->reach_end = 1;
			traceback(&d, rev, ez->mqe_t, qlen - 1, &g);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) traceback(&d, rev, ez->max_t, ez->max_q, &g);
		ez->n_cigar = g.n, ez->cigar_overflow = g.ovf;
		free(d.dir); free(d.off);
	}
}

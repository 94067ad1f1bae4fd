/* oracle/ksw_exts2.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Scalar, lane-exact restatement of the reference's splice-aware extension DP (ksw_exts2_sse,
 * /root/reference/ksw2_exts2_sse.c:33-465, the SSE4.1 code path) with its traceback (ksw_backtrack, ksw2.h:130-162, where
 * state 3 becomes an N operation).  Same contract as ksw_extd2.c.  What differs from the dual-affine kernel: there is no band
 * (:226-230); the second gap state is an intron state on the target only (x2, no y2) that costs q2 to open, nothing to extend,
 * and whose entry/exit is priced per target position by the donor/acceptor arrays derived from the neighbouring bases
 * (:120-194, miniprot-style model with KSW_EZ_SPLICE_CMPLX); the running value is not clamped; Z-drop ignores the diagonal
 * (:434,:451 pass e = 0).  Arrays u v x y x2 donor acceptor s | sf | qr are contiguous (:104-107).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static inline int8_t w8(int v) { return (int8_t)(uint8_t)v; } /* mod-256 wrap, like _mm_add/sub_epi8 */

typedef struct {
	int qlen, tlen, T16, ncol;
	int8_t *u, *v, *x, *y, *x2, *donor, *acceptor, *s;
	uint8_t *sf, *qr;
	uint8_t *dir;
	int *off, *off_end;
} dps_t;

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

static void traceback(const dps_t *d, int is_rev, int min_intron_len, int i0, int j0, cig_t *g) /* ksw2.h:130-162, is_rot=1 */
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

/* junc may be NULL.  KSW_SPSC_OFFSET is 64 (ksw2.h:22). */
void ora_ksw_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int end_bonus, int8_t junc_bonus, int8_t junc_pen, int flag,
                   const uint8_t *junc, ora_ez_t *ez, uint32_t *cigar, int cigar_cap)
{
	const int with_cigar = !(flag & ORA_EZ_SCORE_ONLY), approx_max = !!(flag & ORA_EZ_APPROX_MAX), right = !!(flag & ORA_EZ_RIGHT);
	const int qe = q + e, is_for = !!(flag & ORA_EZ_SPLICE_FOR), is_rev = !!(flag & ORA_EZ_SPLICE_REV), rev_cigar = !!(flag & ORA_EZ_REV_CIGAR);
	int r, t, Q16, last_st = -1, last_en = -1, min_sc, long_thres, long_diff;
	int32_t *H = 0, H0 = 0, last_H0_t = 0;
	int8_t sc_mch, sc_mis, sc_N;
	uint8_t *mem;
	dps_t d;

	ez_reset(ez);
	if (m <= 1 || qlen <= 0 || tlen <= 0 || q2 <= q + e) return;                        /* :77 */
	sc_mch = mat[0], sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? w8(-e) : mat[m * m - 1];                               /* :86 */
	d.qlen = qlen, d.tlen = tlen;
	d.T16 = (tlen + 15) / 16 * 16, Q16 = (qlen + 15) / 16 * 16;
	d.ncol = (((qlen < tlen ? qlen : tlen) + 15) / 16 + 1) * 16;                        /* :90 */
	for (t = 1, min_sc = mat[1]; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
	if (-min_sc > 2 * (q + e)) return;                                                  /* :96 */
	long_thres = (q2 - q) / e - 1;                                                      /* :98-101 */
	if (q2 > q + e + long_thres * e) ++long_thres;
	long_diff = long_thres * e - (q2 - q);

	mem = (uint8_t*)calloc((size_t)d.T16 * 10 + Q16 + 16, 1);
	d.u = (int8_t*)mem, d.v = d.u + d.T16, d.x = d.v + d.T16, d.y = d.x + d.T16, d.x2 = d.y + d.T16;
	d.donor = d.x2 + d.T16, d.acceptor = d.donor + d.T16;
	d.s = d.acceptor + d.T16, d.sf = (uint8_t*)(d.s + d.T16), d.qr = d.sf + d.T16;
	memset(d.u, w8(-q - e), (size_t)d.T16 * 4);
	memset(d.x2, w8(-q2), d.T16);
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

	if (is_for || is_rev) {                                                             /* donor / acceptor costs, :120-194 */
		const int sp0[4] = { 8, 15, 21, 30 };
		int sp[4];
		if (flag & ORA_EZ_SPLICE_CMPLX) for (t = 0; t < 4; ++t) sp[t] = (int)((double)sp0[t] / 3. + .499);
		else sp[0] = flag & ORA_EZ_SPLICE_FLANK ? noncan / 2 : 0, sp[1] = sp[2] = sp[3] = noncan;
		memset(d.donor, w8(-sp[3]), d.T16);
		memset(d.acceptor, w8(-sp[3]), d.T16);
		if (!rev_cigar) {
			for (t = 0; t < tlen - 4; ++t) {
				int z = 3;
				if (is_for) {
					if (target[t+1] == 2 && target[t+2] == 3) z = target[t+3] == 0 || target[t+3] == 2 ? -1 : 0;
					else if (target[t+1] == 2 && target[t+2] == 1) z = 1;
					else if (target[t+1] == 0 && target[t+2] == 3) z = 2;
				} else if (is_rev) {
					if (target[t+1] == 1 && target[t+2] == 3) z = target[t+3] == 0 || target[t+3] == 2 ? -1 : 0;
					else if (target[t+1] == 2 && target[t+2] == 3) z = 2;
				}
				d.donor[t] = z < 0 ? 0 : w8(-sp[z]);
			}
			for (t = 2; t < tlen; ++t) {
				int z = 3;
				if (is_for) {
					if (target[t-1] == 0 && target[t] == 2) z = target[t-2] == 1 || target[t-2] == 3 ? -1 : 0;
					else if (target[t-1] == 0 && target[t] == 1) z = 2;
				} else if (is_rev) {
					if (target[t-1] == 0 && target[t] == 1) z = target[t-2] == 1 || target[t-2] == 3 ? -1 : 0;
					else if (target[t-1] == 2 && target[t] == 1) z = 1;
					else if (target[t-1] == 0 && target[t] == 3) z = 2;
				}
				d.acceptor[t] = z < 0 ? 0 : w8(-sp[z]);
			}
		} else {
			for (t = 0; t < tlen - 4; ++t) {
				int z = 3;
				if (is_for) {
					if (target[t+1] == 2 && target[t+2] == 0) z = target[t+3] == 1 || target[t+3] == 3 ? -1 : 0;
					else if (target[t+1] == 1 && target[t+2] == 0) z = 2;
				} else if (is_rev) {
					if (target[t+1] == 1 && target[t+2] == 0) z = target[t+3] == 1 || target[t+3] == 3 ? -1 : 0;
					else if (target[t+1] == 1 && target[t+2] == 2) z = 1;
					else if (target[t+1] == 3 && target[t+2] == 0) z = 2;
				}
				d.donor[t] = z < 0 ? 0 : w8(-sp[z]);
			}
			for (t = 2; t < tlen; ++t) {
				int z = 3;
				if (is_for) {
					if (target[t-1] == 3 && target[t] == 2) z = target[t-2] == 0 || target[t-2] == 2 ? -1 : 0;
					else if (target[t-1] == 1 && target[t] == 2) z = 1;
					else if (target[t-1] == 3 && target[t] == 0) z = 2;
				} else if (is_rev) {
					if (target[t-1] == 3 && target[t] == 1) z = target[t-2] == 0 || target[t-2] == 2 ? -1 : 0;
					else if (target[t-1] == 3 && target[t] == 2) z = 2;
				}
				d.acceptor[t] = z < 0 ? 0 : w8(-sp[z]);
			}
		}
	}
	if (junc && (flag & ORA_EZ_SPLICE_SCORE)) {                                         /* :196-203 */
		const uint8_t donor_val = is_for == !rev_cigar ? 0 : 1;
		for (t = 0; t < tlen - 1; ++t)
			d.donor[t] = w8(d.donor[t] + (junc[t+1] == 0xff || (junc[t+1] & 1) != donor_val ? -junc_pen : (int8_t)(junc[t+1] >> 1) - (int8_t)64));
		for (t = 0; t < tlen - 1; ++t)
			d.acceptor[t] = w8(d.acceptor[t] + (junc[t+1] == 0xff || (junc[t+1] & 1) != !donor_val ? -junc_pen : (int8_t)(junc[t+1] >> 1) - (int8_t)64));
	} else if (junc) {                                                                  /* :204-222 */
		if (!rev_cigar) {
			for (t = 0; t < tlen - 1; ++t) if ((is_for && (junc[t+1] & 1)) || (is_rev && (junc[t+1] & 8))) d.donor[t] = w8(d.donor[t] + junc_bonus);
			for (t = 0; t < tlen; ++t) if ((is_for && (junc[t] & 2)) || (is_rev && (junc[t] & 4))) d.acceptor[t] = w8(d.acceptor[t] + junc_bonus);
		} else {
			for (t = 0; t < tlen - 1; ++t) if ((is_for && (junc[t+1] & 2)) || (is_rev && (junc[t+1] & 4))) d.donor[t] = w8(d.donor[t] + junc_bonus);
			for (t = 0; t < tlen; ++t) if ((is_for && (junc[t] & 1)) || (is_rev && (junc[t] & 8))) d.acceptor[t] = w8(d.acceptor[t] + junc_bonus);
		}
	}

	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21, v1, bnd;
		const uint8_t *qrr = d.qr + (qlen - 1 - r);
		if (st < r - qlen + 1) st = r - qlen + 1;                                       /* :230-233: no band */
		if (en > r) en = r;
		st0 = st, en0 = en;
		st = st / 16 * 16, en = (en + 16) / 16 * 16 - 1;
		bnd = r == 0 ? w8(-q - e) : r < long_thres ? w8(-e) : r == long_thres ? w8(long_diff) : 0;   /* :241,:245 */
		if (st > 0) {
			if (st - 1 >= last_st && st - 1 <= last_en) x1 = d.x[st - 1], x21 = d.x2[st - 1], v1 = d.v[st - 1];
			else x1 = w8(-q - e), x21 = w8(-q2), v1 = w8(-q - e);
		} else x1 = w8(-q - e), x21 = w8(-q2), v1 = bnd;
		if (en >= r) d.y[r] = w8(-q - e), d.u[r] = bnd;
		if (!(flag & ORA_EZ_GENERIC_SC)) {                                              /* :248-266 */
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
			int8_t cx = x1, cv = v1, cx2 = x21;
			uint8_t *pr = with_cigar ? d.dir + (size_t)r * d.ncol - st : 0;
			if (with_cigar) d.off[r] = st, d.off_end[r] = en;
			for (t = st; t <= en; ++t) {
				int8_t z = d.s[t], xt1 = cx, vt1 = cv, x2t1 = cx2, ut = d.u[t], a, b, a2, a2a, tmp, dn;
				uint8_t dd = 0;
				cx = d.x[t], cv = d.v[t], cx2 = d.x2[t];
				a = w8(xt1 + vt1), b = w8(d.y[t] + ut), a2 = w8(x2t1 + vt1), a2a = w8(a2 + d.acceptor[t]);   /* __dp_code_block1 */
				if (!with_cigar) { z = z > a ? z : a; z = z > b ? z : b; z = z > a2a ? z : a2a; }             /* :283-285 */
				else if (!right) {                                                                             /* :312-318 */
					dd = a > z ? 1 : 0;    z = z > a ? z : a;
					dd = b > z ? 2 : dd;   z = z > b ? z : b;
					dd = a2a > z ? 3 : dd; z = z > a2a ? z : a2a;
				} else {                                                                                       /* :355-361 */
					dd = z > a ? 0 : 1;    z = z > a ? z : a;
					dd = z > b ? dd : 2;   z = z > b ? z : b;
					dd = z > a2a ? dd : 3; z = z > a2a ? z : a2a;
				}
				d.u[t] = w8(z - vt1), d.v[t] = w8(z - ut);                                                     /* __dp_code_block2 */
				tmp = w8(z - q); a = w8(a - tmp), b = w8(b - tmp);
				a2 = w8(a2 - w8(z - q2));
				dn = d.donor[t];
				if (!with_cigar || !right) {
					d.x[t] = w8((a > 0 ? a : 0) - qe); if (with_cigar && a > 0) dd |= 0x08;                    /* :333-338 */
					d.y[t] = w8((b > 0 ? b : 0) - qe); if (with_cigar && b > 0) dd |= 0x10;
					d.x2[t] = w8((a2 > dn ? a2 : dn) - q2); if (with_cigar && a2 > dn) dd |= 0x20;            /* :340-348 */
				} else {
					d.x[t] = w8((0 > a ? 0 : a) - qe); if (!(0 > a)) dd |= 0x08;                               /* :377-382 */
					d.y[t] = w8((0 > b ? 0 : b) - qe); if (!(0 > b)) dd |= 0x10;
					d.x2[t] = w8((dn > a2 ? dn : a2) - q2); if (!(dn > a2)) dd |= 0x20;                        /* :384-392 */
				}
				if (with_cigar) pr[t] = dd;
			}
		}
		if (!approx_max) {                                                              /* :396-437 */
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
			} else H[0] = d.v[0] - qe, max_H = H[0], max_t = 0;
			if (en0 == tlen - 1 && H[en0] > ez->mte) ez->mte = H[en0], ez->mte_q = r - en0;
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) ez->mqe = H[st0], ez->mqe_t = st0;
			if (zdrop_test(ez, max_H, r, max_t, zdrop, 0)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {                                                                        /* :438-454 */
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					int32_t d0 = d.v[last_H0_t], d1 = d.u[last_H0_t + 1];
					if (d0 > d1) H0 += d0; else H0 += d1, ++last_H0_t;
				} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += d.v[last_H0_t];
				else ++last_H0_t, H0 += d.u[last_H0_t];
			} else H0 = d.v[0] - qe, last_H0_t = 0;
			if ((flag & ORA_EZ_APPROX_DROP) && zdrop_test(ez, H0, r, last_H0_t, zdrop, 0)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
		last_st = st, last_en = en;
	}
	free(mem); free(H);
	if (with_cigar) {                                                                   /* :459-469: min_intron_len = long_thres */
		cig_t g = { cigar, 0, cigar_cap, 0, 0xf };
		if (!ez->zdropped && !(flag & ORA_EZ_EXTZ_ONLY)) traceback(&d, rev_cigar, long_thres, tlen - 1, qlen - 1, &g);
		else if (!ez->zdropped && (flag & ORA_EZ_EXTZ_ONLY) && ez->mqe + end_bonus > ez->max) {
			ez->reach_end = 1;
			traceback(&d, rev_cigar, long_thres, ez->mqe_t, qlen - 1, &g);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) traceback(&d, rev_cigar, long_thres, ez->max_t, ez->max_q, &g);
		ez->n_cigar = g.n, ez->cigar_overflow = g.ovf;
		free(d.dir); free(d.off);
	}
}

/* oracle/lchain.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Restatement of the chaining DP: mg_lchain_dp (/root/reference/lchain.c:148-217), its scoring function comput_sc
 * (:113-138) with the bit-trick log2 (mmpriv.h:139-147), the backtrack (:9-76) and the compaction (:78-111).
 * Compile with -ffp-contract=off: the reference's x86-64 -O2 build performs the float mul/add separately. */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define SEG_SHIFT 48
#define SEG_MASK (0xffULL << SEG_SHIFT)

static inline float fast_log2(float x) /* mg_log2; only meaningful for x >= 2 */
{
	union { float f; uint32_t i; } z = { x };
	float l = (float)((int)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	l += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return l;
}

static inline int32_t link_score(const ora128_t *ai, const ora128_t *aj, int32_t max_dist_x, int32_t max_dist_y, int32_t bw,
                                 float pen_gap, float pen_skip, int is_cdna, int n_seg)
{
	int32_t dq = (int32_t)ai->y - (int32_t)aj->y, dr, dd, dg, span, sc;
	int32_t si = (int32_t)((ai->y & SEG_MASK) >> SEG_SHIFT), sj = (int32_t)((aj->y & SEG_MASK) >> SEG_SHIFT);
	if (dq <= 0 || dq > max_dist_x) return INT32_MIN;
	dr = (int32_t)(ai->x - aj->x);
	if (si == sj && (dr == 0 || dq > max_dist_y)) return INT32_MIN;
	dd = dr > dq ? dr - dq : dq - dr;
	if (si == sj && dd > bw) return INT32_MIN;
	if (n_seg > 1 && !is_cdna && si == sj && dr > max_dist_y) return INT32_MIN;
	dg = dr < dq ? dr : dq;
	span = (int32_t)(aj->y >> 32 & 0xff);
	sc = span < dg ? span : dg;
	if (dd || dg > span) {
		float lin = pen_gap * (float)dd + pen_skip * (float)dg;
		float lg = dd >= 1 ? fast_log2((float)(dd + 1)) : 0.0f;
		if (is_cdna || si != sj) {
			if (si != sj && dr == 0) ++sc;
			else if (dr > dq || si != sj) sc -= (int)(lin < lg ? lin : lg);
			else sc -= (int)(lin + .5f * lg);
		} else sc -= (int)(lin + .5f * lg);
	}
	return sc;
}

void ora_lchain_fill(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, float pen_gap, float pen_skip,
                     int is_cdna, int n_seg, int64_t n, const ora128_t *a, int32_t *f, int64_t *p, int32_t *v, int32_t *t)
{
	int64_t i, j, best_ii = -1, st = 0;
	if (max_dist_x < bw) max_dist_x = bw;
	if (max_dist_y < bw && !is_cdna) max_dist_y = bw;
	memset(t, 0, sizeof(int32_t) * n);
	for (i = 0; i < n; ++i) {
		int64_t best_j = -1, end_j;
		int32_t best = (int32_t)(a[i].y >> 32 & 0xff), n_skip = 0;
		while (st < i && (a[i].x >> 32 != a[st].x >> 32 || a[i].x > a[st].x + max_dist_x)) ++st;
		if (i - st > max_iter) st = i - max_iter;
		for (j = i - 1; j >= st; --j) {
			int32_t sc = link_score(&a[i], &a[j], max_dist_x, max_dist_y, bw, pen_gap, pen_skip, is_cdna, n_seg);
			if (sc == INT32_MIN) continue;
			sc += f[j];
			if (sc > best) {
				best = sc, best_j = j;
				if (n_skip > 0) --n_skip;
			} else if (t[j] == (int32_t)i) {
				if (++n_skip > max_skip) break;
			}
			if (p[j] >= 0) t[p[j]] = (int32_t)i;
		}
		end_j = j;
		if (best_ii < 0 || a[i].x - a[best_ii].x > (uint64_t)(int64_t)max_dist_x) { /* NB: unsigned subtraction vs int64 cast (:189) */
			int32_t m = INT32_MIN;
			best_ii = -1;
			for (j = i - 1; j >= st; --j) if (m < f[j]) m = f[j], best_ii = j;
		}
		if (best_ii >= 0 && best_ii < end_j) {
			int32_t tmp = link_score(&a[i], &a[best_ii], max_dist_x, max_dist_y, bw, pen_gap, pen_skip, is_cdna, n_seg);
			if (tmp != INT32_MIN && best < tmp + f[best_ii]) best = tmp + f[best_ii], best_j = best_ii;
		}
		f[i] = best, p[i] = best_j;
		v[i] = best_j >= 0 && v[best_j] > best ? v[best_j] : best;
		if (best_ii < 0 || (a[i].x - a[best_ii].x <= (uint64_t)(int64_t)max_dist_x && f[best_ii] < f[i])) best_ii = i;
	}
}

/* where the chain ending at z[k] should stop when walked backwards (mg_chain_bk_end, :9-25) */
static int64_t chain_stop(int32_t max_drop, const ora128_t *z, const int32_t *f, const int64_t *p, int32_t *t, int64_t k)
{
	int64_t i = (int64_t)z[k].y, end_i = -1, max_i = i;
	int32_t max_s = 0;
	if (i < 0 || t[i] != 0) return i;
	do {
		int32_t s;
		t[i] = 2;
		end_i = i = p[i];
		s = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
		if (s > max_s) max_s = s, max_i = i;
		else if (max_s - s > max_drop) break;
	} while (i >= 0 && t[i] == 0);
	for (i = (int64_t)z[k].y; i >= 0 && i != end_i; i = p[i]) t[i] = 0;
	return max_i;
}

int ora_lchain_dp(int max_dist_x, int max_dist_y, int bw, int max_skip, int max_iter, int min_cnt, int min_sc,
                  float pen_gap, float pen_skip, int is_cdna, int n_seg, int64_t n, ora128_t *a,
                  uint64_t *u_out, int64_t *n_a_out)
{
	int32_t *f, *t, *v, n_u = 0, max_drop = is_cdna ? INT32_MAX : bw;
	int64_t *p, i, k, n_z = 0, n_v = 0;
	ora128_t *z, *b, *w;
	uint64_t *u;

	*n_a_out = 0;
	if (n == 0) return 0;
	p = (int64_t*)malloc(n * 8), f = (int32_t*)malloc(n * 4), v = (int32_t*)malloc(n * 4), t = (int32_t*)malloc(n * 4);
	ora_lchain_fill(max_dist_x, max_dist_y, bw, max_skip, max_iter, pen_gap, pen_skip, is_cdna, n_seg, n, a, f, p, v, t);

	/* backtrack (:27-76): visit chain ends from the highest f down */
	for (i = 0; i < n; ++i) if (f[i] >= min_sc) ++n_z;
	if (n_z == 0) { free(p); free(f); free(v); free(t); return 0; }
	z = (ora128_t*)malloc(n_z * sizeof(ora128_t));
	for (i = 0, k = 0; i < n; ++i) if (f[i] >= min_sc) z[k].x = (uint64_t)f[i], z[k++].y = (uint64_t)i;
	ora_radix_sort_128x(z, z + n_z);
	memset(t, 0, n * 4);
	u = u_out;
	for (k = n_z - 1; k >= 0; --k) {
		if (t[z[k].y] == 0) {
			int64_t n_v0 = n_v, end_i = chain_stop(max_drop, z, f, p, t, k);
			int32_t sc;
			for (i = (int64_t)z[k].y; i != end_i; i = p[i]) v[n_v++] = (int32_t)i, t[i] = 1;
			sc = i < 0 ? (int32_t)z[k].x : (int32_t)z[k].x - f[i];
			if (sc >= min_sc && n_v > n_v0 && n_v - n_v0 >= min_cnt) u[n_u++] = (uint64_t)sc << 32 | (uint64_t)(n_v - n_v0);
			else n_v = n_v0;
		}
	}
	free(z); free(p); free(f); free(t);
	if (n_u == 0) { free(v); return 0; }

	/* compaction (:78-111): chains in ascending anchor order, then chains sorted by target position of their first anchor */
	b = (ora128_t*)malloc(n_v * sizeof(ora128_t));
	for (i = 0, k = 0; i < n_u; ++i) {
		int32_t k0 = (int32_t)k, ni = (int32_t)u[i], j;
		for (j = 0; j < ni; ++j) b[k++] = a[v[k0 + (ni - j - 1)]];
	}
	free(v);
	w = (ora128_t*)malloc(n_u * sizeof(ora128_t));
	for (i = k = 0; i < n_u; ++i) w[i].x = b[k].x, w[i].y = (uint64_t)k << 32 | (uint64_t)i, k += (int32_t)u[i];
	ora_radix_sort_128x(w, w + n_u);
	{
		uint64_t *u2 = (uint64_t*)malloc(n_u * 8);
		for (i = k = 0; i < n_u; ++i) {
			int32_t j = (int32_t)w[i].y, cnt = (int32_t)u[j];
			u2[i] = u[j];
			memcpy(&a[k], &b[w[i].y >> 32], cnt * sizeof(ora128_t));
			k += cnt;
		}
		memcpy(u, u2, n_u * 8);
		free(u2);
	}
	free(b); free(w);
	*n_a_out = k;
	return n_u;
}

// Local (Smith-Waterman) score with end coordinates, as used by minimap2 for inversion tests:
// ksw_ll_qinit + ksw_ll_i16 (ksw2_ll_sse.c:37-152).  The reference runs Farrar's striped algorithm on 8 x int16
// lanes with saturating arithmetic; the end coordinates it reports depend on the striped scan order (the LAST row
// reaching the maximum, and within that row the last striped slot holding it), so we keep the same data layout:
// query position p lives in segment j = p % slen, lane l = p / slen.  This path is rare (only after a large
// Z-drop) and tiny, so it stays a scalar host routine.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include "align.hpp"

namespace mm2amd {

namespace {
inline int16_t adds16(int16_t a, int16_t b) { int v = (int)a + b; return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }
inline int16_t subsu16(int16_t a, int16_t b) { uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(x > y ? x - y : 0); } // _mm_subs_epu16
inline int16_t max16(int16_t a, int16_t b) { return a > b ? a : b; }
}

static int ll_local_score_striped(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25], int gapo, int gape, int *qe, int *te)
{
	const int m = 5, L = 8;
	const int slen = (qlen + L - 1) / L;
	*qe = *te = -1;
	if (slen <= 0) return 0;
	std::vector<int16_t> prof((size_t)m * slen * L), H0((size_t)slen * L, 0), H1((size_t)slen * L, 0), E((size_t)slen * L, 0), Hmax((size_t)slen * L, 0);
	for (int a = 0; a < m; ++a)
		for (int j = 0; j < slen; ++j)
			for (int l = 0; l < L; ++l) {
				const int k = j + l * slen;
				prof[((size_t)a * slen + j) * L + l] = k >= qlen ? -1 : mat[a * m + query[k]];
			}
	const int16_t go_e = (int16_t)(gapo + gape), ge = (int16_t)gape;
	int gmax = 0;
	int16_t *h0 = H0.data(), *h1 = H1.data();
	for (int i = 0; i < tlen; ++i) {
		int16_t f[L], h[L], mx[L];
		const int16_t *S = &prof[(size_t)target[i] * slen * L];
		for (int l = 0; l < L; ++l) f[l] = 0, mx[l] = 0;
		h[0] = 0;
		for (int l = 1; l < L; ++l) h[l] = h0[(size_t)(slen - 1) * L + l - 1]; // previous row, shifted by one lane
		for (int j = 0; j < slen; ++j) {
			for (int l = 0; l < L; ++l) {
				int16_t hv = adds16(h[l], S[(size_t)j * L + l]);
				int16_t e = E[(size_t)j * L + l];
				hv = max16(hv, e);
				hv = max16(hv, f[l]);
				mx[l] = max16(mx[l], hv);
				h1[(size_t)j * L + l] = hv;
				hv = subsu16(hv, go_e);
				e = subsu16(e, ge);
				E[(size_t)j * L + l] = max16(e, hv);
				f[l] = max16(subsu16(f[l], ge), hv);
				h[l] = h0[(size_t)j * L + l];
			}
		}
		// lazy-F: propagate vertical gaps across the stripe boundaries until nothing changes
		bool settled = false;
		for (int k = 0; k < L && !settled; ++k) {
			for (int l = L - 1; l > 0; --l) f[l] = f[l - 1];
			f[0] = 0;
			for (int j = 0; j < slen; ++j) {
				bool any = false;
				for (int l = 0; l < L; ++l) {
					int16_t hv = max16(h1[(size_t)j * L + l], f[l]);
					h1[(size_t)j * L + l] = hv;
					hv = subsu16(hv, go_e);
					f[l] = subsu16(f[l], ge);
					if (f[l] > hv) any = true;
				}
				if (!any) { settled = true; break; }
			}
		}
		int imax = mx[0];
		for (int l = 1; l < L; ++l) imax = imax > mx[l] ? imax : mx[l];
		if (imax >= gmax) {
			gmax = imax, *te = i;
			memcpy(Hmax.data(), h1, sizeof(int16_t) * slen * L);
		}
		std::swap(h0, h1);
	}
	for (int i = 0; i < slen * L; ++i)
		if ((int)(uint16_t)Hmax[i] == gmax) *qe = i / L + i % L * slen;
	return gmax;
}

namespace {
// One sweep over the anti-diagonals r = i + j of the Smith-Waterman matrix (i target, j query), arrays indexed by i.  row_max[i] (when given) receives the maximum
// of target row i; row_of / row_vals (when given) the scores H(row_of, 0..qlen) -- the sweep then stops after the last anti-diagonal that row touches.
// Returns false if a score left the range a 16-bit lane holds without saturating.
#if defined(__x86_64__) && defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default"))) // (16 scores per vector instruction where the host has AVX2; resolved once, at load time)
#endif
#if defined(__GNUC__) && !defined(__clang__)
__attribute__((optimize("O3", "tree-vectorize")))
#endif
bool ll_sweep(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25], int goe_, int ge_, int16_t *row_max, int row_of, int16_t *row_vals, int16_t *keep,
              int16_t *work /* 8 * (tlen + 2), zeroed */, uint8_t *qrev /* qlen */)
{
	// three score diagonals (r - 2, r - 1, r) and two of each gap state; one slot of padding below index 0 and one above the last: a neighbour outside the matrix
	// reads as zero there (every array starts zeroed, and the slot just past a diagonal's last cell is cleared when the diagonal is written)
	// (no containers in here: the function is cloned per instruction set, and the clones must not carry template instantiations of their own)
	const size_t W = (size_t)tlen + 2;
	int16_t *const sub = work + 7 * W;
	for (int j = 0; j < qlen; ++j) qrev[j] = query[qlen - 1 - j];
	int16_t *H2 = work + 1, *H1 = H2 + W, *H0 = H1 + W, *E1 = H0 + W, *E0 = E1 + W, *F1 = E0 + W, *F0 = F1 + W;
	// a plain DNA matrix (one score for a match, one for a mismatch, one whenever either base is ambiguous) is scored by comparisons; anything else by look-up
	bool simple = true;
	for (int x = 0; x < 5 && simple; ++x)
		for (int y = 0; y < 5; ++y) {
			const int want = (x == 4 || y == 4) ? mat[24] : x == y ? mat[0] : mat[1];
			if (mat[x * 5 + y] != want) { simple = false; break; }
		}
	const int16_t sc_mch = mat[0], sc_mis = mat[1], sc_n = mat[24], goe = (int16_t)goe_, ge = (int16_t)ge_;
	const int last_r = row_vals ? row_of + qlen - 1 : qlen + tlen - 2;
	int16_t top = 0;
	for (int r = 0; r <= last_r; ++r) {
		const int lo = r - qlen + 1 > 0 ? r - qlen + 1 : 0, hi = r < tlen - 1 ? r : tlen - 1;
		const uint8_t *__restrict tq = target, *__restrict qq = qrev + (qlen - 1 - r); // query[r - i] = qrev[qlen - 1 - r + i]
		int16_t *__restrict sb = sub;
		if (simple) for (int i = lo; i <= hi; ++i) { const uint8_t x = tq[i], y = qq[i]; sb[i] = (x | y) >= 4 ? sc_n : x == y ? sc_mch : sc_mis; }
		else for (int i = lo; i <= hi; ++i) sb[i] = mat[tq[i] * 5 + qq[i]];
		const int16_t *__restrict h1 = H1, *__restrict h2 = H2, *__restrict e1 = E1, *__restrict f1 = F1;
		int16_t *__restrict h0 = H0, *__restrict e0 = E0, *__restrict f0 = F0;
		int16_t dmax = 0;
		for (int i = lo; i <= hi; ++i) {
			// E: the gap arriving from (i - 1, j) -- diagonal r - 1, index i - 1; F: from (i, j - 1) -- diagonal r - 1, index i; the reference's unsigned saturating
			// subtraction is max(., 0) on these non-negative values
			int16_t e = (int16_t)(e1[i - 1] - ge), eh = (int16_t)(h1[i - 1] - goe), f = (int16_t)(f1[i] - ge), fh = (int16_t)(h1[i] - goe);
			e = e > eh ? e : eh, e = e > 0 ? e : (int16_t)0;
			f = f > fh ? f : fh, f = f > 0 ? f : (int16_t)0;
			int16_t h = (int16_t)(h2[i - 1] + sb[i]);
			h = h > e ? h : e, h = h > f ? h : f;
			e0[i] = e, f0[i] = f, h0[i] = h;
			dmax = dmax > h ? dmax : h;
		}
		top = top > dmax ? top : dmax;
		if (keep) { memcpy(keep, H0 + lo, sizeof(int16_t) * (size_t)(hi - lo + 1)); keep += hi - lo + 1; } // every score, diagonal after diagonal: the caller reads one row out of it
		if (row_max) for (int i = lo; i <= hi; ++i) row_max[i] = row_max[i] > H0[i] ? row_max[i] : H0[i];
		if (row_vals && row_of >= lo && row_of <= hi) row_vals[r - row_of] = H0[row_of];
		H0[hi + 1] = E0[hi + 1] = F0[hi + 1] = 0; // (the next diagonals read one slot past this one's last cell when they start a new target row)
		int16_t *t = H2; H2 = H1, H1 = H0, H0 = t;
		t = E1, E1 = E0, E0 = t;
		t = F1, F1 = F0, F0 = t;
	}
	return top < 32000;
}
}

int ll_local_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25], int gapo, int gape, int *qe, int *te)
{
	*qe = *te = -1;
	if (qlen <= 0 || tlen <= 0) return ll_local_score_striped(qlen, query, tlen, target, mat, gapo, gape, qe, te);
	int worst = 0, best = 0;
	for (int k = 0; k < 25; ++k) worst = std::min<int>(worst, mat[k]), best = std::max<int>(best, mat[k]);
	// the plain matrix is the striped routine's whenever opening a gap right after a gap of the other kind cannot beat a substitution, and nothing saturates
	// (gapo >= 1: with a free gap opening the striped routine itself leaves the plain matrix -- tests/cpucheck/ksw_ll_test.cpp shows both -- and stays the only form)
	const bool plain = -worst <= 2 * (gapo + gape) && gapo >= 1 && gape > 0 && (long)best * std::min(qlen, tlen) < 32000 && static_cast<unsigned>(gapo + gape) < 16000u;
	if (!plain) return ll_local_score_striped(qlen, query, tlen, target, mat, gapo, gape, qe, te);
	std::vector<int16_t> row_max((size_t)tlen, 0), row((size_t)qlen, 0);
	const size_t n_cells = (size_t)qlen * (size_t)tlen;
	std::vector<int16_t> all(n_cells <= ((size_t)24 << 20) ? n_cells : 0); // up to 48 MB: one sweep that keeps every score; beyond that a second sweep recomputes the row
	std::vector<int16_t> work((size_t)8 * ((size_t)tlen + 2), 0);
	std::vector<uint8_t> qrev((size_t)qlen);
	if (!ll_sweep(qlen, query, tlen, target, mat, gapo + gape, gape, row_max.data(), -1, nullptr, all.empty() ? nullptr : all.data(), work.data(), qrev.data())) return ll_local_score_striped(qlen, query, tlen, target, mat, gapo, gape, qe, te);
	int gmax = 0;
	for (int i = 0; i < tlen; ++i) if (row_max[i] >= gmax) gmax = row_max[i], *te = i; // the LAST row that reaches the running maximum (ksw2_ll_sse.c:143-146)
	if (gmax == 0) { *qe = 8 * ((qlen + 7) / 8) - 1; return 0; } // nothing scores: the striped scan's last slot holds the "maximum" too, padding included (:149-151)
	if (all.empty()) { std::fill(work.begin(), work.end(), (int16_t)0); ll_sweep(qlen, query, tlen, target, mat, gapo + gape, gape, nullptr, *te, row.data(), nullptr, work.data(), qrev.data()); }
	else { // H(te, j) sits on diagonal te + j at index te - lo(te + j); diagonal r starts at the sum of the lengths before it
		size_t off = 0;
		for (int r = 0; r < *te + qlen; ++r) {
			const int lo = r - qlen + 1 > 0 ? r - qlen + 1 : 0, hi = r < tlen - 1 ? r : tlen - 1;
			if (r >= *te) row[r - *te] = all[off + (size_t)(*te - lo)];
			off += (size_t)(hi - lo + 1);
		}
	}
	// the cell of that row the striped scan finds last (:149-151): positions are visited segment by segment (p mod slen), within a segment lane by lane (p / slen)
	const int slen = (qlen + 7) / 8;
	int last_key = -1;
	for (int p = 0; p < qlen; ++p)
		if (row[p] == gmax) { const int key = p % slen * 8 + p / slen; if (key > last_key) last_key = key, *qe = p; }
	return gmax;
}

} // namespace mm2amd

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

int ll_local_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25], int gapo, int gape, int *qe, int *te)
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

} // namespace mm2amd

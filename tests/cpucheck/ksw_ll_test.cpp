// TEST INFRASTRUCTURE.  minimap2_amd/csrc/ksw_ll.cpp's ll_local_score -- the inversion test's local alignment score with end coordinates (align.c:86-103) -- against
// the reference's own ksw_ll_qinit + ksw_ll_i16 (ksw2_ll_sse.c:37-152, from oracle/_ref/libminimap2_ref.a): score, query end and target end on random, related,
// shifted and N-bearing sequence pairs under random scorings.  Round 6's anti-diagonal form answers for scorings with gapo >= 1 and b <= 2 (gapo + gape); outside that
// (a free gap opening: the striped routine itself leaves the plain Smith-Waterman matrix there) and when nothing scores, the lane-by-lane form or the padding rule
// must -- the cases cover all three, and the test says how many took which.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <random>
#include "../../minimap2_amd/csrc/align.hpp"

extern "C" {
void *ksw_ll_qinit(void *km, int size, int qlen, const uint8_t *query, int m, const int8_t *mat);
int ksw_ll_i16(void *q, int tlen, const uint8_t *target, int gapo, int gape, int *qe, int *te);
void kfree(void *km, void *p);
}

int main(int argc, char **argv)
{
	const int n_cases = argc > 1 ? atoi(argv[1]) : 30000;
	std::mt19937 r(7);
	long n_bad = 0, n_free_open = 0, n_zero = 0, n_plain = 0;
	for (int it = 0; it < n_cases; ++it) {
		const int ql = 1 + r() % 300, tl = 1 + r() % 300, mode = it % 4;
		std::vector<uint8_t> q(ql), t(tl);
		for (auto &c : t) c = r() % (mode == 3 ? 5 : 4);
		if (mode == 0) for (auto &c : q) c = r() % 4;
		else for (int i = 0; i < ql; ++i) q[i] = (i < tl && r() % 8) ? t[(i + (mode == 2 ? 5 : 0)) % tl] : r() % 4;
		const int a = 1 + r() % 3, b = 1 + r() % 8, go = r() % 8, ge = 1 + r() % 3, scn = (r() % 2) ? -1 : -(int)(r() % 3);
		int8_t mat[25];
		for (int x = 0; x < 5; ++x) for (int y = 0; y < 5; ++y) mat[x * 5 + y] = (x == 4 || y == 4) ? scn : x == y ? a : -b;
		if (it % 7 == 0) mat[1 * 5 + 2] = mat[2 * 5 + 1] = -1; // a transition score: the look-up path
		int qe1, te1, qe2, te2;
		const int s1 = mm2amd::ll_local_score(ql, q.data(), tl, t.data(), mat, go, ge, &qe1, &te1);
		void *qp = ksw_ll_qinit(0, 2, ql, q.data(), 5, mat);
		const int s2 = ksw_ll_i16(qp, tl, t.data(), go, ge, &qe2, &te2);
		kfree(0, qp);
		n_free_open += go == 0, n_zero += s2 == 0, n_plain += go >= 1 && b <= 2 * (go + ge) && s2 > 0;
		if (s1 != s2 || qe1 != qe2 || te1 != te2) {
			if (n_bad++ < 5) fprintf(stderr, "case %d (q %d, t %d, a %d b %d gapo %d gape %d): score %d qe %d te %d, the reference %d %d %d\n", it, ql, tl, a, b, go, ge, s1, qe1, te1, s2, qe2, te2);
		}
	}
	if (n_bad || !n_free_open || !n_zero || !n_plain) { fprintf(stderr, "%ld of %d cases differ (free opening %ld, nothing scores %ld, anti-diagonal form %ld)\n", n_bad, n_cases, n_free_open, n_zero, n_plain); return 1; }
	printf("OK %d cases == ksw_ll_i16: %ld through the anti-diagonal form, %ld with a free gap opening, %ld where nothing scores\n", n_cases, n_plain, n_free_open, n_zero);
	return 0;
}

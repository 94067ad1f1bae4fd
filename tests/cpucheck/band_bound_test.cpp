// TEST INFRASTRUCTURE.  ksw_band.hpp's acceptance rule against brute force: for small windows and small bands (the rule does not know that the kernel's band has
// 128 or 256 diagonals), the best score ANY alignment that touches a cell outside the band can reach -- every aligned pair scoring the best substitution score,
// gaps at the dual affine cost -- is computed by dynamic programming over (cell, last move, touched-outside) and must
//   (1) never exceed band_outside_bound()  (the rule is sound), and
//   (2) reach it for windows where an outside cell exists  (the rule is tight: equality happens, hence the kernel's STRICT comparison).
// Also checked: band_holds_corners() is what it says, the band's geometry puts both corners inside with margins that differ by at most 2.
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <vector>
#include <algorithm>
#include "../../minimap2_amd/csrc/ksw_band.hpp"

using namespace mm2amd;

static int check(int W, int a, int q, int e, int q2, int e2, int max_len, long &n_windows, long &n_tight)
{
	const int NEG = INT_MIN / 4;
	for (int ql = 1; ql <= max_len; ++ql)
		for (int tl = 1; tl <= max_len; ++tl) {
			const int D = tl - ql, c = band_c(ql, tl, W), dlo = -2 * c, dhi = dlo + W - 1;
			const bool holds = dlo <= 0 && dlo <= D && dhi >= 0 && dhi >= D;
			if (holds != band_holds_corners(ql, tl, W)) { fprintf(stderr, "band_holds_corners(%d, %d, %d) disagrees\n", ql, tl, W); return 1; }
			if (!holds) continue;
			{ // the margins below min(0, D) and above max(0, D) are balanced
				const int below = std::min(0, D) - dlo, above = dhi - std::max(0, D);
				if (std::abs(below - above) > 3) { fprintf(stderr, "unbalanced band: q %d t %d W %d: %d below, %d above\n", ql, tl, W, below, above); return 1; }
			}
			// best[(i + 1) * (ql + 1) + (j + 1)][move][touched], i in -1 .. tl - 1, j in -1 .. ql - 1; move: 0 aligned pair (or the origin), 1 gap over target bases, 2 over query bases
			const int NI = tl + 1, NJ = ql + 1;
			std::vector<int> best((size_t)NI * NJ * 3 * 2, NEG);
			auto at = [&](int i, int j, int mv, int touched) -> int & { return best[(((size_t)(i + 1) * NJ + (j + 1)) * 3 + mv) * 2 + touched]; };
			auto outside = [&](int i, int j) { return i >= 0 && j >= 0 && (i - j < dlo || i - j > dhi); }; // (border positions are not cells)
			at(-1, -1, 0, 0) = 0;
			for (int i = -1; i < tl; ++i)
				for (int j = -1; j < ql; ++j)
					for (int mv = 0; mv < 3; ++mv)
						for (int tc = 0; tc < 2; ++tc) {
							const int v = at(i, j, mv, tc);
							if (v == NEG) continue;
							if (i + 1 < tl && j + 1 < ql) { int &d = at(i + 1, j + 1, 0, tc | outside(i + 1, j + 1)); d = std::max(d, v + a); }
							if (mv != 1) for (int L = 1; i + L < tl; ++L) { // the gap's cells (i + 1 .. i + L, j) all count as touched
								int t2 = tc;
								for (int k = 1; k <= L; ++k) t2 |= outside(i + k, j);
								int &d = at(i + L, j, 1, t2); d = std::max(d, v - band_gap_cost(L, q, e, q2, e2));
							}
							if (mv != 2) for (int L = 1; j + L < ql; ++L) {
								int t2 = tc;
								for (int k = 1; k <= L; ++k) t2 |= outside(i, j + k);
								int &d = at(i, j + L, 2, t2); d = std::max(d, v - band_gap_cost(L, q, e, q2, e2));
							}
						}
			int top = NEG;
			for (int mv = 0; mv < 3; ++mv) top = std::max(top, at(tl - 1, ql - 1, mv, 1));
			const int ub = band_outside_bound(ql, tl, W, a, q, e, q2, e2);
			++n_windows;
			if (ub == INT32_MIN) { if (top != NEG) { fprintf(stderr, "q %d t %d W %d: a path goes outside (%d) but the bound says no cell is\n", ql, tl, W, top); return 1; } continue; }
			if (top > ub) { fprintf(stderr, "UNSOUND: q %d t %d W %d (a %d, gaps %d %d %d %d): a path through an outside cell scores %d > bound %d\n", ql, tl, W, a, q, e, q2, e2, top, ub); return 1; }
			if (top == ub) ++n_tight;
		}
	return 0;
}

int main()
{
	long n_windows = 0, n_tight = 0;
	const int sc[][5] = { { 2, 4, 2, 24, 1 }, { 1, 6, 2, 26, 1 }, { 2, 24, 1, 4, 2 }, { 1, 39, 3, 81, 1 }, { 2, 2, 1, 2, 1 }, { 3, 1, 1, 5, 0 } }; // match score; q, e, q2, e2
	for (const auto &s : sc)
		for (int W : { 4, 8, 12, 20 })
			if (check(W, s[0], s[1], s[2], s[3], s[4], 18, n_windows, n_tight)) return 1;
	if (n_tight * 4 < n_windows) { fprintf(stderr, "the bound is reached in only %ld of %ld windows: it should be tight wherever an outside cell exists and a path can use it\n", n_tight, n_windows); return 1; }
	printf("band_outside_bound: sound on %ld windows, reached exactly in %ld\n", n_windows, n_tight);
	return 0;
}

// Geometry and acceptance test of the BANDED gap-fill kernel (ksw_band.hip), shared by the kernel, by the launch classes (ksw_classify.hpp: which
// windows are worth trying in a band) and by the host-side check of the rule (tests/cpucheck/band_bound_test.cpp).
//
// The reference fills the gap between two anchors with ksw_extd2_sse and a band that cannot bind (align.c:810-844: bw_long), i.e. it computes the
// whole qlen x tlen rectangle.  A cell far from the diagonals of the two corners cannot lie on an optimal alignment, and the reference's traceback
// only ever visits cells of an optimal alignment.  ksw_band.hip computes the W = 128 * NB diagonals d = target index - query index in [dlo, dlo + W)
// around the corners' diagonals 0 and D = tlen - qlen, nothing else, and ACCEPTS its result only when the score it found proves that nothing outside
// that band can matter (band_outside_bound below); a window that fails the test is computed again in a wider band or as the full rectangle by the
// kernels that were there before.  So the band is a question of speed only: what leaves the kernel is what the full rectangle gives.
//
// Why an accepted result is the rectangle's (DESIGN.md section 4 has the argument in full):
//   (1) every value the banded recurrences hold is the score of some real alignment prefix (possibly one that steps onto a cell just outside the
//       band: the band's edge constants are the reference's own -- a neighbour that was not computed counts as "open a gap from there", -(q + e),
//       ksw2_extd2_sse.c:111-116, :148-155), or lower (the clamp z <= match score); so banded <= full, cell by cell and state by state;
//   (2) an alignment that touches a cell outside the band scores at most band_outside_bound() -- on diagonal d > max(0, D) at most tlen - d columns can
//       be matches, and the path needs d target-only moves to get there and d - D query-only moves to get back: a (tlen - d) - gap(d) - gap(d - D),
//       gap(l) = min(q + e l, q2 + e2 l) being concave hence subadditive over several gaps; symmetrically below min(0, D); both fall with the distance,
//       so the two diagonals next to the band decide;
//   (3) the banded score S is the score of a real alignment, so S <= optimum.  If S > bound (STRICTLY), every alignment through an outside cell scores
//       less than the optimum: all optimal alignments -- and all prefixes that TIE with a prefix of one, which is what the reference's "first of
//       (diagonal, E, F, E2, F2) that reaches the maximum" rule looks at -- lie inside the band, where (1) holds with equality; the candidate that
//       wins a cell of the traced path wins it with the same value against candidates that are equal (then also exact) or lower.  Hence the same
//       direction byte at every cell the traceback reads, the same CIGAR and the same corner score.
#pragma once
#include <cstdint>
#include "exact_rsort.hpp" // MM2_HD

namespace mm2amd {

// lanes hold diagonal pairs: band index k = d - dlo in [0, W), lane k >> 1 of register set k >> 7; dlo = -2 c is even, c = band_c() centres [dlo, dlo + W) on D / 2
MM2_HD inline int band_c(int qlen, int tlen, int W) { return (W - (tlen - qlen)) >> 2; }
MM2_HD inline int band_gap_cost(int l, int q, int e, int q2, int e2) { if (l <= 0) return 0; const int a = q + e * l, b = q2 + e2 * l; return a < b ? a : b; }
MM2_HD inline bool band_holds_corners(int qlen, int tlen, int W)
{
	const int D = tlen - qlen, dlo = -2 * band_c(qlen, tlen, W), dhi = dlo + W - 1;
	return dlo <= 0 && dlo <= D && dhi >= 0 && dhi >= D;
}
// the most an alignment of the window can score if it touches a cell outside the band (sc_max = the largest substitution score, >= 0); INT32_MIN: no cell is outside
MM2_HD inline int band_outside_bound(int qlen, int tlen, int W, int sc_max, int q, int e, int q2, int e2)
{
	const int D = tlen - qlen, dlo = -2 * band_c(qlen, tlen, W), dhi = dlo + W - 1;
	int ub = INT32_MIN;
	const int dp = dhi + 1;  // the first diagonal above: cells (i, i - dp), dp <= i <= tlen - 1
	if (dp <= tlen - 1) {
		const int v = sc_max * (tlen - dp) - band_gap_cost(dp, q, e, q2, e2) - band_gap_cost(dp - D, q, e, q2, e2);
		ub = v > ub ? v : ub;
	}
	const int dm = 1 - dlo;  // the first diagonal below is -dm: cells (j - dm, j), dm <= j <= qlen - 1
	if (dm <= qlen - 1) {
		const int v = sc_max * (qlen - dm) - band_gap_cost(dm, q, e, q2, e2) - band_gap_cost(dm + D, q, e, q2, e2);
		ub = v > ub ? v : ub;
	}
	return ub;
}

} // namespace mm2amd

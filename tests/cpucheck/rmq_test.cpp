// TEST INFRASTRUCTURE.  minimap2_amd/csrc/rmq_chain.cpp (chain_rmq: the tie-exact range-minimum tree + the narrow-window scan)
// against the reference's own mg_lchain_rmq (lchain.c:250-368 over krmq.h, linked from oracle/_ref/libminimap2_ref.a) on anchor
// sets built to make priorities TIE -- pen_gap 0 (priority = -score), anchors on a lattice, equal spans, repeated coordinates --
// and to exercise every removal path: small max_dist (constant eviction), a size cap below the window, several target sequences.
// Prints "OK <cases> <anchors>" or fails.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>
#include "../../minimap2_amd/csrc/chain_host.hpp"

typedef struct { uint64_t x, y; } mm128_t;
extern "C" mm128_t *mg_lchain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc, float chn_pen_gap,
                                  float chn_pen_skip, int64_t n, mm128_t *a, int *n_u_, uint64_t **_u, void *km);

int main(int argc, char **argv)
{
	const int n_case = argc > 1 ? atoi(argv[1]) : 4000;
	std::mt19937_64 rng(11);
	long n_anchor = 0;
	for (int it = 0; it < n_case; ++it) {
		const int n = 1 + (int)(rng() % (it % 20 == 0 ? 6000 : 400));
		const int grid = (int[]){ 1, 1, 5, 15, 50 }[rng() % 5];       // coarse grids put many anchors on the same anti-diagonal
		const int span = (int[]){ 15, 15, 19, 10 }[rng() % 4];
		const int n_rid = 1 + (int)(rng() % 3), range = 200 + (int)(rng() % 30000);
		std::vector<mm2amd::Anchor> a(n);
		for (int i = 0; i < n; ++i) {
			const uint64_t rid = rng() % n_rid, x = (rng() % range) / grid * grid + span;
			uint64_t y = (it % 3 == 0 ? x + (rng() % 7 - 3) * grid : rng() % range / grid * grid) + span; // near the diagonal, or anywhere
			a[i].x = rid << 32 | x;
			a[i].y = (uint64_t)(it % 5 == 0 ? span : 8 + rng() % 20) << 32 | (y & 0x7fffffff);
		}
		std::sort(a.begin(), a.end(), [](const mm2amd::Anchor &p, const mm2amd::Anchor &q) { return p.x < q.x || (p.x == q.x && p.y < q.y); });
		a[0].y = (a[0].y >> 32) << 32 | 0x7ffffff0u; // the reference's upper key (y_i, 0) would admit anchor 0 at the SAME query coordinate and then trip its own assertion (lchain.c:352)
		const int max_dist = (int[]){ 100, 500, 5000, 20000 }[rng() % 4], inner = (int[]){ 0, 50, 1000, 30000 }[rng() % 4];
		const int bw = (int[]){ 50, 500, 20000 }[rng() % 3], cap = (int[]){ 4, 37, 1000, 100000 }[rng() % 4], skip = (int[]){ 0, 3, 25 }[rng() % 3];
		const float pen_gap = (float[]){ 0.f, 0.f, 0.12f, 1.f }[rng() % 4], pen_skip = (float[]){ 0.f, 0.15f }[rng() % 2];
		const int min_cnt = 1 + (int)(rng() % 3), min_sc = (int[]){ 0, 20, 40 }[rng() % 3];
		mm128_t *ra = (mm128_t *)malloc(sizeof(mm128_t) * n);
		memcpy(ra, a.data(), sizeof(mm128_t) * n);
		int n_u_ref = 0;
		uint64_t *u_ref = nullptr;
		mm128_t *out_ref = mg_lchain_rmq(max_dist, inner, bw, skip, cap, min_cnt, min_sc, pen_gap, pen_skip, n, ra, &n_u_ref, &u_ref, nullptr);
		std::vector<uint64_t> u;
		std::vector<mm2amd::Anchor> out;
		mm2amd::ChainScratch sc;
		mm2amd::chain_rmq(max_dist, inner, bw, skip, cap, min_cnt, min_sc, pen_gap, pen_skip, n, a.data(), u, out, sc);
		size_t n_out_ref = 0;
		for (int k = 0; k < n_u_ref; ++k) n_out_ref += (uint32_t)u_ref[k];
		const bool same = (int)u.size() == n_u_ref && (n_u_ref == 0 || memcmp(u.data(), u_ref, 8 * (size_t)n_u_ref) == 0) && out.size() == n_out_ref &&
		                  (n_out_ref == 0 || memcmp(out.data(), out_ref, 16 * n_out_ref) == 0);
		if (!same) {
			fprintf(stderr, "case %d: n %d max_dist %d inner %d bw %d cap %d skip %d pen %g %g: chains %zu vs %d, anchors %zu vs %zu\n", it, n, max_dist, inner, bw, cap, skip,
			        pen_gap, pen_skip, u.size(), n_u_ref, out.size(), n_out_ref);
			return 1;
		}
		free(u_ref), free(out_ref);
		n_anchor += n;
	}
	printf("OK %d %ld\n", n_case, n_anchor);
	return 0;
}

// Host-side pieces of chaining: the backtrack/compaction that turns the DP arrays (f, p) produced by the
// chain-fill kernel into chains (mg_chain_backtrack + compact_a, lchain.c:9-111), and the RMQ re-chaining used
// by the long-join branch (mg_lchain_rmq, lchain.c:250-368).
#pragma once
#include <vector>
#include "types.hpp"

namespace mm2amd {

struct ChainScratch {
	std::vector<int32_t> t, v;
	std::vector<Anchor> z, b, w;
	std::vector<uint64_t> u2;
};

// Input: sorted anchors a[0..n), DP results f[], p[] (p as 32-bit index, -1 = none).
// Output: u (score<<32|cnt per chain) and the compacted anchors, chain by chain, chains ordered by target position.
void chain_backtrack_compact(int64_t n, const Anchor *a, const int32_t *f, const int32_t *p, int min_cnt, int min_sc, int max_drop,
                             std::vector<uint64_t> &u, std::vector<Anchor> &out, ChainScratch &sc);

// mg_lchain_rmq: anchors a[0..n) sorted by x; results as above.
void chain_rmq(int max_dist, int max_dist_inner, int bw, int max_chn_skip, int cap_rmq_size, int min_cnt, int min_sc,
               float chn_pen_gap, float chn_pen_skip, int64_t n, const Anchor *a, std::vector<uint64_t> &u, std::vector<Anchor> &out,
               ChainScratch &sc);

// mg_log2 (mmpriv.h:139-147)
inline float fast_log2(float x)
{
	union { float f; uint32_t i; } z = { x };
	float l = (float)((int)((z.i >> 23) & 255) - 128);
	z.i &= ~(255u << 23);
	z.i += 127u << 23;
	l += (-0.34484843f * z.f + 2.02466578f) * z.f - 0.67487759f;
	return l;
}

// where the chain ending at anchor `end` with score `total` is cut (chain_host.cpp; exposed for tests/cpucheck/region_rules_test.cpp)
int64_t chain_cut(int32_t max_drop, int32_t total, int64_t end, const int32_t *f, const int32_t *p, const int32_t *t);

} // namespace mm2amd

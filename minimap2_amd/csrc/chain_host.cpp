#include <cstring>
#include <cassert>
#include "chain_host.hpp"

namespace mm2amd {

// Where the chain that ends at anchor `end` (chaining score `total`) is cut (what mg_chain_bk_end, lchain.c:9-25, returns): following the predecessor links, the
// chain's score SINCE an anchor is total - f[that anchor]; the cut goes to the anchor where that gain peaks, and the walk gives up once the gain has fallen
// max_drop below its peak or reaches an anchor another chain has claimed (t != 0).  Links only point backwards, so no anchor is seen twice and nothing needs
// marking on the way (the reference marks and unmarks).  Returns the anchor BEFORE the chain's first (-1: the chain starts the read's anchors).
int64_t chain_cut(int32_t max_drop, int32_t total, int64_t end, const int32_t *f, const int32_t *p, const int32_t *t)
{
	if (end < 0 || t[end] != 0) return end;
	int64_t cut = end;
	int32_t peak = 0;
	for (int64_t i = end;;) {
		i = p[i];
		const int32_t gain = i < 0 ? total : total - f[i];
		if (gain > peak) peak = gain, cut = i;
		else if (peak - gain > max_drop) break;
		if (i < 0 || t[i] != 0) break;
	}
	return cut;
}

void chain_backtrack_compact(int64_t n, const Anchor *a, const int32_t *f, const int32_t *p, int min_cnt, int min_sc, int max_drop,
                             std::vector<uint64_t> &u, std::vector<Anchor> &out, ChainScratch &sc)
{
	u.clear(); out.clear();
	if (n <= 0) return;
	// chain ends, best first (lchain.c:35-41).  Equal scores are ordered by the reference's unstable sort.
	sc.z.clear();
	for (int64_t i = 0; i < n; ++i)
		if (f[i] >= min_sc) sc.z.push_back(Anchor{(uint64_t)f[i], (uint64_t)i});
	const int64_t n_z = (int64_t)sc.z.size();
	if (n_z == 0) return;
	sort_by_x(sc.z.data(), sc.z.data() + n_z);
	sc.t.assign(n, 0);
	sc.v.clear();
	for (int64_t k = n_z - 1; k >= 0; --k) { // (lchain.c:57-72; the counting pre-pass :43-56 computes the same thing)
		if (sc.t[sc.z[k].y] != 0) continue;
		const size_t n_v0 = sc.v.size();
		const int64_t end_i = chain_cut(max_drop, (int32_t)sc.z[k].x, (int64_t)sc.z[k].y, f, p, sc.t.data());
		int64_t i;
		for (i = (int64_t)sc.z[k].y; i != end_i; i = p[i]) sc.v.push_back((int32_t)i), sc.t[i] = 1;
		const int32_t s = i < 0 ? (int32_t)sc.z[k].x : (int32_t)sc.z[k].x - f[i];
		if (s >= min_sc && sc.v.size() > n_v0 && (int64_t)(sc.v.size() - n_v0) >= min_cnt) u.push_back((uint64_t)s << 32 | (uint64_t)(sc.v.size() - n_v0));
		else sc.v.resize(n_v0);
	}
	const size_t n_u = u.size();
	if (n_u == 0) return;
	// anchors of each chain in ascending order (lchain.c:84-91)
	sc.b.resize(sc.v.size());
	size_t k = 0;
	for (size_t i = 0; i < n_u; ++i) {
		const size_t k0 = k, ni = (uint32_t)u[i];
		for (size_t j = 0; j < ni; ++j) sc.b[k++] = a[sc.v[k0 + (ni - j - 1)]];
	}
	// chains ordered by the target position of their first anchor (lchain.c:93-106)
	sc.w.resize(n_u);
	k = 0;
	for (size_t i = 0; i < n_u; ++i) { sc.w[i] = Anchor{sc.b[k].x, (uint64_t)k << 32 | (uint64_t)i}; k += (uint32_t)u[i]; }
	sort_by_x(sc.w.data(), sc.w.data() + n_u);
	sc.u2.resize(n_u);
	out.resize(sc.b.size());
	k = 0;
	for (size_t i = 0; i < n_u; ++i) {
		const uint32_t j = (uint32_t)sc.w[i].y, cnt = (uint32_t)u[j];
		sc.u2[i] = u[j];
		memcpy(&out[k], &sc.b[sc.w[i].y >> 32], cnt * sizeof(Anchor));
		k += cnt;
	}
	u.assign(sc.u2.begin(), sc.u2.end());
}

} // namespace mm2amd

// TEST INFRASTRUCTURE.  minimap2_amd/csrc/region_rules.hpp (the chain -> window rules shared by region_plan_kernel and align.cpp), chain_host.cpp's chain_cut and
// align.cpp's append_cigar against the reference's OWN static functions -- mm_filter_bad_seeds, mm_filter_bad_seeds_alt, mm_fix_bad_ends, mm_max_stretch, mm_append_cigar
// (align.c) and mg_chain_bk_end (lchain.c), compiled from the reference's sources by oracle/ref_align_shim.c / ref_lchain_shim.c -- on random chains built to
// reach the rules: long gaps that compensate each other within and beyond the reach limits, clusters of gaps with little room between them, chain ends that sit
// off the diagonal, long-join flags, spans that vary (homopolymer-compressed seeds).  Each case must also take effect somewhere (the test fails if no case
// flags a seed, joins a cluster, trims an end or cuts a chain: the rules would not have been reached).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <random>
#include <vector>
#include "../../minimap2_amd/csrc/region_rules.hpp"
#include "../../minimap2_amd/csrc/align.hpp"
#include "../../minimap2_amd/csrc/chain_host.hpp"

using namespace mm2amd;

extern "C" {
void refshim_filter_bad_seeds(int as1, int cnt1, void *a, int min_gap, int diff_thres, int max_ext_len, int max_ext_cnt);
void refshim_filter_bad_seeds_alt(int as1, int cnt1, void *a, int min_gap, int max_ext);
void refshim_max_stretch(int as, int cnt, const void *a, int32_t *as1, int32_t *cnt1);
void refshim_fix_bad_ends(int as, int cnt, int mlen, const void *a, int bw, int min_match, int32_t *as1, int32_t *cnt1);
int refshim_append_cigar(int n_head, const uint32_t *head, int n_tail, const uint32_t *tail, uint32_t *out);
int64_t refshim_chain_bk_end(int32_t max_drop, int32_t z_x, int64_t z_y, const int32_t *f, const int64_t *p, int32_t *t);
}


static std::mt19937_64 rng(20260930);
static int rnd(int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); }

// a chain of n anchors on one diagonal, with long gaps (some compensating, some clustered) sprinkled in
static std::vector<Anchor> random_chain(int n, int flavour)
{
	std::vector<Anchor> a((size_t)n);
	int32_t x = rnd(100, 5000), y = rnd(20, 300);
	int pending = 0; // a gap to be compensated by an opposite one soon
	for (int i = 0; i < n; ++i) {
		const int span = flavour == 3 ? rnd(15, 40) : 15;
		int dx = rnd(8, flavour == 1 ? 60 : 200), dy = dx;
		const int roll = rnd(0, 99);
		if (pending && rnd(0, 2) == 0) { if (pending > 0) dx += pending; else dy -= pending; pending = 0; } // the opposite gap
		else if (roll < (flavour == 2 ? 25 : 10)) { const int g = rnd(11, 400); if (rnd(0, 1)) dy += g, pending = rnd(0, 1) ? g + rnd(-5, 5) : 0; else dx += g, pending = rnd(0, 1) ? -(g + rnd(-5, 5)) : 0; }
		else if (roll < 20) { const int g = rnd(1, 9); if (rnd(0, 1)) dy += g; else dx += g; }
		if (i) x += dx, y += dy;
		a[i].x = (uint64_t)1 << 32 | (uint32_t)x;
		a[i].y = (uint64_t)span << 32 | (uint32_t)y;
		if (flavour == 4 && i > 0 && rnd(0, 30) == 0) a[i].y |= ref::SEED_LONG_JOIN;
	}
	return a;
}

static std::vector<int32_t> sites_of(const Anchor *chain, int cnt1, int min_gap)
{
	std::vector<int32_t> K;
	for (int i = 1; i < cnt1; ++i) if (rr_is_long_gap(chain, i, min_gap)) K.push_back(i);
	if (K.size() <= 1) K.clear();
	return K;
}

int main()
{
	long n_flagged = 0, n_joined = 0, n_trimmed = 0, n_cut = 0, n_merged = 0, n_runs = 0;
	for (int it = 0; it < 40000; ++it) {
		const int n = rnd(3, 120), flavour = it % 5;
		std::vector<Anchor> base = random_chain(n, flavour);
		const int as1 = rnd(0, 2) == 0 ? rnd(0, n / 3) : 0, cnt1 = n - as1 - (rnd(0, 2) == 0 ? rnd(0, (n - as1) / 3) : 0);
		if (cnt1 < 2) continue;
		const int max_gap = rnd(0, 3) == 0 ? rnd(50, 800) : 5000;
		{ // filter 1, then filter 2 on its output, as mm_align1 runs them (align.c:670-673)
			std::vector<Anchor> mine = base, theirs = base;
			const int thres = rnd(0, 4) == 0 ? rnd(0, 100) : 40, max_cnt = rnd(0, 4) == 0 ? rnd(1, 4) : 10;
			std::vector<int32_t> K = sites_of(mine.data() + as1, cnt1, 10);
			rr_drop_compensating_gaps(mine.data() + as1, K.data(), (int)K.size(), thres, max_gap >> 1, max_cnt);
			refshim_filter_bad_seeds(as1, cnt1, theirs.data(), 10, thres, max_gap >> 1, max_cnt);
			if (memcmp(mine.data(), theirs.data(), sizeof(Anchor) * (size_t)n)) { fprintf(stderr, "case %d: rr_drop_compensating_gaps differs from mm_filter_bad_seeds\n", it); return 1; }
			for (int i = 0; i < n; ++i) n_flagged += (mine[i].y & ref::SEED_IGNORE) != 0;
			K = sites_of(mine.data() + as1, cnt1, 30);
			rr_join_gap_clusters(mine.data() + as1, K.data(), (int)K.size(), max_gap >> 1);
			refshim_filter_bad_seeds_alt(as1, cnt1, theirs.data(), 30, max_gap >> 1);
			if (memcmp(mine.data(), theirs.data(), sizeof(Anchor) * (size_t)n)) { fprintf(stderr, "case %d: rr_join_gap_clusters differs from mm_filter_bad_seeds_alt\n", it); return 1; }
			for (int i = 0; i < n; ++i) n_joined += (mine[i].y & ref::SEED_LONG_JOIN) != 0 && !(base[i].y & ref::SEED_LONG_JOIN);
		}
		{ // end trimming
			ref::Reg1 r;
			memset(&r, 0, sizeof r);
			r.as = as1, r.cnt = cnt1, r.mlen = rnd(0, 3) == 0 ? rnd(10, 400) : 15 * cnt1;
			const int bw = rnd(0, 3) == 0 ? rnd(10, 200) : 500, min_match = rnd(0, 3) == 0 ? rnd(10, 100) : 80;
			int32_t a0, c0, a1, c1;
			rr_trim_ends(r, base.data(), bw, min_match, &a0, &c0);
			refshim_fix_bad_ends(as1, cnt1, r.mlen, base.data(), bw, min_match, &a1, &c1);
			if (a0 != a1 || c0 != c1) { fprintf(stderr, "case %d: rr_trim_ends (%d, %d) differs from mm_fix_bad_ends (%d, %d)\n", it, a0, c0, a1, c1); return 1; }
			n_trimmed += a0 != as1 || c0 != cnt1;
		}
		{ // the best run on one diagonal (short reads)
			ref::Reg1 r;
			memset(&r, 0, sizeof r);
			r.as = as1, r.cnt = cnt1;
			int32_t a0, c0, a1, c1;
			rr_best_diagonal_run(r, base.data(), &a0, &c0);
			refshim_max_stretch(as1, cnt1, base.data(), &a1, &c1);
			if (a0 != a1 || c0 != c1) { fprintf(stderr, "case %d: rr_best_diagonal_run (%d, %d) differs from mm_max_stretch (%d, %d)\n", it, a0, c0, a1, c1); return 1; }
			n_runs += c0 > 1 && c0 < cnt1;
		}
		{ // chain cut: random scores and links
			std::vector<int32_t> f((size_t)n), p32((size_t)n), t((size_t)n), t2;
			std::vector<int64_t> p64((size_t)n);
			for (int i = 0; i < n; ++i) {
				p32[i] = rnd(0, 9) == 0 ? -1 : (i == 0 ? -1 : rnd(std::max(0, i - 4), i - 1)), p64[i] = p32[i];
				f[i] = (p32[i] < 0 ? 0 : f[p32[i]]) + rnd(-40, 60);
				t[i] = rnd(0, 7) == 0 ? 1 : 0;
			}
			const int end = rnd(0, n - 1), max_drop = rnd(0, 2) == 0 ? rnd(0, 60) : 300;
			t2 = t;
			const int64_t mine = chain_cut(max_drop, f[end], end, f.data(), p32.data(), t.data());
			const int64_t theirs = refshim_chain_bk_end(max_drop, f[end], end, f.data(), p64.data(), t2.data());
			if (mine != theirs || t2 != t) { fprintf(stderr, "case %d: chain_cut %ld differs from mg_chain_bk_end %ld (or the marks were left)\n", it, (long)mine, (long)theirs); return 1; }
			n_cut += mine != end && mine != -1 && t[end] == 0;
		}
		{ // append_cigar
			const int nh = rnd(0, 6), nt = rnd(0, 6);
			std::vector<uint32_t> head((size_t)nh), tail((size_t)nt), out((size_t)(nh + nt + 1));
			for (auto &c : head) c = (uint32_t)rnd(1, 300) << 4 | (uint32_t)rnd(0, 2);
			for (auto &c : tail) c = (uint32_t)rnd(1, 300) << 4 | (uint32_t)rnd(0, 2);
			const int want = refshim_append_cigar(nh, head.data(), nt, tail.data(), out.data());
			ref::Reg1 r;
			memset(&r, 0, sizeof r);
			append_cigar(r, (uint32_t)nh, head.data());
			append_cigar(r, (uint32_t)nt, tail.data());
			const int got = r.p ? (int)r.p->n_cigar : 0;
			if (got != want || (got && memcmp(r.p->cigar, out.data(), (size_t)got * 4))) { fprintf(stderr, "case %d: append_cigar differs from mm_append_cigar\n", it); return 1; }
			n_merged += got < nh + nt;
			free(r.p);
		}
	}
	if (!n_flagged || !n_joined || !n_trimmed || !n_cut || !n_merged || !n_runs) { fprintf(stderr, "a rule was never reached: flagged %ld joined %ld trimmed %ld cut %ld merged %ld\n", n_flagged, n_joined, n_trimmed, n_cut, n_merged); return 1; }
	printf("region rules == the reference's statics: %ld seeds flagged, %ld clusters joined, %ld ends trimmed, %ld chains cut, %ld CIGAR joins\n", n_flagged, n_joined, n_trimmed, n_cut, n_merged);
	return 0;
}

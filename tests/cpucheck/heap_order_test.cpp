// TEST INFRASTRUCTURE.  Host build of the header the device kernel anchor_heap_order_kernel is made of
// (minimap2_amd/csrc/heap_order.hpp), checked against the oracle's restatement of collect_seed_hits_heap (oracle/seed.c, itself
// pinned to the reference's --print-seeds output by tests/test_host_pipeline.py): random seed sets in which several query
// minimizers share one k-mer, so that equal index entries meet in the heap.  Prints "OK <cases> <cases with ties>" or fails.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>
#include "../../minimap2_amd/csrc/heap_order.hpp"
#include "../../oracle/oracle.h"

using namespace mm2amd;

struct Idx { std::map<uint64_t, std::vector<uint64_t>> lists; };
static const uint64_t *idx_get(const void *idx, uint64_t minier, int *n)
{
	const Idx *I = (const Idx *)idx;
	auto it = I->lists.find(minier);
	if (it == I->lists.end()) { *n = 0; return nullptr; }
	*n = (int)it->second.size();
	return it->second.data();
}

int main(int argc, char **argv)
{
	const int n_case = argc > 1 ? atoi(argv[1]) : 2000;
	std::mt19937_64 rng(20240917);
	long n_tied_cases = 0;
	for (int it = 0; it < n_case; ++it) {
		Idx I;
		const int n_kmer = 1 + (int)(rng() % 12), qlen = 150 + (int)(rng() % 200);
		std::vector<uint64_t> hashes;
		for (int h = 0; h < n_kmer; ++h) {
			const uint64_t hash = (rng() >> 20) + 1;
			std::vector<uint64_t> &l = I.lists[hash];
			const int n = 1 + (int)(rng() % (it % 3 == 0 ? 40 : 4));
			std::map<uint64_t, int> seen;
			for (int k = 0; k < n; ++k) seen[(uint64_t)(rng() % 3) << 32 | (uint64_t)(rng() % 5000) << 1 | (rng() & 1)] = 1;
			for (auto &kv : seen) l.push_back(kv.first); // ascending, as in the index (index.c:265)
			hashes.push_back(hash);
		}
		// the read's minimizers: ascending positions, k-mers drawn with repetition, either strand
		std::vector<ora128_t> mv;
		int pos = 20;
		const int n_mz = 1 + (int)(rng() % 25);
		for (int m = 0; m < n_mz && pos < qlen; ++m, pos += 1 + (int)(rng() % 12)) {
			ora128_t z;
			z.x = hashes[rng() % hashes.size()] << 8 | 21, z.y = (uint64_t)pos << 1 | (rng() & 1);
			mv.push_back(z);
		}
		const int64_t flag = 0x400000LL | (it % 7 == 0 ? 0x100000LL : 0) | (it % 11 == 0 ? 0x200000LL : 0); // HEAP_SORT, sometimes one strand only
		if ((flag & 0x300000LL) == 0x300000LL) continue;
		// oracle
		std::vector<ora128_t> mv2 = mv;
		ora128_t *a = nullptr; uint64_t *mp = nullptr; int64_t n_a = 0; int n_mp = 0, rep_len = 0;
		ora_collect_seed_hits(&I, idx_get, flag, qlen, 1000000, 1000000, 0, 0.0f, mv2.data(), (int64_t)mv2.size(), &a, &n_a, &mp, &n_mp, &rep_len);
		// the header, driven the way the kernel drives it
		struct Seed { const uint64_t *cr; uint32_t n, qp, span; bool tandem; };
		std::vector<Seed> sd;
		for (size_t i = 0; i < mv.size(); ++i) {
			int n; const uint64_t *cr = idx_get(&I, mv[i].x >> 8, &n);
			Seed s{cr, (uint32_t)n, (uint32_t)mv[i].y, (uint32_t)(mv[i].x & 0xff), false};
			if (i > 0 && mv[i].x >> 8 == mv[i - 1].x >> 8) s.tandem = true;
			if (i + 1 < mv.size() && mv[i].x >> 8 == mv[i + 1].x >> 8) s.tandem = true;
			sd.push_back(s);
		}
		std::vector<uint64_t> hx(sd.size() + 1), hy(sd.size() + 1);
		std::vector<ora128_t> out((size_t)n_a + 1);
		uint32_t n_for = 0, n_rev = 0;
		const uint32_t n = (uint32_t)n_a;
		bool overflow = false;
		heap_merge_order((uint32_t)sd.size(), hx.data(), hy.data(),
			[&](uint32_t i, uint32_t *cnt) { *cnt = sd[i].n; return sd[i].cr; },
			[&](uint32_t i, uint64_t rr) {
				const uint32_t qp = sd[i].qp, span = sd[i].span, rpos = (uint32_t)rr >> 1;
				const bool fwd = (rr & 1) == (qp & 1);
				if (fwd ? (flag & 0x200000LL) != 0 : (flag & 0x100000LL) != 0) return;
				ora128_t p;
				if (fwd) p.x = (rr & 0xffffffff00000000ULL) | rpos, p.y = (uint64_t)span << 32 | (uint64_t)(qp >> 1);
				else p.x = 1ULL << 63 | (rr & 0xffffffff00000000ULL) | rpos, p.y = (uint64_t)span << 32 | (uint64_t)(uint32_t)(qlen - ((int)(qp >> 1) + 1 - (int)span) - 1);
				if (sd[i].tandem) p.y |= 1ULL << 42;
				if (n_for + n_rev >= n) { overflow = true; return; }
				if (p.x >> 63) out[n - (++n_rev)] = p; else out[n_for++] = p;
			});
		for (uint32_t j = 0; j < n_rev >> 1; ++j) std::swap(out[n - 1 - j], out[n - n_rev + j]);
		bool tie = false;
		for (int64_t j = 1; j < n_a; ++j) tie |= a[j].x == a[j - 1].x;
		n_tied_cases += tie;
		if (overflow || n_for + n_rev != n || memcmp(out.data(), a, (size_t)n_a * sizeof(ora128_t)) != 0) {
			fprintf(stderr, "case %d: header and oracle disagree (n_a=%ld, header %u+%u)\n", it, (long)n_a, n_for, n_rev);
			return 1;
		}
		free(a); free(mp);
	}
	printf("OK %d %ld\n", n_case, n_tied_cases);
	return 0;
}

// TEST INFRASTRUCTURE.  Host build of the header the sketch kernels are made of (minimap2_amd/csrc/sketch_dev.hpp: the (w,k)-minimizer
// automaton started in the middle of a sequence), checked against the oracle's restatement of mm_sketch (oracle/sketch.c, itself
// pinned to the reference): sequences with N runs, homopolymers, low-complexity stretches and strand-symmetric k-mers are cut the
// way sketch_wave_kernel cuts a read (64 stretches of at least 32 bases) and the way idx_sketch_kernel cuts a contig (fixed chunks);
// the concatenation of the stretches' outputs must be the reference's list, every stretch must report owned positions only, each
// at most once -- which is what lets a lane stage its minimizers in the slots of its own stretch (seed_chain.hip).
// Prints "OK <cases> <minimizers>" or fails.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../minimap2_amd/csrc/sketch_dev.hpp"
#include "../../oracle/oracle.h"

using namespace mm2amd;

static const char ACGT[] = "ACGT";

static std::string make_seq(std::mt19937_64 &rng, int len)
{
	std::string s;
	s.reserve(len);
	while ((int)s.size() < len) {
		const int kind = (int)(rng() % 10);
		int n = 1 + (int)(rng() % 200);
		if (kind == 0) s.append(std::min(n, 40), 'N');                                        // ambiguous run
		else if (kind == 1) s.append(n, ACGT[rng() & 3]);                                       // homopolymer
		else if (kind == 2) { const char a = ACGT[rng() & 3], b = ACGT[rng() & 3]; for (int i = 0; i < n; ++i) s.push_back(i & 1 ? a : b); } // dinucleotide repeat
		else if (kind == 3) { std::string u; for (int i = 0; i < 7; ++i) u.push_back(ACGT[rng() & 3]); for (int i = 0; i < n; ++i) s.push_back(u[i % 7]); }
		else if (kind == 4) { // a palindrome in the reverse-complement sense: strand-symmetric k-mers
			std::string h;
			for (int i = 0; i < 20; ++i) h.push_back(ACGT[rng() & 3]);
			s += h;
			for (int i = 19; i >= 0; --i) s.push_back(h[i] == 'A' ? 'T' : h[i] == 'C' ? 'G' : h[i] == 'G' ? 'C' : 'A');
		} else for (int i = 0; i < n; ++i) s.push_back(ACGT[rng() & 3]);
	}
	s.resize(len);
	return s;
}

template <bool HPC>
static long check(const std::string &s, int w, int k, int64_t chunk, uint32_t rid)
{
	const int len = (int)s.size();
	std::vector<uint8_t> store(len + 64 + 16, 4);
	uint8_t *nt4 = store.data() + 8; // the kernels' buffers are padded on both sides (aligned 8-byte fetches)
	nt4 = (uint8_t *)(((uintptr_t)nt4 + 7) & ~(uintptr_t)7);
	for (int i = 0; i < len; ++i) nt4[i] = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : s[i] == 'T' ? 3 : 4;
	std::vector<ora128_t> want(len + 1);
	const int64_t n_want = ora_sketch(s.c_str(), len, w, k, rid, HPC ? 1 : 0, want.data(), len + 1);
	std::vector<uint64_t> bx(256), by(256), gx, gy;
	for (int64_t cs = 0; cs < len; cs += chunk) {
		const int64_t ce = cs + chunk < len ? cs + chunk : len;
		const size_t before = gx.size();
		sketch_chunk<HPC>(nt4, len, cs, ce, w, k, rid, bx.data(), by.data(), 1, [&](uint64_t x, uint64_t y) { gx.push_back(x), gy.push_back(y); });
		if ((int64_t)(gx.size() - before) > ce - cs) { fprintf(stderr, "stretch [%ld,%ld) reported %zu minimizers\n", (long)cs, (long)ce, gx.size() - before); exit(1); }
		for (size_t i = before; i < gx.size(); ++i) {
			const int64_t pos = (int64_t)((uint32_t)gy[i] >> 1);
			if (pos < cs || pos >= ce) { fprintf(stderr, "stretch [%ld,%ld) reported position %ld\n", (long)cs, (long)ce, (long)pos); exit(1); }
			for (size_t j = before; j < i; ++j) if (gy[j] == gy[i]) { fprintf(stderr, "position %ld reported twice\n", (long)pos); exit(1); }
		}
	}
	if ((int64_t)gx.size() != n_want) { fprintf(stderr, "w=%d k=%d hpc=%d len=%d chunk=%ld: %zu minimizers, reference %ld\n", w, k, (int)HPC, len, (long)chunk, gx.size(), (long)n_want); exit(1); }
	for (int64_t i = 0; i < n_want; ++i)
		if (gx[i] != want[i].x || gy[i] != want[i].y) { fprintf(stderr, "w=%d k=%d hpc=%d len=%d chunk=%ld: minimizer %ld differs\n", w, k, (int)HPC, len, (long)chunk, (long)i); exit(1); }
	return (long)n_want;
}

// sketch_wave_kernel's configuration of the same automaton: bases from the 2-bit packed copy (sk_pack16 / sk_base_at), y halves in the ring,
// warm-up of w + k + 8, 32-bit k-mer registers when 2k <= 32 -- and every reported record rebuilt from the packed bases alone
// (sk_minimizer_at), as the kernel's emit phase does from its position marks.
template <bool K32, int W = 0>
static long check_packed(const std::string &s, int w, int k, int64_t chunk)
{
	const int len = (int)s.size();
	std::vector<uint8_t> nt4(len + 32, 4);
	for (int i = 0; i < len; ++i) nt4[i] = s[i] == 'A' ? 0 : s[i] == 'C' ? 1 : s[i] == 'G' ? 2 : s[i] == 'T' ? 3 : 4;
	const int n_words = (len + 15) / 16 + 1;
	std::vector<uint32_t> store(n_words + 2, 0);
	std::vector<uint16_t> amb(n_words, 0);
	uint32_t *pk = store.data() + 2;
	for (int c = 0; c < len; c += 16) {
		uint32_t q[4], packed, flags;
		memcpy(q, nt4.data() + c, 16);
		sk_pack16(q, &packed, &flags);
		pk[c >> 4] = packed, amb[c >> 4] = (uint16_t)flags;
	}
	for (int i = 0; i < len; ++i) if (sk_base_at(pk, amb.data(), i) != nt4[i]) { fprintf(stderr, "packed base %d differs\n", i); exit(1); }
	std::vector<ora128_t> want(len + 1);
	const int64_t n_want = ora_sketch(s.c_str(), len, w, k, 0, 0, want.data(), len + 1);
	std::vector<uint64_t> bx(256), gx, gy;
	std::vector<uint32_t> by(256);
	auto base_at = [&](int64_t i) -> int { return sk_base_at(pk, amb.data(), i); };
	for (int64_t cs = 0; cs < len; cs += chunk) {
		const int64_t ce = cs + chunk < len ? cs + chunk : len;
		sketch_chunk_core<false, K32, uint32_t, W>(base_at, len, cs, ce, w, k, 0u, bx.data(), by.data(), 1, [&](uint64_t x, uint64_t y) { gx.push_back(x), gy.push_back(y); }, (int64_t)(w + k + 8));
	}
	if ((int64_t)gx.size() != n_want) { fprintf(stderr, "packed: w=%d k=%d len=%d chunk=%ld: %zu minimizers, reference %ld\n", w, k, len, (long)chunk, gx.size(), (long)n_want); exit(1); }
	for (int64_t i = 0; i < n_want; ++i) {
		if (gx[i] != want[i].x || gy[i] != want[i].y) { fprintf(stderr, "packed: w=%d k=%d len=%d chunk=%ld: minimizer %ld differs\n", w, k, len, (long)chunk, (long)i); exit(1); }
		if (i > 0 && gy[i] <= gy[i - 1]) { fprintf(stderr, "positions not increasing at %ld\n", (long)i); exit(1); } // what lets a bit mask carry the list
		uint64_t x, y;
		const int64_t pos = (int64_t)((uint32_t)gy[i] >> 1);
		sk_minimizer_at(pk, pos, pos, k, &x, &y);
		if (x != gx[i] || y != gy[i]) { fprintf(stderr, "packed: w=%d k=%d len=%d: record of position %ld rebuilt as %llx/%llx, automaton %llx/%llx\n", w, k, len, (long)pos, (unsigned long long)x, (unsigned long long)y, (unsigned long long)gx[i], (unsigned long long)gy[i]); exit(1); }
	}
	return (long)n_want;
}

int main(int argc, char **argv)
{
	const int n_case = argc > 1 ? atoi(argv[1]) : 300;
	std::mt19937_64 rng(20260922);
	static const int WK[][2] = { { 10, 15 }, { 19, 19 }, { 11, 21 }, { 5, 15 }, { 1, 11 }, { 32, 28 } };
	static const int WK2[][2] = { { 10, 15 }, { 19, 19 }, { 10, 16 }, { 5, 14 }, { 4, 8 }, { 32, 28 }, { 11, 21 }, { 3, 17 } }; // even k: strand-symmetric k-mers exist
	long total = 0;
	for (int c = 0; c < n_case; ++c) {
		const int len = c % 7 == 0 ? 1 + (int)(rng() % 80) : 200 + (int)(rng() % 6000);
		const std::string s = make_seq(rng, len);
		const int w = WK[c % 6][0], k = WK[c % 6][1];
		int64_t chunk = (len + 63) / 64; // sketch_wave_kernel: one stretch per lane, at least 32 bases
		if (chunk < 32) chunk = 32;
		total += check<false>(s, w, k, chunk, 0u);
		total += check<false>(s, w, k, 1 + (int64_t)(rng() % 300), (uint32_t)(c & 3)); // arbitrary cuts, as the index build makes them
		if (c % 3 == 0) total += check<true>(s, w, k, 64 + (int64_t)(rng() % 500), (uint32_t)(c & 3));
		const int w2 = WK2[c % 8][0], k2 = WK2[c % 8][1];
		const int64_t chunk2 = c & 1 ? chunk : 32 + (int64_t)(rng() % 200);
		total += 2 * k2 <= 32 ? check_packed<true>(s, w2, k2, chunk2) : check_packed<false>(s, w2, k2, chunk2);
		if (2 * k2 <= 32) total += check_packed<false>(s, w2, k2, chunk2); // the wide registers on a narrow k
		// the instantiations with the window size compiled in (unrolled ring scans)
		if (w2 == 10 && k2 == 15) total += check_packed<true, 10>(s, 10, 15, chunk2);
		if (w2 == 10 && k2 == 16) total += check_packed<true, 10>(s, 10, 16, chunk2);
		if (w2 == 19) total += check_packed<false, 19>(s, 19, k2, chunk2);
		if (w2 == 5) total += check_packed<true, 5>(s, 5, k2, chunk2);
		if (c % 8 == 7) total += check_packed<false, 10>(s, 10, 20, chunk2);
	}
	printf("OK %d %ld\n", n_case, total);
	return 0;
}

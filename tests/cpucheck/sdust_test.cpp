// TEST INFRASTRUCTURE.  Host build of minimap2_amd/csrc/sdust_core.hpp (what dust_filter_kernel runs per read) against the
// reference's own sdust() (sdust.c, linked from oracle/_ref/libminimap2_ref.a): random sequences, low-complexity sequences of every
// kind (homopolymers, short tandem repeats with mutations, mixtures), Ns, several thresholds.  Prints "OK <cases> <cases with
// masked regions> <largest number of perfect intervals seen>" or fails.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../minimap2_amd/csrc/sdust_core.hpp"

extern "C" uint64_t *sdust(void *km, const uint8_t *seq, int l_seq, int T, int W, int *n);

int main(int argc, char **argv)
{
	const int n_case = argc > 1 ? atoi(argv[1]) : 3000;
	std::mt19937_64 rng(7);
	long n_masked = 0;
	int max_np = 0;
	static const char acgt[] = "ACGT";
	for (int it = 0; it < n_case; ++it) {
		const int len = (int)(rng() % (it % 10 == 0 ? 20000 : 600)) + 1;
		std::string s(len, 'A');
		int pos = 0;
		while (pos < len) { // pieces: random / homopolymer / tandem repeat with a few mutations / N run
			const int kind = (int)(rng() % 5), piece = 1 + (int)(rng() % 300);
			std::string unit;
			for (int k = 0, ul = 1 + (int)(rng() % 7); k < ul; ++k) unit += acgt[rng() % 4];
			for (int k = 0; k < piece && pos < len; ++k, ++pos) {
				if (kind == 0) s[pos] = acgt[rng() % 4];
				else if (kind == 1) s[pos] = unit[0];
				else if (kind == 2 || kind == 3) s[pos] = (rng() % 40 == 0) ? acgt[rng() % 4] : unit[k % unit.size()];
				else s[pos] = (piece < 6 || k < 3) ? 'N' : acgt[rng() % 4];
			}
		}
		if (it % 7 == 0) for (char &c : s) if (rng() % 3 == 0) c = (char)(c + 32); // lower case
		const int T = (int[]){20, 20, 10, 30, 5, 100}[it % 6];
		int n_ref = 0;
		uint64_t *ref = sdust(0, (const uint8_t *)s.data(), len, T, 64, &n_ref);
		std::vector<uint8_t> nt4(len);
		for (int i = 0; i < len; ++i) { const char c = s[i] & ~32; nt4[i] = c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }
		static mm2amd::SdustState S; static mm2amd::SdustState::Perf Pbuf[mm2amd::SdustState::PCAP]; S.P = Pbuf;
		std::vector<uint64_t> got;
		mm2amd::sdust_scan(nt4.data(), len, T, S, [&](int st, int en) { got.push_back((uint64_t)st << 32 | (uint32_t)en); });
		if (S.max_nP > max_np) max_np = S.max_nP;
		n_masked += n_ref > 0;
		if (S.overflow || (int)got.size() != n_ref || (n_ref && memcmp(got.data(), ref, (size_t)n_ref * 8) != 0)) {
			fprintf(stderr, "case %d (len %d, T %d): %d regions vs %zu, overflow %d\n", it, len, T, n_ref, got.size(), S.overflow);
			return 1;
		}
		free(ref);
	}
	printf("OK %d %ld %d\n", n_case, n_masked, max_np);
	return 0;
}

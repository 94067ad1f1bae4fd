// SDUST low-complexity masking of a read and the minimizer filter built on it (mm_mapopt_t::sdust_thres, `-T`):
// sdust_core (sdust.c:137-170, with shift_window :70-90, save_masked_regions :92-108, find_perfect :110-135) and
// mm_dust_minier (map.c:34-57).
//
// One thread runs the scan for one read: the state is a 64-entry window of 3-mers, two 64-entry count tables and a list of
// perfect intervals (caller-provided storage).  Shared by the device kernel (seed_chain.hip: dust_filter_kernel), the oracle-backed check backend and a host
// unit test against the reference's sdust() (tests/cpucheck/sdust_test.cpp).
#pragma once
#include <cstdint>
#include "backend.hpp"

namespace mm2amd {

struct SdustState {
	static constexpr int W = 64;          // window size minimap2 uses (map.c:41)
	int32_t w[64], front, count;          // the window: a FIFO of 3-mer codes (kdq_t(int)); at most W - 2 entries
	int32_t cv[64], cw[64];
	// perfect intervals of the window, descending start then ascending finish.  In a homopolymer every (start, finish) pair of the
	// window is one, so the list holds up to W*W entries (an entry lives for at most W steps, a step adds at most W) (the reference grows a vector): the caller supplies PCAP of them.
	static constexpr int PCAP = 4096;
	struct Perf { int32_t start, finish, r, l; };
	Perf *P;
	int32_t nP, max_nP, overflow;
	int32_t rv, rw, L;
	int32_t last_s, last_f;               // the last masked region (it may still grow); last_f < 0: none yet
};

// intervals are reported through emit(start, finish) in ascending order, each one final
template <class Emit>
MM2_HD inline void sdust_save_masked(SdustState &S, int start, Emit &emit)
{
	if (S.nP == 0 || S.P[S.nP - 1].start >= start) return;
	const SdustState::Perf &p = S.P[S.nP - 1];
	bool saved = false;
	if (S.last_f >= 0 && p.start <= S.last_f) { // overlapping with or adjacent to the previous region
		saved = true;
		if (p.finish > S.last_f) S.last_f = p.finish;
	}
	if (!saved) {
		if (S.last_f >= 0) emit(S.last_s, S.last_f);
		S.last_s = p.start, S.last_f = p.finish;
	}
	int i;
	for (i = S.nP - 1; i >= 0 && S.P[i].start < start; --i) {} // perfect intervals that have fallen out of the window
	S.nP = i + 1;
}

MM2_HD inline void sdust_shift_window(SdustState &S, int t, int T)
{
	if (S.count >= SdustState::W - 3 + 1) {
		const int s = S.w[S.front];
		S.front = (S.front + 1) & 63, --S.count;
		S.rw -= --S.cw[s];
		if (S.L > S.count) --S.L, S.rv -= --S.cv[s];
	}
	S.w[(S.front + S.count) & 63] = t, ++S.count;
	++S.L;
	S.rw += S.cw[t]++;
	S.rv += S.cv[t]++;
	if (S.cv[t] * 10 > T << 1) {
		int s;
		do {
			s = S.w[(S.front + S.count - S.L) & 63];
			S.rv -= --S.cv[s];
			--S.L;
		} while (s != t);
	}
}

MM2_HD inline void sdust_find_perfect(SdustState &S, int T, int start)
{
	int32_t c[64];
	int r = S.rv, max_r = 0, max_l = 0;
	for (int k = 0; k < 64; ++k) c[k] = S.cv[k];
	for (int i = S.count - S.L - 1; i >= 0; --i) {
		const int t = S.w[(S.front + i) & 63];
		r += c[t]++;
		const int new_r = r, new_l = S.count - i - 1;
		if (new_r * 10 > T * new_l) {
			int j;
			for (j = 0; j < S.nP && S.P[j].start >= i + start; ++j) { // insertion position
				const SdustState::Perf &p = S.P[j];
				if (max_r == 0 || p.r * max_l > max_r * p.l) max_r = p.r, max_l = p.l;
			}
			if (max_r == 0 || new_r * max_l >= max_r * new_l) {
				max_r = new_r, max_l = new_l;
				if (S.nP >= SdustState::PCAP) { S.overflow = 1; continue; }
				for (int k = S.nP; k > j; --k) S.P[k] = S.P[k - 1];
				++S.nP;
				if (S.nP > S.max_nP) S.max_nP = S.nP;
				S.P[j].start = i + start, S.P[j].finish = S.count + (3 - 1) + start;
				S.P[j].r = new_r, S.P[j].l = new_l;
			}
		}
	}
}

// seq: nt4 codes (0-3, anything else breaks the sequence like an N), T: score threshold
template <class Emit>
MM2_HD inline void sdust_scan(const uint8_t *seq, int l_seq, int T, SdustState &S, Emit emit)
{
	S.front = S.count = 0, S.nP = 0, S.max_nP = 0, S.overflow = 0, S.rv = S.rw = S.L = 0, S.last_s = 0, S.last_f = -1;
	for (int k = 0; k < 64; ++k) S.cv[k] = S.cw[k] = 0;
	int l = 0, start;
	unsigned t = 0;
	for (int i = 0; i <= l_seq; ++i) {
		const int b = i < l_seq ? seq[i] : 4;
		if (b < 4) {
			++l, t = (t << 2 | (unsigned)b) & 63u;
			if (l >= 3) { // a complete word
				start = (l - SdustState::W > 0 ? l - SdustState::W : 0) + (i + 1 - l);
				sdust_save_masked(S, start, emit);
				sdust_shift_window(S, (int)t, T);
				if (S.rw * 10 > S.L * T) sdust_find_perfect(S, T, start);
			}
		} else { // an N or the end: flush (the window itself is not cleared, as in the reference)
			start = (l - SdustState::W + 1 > 0 ? l - SdustState::W + 1 : 0) + (i + 1 - l);
			while (S.nP) sdust_save_masked(S, start++, emit);
			l = 0, t = 0;
		}
	}
	if (S.last_f >= 0) emit(S.last_s, S.last_f);
}

// mm_dust_minier (map.c:34-57): keep a minimizer unless more than half of it lies in masked regions.  reg(u, &st, &en) returns
// region u; x / y are the minimizer arrays (x = hash << 8 | span, y = ... | pos << 1 | strand), compacted in place.
// pos_off is added to the minimizer positions: the reference filters the second read of a pair after it has shifted its
// positions by the first read's length, against regions in the read's own coordinates (map.c:66-69).
template <class RegionAt>
MM2_HD inline int dust_filter_minimizers(int n, uint64_t *x, uint64_t *y, int n_reg, RegionAt reg, int pos_off)
{
	int u = 0, k = 0;
	for (int j = 0; j < n; ++j) {
		const int32_t qpos = (int32_t)((uint32_t)y[j] >> 1) + pos_off, span = (int32_t)(x[j] & 0xff);
		const int32_t s = qpos - (span - 1), e = s + span;
		int32_t st, en;
		while (u < n_reg) { reg(u, &st, &en); if (en <= s) ++u; else break; }
		bool keep = true;
		if (u < n_reg) {
			reg(u, &st, &en);
			if (st < e) {
				int l = 0;
				for (int v = u; v < n_reg; ++v) {
					reg(v, &st, &en);
					if (!(st < e)) break;
					const int ss = s > st ? s : st, ee = e < en ? e : en;
					l += ee - ss;
				}
				keep = l <= span >> 1;
			}
		}
		if (keep) { x[k] = x[j], y[k] = y[j]; ++k; }
	}
	return k;
}

} // namespace mm2amd

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include "flat_index.hpp"

namespace mm2amd {

void FlatIndex::build_tables(const std::vector<std::pair<uint64_t, uint64_t>> &sp)
{
	keys.clear(); val_off.clear(); pos.clear();
	pos.reserve(sp.size());
	for (size_t i = 0; i < sp.size(); ++i) {
		if (i == 0 || sp[i].first != sp[i - 1].first) { keys.push_back(sp[i].first); val_off.push_back((uint32_t)pos.size()); }
		pos.push_back(sp[i].second);
	}
	val_off.push_back((uint32_t)pos.size());
	// direct table on the top bits of the 2k-bit hash; about one key per bucket, capped at 2^28 buckets
	int hash_bits = 2 * k, want = 1;
	while ((1ull << want) < keys.size() && want < 28) ++want;
	bucket_bits = std::min(hash_bits, std::max(8, want));
	key_shift = hash_bits - bucket_bits;
	bucket_start.assign((1ull << bucket_bits) + 1, 0);
	for (uint64_t h : keys) ++bucket_start[(h >> key_shift) + 1];
	for (size_t i = 1; i < bucket_start.size(); ++i) bucket_start[i] += bucket_start[i - 1];
}

void FlatIndex::from_reference(const ref::Idx *mi, bool tables)
{
	if (!mi || !mi->B) throw std::invalid_argument("[mm2amd] null reference index");
	k = mi->k, w = mi->w, flag = mi->flag, n_seq = mi->n_seq, n_alt = mi->n_alt;
	names.resize(n_seq); seq_off.resize(n_seq); seq_len.resize(n_seq); is_alt.assign(n_seq, 0);
	sum_len = 0;
	for (uint32_t i = 0; i < n_seq; ++i) {
		names[i] = mi->seq[i].name ? mi->seq[i].name : "";
		is_alt[i] = mi->seq[i].is_alt ? 1 : 0;
		seq_off[i] = mi->seq[i].offset, seq_len[i] = mi->seq[i].len;
		sum_len += mi->seq[i].len;
	}
	S = mi->S; // borrowed: the reference index outlives the mapper (map.c:663-686)
	junc.clear(), has_junc = false;
	if (mi->I) { // junction annotation (--junc-bed): copied, the mapper makes the per-window junc[] of mm_idx_bed_junc from it
		const ref::IntvList *I = (const ref::IntvList *)mi->I;
		junc.resize(n_seq);
		for (uint32_t i = 0; i < n_seq; ++i)
			for (int32_t j = 0; j < I[i].n; ++j) junc[i].push_back(Junc{I[i].a[j].st, I[i].a[j].en, I[i].a[j].strand});
		has_junc = true;
	}
	jump.clear(), has_jump = false;
	if (mi->J) { // jump annotation (-j / --pass1): used by the host's mm_jump_split step only
		const ref::JJumpList *J = (const ref::JJumpList *)mi->J;
		jump.resize(n_seq);
		for (uint32_t i = 0; i < n_seq; ++i) jump[i].assign(J[i].a, J[i].a + J[i].n);
		has_jump = true;
	}
	spsc.clear(), has_spsc = false;
	if (mi->spsc) { // splice scores (--spsc): what mm_idx_spsc_get reads
		const ref::SpscList *P = (const ref::SpscList *)mi->spsc;
		spsc.resize((size_t)n_seq * 2);
		for (uint32_t i = 0; i < n_seq * 2; ++i) if (P[i].n) spsc[i].assign(P[i].a, P[i].a + P[i].n);
		has_spsc = true;
	}
	if (!tables) { keys.clear(), val_off.assign(1, 0), pos.clear(), bucket_start.clear(); return; }
	std::vector<std::pair<uint64_t, uint64_t>> pairs;
	const uint32_t nb = 1u << mi->b;
	for (uint32_t b = 0; b < nb; ++b) {
		const ref::Bucket &B = mi->B[b];
		const ref::KhashIdx *h = B.h;
		if (!h) continue;
		for (uint32_t s = 0; s < h->n_buckets; ++s) {
			if ((h->flags[s >> 4] >> ((s & 0xfU) << 1)) & 3) continue; // empty or deleted slot
			const uint64_t key = h->keys[s], val = h->vals[s];
			const uint64_t hash = (key >> 1) << mi->b | b;
			if (key & 1) pairs.emplace_back(hash, val);
			else {
				const uint64_t *p = &B.p[val >> 32];
				for (uint32_t j = 0, n = (uint32_t)val; j < n; ++j) pairs.emplace_back(hash, p[j]);
			}
		}
	}
	std::sort(pairs.begin(), pairs.end());
	build_tables(pairs);
}

void FlatIndex::from_sequences(int k_, int w_, int flag_, int n, const char *const *seqs, const char *const *nms,
                               void (*sketch)(const char *, int, int, int, uint32_t, int, std::vector<ref::mm128> &))
{
	k = k_, w = w_, flag = flag_, n_seq = (uint32_t)n, n_alt = 0;
	names.resize(n); seq_off.resize(n); seq_len.resize(n);
	sum_len = 0;
	for (int i = 0; i < n; ++i) {
		names[i] = nms && nms[i] ? nms[i] : "";
		seq_off[i] = sum_len, seq_len[i] = (uint32_t)strlen(seqs[i]);
		sum_len += seq_len[i];
	}
	S_own.assign((sum_len + 7) / 8, 0);
	std::vector<std::pair<uint64_t, uint64_t>> pairs;
	std::vector<ref::mm128> mz;
	extern const uint8_t kNt4Table[256];
	for (int i = 0; i < n; ++i) {
		for (uint32_t j = 0; j < seq_len[i]; ++j) {
			const uint64_t o = seq_off[i] + j;
			S_own[o >> 3] |= (uint32_t)kNt4Table[(uint8_t)seqs[i][j]] << ((o & 7) << 2);
		}
		mz.clear();
		if (seq_len[i] > 0) sketch(seqs[i], (int)seq_len[i], w, k, (uint32_t)i, flag & ref::I_HPC, mz);
		for (const ref::mm128 &m : mz) pairs.emplace_back(m.x >> 8, m.y);
	}
	S = S_own.data();
	std::sort(pairs.begin(), pairs.end());
	build_tables(pairs);
}

int32_t FlatIndex::cal_max_occ(float f) const
{
	const size_t n = keys.size();
	if (f <= 0.f || n == 0) return INT32_MAX;
	std::vector<uint32_t> a(n);
	for (size_t i = 0; i < n; ++i) a[i] = val_off[i + 1] - val_off[i];
	const size_t kk = (uint32_t)((1. - f) * n);
	std::nth_element(a.begin(), a.begin() + kk, a.end());
	return (int32_t)(a[kk] + 1);
}

} // namespace mm2amd

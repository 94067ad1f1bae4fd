// Flat, position-independent minimizer index: the device-friendly mirror of the reference's bucketed khash
// (mm_idx_t::B, index.c:28-33,93-110).  Any layout is allowed as long as a lookup returns the same
// (count, ascending position list) as mm_idx_get; we use a two-level direct table:
//     bucket_start[hash >> key_shift]  ->  run of distinct keys (sorted)  ->  val_off[]  ->  pos[]
// so a lookup costs one dependent load per level and touches 1-2 sectors per level.
#pragma once
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include "abi_ref.hpp"

namespace mm2amd {

struct FlatIndex {
	int k = 0, w = 0, flag = 0;
	uint32_t n_seq = 0;
	int n_alt = 0;
	std::vector<uint8_t> is_alt;        // per sequence: ALT contig (mm_idx_seq_t::is_alt, set by mm_idx_alt_read, index.c:648-670)
	// annotated introns per sequence, ascending start (mm_idx_t::I as mm_idx_bed_read leaves it, index.c:797-801): st, en, strand
	struct Junc { int32_t st, en, strand; };
	std::vector<std::vector<Junc>> junc;
	bool has_junc = false;
	// junctions a clipped alignment end may jump across (mm_idx_t::J as mm_idx_jjump_read leaves it: ascending off), per sequence
	std::vector<std::vector<ref::JJump1>> jump;
	bool has_jump = false;
	// splice scores (--spsc): per sequence and strand (index rid << 1 | minus), ascending  pos << 8 | (score + 64) << 1 | is_acceptor
	std::vector<std::vector<uint64_t>> spsc;
	bool has_spsc = false;
	std::vector<std::string> names;
	std::vector<uint64_t> seq_off;      // offset of each sequence in S (bases)
	std::vector<uint32_t> seq_len;
	const uint32_t *S = nullptr;        // 4-bit packed bases, 8 per word (mmpriv.h:34-35); borrowed or owned
	std::vector<uint32_t> S_own;
	uint64_t sum_len = 0;

	int bucket_bits = 0, key_shift = 0; // bucket id = hash >> key_shift
	std::vector<uint32_t> bucket_start; // (1<<bucket_bits) + 1 entries into keys[]
	std::vector<uint64_t> keys;         // distinct minimizer hashes, ascending
	std::vector<uint32_t> val_off;      // keys.size() + 1 entries into pos[]
	std::vector<uint64_t> pos;          // rid<<32 | last_pos<<1 | strand, ascending within a key (index.c:265)

	// mm_idx_get (index.c:93-110)
	const uint64_t *get(uint64_t hash, int *n) const
	{
		*n = 0;
		if (keys.empty()) return nullptr;
		const uint64_t b = hash >> key_shift;
		if (b >= (1ull << bucket_bits)) return nullptr;
		for (uint32_t i = bucket_start[b], e = bucket_start[b + 1]; i < e; ++i)
			if (keys[i] == hash) { *n = (int)(val_off[i + 1] - val_off[i]); return &pos[val_off[i]]; }
		return nullptr;
	}
	uint8_t base(uint32_t rid, uint32_t p) const // mm_seq4_get via mm_idx_getseq (index.c:164-174)
	{
		const uint64_t o = seq_off[rid] + p;
		return (uint8_t)(S[o >> 3] >> ((o & 7) << 2) & 0xf);
	}
	void getseq(uint32_t rid, uint32_t st, uint32_t en, uint8_t *out) const // one packed word (8 bases) at a time
	{
		uint64_t o = seq_off[rid] + st;
		const uint64_t oe = seq_off[rid] + en;
		for (; o < oe && (o & 7); ++o) *out++ = (uint8_t)(S[o >> 3] >> ((o & 7) << 2) & 0xf);
		for (; o + 8 <= oe; o += 8, out += 8) { // eight nibbles spread to eight bytes (little endian: base 0 is the low nibble and the first byte)
			uint64_t x = S[o >> 3];
			x = (x | x << 16) & 0x0000FFFF0000FFFFull;
			x = (x | x << 8) & 0x00FF00FF00FF00FFull;
			x = (x | x << 4) & 0x0F0F0F0F0F0F0F0Full;
			memcpy(out, &x, 8);
		}
		for (; o < oe; ++o) *out++ = (uint8_t)(S[o >> 3] >> ((o & 7) << 2) & 0xf);
	}
	void getseq2(bool is_rev, uint32_t rid, uint32_t st, uint32_t en, uint8_t *out) const // mm_idx_getseq2 (index.c:176-196): the window [st, en) of the reverse-complemented sequence when is_rev
	{
		if (!is_rev) { getseq(rid, st, en, out); return; }
		const uint32_t len = seq_len[rid];
		getseq(rid, len - en, len - st, out);
		if (en <= st) return;
		uint32_t i = 0, j = en - st - 1;
		for (; i < j; ++i, --j) { const uint8_t a = out[i], b = out[j]; out[i] = b < 4 ? 3 - b : b, out[j] = a < 4 ? 3 - a : a; }
		if (i == j) out[i] = out[i] < 4 ? 3 - out[i] : out[i];
	}
	int32_t cal_max_occ(float f) const; // mm_idx_cal_max_occ (index.c:198-220)

	// (hash, pos) pairs -> tables.  pairs must be sorted by (hash, pos).
	void build_tables(const std::vector<std::pair<uint64_t, uint64_t>> &sorted_pairs);
	// Flatten a reference-built index (read-only view of its private hash buckets).
	// from a reference mm_idx_t: sequence table, packed sequence (borrowed) and annotations always; the minimizer tables only with
	// `tables` (walking the reference's 2^b hash tables and sorting their content: ~40 s on one thread for a 3 Gb index -- the HIP
	// product rebuilds them on the device from the packed sequence instead, DeviceIndexBuilder::build_from_packed)
	void from_reference(const ref::Idx *mi, bool tables = true);
	// Build from raw sequences with our own minimizer code path (host version; used when no reference index exists).
	void from_sequences(int k, int w, int flag, int n, const char *const *seqs, const char *const *names,
	                    void (*sketch)(const char *, int, int, int, uint32_t, int, std::vector<ref::mm128> &));
};

} // namespace mm2amd

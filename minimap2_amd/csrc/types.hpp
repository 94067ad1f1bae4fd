// Shared host-side records of the batched mapper.
#pragma once
#include <cstdint>
#include <cstddef>
#include <string>
#include <vector>
#include "abi_ref.hpp"
#include "exact_rsort.hpp"

namespace mm2amd {

using Anchor = ref::mm128;   // x = rev<<63 | rid<<32 | rpos ; y = flags | seg<<48 | q_span<<32 | qpos   (lchain.c:140-143)

struct KeyX { MM2_HD uint64_t operator()(const Anchor &a) const { return a.x; } };
struct KeyU64 { MM2_HD uint64_t operator()(uint64_t a) const { return a; } };

inline void sort_by_x(Anchor *b, Anchor *e) { RsortScratch sc; exact_radix_sort(b, e, KeyX(), sc); }   // radix_sort_128x
inline void sort_u64(uint64_t *b, uint64_t *e) { RsortScratch sc; exact_radix_sort(b, e, KeyU64(), sc); } // radix_sort_64

// Parameters of the per-read path that come from the index rather than from mm_mapopt_t.
struct IdxParams {
	int k = 15, w = 10, flag = 0;
};

// One fragment handed to the mapper: a read, or the two reads of a pair (seq2/len2 set) already in the orientation they are
// mapped in (worker_for reverse-complements a mate according to pe_ori before mapping, map.c:436-442).
struct ReadView {
	const char *seq = nullptr;   // ASCII
	int len = 0;
	const char *name = nullptr;  // may be null
	const char *seq2 = nullptr;  // second segment of a two-segment fragment
	int len2 = 0;
	bool paired() const { return seq2 != nullptr; }
	int total() const { return len + len2; }
};

// What seeding + chaining produces for one read (the state of mm_map_frag_core after map.c:316).
struct ReadChains {
	// Views of the results; they point either into the owned vectors below (view_own()) or into backend-owned buffers that
	// stay valid until the backend lane is used for the next sub-batch.  The anchors are modified in place by the aligner.
	const uint64_t *u_p = nullptr, *mp_p = nullptr;
	Anchor *a_p = nullptr;
	int32_t n_u = 0, n_mp = 0;
	int64_t n_a = 0;
	void take_ownership() // copy the viewed results into the owned vectors (no-op for what is already owned)
	{
		if (u_p != u.data()) u.assign(u_p, u_p + n_u);
		if (a_p != a.data()) a.assign(a_p, a_p + n_a);
		if (mp_p != mini_pos.data()) mini_pos.assign(mp_p, mp_p + n_mp);
		view_own();
	}
	void view_own() { u_p = u.data(), n_u = (int32_t)u.size(), a_p = a.data(), n_a = (int64_t)a.size(), mp_p = mini_pos.data(), n_mp = (int32_t)mini_pos.size(); }
	std::vector<uint64_t> u;        // per chain: score<<32 | n_anchors
	std::vector<Anchor> a;          // anchors of all chains, chain by chain
	std::vector<uint64_t> mini_pos; // q_span<<32 | q_pos of every minimizer that was looked up and kept (seed.c:124)
	int rep_len = 0;
	bool long_join_done = false;    // the long-join re-chaining question (map.c:283-292) has been settled for this read: asked and answered no, or re-chained by the backend
	bool long_joined = false;       // ... and the chains are the re-chained ones
	// where the backend keeps this read's chains on the device (Backend::align_regions): -1 not there; 0 the first backtrack's arrays, 1 the
	// long-join re-chain's; offsets into the arrays' anchors / chain records
	int8_t dev_src = -1;
	uint64_t dev_a_off = 0, dev_u_off = 0;
	bool chained = true;            // false: a_p / n_a are the read's SORTED anchors and the caller still has to chain them (MM_F_RMQ on a backend without,
	                                // or a read its RMQ kernel handed back)
};

} // namespace mm2amd

// Device-resident index tables and their construction on the GPU (index_build.hip).
#pragma once
#include <cstdint>
#include <vector>
#include "hip_util.hpp"
#include "flat_index.hpp"

namespace mm2amd {

// The flat minimizer tables of flat_index.hpp living in HBM, plus the 4-bit packed reference.
// One probe record per distinct minimizer: the key, where its positions start and how many there are -- the key scan of a bucket and the count
// come out of ONE 64-byte sector (round 5; keys[] and val_off[] stay for the export and the occurrence statistics, the probes of seed_collect_kernel
// read this: bucket_start -> slot instead of bucket_start -> keys -> val_off, val_off + 1)
struct alignas(16) IdxSlot { uint64_t key; uint32_t off, cnt; };
// Round 6: one such record PER BUCKET as well -- the bucket's first key inline, its count's top bit set when the bucket holds more keys, an impossible key when it
// holds none.  A probe reads this one record first: an empty bucket, a bucket whose only key is another one, and a hit on a bucket's first key (three probes in four
// at the index' load factor of 0.65 keys per bucket) are answered by ONE sector read instead of two dependent ones (bucket_start, then the slots).
constexpr uint32_t kIdxMoreKeys = 1u << 31;
constexpr uint64_t kIdxNoKey = ~0ull; // (a minimizer key is hash << 8 | span: never all ones)

struct DeviceIndexTables {
	DevBuf<uint32_t> bucket_start, val_off, S;
	DevBuf<uint64_t> keys, pos;
	DevBuf<IdxSlot> slots, first;          // per distinct minimizer | per bucket (above)
	void make_slots(hipStream_t stream); // from keys / val_off (and bucket_start)
	uint64_t n_keys = 0, n_pos = 0;
	int bucket_bits = 0, key_shift = 0;
	std::vector<unsigned long long> occ_hist; // occ_hist[c] = number of distinct minimizers occurring c times (last bin: >=)
	int32_t cal_max_occ(float f) const;       // mm_idx_cal_max_occ (index.c:198-220) from the histogram
	void upload(const FlatIndex &fi, hipStream_t stream); // mirror host-side tables (index flattened from a reference mm_idx_t)
	void clone_from(const DeviceIndexTables &src, int src_device, int dst_device); // replica on another GPU: device-to-device copies (xGMI); the calling thread's current device must be dst_device
};

struct DeviceIndexBuilder {
	// In-memory index construction, the device counterpart of mm_idx_str (index.c:421-470) / mm_idx_gen (index.c:389-408).
	// Fills the host-side part of `fi` (names, lengths, packed S; no host hash tables) and the device tables `T`.
	static void build(FlatIndex &fi, DeviceIndexTables &T, int k, int w, int flag, int n_seq, const char *const *seqs, const uint64_t *lens,
	                  const char *const *names, hipStream_t stream);
	// Same tables from an index that already carries its packed sequence (fi.S and the sequence table set: FlatIndex::from_reference
	// without the host tables): the device part of mm_gpu_init.
	static void build_from_packed(FlatIndex &fi, DeviceIndexTables &T, hipStream_t stream);
private:
	static void tables_from_nt4(FlatIndex &fi, DeviceIndexTables &T, DevBuf<uint8_t> &d_nt4, hipStream_t stream);
};

} // namespace mm2amd

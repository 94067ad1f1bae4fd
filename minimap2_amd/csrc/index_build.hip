// Device-side index construction: the GPU counterpart of mm_idx_gen's sketch -> bucket -> sort -> hash pipeline
// (index.c:226-278, :369-408), producing the flat tables of flat_index.hpp directly in HBM.
//
//   1. encode   : ASCII -> nt4 (1 B/base) and 4-bit packed S (mmpriv.h:34-35)
//   2. sketch   : mm_sketch (sketch.c:77-143) over fixed-size chunks.  The reference's window automaton is
//                 sequential, but its state after any stretch of w+k consecutive valid slots is a function of that
//                 stretch alone (hard part 7), so a lane that starts early enough before its
//                 chunk reaches the true state before its first owned position; each lane emits only minimizers
//                 whose position lies in its own chunk, which partitions the reference's output exactly.
//   3. sort     : (hash, position) pairs by hash, stably, so that positions stay ascending within a hash (device_sort.hip: a hand-written
//                 device-wide LSD radix sort over the 2k hash bits)
//   4. tables   : distinct keys, value offsets (heads of the runs of equal hashes, ranked by a tile count + prefix sum), top-bits direct table
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>
#include "hip_util.hpp"
#include "index_build.hpp"
#include "device_sort.hpp"
#include "device_sort_dev.hpp"
#include "sketch_dev.hpp"

namespace mm2amd {

extern const uint8_t kNt4Table[256];
__constant__ uint8_t c_nt4_idx[256];

__global__ void __launch_bounds__(256) idx_encode_kernel(const char *ascii, uint8_t *nt4, uint32_t *S, uint64_t total)
{
	// one thread per packed word (8 bases)
	const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t b0 = wi * 8;
	if (b0 >= total) return;
	uint32_t word = 0;
	for (int j = 0; j < 8 && b0 + j < total; ++j) {
		const uint8_t c = c_nt4_idx[(uint8_t)ascii[b0 + j]];
		nt4[b0 + j] = c;
		word |= (uint32_t)c << (j << 2);
	}
	S[wi] = word;
}

// the reverse: 4-bit packed S (an index loaded from a .mmi or built by the reference carries it, index.c:242-246) -> nt4 bytes
__global__ void __launch_bounds__(256) idx_decode_kernel(const uint32_t *S, uint8_t *nt4, uint64_t total)
{
	const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t b0 = wi * 8;
	if (b0 >= total) return;
	const uint32_t word = S[wi];
	for (int j = 0; j < 8 && b0 + j < total; ++j) nt4[b0 + j] = (uint8_t)(word >> (j << 2) & 0xf);
}

struct ChunkDesc { uint32_t rid; uint32_t start; }; // chunk = [start, start + CHUNK) clipped to the sequence

// one lane per chunk (sketch_dev.hpp)
template <bool EMIT, int WMAX, bool HPC>
__global__ void __launch_bounds__(64) idx_sketch_kernel(const uint8_t *nt4, const uint64_t *seq_off, const uint32_t *seq_len, const ChunkDesc *chunks,
                                                         uint64_t n_chunks, int chunk_len, int w, int k, uint32_t *cnt, const uint64_t *out_off,
                                                         uint64_t *out_hash, uint64_t *out_pos)
{
	const uint64_t ci = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (ci >= n_chunks) return;
	const uint32_t rid = chunks[ci].rid;
	const int64_t cs = chunks[ci].start, len = seq_len[rid];
	const int64_t ce = cs + chunk_len < len ? cs + chunk_len : len;
	uint64_t bx[WMAX], by[WMAX];
	uint32_t n_out = 0;
	uint64_t *oh = nullptr, *op = nullptr;
	if (EMIT) oh = out_hash + out_off[ci], op = out_pos + out_off[ci];
	sketch_chunk<HPC>(nt4 + seq_off[rid], len, cs, ce, w, k, rid, bx, by, 1, [&](uint64_t x, uint64_t y) {
		if (EMIT) { oh[n_out] = x >> 8; op[n_out] = y; }
		++n_out;
	});
	if (!EMIT) cnt[ci] = n_out;
}

// A head is the first pair of a run of equal hashes: one per distinct minimizer, in sorted order.  Tiles of kSortTile pairs; wave w of
// a tile owns its pairs [1024 w, 1024 (w+1)) and visits them 64 at a time, so a ballot ranks the heads of a round.
__global__ void __launch_bounds__(kSortThreads) idx_count_heads_kernel(const uint64_t *hash, uint64_t n, uint32_t *tile_cnt)
{
	__shared__ uint32_t sh[4];
	const int tid = threadIdx.x;
	const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
	uint32_t c = 0;
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint64_t i = base + (uint64_t)(r * kSortThreads + tid);
		if (i < n && (i == 0 || hash[i] != hash[i - 1])) ++c;
	}
	uint32_t total;
	(void)block_exclusive_sum(c, sh, total);
	if (tid == 0) tile_cnt[blockIdx.x] = total;
}

// keys[] = the distinct hashes, val_off[] = where each one's run of positions starts; tile_off = exclusive prefix sum of the tiles' head counts
__global__ void __launch_bounds__(kSortThreads) idx_scatter_keys_kernel(const uint64_t *hash, uint64_t n, const uint32_t *tile_off, uint64_t *keys, uint32_t *val_off, uint64_t n_keys)
{
	__shared__ uint32_t sh[4];
	const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
	const uint64_t base = (uint64_t)blockIdx.x * kSortTile + (uint64_t)(w * (kSortItems * 64) + lane);
	uint64_t key[kSortItems];
	uint32_t heads = 0, wave_cnt = 0;
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const uint64_t i = base + (uint64_t)(r * 64);
		key[r] = i < n ? hash[i] : 0;
		const bool head = i < n && (i == 0 || hash[i - 1] != key[r]);
		heads |= (uint32_t)head << r;
		wave_cnt += (uint32_t)__popcll(__ballot(head));
	}
	if (lane == 0) sh[w] = wave_cnt;
	__syncthreads();
	uint32_t run = tile_off[blockIdx.x];
	for (int i = 0; i < w; ++i) run += sh[i];
	const uint64_t below = (1ull << lane) - 1;
#pragma unroll
	for (int r = 0; r < kSortItems; ++r) {
		const bool head = heads >> r & 1;
		const uint64_t bal = __ballot(head);
		if (head) {
			const uint32_t rank = run + (uint32_t)__popcll(bal & below);
			keys[rank] = key[r];
			val_off[rank] = (uint32_t)(base + (uint64_t)(r * 64));
		}
		run += (uint32_t)__popcll(bal);
	}
	if (blockIdx.x == 0 && tid == 0) val_off[n_keys] = (uint32_t)n;
}

__global__ void __launch_bounds__(256) idx_bucket_start_kernel(const uint64_t *keys, uint64_t n_keys, int key_shift, uint64_t n_buckets, uint32_t *bucket_start)
{
	const uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b > n_buckets) return;
	// first key whose bucket id is >= b
	uint64_t lo = 0, hi = n_keys;
	while (lo < hi) {
		const uint64_t mid = (lo + hi) >> 1;
		if ((keys[mid] >> key_shift) < b) lo = mid + 1; else hi = mid;
	}
	bucket_start[b] = (uint32_t)lo;
}

__global__ void __launch_bounds__(256) idx_occ_hist_kernel(const uint32_t *val_off, uint64_t n_keys, unsigned long long *hist, int n_bins)
{
	// almost every minimizer occurs a handful of times: count the low bins per block in LDS, the tail with global atomics
	__shared__ unsigned int low[256];
	low[threadIdx.x] = 0;
	__syncthreads();
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += stride) {
		uint32_t c = val_off[i + 1] - val_off[i];
		if (c < 256) atomicAdd(&low[c], 1u);
		else atomicAdd(&hist[c < (uint32_t)n_bins ? c : (uint32_t)n_bins - 1], 1ull);
	}
	__syncthreads();
	if (low[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)low[threadIdx.x]);
}

void DeviceIndexBuilder::build(FlatIndex &fi, DeviceIndexTables &T, int k, int w, int flag, int n_seq, const char *const *seqs, const uint64_t *lens,
                               const char *const *names, hipStream_t stream)
{
	if (w <= 0 || w >= 256 || k <= 0 || k > 28) throw std::invalid_argument("[mm2amd] index build: need 0<w<256 and 0<k<=28");
	HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_nt4_idx), kNt4Table, 256, 0, hipMemcpyHostToDevice, stream));
	fi.k = k, fi.w = w, fi.flag = flag, fi.n_seq = (uint32_t)n_seq, fi.n_alt = 0;
	fi.names.resize(n_seq), fi.seq_off.resize(n_seq), fi.seq_len.resize(n_seq);
	uint64_t total = 0;
	for (int i = 0; i < n_seq; ++i) {
		if (lens[i] >= (1ull << 31)) throw std::invalid_argument("[mm2amd] reference sequences must be shorter than 2^31 bases");
		fi.names[i] = names && names[i] ? names[i] : "";
		fi.seq_off[i] = total, fi.seq_len[i] = (uint32_t)lens[i];
		total += lens[i];
	}
	fi.sum_len = total;
	const uint64_t n_words = (total + 7) / 8;
	// 1. upload + encode
	DevBuf<char> d_ascii;
	DevBuf<uint8_t> d_nt4;
	d_ascii.ensure(total + 8, 1.0), d_nt4.ensure(total + 8, 1.0);
	T.S.ensure(n_words + 1, 1.0);
	for (int i = 0; i < n_seq; ++i)
		if (lens[i]) HIP_CHECK(hipMemcpyAsync(d_ascii.p + fi.seq_off[i], seqs[i], lens[i], hipMemcpyHostToDevice, stream));
	if (n_words) hipLaunchKernelGGL(idx_encode_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, d_ascii.p, d_nt4.p, T.S.p, total);
	HIP_CHECK(hipGetLastError());
	HIP_CHECK(hipStreamSynchronize(stream));
	d_ascii.release();
	fi.S_own.resize(n_words);
	if (n_words) HIP_CHECK(hipMemcpy(fi.S_own.data(), T.S.p, n_words * 4, hipMemcpyDeviceToHost));
	fi.S = fi.S_own.data();
	tables_from_nt4(fi, T, d_nt4, stream);
}

// The minimizer tables of an index whose sequence is already packed (fi.S; k, w, flag, the sequence table and sum_len set): what
// mm_gpu_init does with a reference mm_idx_t instead of walking its 2^14 hash tables on the host -- a 3 Gb index is rebuilt in
// about a second, and tools/e2e_wall.py checks once at full size that the result equals the reference's own mm_idx_gen.
void DeviceIndexBuilder::build_from_packed(FlatIndex &fi, DeviceIndexTables &T, hipStream_t stream)
{
	if (fi.w <= 0 || fi.w >= 256 || fi.k <= 0 || fi.k > 28) throw std::invalid_argument("[mm2amd] index build: need 0<w<256 and 0<k<=28");
	if (!fi.S) throw std::invalid_argument("[mm2amd] index without sequence");
	const uint64_t total = fi.sum_len, n_words = (total + 7) / 8;
	DevBuf<uint8_t> d_nt4;
	d_nt4.ensure(total + 8, 1.0);
	T.S.ensure(n_words + 1, 1.0);
	if (n_words) HIP_CHECK(hipMemcpyAsync(T.S.p, fi.S, n_words * 4, hipMemcpyHostToDevice, stream));
	if (n_words) hipLaunchKernelGGL(idx_decode_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, stream, T.S.p, d_nt4.p, total);
	HIP_CHECK(hipGetLastError());
	tables_from_nt4(fi, T, d_nt4, stream);
}

// steps 2-4: sketch -> sort -> tables, from the nt4 bytes of the concatenated sequences
void DeviceIndexBuilder::tables_from_nt4(FlatIndex &fi, DeviceIndexTables &T, DevBuf<uint8_t> &d_nt4, hipStream_t stream)
{
	const int k = fi.k, w = fi.w, flag = fi.flag, n_seq = (int)fi.n_seq;
	// 2. chunked sketch: count, scan, emit
	const int chunk_len = 2048;
	std::vector<ChunkDesc> chunks;
	for (int i = 0; i < n_seq; ++i)
		for (uint64_t s = 0; s < fi.seq_len[i]; s += chunk_len) chunks.push_back(ChunkDesc{(uint32_t)i, (uint32_t)s});
	const uint64_t n_chunks = chunks.size();
	DevBuf<ChunkDesc> d_chunks;
	DevBuf<uint64_t> d_seq_off, d_out_off;
	DevBuf<uint32_t> d_seq_len, d_cnt;
	d_chunks.ensure(n_chunks + 1, 1.0), d_seq_off.ensure(n_seq + 1, 1.0), d_seq_len.ensure(n_seq + 1, 1.0), d_cnt.ensure(n_chunks + 1, 1.0), d_out_off.ensure(n_chunks + 2, 1.0);
	if (n_chunks) HIP_CHECK(hipMemcpyAsync(d_chunks.p, chunks.data(), n_chunks * sizeof(ChunkDesc), hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_seq_off.p, fi.seq_off.data(), n_seq * 8, hipMemcpyHostToDevice, stream));
	HIP_CHECK(hipMemcpyAsync(d_seq_len.p, fi.seq_len.data(), n_seq * 4, hipMemcpyHostToDevice, stream));
	const dim3 sgrid((unsigned)((n_chunks + 63) / 64)), sblock(64);
	const bool hpc = flag & ref::I_HPC;
	auto run_sketch = [&](bool emit, uint64_t *oh, uint64_t *op) {
		if (n_chunks == 0) return;
#define MM2_IDX_SKETCH(E, W, H) hipLaunchKernelGGL((idx_sketch_kernel<E, W, H>), sgrid, sblock, 0, stream, d_nt4.p, d_seq_off.p, d_seq_len.p, d_chunks.p, n_chunks, chunk_len, w, k, d_cnt.p, d_out_off.p, oh, op)
		if (w <= 32) {
			if (hpc) { if (emit) MM2_IDX_SKETCH(true, 32, true); else MM2_IDX_SKETCH(false, 32, true); }
			else { if (emit) MM2_IDX_SKETCH(true, 32, false); else MM2_IDX_SKETCH(false, 32, false); }
		} else {
			if (hpc) { if (emit) MM2_IDX_SKETCH(true, 256, true); else MM2_IDX_SKETCH(false, 256, true); }
			else { if (emit) MM2_IDX_SKETCH(true, 256, false); else MM2_IDX_SKETCH(false, 256, false); }
		}
#undef MM2_IDX_SKETCH
		HIP_CHECK(hipGetLastError());
	};
	run_sketch(false, nullptr, nullptr);
	std::vector<uint32_t> h_cnt(n_chunks);
	if (n_chunks) HIP_CHECK(hipMemcpyAsync(h_cnt.data(), d_cnt.p, n_chunks * 4, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	std::vector<uint64_t> h_off(n_chunks + 1);
	h_off[0] = 0;
	for (uint64_t i = 0; i < n_chunks; ++i) h_off[i + 1] = h_off[i] + h_cnt[i];
	const uint64_t n_mz = h_off[n_chunks];
	if (n_mz >= (1ull << 32)) throw std::invalid_argument("[mm2amd] more than 2^32 minimizers: split the reference (the reference's -I does the same)");
	HIP_CHECK(hipMemcpyAsync(d_out_off.p, h_off.data(), (n_chunks + 1) * 8, hipMemcpyHostToDevice, stream));
	DevBuf<uint64_t> d_hash, d_pos, d_hash2, d_pos2;
	d_hash.ensure(n_mz + 1, 1.0), d_pos.ensure(n_mz + 1, 1.0), d_hash2.ensure(n_mz + 1, 1.0), d_pos2.ensure(n_mz + 1, 1.0);
	run_sketch(true, d_hash.p, d_pos.p);
	d_nt4.release();
	// 3. sort by (hash, pos): positions are already ascending within a chunk and chunks are in (rid, start) order, so the
	//    pairs are sorted by pos; one stable sort by hash finishes the job (index.c:236 + :265 yield the same order).
	if (device_sort_pairs_u64(d_hash.p, d_pos.p, d_hash2.p, d_pos2.p, n_mz, 2 * k, stream) == 1)
		std::swap(d_hash.p, d_hash2.p), std::swap(d_hash.cap, d_hash2.cap), std::swap(d_pos.p, d_pos2.p), std::swap(d_pos.cap, d_pos2.cap);
	d_hash2.release(), d_pos2.release();
	// 4. tables
	const uint64_t n_tiles = (n_mz + kSortTile - 1) / kSortTile;
	DevBuf<uint32_t> d_tile;
	d_tile.ensure(n_tiles + 2, 1.0);
	uint64_t n_keys = 0;
	if (n_mz) {
		hipLaunchKernelGGL(idx_count_heads_kernel, dim3((unsigned)n_tiles), dim3(kSortThreads), 0, stream, d_hash.p, n_mz, d_tile.p);
		HIP_CHECK(hipGetLastError());
		device_exclusive_sum_u32(d_tile.p, d_tile.p, n_tiles, stream);
		uint32_t total = 0;
		HIP_CHECK(hipMemcpyAsync(&total, d_tile.p + n_tiles, 4, hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		n_keys = total;
	}
	T.keys.ensure(n_keys + 1, 1.0), T.val_off.ensure(n_keys + 2, 1.0);
	if (n_mz) hipLaunchKernelGGL(idx_scatter_keys_kernel, dim3((unsigned)n_tiles), dim3(kSortThreads), 0, stream, d_hash.p, n_mz, (const uint32_t *)d_tile.p, T.keys.p, T.val_off.p, n_keys);
	else HIP_CHECK(hipMemsetAsync(T.val_off.p, 0, 4, stream));
	HIP_CHECK(hipGetLastError());
	// hand the sorted positions over
	T.pos.release();
	T.pos.p = d_pos.p, T.pos.cap = d_pos.cap, d_pos.p = nullptr, d_pos.cap = 0;
	int hash_bits = 2 * k, want = 1;
	while ((1ull << want) < n_keys && want < 28) ++want;
	T.bucket_bits = std::min(hash_bits, std::max(8, want));
	T.key_shift = hash_bits - T.bucket_bits;
	const uint64_t n_buckets = 1ull << T.bucket_bits;
	T.bucket_start.ensure(n_buckets + 2, 1.0);
	hipLaunchKernelGGL(idx_bucket_start_kernel, dim3((unsigned)((n_buckets + 1 + 255) / 256)), dim3(256), 0, stream, T.keys.p, n_keys, T.key_shift, n_buckets, T.bucket_start.p);
	HIP_CHECK(hipGetLastError());
	T.n_keys = n_keys, T.n_pos = n_mz;
	// occurrence histogram for mm_idx_cal_max_occ (index.c:198-220)
	const int n_bins = 1 << 16;
	DevBuf<unsigned long long> d_hist;
	d_hist.ensure(n_bins, 1.0);
	HIP_CHECK(hipMemsetAsync(d_hist.p, 0, n_bins * 8, stream));
	if (n_keys) hipLaunchKernelGGL(idx_occ_hist_kernel, dim3((unsigned)std::min<uint64_t>((n_keys + 255) / 256, 4096)), dim3(256), 0, stream, T.val_off.p, n_keys, d_hist.p, n_bins);
	T.occ_hist.resize(n_bins);
	HIP_CHECK(hipMemcpyAsync(T.occ_hist.data(), d_hist.p, n_bins * 8, hipMemcpyDeviceToHost, stream));
	HIP_CHECK(hipStreamSynchronize(stream));
	fi.bucket_bits = T.bucket_bits, fi.key_shift = T.key_shift;
	T.make_slots(stream);
}

__global__ void __launch_bounds__(256) idx_make_slots_kernel(const uint64_t *keys, const uint32_t *val_off, uint64_t n_keys, IdxSlot *slots)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_keys; i += stride) {
		IdxSlot s;
		s.key = keys[i], s.off = val_off[i], s.cnt = val_off[i + 1] - val_off[i];
		slots[i] = s;
	}
}
__global__ void __launch_bounds__(256) idx_make_first_kernel(const uint32_t *bucket_start, const IdxSlot *slots, uint64_t n_buckets, IdxSlot *first)
{
	const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
	for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_buckets; b += stride) {
		const uint32_t s = bucket_start[b], e = bucket_start[b + 1];
		IdxSlot f;
		f.key = kIdxNoKey, f.off = 0, f.cnt = 0;
		if (s < e) { f = slots[s]; if (e - s > 1) f.cnt |= kIdxMoreKeys; }
		first[b] = f;
	}
}
void DeviceIndexTables::make_slots(hipStream_t stream)
{
	slots.ensure(n_keys + 1, 1.0);
	if (n_keys) hipLaunchKernelGGL(idx_make_slots_kernel, dim3((unsigned)std::min<uint64_t>((n_keys + 255) / 256, 16384)), dim3(256), 0, stream, keys.p, val_off.p, n_keys, slots.p);
	const uint64_t n_buckets = 1ull << bucket_bits;
	first.ensure(n_buckets + 1, 1.0);
	hipLaunchKernelGGL(idx_make_first_kernel, dim3((unsigned)std::min<uint64_t>((n_buckets + 255) / 256, 16384)), dim3(256), 0, stream, bucket_start.p, slots.p, n_buckets, first.p);
	HIP_CHECK(hipGetLastError());
	stream_wait(stream);
}

void DeviceIndexTables::upload(const FlatIndex &fi, hipStream_t stream)
{
	auto up32 = [&](DevBuf<uint32_t> &d, const std::vector<uint32_t> &h) { d.ensure(h.size() + 1, 1.0); if (!h.empty()) HIP_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * 4, hipMemcpyHostToDevice, stream)); };
	auto up64 = [&](DevBuf<uint64_t> &d, const std::vector<uint64_t> &h) { d.ensure(h.size() + 1, 1.0); if (!h.empty()) HIP_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, stream)); };
	up32(bucket_start, fi.bucket_start), up32(val_off, fi.val_off), up64(keys, fi.keys), up64(pos, fi.pos);
	const size_t s_words = (fi.sum_len + 7) / 8;
	S.ensure(s_words + 1, 1.0);
	if (s_words && fi.S) HIP_CHECK(hipMemcpyAsync(S.p, fi.S, s_words * 4, hipMemcpyHostToDevice, stream)); // an index without sequence (MM_I_NO_SEQ) only serves chain-level mapping
	HIP_CHECK(hipStreamSynchronize(stream));
	n_keys = fi.keys.size(), n_pos = fi.pos.size(), bucket_bits = fi.bucket_bits, key_shift = fi.key_shift;
	occ_hist.assign(1 << 16, 0);
	for (size_t i = 0; i < fi.keys.size(); ++i) { uint32_t c = fi.val_off[i + 1] - fi.val_off[i]; ++occ_hist[c < occ_hist.size() ? c : occ_hist.size() - 1]; }
	make_slots(stream);
}

void DeviceIndexTables::clone_from(const DeviceIndexTables &src, int src_device, int dst_device)
{
	n_keys = src.n_keys, n_pos = src.n_pos, bucket_bits = src.bucket_bits, key_shift = src.key_shift, occ_hist = src.occ_hist;
	auto cp = [&](void *dst, const void *from, size_t bytes) { if (bytes) HIP_CHECK(hipMemcpyPeer(dst, dst_device, from, src_device, bytes)); };
	bucket_start.ensure(((size_t)1 << bucket_bits) + 2, 1.0), cp(bucket_start.p, src.bucket_start.p, (((size_t)1 << bucket_bits) + 1) * 4);
	keys.ensure(n_keys + 1, 1.0), cp(keys.p, src.keys.p, n_keys * 8);
	val_off.ensure(n_keys + 2, 1.0), cp(val_off.p, src.val_off.p, (n_keys + 1) * 4);
	pos.ensure(n_pos + 1, 1.0), cp(pos.p, src.pos.p, n_pos * 8);
	const size_t s_words = src.S.cap; // the packed reference as allocated (its exact length lives in the host index)
	S.ensure(s_words + 1, 1.0), cp(S.p, src.S.p, s_words * 4);
	slots.ensure(n_keys + 1, 1.0), cp(slots.p, src.slots.p, n_keys * sizeof(IdxSlot));
	first.ensure(((size_t)1 << bucket_bits) + 1, 1.0), cp(first.p, src.first.p, ((size_t)1 << bucket_bits) * sizeof(IdxSlot));
	HIP_CHECK(hipDeviceSynchronize());
}

int32_t DeviceIndexTables::cal_max_occ(float f) const
{
	if (f <= 0.f || n_keys == 0) return INT32_MAX;
	const uint64_t kk = (uint32_t)((1. - f) * n_keys); // 0-based rank of the wanted occurrence count
	uint64_t acc = 0;
	for (size_t c = 0; c < occ_hist.size(); ++c) {
		acc += occ_hist[c];
		if (acc > kk) {
			if (c + 1 == occ_hist.size()) throw std::runtime_error("[mm2amd] occurrence count beyond histogram range");
			return (int32_t)c + 1;
		}
	}
	return INT32_MAX;
}

} // namespace mm2amd

// The product's Backend: hand-written gfx950 kernels (seed_chain.hip, ksw_extd2.hip) plus the buffer management
// around them.  There is deliberately no CPU path here: constructing it without a usable HIP device throws.
//
// A batch's sequences are made resident once (begin_batch).  The mapper then runs sub-batches of reads through
// seed_chain()/ksw() on independent LANES: each lane owns a HIP stream and its device/pinned work buffers, so several host
// driver threads can keep different sub-batches in flight and the GPU stages of one overlap the host stages of another.
#include <atomic>
#include <condition_variable>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <thread>
#include "backend.hpp"
#include "chain_host.hpp"
#include "device_ctx.hpp"
#include "ksw_host.hpp"
#include "seed_chain_dev.hpp"
#include "index_build.hpp"
#include "kernel_prof.hpp"
#include "threads.hpp"
#include "trace.hpp"
#include "sdust_core.hpp"

namespace mm2amd {

extern const uint8_t kNt4Table[256];

namespace {

struct Lane {
	int id = 0;
	int set = 0; // the resident set (HipBackend::res_) this lane's calls work on: Backend::bind_lane
	hipStream_t stream = nullptr;
	SeedChainBuffers B{};
	KswRunner ksw;
	DevBuf<uint64_t> d_a_off, d_mp_off, d_mz_x, d_mz_y, d_minipos, d_skey_in, d_sval_in, d_skey_out, d_sval_out;
	DevBuf<uint32_t> d_mz_cnt, d_sd_n, d_sd_off, d_sd_aoff, d_sd_qpos, d_sd_info, d_n_anchor, d_n_minipos, d_n_seedhit, d_tie;
	DevBuf<int32_t> d_rep_len, d_f, d_p, d_t;
	DevBuf<Anchor> d_anchors;
	DevBuf<uint8_t> d_tbytes;
	DevBuf<uint32_t> d_sort_list;
	PinBuf<uint32_t> h_sort_list;
	DevBuf<uint32_t> d_pieces;                 // the chaining kernels' work list when a read of the launch is cut into pieces (make_pieces)
	PinBuf<uint32_t> h_pieces, h_lj_pieces;
	DevBuf<Anchor> d_lj_out_a;                 // long-join re-chaining (second backtrack's anchors; the first one's are its input)
	DevBuf<uint64_t> d_lj_src;
	PinBuf<Anchor> h_lj_a;
	PinBuf<uint64_t> h_lj_u, h_lj_off, h_lj_aoff, h_lj_uoff;
	PinBuf<int32_t> h_lj_nu, h_lj_nv;
	PinBuf<uint32_t> h_lj_tie, h_lj_list;
	PinBuf<unsigned long long> h_lj_cursor;
	DevBuf<FinRegion> d_fin_regions; DevBuf<FinPiece> d_fin_pieces; DevBuf<uint32_t> d_fin_out; DevBuf<FinResult> d_fin_res; // region_finish.hip
	PinBuf<FinRegion> h_fin_regions; PinBuf<FinPiece> h_fin_pieces; PinBuf<uint32_t> h_fin_out; PinBuf<FinResult> h_fin_res;
	// align_regions (region_dev.hpp): the lane's last seed_chain() in device terms, the hit / window / piece arrays, the results' host copies
	long rg_lo = 0; size_t rg_n = 0; uint64_t rg_n_v = 0, rg_n_v2 = 0, rg_n_u = 0, rg_n_u2 = 0;
	std::vector<int8_t> rg_src; std::vector<uint64_t> rg_aoff, rg_uoff; std::vector<int32_t> rg_nu, rg_nv;
	DevBuf<uint64_t> d_lj_out_u;
	bool rg_lazy = false;
	DevBuf<int32_t> d_span; PinBuf<int32_t> h_span;
	DevBuf<RgnGather> d_gather; PinBuf<RgnGather> h_gather; DevBuf<Anchor> d_fetch_a; DevBuf<uint64_t> d_fetch_mp; PinBuf<Anchor> h_fetch_a; PinBuf<uint64_t> h_fetch_mp;
	DevBuf<RgnRead> d_rg_reads; PinBuf<RgnRead> h_rg_reads;
	DevBuf<Anchor> d_rg_sq; DevBuf<ref::Reg1> d_rg_regs; DevBuf<RgnAux> d_rg_aux; DevBuf<RgnReadOut> d_rg_rout; DevBuf<unsigned int> d_rg_cur;
	DevBuf<RgnPlan> d_rg_plan; DevBuf<RgnWin> d_rg_win; DevBuf<KswJob> d_rg_jobs; DevBuf<int32_t> d_rg_sites; DevBuf<FinRegion> d_rg_fin; DevBuf<FinPiece> d_rg_pieces;
	DevBuf<FinResult> d_rg_finres; DevBuf<uint32_t> d_rg_out;
	PinBuf<ref::Reg1> h_rg_regs; PinBuf<RgnAux> h_rg_aux; PinBuf<RgnReadOut> h_rg_rout; PinBuf<unsigned int> h_rg_cur; PinBuf<RgnPlan> h_rg_plan; PinBuf<KswJob> h_rg_jobs;
	PinBuf<FinRegion> h_rg_fin; PinBuf<FinResult> h_rg_finres; PinBuf<uint32_t> h_rg_out;
	PinBuf<Anchor> h_anchors, h_redo;
	PinBuf<int32_t> h_rep, h_nu, h_nv;
	PinBuf<uint64_t> h_minipos, h_off, h_u, h_aoff, h_uoff;
	PinBuf<unsigned long long> h_cursor;
	DevBuf<unsigned long long> d_bt_cursor;
	DevBuf<Anchor> d_bt_out_a;
	DevBuf<uint64_t> d_bt_out_u, d_bt_aoff, d_bt_uoff;
	DevBuf<int32_t> d_bt_nu, d_bt_nv;
	PinBuf<uint32_t> h_na, h_nmp, h_dust_n, h_dust_s, h_dust_e, h_tie;
	std::vector<uint64_t> a_off, mp_off;
	~Lane() { if (stream) (void)hipStreamDestroy(stream); }
};

class HipBackend : public Backend {
public:
	// `tables` may be null: the backend then mirrors the host tables of `fi` (an index flattened from a reference mm_idx_t)
	// `device` < 0: the process's default device.  `tables_device`: where `tables` live; a replica on another device copies them
	// (hipMemcpyPeer over xGMI), one on the same device shares them.
	HipBackend(const FlatIndex &fi, DeviceIndexTables *tables, int n_threads, int device, int replica, int tables_device) : T_(tables)
	{
		replica_ = replica;
		DeviceCtx &d = device_ctx(device);
		std::lock_guard<std::mutex> lk(d.mu);
		ensure_device(d);
		dev_ = d.device_id;
		stream_ = d.stream;
		n_cu_ = d.n_cu;
		n_threads_ = n_threads > 0 ? n_threads : (int)std::max(1u, std::thread::hardware_concurrency());
		if (!T_) { own_.upload(fi, stream_); T_ = &own_; }
		else if (tables_device >= 0 && tables_device != dev_) { own_.clone_from(*T_, tables_device, dev_); T_ = &own_; }
		I_.bucket_start = T_->bucket_start.p, I_.keys = T_->keys.p, I_.val_off = T_->val_off.p, I_.slots = T_->slots.p, I_.first = getenv("MM2AMD_NO_FIRST_SLOT") ? nullptr : T_->first.p, I_.pos = T_->pos.p, I_.S = T_->S.p;
		I_.bucket_bits = T_->bucket_bits, I_.key_shift = T_->key_shift;
		I_.name_rank = nullptr, I_.seq_len = nullptr;
		fi_names_ = &fi.names, fi_seq_len_ = &fi.seq_len, fi_seq_off_ = &fi.seq_off; // fi outlives the backend (it is the mapper's index)
		while ((1ull << rid_bits_) < fi.n_seq) ++rid_bits_;
		n_lanes_ = 8; // sub-batches in flight.  Rounds 1-3: 5 (more made no difference while the lanes' waits spun on the CPU quota); with the lanes starting
		              // on the next batch early, 8-10 keep seeding, sorting and DP kernels of different sub-batches on the GPU together: +3-4 % (profiles/r04, call 16)
		if (const char *e = getenv("MM2AMD_LANES")) n_lanes_ = std::max(1, std::min(kMaxProfLanes, atoi(e)));
		for (int i = 0; i < n_lanes_; ++i) {
			lanes_.emplace_back(new Lane);
			lanes_.back()->id = i;
			HIP_CHECK(hipStreamCreateWithFlags(&lanes_.back()->stream, hipStreamNonBlocking));
			lanes_.back()->ksw.n_cu = n_cu_;
			lanes_.back()->ksw.disable_fast = getenv("MM2AMD_KSW_EXACT_ONLY") != nullptr;
		}
		// the lanes' work buffers come out of arenas (hip_util.hpp): the first chunks are taken now, before any batch, by a thread of its own -- the index tables are
		// still on their way to the device, and what a 100 000-read batch of long reads settles at (116 GB in eight lanes' buffers, 30 ms per GB of hipMalloc, 185 ms
		// per GB of pinned memory) is seconds of allocation that the first batch would otherwise pay for; never more than half of what the device has free
		{
			size_t free_b = 0, total_b = 0, want = arena_env_gb("MM2AMD_ARENA_DEV_GB", 112) << 30;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) want = std::min(want, free_b / 2);
			const size_t want_pin = arena_env_gb("MM2AMD_ARENA_PIN_GB", 4) << 30;
			const int dev = dev_;
			if (MemArena::enabled() && (want || want_pin)) std::thread([dev, want, want_pin] {
				if (hipSetDevice(dev) != hipSuccess) return;
				try { pin_arena().reserve(want_pin); dev_arena().reserve(want); } catch (...) {} // (a reservation that fails leaves the buffers to allocate as they grow)
			}).detach();
		}
	}

	~HipBackend() override
	{
		(void)hipSetDevice(dev_);
		if (stage_stream_) (void)hipStreamDestroy(stage_stream_);
		for (int l = 0; l < n_lanes_; ++l) kernel_profiler(l, replica_).drop_events();
	}
	int n_lanes() const override { return n_lanes_; }
	void enable_name_rules() override
	{
		HIP_CHECK(hipSetDevice(dev_)); // HIP's current device is per thread, and the mapper drives every lane from a thread of its own
		if (name_rules_ || fi_names_->empty()) return;
		const std::vector<std::string> &nm = *fi_names_;
		sorted_names_ = nm;
		std::sort(sorted_names_.begin(), sorted_names_.end());
		sorted_names_.erase(std::unique(sorted_names_.begin(), sorted_names_.end()), sorted_names_.end());
		std::vector<int32_t> rank(nm.size());
		for (size_t i = 0; i < nm.size(); ++i) rank[i] = (int32_t)(std::lower_bound(sorted_names_.begin(), sorted_names_.end(), nm[i]) - sorted_names_.begin());
		d_name_rank_.ensure(nm.size()), d_ref_len_.ensure(nm.size());
		HIP_CHECK(hipMemcpyAsync(d_name_rank_.p, rank.data(), nm.size() * 4, hipMemcpyHostToDevice, stream_));
		HIP_CHECK(hipMemcpyAsync(d_ref_len_.p, fi_seq_len_->data(), nm.size() * 4, hipMemcpyHostToDevice, stream_));
		stream_wait(stream_);
		I_.name_rank = d_name_rank_.p, I_.seq_len = d_ref_len_.p;
		name_rules_ = true;
	}
	void enable_seq_len() override
	{
		HIP_CHECK(hipSetDevice(dev_)); // HIP's current device is per thread, and the mapper drives every lane from a thread of its own
		if (I_.seq_len) return;
		d_ref_len_.ensure(fi_seq_len_->size());
		HIP_CHECK(hipMemcpyAsync(d_ref_len_.p, fi_seq_len_->data(), fi_seq_len_->size() * 4, hipMemcpyHostToDevice, stream_));
		stream_wait(stream_);
		I_.seq_len = d_ref_len_.p;
	}
	bool supports_rmq() const override { return getenv("MM2AMD_RMQ_ON_HOST") == nullptr; } // chain_rmq_kernel; MM2AMD_RMQ_ON_HOST=1: A/B against rmq_chain.cpp
	void set_active_lanes(int n) override { active_lanes_ = std::max(1, std::min(n, n_lanes_)); }

	void begin_batch(const std::vector<ReadView> &reads, std::vector<uint64_t> &qpool_off) override
	{
		HIP_CHECK(hipSetDevice(dev_)); // HIP's current device is per thread, and the mapper drives every lane from a thread of its own
		Resident &R = res_[1 - cur_]; // the set that is not being mapped
		if (!stage_stream_) HIP_CHECK(hipStreamCreateWithFlags(&stage_stream_, hipStreamNonBlocking));
		hipStream_t stream_ = stage_stream_;
		std::vector<uint64_t> &seq_off_ = R.seq_off, &unit_off_ = R.unit_off;
		std::vector<int32_t> &unit_first_ = R.unit_first, &name_key_ = R.name_key;
		bool &has_pairs_ = R.has_pairs, &have_read_names_ = R.have_read_names;
		size_t &n_units_ = R.n_units;
		PinBuf<char> &h_ascii_ = R.h_ascii;
		DevBuf<uint8_t> &d_qpool_ = R.d_qpool;
		DevBuf<uint64_t> &d_seq_off_ = R.d_seq_off, &d_unit_off_ = R.d_unit_off;
		DevBuf<int32_t> &d_unit_first_ = R.d_unit_first, &d_name_key_ = R.d_name_key;
		std::vector<const char *> &read_names_ = R.read_names;
		const size_t n = reads.size();
		// fragment-level offsets (what seeding and chaining see: a pair is one query, the concatenation of its two reads) ...
		seq_off_.resize(n + 1);
		seq_off_[0] = 0;
		has_pairs_ = false;
		for (size_t i = 0; i < n; ++i) seq_off_[i + 1] = seq_off_[i] + (uint64_t)reads[i].total(), has_pairs_ |= reads[i].paired();
		const uint64_t total = seq_off_[n];
		qpool_off.resize(n);
		for (size_t i = 0; i < n; ++i) qpool_off[i] = 2 * seq_off_[i];
		char *h = h_ascii_.ensure(total + 1);
		parallel_for_side(n_threads_, (long)n, [&](long i, int) {
			memcpy(h + seq_off_[i], reads[i].seq, reads[i].len);
			if (reads[i].paired()) memcpy(h + seq_off_[i] + reads[i].len, reads[i].seq2, reads[i].len2);
		}, 64);
		// The ASCII bases stay in the pinned host buffer: encode_kernel reads them from there, once, sub-batch by sub-batch (zero-copy
		// over PCIe, spread over the mapping step).  A bulk H2D copy of the next batch beside the mapping of the current one made the
		// mapping's own small, latency-critical copies queue behind it (profiles/r03: mapping calls 25 % slower in the pipeline).
		d_qpool_.ensure(2 * total + 16);
		d_seq_off_.ensure(n + 1);
		// ... and unit-level offsets (what encoding and sketching see: every read of a pair on its own, so that no k-mer spans the
		// two and each has its own forward | reverse-complement block in the query pool)
		n_units_ = n;
		if (has_pairs_) {
			unit_first_.resize(n + 1);
			unit_first_[0] = 0;
			for (size_t i = 0; i < n; ++i) unit_first_[i + 1] = unit_first_[i] + (reads[i].paired() ? 2 : 1);
			n_units_ = (size_t)unit_first_[n];
			unit_off_.resize(n_units_ + 1);
			for (size_t i = 0; i < n; ++i) {
				unit_off_[unit_first_[i]] = seq_off_[i];
				if (reads[i].paired()) unit_off_[unit_first_[i] + 1] = seq_off_[i] + (uint64_t)reads[i].len;
			}
			unit_off_[n_units_] = total;
			d_unit_off_.ensure(n_units_ + 1), d_unit_first_.ensure(n + 1);
			HIP_CHECK(hipMemcpyAsync(d_unit_off_.p, unit_off_.data(), (n_units_ + 1) * 8, hipMemcpyHostToDevice, stream_));
			HIP_CHECK(hipMemcpyAsync(d_unit_first_.p, unit_first_.data(), (n + 1) * 4, hipMemcpyHostToDevice, stream_));
		}
		HIP_CHECK(hipMemcpyAsync(d_seq_off_.p, seq_off_.data(), (n + 1) * 8, hipMemcpyHostToDevice, stream_));
		read_names_.clear();
		if (getenv("MM2AMD_SEED_DUMP")) for (size_t i = 0; i < n; ++i) read_names_.push_back(reads[i].name);
		have_read_names_ = false;
		if (name_rules_ && n > 0) { // strcmp(qname, target name) as two ranks per read (see skip_hit in seed_chain.hip)
			have_read_names_ = true;
			for (size_t i = 0; i < n; ++i) if (!reads[i].name) { have_read_names_ = false; break; } // the reference skips the rules without a name (map.c:81)
		}
		if (have_read_names_) {
			name_key_.resize(2 * n);
			parallel_for_side(n_threads_, (long)n, [&](long i, int) {
				const std::string q(reads[i].name);
				const size_t lb = (size_t)(std::lower_bound(sorted_names_.begin(), sorted_names_.end(), q) - sorted_names_.begin());
				name_key_[i] = (int32_t)lb;
				name_key_[n + i] = lb < sorted_names_.size() && sorted_names_[lb] == q ? (int32_t)lb : -1;
			}, 256);
			d_name_key_.ensure(2 * n);
			HIP_CHECK(hipMemcpyAsync(d_name_key_.p, name_key_.data(), 2 * n * 4, hipMemcpyHostToDevice, stream_));
		}
		stream_wait(stream_); // the batch is resident; everything after this is the hot path
	}
	void activate_batch() override { cur_ = 1 - cur_; }
	int current_set() const override { return cur_; }
	void bind_lane(int lane, int set) override { lanes_.at(lane)->set = set; }
	bool stages_beside_mapping() const override { return true; }

	void seed_chain(const SeedChainParams &P, long lo, long hi, int lane_id, int n_threads, std::vector<ReadChains> &out) override
	{
		HIP_CHECK(hipSetDevice(dev_)); // HIP's current device is per thread, and the mapper drives every lane from a thread of its own
		Lane &ln = *lanes_.at(lane_id);
		hipStream_t st = ln.stream;
		SeedChainBuffers &B = ln.B;
		const Resident &R = res_[ln.set];
		const std::vector<uint64_t> &seq_off_ = R.seq_off, &unit_off_ = R.unit_off;
		const std::vector<int32_t> &unit_first_ = R.unit_first;
		const bool has_pairs_ = R.has_pairs, have_read_names_ = R.have_read_names;
		const DevBuf<uint8_t> &d_qpool_ = R.d_qpool;
		const DevBuf<uint64_t> &d_seq_off_ = R.d_seq_off, &d_unit_off_ = R.d_unit_off;
		const DevBuf<int32_t> &d_unit_first_ = R.d_unit_first, &d_name_key_ = R.d_name_key;
		const PinBuf<char> &h_ascii_ = R.h_ascii;
		const std::vector<const char *> &read_names_ = R.read_names;
		const size_t n = (size_t)(hi - lo);
		out.clear();
		out.resize(n);
		if (n == 0) return;
		B = SeedChainBuffers();
		B.n_reads = (int)n, B.seq_off = d_seq_off_.p + lo, B.ascii = h_ascii_.p, B.qpool = d_qpool_.p; // (pinned host memory, device-visible)
		if (have_read_names_) B.name_lb = d_name_key_.p + lo, B.name_eq = d_name_key_.p + seq_off_.size() - 1 + lo;
		KernelProfiler &kp = kernel_profiler(lane_id, replica_);
		double tt = Trace::now();
		const double L = (double)(seq_off_[hi] - seq_off_[lo]);
		// encoding and sketching run over units (see begin_batch); without pairs a unit is a fragment
		SeedChainBuffers Bu = B;
		const size_t ulo = has_pairs_ ? (size_t)unit_first_[lo] : (size_t)lo, n_unit = has_pairs_ ? (size_t)unit_first_[hi] - ulo : n;
		if (has_pairs_) Bu.n_reads = (int)n_unit, Bu.seq_off = d_unit_off_.p + ulo;
		kp.begin(st); launch_encode(Bu, st); kp.end(st, "encode_kernel", 3 * L);
		// 1. minimizers, written from slot seq_off[r] of the minimizer arrays (at most one per base)
		const uint64_t base0 = seq_off_[lo], cap_mz = seq_off_[hi] - base0;
		ln.d_mz_cnt.ensure(n_unit);
		ln.d_mz_x.ensure(cap_mz + 1), ln.d_mz_y.ensure(cap_mz + 1);
		ln.d_sd_n.ensure(cap_mz + 1), ln.d_sd_off.ensure(cap_mz + 1), ln.d_sd_aoff.ensure(cap_mz + 1), ln.d_sd_qpos.ensure(cap_mz + 1), ln.d_sd_info.ensure(cap_mz + 1);
		// the per-read slot offsets are the batch-wide base offsets; shifting the array bases by the sub-batch's first offset makes them local
		B.mz_cnt = ln.d_mz_cnt.p, B.mz_off = B.seq_off, B.mz_x = ln.d_mz_x.p - base0, B.mz_y = ln.d_mz_y.p - base0;
		B.sd_n = ln.d_sd_n.p - base0, B.sd_off = ln.d_sd_off.p - base0, B.sd_aoff = ln.d_sd_aoff.p - base0, B.sd_qpos = ln.d_sd_qpos.p - base0, B.sd_info = ln.d_sd_info.p - base0;
		const double est_mz = 2.0 * L / (P.w + 1);
		Bu.mz_cnt = B.mz_cnt, Bu.mz_off = Bu.seq_off, Bu.mz_x = B.mz_x, Bu.mz_y = B.mz_y;
		int max_len = 0; // sizes the sketch kernel's LDS tile
		for (size_t u = 0; u < n_unit; ++u) {
			const uint64_t l_u = has_pairs_ ? unit_off_[ulo + u + 1] - unit_off_[ulo + u] : seq_off_[lo + u + 1] - seq_off_[lo + u];
			max_len = (int)std::max<uint64_t>((uint64_t)max_len, l_u);
		}
		kp.begin(st); launch_sketch(Bu, P, max_len, st); kp.end(st, "sketch_kernel", L + 16.0 * est_mz);
		if (has_pairs_) // seed_collect joins the minimizer lists of a pair's two units (collect_minimizers, map.c:59-72); indices are batch-wide
			B.unit_first = d_unit_first_.p + lo, B.unit_off = d_unit_off_.p, B.unit_cnt = ln.d_mz_cnt.p - ulo, B.mz_cnt = nullptr;
		if (P.sdust_thres > 0) { // -T: drop minimizers in low-complexity regions (the seed arrays are still unused: they carry the regions)
			uint32_t *h_n = ln.h_dust_n.ensure(cap_mz + 1), *h_s = ln.h_dust_s.ensure(cap_mz + 1), *h_e = ln.h_dust_e.ensure(cap_mz + 1);
			const char *asc = h_ascii_.p;
			parallel_for(n_threads, (long)n_unit, [&](long u, int) { // sdust_core per read on the host, while the sketch kernel runs
				const uint64_t o = has_pairs_ ? unit_off_[ulo + u] : seq_off_[lo + u];
				const int len = (int)((has_pairs_ ? unit_off_[ulo + u + 1] : seq_off_[lo + u + 1]) - o);
				thread_local std::vector<SdustState::Perf> perf(SdustState::PCAP);
				thread_local std::vector<uint8_t> codes;
				codes.resize((size_t)len + 1);
				for (int j = 0; j < len; ++j) codes[j] = kNt4Table[(uint8_t)asc[o + j]];
				SdustState S;
				S.P = perf.data();
				uint32_t k = 0;
				sdust_scan(codes.data(), len, P.sdust_thres, S, [&](int s0, int e0) { h_s[o - base0 + k] = (uint32_t)s0, h_e[o - base0 + k] = (uint32_t)e0; ++k; });
				if (len > 0) h_n[o - base0] = k;
			}, 16);
			HIP_CHECK(hipMemcpyAsync(ln.d_sd_n.p, h_n, cap_mz * 4, hipMemcpyHostToDevice, st));
			HIP_CHECK(hipMemcpyAsync(ln.d_sd_off.p, h_s, cap_mz * 4, hipMemcpyHostToDevice, st));
			HIP_CHECK(hipMemcpyAsync(ln.d_sd_aoff.p, h_e, cap_mz * 4, hipMemcpyHostToDevice, st));
			kp.begin(st); launch_dust_filter(B, st); kp.end(st, "dust_filter_kernel", 16.0 * est_mz);
		}
		// 2. seeds: probe, filter, count anchors
		ln.d_n_anchor.ensure(n), ln.d_n_minipos.ensure(n), ln.d_n_seedhit.ensure(n), ln.d_rep_len.ensure(n);
		B.n_anchor = ln.d_n_anchor.p, B.n_minipos = ln.d_n_minipos.p, B.n_seedhit = ln.d_n_seedhit.p, B.rep_len = ln.d_rep_len.p;
		kp.begin(st); launch_seed_collect(B, I_, P, st); kp.end(st, "seed_collect_kernel", 36.0 * est_mz); // per minimizer: 16 B record + 8 key + 8 val + 4 flags
		uint32_t *h_na = ln.h_na.ensure(n), *h_nmp = ln.h_nmp.ensure(n);
		int32_t *h_rep = ln.h_rep.ensure(n);
		HIP_CHECK(hipMemcpyAsync(h_na, ln.d_n_anchor.p, n * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(h_nmp, ln.d_n_minipos.p, n * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(h_rep, ln.d_rep_len.p, n * 4, hipMemcpyDeviceToHost, st));
		stream_wait(st);
		Trace::get().add(lane_id, "gpu:sketch+collect", tt, Trace::now()); tt = Trace::now();
		std::vector<uint64_t> &a_off = ln.a_off, &mp_off = ln.mp_off;
		a_off.resize(n + 1), mp_off.resize(n + 1);
		a_off[0] = mp_off[0] = 0;
		for (size_t i = 0; i < n; ++i) a_off[i + 1] = a_off[i] + h_na[i], mp_off[i + 1] = mp_off[i] + h_nmp[i];
		const uint64_t n_a = a_off[n], n_mp = mp_off[n];
		uint64_t *h_off = ln.h_off.ensure(2 * (n + 1));
		memcpy(h_off, a_off.data(), (n + 1) * 8), memcpy(h_off + n + 1, mp_off.data(), (n + 1) * 8);
		ln.d_a_off.ensure(n + 1), ln.d_mp_off.ensure(n + 1);
		HIP_CHECK(hipMemcpyAsync(ln.d_a_off.p, h_off, (n + 1) * 8, hipMemcpyHostToDevice, st));
		HIP_CHECK(hipMemcpyAsync(ln.d_mp_off.p, h_off + n + 1, (n + 1) * 8, hipMemcpyHostToDevice, st));
		ln.d_anchors.ensure(n_a + 1), ln.d_minipos.ensure(n_mp + 1), ln.d_f.ensure(n_a + 1), ln.d_p.ensure(n_a + 1), ln.d_t.ensure(n_a + 1);
		ln.d_skey_in.ensure(n_a + 1), ln.d_sval_in.ensure(n_a + 1), ln.d_skey_out.ensure(n_a + 1), ln.d_sval_out.ensure(n_a + 1), ln.d_tie.ensure(2 * n + 1);
		B.sort_key_in = ln.d_skey_in.p, B.sort_val_in = ln.d_sval_in.p, B.sort_key_out = ln.d_skey_out.p, B.sort_val_out = ln.d_sval_out.p, B.tie_flag = ln.d_tie.p;
		B.tie_list = ln.d_tie.p + n, B.tie_count = ln.d_tie.p + 2 * n;
		B.rid_bits = rid_bits_;
		// the per-read anchor sort's launch classes (by anchors per read): the reads of a class are listed together
		int n_class[kAnchorSortClasses] = { 0 }, class_first[kAnchorSortClasses + 1] = { 0 };
		double a_class[kAnchorSortClasses] = { 0 };
		for (size_t i = 0; i < n; ++i) if (h_na[i]) { const int c = anchor_sort_class(h_na[i], rid_bits_); ++n_class[c], a_class[c] += h_na[i]; }
		for (int c = 0; c < kAnchorSortClasses; ++c) class_first[c + 1] = class_first[c] + n_class[c];
		uint32_t *h_list = ln.h_sort_list.ensure(n + 1);
		{
			int fill[kAnchorSortClasses];
			for (int c = 0; c < kAnchorSortClasses; ++c) fill[c] = class_first[c];
			for (size_t i = 0; i < n; ++i) if (h_na[i]) h_list[fill[anchor_sort_class(h_na[i], rid_bits_)]++] = (uint32_t)i;
		}
		ln.d_sort_list.ensure(n + 1);
		if (class_first[kAnchorSortClasses]) HIP_CHECK(hipMemcpyAsync(ln.d_sort_list.p, h_list, (size_t)class_first[kAnchorSortClasses] * 4, hipMemcpyHostToDevice, st));
		B.a_off = ln.d_a_off.p, B.mp_off = ln.d_mp_off.p, B.anchors = ln.d_anchors.p, B.mini_pos = ln.d_minipos.p;
		B.f = ln.d_f.p, B.p = ln.d_p.p, B.t = ln.d_t.p;
		// 3. anchors: expand, sort, chain
		kp.begin(st); launch_seed_expand(B, I_, P, st); kp.end(st, "seed_expand_kernel", 24.0 * n_a);
		launch_anchor_sort(B, I_, P, ln.d_sort_list.p, n_class, a_class, st, &kp);
		if (getenv("MM2AMD_TIE_COUNT")) { // diagnostics: how many reads of this sub-batch had two anchors with the same x (the replayed ones)
			std::vector<uint32_t> tf(n);
			stream_wait(st);
			HIP_CHECK(hipMemcpy(tf.data(), ln.d_tie.p, n * 4, hipMemcpyDeviceToHost));
			size_t n_tied = 0;
			for (size_t i = 0; i < n; ++i) n_tied += tf[i] != 0;
			fprintf(stderr, "[mm2amd] anchor sort: %zu of %zu reads have duplicated keys (%d / %d / %d reads in the 7 k / 10 k / global classes)\n", n_tied, n, n_class[3], n_class[4], n_class[5]);
		}
		if (const char *dump = getenv("MM2AMD_SEED_DUMP")) { // diagnostics: every read's sorted anchors, as the reference's --print-seeds prints them (map.c:255-260)
			std::vector<Anchor> all(n_a + 1);
			stream_wait(st);
			if (n_a) HIP_CHECK(hipMemcpy(all.data(), ln.d_anchors.p, n_a * sizeof(Anchor), hipMemcpyDeviceToHost));
			static std::mutex dump_mu;
			std::lock_guard<std::mutex> lk(dump_mu);
			if (FILE *fp = fopen(dump, "a")) {
				for (size_t i = 0; i < n; ++i) {
					fprintf(fp, "QR\t%s\t%d\nRS\t%d\n", read_names_.empty() || !read_names_[lo + i] ? "*" : read_names_[lo + i], P.mid_occ, h_rep[i]);
					for (uint64_t j = a_off[i]; j < a_off[i + 1]; ++j) {
						const Anchor &a = all[j];
						fprintf(fp, "SD\t%s\t%d\t%c\t%d\t%d\t%d\n", (*fi_names_)[a.x << 1 >> 33].c_str(), (int32_t)a.x, "+-"[a.x >> 63], (int32_t)a.y, (int32_t)(a.y >> 32 & 0xff),
						        j == a_off[i] ? 0 : ((int32_t)a.y - (int32_t)all[j - 1].y) - ((int32_t)a.x - (int32_t)all[j - 1].x));
					}
				}
				fclose(fp);
			}
		}
		if (P.anchors_only) { // the caller chains (RMQ): hand over the sorted anchors as they are
			Anchor *ha = ln.h_anchors.ensure(n_a + 1);
			uint64_t *hmp = ln.h_minipos.ensure(n_mp + 1);
			if (n_a) HIP_CHECK(hipMemcpyAsync(ha, ln.d_anchors.p, n_a * sizeof(Anchor), hipMemcpyDeviceToHost, st));
			if (n_mp) HIP_CHECK(hipMemcpyAsync(hmp, ln.d_minipos.p, n_mp * 8, hipMemcpyDeviceToHost, st));
			stream_wait(st);
			Trace::get().add(lane_id, "gpu:expand+sort, d2h:anchors", tt, Trace::now());
			kp.collect();
			parallel_for(n_threads, (long)n, [&](long i, int) {
				ReadChains &c = out[i];
				c.rep_len = h_rep[i];
				c.mp_p = hmp + mp_off[i], c.n_mp = (int32_t)(mp_off[i + 1] - mp_off[i]);
				c.u_p = nullptr, c.n_u = 0, c.chained = false;
				c.a_p = ha + a_off[i], c.n_a = (int64_t)(a_off[i + 1] - a_off[i]);
			}, 64);
			return;
		}
		make_pieces(B, ln, ln.h_pieces, n, [&](size_t i) { return (uint64_t)h_na[i]; }, P.rmq ? rmq_piece_len() : fill_piece_len(), st);
		if (P.rmq) { kp.begin(st); launch_chain_rmq(B, P, st); kp.end(st, "chain_rmq_kernel", 32.0 * n_a); }
		else { kp.begin(st); launch_chain_fill(B, P, st); kp.end(st, "chain_fill_kernel", 24.0 * n_a); }
		B.pieces = nullptr, B.n_pieces = 0;
		// 4. chains: backtrack + compaction on the device, then only the chained anchors travel to the host
		ln.d_bt_cursor.ensure(2), ln.d_bt_out_a.ensure(n_a + 1), ln.d_bt_out_u.ensure((P.min_cnt >= 2 ? n_a / 2 : n_a) + n + 1); // a chain has at least max(1, min_cnt) anchors (lchain.c:66)
		ln.d_bt_nu.ensure(n), ln.d_bt_nv.ensure(n), ln.d_bt_aoff.ensure(n), ln.d_bt_uoff.ensure(n);
		B.bt_cursor = ln.d_bt_cursor.p, B.bt_out_a = ln.d_bt_out_a.p, B.bt_out_u = ln.d_bt_out_u.p;
		B.bt_nu = ln.d_bt_nu.p, B.bt_nv = ln.d_bt_nv.p, B.bt_aoff = ln.d_bt_aoff.p, B.bt_uoff = ln.d_bt_uoff.p;
		kp.begin(st); launch_chain_backtrack(B, P, st); kp.end(st, "chain_backtrack_kernel", 8.0 * n_a);
		unsigned long long *h_cur = ln.h_cursor.ensure(2);
		int32_t *h_nu = ln.h_nu.ensure(n), *h_nv = ln.h_nv.ensure(n);
		uint64_t *h_aoff = ln.h_aoff.ensure(n), *h_uoff = ln.h_uoff.ensure(n);
		uint64_t *hmp = ln.h_minipos.ensure(n_mp + 1);
		uint32_t *h_tie = P.rmq ? ln.h_tie.ensure(n) : nullptr;
		HIP_CHECK(hipMemcpyAsync(h_cur, ln.d_bt_cursor.p, 16, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(h_nu, ln.d_bt_nu.p, n * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(h_nv, ln.d_bt_nv.p, n * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(h_aoff, ln.d_bt_aoff.p, n * 8, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(h_uoff, ln.d_bt_uoff.p, n * 8, hipMemcpyDeviceToHost, st));
		if (h_tie) HIP_CHECK(hipMemcpyAsync(h_tie, ln.d_tie.p, n * 4, hipMemcpyDeviceToHost, st));
		const bool lazy = P.lazy_chains != 0 && !P.anchors_only; // the chains stay on the device for align_regions()
		ln.rg_lazy = lazy;
		int32_t *h_span = nullptr;
		if (lazy) { // ... and of the anchors only what the long-join question needs comes back
			ln.d_span.ensure(2 * n), h_span = ln.h_span.ensure(2 * n);
			launch_first_chain_span((int)n, ln.d_bt_out_a.p, ln.d_bt_out_u.p, ln.d_bt_aoff.p, ln.d_bt_uoff.p, ln.d_bt_nu.p, ln.d_span.p, st);
			HIP_CHECK(hipMemcpyAsync(h_span, ln.d_span.p, 2 * n * 4, hipMemcpyDeviceToHost, st));
		} else if (n_mp) HIP_CHECK(hipMemcpyAsync(hmp, ln.d_minipos.p, n_mp * 8, hipMemcpyDeviceToHost, st));
		stream_wait(st);
		Trace::get().add(lane_id, "gpu:expand..backtrack", tt, Trace::now()); tt = Trace::now();
		const uint64_t n_v = h_cur[0], n_u = h_cur[1];
		Anchor *ha = ln.h_anchors.ensure(n_v + 1);
		uint64_t *hu = ln.h_u.ensure(n_u + 1);
		if (n_v && !lazy) HIP_CHECK(hipMemcpyAsync(ha, ln.d_bt_out_a.p, n_v * sizeof(Anchor), hipMemcpyDeviceToHost, st));
		if (n_u) HIP_CHECK(hipMemcpyAsync(hu, ln.d_bt_out_u.p, n_u * 8, hipMemcpyDeviceToHost, st));
		// reads the RMQ kernel left to the host (a tie in a range minimum, an over-full neighbourhood, a whole contig): their sorted
		// anchors travel as they are and ReadChains::chained stays false -- the mapper chains them with rmq_chain.cpp
		std::vector<std::pair<size_t, size_t>> redo; // (read, offset into the fallback buffer)
		size_t redo_total = 0;
		if (h_tie) for (size_t i = 0; i < n; ++i) if (h_tie[i]) redo.emplace_back(i, redo_total), redo_total += (size_t)(a_off[i + 1] - a_off[i]);
		Anchor *h_redo = redo_total ? ln.h_redo.ensure(redo_total + 1) : nullptr;
		for (const auto &rd : redo) {
			const size_t cnt = (size_t)(a_off[rd.first + 1] - a_off[rd.first]);
			if (cnt) HIP_CHECK(hipMemcpyAsync(h_redo + rd.second, ln.d_anchors.p + a_off[rd.first], cnt * sizeof(Anchor), hipMemcpyDeviceToHost, st));
			//  with the chains left on the device the minimizer positions stay there too -- but a handed-back read is chained and finished by the
			// host path, whose mm_est_err walks them: its slice comes along
			const size_t n_pos = (size_t)(mp_off[rd.first + 1] - mp_off[rd.first]);
			if (lazy && n_pos) HIP_CHECK(hipMemcpyAsync(hmp + mp_off[rd.first], ln.d_minipos.p + mp_off[rd.first], n_pos * 8, hipMemcpyDeviceToHost, st));
		}
		stream_wait(st);
		Trace::get().add(lane_id, "d2h:chains", tt, Trace::now()); tt = Trace::now();
		kp.collect();
		TraceScope ts(lane_id, "host:chains->vectors");
		parallel_for(n_threads, (long)n, [&](long i, int) {
			ReadChains &c = out[i];
			c.rep_len = h_rep[i];
			c.mp_p = lazy ? nullptr : hmp + mp_off[i], c.n_mp = (int32_t)(mp_off[i + 1] - mp_off[i]);
			c.u_p = hu + h_uoff[i], c.n_u = h_nu[i];
			c.a_p = lazy ? nullptr : ha + h_aoff[i], c.n_a = h_nv[i];
			c.chained = true;
		}, 64);
		ln.rg_lo = lo, ln.rg_n = n, ln.rg_n_v = n_v, ln.rg_n_u = n_u, ln.rg_n_v2 = ln.rg_n_u2 = 0;
		for (size_t i = 0; i < n; ++i) out[i].dev_src = 0, out[i].dev_a_off = h_aoff[i], out[i].dev_u_off = h_uoff[i];
		for (const auto &rd : redo) {
			ReadChains &c = out[rd.first];
			c.u_p = nullptr, c.n_u = 0, c.chained = false, c.dev_src = -1;
			c.mp_p = hmp + mp_off[rd.first];
			c.a_p = h_redo + rd.second, c.n_a = (int64_t)(a_off[rd.first + 1] - a_off[rd.first]);
		}
		if (P.long_join && !has_pairs_ && !getenv("MM2AMD_LONG_JOIN_ON_HOST")) long_join(P, lo, n, ln, kp, out, ha, hu, h_nu, h_nv, h_aoff, h_uoff, h_span); // (the diagnostic switch: every re-chain through rmq_chain.cpp)
	}

	// The chaining kernels give a wavefront to a read.  A read with far more anchors than the others (it crosses a multi-copy element of the reference) makes the whole
	// launch wait for its wavefront, so when a read has more than `len` anchors the launch gets a work list instead: every read's ceil(anchors / len) pieces (seed_chain.hip:
	// chain_piece_bounds moves the cuts to cluster heads, where the sequential rules restart), the reads with several pieces first.  Launches without such a read --
	// every launch of the uniform benchmark -- keep the plain read-per-wavefront grid.  MM2AMD_CHAIN_PIECE / MM2AMD_RMQ_PIECE: the piece lengths (tests set small ones; 0 = never cut).
	static int fill_piece_len() { const char *e = getenv("MM2AMD_CHAIN_PIECE"); return e ? atoi(e) : 4096; }
	static int rmq_piece_len() { const char *e = getenv("MM2AMD_RMQ_PIECE"); return e ? atoi(e) : 512; }
	template <class Count>
	static void make_pieces(SeedChainBuffers &B, Lane &ln, PinBuf<uint32_t> &hbuf, size_t n, Count count, int len, hipStream_t st)
	{
		B.pieces = nullptr, B.n_pieces = 0, B.piece_len = len;
		{ const char *e = getenv("MM2AMD_RMQ_RANK_MAX"); B.rmq_rank_max = e ? std::min(256, std::max(0, atoi(e))) : 256; }
		{ const char *e = getenv("MM2AMD_RMQ_DENSE"); B.piece_dense = e ? atoi(e) : 1024; } // (the RMQ kernel's long clusters go to workgroups: seed_chain.hip chain_rmq_wide_kernel; tests set it low, 0 = off)
		if (len <= 0) return;
		size_t n_p = 0;
		bool any = false;
		for (size_t i = 0; i < n; ++i) { const uint64_t c = count(i); n_p += (size_t)((c + len - 1) / len); any |= c > (uint64_t)len; }
		if (!any || n_p > (size_t)INT32_MAX) return;
		uint32_t *h = hbuf.ensure(2 * n_p);
		size_t k = 0;
		for (int pass = 0; pass < 2; ++pass) // reads with several pieces first: their wavefronts start with the launch
			for (size_t i = 0; i < n; ++i) {
				const uint64_t c = count(i), m = (c + len - 1) / len;
				if (m == 0 || (m > 1) != (pass == 0)) continue;
				for (uint64_t q = 0; q < m; ++q) h[2 * k] = (uint32_t)i, h[2 * k + 1] = (uint32_t)q, ++k;
			}
		ln.d_pieces.ensure(2 * n_p);
		HIP_CHECK(hipMemcpyAsync(ln.d_pieces.p, h, 2 * n_p * 4, hipMemcpyHostToDevice, st));
		B.pieces = ln.d_pieces.p, B.n_pieces = (int)n_p;
		if (getenv("MM2AMD_PIECE_DEBUG")) fprintf(stderr, "[mm2amd] chaining work list: %zu reads in %zu pieces of %d anchors\n", n, n_p, len);
	}

	// map.c:283-292 on the device.  Which reads re-chain is decided here from the first chains (the reference's two conditions); their CHAINED
	// anchors -- still in the backtrack's device output -- go through the per-read sort (by reference position, the reference's tie order),
	// chain_rmq_kernel with bw_long, and the backtrack again.  A read the RMQ kernel hands back (a range minimum that is not unique, an
	// over-full neighbourhood) keeps its first chains and long_join_done unset: the caller re-chains it with rmq_chain.cpp.
	void long_join(const SeedChainParams &P, long lo, size_t n, Lane &ln, KernelProfiler &kp, std::vector<ReadChains> &out, const Anchor *ha, const uint64_t *hu,
	               const int32_t *h_nu, const int32_t *h_nv, const uint64_t *h_aoff, const uint64_t *h_uoff, const int32_t *h_span)
	{
		const bool lazy = h_span != nullptr; // (the chained anchors stayed on the device: the first chain's ends came back on their own)
		hipStream_t st = ln.stream;
		const std::vector<uint64_t> &seq_off = res_[ln.set].seq_off;
		std::vector<uint32_t> sel;
		for (size_t i = 0; i < n; ++i) {
			ReadChains &c = out[i];
			if (!c.chained) continue; // (an RMQ preset's hand-back: the caller chains it first and asks the question itself)
			c.long_join_done = true;
			if (h_nu[i] <= 1) continue;
			const int qlen = (int)(seq_off[lo + i + 1] - seq_off[lo + i]);
			const int32_t a0 = lazy ? h_span[2 * i] : (int32_t)ha[h_aoff[i]].y, a1 = lazy ? h_span[2 * i + 1] : (int32_t)ha[h_aoff[i] + (uint64_t)((int32_t)hu[h_uoff[i]] - 1)].y;
			if (qlen - (a1 - a0) > P.rmq_rescue_size || a1 - a0 > qlen * P.rmq_rescue_ratio) sel.push_back((uint32_t)i);
		}
		const size_t n2 = sel.size();
		if (n2 == 0) return;
		// the re-chain list as a batch of its own, in the first pass's (now dead) device buffers; only the backtrack's anchors must survive
		uint64_t *h_off = ln.h_lj_off.ensure(2 * (n2 + 1));
		uint64_t *h_src = h_off + n2 + 1;
		h_off[0] = 0;
		for (size_t k = 0; k < n2; ++k) h_off[k + 1] = h_off[k] + (uint64_t)h_nv[sel[k]], h_src[k] = h_aoff[sel[k]];
		const uint64_t n_a2 = h_off[n2];
		SeedChainBuffers B2 = ln.B;
		B2.n_reads = (int)n2, B2.seq_off = nullptr, B2.mz_cnt = nullptr, B2.unit_first = nullptr;
		ln.d_lj_src.ensure(n2 + 1), ln.d_lj_out_a.ensure(n_a2 + 1);
		HIP_CHECK(hipMemcpyAsync(ln.d_a_off.p, h_off, (n2 + 1) * 8, hipMemcpyHostToDevice, st));
		HIP_CHECK(hipMemcpyAsync(ln.d_lj_src.p, h_src, n2 * 8, hipMemcpyHostToDevice, st));
		B2.a_off = ln.d_a_off.p;
		int n_class[kAnchorSortClasses] = { 0 }, class_first[kAnchorSortClasses + 1] = { 0 };
		double a_class[kAnchorSortClasses] = { 0 };
		for (size_t k = 0; k < n2; ++k) { const int c = anchor_sort_class((uint64_t)h_nv[sel[k]], rid_bits_); ++n_class[c], a_class[c] += h_nv[sel[k]]; }
		for (int c = 0; c < kAnchorSortClasses; ++c) class_first[c + 1] = class_first[c] + n_class[c];
		uint32_t *h_list = ln.h_lj_list.ensure(n2 + 1);
		{
			int fill[kAnchorSortClasses];
			for (int c = 0; c < kAnchorSortClasses; ++c) fill[c] = class_first[c];
			for (size_t k = 0; k < n2; ++k) h_list[fill[anchor_sort_class((uint64_t)h_nv[sel[k]], rid_bits_)]++] = (uint32_t)k;
		}
		HIP_CHECK(hipMemcpyAsync(ln.d_sort_list.p, h_list, n2 * 4, hipMemcpyHostToDevice, st));
		kp.begin(st); launch_rechain_gather(B2, ln.d_bt_out_a.p, ln.d_lj_src.p, st); kp.end(st, "rechain_gather_kernel", 32.0 * n_a2);
		SeedChainParams P2 = P;
		P2.rmq = 1, P2.bw = P.bw_long, P2.flag &= ~(int64_t)ref::F_HEAP_SORT; // (the heap-merge order belongs to the seeding; this sort is radix_sort_128x)
		launch_anchor_sort(B2, I_, P2, ln.d_sort_list.p, n_class, a_class, st, &kp);
		make_pieces(B2, ln, ln.h_lj_pieces, n2, [&](size_t k) { return (uint64_t)h_nv[sel[k]]; }, rmq_piece_len(), st);
		kp.begin(st); launch_chain_rmq(B2, P2, st); kp.end(st, "chain_rmq_kernel[long-join]", 32.0 * n_a2);
		ln.d_lj_out_u.ensure((P2.min_cnt >= 2 ? n_a2 / 2 : n_a2) + n2 + 1);
		B2.bt_out_a = ln.d_lj_out_a.p, B2.bt_out_u = ln.d_lj_out_u.p; // (counts and offsets reuse the first pass's arrays: they have been copied out; the first pass's chains stay for align_regions)
		kp.begin(st); launch_chain_backtrack(B2, P2, st); kp.end(st, "chain_backtrack_kernel[long-join]", 8.0 * n_a2);
		unsigned long long *h_cur = ln.h_lj_cursor.ensure(2);
		int32_t *nu2 = ln.h_lj_nu.ensure(n2), *nv2 = ln.h_lj_nv.ensure(n2);
		uint64_t *aoff2 = ln.h_lj_aoff.ensure(n2), *uoff2 = ln.h_lj_uoff.ensure(n2);
		uint32_t *tie2 = ln.h_lj_tie.ensure(n2);
		HIP_CHECK(hipMemcpyAsync(h_cur, ln.d_bt_cursor.p, 16, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(nu2, ln.d_bt_nu.p, n2 * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(nv2, ln.d_bt_nv.p, n2 * 4, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(aoff2, ln.d_bt_aoff.p, n2 * 8, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(uoff2, ln.d_bt_uoff.p, n2 * 8, hipMemcpyDeviceToHost, st));
		HIP_CHECK(hipMemcpyAsync(tie2, ln.d_tie.p, n2 * 4, hipMemcpyDeviceToHost, st));
		stream_wait(st);
		const uint64_t n_v2 = h_cur[0], n_u2 = h_cur[1];
		Anchor *ha2 = ln.h_lj_a.ensure(n_v2 + 1);
		uint64_t *hu2 = ln.h_lj_u.ensure(n_u2 + 1);
		if (n_v2 && !lazy) HIP_CHECK(hipMemcpyAsync(ha2, ln.d_lj_out_a.p, n_v2 * sizeof(Anchor), hipMemcpyDeviceToHost, st));
		if (n_u2) HIP_CHECK(hipMemcpyAsync(hu2, ln.d_lj_out_u.p, n_u2 * 8, hipMemcpyDeviceToHost, st));
		ln.rg_n_v2 = n_v2, ln.rg_n_u2 = n_u2;
		stream_wait(st);
		kp.collect();
		for (size_t k = 0; k < n2; ++k) {
			ReadChains &c = out[sel[k]];
			if (tie2[k]) { c.long_join_done = false; continue; } // the host's tie-exact tree does this one
			c.u_p = hu2 + uoff2[k], c.n_u = nu2[k];
			c.a_p = lazy ? nullptr : ha2 + aoff2[k], c.n_a = nv2[k];
			c.long_joined = true;
			c.dev_src = 1, c.dev_a_off = aoff2[k], c.dev_u_off = uoff2[k];
		}
	}

	void ksw(const std::vector<KswJob> &jobs, const KswScoring &sc, int lane_id, int n_threads, std::vector<KswRes> &res, const uint32_t **cigar) override
	{
		HIP_CHECK(hipSetDevice(dev_)); // HIP's current device is per thread, and the mapper drives every lane from a thread of its own
		Lane &ln = *lanes_.at(lane_id);
		res.resize(jobs.size());
		size_t n_cig = 0;
		ln.ksw.n_threads = n_threads;
		ln.ksw.prof = &kernel_profiler(lane_id, replica_);
		// the DP scratch (one direction-matrix slot per persistent wave) is the big per-lane allocation: split the budget
		// HBM the lanes may spend on direction matrices (1 B per DP cell, one slot per persistent wave); splice gap fills have
		// matrices of tens of MB each, and 288 GB of HBM is what lets thousands of them be in flight
		size_t dir_gb = 160;
		if (const char *e = getenv("MM2AMD_DIR_BUDGET_GB")) dir_gb = atol(e) > 0 ? (size_t)atol(e) : dir_gb;
		ln.ksw.dir_budget = std::min<size_t>((dir_gb << 30) / (size_t)active_lanes_, (size_t)96 << 30); // scratch only grows: keep one lane's share bounded
		ln.ksw.lane = lane_id;
		const uint8_t *d_tbytes = nullptr;
		if (sc.n_tbytes) { // composed targets (mm_align_sr_rna): a small byte pool per call
			ln.d_tbytes.ensure(sc.n_tbytes + 1);
			HIP_CHECK(hipMemcpyAsync(ln.d_tbytes.p, sc.tbytes, sc.n_tbytes, hipMemcpyHostToDevice, ln.stream));
			d_tbytes = ln.d_tbytes.p;
		}
		ln.ksw.run(jobs, res_[ln.set].d_qpool.p, d_tbytes, T_->S.p, sc, res.data(), cigar, &n_cig, ln.stream);
		kernel_profiler(lane_id, replica_).collect();
	}

	void fetch_chains(int lane_id, const std::vector<long> &reads, std::vector<ReadChains> &chains) override
	{
		if (reads.empty()) return;
		HIP_CHECK(hipSetDevice(dev_));
		Lane &ln = *lanes_.at(lane_id);
		hipStream_t st = ln.stream;
		RgnGather *g = ln.h_gather.ensure(reads.size());
		uint64_t na = 0, nmp = 0;
		for (size_t k = 0; k < reads.size(); ++k) {
			const ReadChains &c = chains[reads[k]];
			RgnGather &G = g[k];
			G.a_src = c.dev_a_off, G.a_dst = na, G.mp_src = ln.mp_off[reads[k]], G.mp_dst = nmp;
			G.n_a = (int32_t)c.n_a, G.n_mp = c.n_mp, G.src = c.dev_src == 1 ? 1u : 0u, G.pad = 0;
			na += (uint64_t)c.n_a, nmp += (uint64_t)c.n_mp;
		}
		ln.d_gather.ensure(reads.size()), ln.d_fetch_a.ensure(na + 1), ln.d_fetch_mp.ensure(nmp + 1);
		Anchor *ha = ln.h_fetch_a.ensure(na + 1);
		uint64_t *hmp = ln.h_fetch_mp.ensure(nmp + 1);
		HIP_CHECK(hipMemcpyAsync(ln.d_gather.p, g, reads.size() * sizeof(RgnGather), hipMemcpyHostToDevice, st));
		launch_gather_chains((int)reads.size(), ln.d_gather.p, ln.d_bt_out_a.p, ln.d_lj_out_a.p, ln.d_minipos.p, ln.d_fetch_a.p, ln.d_fetch_mp.p, st);
		if (na) HIP_CHECK(hipMemcpyAsync(ha, ln.d_fetch_a.p, na * sizeof(Anchor), hipMemcpyDeviceToHost, st));
		if (nmp) HIP_CHECK(hipMemcpyAsync(hmp, ln.d_fetch_mp.p, nmp * 8, hipMemcpyDeviceToHost, st));
		stream_wait(st);
		for (size_t k = 0; k < reads.size(); ++k) {
			ReadChains &c = chains[reads[k]];
			c.a_p = ha + g[k].a_dst, c.mp_p = hmp + g[k].mp_dst;
		}
	}

	// region_dev.hpp: chains -> hits -> windows -> DP -> consume -> finish on the device, for the reads of this lane's last seed_chain()
	bool aligns_regions() const override
	{
		const char *e = getenv("MM2AMD_DEVICE_REGIONS"); // =0: the host's chains -> hits, window planning and consumption (A/B checks)
		return !(e && *e == '0');
	}
	void align_regions(int lane_id, const RgnOpts &O, const KswScoring &sc, bool log_gap, const std::vector<ReadChains> &chains, const std::vector<RegionReadIn> &in, int n_threads,
	                   RegionBatchOut &out) override
	{
		HIP_CHECK(hipSetDevice(dev_));
		Lane &ln = *lanes_.at(lane_id);
		hipStream_t st = ln.stream;
		const Resident &R = res_[ln.set];
		const size_t n = ln.rg_n;
		out = RegionBatchOut();
		if (n == 0 || chains.size() < n || in.size() < n) return;
		if (!ref_ready_.load(std::memory_order_acquire)) { // once: the reference sequences' offsets and lengths (the flag is set after the copies have landed)
			std::lock_guard<std::mutex> lk(ref_mu_);
			if (!ref_ready_.load(std::memory_order_relaxed)) {
				d_ref_off_.ensure(fi_seq_off_->size() + 1), d_ref_len2_.ensure(fi_seq_len_->size() + 1);
				HIP_CHECK(hipMemcpyAsync(d_ref_off_.p, fi_seq_off_->data(), fi_seq_off_->size() * 8, hipMemcpyHostToDevice, st));
				HIP_CHECK(hipMemcpyAsync(d_ref_len2_.p, fi_seq_len_->data(), fi_seq_len_->size() * 4, hipMemcpyHostToDevice, st));
				stream_wait(st);
				ref_ready_.store(true, std::memory_order_release);
			}
		}
		KernelProfiler &kp = kernel_profiler(lane_id, replica_);
		double tt = Trace::now();
		// the reads in device terms
		RgnRead *hr = ln.h_rg_reads.ensure(n);
		uint64_t sq = 0, n_chain = 0;
		int max_nu = 1;
		const int stride = R.has_pairs ? 2 : 1; // two-segment fragments: their chains are cut per segment on the device (chain_regs_kernel), a segment is a read of its own from there on
		for (size_t i = 0; i < n; ++i) {
			const ReadChains &c = chains[i];
			RgnRead &r = hr[i];
			const bool skip = in[i].skip || c.dev_src < 0 || !c.chained;
			r.a_off = c.dev_a_off, r.u_off = c.dev_u_off, r.sq_off = sq, r.mp_off = ln.mp_off[i];
			r.qpool_fwd = 2 * R.seq_off[ln.rg_lo + i];
			r.n_u = skip ? 0 : c.n_u, r.n_a = skip ? 0 : (int32_t)c.n_a, r.n_mp = (int32_t)(ln.mp_off[i + 1] - ln.mp_off[i]);
			r.qlen = (int32_t)(R.seq_off[ln.rg_lo + i + 1] - R.seq_off[ln.rg_lo + i]);
			r.qlen2 = 0, r.gap_ref = in[i].gap_ref;
			if (R.has_pairs) {
				const int32_t u0 = R.unit_first[ln.rg_lo + i];
				if (R.unit_first[ln.rg_lo + i + 1] - u0 == 2) r.qlen = (int32_t)(R.unit_off[u0 + 1] - R.unit_off[u0]), r.qlen2 = (int32_t)(R.unit_off[u0 + 2] - R.unit_off[u0 + 1]);
			}
			r.hash = in[i].hash, r.src = skip ? RGN_SRC_SKIP : (c.dev_src == 1 ? RGN_SRC_LJ : 0);
			if (!skip) {
				const uint64_t k = r.qlen2 > 0 ? 2 : 1; // (a chain can have anchors on both segments: each segment's slice is as long as the fragment's chained anchors)
				sq += k * (uint64_t)c.n_a, n_chain += k * (uint64_t)c.n_u, max_nu = std::max(max_nu, (int)c.n_u);
			}
		}
		if (n_chain >= (1ull << 31)) throw std::runtime_error("[mm2amd] align_regions: a sub-batch with more than 2^31 chains");
		const uint32_t max_regs = (uint32_t)n_chain;
		const uint64_t max_jobs64 = std::max<uint64_t>(sq + 2 * n_chain + 1, 3 * n_chain + 1); // (one window per anchor at most, two extensions; short reads: three pieces per hit)
		if (max_jobs64 >= (1ull << 31)) throw std::runtime_error("[mm2amd] align_regions: a sub-batch with more than 2^31 anchors");
		RgnBuffers B{};
		B.n_reads = (int)n;
		B.rout_stride = stride;
		ln.d_rg_reads.ensure(n), ln.d_rg_rout.ensure(n * (size_t)stride), ln.d_rg_cur.ensure(RGN_CUR_N);
		ln.d_rg_sq.ensure(sq + 1), ln.d_rg_sites.ensure(sq + 1);
		ln.d_rg_regs.ensure(max_regs + 1), ln.d_rg_aux.ensure(max_regs + 1), ln.d_rg_plan.ensure(max_regs + 1), ln.d_rg_fin.ensure(max_regs + 1), ln.d_rg_finres.ensure(max_regs + 1);
		ln.d_rg_win.ensure(max_jobs64), ln.d_rg_jobs.ensure(max_jobs64), ln.d_rg_pieces.ensure(max_jobs64);
		HIP_CHECK(hipMemcpyAsync(ln.d_rg_reads.p, hr, n * sizeof(RgnRead), hipMemcpyHostToDevice, st));
		HIP_CHECK(hipMemsetAsync(ln.d_rg_cur.p, 0, RGN_CUR_N * sizeof(unsigned int), st));
		B.reads = ln.d_rg_reads.p;
		B.a_src[0] = ln.d_bt_out_a.p, B.a_src[1] = ln.d_lj_out_a.p, B.u_src[0] = ln.d_bt_out_u.p, B.u_src[1] = ln.d_lj_out_u.p;
		B.mini_pos = ln.d_minipos.p, B.sq_a = ln.d_rg_sq.p, B.regs = ln.d_rg_regs.p, B.aux = ln.d_rg_aux.p, B.rout = ln.d_rg_rout.p, B.cursors = ln.d_rg_cur.p;
		B.ref_len = d_ref_len2_.p, B.ref_off = d_ref_off_.p, B.max_regs = max_regs;
		B.qpool = R.d_qpool.p, B.S = T_->S.p;
		B.lds_chains = std::min(256, (max_nu + 63) / 64 * 64);
		B.plan = ln.d_rg_plan.p, B.win = ln.d_rg_win.p, B.jobs = ln.d_rg_jobs.p, B.gap_sites = ln.d_rg_sites.p, B.max_jobs = (uint32_t)max_jobs64;
		B.fin = ln.d_rg_fin.p, B.pieces = ln.d_rg_pieces.p;
		kp.begin(st); launch_chain_regs(B, O, st); kp.end(st, "chain_regs_kernel", 32.0 * (double)sq + 80.0 * (double)n_chain);
		kp.begin(st); launch_region_plan(B, O, st); kp.end(st, "region_plan_kernel", 16.0 * (double)sq);
		unsigned int *cur = ln.h_rg_cur.ensure(RGN_CUR_N);
		HIP_CHECK(hipMemcpyAsync(cur, ln.d_rg_cur.p, RGN_CUR_N * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
		stream_wait(st);
		const uint32_t n_regs = std::min<uint32_t>(cur[RGN_CUR_REGS], max_regs);
		const size_t n_jobs = std::min<size_t>(cur[RGN_CUR_JOBS], (size_t)max_jobs64);
		Trace::get().add(lane_id, "gpu:regs+plan", tt, Trace::now()); tt = Trace::now();
		const uint32_t *d_cigar = nullptr;
		size_t n_cig = 0;
		if (n_jobs) { // the DP: the job records stay where region_plan_kernel wrote them; launch classes, order and sizes are made on the device (ksw_order.hip)
			ln.ksw.n_threads = n_threads, ln.ksw.prof = &kp, ln.ksw.lane = lane_id;
			size_t dir_gb = 160;
			if (const char *e = getenv("MM2AMD_DIR_BUDGET_GB")) dir_gb = atol(e) > 0 ? (size_t)atol(e) : dir_gb;
			ln.ksw.dir_budget = std::min<size_t>((dir_gb << 30) / (size_t)active_lanes_, (size_t)96 << 30);
			static const bool host_order = getenv("MM2AMD_KSW_ORDER_ON_HOST") != nullptr; // A/B checks: the job records cross for the host's ordering
			if (host_order) {
				KswJob *hj = ln.h_rg_jobs.ensure(n_jobs);
				HIP_CHECK(hipMemcpyAsync(hj, ln.d_rg_jobs.p, n_jobs * sizeof(KswJob), hipMemcpyDeviceToHost, st));
				stream_wait(st);
				for (size_t j = 0; j < n_jobs; ++j) out.dp_cells += (double)hj[j].qlen * hj[j].tlen;
				ln.ksw.run_jobs(hj, n_jobs, R.d_qpool.p, nullptr, T_->S.p, sc, nullptr, &d_cigar, &n_cig, st);
			} else {
				{
					DpGate gate(*this); // at most MM2AMD_DP_GATE lanes (default 4) in their DP launches at once
					ln.ksw.run_jobs(nullptr, n_jobs, R.d_qpool.p, nullptr, T_->S.p, sc, nullptr, &d_cigar, &n_cig, st, ln.d_rg_jobs.p);
				}
				out.dp_cells += ln.ksw.last_cells;
			}
		}
		tt = Trace::now();
		B.res = ln.ksw.d_res.p, B.perm = ln.ksw.d_perm.p;
		kp.begin(st); launch_region_consume(B, O, d_cigar, st); kp.end(st, "region_consume_kernel", 68.0 * (double)n_jobs);
		HIP_CHECK(hipMemcpyAsync(cur, ln.d_rg_cur.p, RGN_CUR_N * sizeof(unsigned int), hipMemcpyDeviceToHost, st));
		stream_wait(st);
		const size_t out_words = cur[RGN_CUR_OUT];
		ln.d_rg_out.ensure(out_words + 1);
		if (n_regs && cur[RGN_CUR_N_FIN]) {
			FinParams P;
			P.regions = ln.d_rg_fin.p, P.n_regions = (int)n_regs, P.pieces = ln.d_rg_pieces.p, P.cigar_pool = d_cigar, P.out_pool = ln.d_rg_out.p, P.results = ln.d_rg_finres.p;
			P.qpool = R.d_qpool.p, P.S = T_->S.p;
			memcpy(P.mat, sc.mat, 25);
			P.q = sc.q, P.e = sc.e, P.log_gap = log_gap ? 1 : 0;
			P.cap_ops = (int)std::min<uint32_t>((std::max<uint32_t>(cur[RGN_CUR_MAX_OPS], 1u) + 63) & ~63u, (uint32_t)kFinMaxOps);
			kp.begin(st);
			region_finish_launch(P, st);
			kp.end(st, "region_finish_kernel", (double)out_words * 8.0 + (double)cur[RGN_CUR_N_FIN] * (sizeof(FinRegion) + sizeof(FinResult)), (double)cur[RGN_CUR_N_FIN]);
		}
		RgnReadOut *h_rout = ln.h_rg_rout.ensure(n * (size_t)stride);
		ref::Reg1 *h_regs = ln.h_rg_regs.ensure(n_regs + 1);
		RgnAux *h_aux = ln.h_rg_aux.ensure(n_regs + 1);
		RgnPlan *h_plan = ln.h_rg_plan.ensure(n_regs + 1);
		FinRegion *h_fin = ln.h_rg_fin.ensure(n_regs + 1);
		FinResult *h_finres = ln.h_rg_finres.ensure(n_regs + 1);
		uint32_t *h_out = ln.h_rg_out.ensure(out_words + 1);
		HIP_CHECK(hipMemcpyAsync(h_rout, ln.d_rg_rout.p, n * (size_t)stride * sizeof(RgnReadOut), hipMemcpyDeviceToHost, st));
		if (n_regs) {
			HIP_CHECK(hipMemcpyAsync(h_regs, ln.d_rg_regs.p, n_regs * sizeof(ref::Reg1), hipMemcpyDeviceToHost, st));
			HIP_CHECK(hipMemcpyAsync(h_aux, ln.d_rg_aux.p, n_regs * sizeof(RgnAux), hipMemcpyDeviceToHost, st));
			HIP_CHECK(hipMemcpyAsync(h_plan, ln.d_rg_plan.p, n_regs * sizeof(RgnPlan), hipMemcpyDeviceToHost, st));
			HIP_CHECK(hipMemcpyAsync(h_fin, ln.d_rg_fin.p, n_regs * sizeof(FinRegion), hipMemcpyDeviceToHost, st));
			HIP_CHECK(hipMemcpyAsync(h_finres, ln.d_rg_finres.p, n_regs * sizeof(FinResult), hipMemcpyDeviceToHost, st));
		}
		if (out_words) HIP_CHECK(hipMemcpyAsync(h_out, ln.d_rg_out.p, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		stream_wait(st);
		Trace::get().add(lane_id, "gpu:consume+finish, d2h:hits", tt, Trace::now());
		kp.collect();
		out.reads = h_rout, out.regs = h_regs, out.aux = h_aux, out.plan = h_plan, out.fin = h_fin, out.fin_res = h_finres, out.cigars = h_out;
		out.n_regs = n_regs, out.n_jobs = n_jobs, out.rout_stride = stride;
	}

	// MM2AMD_DEVICE_FINISH=1 / =0 decides; unset: the device finishes the regions when this mapper has fewer than 12 host threads.  The kernel saves
	// 1.4 host core-seconds per 1-Gbase step (8.2 -> 6.8) for 56 ms of latency-bound launches: with 16 threads for one GPU the pipeline comes out
	// 1 % slower with it (1.486 / 1.487 against 1.504 / 1.499 Gbases/s, A/B on one box: profiles/README.md); a rank of a multi-GPU job that shares
	// the node's CPU quota with seven others (two threads each under a 16-CPU quota) is bounded by exactly those core-seconds.  Read per batch.
	bool supports_long_join() const override { return true; }
	bool finishes_regions() const override
	{
		// Round 4: on by default.  The kernel costs 18 ms per 1-Gbase step (round 3: 56) and takes 1.5 host core-seconds per step off the host (5.5 -> 3.9):
		// at 16 threads the pipeline's rate is the same either way, with fewer it is higher.  MM2AMD_DEVICE_FINISH=0: the host's mm_update_extra (A/B checks).
		const char *e = getenv("MM2AMD_DEVICE_FINISH");
		if (e && *e) return *e != '0';
		return true;
	}
	void finish_regions(int lane_id, const std::vector<FinRegion> &regions, const std::vector<FinPiece> &pieces, size_t out_words, const int8_t *mat25, int q, int e, bool log_gap,
	                    std::vector<FinResult> &results, const uint32_t **cigars) override
	{
		HIP_CHECK(hipSetDevice(dev_));
		Lane &ln = *lanes_.at(lane_id);
		const size_t n = regions.size();
		results.resize(n);
		*cigars = nullptr;
		if (n == 0) return;
		memcpy(ln.h_fin_regions.ensure(n), regions.data(), n * sizeof(FinRegion));
		memcpy(ln.h_fin_pieces.ensure(pieces.size() + 1), pieces.data(), pieces.size() * sizeof(FinPiece));
		ln.d_fin_regions.ensure(n), ln.d_fin_pieces.ensure(pieces.size() + 1), ln.d_fin_out.ensure(out_words + 1), ln.d_fin_res.ensure(n);
		ln.h_fin_out.ensure(out_words + 1), ln.h_fin_res.ensure(n);
		HIP_CHECK(hipMemcpyAsync(ln.d_fin_regions.p, ln.h_fin_regions.p, n * sizeof(FinRegion), hipMemcpyHostToDevice, ln.stream));
		HIP_CHECK(hipMemcpyAsync(ln.d_fin_pieces.p, ln.h_fin_pieces.p, pieces.size() * sizeof(FinPiece), hipMemcpyHostToDevice, ln.stream));
		FinParams P;
		P.regions = ln.d_fin_regions.p, P.n_regions = (int)n, P.pieces = ln.d_fin_pieces.p, P.cigar_pool = ln.ksw.d_cigar.p, P.out_pool = ln.d_fin_out.p, P.results = ln.d_fin_res.p;
		P.qpool = res_[ln.set].d_qpool.p, P.S = T_->S.p;
		memcpy(P.mat, mat25, 25);
		P.q = (int8_t)q, P.e = (int8_t)e, P.log_gap = log_gap ? 1 : 0;
		uint32_t longest = 1; // (a region's room in the output pool is the sum of its pieces: the next region's offset minus its own)
		for (size_t i = 0; i < n; ++i) longest = std::max<uint32_t>(longest, (uint32_t)((i + 1 < n ? regions[i + 1].out_off : (uint32_t)out_words) - regions[i].out_off));
		P.cap_ops = (int)std::min<uint32_t>((longest + 63) & ~63u, (uint32_t)kFinMaxOps);
		KernelProfiler &prof = kernel_profiler(lane_id, replica_);
		prof.begin(ln.stream);
		region_finish_launch(P, ln.stream);
		prof.end(ln.stream, "region_finish_kernel", (double)out_words * 8.0 + (double)n * (sizeof(FinRegion) + sizeof(FinResult)), (double)n);
		HIP_CHECK(hipMemcpyAsync(ln.h_fin_res.p, ln.d_fin_res.p, n * sizeof(FinResult), hipMemcpyDeviceToHost, ln.stream));
		HIP_CHECK(hipMemcpyAsync(ln.h_fin_out.p, ln.d_fin_out.p, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, ln.stream));
		stream_wait(ln.stream);
		memcpy(results.data(), ln.h_fin_res.p, n * sizeof(FinResult));
		*cigars = ln.h_fin_out.p;
		static const bool check = getenv("MM2AMD_FIN_CHECK") != nullptr; // diagnostic: every returned CIGAR must cover its windows
		if (check)
			for (size_t i = 0; i < n; ++i) {
				const FinResult &f = results[i];
				long q = f.qshift, t = f.tshift;
				for (int k = 0; k < f.n_cigar; ++k) {
					const uint32_t c = ln.h_fin_out.p[regions[i].out_off + k], op = c & 0xf, len = c >> 4;
					if (op == 0 || op == 7 || op == 8) q += len, t += len;
					else if (op == 1) q += len;
					else if (op == 2 || op == 3) t += len;
				}
				if (f.n_cigar < 0 || q != regions[i].q_len || t != regions[i].t_len)
					fprintf(stderr, "[mm2amd] FIN_CHECK lane %d region %zu of %zu: n_cigar %d qshift %d tshift %d covers %ld/%ld of %d/%d (pieces %u from %u, out_off %u)\n", lane_id, i, n,
					        f.n_cigar, f.qshift, f.tshift, q, t, regions[i].q_len, regions[i].t_len, regions[i].n_pieces, regions[i].piece0, regions[i].out_off);
			}
		prof.collect();
	}

private:
	// At most dp_gate_ lanes between the ordering of their DP batch and its last kernel (0: no limit).  The DP kernels are VALU-bound persistent launches: eight that
	// coincide share the SIMDs round-robin and ALL finish late; admitted four at a time the first ones finish early and their lanes go on to consume, finish and the next
	// sub-batch's seeding while the others compute.  Measured (calls 23-24, 8-step runs in one call each): no gate 2.24-2.28 Gbases/s with 47-52 ms per streaming
	// launch, four lanes 2.30 / 2.30 with 29-34 ms, two lanes 2.22-2.28 with 18-21 ms, one lane 2.08.  (Round 2's stage gates serialised ALL GPU stages of the lanes
	// and lost 7-18 %; this one leaves seeding, chaining and the region kernels free to run beside the DP.)
	struct DpGate {
		HipBackend &b;
		explicit DpGate(HipBackend &be) : b(be) { if (b.dp_gate_ > 0) { std::unique_lock<std::mutex> lk(b.gate_mu_); b.gate_cv_.wait(lk, [&] { return b.gate_in_ < b.dp_gate_; }); ++b.gate_in_; } }
		~DpGate() { if (b.dp_gate_ > 0) { { std::lock_guard<std::mutex> lk(b.gate_mu_); --b.gate_in_; } b.gate_cv_.notify_one(); } }
	};
	int dp_gate_ = getenv("MM2AMD_DP_GATE") ? atoi(getenv("MM2AMD_DP_GATE")) : 4, gate_in_ = 0;
	std::mutex gate_mu_;
	std::condition_variable gate_cv_;
	hipStream_t stream_ = nullptr;
	int n_threads_ = 1, n_cu_ = 256, n_lanes_ = 1, rid_bits_ = 1, dev_ = 0, replica_ = 0;
	std::atomic<int> active_lanes_{1};
	DevIndex I_{};
	DeviceIndexTables own_;
	DeviceIndexTables *T_ = nullptr;
	std::vector<std::unique_ptr<Lane>> lanes_;
	// The resident batch (shared by all lanes, read-only during run).  Two sets: begin_batch() fills the one that is NOT being mapped, so the
	// hand-over of batch k+1 (pack into pinned memory, H2D) runs beside the mapping of batch k; activate_batch() swaps them.
	struct Resident {
		DevBuf<uint8_t> d_qpool;
		DevBuf<uint64_t> d_seq_off;
		PinBuf<char> h_ascii;
		std::vector<uint64_t> seq_off;
		// pairs: unit = one read of a pair (or a single read); see begin_batch
		bool has_pairs = false;
		size_t n_units = 0;
		std::vector<uint64_t> unit_off;
		std::vector<int32_t> unit_first;
		DevBuf<uint64_t> d_unit_off;
		DevBuf<int32_t> d_unit_first;
		// all-vs-all name rules: two ranks per read (see skip_hit in seed_chain.hip)
		bool have_read_names = false;
		std::vector<int32_t> name_key;
		DevBuf<int32_t> d_name_key;
		std::vector<const char *> read_names; // MM2AMD_SEED_DUMP only
	};
	Resident res_[2];
	int cur_ = 0;
	hipStream_t stage_stream_ = nullptr; // the hand-over's copies (the device's shared stream serves set-up only)
	// all-vs-all name rules
	bool name_rules_ = false;
	const std::vector<std::string> *fi_names_ = nullptr;
	const std::vector<uint32_t> *fi_seq_len_ = nullptr;
	const std::vector<uint64_t> *fi_seq_off_ = nullptr;
	DevBuf<uint64_t> d_ref_off_;
	DevBuf<uint32_t> d_ref_len2_;
	std::mutex ref_mu_;
	std::atomic<bool> ref_ready_{false};
	std::vector<std::string> sorted_names_;
	DevBuf<int32_t> d_name_rank_;
	DevBuf<uint32_t> d_ref_len_;
};

} // namespace

Backend *make_backend(const FlatIndex &fi, void *device_tables, int n_threads, int device, int replica, int tables_device)
{
	return new HipBackend(fi, (DeviceIndexTables *)device_tables, n_threads, device, replica, tables_device);
}
void apply_hw_queue_default(); // capi_common.cpp
int backend_device_count() { apply_hw_queue_default(); int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

void *backend_build_index_tables(FlatIndex &fi, int device, int *on_device)
{
	DeviceCtx &d = device_ctx(device);
	std::lock_guard<std::mutex> lk(d.mu);
	ensure_device(d);
	std::unique_ptr<DeviceIndexTables> T(new DeviceIndexTables);
	DeviceIndexBuilder::build_from_packed(fi, *T, d.stream);
	if (on_device) *on_device = d.device_id;
	return T.release();
}
void backend_free_index_tables(void *tables) { delete (DeviceIndexTables *)tables; }
const char *backend_name() { return "hip:gfx950"; }

} // namespace mm2amd

// The product's Backend: hand-written gfx950 kernels (seed_chain.hip, ksw_extd2.hip) plus the buffer management
// around them.  There is deliberately no CPU path here: constructing it without a usable HIP device throws.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include "backend.hpp"
#include "chain_host.hpp"
#include "device_ctx.hpp"
#include "ksw_host.hpp"
#include "seed_chain_dev.hpp"
#include "index_build.hpp"
#include "kernel_prof.hpp"
#include "threads.hpp"
#include <thread>

namespace mm2amd {

namespace {

template <typename T>
void upload(DevBuf<T> &d, const std::vector<T> &h, hipStream_t s)
{
	d.ensure(h.size() ? h.size() : 1, 1.0);
	if (!h.empty()) HIP_CHECK(hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

class HipBackend : public Backend {
public:
	// `tables` may be null: the backend then mirrors the host tables of `fi` (an index flattened from a reference mm_idx_t)
	HipBackend(const FlatIndex &fi, DeviceIndexTables *tables) : fi_(fi), T_(tables)
	{
		DeviceCtx &d = device_ctx();
		std::lock_guard<std::mutex> lk(d.mu);
		ensure_device(d);
		stream_ = d.stream;
		ksw_.n_cu = d.n_cu;
		n_threads_ = std::max(1u, std::thread::hardware_concurrency());
		if (!T_) { own_.upload(fi, stream_); T_ = &own_; }
		I_.bucket_start = T_->bucket_start.p, I_.keys = T_->keys.p, I_.val_off = T_->val_off.p, I_.pos = T_->pos.p, I_.S = T_->S.p;
		I_.bucket_bits = T_->bucket_bits, I_.key_shift = T_->key_shift;
		while ((1ull << rid_bits_) < fi.n_seq) ++rid_bits_;
	}

	long max_reads_per_call() const override { return 1L << (31 - rid_bits_); } // the anchor sort's composite key: read | strand | rid | rpos in 64 bits

	void begin_batch(const std::vector<ReadView> &reads, std::vector<uint64_t> &qpool_off) override
	{
		const size_t n = reads.size();
		n_reads_ = (int)n;
		seq_off_.resize(n + 1);
		seq_off_[0] = 0;
		for (size_t i = 0; i < n; ++i) seq_off_[i + 1] = seq_off_[i] + (uint64_t)reads[i].len;
		const uint64_t total = seq_off_[n];
		qpool_off.resize(n);
		for (size_t i = 0; i < n; ++i) qpool_off[i] = 2 * seq_off_[i];
		char *h = h_ascii_.ensure(total + 1);
		parallel_for(n_threads_, (long)n, [&](long i, int) { memcpy(h + seq_off_[i], reads[i].seq, reads[i].len); }, 64);
		d_ascii_.ensure(total + 1);
		d_qpool_.ensure(2 * total + 16);
		d_seq_off_.ensure(n + 1);
		HIP_CHECK(hipMemcpyAsync(d_ascii_.p, h, total, hipMemcpyHostToDevice, stream_));
		HIP_CHECK(hipMemcpyAsync(d_seq_off_.p, seq_off_.data(), (n + 1) * 8, hipMemcpyHostToDevice, stream_));
		B_ = SeedChainBuffers();
		B_.n_reads = n_reads_, B_.seq_off = d_seq_off_.p, B_.ascii = d_ascii_.p, B_.qpool = d_qpool_.p;
		HIP_CHECK(hipStreamSynchronize(stream_)); // the batch is resident; everything after this is the hot path
	}

	void seed_chain(const SeedChainParams &P, long lo, long hi, std::vector<ReadChains> &out) override
	{
		const size_t n = (size_t)(hi - lo);
		B_.n_reads = (int)n, B_.seq_off = d_seq_off_.p + lo;
		out.clear();
		out.resize(n);
		if (n == 0) return;
		if (P.flag & (ref::F_FOR_ONLY | ref::F_REV_ONLY)) throw std::invalid_argument("[mm2amd] --for-only/--rev-only are not implemented on the device path");
		KernelProfiler &kp = kernel_profiler();
		const double L = (double)(seq_off_[hi] - seq_off_[lo]);
		kp.begin(stream_); launch_encode(B_, stream_); kp.end(stream_, "encode_kernel", 3 * L);
		// 1. minimizers, written from slot seq_off[r] of the minimizer arrays (at most one per base)
		const uint64_t base0 = seq_off_[lo], cap_mz = seq_off_[hi] - base0;
		d_mz_cnt_.ensure(n);
		d_mz_x_.ensure(cap_mz + 1), d_mz_y_.ensure(cap_mz + 1);
		d_sd_n_.ensure(cap_mz + 1), d_sd_off_.ensure(cap_mz + 1), d_sd_aoff_.ensure(cap_mz + 1), d_sd_qpos_.ensure(cap_mz + 1), d_sd_info_.ensure(cap_mz + 1);
		// the per-read slot offsets are the batch-wide base offsets; shifting the array bases by the sub-batch's first offset makes them local
		B_.mz_cnt = d_mz_cnt_.p, B_.mz_off = B_.seq_off, B_.mz_x = d_mz_x_.p - base0, B_.mz_y = d_mz_y_.p - base0;
		B_.sd_n = d_sd_n_.p - base0, B_.sd_off = d_sd_off_.p - base0, B_.sd_aoff = d_sd_aoff_.p - base0, B_.sd_qpos = d_sd_qpos_.p - base0, B_.sd_info = d_sd_info_.p - base0;
		kp.begin(stream_); launch_sketch(B_, P, stream_); kp.end(stream_, "sketch_kernel", L + 16.0 * (2.0 * L / (P.w + 1)));
		// 2. seeds: probe, filter, count anchors
		d_n_anchor_.ensure(n), d_n_minipos_.ensure(n), d_n_seedhit_.ensure(n), d_rep_len_.ensure(n);
		B_.n_anchor = d_n_anchor_.p, B_.n_minipos = d_n_minipos_.p, B_.n_seedhit = d_n_seedhit_.p, B_.rep_len = d_rep_len_.p;
		kp.begin(stream_); launch_seed_collect(B_, I_, P, stream_); kp.end(stream_, "seed_collect_kernel", 36.0 * (2.0 * L / (P.w + 1))); // per minimizer: 16 B record + 8 key + 8 val + 4 flags
		h_na_.resize(n), h_nmp_.resize(n), h_rep_.resize(n);
		HIP_CHECK(hipMemcpyAsync(h_na_.data(), d_n_anchor_.p, n * 4, hipMemcpyDeviceToHost, stream_));
		HIP_CHECK(hipMemcpyAsync(h_nmp_.data(), d_n_minipos_.p, n * 4, hipMemcpyDeviceToHost, stream_));
		HIP_CHECK(hipMemcpyAsync(h_rep_.data(), d_rep_len_.p, n * 4, hipMemcpyDeviceToHost, stream_));
		HIP_CHECK(hipStreamSynchronize(stream_));
		a_off_.resize(n + 1), mp_off_.resize(n + 1);
		a_off_[0] = mp_off_[0] = 0;
		for (size_t i = 0; i < n; ++i) a_off_[i + 1] = a_off_[i] + h_na_[i], mp_off_[i + 1] = mp_off_[i] + h_nmp_[i];
		const uint64_t n_a = a_off_[n], n_mp = mp_off_[n];
		d_a_off_.ensure(n + 1), d_mp_off_.ensure(n + 1);
		HIP_CHECK(hipMemcpyAsync(d_a_off_.p, a_off_.data(), (n + 1) * 8, hipMemcpyHostToDevice, stream_));
		HIP_CHECK(hipMemcpyAsync(d_mp_off_.p, mp_off_.data(), (n + 1) * 8, hipMemcpyHostToDevice, stream_));
		d_anchors_.ensure(n_a + 1), d_minipos_.ensure(n_mp + 1), d_f_.ensure(n_a + 1), d_p_.ensure(n_a + 1), d_t_.ensure(n_a + 1);
		d_skey_in_.ensure(n_a + 1), d_sval_in_.ensure(n_a + 1), d_skey_out_.ensure(n_a + 1), d_sval_out_.ensure(n_a + 1), d_tie_.ensure(n);
		B_.sort_key_in = d_skey_in_.p, B_.sort_val_in = d_sval_in_.p, B_.sort_key_out = d_skey_out_.p, B_.sort_val_out = d_sval_out_.p, B_.tie_flag = d_tie_.p;
		B_.rid_bits = rid_bits_;
		int read_bits = 1;
		while ((1ull << read_bits) < n) ++read_bits;
		const int end_bit = 33 + rid_bits_ + read_bits;
		const size_t sort_tmp = anchor_sort_temp_bytes(n_a, end_bit);
		d_sort_tmp_.ensure(sort_tmp + 16);
		B_.a_off = d_a_off_.p, B_.mp_off = d_mp_off_.p, B_.anchors = d_anchors_.p, B_.mini_pos = d_minipos_.p;
		B_.f = d_f_.p, B_.p = d_p_.p, B_.t = d_t_.p;
		// 3. anchors: expand, sort, chain
		kp.begin(stream_); launch_seed_expand(B_, I_, P, stream_); kp.end(stream_, "seed_expand_kernel", 24.0 * n_a);
		kp.begin(stream_); launch_anchor_sort(B_, n_a, end_bit, d_sort_tmp_.p, sort_tmp, stream_); kp.end(stream_, "anchor_sort", 32.0 * n_a);
		kp.begin(stream_); launch_chain_fill(B_, P, stream_); kp.end(stream_, "chain_fill_kernel", 24.0 * n_a);
		// 4. back to the host for the (scalar, order-sensitive) backtrack
		Anchor *ha = h_anchors_.ensure(n_a + 1);
		int32_t *hf = h_f_.ensure(n_a + 1), *hp = h_p_.ensure(n_a + 1);
		uint64_t *hmp = h_minipos_.ensure(n_mp + 1);
		if (n_a) {
			HIP_CHECK(hipMemcpyAsync(ha, d_anchors_.p, n_a * sizeof(Anchor), hipMemcpyDeviceToHost, stream_));
			HIP_CHECK(hipMemcpyAsync(hf, d_f_.p, n_a * 4, hipMemcpyDeviceToHost, stream_));
			HIP_CHECK(hipMemcpyAsync(hp, d_p_.p, n_a * 4, hipMemcpyDeviceToHost, stream_));
		}
		if (n_mp) HIP_CHECK(hipMemcpyAsync(hmp, d_minipos_.p, n_mp * 8, hipMemcpyDeviceToHost, stream_));
		HIP_CHECK(hipStreamSynchronize(stream_));
		kp.collect();
		const int max_drop = P.is_cdna ? INT32_MAX : P.bw;
		std::vector<ChainScratch> scratch(n_threads_);
		parallel_for(n_threads_, (long)n, [&](long i, int tid) {
			ReadChains &c = out[i];
			c.rep_len = h_rep_[i];
			c.mini_pos.assign(hmp + mp_off_[i], hmp + mp_off_[i + 1]);
			const int64_t na = (int64_t)(a_off_[i + 1] - a_off_[i]);
			chain_backtrack_compact(na, ha + a_off_[i], hf + a_off_[i], hp + a_off_[i], P.min_cnt, P.min_chain_score, max_drop, c.u, c.a, scratch[tid]);
		});
	}

	void ksw(const std::vector<KswJob> &jobs, const KswScoring &sc, std::vector<KswRes> &res, const uint32_t **cigar) override
	{
		res.resize(jobs.size());
		size_t n_cig = 0;
		ksw_.n_threads = n_threads_;
		ksw_.run(jobs, d_qpool_.p, nullptr, T_->S.p, sc, res.data(), cigar, &n_cig, stream_);
		kernel_profiler().collect();
	}

private:
	const FlatIndex &fi_;
	hipStream_t stream_ = nullptr;
	int n_threads_ = 1, n_reads_ = 0;
	DevIndex I_{};
	SeedChainBuffers B_{};
	KswRunner ksw_;
	DeviceIndexTables own_;
	DeviceIndexTables *T_ = nullptr;
	DevBuf<char> d_ascii_;
	DevBuf<uint8_t> d_qpool_;
	DevBuf<uint64_t> d_seq_off_, d_mz_off_, d_a_off_, d_mp_off_, d_mz_x_, d_mz_y_, d_minipos_;
	DevBuf<uint32_t> d_mz_cnt_, d_sd_n_, d_sd_off_, d_sd_aoff_, d_sd_qpos_, d_sd_info_, d_n_anchor_, d_n_minipos_, d_n_seedhit_;
	DevBuf<int32_t> d_rep_len_, d_f_, d_p_, d_t_;
	DevBuf<Anchor> d_anchors_;
	DevBuf<uint64_t> d_skey_in_, d_sval_in_, d_skey_out_, d_sval_out_;
	DevBuf<uint32_t> d_tie_;
	DevBuf<uint8_t> d_sort_tmp_;
	int rid_bits_ = 1;
	PinBuf<char> h_ascii_;
	PinBuf<Anchor> h_anchors_;
	PinBuf<int32_t> h_f_, h_p_;
	PinBuf<uint64_t> h_minipos_;
	std::vector<uint64_t> seq_off_, mz_off_, a_off_, mp_off_;
	std::vector<uint32_t> h_cnt_, h_na_, h_nmp_;
	std::vector<int32_t> h_rep_;
};

} // namespace

Backend *make_backend(const FlatIndex &fi, void *device_tables) { return new HipBackend(fi, (DeviceIndexTables *)device_tables); }
const char *backend_name() { return "hip:gfx950"; }

} // namespace mm2amd

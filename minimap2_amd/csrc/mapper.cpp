#include <chrono>
#include <ctime>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <malloc.h>
#include <stdexcept>
#include <new>
#include "mapper.hpp"
#include "host_prof.hpp"
#include "chain_host.hpp"
#include "threads.hpp"
#include "trace.hpp"
#include <atomic>
#include <cstdio>
#include <mutex>
#include <thread>

namespace mm2amd {

using namespace ref;

namespace {
inline uint32_t wang_hash(uint32_t key) // __ac_Wang_hash, khash.h:400-409
{
	key += ~(key << 15); key ^= (key >> 10); key += (key << 3); key ^= (key >> 6); key += ~(key << 11); key ^= (key >> 16);
	return key;
}
inline uint32_t x31_hash(const char *s) // __ac_X31_hash_string, khash.h:383-388
{
	uint32_t h = (uint32_t)*s;
	if (h) for (++s; *s; ++s) h = (h << 5) - h + (uint32_t)*s;
	return h;
}
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
double cpu_now() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
double thr_now() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
}

// Backend::align_regions takes single-segment reads of the plain long-read configurations: dual- or single-affine DNA alignment with CIGARs.  What it leaves
// to the host path -- each a rule the device kernels do not carry: spliced and short-read alignment (two strands per region, ungapped shortcuts, composed
// targets), query-strand coordinates, =/X CIGARs, chains reported as hits, ALT contigs, junction / jump annotation, and the re-seeding of reads without chains
// (-f x,y: map.c:293-316).
bool region_path_supported(const MapOpt &opt, int idx_flag, int n_alt, bool has_annotation)
{
	if (!(opt.flag & F_CIGAR)) return false;
	if (opt.flag & (F_SPLICE | F_SR_RNA | F_QSTRAND | F_EQX | F_ALL_CHAINS)) return false;
	if (n_alt > 0 || has_annotation) return false;
	(void)idx_flag; // (round 5, late: homopolymer-compressed indices take the device path too -- region_plan_kernel's window boundaries, the summed minimizer spans)
	// (round 6: short reads and their pairs take the device path -- the best diagonal run, the ungapped window, the fragment's chains cut per segment)
	if (opt.max_occ > opt.mid_occ && !(opt.flag & (F_RMQ | F_SR))) return false; // (the second seeding of map.c:293-316: process_sub keeps the reads concerned on the host; only the short-read path is arranged for it)
	if (opt.split_prefix) return false;
	return true;
}

uint32_t read_hash(const char *qname, int qlen, const MapOpt &opt)
{
	uint32_t h = qname && !(opt.flag & F_NO_HASH_NAME) ? x31_hash(qname) : 0;
	h ^= wang_hash((uint32_t)qlen) + wang_hash((uint32_t)opt.seed);
	return wang_hash(h);
}

Mapper::Mapper(const FlatIndex &fi, const MapOpt &opt, Backend &be, int n_threads) : fi_(fi), opt_(opt), be_(be), n_threads_(n_threads < 1 ? 1 : n_threads)
{
	{ // one scratch object per lane the backend has, made before any lane thread exists
		const size_t cap = (size_t)std::max(16, be.n_lanes());
		scratch_.reserve(cap), drivers_.reserve(cap);
		for (int i = 0; i < std::max(1, be.n_lanes()); ++i) scratch_.emplace_back(new DriverScratch);
	}
	{ // what the device's chains -> hits -> windows path reads from the options (region_dev.hpp)
		RgnOpts &O = rgn_opts_;
		O.flag = opt.flag, O.max_sw_mat = opt.max_sw_mat, O.k = fi.k;
		O.mask_level = opt.mask_level, O.pri_ratio = opt.pri_ratio, O.mask_len = opt.mask_len, O.best_n = opt.best_n;
		O.sub_diff = opt.a * 2 + opt.b, O.min_strand_sc = (int)(opt.max_gap * 0.8);
		O.max_gap = opt.max_gap, O.min_cnt = opt.min_cnt, O.min_chain_score = opt.min_chain_score, O.bw = opt.bw;
		O.bw_ext = (int)(opt.bw * 1.5 + 1.), O.bw_gap = (int)(opt.bw_long * 1.5 + 1.);
		if (O.bw_gap < O.bw_ext) O.bw_gap = O.bw_ext;
		O.a = opt.a, O.b = opt.b, O.q = opt.q, O.e = opt.e, O.zdrop = opt.zdrop, O.zdrop_inv = opt.zdrop_inv, O.end_bonus = opt.end_bonus, O.min_ksw_len = opt.min_ksw_len;
		O.transition = opt.transition, O.hpc = (fi.flag & I_HPC) ? 1 : 0;
		O.sc_ambi = opt.sc_ambi;
		gen_score_matrix(opt, O.mat);
		rgn_ok_ = region_path_supported(opt, fi.flag, fi.n_alt, fi.has_junc || fi.has_jump || fi.has_spsc);
	}
	// MM_F_INDEPEND_SEG / MM_F_WEAK_PAIRING are resolved at the boundary (capi_map.cpp)
	if (opt.flag & F_QSTRAND) { // reverse-strand hits in query-strand coordinates: DP targets are composed (reverse-complemented) into the byte pool
		if (!be.supports_byte_targets()) throw std::invalid_argument("[mm2amd] --qstrand needs a backend with composed DP targets");
		be.enable_seq_len();
		if ((opt.flag & (F_OUT_SAM | F_SPLICE | F_FRAG_MODE | F_SR | F_HEAP_SORT)) || (fi.flag & I_HPC))
			throw std::invalid_argument("[mm2amd] --qstrand doesn't work with -a, -H, --frag, --sr, --heap-sort or --splice (options.c:271)");
	}
	if ((opt.flag & F_SR_RNA) && (opt.flag & F_SPLICE)) {
		if (!be.supports_byte_targets()) throw std::invalid_argument("[mm2amd] splice:sr needs a backend with composed DP targets");
		if (fi.has_junc || fi.has_spsc) throw std::invalid_argument("[mm2amd] splice:sr with --junc-bed / --spsc is not implemented (use -j)");
	}
	if ((opt.flag & F_SR) && (fi.flag & I_HPC)) throw std::invalid_argument("[mm2amd] short-read mode does not work with an HPC index (align.c:655)");
	if ((opt.flag & F_SPLICE) && fi.has_jump && (opt.flag & F_EQX)) throw std::invalid_argument("[mm2amd] jump annotation (-j) does not work with --eqx (jump.c:197)");
	if ((opt.flag & F_SPLICE) && (fi.has_junc || fi.has_spsc) && !be.supports_junctions())
		throw std::invalid_argument("[mm2amd] spliced alignment with junction annotation or splice scores (--junc-bed, --spsc) is not supported by this backend");
	if (opt.flag & (F_NO_DIAG | F_NO_DUAL)) be.enable_name_rules(); // all-vs-all: skip_seed compares read and target names (map.c:81-91)
	if ((opt.flag & F_CIGAR) && !fi.S) throw std::invalid_argument("[mm2amd] base-level alignment needs an index with sequence (MM_I_NO_SEQ is set)");
	if (opt.sdust_thres > 0 && !be.supports_sdust())
		throw std::invalid_argument("[mm2amd] SDUST masking (-T) is not supported by this backend");
	// The host stages allocate and free hundreds of MB of per-read records per sub-batch from hundreds of threads; letting glibc
	// hand that memory back to the kernel every time turns into page-fault and mmap-lock storms (the reference sidesteps the same
	// problem with its own kalloc arenas).  MM2AMD_MALLOPT=1 keeps freed memory in the process instead.  It is opt-in because it
	// changes malloc behaviour of the whole embedding process (bench.py and the drop-in driver set it).
	if (const char *e = getenv("MM2AMD_MALLOPT")) if (atoi(e) > 0) {
		mallopt(M_MMAP_THRESHOLD, 32 << 20);
		mallopt(M_TRIM_THRESHOLD, 1 << 30);
		mallopt(M_TOP_PAD, 64 << 20);
	}
}

void Mapper::stage(const std::vector<ReadView> &reads, bool may_start_early)
{
	{
		std::unique_lock<std::mutex> lk(mu_);
		cancel_next_locked(lk); // (a staged batch the lanes had started on is being replaced: not a pipeline's pattern, but it must not corrupt anything)
		pending_ = false, early_ok_ = false;
	}
	Staged &S = sets_[1 - cur_set_];
	S.n = (long)reads.size();
	S.live.clear(), S.live_id.clear(), S.qoff.clear();
	// which reads are mapped at all (map.c:243-244)
	for (long i = 0; i < S.n; ++i)
		if (reads[i].total() > 0 && !(opt_.max_qlen > 0 && reads[i].total() > opt_.max_qlen)) S.live.push_back(reads[i]), S.live_id.push_back(i);
	be_.begin_batch(S.live, S.qoff); // (an empty batch too: the backend's sets and ours swap together)
	static const bool no_early = getenv("MM2AMD_NO_EARLY_START") != nullptr; // A/B checks: every batch starts at its own run() call
	{
		std::lock_guard<std::mutex> lk(mu_);
		pending_ = true, early_ok_ = may_start_early && !no_early && be_.stages_beside_mapping();
	}
	cv_work_.notify_all(); // idle lanes of a batch on its last sub-batches may begin with this one
}

void Mapper::take_locked()
{
	cur_set_ = 1 - cur_set_;
	be_.activate_batch();
	pending_ = false;
}

void Mapper::take()
{
	std::lock_guard<std::mutex> lk(mu_);
	if (next_run_) { next_adopted_ = true; return; } // the lanes have taken it already; from here on it is the caller's batch (a hand-over that follows must not drop it)
	if (pending_) take_locked();
}

void Mapper::discard()
{
	std::unique_lock<std::mutex> lk(mu_);
	cancel_next_locked(lk);
	pending_ = false, early_ok_ = false;
}

// drops a batch the lanes started early: no further sub-batch of it is handed out, those under way are awaited
void Mapper::cancel_next_locked(std::unique_lock<std::mutex> &lk)
{
	if (!next_run_ || next_adopted_) return;
	next_run_->cancelled = true;
	std::shared_ptr<BatchRun> b = next_run_;
	cv_done_.wait(lk, [&] { return b->n_done == b->n_taken; });
	next_run_.reset();
	cur_set_ = 1 - cur_set_; // un-take: the set that was staged is the staged set again (and about to be refilled or forgotten)
	be_.activate_batch();
}

Mapper::~Mapper()
{
	{
		std::lock_guard<std::mutex> lk(mu_);
		stop_ = true;
	}
	cv_work_.notify_all();
	for (std::thread &t : drivers_) t.join();
}

// the batch in sets_[set] cut into sub-batches, with everything a lane needs to take one through its stages
std::shared_ptr<Mapper::BatchRun> Mapper::make_run(int set)
{
	std::shared_ptr<BatchRun> bp(new BatchRun);
	BatchRun &b = *bp;
	b.set = set, b.be_set = be_.current_set();
	b.out.resize(sets_[set].n);
	const std::vector<ReadView> &live = sets_[set].live;
	const long m_all = (long)live.size();

	// chaining parameters (map.c:262-274); single segment, not sr
	SeedChainParams &sp = b.sp;
	sp.k = fi_.k, sp.w = fi_.w, sp.is_hpc = fi_.flag & I_HPC;
	sp.sdust_thres = opt_.sdust_thres;
	sp.mid_occ = sp.q_mid_occ = opt_.mid_occ, sp.max_max_occ = opt_.max_max_occ, sp.occ_dist = opt_.occ_dist, sp.q_occ_frac = opt_.q_occ_frac;
	sp.flag = opt_.flag;
	sp.max_gap = opt_.max_gap, sp.max_gap_ref = opt_.max_gap_ref, sp.max_frag_len = opt_.max_frag_len, sp.is_sr = (opt_.flag & F_SR) ? 1 : 0; // per read: chain_gaps()
	sp.bw = opt_.bw, sp.max_chain_skip = opt_.max_chain_skip, sp.max_chain_iter = opt_.max_chain_iter;
	sp.min_cnt = opt_.min_cnt, sp.min_chain_score = opt_.min_chain_score;
	sp.chn_pen_gap = (float)(opt_.chain_gap_scale * 0.01 * fi_.k);
	sp.chn_pen_skip = (float)(opt_.chain_skip_scale * 0.01 * fi_.k);
	sp.is_cdna = (opt_.flag & F_SPLICE) ? 1 : 0; // map.c:230,280
	// map.c:275-277: mg_lchain_rmq instead of mg_lchain_dp -- on the device when the backend has the kernel (reads it hands back arrive
	// with their sorted anchors and ReadChains::chained unset), otherwise on the host over the device-sorted anchors
	sp.rmq = (opt_.flag & F_RMQ) && be_.supports_rmq() ? 1 : 0;
	sp.anchors_only = (opt_.flag & F_RMQ) && !sp.rmq ? 1 : 0;
	sp.rmq_inner_dist = opt_.rmq_inner_dist, sp.rmq_size_cap = opt_.rmq_size_cap;
	if (const char *e = getenv("MM2AMD_RMQ_DEV_MAX_ANCHORS")) sp.rmq_dev_max_anchors = atoi(e); // tests: force the hand-back of RMQ-preset reads
	// map.c:283-292: long-join re-chaining, by the backend when it can (reads it leaves alone are re-chained in process_sub)
	sp.long_join = opt_.bw_long > opt_.bw && (opt_.flag & (F_SPLICE | F_SR | F_NO_LJOIN)) == 0 && be_.supports_long_join() ? 1 : 0;
	sp.bw_long = opt_.bw_long, sp.rmq_rescue_size = opt_.rmq_rescue_size, sp.rmq_rescue_ratio = opt_.rmq_rescue_ratio;

	// Sub-batches bound the device working set (anchors and DP scratch scale with the number of reads in flight) and are the
	// unit of pipelining: each of the backend's lanes is driven by one host thread that takes the next sub-batch through all of
	// its stages, so the GPU stages of one sub-batch overlap the host stages of the others.
	long sub_bases = 100000000;
	if (const char *e = getenv("MM2AMD_SUBBATCH_BASES")) sub_bases = atol(e) > 0 ? atol(e) : sub_bases;
	std::vector<std::pair<long, long>> &subs = b.subs;
	if (m_all > 0) {
		long max_reads = be_.max_reads_per_call(), sub_reads = 25000; // bound the read count too, so that a second lane overlaps the host stages; larger sub-batches keep the DP launches' tails short
		uint64_t tot = 0;
		for (long i = 0; i < m_all; ++i) tot += (uint64_t)live[i].total();
		if (m_all > 0 && tot / (uint64_t)m_all < 1000) sub_reads = 400000; // Illumina-sized reads: 25 k of them are a few Mbases, far too little work per kernel launch -- let the base budget decide
		if (const char *e = getenv("MM2AMD_SUBBATCH_READS")) sub_reads = atol(e) > 0 ? atol(e) : sub_reads;
		if (sub_reads < max_reads) max_reads = sub_reads;
		if (m_all > 0 && tot / (uint64_t)m_all < 1000 && !getenv("MM2AMD_SUBBATCH_READS")) {
			// (round 6) short reads are cut by the read cap, not by bases: equal shares in whole rounds of the lanes (a million pairs were ten shares of 100 000 on
			// eight lanes -- a round of eight, then a round of two)
			// (measured on a million pairs, call 41: shares of 100 000 / 62 500 / 41 667 / 31 250 pairs: 0.695 / 0.676 / 0.714 / 0.738 Gbases/s -- more, smaller shares overlap better)
			if (rgn_ok_ && be_.aligns_regions() && max_reads > 40000) max_reads = 40000;
			const long lanes = std::max(1, be_.n_lanes()), rounds = (m_all + lanes * max_reads - 1) / (lanes * max_reads);
			max_reads = std::max<long>(1, (m_all + rounds * lanes - 1) / (rounds * lanes));
		}
		// equal shares instead of full sub-batches plus a remainder, and at least two of them when there is enough work to overlap
		// (a rank of an 8-GPU job gets an eighth of the batch: 125 Mbases map 6 % faster as 2 x 62 than as 100 + 25)
		long n_sub = (long)((tot + (uint64_t)sub_bases - 1) / (uint64_t)sub_bases);
		if (n_sub < 2 && tot >= 40000000 && !(opt_.flag & F_SPLICE)) n_sub = 2; // not for spliced reads: their DP launch classes need the whole batch's jobs to hide their tails
		// (round 5) with the per-read host stages on the device a sub-batch has no host work to hide, only kernel latencies to overlap: a batch of an eighth of
		// a Gbase -- a rank's share of an 8-GPU job -- maps 4 % faster as four sub-batches than as two (72 against 75 ms, profiles/r05_share_*_v18.json)
		static const long min_subs = getenv("MM2AMD_MIN_SUBBATCHES") ? atol(getenv("MM2AMD_MIN_SUBBATCHES")) : 4; // (A/B checks)
		if (n_sub < min_subs && tot >= 80000000 && rgn_ok_ && be_.aligns_regions()) n_sub = min_subs;
		if (n_sub > 0) sub_bases = (long)((tot + (uint64_t)n_sub - 1) / (uint64_t)n_sub);
		// (Staggering the first round's shares so that the lanes fall out of step was measured: 25 % slower.  The lanes' lockstep -- all
		// seeding, then all in the DP -- is the better regime while the DP kernels are persistent waves that fill every SIMD.)
		for (long lo = 0, hi; lo < m_all; lo = hi) {
			long bases = 0;
			for (hi = lo; hi < m_all && hi - lo < max_reads && bases < sub_bases; ++hi) bases += live[hi].total(); // the read that crosses the share's end still belongs to it
			subs.emplace_back(lo, hi);
		}
	}
	int n_drivers = (int)std::min<size_t>((size_t)std::max(1, be_.n_lanes()), std::max<size_t>(1, subs.size()));
	if (const char *e = getenv("MM2AMD_ACTIVE_LANES")) n_drivers = std::max(1, std::min(n_drivers, atoi(e))); // read per run: bench.py takes its un-overlapped kernel times with one lane
	b.n_drivers = n_drivers;
	b.device_finish = be_.finishes_regions(); // one answer for the whole batch
	b.device_regions = rgn_ok_ && b.device_finish && be_.aligns_regions();
	return bp;
}

void Mapper::ensure_drivers(int n)
{
	// (scratch_ never reallocates: the constructor reserved the backend's lane count, and lane threads read scratch_.at(lane) without the lock)
	while ((int)scratch_.size() < n) scratch_.emplace_back(new DriverScratch);
	while ((int)drivers_.size() < n) { const int lane = (int)drivers_.size(); drivers_.emplace_back([this, lane] { driver_loop(lane); }); }
}

// A lane's thread: takes the next sub-batch of the batch run() waits for; when that batch has none left and the next one has been handed over
// by a pipeline (stage(.., may_start_early)), starts on that one -- its run() call then finds part of the work done.
void Mapper::driver_loop(int lane)
{
	name_thread("mm2lane");
	for (;;) {
		std::shared_ptr<BatchRun> b;
		size_t si = 0;
		{
			std::unique_lock<std::mutex> lk(mu_);
			for (;;) {
				if (stop_) return;
				if (cur_run_ && lane < cur_run_->n_drivers && !cur_run_->cancelled && cur_run_->next_sub < cur_run_->subs.size()) { b = cur_run_; break; }
				if (!next_run_ && cur_run_ && pending_ && early_ok_ && lane < lane_cap_) { // the staged batch: ours from here on
					take_locked();
					early_ok_ = false;
					next_run_ = make_run(cur_set_);
				}
				if (next_run_ && lane < next_run_->n_drivers && !next_run_->cancelled && next_run_->next_sub < next_run_->subs.size()) { b = next_run_; break; }
				cv_work_.wait(lk);
			}
			si = b->next_sub++;
			++b->n_taken;
			if (b != cur_run_) ++b->stats.n_early_sub;
		}
		MapperStats st; // this sub-batch's share, merged below
		std::exception_ptr err;
		try {
			// per-lane state that lives as long as the mapper: one Aligner per pool thread (they hold scratch buffers) and the big
			// per-sub-batch arrays, so that steady-state batches allocate (and page-fault) nothing
			DriverScratch &ds = *scratch_.at(lane);
			std::vector<std::unique_ptr<Aligner>> &al = ds.al;
			if (al.empty()) { al.resize(n_threads_); for (auto &p : al) p.reset(new Aligner(opt_, fi_)); }
			for (auto &p : al) p->device_finish(b->device_finish);
			process_sub(*b, b->subs[si].first, b->subs[si].second, lane, al, ds, st);
		} catch (...) {
			err = std::current_exception();
		}
		{
			std::lock_guard<std::mutex> lk(mu_);
			MapperStats &stats = b->stats;
			stats.t_seed_chain += st.t_seed_chain, stats.t_host_pre += st.t_host_pre, stats.t_plan += st.t_plan, stats.t_ksw += st.t_ksw;
			stats.t_consume += st.t_consume, stats.t_finish += st.t_finish, stats.n_jobs += st.n_jobs, stats.n_rounds += st.n_rounds, stats.dp_cells += st.dp_cells;
			stats.c_seed_chain += st.c_seed_chain, stats.c_host_pre += st.c_host_pre, stats.c_plan += st.c_plan, stats.c_ksw += st.c_ksw, stats.c_consume += st.c_consume, stats.c_finish += st.c_finish;
			stats.n_long_join_dev += st.n_long_join_dev, stats.n_long_join_host += st.n_long_join_host;
			stats.n_region_reads_dev += st.n_region_reads_dev, stats.n_region_reads_host += st.n_region_reads_host;
			stats.d_seed_chain += st.d_seed_chain, stats.d_host_pre += st.d_host_pre, stats.d_plan += st.d_plan, stats.d_ksw += st.d_ksw, stats.d_consume += st.d_consume, stats.d_finish += st.d_finish;
			if (err) { if (!b->err) b->err = err; b->cancelled = true; } // no further sub-batch of it is handed out
			++b->n_done;
		}
		cv_done_.notify_all();
	}
}

void Mapper::run(std::vector<ReadResult> &out)
{
	// (No take() here: the caller has taken the batch it wants mapped, under the lock that also orders the hand-overs.  A take at this point
	// raced with the hand-over of the NEXT batch -- when that finished between the caller's take and this line, run() mapped the next batch's
	// reads against the caller's records of this one: tests/test_wave_emu.py::test_pipeline_equals_batch_by_batch with tiny batches.)
	std::shared_ptr<BatchRun> b;
	{
		std::unique_lock<std::mutex> lk(mu_);
		if (next_run_ && next_adopted_) b = next_run_, next_run_.reset(), next_adopted_ = false; // the lanes have started on it
		else b = make_run(cur_set_);
		cur_run_ = b;
		lane_cap_ = std::max(1, be_.n_lanes());
		if (const char *e = getenv("MM2AMD_ACTIVE_LANES")) lane_cap_ = std::max(1, std::min(lane_cap_, atoi(e)));
		// shared budgets (the DP kernels' direction-matrix scratch) are split among this batch's lanes; lanes that start on the NEXT batch early find
		// their share bounded by what the device still has free (KswRunner::run_jobs: the scratch only grows, so the bound is taken where it grows)
		be_.set_active_lanes(b->n_drivers);
		ensure_drivers(std::max(lane_cap_, b->n_drivers)); // (lanes beyond this batch's sub-batches exist too: they are the ones free to start on the next batch)
		cv_work_.notify_all();
		cv_done_.wait(lk, [&] { return b->n_done == b->n_taken && (b->cancelled || b->next_sub >= b->subs.size()); });
		cur_run_.reset();
		stats = b->stats;
	}
	Trace::get().flush();
	out.swap(b->out);
	if (b->err) std::rethrow_exception(b->err);
}

// A read the device finished (Backend::align_regions): its hit records as mm_align_skeleton leaves them (align.c:1048-1108) -- the records of
// chain_regs_kernel with the coordinates region_consume_kernel set, mm_est_err's divergence from the kernel's counts (libm's pow stays here),
// the CIGAR container as mm_update_extra left it (region_finish_kernel), in a libc block the caller frees.
void Mapper::device_hits(const Backend::RegionBatchOut &rb, const ReadChains &c, long i, int qlen, RegVec &regs) const
{
	regs.clear();
	const RgnReadOut &ro = rb.reads[i];
	if (ro.n_regs <= 0) return;
	regs.reserve((size_t)ro.n_regs);
	const float avg_k = ro.avg_k; // esterr.c:37-40: the mean minimizer span, as chain_regs_kernel computed it (with an HPC index the spans are summed: the positions stay on the device)
	(void)c;
	constexpr uint32_t kHdr = sizeof(Extra) / 4;
	//  every slot of the read is checked BEFORE the first block is allocated, and an allocation that fails mid-read frees what the read already holds:
	// an exception out of here must not leave libc blocks behind in a vector nobody hands to the caller
	for (int32_t p = 0; p < ro.n_regs; ++p) {
		const uint32_t slot = ro.reg0 + (uint32_t)p;
		if (rb.plan[slot].status != 0 || rb.fin_res[slot].n_cigar < 0 || (uint32_t)rb.fin_res[slot].n_cigar + kHdr > rb.plan[slot].capacity)
			throw std::runtime_error("[mm2amd] align_regions: a finished region without a consistent CIGAR");
	}
	for (int32_t p = 0; p < ro.n_regs; ++p) {
		const uint32_t slot = ro.reg0 + (uint32_t)p;
		Reg r = rb.regs[slot];
		const RgnAux &x = rb.aux[slot];
		const RgnPlan &pl = rb.plan[slot];
		const FinResult &f = rb.fin_res[slot];
		r.div = x.n_tot < 0 ? -1.0f : x.n_match >= x.n_tot ? 0.0f : (float)(1.0 - pow((double)x.n_match / x.n_tot, 1.0 / avg_k)); // esterr.c:61
		r.p = (Extra *)malloc((size_t)pl.capacity * 4); // (the header is set below, the operations copied; what lies beyond them is as undefined as after the reference's realloc)
		if (!r.p) { for (Reg &done : regs) free(done.p); regs.clear(); throw std::bad_alloc(); }
		memset(r.p, 0, sizeof(Extra));
		r.p->capacity = pl.capacity;
		r.p->n_cigar = (uint32_t)f.n_cigar;
		memcpy(r.p->cigar, rb.cigars + rb.fin[slot].out_off, (size_t)f.n_cigar * 4);
		r.p->dp_score = pl.dp_score, r.p->dp_max = r.p->dp_max0 = f.dp_max, r.p->n_ambi = (uint32_t)f.n_ambi;
		r.blen = f.blen, r.mlen = f.mlen, r.is_spliced = f.is_spliced;
		if (f.qshift) { if (r.rev) r.qe -= f.qshift; else r.qs += f.qshift; } // mm_fix_cigar's dropped leading gap (align.c:171-180)
		r.rs += f.tshift;
		regs.push_back(r);
	}
	(void)qlen;
}

void Mapper::process_sub(BatchRun &batch, long lo, long hi, int lane, std::vector<std::unique_ptr<Aligner>> &al, DriverScratch &ds, MapperStats &stats)
{
	const SeedChainParams &sp = batch.sp;
	std::vector<ReadResult> &out = batch.out;
	const bool device_finish_ = batch.device_finish;
	const std::vector<ReadView> &live = sets_[batch.set].live;
	const std::vector<long> &live_id = sets_[batch.set].live_id;
	const std::vector<uint64_t> &qoff = sets_[batch.set].qoff;
	be_.bind_lane(lane, batch.be_set); // the lane's kernels read this batch's resident set (two batches can be under way: a batch's tail and the next one's start)
	{
		const long m = hi - lo;
		double t0 = now(), c0 = cpu_now(), d0 = thr_now();
		std::vector<ReadChains> &chains = ds.chains;
		SeedChainParams sp_lazy = sp;
		bool lazy = false;
		if (batch.device_regions) { // the chains stay on the device when no read of the sub-batch is a pair (a pair's chains are cut per segment on the host)
			lazy = true;
			for (long i = lo; i < hi && lazy; ++i) lazy = !live[i].paired();
			sp_lazy.lazy_chains = lazy ? 1 : 0;
		}
		be_.seed_chain(sp_lazy, lo, hi, lane, n_threads_, chains);
		std::vector<long> again; // reads to be seeded a second time (below, after the device has taken the others)
		if (opt_.max_occ > opt_.mid_occ && !(opt_.flag & F_RMQ)) {
			// map.c:293-316 for single-segment reads: a read that found no chain although it has repetitive minimizers is seeded
			// again with the occurrence cap raised to max_occ and chained again.  Rare (only with -f x,y): the whole sub-batch is
			// seeded a second time and the results of the reads concerned replace the first ones.
			for (long i = 0; i < m; ++i) {
				const ReadChains &c = chains[i];
				if (c.rep_len <= 0) continue;
				bool rechain = c.n_u == 0;
				if (!rechain && live[lo + i].paired()) { // does the best chain hold anchors of both segments? (map.c:295-306)
					int max = 0, max_i = -1, n_chained_segs = 1;
					int64_t max_off = -1, off = 0;
					for (int32_t k = 0; k < c.n_u; ++k) {
						if (max < (int)(c.u_p[k] >> 32)) max = (int)(c.u_p[k] >> 32), max_i = k, max_off = off;
						off += (uint32_t)c.u_p[k];
					}
					if (max_i >= 0)
						for (int32_t k = 1; k < (int32_t)c.u_p[max_i]; ++k)
							if ((c.a_p[max_off + k].y & SEED_SEG_MASK) != (c.a_p[max_off + k - 1].y & SEED_SEG_MASK)) ++n_chained_segs;
					rechain = n_chained_segs < 2;
				}
				if (rechain) again.push_back(i);
			}
		}
		auto seed_again = [&](const std::vector<uint8_t> *on_device) {
			if (again.empty()) return;
			for (long i = 0; i < m; ++i) if (!on_device || !(*on_device)[i]) chains[i].take_ownership(); // the backend's buffers are about to be reused (a read the device finished needs its chains no more)
			SeedChainParams sp2 = sp;
			sp2.mid_occ = opt_.max_occ;
			sp2.long_join = 0; // map.c:293: this branch is the ELSE of the long-join: its chains are final
			std::vector<ReadChains> second;
			be_.seed_chain(sp2, lo, hi, lane, n_threads_, second);
			for (long i : again) { chains[i] = second[i]; chains[i].take_ownership(); chains[i].long_join_done = true; }
		};
		if (!batch.device_regions) seed_again(nullptr);
		stats.t_seed_chain += now() - t0; t0 = now();
		stats.c_seed_chain += cpu_now() - c0; c0 = cpu_now(); stats.d_seed_chain += thr_now() - d0; d0 = thr_now();

		// ---- chains -> hits -> DP windows -> DP -> finished regions on the device (region_dev.hpp), for every read it can decide alone ----
		const bool is_sr = (opt_.flag & (F_SR | F_SR_RNA)) != 0;
		KswScoring sc;
		Aligner aligner(opt_, fi_);
		memcpy(sc.mat, aligner.mat(), 25);
		sc.m = 5, sc.q = (int8_t)opt_.q, sc.e = (int8_t)opt_.e, sc.q2 = (int8_t)opt_.q2, sc.e2 = (int8_t)opt_.e2, sc.noncan = (int8_t)opt_.noncan;
		sc.single = (opt_.flag & F_SPLICE) ? 2 : (opt_.q == opt_.q2 && opt_.e == opt_.e2) ? 1 : 0; // which DP mm_align_pair picks (align.c:352-360)
		Backend::RegionBatchOut rb;
		std::vector<uint8_t> &on_dev = ds.on_dev;
		on_dev.assign((size_t)m, 0);
		if (batch.device_regions) {
			std::vector<Backend::RegionReadIn> &in = ds.rg_in;
			in.resize((size_t)m);
			const bool host_long_join = opt_.bw_long > opt_.bw && (opt_.flag & (F_SPLICE | F_SR | F_NO_LJOIN)) == 0;
			const bool pairs_on_device = (opt_.flag & F_SR) != 0; // (two-segment fragments of the other presets go through mm_est_err on the fragment first, map.c:333: the host's)
			parallel_for(n_threads_, m, [&](long i, int) {
				const ReadChains &c = chains[i];
				const ReadView &rv = live[lo + i];
				// the host keeps: pairs (but the short-read path's), reads the backend did not chain, reads whose long-join question (map.c:283-292) is still open
				in[i].skip = (rv.paired() && !pairs_on_device) || !c.chained || c.dev_src < 0 || (host_long_join && !c.long_join_done && c.n_u > 1);
				in[i].hash = read_hash(rv.name, rv.total(), opt_);
				int gap_qry;
				chain_gaps(sp, rv.total(), &in[i].gap_ref, &gap_qry);
			}, 1024);
			for (long i : again) in[i].skip = true; // (seeded again below: the host path takes them from their second chains)
			be_.align_regions(lane, rgn_opts_, sc, !is_sr, chains, in, n_threads_, rb);
			long n_dev = 0;
			for (long i = 0; i < m; ++i) on_dev[i] = !in[i].skip && rb.reads[(size_t)i * (size_t)rb.rout_stride].flags == 0, n_dev += on_dev[i];
			if (lazy) { // the hand-backs' chains come to the host now
				std::vector<long> back;
				for (long i = 0; i < m; ++i) if (!on_dev[i] && chains[i].chained && chains[i].dev_src >= 0) back.push_back(i);
				be_.fetch_chains(lane, back, chains);
			}
			seed_again(&on_dev);
			stats.n_region_reads_dev += n_dev, stats.n_region_reads_host += m - n_dev;
			stats.n_jobs += (long)rb.n_jobs, stats.dp_cells += rb.dp_cells, stats.n_rounds += rb.n_jobs ? 1 : 0;
			stats.t_ksw += now() - t0; t0 = now();
			stats.c_ksw += cpu_now() - c0; c0 = cpu_now(); stats.d_ksw += thr_now() - d0; d0 = thr_now();
		} else stats.n_region_reads_host += m;

		// ---- host: chains -> hits, primary/secondary marking, divergence (map.c:283-336) ----
		// A two-segment fragment (paired-end reads) is seeded and chained as one query -- the concatenation of its segments -- and
		// then split: each segment becomes an alignment UNIT of its own (map.c:343-351); a plain read is one unit.
		std::vector<ReadAlign> &ra = ds.ra;
		std::vector<RegVec> &regs0 = ds.regs0;
		std::vector<long> &unit0 = ds.unit0;
		unit0.resize(m + 1);
		unit0[0] = 0;
		for (long i = 0; i < m; ++i) unit0[i + 1] = unit0[i] + (live[lo + i].paired() ? 2 : 1);
		const long mu = unit0[m];
		if ((long)regs0.size() < m) regs0.resize(m);
		if ((long)ra.size() < mu) ra.resize(mu);
		if ((long)ds.seg_regs.size() < mu) ds.seg_regs.resize(mu), ds.seg_a.resize(mu);
		// nt4 copies of the reads for the host-side checks (Z-drop rescoring, CIGAR fixes): one arena per driver instead of one
		// heap block per read
		uint64_t q4_total = 0;
		ds.q4_off.resize(mu + 1);
		for (long i = 0; i < m; ++i) {
			ds.q4_off[unit0[i]] = q4_total, q4_total += 2 * q4_stride(live[lo + i].len);
			if (live[lo + i].paired()) ds.q4_off[unit0[i] + 1] = q4_total, q4_total += 2 * q4_stride(live[lo + i].len2);
		}
		if (ds.q4.size() < q4_total + 16) ds.q4.resize(q4_total + q4_total / 4 + 16); // (every strand block is followed by >= 15 bytes of its own: update_extra compares 16 columns per load, q4_stride)
		std::atomic<long> n_lj_dev{0}, n_lj_host{0};
		parallel_for(n_threads_, m, [&](long i, int) {
			hostprof::Scope hp(hostprof::CHAINS_TO_HITS);
			ReadChains &c = chains[i];
			const ReadView &rv = live[lo + i];
			const int qlen = rv.total(), n_segs = rv.paired() ? 2 : 1, qlens[2] = { rv.len, rv.len2 };
			const long u0 = unit0[i];
			ReadResult &res = out[live_id[lo + i]];
			if (c.long_joined) ++n_lj_dev;
			if (on_dev[i]) { // the device has this read's hits already: nothing to prepare
				int gap_ref, gap_qry;
				chain_gaps(sp, qlen, &gap_ref, &gap_qry);
				res.frag_gap = gap_ref, res.rep_len = c.rep_len; // map.c:317-318
				for (long k = u0; k < unit0[i + 1]; ++k) ra[k].tasks.clear(), ra[k].order.clear(), ra[k].finish_queue.clear();
				return;
			}
			const uint32_t hash = read_hash(rv.name, qlen, opt_);
			if ((opt_.flag & F_RMQ) && !c.chained) { // mg_lchain_rmq as the primary chainer (map.c:275-277), for the reads the backend did not chain
				ChainScratch sc;
				std::vector<uint64_t> u2;
				std::vector<Anchor> out_a;
				chain_rmq(opt_.max_gap, opt_.rmq_inner_dist, opt_.bw, opt_.max_chain_skip, opt_.rmq_size_cap, opt_.min_cnt, opt_.min_chain_score,
				          sp.chn_pen_gap, sp.chn_pen_skip, c.n_a, c.a_p, u2, out_a, sc);
				c.u.swap(u2), c.a.swap(out_a);
				c.u_p = c.u.data(), c.n_u = (int32_t)c.u.size(), c.a_p = c.a.data(), c.n_a = (int64_t)c.a.size();
			}
			if (!c.long_join_done && opt_.bw_long > opt_.bw && (opt_.flag & (F_SPLICE | F_SR | F_NO_LJOIN)) == 0 && n_segs == 1 && c.n_u > 1) { // long-join re-chaining (map.c:283-292): the reads the backend left alone
				const int32_t st = (int32_t)c.a_p[0].y, en = (int32_t)c.a_p[(int32_t)c.u_p[0] - 1].y;
				if (qlen - (en - st) > opt_.rmq_rescue_size || en - st > qlen * opt_.rmq_rescue_ratio) {
					ChainScratch sc;
					std::vector<Anchor> a2(c.a_p, c.a_p + c.n_a);
					sort_by_x(a2.data(), a2.data() + a2.size());
					std::vector<uint64_t> u2;
					std::vector<Anchor> out_a;
					chain_rmq(opt_.max_gap, opt_.rmq_inner_dist, opt_.bw_long, opt_.max_chain_skip, opt_.rmq_size_cap, opt_.min_cnt, opt_.min_chain_score,
					          sp.chn_pen_gap, sp.chn_pen_skip, (int64_t)a2.size(), a2.data(), u2, out_a, sc);
					c.u.swap(u2), c.a.swap(out_a);
					c.u_p = c.u.data(), c.n_u = (int32_t)c.u.size(), c.a_p = c.a.data(), c.n_a = (int64_t)c.a.size();
					++n_lj_host;
				}
			}
			int gap_ref, gap_qry;
			chain_gaps(sp, qlen, &gap_ref, &gap_qry);
			res.frag_gap = gap_ref, res.rep_len = c.rep_len; // map.c:317-318
			RegVec &r0 = regs0[i];
			{ hostprof::Scope hp2(hostprof::GEN_REGS); gen_regs(hash, qlen, c.u_p, c.n_u, c.a_p, (opt_.flag & F_QSTRAND) != 0, r0); }
			if (fi_.n_alt) { // mm_mark_alt + re-sort with ALT hits handicapped (map.c:321-324)
				for (Reg &r : r0) if (fi_.is_alt[r.rid]) r.is_alt = 1;
				hit_sort(r0, opt_.alt_drop);
			}
			if (!(opt_.flag & F_ALL_CHAINS)) { // chain_post (map.c:206-213)
				hostprof::Scope hp2(hostprof::PARENT_SELECT);
				set_parent(opt_.mask_level, opt_.mask_len, r0, opt_.a * 2 + opt_.b, opt_.flag & F_HARD_MLEVEL, opt_.alt_drop);
				if (n_segs <= 1) select_sub(opt_.pri_ratio, fi_.k * 2, opt_.best_n, true, (int)(opt_.max_gap * 0.8), r0);
				else select_sub_multi(opt_.pri_ratio, 0.2f, 0.7f, gap_ref, fi_.k * 2, opt_.best_n, n_segs, qlens, r0);
			}
			if (!(opt_.flag & (F_SR | F_QSTRAND))) { // map.c:333-336
				hostprof::Scope hp2(hostprof::EST_ERR);
				est_err(fi_, qlen, r0, c.a_p, c.mp_p, c.n_mp);
				filter_strand_retained(r0);
			}
			if (n_segs == 1) {
				if (!(opt_.flag & F_CIGAR)) { // mapping without base-level alignment: the chains are the hits (align_regs returns early, map.c:217)
					res.regs = r0;
					set_mapq(res.regs, opt_.min_chain_score, opt_.a, res.rep_len, is_sr, opt_.flag & F_SPLICE);
					return;
				}
				hostprof::Scope hp2(hostprof::BEGIN_READ);
				aligner.begin_read(ra[u0], rv.seq, qlen, r0, c.a_p, qoff[lo + i], qoff[lo + i] + (uint64_t)qlen, ds.q4.data() + ds.q4_off[u0]);
				return;
			}
			// two segments (map.c:343-351): per-segment chains and anchors, primaries chosen again per segment
			seg_gen(hash, 2, qlens, r0, c.a_p, &ds.seg_regs[u0], &ds.seg_a[u0]);
			for (int s = 0; s < 2; ++s) {
				RegVec &rs = ds.seg_regs[u0 + s];
				set_parent(opt_.mask_level, opt_.mask_len, rs, opt_.a * 2 + opt_.b, opt_.flag & F_HARD_MLEVEL, opt_.alt_drop);
				if (!(opt_.flag & F_CIGAR)) {
					RegVec &dst = s == 0 ? res.regs : res.regs2;
					dst = rs;
					set_mapq(dst, opt_.min_chain_score, opt_.a, res.rep_len, is_sr, opt_.flag & F_SPLICE);
					continue;
				}
				// in the query pool every read of a pair has its own  forward | reverse-complement  block, one after the other
				const uint64_t fwd = qoff[lo + i] + (s == 0 ? 0 : 2 * (uint64_t)rv.len), rev = fwd + (uint64_t)qlens[s];
				aligner.begin_read(ra[u0 + s], s == 0 ? rv.seq : rv.seq2, qlens[s], rs, ds.seg_a[u0 + s].data(), fwd, rev, ds.q4.data() + ds.q4_off[u0 + s]);
			}
		});
		Trace::get().add(lane, "host:pre", t0, now());
		stats.n_long_join_dev += n_lj_dev.load(), stats.n_long_join_host += n_lj_host.load();
		stats.t_host_pre += now() - t0;
		stats.c_host_pre += cpu_now() - c0; stats.d_host_pre += thr_now() - d0; d0 = thr_now();

		if (!(opt_.flag & F_CIGAR)) return; // no base-level alignment asked for
		// ---- rounds of plan -> batched DP -> consume (mm_align_skeleton, align.c:1048-1120) ----
		std::vector<std::vector<KswJob>> &per_read_jobs = ds.per_read_jobs;
		if ((long)per_read_jobs.size() < mu) per_read_jobs.resize(mu);
		std::vector<size_t> &job_base = ds.job_base;
		job_base.resize(mu + 1);
		std::vector<KswJob> &jobs = ds.jobs;
		std::vector<KswRes> &kres = ds.kres;
		const uint32_t *cigars = nullptr;
		std::vector<uint8_t> active(mu, 1);
		for (long i = 0; i < m; ++i) if (on_dev[i]) for (long k = unit0[i]; k < unit0[i + 1]; ++k) active[k] = 0; // (a read the device finished; both segments of a pair)
		for (int round = 0;; ++round) {
			t0 = now(), c0 = cpu_now(), d0 = thr_now();
			parallel_for(n_threads_, mu, [&](long i, int tid) {
				per_read_jobs[i].clear();
				if (active[i]) al[tid]->schedule(ra[i], per_read_jobs[i]);
			});
			job_base[0] = 0;
			for (long i = 0; i < mu; ++i) job_base[i + 1] = job_base[i] + per_read_jobs[i].size();
			if (job_base[mu] == 0) break;
			jobs.resize(job_base[mu]);
			parallel_for(n_threads_, mu, [&](long i, int) {
				if (!per_read_jobs[i].empty()) memcpy(&jobs[job_base[i]], per_read_jobs[i].data(), per_read_jobs[i].size() * sizeof(KswJob));
			}, 256);
			if ((fi_.has_junc || fi_.has_spsc) && (opt_.flag & F_SPLICE)) { // the units' junction / splice-score entries as one pool; the jobs' offsets become pool-wide
				std::vector<size_t> &jb = ds.junc_base;
				jb.resize(mu + 1);
				jb[0] = 0;
				for (long i = 0; i < mu; ++i) jb[i + 1] = jb[i] + (per_read_jobs[i].empty() ? 0 : ra[i].juncs.size());
				ds.juncs.resize(jb[mu] + 1);
				parallel_for(n_threads_, mu, [&](long i, int) {
					if (per_read_jobs[i].empty()) return;
					if (!ra[i].juncs.empty()) memcpy(&ds.juncs[jb[i]], ra[i].juncs.data(), ra[i].juncs.size() * 4);
					for (size_t k = job_base[i]; k < job_base[i + 1]; ++k) if (jobs[k].reserved) jobs[k].tag += (uint32_t)jb[i];
				}, 256);
				sc.juncs = ds.juncs.data(), sc.n_juncs = jb[mu], sc.junc_bonus = (int8_t)opt_.junc_bonus, sc.junc_pen = (int8_t)opt_.junc_pen;
			}
			if (opt_.flag & (F_SR_RNA | F_QSTRAND)) { // composed targets (Aligner::add_flank_job, --qstrand windows) as one byte pool; the jobs' offsets become pool-wide
				std::vector<size_t> &tb = ds.tbyte_base;
				tb.resize(mu + 1);
				tb[0] = 0;
				for (long i = 0; i < mu; ++i) tb[i + 1] = tb[i] + (per_read_jobs[i].empty() ? 0 : ra[i].tbytes.size());
				ds.tbytes.resize(tb[mu] + 1);
				parallel_for(n_threads_, mu, [&](long i, int) {
					if (per_read_jobs[i].empty() || ra[i].tbytes.empty()) return;
					memcpy(&ds.tbytes[tb[i]], ra[i].tbytes.data(), ra[i].tbytes.size());
					for (size_t k = job_base[i]; k < job_base[i + 1]; ++k) if (!(jobs[k].flag & KSWJ_T_PACKED)) jobs[k].t_off += tb[i];
				}, 256);
				sc.tbytes = ds.tbytes.data(), sc.n_tbytes = tb[mu];
			}
			for (const KswJob &j : jobs) stats.dp_cells += (double)j.qlen * j.tlen;
			if (const char *dump = getenv("MM2AMD_DUMP_JOBS")) { // debugging aid: the shapes of the DP jobs of every round
				FILE *fp = fopen(dump, "a");
				if (fp) { for (const KswJob &j : jobs) fprintf(fp, "%d\t%d\t%d\t0x%x\t%d\n", round, j.qlen, j.tlen, j.flag & 0x1fff, j.w); fclose(fp); }
			}
			Trace::get().add(lane, "host:plan", t0, now());
			stats.t_plan += now() - t0; t0 = now();
			stats.c_plan += cpu_now() - c0; c0 = cpu_now(); stats.d_plan += thr_now() - d0; d0 = thr_now();
			be_.ksw(jobs, sc, lane, n_threads_, kres, &cigars);
			stats.n_jobs += (long)jobs.size(), ++stats.n_rounds;
			stats.t_ksw += now() - t0; t0 = now();
			stats.c_ksw += cpu_now() - c0; c0 = cpu_now(); stats.d_ksw += thr_now() - d0; d0 = thr_now();
			parallel_for(n_threads_, mu, [&](long i, int tid) {
				if (active[i]) active[i] = al[tid]->consume(ra[i], kres.data() + job_base[i], cigars) ? 1 : 0;
			});
			// the regions whose windows all came back: stitched, left-aligned and counted on the device (region_finish.hip), then completed here
			if (device_finish_) {
				std::vector<FinRegion> &fregs = ds.fin_regions;
				std::vector<FinPiece> &fpieces = ds.fin_pieces;
				std::vector<size_t> &fbase = ds.fin_base; // first queued region of each unit
				fbase.resize(mu + 1);
				fbase[0] = 0;
				for (long i = 0; i < mu; ++i) fbase[i + 1] = fbase[i] + ra[i].finish_queue.size();
				if (fbase[mu] > 0) {
					fregs.resize(fbase[mu]);
					size_t n_pieces = 0, out_words = 0;
					for (long i = 0; i < mu; ++i)
						for (size_t k = 0; k < ra[i].finish_queue.size(); ++k) {
							FinRegion &fr = fregs[fbase[i] + k];
							al[0]->describe_finish(ra[i], ra[i].finish_queue[k], fr);
							fr.piece0 = (uint32_t)n_pieces, fr.out_off = (uint32_t)out_words;
							n_pieces += fr.n_pieces;
							for (const FinPiece &pc : ra[i].tasks[ra[i].finish_queue[k]].pieces) out_words += pc.n;
						}
					if (out_words >= (1ull << 32)) throw std::runtime_error("[mm2amd] region_finish: CIGAR pool of a sub-batch exceeds 32-bit offsets");
					fpieces.resize(n_pieces);
					parallel_for(n_threads_, mu, [&](long i, int) {
						for (size_t k = 0; k < ra[i].finish_queue.size(); ++k) {
							const std::vector<FinPiece> &src = ra[i].tasks[ra[i].finish_queue[k]].pieces;
							memcpy(&fpieces[fregs[fbase[i] + k].piece0], src.data(), src.size() * sizeof(FinPiece));
						}
					}, 256);
					const uint32_t *fin_cigars = nullptr;
					be_.finish_regions(lane, fregs, fpieces, out_words, aligner.mat(), opt_.q, opt_.e, !(opt_.flag & (F_SR | F_SR_RNA)), ds.fin_results, &fin_cigars);
					parallel_for(n_threads_, mu, [&](long i, int tid) {
						if (!ra[i].finish_queue.empty()) active[i] = al[tid]->complete_finished(ra[i], ds.fin_results.data() + fbase[i], fregs.data() + fbase[i], fin_cigars) ? 1 : 0;
					});
				}
			}
			Trace::get().add(lane, "host:consume", t0, now());
			stats.t_consume += now() - t0;
			stats.c_consume += cpu_now() - c0; stats.d_consume += thr_now() - d0; d0 = thr_now();
			if (round > 1000) throw std::runtime_error("[mm2amd] alignment rounds did not converge");
		}

		// ---- final hit selection and MAPQ (map.c:215-225, :339-342), pairing (map.c:353-354) ----
		t0 = now(), c0 = cpu_now(), d0 = thr_now();
		parallel_for(n_threads_, m, [&](long i, int tid) {
			ReadResult &res = out[live_id[lo + i]];
			const ReadView &rv = live[lo + i];
			const int n_segs = rv.paired() ? 2 : 1;
			for (int s = 0; s < n_segs; ++s) {
				RegVec &regs = s == 0 ? res.regs : res.regs2;
				if (on_dev[i]) device_hits(rb, chains[i], i * rb.rout_stride + s, s == 0 ? rv.len : rv.len2, regs), al[tid]->finish_regs(s == 0 ? rv.len : rv.len2, regs);
				else al[tid]->finish_read(ra[unit0[i] + s], regs);
				if (!(opt_.flag & F_ALL_CHAINS)) {
					set_parent(opt_.mask_level, opt_.mask_len, regs, opt_.a * 2 + opt_.b, opt_.flag & F_HARD_MLEVEL, opt_.alt_drop);
					select_sub(opt_.pri_ratio, fi_.k * 2, opt_.best_n, false, (int)(opt_.max_gap * 0.8), regs);
					set_sam_pri(regs);
				}
				set_mapq(regs, opt_.min_chain_score, opt_.a, res.rep_len, is_sr, opt_.flag & F_SPLICE);
			}
			if (fi_.has_jump && n_segs == 1 && (opt_.flag & F_SPLICE)) // map.c:362-364: clipped ends hop over annotated junctions
				for (Reg &r : res.regs) jump_split(fi_, opt_, rv.len, rv.seq, r, 0);
			if (n_segs == 2 && opt_.pe_ori >= 0) {
				const int qlens[2] = { rv.len, rv.len2 };
				RegVec both[2];
				both[0].swap(res.regs), both[1].swap(res.regs2);
				pair_hits(res.frag_gap, opt_.pe_bonus, opt_.a * 2 + opt_.b, opt_.a, qlens, both);
				both[0].swap(res.regs), both[1].swap(res.regs2);
			}
		});
		Trace::get().add(lane, "host:finish", t0, now());
		stats.t_finish += now() - t0;
		stats.c_finish += cpu_now() - c0; stats.d_finish += thr_now() - d0; d0 = thr_now();
	}
}

} // namespace mm2amd

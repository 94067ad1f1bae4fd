// The batched replacement of the reference's per-read driver: what kt_for(worker_for) + mm_map_frag_core
// (map.c:227-378, :425-474) compute for a mini-batch of reads, restructured into whole-batch stages.
#pragma once
#include <condition_variable>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <memory>
#include <vector>
#include "types.hpp"
#include "hits.hpp"
#include "align.hpp"
#include "backend.hpp"

namespace mm2amd {

struct ReadResult {
	RegVec regs;       // final hits; each regs[i].p is libc-allocated (caller frees)
	RegVec regs2;      // ... of the second segment of a two-segment fragment
	int rep_len = 0;   // mm_tbuf_t::rep_len (map.c:318)
	int frag_gap = 0;  // mm_tbuf_t::frag_gap (map.c:317)
};

struct MapperStats { // wall-clock seconds per stage of the last map_batch (for bench / DESIGN.md)
	double t_seed_chain = 0, t_host_pre = 0, t_plan = 0, t_ksw = 0, t_consume = 0, t_finish = 0;
	// process CPU seconds spent while each stage ran (all threads): meaningful when ONE lane drives the sub-batches one after the other
	// (MM2AMD_ACTIVE_LANES=1: bench.py's un-overlapped pass) -- where the host side's cores go when the process runs under a CPU quota
	double c_seed_chain = 0, c_host_pre = 0, c_plan = 0, c_ksw = 0, c_consume = 0, c_finish = 0;
	// CPU seconds of the lane DRIVER threads themselves per stage (CLOCK_THREAD_CPUTIME_ID: launches, copies, waits, serial glue and the driver's own share
	// of the parallel loops); these add up over lanes, whatever runs beside them
	double d_seed_chain = 0, d_host_pre = 0, d_plan = 0, d_ksw = 0, d_consume = 0, d_finish = 0;
	long n_jobs = 0, n_rounds = 0;
	double dp_cells = 0;
	long n_early_sub = 0; // sub-batches of this batch that a lane started before the batch's own run() call (Mapper::stage may_start_early)
	long n_region_reads_dev = 0, n_region_reads_host = 0; // reads whose hits, windows and DP results were handled on the device (Backend::align_regions) / by the host path (a hand-back, or a configuration the device path does not take)
	long n_long_join_dev = 0, n_long_join_host = 0; // reads re-chained by the long-join rule (map.c:283-292): on the device / by the host's tie-exact tree
};

class Mapper {
public:
	Mapper(const FlatIndex &fi, const ref::MapOpt &opt, Backend &be, int n_threads);
	void map_batch(const std::vector<ReadView> &reads, std::vector<ReadResult> &out) { stage(reads); take(); run(out); }
	// the two halves of map_batch: stage() makes the batch resident on the device (the hand-over the reference's pipeline
	// step 0 performs), run() is the hot path proper.  The ReadViews must stay valid until run() returns.  stage() prepares the
	// NEXT batch and run() takes it over: with a backend that stages beside mapping, stage() of batch k+1 may be called from
	// another thread while run() of batch k is under way (the caller orders them: every stage() is followed by one take()+run()).
	// may_start_early (round 4): the caller promises that this staged batch will be mapped exactly once, by the next take() + run() -- a
	// pipeline's queued hand-over.  Lanes that run out of sub-batches of the batch being mapped then start on this one before its own
	// run() call arrives (the tail of a batch -- one or two lanes still busy, the others idle -- was 18 % of the GPU's time: profiles/r04).
	void stage(const std::vector<ReadView> &reads, bool may_start_early = false);
	void take();                                      // the staged batch becomes the one run() maps; the caller calls it, under the lock that orders the hand-overs (run() does not)
	void run(std::vector<ReadResult> &out);
	void discard();                                   // the staged batch will not be mapped (a pipeline shutting down): whatever was started of it is awaited and dropped
	~Mapper();
	bool stages_beside_mapping() const { return be_.stages_beside_mapping(); }
	MapperStats stats;
private:
	struct DriverScratch {
		std::vector<ReadChains> chains;
		std::vector<ReadAlign> ra;
		std::vector<RegVec> regs0;
		std::vector<long> unit0;                 // first alignment unit of each fragment of the sub-batch (a pair has two)
		std::vector<RegVec> seg_regs;            // per unit: the segment's chains (pairs only)
		std::vector<std::vector<Anchor>> seg_a;  // per unit: the segment's anchors (pairs only)
		std::vector<std::vector<KswJob>> per_read_jobs;
		std::vector<size_t> job_base;
		std::vector<KswJob> jobs;
		std::vector<uint32_t> juncs;             // junction annotation entries of the round's jobs (KswScoring::juncs)
		std::vector<size_t> junc_base;
		std::vector<uint8_t> tbytes;             // composed DP targets of the round's jobs (KswScoring::tbytes)
		std::vector<size_t> tbyte_base;
		std::vector<KswRes> kres;
		std::vector<uint8_t> q4;
		std::vector<uint64_t> q4_off;
		std::vector<std::unique_ptr<Aligner>> al;
		std::vector<FinRegion> fin_regions;      // the round's regions for Backend::finish_regions, their windows' CIGARs, the results
		std::vector<FinPiece> fin_pieces;
		std::vector<FinResult> fin_results;
		std::vector<size_t> fin_base;
		std::vector<Backend::RegionReadIn> rg_in;   // Backend::align_regions: per read the hash of map.c:246-248 and whether the host keeps it
		std::vector<uint8_t> on_dev;                // per read: finished by the device path
	};
	std::vector<std::unique_ptr<DriverScratch>> scratch_;
	// One batch on its way through the lanes: its resident set (ours and the backend's), its sub-batches and who has taken them, its results.
	struct BatchRun {
		int set = 0, be_set = 0, n_drivers = 1;
		bool device_finish = false, device_regions = false, cancelled = false;
		SeedChainParams sp;
		std::vector<std::pair<long, long>> subs;
		size_t next_sub = 0, n_done = 0, n_taken = 0; // guarded by mu_
		std::vector<ReadResult> out;
		MapperStats stats;
		std::exception_ptr err;
	};
	void device_hits(const Backend::RegionBatchOut &rb, const ReadChains &c, long i, int qlen, RegVec &regs) const;
	void process_sub(BatchRun &b, long lo, long hi, int lane, std::vector<std::unique_ptr<Aligner>> &al, DriverScratch &ds, MapperStats &st);
	std::shared_ptr<BatchRun> make_run(int set);  // call with mu_ held
	void take_locked();                           // the staged batch becomes the current set; call with mu_ held
	void driver_loop(int lane);
	void ensure_drivers(int n);
	void cancel_next_locked(std::unique_lock<std::mutex> &lk);
	std::mutex mu_;
	std::condition_variable cv_work_, cv_done_;
	std::shared_ptr<BatchRun> cur_run_, next_run_; // the batch run() is waiting for; the staged batch the lanes have started early
	std::vector<std::thread> drivers_;             // one persistent thread per lane
	int lane_cap_ = 1; // lanes that may work at all (the backend's, or MM2AMD_ACTIVE_LANES)
	bool stop_ = false, early_ok_ = false, next_adopted_ = false; // next_adopted_: take() has made next_run_ the caller's batch
	const FlatIndex &fi_;
	ref::MapOpt opt_;
	Backend &be_;
	int n_threads_;
	RgnOpts rgn_opts_{};
	bool rgn_ok_ = false; // the options and the index allow Backend::align_regions (region_path_supported)
	struct Staged { long n = 0; std::vector<ReadView> live; std::vector<long> live_id; std::vector<uint64_t> qoff; };
	Staged sets_[2];
	int cur_set_ = 0;
	bool pending_ = false; // a staged batch run() has not taken over yet
};

uint32_t read_hash(const char *qname, int qlen, const ref::MapOpt &opt); // map.c:246-248

} // namespace mm2amd

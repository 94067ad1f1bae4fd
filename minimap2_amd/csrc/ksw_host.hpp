// Host-side driver of the batched extension-DP kernels: tiers jobs by LDS footprint, orders them by cost,
// sizes the direction-matrix scratch and launches persistent waves.
#pragma once
#include <vector>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_classify.hpp"

namespace mm2amd {

struct KswRunner {
	DevBuf<KswJob> d_jobs;
	DevBuf<KswRes> d_res;
	DevBuf<uint32_t> d_cigar, d_cigar_tmp, d_cursor, d_juncs;
	DevBuf<uint8_t> d_dir, d_dir2, d_dir3, d_state;   // direction-matrix scratch of the concurrent launch groups (ksw_host.cpp)
	DevBuf<uint32_t> d_cigar_tmp2, d_cigar_tmp3;
	hipStream_t side2 = nullptr;              // round 6: the strip kernel's few long gap fills, beside the banded kernel's launches
	hipEvent_t ev_ready2 = nullptr, ev_side2_done = nullptr;
	hipStream_t side = nullptr;               // the lane-exact kernel's launches run here, beside the register-resident kernels on the caller's stream
	hipEvent_t ev_ready = nullptr, ev_side_done = nullptr;
	~KswRunner()
	{
		if (side) { (void)hipStreamDestroy(side); (void)hipEventDestroy(ev_ready); (void)hipEventDestroy(ev_side_done); }
		if (side2) { (void)hipStreamDestroy(side2); (void)hipEventDestroy(ev_ready2); (void)hipEventDestroy(ev_side2_done); }
	}
	DevBuf<int32_t> d_counter;
	PinBuf<KswJob> sorted;            // jobs in launch order (tier, then decreasing cost), pinned for the H2D copy
	PinBuf<KswRes> tmp_res;
	PinBuf<uint32_t> cigar_host;      // the batch's CIGARs as the kernel packed them
	PinBuf<uint32_t> h_cursor;        // the CIGAR pool's cursor and overflow flag, read back after the launches
	std::vector<uint32_t> perm, bucket, chunk_hist;
	int n_threads = 1, lane = 0;
	bool disable_fast = false;       // route every job through the lane-exact kernel (MM2AMD_KSW_EXACT_ONLY=1; for A/B checks)
	class KernelProfiler *prof = nullptr; // optional per-launch timing
	size_t dir_budget = (size_t)12 << 30; // bytes of HBM we allow for direction matrices
	int n_cu = 256;

	// Pools are device pointers.  Results land in res[i] (input order); the CIGARs stay in this runner's pinned buffer
	// (*cigar_out, valid until the next run) and are addressed by res[i].cigar_off / n_cigar.
	void run(const std::vector<KswJob> &jobs, const uint8_t *d_qpool, const uint8_t *d_tpool, const uint32_t *d_S,
	         const KswScoring &sc, KswRes *res, const uint32_t **cigar_out, size_t *n_cigar_out, hipStream_t stream)
	{ run_jobs(jobs.data(), jobs.size(), d_qpool, d_tpool, d_S, sc, res, cigar_out, n_cigar_out, stream); }
	// res == nullptr: the results STAY on the device for a consumer there (region_consume_kernel): d_res in launch order, d_perm[i] = launch
	// position of job i, the CIGARs in d_cigar; nothing but the pool's cursor comes back.  *n_cigar_out = entries used in d_cigar.
	// d_jobs_in (with res == nullptr): the n job records are ON THE DEVICE already (region_plan_kernel's): they are classed and ordered there
	// (ksw_order.hip), `jobs` is not read, and only the per-class sizing figures come to the host.  last_cells = the DP cells of that batch.
	void run_jobs(const KswJob *jobs, size_t n, const uint8_t *d_qpool, const uint8_t *d_tpool, const uint32_t *d_S,
	              const KswScoring &sc, KswRes *res, const uint32_t **cigar_out, size_t *n_cigar_out, hipStream_t stream, const KswJob *d_jobs_in = nullptr);
	DevBuf<uint32_t> d_perm, d_order_work;
	PinBuf<uint32_t> h_perm;
	DevBuf<struct KswOrderResult> d_order_out;
	PinBuf<struct KswOrderResult> h_order_out;
	double last_cells = 0;
	// the banded gap fill (ksw_band.hip): its two lists (windows for the wider band | for the rectangle), how its classes fared, and the share of the best
	// possible score a window is expected to reach (starts at what 12 %-error reads give; follows the accepted windows of the batches before)
	DevBuf<uint32_t> d_band_lists;
	struct BandStats { unsigned long long n_band1 = 0, n_band2 = 0, n_band4 = 0, n_widened = 0, n_retried = 0, n_retried_big = 0; } band_stats;
	double band_rho = 0.5;
};

} // namespace mm2amd

#include <algorithm>
#include <numeric>
#include <cstring>
#include "ksw_host.hpp"
#include "ksw_classify.hpp"
#include "host_prof.hpp"
#include "kernel_prof.hpp"
#include "threads.hpp"
#include "trace.hpp"
#include <atomic>
#include <mutex>
#include <cmath>

namespace mm2amd {

namespace {
// (the launch classes themselves: ksw_classify.hpp)
static const char *const kExtNames[4] = {"ksw_ext_kernel[left-aligned]", "ksw_ext_kernel[right-aligned]", "ksw_ext_kernel[left-aligned,t512]", "ksw_ext_kernel[right-aligned,t512]"};
static const char *const kExtqNames[kExtClasses] = {"ksw_extq_kernel[left-aligned,q128]", "ksw_extq_kernel[right-aligned,q128]", "ksw_extq_kernel[left-aligned,q256]", "ksw_extq_kernel[right-aligned,q256]",
                                                    "ksw_extq_kernel[left-aligned,q512]", "ksw_extq_kernel[right-aligned,q512]"};
// + 0/1: targets up to 256 (left- / right-aligned gaps), + 2/3: up to 512 (eight register sets; these launches hold a few hundred long jobs and are as
// latency-bound as the lane-exact kernel's: they run beside it on the side stream).  Twelve sets (targets up to 768) were measured and dropped: 5 Gcells/s,
// three times the time the lane-exact kernel needs for the same jobs.
const int kSpliceSets[kSpliceClasses] = { 2, 4, 4 };
const bool kSpliceSelf[kSpliceClasses] = { false, false, true };
const int kSpliceWaves[kSpliceClasses] = { 4, 4, 4 };             // waves per block (splice_wpb in ksw_splice.hip)
const int kSpliceBlocksPerCU[kSpliceClasses] = { 4, 4, 4 };
const int kFastQCap[kFirstExact] = { 512, 512, 512, 1024, 1024, 1024 };
const int kRingSize[kRingClasses] = { 256, 512, 1024, 2048, 4096, 8192, 0 };
const int kRingWaves[kRingClasses] = { 4, 4, 4, 1, 1, 1, 4 }; // waves per block where a wave has a job of its own
// wavefronts per job (round 4): anti-diagonals of up to 192 cells stay with one wave; wider ones are swept by a workgroup of 4 or 8 waves, one job
// per workgroup (a band-751 anti-diagonal is twelve 64-lane chunks: one wave needed three passes of twelve chunks per row, and the ~3 k such
// extensions of a step held their launches for as long as the longest one took)
const int kRingTeam[kRingClasses] = { 1, 4, 8, 8, 8, 8, 1 };
constexpr int kMaxWavesPerCU = 20;    // exact kernel: <= 96 VGPRs -> 5 waves/SIMD
inline int fast_waves(int tier) { return kFastQCap[tier] > 512 ? 4 : 6; } // waves per SIMD the gap-fill kernel is compiled for (= blocks of four waves per CU)
inline int stream_sets(int tier) { return ksw_stream_sets(tier); }
}

void ksw_gapfill_launch(const KswLaunch &L, int n_slots, int qcap, void *stream); // ksw_gapfill.hip
void ksw_stream_launch(const KswLaunch &L, int n_slots, int n_sets, void *stream);  // ksw_stream.hip
size_t ksw_stream_slot_bytes(int n_sets);
int ksw_stream_waves(int n_sets);
void ksw_splice_launch(const KswLaunch &L, int n_slots, int n_sets, bool self, void *stream); // ksw_splice.hip
void ksw_ext_launch(const KswLaunch &L, int n_slots, bool right, int n_sets, void *stream);               // ksw_ext.hip
void ksw_extq_launch(const KswLaunch &L, int n_slots, bool right, int n_sets, void *stream);              // ksw_extq.hip
void ksw_band_launch(const KswLaunch &L, int n_slots, int n_sets, void *stream);    // ksw_band.hip
int ksw_band_waves(int n_sets);
int ksw_band_slots(int n_sets);
size_t ksw_band_slot_bytes(int n_sets, int max_rows);

void KswRunner::run_jobs(const KswJob *jobs, size_t n, const uint8_t *d_qpool, const uint8_t *d_tpool, const uint32_t *d_S,
                         const KswScoring &sc, KswRes *res, const uint32_t **cigar_out, size_t *n_cigar_out, hipStream_t stream, const KswJob *d_jobs_in)
{
	const bool resident = res == nullptr; // the results stay on the device (ksw_host.hpp)
	if (d_jobs_in && !resident) throw std::invalid_argument("[mm2amd] KswRunner: jobs on the device go with results on the device");
	*cigar_out = nullptr, *n_cigar_out = 0;
	if (n == 0) return;
	double tt = Trace::now();
	// launch order: tier ascending, then cost (rows * row width) roughly descending (longest-job-first for the persistent
	// waves).  An exact order is not needed, so a counting sort on sqrt(cost) does it.  Jobs that were born on the device (d_jobs_in:
	// region_plan_kernel's) are classed, counted and ordered THERE (ksw_order.hip) and only the per-class sizing figures come back; jobs from the
	// host are cut into chunks, each chunk classified and histogrammed by one pool thread, a short serial prefix turns the histograms into stable
	// scatter offsets, and the chunks scatter in parallel.
	constexpr int NB = kOrderBuckets; // cost buckets per tier
	const bool stream_on = !getenv("MM2AMD_NO_STREAM"); // diagnostic: every gap fill through the strip kernel
	constexpr size_t CH = 32768;
	const size_t NBINS = (size_t)kNTiers * NB;
	static const bool ext_on = !getenv("MM2AMD_NO_EXT_KERNEL"); // diagnostic: every extension through the lane-exact kernel
	static const bool ext_by_target = getenv("MM2AMD_EXT_BY_TARGET") != nullptr; // A/B: round 3's extension kernel (the target across the lanes, targets <= 512) instead of ksw_extq.hip
	static const int ext_max_t = getenv("MM2AMD_EXT_MAX_T") ? atoi(getenv("MM2AMD_EXT_MAX_T")) : ext_by_target ? kExtMaxT : kExtqMaxT; // A/B: longer targets to the lane-exact kernel's workgroups
	int min_sc = sc.mat[1];
	for (int t = 1; t < sc.m * sc.m; ++t) min_sc = std::min<int>(min_sc, sc.mat[t]);
	const bool single_affine = sc.single == 1, splice = sc.single == 2; // the gap-fill kernel is dual-affine only
	const bool scoring_ok = sc.m == 5 && !disable_fast && !single_affine && !splice && -min_sc <= 2 * (std::min(sc.q + sc.e, sc.q2 + sc.e2)); // else ksw_extd2 returns early (ksw2_extd2_sse.c:73)
	int max_abs = 0;
	for (int t = 0; t < sc.m * sc.m; ++t) max_abs = std::max<int>(max_abs, std::abs((int)sc.mat[t]));
	const bool splice_ok = sc.m == 5 && !disable_fast && splice && -min_sc <= 2 * (sc.q + sc.e) && sc.q2 > sc.q + sc.e && sc.e > 0 && sc.q >= 0 && sc.noncan >= 0 &&
	                       sc.q + sc.e + sc.q2 + sc.noncan + max_abs <= 100;
	KswClassCtx cctx;
	cctx.scoring_ok = scoring_ok, cctx.splice_ok = splice_ok, cctx.splice = splice, cctx.stream_on = stream_on, cctx.ext_on = ext_on, cctx.ext_max_t = ext_max_t;
	cctx.ext_by_target = ext_by_target;
	// Extensions with queries beyond 256 stay with the lane-exact kernel: its launches run anyway (the extensions whose band binds), last as long as their longest
	// job whatever they hold, and did not get shorter when these jobs left (r512 34.8 against 36.2 ms per step un-overlapped, call v7) -- while an eight-set class of
	// ksw_extq.hip is a launch of a few hundred pairs at 2.5 us per row of its own: 40 ms per step and direction.  MM2AMD_EXT_MAX_Q=512 brings the class back (A/B).
	const int ext_max_q = getenv("MM2AMD_EXT_MAX_Q") ? atoi(getenv("MM2AMD_EXT_MAX_Q")) : ext_by_target ? kExtMaxQ : 256;
	cctx.ext_max_q = ext_max_q;
	cctx.merge_rings = getenv("MM2AMD_KSW_SPLIT_RINGS") ? 0 : 1;
	// the banded gap fill (ksw_band.hip): which windows try a band first is decided from the score their length lets one expect -- a share of the best possible
	// score that follows what the kernel's accepted windows actually reached (band_rho; MM2AMD_BAND_RHO pins it) -- never the results
	const bool band_env_off = getenv("MM2AMD_NO_BAND") != nullptr; // (read per batch: tests switch it)
	const double rho_env = getenv("MM2AMD_BAND_RHO") ? atof(getenv("MM2AMD_BAND_RHO")) : -1.0;
	cctx.band_on = scoring_ok && stream_on && !band_env_off;
	cctx.sc_max = 0;
	for (int t = 0; t < sc.m * sc.m; ++t) cctx.sc_max = std::max<int>(cctx.sc_max, sc.mat[t]);
	cctx.gq = sc.q, cctx.ge = sc.e, cctx.gq2 = sc.q2, cctx.ge2 = sc.e2;
	cctx.band_rho256 = (int)(256.0 * (rho_env >= 0 ? rho_env : band_rho));
	if (getenv("MM2AMD_BAND_MAX")) cctx.band_max = atoi(getenv("MM2AMD_BAND_MAX")); // A/B: 512 = no four-set class (windows beyond 512 x 512 as rectangles, as before it existed)
	const int band_reject = getenv("MM2AMD_BAND_REJECT") ? atoi(getenv("MM2AMD_BAND_REJECT")) : 0; // tests: 1 = every first attempt fails (the lists and their launches run), 2 = ... straight to the rectangle
	auto r16 = [](int v) { return (v + 15) / 16 * 16; };
	struct ClassStat { size_t slot_bytes = 16, tmp_cap = 16; int max_ring = 64, max_Q16 = 16, max_rows = 1, max_ncol = 64; double alg_bytes = 0, cells = 0, sum_len = 0; };
	size_t sum_len = 0;
	ClassStat cls[kNTiers];
	size_t tier_beg[kNTiers + 1];
	d_jobs.ensure(n);
	if (d_jobs_in) { // the ordering on the device; nothing of the jobs crosses PCIe
		d_perm.ensure(n), d_order_work.ensure(ksw_order_work_words(n)), d_order_out.ensure(1);
		KswOrderResult *ho = h_order_out.ensure(1);
		ksw_order_device(d_jobs_in, n, cctx, d_jobs.p, d_perm.p, d_order_work.p, d_order_out.p, stream);
		HIP_CHECK(hipMemcpyAsync(ho, d_order_out.p, sizeof(KswOrderResult), hipMemcpyDeviceToHost, stream));
		stream_wait(stream);
		for (int t = 0; t < kNTiers; ++t) {
			const KswClassStat &c = ho->cls[t];
			cls[t].slot_bytes = std::max<size_t>(16, (size_t)c.slot_bytes), cls[t].tmp_cap = std::max<size_t>(16, (size_t)c.tmp_cap);
			cls[t].max_ring = std::max(64, (int)c.max_ring), cls[t].max_Q16 = std::max(16, (int)c.max_Q16), cls[t].max_rows = std::max(1, (int)c.max_rows), cls[t].max_ncol = std::max(64, (int)c.max_ncol);
			cls[t].alg_bytes = (double)c.alg_bytes, cls[t].cells = (double)c.cells, cls[t].sum_len = (double)c.sum_len;
			sum_len += (size_t)c.sum_len;
		}
		for (int t = 0; t <= kNTiers; ++t) tier_beg[t] = ho->tier_beg[t];
		last_cells = 0;
		for (int t = 0; t < kNTiers; ++t) last_cells += cls[t].cells;
	} else {
	struct ChunkStat { ClassStat cls[kNTiers]; size_t sum_len = 0; bool too_big = false; };
	const size_t n_chunks = (n + CH - 1) / CH;
	bucket.resize(n), perm.resize(n);
	chunk_hist.assign(n_chunks * NBINS, 0);
	std::vector<ChunkStat> cstat(n_chunks);
	parallel_for(n_threads, (long)n_chunks, [&](long c, int) {
		hostprof::Scope hp(hostprof::KSW_CLASSIFY);
		ChunkStat st;
		uint32_t *hist = &chunk_hist[(size_t)c * NBINS];
		const size_t e = std::min(n, ((size_t)c + 1) * CH);
		for (size_t i = (size_t)c * CH; i < e; ++i) {
			const KswJob &j = jobs[i];
			KswClassOut o;
			ksw_classify(j, cctx, o);
			const int tier = o.tier;
			const uint32_t bk = (uint32_t)(tier * NB + (NB - 1 - o.cb));
			bucket[i] = bk;
			++hist[bk];
			// sizing and accounting for the class (query bytes + packed target + job/result records; the 1 B/cell
			// direction matrix only counts when it cannot stay on chip, i.e. exceeds 160 KB of LDS)
			ClassStat &cs = st.cls[tier];
			cs.alg_bytes += sizeof(KswJob) + sizeof(KswRes);
			if (!o.live) continue;
			cs.alg_bytes += (double)j.qlen + ((j.flag & KSWJ_T_PACKED) ? 0.5 : 1.0) * j.tlen;
			cs.cells += (double)j.qlen * (double)j.tlen;
			cs.max_ring = std::max(cs.max_ring, o.ring_need), cs.max_Q16 = std::max(cs.max_Q16, r16(j.qlen));
			if (!(j.flag & KSW_SCORE_ONLY)) {
				if (o.db > 160 * 1024 || ksw_band_sets(tier)) cs.alg_bytes += (double)o.db; // (the banded kernel's direction bytes -- 64 NB per row and job -- go through HBM whatever their size: two waves per CU is what keeping them in LDS would cost)
				cs.slot_bytes = std::max(cs.slot_bytes, o.db), cs.tmp_cap = std::max(cs.tmp_cap, (size_t)j.qlen + j.tlen);
				if (o.fast || o.xfast) cs.max_rows = std::max(cs.max_rows, j.qlen + j.tlen - 1), cs.max_ncol = std::max(cs.max_ncol, (j.tlen + 63) & ~63);
				st.sum_len += (size_t)j.qlen + j.tlen, cs.sum_len += (double)j.qlen + j.tlen;
			}
		}
		cstat[c] = st;
	}, 1);
	for (const ChunkStat &st : cstat) {
		sum_len += st.sum_len;
		for (int t = 0; t < kNTiers; ++t) {
			cls[t].alg_bytes += st.cls[t].alg_bytes, cls[t].cells += st.cls[t].cells, cls[t].sum_len += st.cls[t].sum_len;
			cls[t].slot_bytes = std::max(cls[t].slot_bytes, st.cls[t].slot_bytes), cls[t].tmp_cap = std::max(cls[t].tmp_cap, st.cls[t].tmp_cap);
			cls[t].max_ring = std::max(cls[t].max_ring, st.cls[t].max_ring), cls[t].max_Q16 = std::max(cls[t].max_Q16, st.cls[t].max_Q16);
			cls[t].max_rows = std::max(cls[t].max_rows, st.cls[t].max_rows), cls[t].max_ncol = std::max(cls[t].max_ncol, st.cls[t].max_ncol);
		}
	}
	{
		uint32_t acc = 0;
		for (size_t b = 0; b < NBINS; ++b) {
			if (b % NB == 0) tier_beg[b / NB] = acc;
			for (size_t c = 0; c < n_chunks; ++c) { uint32_t &h = chunk_hist[c * NBINS + b]; const uint32_t v = h; h = acc; acc += v; }
		}
		for (size_t t = NBINS / NB; t <= (size_t)kNTiers; ++t) tier_beg[t] = acc;
	}
	KswJob *sj = sorted.ensure(n);
	parallel_for(n_threads, (long)n_chunks, [&](long c, int) {
		hostprof::Scope hp(hostprof::KSW_SCATTER);
		uint32_t *off = &chunk_hist[(size_t)c * NBINS];
		const size_t e = std::min(n, ((size_t)c + 1) * CH);
		for (size_t i = (size_t)c * CH; i < e; ++i) { const uint32_t pos = off[bucket[i]]++; perm[i] = pos; sj[pos] = jobs[i]; } // perm[i] = launch position of job i
	}, 1);
	HIP_CHECK(hipMemcpyAsync(d_jobs.p, sj, n * sizeof(KswJob), hipMemcpyHostToDevice, stream));
	if (resident) { // the consumer on the device finds job i's result through the launch order
		uint32_t *hp = h_perm.ensure(n);
		memcpy(hp, perm.data(), n * sizeof(uint32_t));
		d_perm.ensure(n);
		HIP_CHECK(hipMemcpyAsync(d_perm.p, hp, n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
	}
	}

	if (getenv("MM2AMD_KSW_CLASS_DEBUG")) { // diagnostics: what each launch class of this batch holds; the lane-exact classes by kind of job
		stream_wait(stream);
		for (int t = 0; t < kNTiers; ++t) {
			const size_t nt = tier_beg[t + 1] - tier_beg[t];
			if (!nt) continue;
			fprintf(stderr, "[mm2amd] ksw class %d: %zu jobs, longest %zu rows, mean %.0f rows, %.3g cells, ring %d\n", t, nt, cls[t].tmp_cap, cls[t].sum_len / (double)nt, cls[t].cells, cls[t].max_ring);
			if (t < kFirstExact || t >= kFirstSplice) continue;
			std::vector<KswJob> hj(nt);
			HIP_CHECK(hipMemcpy(hj.data(), d_jobs.p + tier_beg[t], nt * sizeof(KswJob), hipMemcpyDeviceToHost));
			struct Kind { size_t n = 0; double rows = 0; int longest = 0, max_q = 0, max_t = 0; } kinds[6];
			static const char *names[6] = { "extension, band cannot bind, query <= 512", "extension, band cannot bind, longer", "extension, band binds", "gap fill, band binds", "gap fill, too long", "other" };
			for (const KswJob &j : hj) {
				const int f = j.flag & 0x1fff;
				const bool ext = f == KSW_EXTZ_ONLY || f == (KSW_EXTZ_ONLY | KSW_RIGHT | KSW_REV_CIGAR), nobind = ksw_band_cannot_bind(j);
				Kind &k = kinds[ext ? (nobind ? (j.qlen <= 512 ? 0 : 1) : 2) : f == KSW_APPROX_MAX ? (nobind ? 4 : 3) : 5];
				++k.n, k.rows += j.qlen + j.tlen, k.longest = std::max(k.longest, j.qlen + j.tlen), k.max_q = std::max(k.max_q, j.qlen), k.max_t = std::max(k.max_t, j.tlen);
			}
			for (int k = 0; k < 6; ++k)
				if (kinds[k].n) fprintf(stderr, "[mm2amd]     %-44s %7zu jobs, mean %6.0f rows, longest %6d, longest query %6d, target %6d\n", names[k], kinds[k].n, kinds[k].rows / kinds[k].n, kinds[k].longest, kinds[k].max_q, kinds[k].max_t);
		}
	}
	Trace::get().add(lane, "host:ksw-order", tt, Trace::now()); tt = Trace::now();
	d_res.ensure(n);
	d_counter.ensure(128 + 16);
	static_assert(kNTiers <= 128, "one queue counter per launch class");
	int32_t *const d_band_ctl = d_counter.p + 128; // the banded kernel's lists: [0] windows for the wider band, [1] for the rectangle, [2] / [3] the queue heads of the launches that take them, [4..7] two 64-bit sums over the windows the first attempts computed: score found (less the corners' gap), best possible score, [8] windows beyond 512 x 512 for the rectangle (the strip kernel), [9] their launch's queue head
	d_cursor.ensure(2);
	KswScoring sc_dev = sc; // the junction entries travel with the jobs
	sc_dev.juncs = nullptr, sc_dev.tbytes = nullptr;
	if (sc.n_juncs) {
		d_juncs.ensure(sc.n_juncs);
		HIP_CHECK(hipMemcpyAsync(d_juncs.p, sc.juncs, sc.n_juncs * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
		sc_dev.juncs = d_juncs.p;
	}
	KswRes *tr = resident ? nullptr : tmp_res.ensure(n);

	// CIGARs are much shorter than qlen+tlen; start with a quarter of the worst case and retry in full on overflow
	static const int pool_div = getenv("MM2AMD_CIGAR_POOL_DIV") ? std::max(1, atoi(getenv("MM2AMD_CIGAR_POOL_DIV"))) : 0; // tests: a first pool that is too small, so that the retry runs
	size_t pool_cap = (pool_div ? sum_len / pool_div : std::min<size_t>(sum_len, sum_len / 4 + 64 * n)) + 16;
	for (int attempt = 0;; ++attempt) {
		if (pool_cap >= (1ull << 32)) throw std::runtime_error("[mm2amd] ksw batch too large for a 32-bit CIGAR pool; split the batch");
		d_cigar.ensure(pool_cap);
		HIP_CHECK(hipMemsetAsync(d_counter.p, 0, (128 + 16) * sizeof(int32_t), stream));
		HIP_CHECK(hipMemsetAsync(d_cursor.p, 0, 2 * sizeof(uint32_t), stream));
		// size every launch class first.  The launches form two groups that run CONCURRENTLY: the register-resident kernels back to
		// back on the caller's stream, the lane-exact kernel's classes back to back on a side stream of higher priority (a few long
		// banded extensions per read: launches with a long tail and few busy CUs, which would otherwise sit between the gap-fill
		// kernel and the copy-back of every sub-batch).  One scratch allocation per group serves all of its launches.
		struct Plan { size_t beg = 0, end = 0, slot_bytes = 16, tmp_cap = 16, n_slots = 0; int ring = 64, max_Q16 = 16, wpb = 4, team = 1; bool hbm = false; double alg_bytes = 0, cells = 0; };
		Plan plan[kNTiers];
		size_t need_dir_g[3] = { 16, 16, 16 }, need_tmp_g[3] = { 16, 16, 16 }, need_state = 0;
		// (the extension classes with a few hundred long jobs per launch -- queries beyond 256; with MM2AMD_EXT_BY_TARGET targets beyond 256 -- run beside the lane-exact kernel's)
		// Round 6, group 2 (MEASURED AND OFF: MM2AMD_SIDE2=1 turns it on): the strip kernel's classes (gap fills beyond 512 x 512: a few long jobs per launch, 3-7 ms of
		// latency each on a handful of CUs) on a stream of their own beside the banded kernel's launches instead of in front of them.  One call, A B: the step 270 -> 319 ms,
		// one rank's share of eight 51.5 -> 66.3 ms (profiles/r06_bench_side2_v14.json / _noside2_): a third queue per lane -- 24 in all -- costs the banded launches more
		// than the strip kernel's latency was worth.
		static const bool side2_off = getenv("MM2AMD_SIDE2") == nullptr;
		const bool strip_on_side2 = !side2_off && stream_on && !splice && !getenv("MM2AMD_NO_SIDE_STREAM");
		auto group_of = [strip_on_side2](int tier) {
			if ((tier >= kFirstExact && tier < kFirstSplice) || (tier >= kFirstExt + (ext_by_target ? 2 : 4) && tier < kFirstBand)) return 1;
			return strip_on_side2 && tier < kFirstExact && !ksw_stream_sets(tier) ? 2 : 0;
		};
		const int max_slots_env = getenv("MM2AMD_KSW_MAX_SLOTS") ? atoi(getenv("MM2AMD_KSW_MAX_SLOTS")) : 0; // tests: few persistent waves, so that each takes many jobs
		// Two groups run concurrently only when there are lane-exact launches and the mode allows it; then each gets half of this lane's
		// scratch budget and buffers of its own.  Otherwise the groups run one after the other and SHARE one buffer sized for the larger.
		bool any_side = false;
		for (int tier = 0; tier < kNTiers; ++tier) any_side |= group_of(tier) == 1 && tier_beg[tier + 1] != tier_beg[tier];
		const bool use_side = any_side && !splice && !getenv("MM2AMD_NO_SIDE_STREAM"); // (spliced alignment: both groups hold matrices of tens of MB per job -- one after the other, each with the whole scratch budget) // read per run: bench.py's un-overlapped pass wants every launch on one stream
		// What this lane may spend on direction matrices: its share of the budget -- and never more than the device has free beside what the lane already
		// holds (the share is computed from the batch's own lane count, but with a queued hand-over the other lanes work on the next batch at the
		// same time, and scratch only grows).  Query and growth are serialised among the lanes, so two lanes cannot both take the same free memory.
		static std::mutex grow_mu;
		std::unique_lock<std::mutex> grow_lk(grow_mu);
		size_t budget_now = dir_budget;
		{
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
				const size_t mine = (d_dir.cap + d_dir2.cap + d_dir3.cap) * sizeof(uint8_t), reserve = (size_t)12 << 30; // (room for the other lanes' per-sub-batch arrays)
				const size_t avail = free_b + mine > reserve ? free_b + mine - reserve : 0;
				budget_now = std::min(budget_now, std::max<size_t>(avail, (size_t)1 << 30));
			}
		}
		bool any_side2 = false;
		for (int tier = 0; tier < kNTiers; ++tier) any_side2 |= group_of(tier) == 2 && tier_beg[tier + 1] != tier_beg[tier];
		const bool use_side2 = any_side2; // (group_of only names group 2 when the mode allows it)
		const size_t group_budget = budget_now / (size_t)(1 + (use_side ? 1 : 0) + (use_side2 ? 1 : 0));
		for (int tier = 0; tier < kNTiers; ++tier) {
			Plan &P = plan[tier];
			size_t &need_dir = need_dir_g[group_of(tier)], &need_tmp = need_tmp_g[group_of(tier)];
			P.beg = tier_beg[tier], P.end = tier_beg[tier + 1];
			if (P.end == P.beg) continue;
			const int band_sets = ksw_band_sets(tier);
			const bool xfast = tier >= kFirstExt && !band_sets, sfast = tier >= kFirstSplice && tier < kFirstExt, fast = tier < kFirstExact || sfast || xfast || band_sets; // the register-resident kernels
			const int rc = fast ? 0 : (tier - kFirstExact) / kDirClasses;
			P.slot_bytes = cls[tier].slot_bytes, P.tmp_cap = cls[tier].tmp_cap, P.max_Q16 = cls[tier].max_Q16, P.alg_bytes = cls[tier].alg_bytes, P.cells = cls[tier].cells;
			// the gap-fill kernel keeps ONE matrix per wave for its two jobs, as many rows as the longer and as many columns as the wider
			// of the two needs (two rows x two jobs per dword): a pair's two slots together must hold (rows / 2 + 1) x columns dwords
			const int n_stream = tier < kFirstExact && stream_on ? stream_sets(tier) : 0;
			if (tier < kFirstExact || band_sets) P.tmp_cap = 3 * (P.tmp_cap + 2); // the operations, and two prefix arrays over them for the half-wave's Z-drop walk (gf_zdrop_scan)
			if (band_sets) P.slot_bytes = ksw_band_slot_bytes(band_sets, cls[tier].max_rows);
			if (tier < kFirstExact) P.slot_bytes = n_stream ? ksw_stream_slot_bytes(n_stream) : (size_t)(cls[tier].max_rows + 3) * (size_t)cls[tier].max_ncol;
			const int extq_sets = xfast && !ext_by_target ? 2 << ((tier - kFirstExt) >> 1) : 0; // ksw_extq.hip: 2 / 4 / 8 register sets of 64 query positions
			if (xfast) P.slot_bytes = (size_t)(cls[tier].max_rows + 3) * (size_t)(extq_sets ? 64 * extq_sets : cls[tier].max_ncol); // one matrix per wave for its two jobs, as in the gap-fill kernel
			P.slot_bytes = (P.slot_bytes + 255) / 256 * 256;
			P.hbm = !fast && rc == kHbmRing;
			P.ring = fast ? 64 : P.hbm ? cls[tier].max_ring : kRingSize[rc];
			const int sclass = sfast ? (tier - kFirstSplice) / kDirClasses : 0;
			P.wpb = sfast ? kSpliceWaves[sclass] : fast ? 4 : kRingWaves[rc];
			const size_t region = (ksw_lds_per_wave(P.ring, P.max_Q16) + 15) / 16 * 16;
			if (!fast && !P.hbm && region > 160 * 1024) P.hbm = true; // a very long query next to a wide window: state goes to HBM
			if (P.hbm) P.wpb = 4;
			static const bool no_team = getenv("MM2AMD_KSW_NO_TEAM") != nullptr; // A/B checks: every lane-exact job on one wave
			P.team = fast || P.hbm || no_team ? 1 : kRingTeam[rc];
			// (round 5, measured and dropped: a sixteen-wave workgroup sweeps a band-751 row -- twelve chunks -- in one round instead of two, and its barriers cost more
			// than the round saves: r1k 57 -> 72 ms per step, profiles/r05_bench_team16_v15.json against _team8_)
			if (P.team > 1) P.wpb = 1; // the workgroup IS the slot
			if (!fast && !P.hbm && region * P.wpb > 160 * 1024) P.wpb = 1;
			int blocks_per_cu;
			if (band_sets) blocks_per_cu = ksw_band_waves(band_sets);
			else if (xfast) blocks_per_cu = (extq_sets ? extq_sets > 4 : tier - kFirstExt >= 2) ? 2 : 4; // (eight register sets: 174 VGPRs)
			else if (sfast) blocks_per_cu = kSpliceBlocksPerCU[sclass];
			else if (n_stream) { static const int sb = getenv("MM2AMD_STREAM_BLOCKS") ? atoi(getenv("MM2AMD_STREAM_BLOCKS")) : 0; blocks_per_cu = sb > 0 ? sb : ksw_stream_waves(n_stream); } // (experiments: fewer resident blocks leave LDS to the other lanes' kernels)
			else if (fast) blocks_per_cu = fast_waves(tier);
			else if (P.hbm) blocks_per_cu = 4;
			else blocks_per_cu = (int)std::min<size_t>((160 * 1024) / (region * P.wpb), kMaxWavesPerCU / (P.wpb * P.team));
			if (blocks_per_cu < 1) blocks_per_cu = 1;
			const int wpb = P.wpb;
			const size_t per_slot = fast && !(sfast && kSpliceSelf[sclass]) ? 2 : 1; // the paired gap-fill kernels run two jobs per wave
			P.n_slots = std::min<size_t>((P.end - P.beg + per_slot - 1) / per_slot, (size_t)n_cu * blocks_per_cu * wpb);
			P.n_slots = std::min<size_t>(P.n_slots, std::max<size_t>(1, group_budget / (P.slot_bytes * per_slot)));
			if (max_slots_env > 0) P.n_slots = std::min<size_t>(P.n_slots, (size_t)max_slots_env);
			P.n_slots = (P.n_slots + wpb - 1) / wpb * wpb;
			need_dir = std::max(need_dir, P.n_slots * P.slot_bytes * per_slot), need_tmp = std::max(need_tmp, P.n_slots * P.tmp_cap * per_slot);
			if (P.hbm) need_state = std::max(need_state, P.n_slots * region);
		}
		// The banded kernel's rejects are computed again by launches that read their job lists on the device: band-128 rejects whose score would pass in a band of
		// 256 diagonals by the two-set instantiation, everything else by the streaming kernel's eight-set class (the full rectangle; query and target <= 512).
		// Their grids are sized for the most the lists can hold; a wave that finds its list empty leaves at once.
		// The four-set class (windows beyond 512 x 512) hands its rejects to the strip kernel; its launch gets a wave per four windows of the class (a matrix slot of
		// this kernel is 2 MB per job; with a wave per sixteen the repeats workload's 6 % rejects took 86 ms per step, call v18).
		struct ListPlan { size_t n_slots = 0, slot_bytes = 16, tmp_cap = 16; } widen_plan, retry_plan, big_plan;
		const size_t n_band1 = plan[kFirstBand].end - plan[kFirstBand].beg, n_band2 = plan[kFirstBand + 1].end - plan[kFirstBand + 1].beg, n_band4 = plan[kFirstBand + 2].end - plan[kFirstBand + 2].beg;
		const size_t n_band = n_band1 + n_band2; // lists: [0, n_band] wider band, [n_band + 1, 2 n_band + 1] rectangle, [2 n_band + 2, ...] rectangle of the big ones
		if (n_band + n_band4) {
			d_band_lists.ensure(2 * n_band + n_band4 + 3);
			const int rows_max = std::max(cls[kFirstBand].max_rows, cls[kFirstBand + 1].max_rows);
			auto size_list = [&](ListPlan &lp, size_t n_max, size_t slot_bytes, int blocks_per_cu, size_t tmp_cap) {
				if (n_max == 0) return;
				lp.slot_bytes = (slot_bytes + 255) / 256 * 256, lp.tmp_cap = 3 * (tmp_cap + 2);
				lp.n_slots = std::min<size_t>((n_max + 1) / 2, (size_t)n_cu * blocks_per_cu * 4);
				lp.n_slots = std::min<size_t>(lp.n_slots, std::max<size_t>(1, group_budget / (lp.slot_bytes * 2)));
				if (max_slots_env > 0) lp.n_slots = std::min<size_t>(lp.n_slots, (size_t)max_slots_env);
				lp.n_slots = (lp.n_slots + 3) / 4 * 4;
				need_dir_g[0] = std::max(need_dir_g[0], lp.n_slots * lp.slot_bytes * 2), need_tmp_g[0] = std::max(need_tmp_g[0], lp.n_slots * lp.tmp_cap * 2);
			};
			const size_t tmp_small = std::max(cls[kFirstBand].tmp_cap, cls[kFirstBand + 1].tmp_cap);
			size_list(widen_plan, n_band1, ksw_band_slot_bytes(2, rows_max), ksw_band_waves(2), tmp_small);
			size_list(retry_plan, n_band, ksw_stream_slot_bytes(8), ksw_stream_waves(8), tmp_small);
			size_list(big_plan, n_band4 ? std::max<size_t>(256, n_band4 / 2) : 0, (size_t)(cls[kFirstBand + 2].max_rows + 3) * (size_t)cls[kFirstBand + 2].max_ncol, fast_waves(4), cls[kFirstBand + 2].tmp_cap);
		}
		for (size_t &need_dir : need_dir_g)
			if (need_dir > ((size_t)1 << 30)) need_dir = (need_dir + ((size_t)2 << 30) - 1) >> 31 << 31; // big scratch grows in 2 GB steps: a slightly larger batch must not cost a 30 GB reallocation
		uint8_t *dir_g[3];
		uint32_t *tmp_g[3];
		if (use_side) {
			d_dir.ensure(need_dir_g[0], 1.0), d_dir2.ensure(need_dir_g[1], 1.0);
			d_cigar_tmp.ensure(need_tmp_g[0], 1.0), d_cigar_tmp2.ensure(need_tmp_g[1], 1.0);
			dir_g[0] = d_dir.p, dir_g[1] = d_dir2.p, tmp_g[0] = d_cigar_tmp.p, tmp_g[1] = d_cigar_tmp2.p;
		} else {
			d_dir.ensure(std::max(need_dir_g[0], need_dir_g[1]), 1.0);
			d_cigar_tmp.ensure(std::max(need_tmp_g[0], need_tmp_g[1]), 1.0);
			dir_g[0] = dir_g[1] = d_dir.p, tmp_g[0] = tmp_g[1] = d_cigar_tmp.p;
		}
		dir_g[2] = dir_g[0], tmp_g[2] = tmp_g[0];
		if (use_side2) {
			d_dir3.ensure(need_dir_g[2], 1.0), d_cigar_tmp3.ensure(need_tmp_g[2], 1.0);
			dir_g[2] = d_dir3.p, tmp_g[2] = d_cigar_tmp3.p;
		}
		if (need_state) d_state.ensure(need_state, 1.0);
		grow_lk.unlock();
		if (use_side) {
			if (!side) {
				int lo_prio = 0, hi_prio = 0;
				HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio));
				HIP_CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi_prio));
				HIP_CHECK(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
				HIP_CHECK(hipEventCreateWithFlags(&ev_side_done, hipEventDisableTiming));
			}
			HIP_CHECK(hipEventRecord(ev_ready, stream)); // job records uploaded, queue heads and cursors zeroed
			HIP_CHECK(hipStreamWaitEvent(side, ev_ready, 0));
		}
		if (use_side2) {
			if (!side2) {
				HIP_CHECK(hipStreamCreateWithFlags(&side2, hipStreamNonBlocking));
				HIP_CHECK(hipEventCreateWithFlags(&ev_ready2, hipEventDisableTiming));
				HIP_CHECK(hipEventCreateWithFlags(&ev_side2_done, hipEventDisableTiming));
			}
			HIP_CHECK(hipEventRecord(ev_ready2, stream));
			HIP_CHECK(hipStreamWaitEvent(side2, ev_ready2, 0));
		}
		static const char *kFastNames[kFirstExact] = { "ksw_gapfill_kernel<512>[t256]", "ksw_gapfill_kernel<512>[t512]", "ksw_gapfill_kernel<512>[t1536]",
		                                               "ksw_gapfill_kernel<1024>[t256]", "ksw_gapfill_kernel<1024>[t1024]", "ksw_gapfill_kernel<1024>[t3072]" };
		static const char *kRingNames[kRingClasses] = { "ksw_extd2_kernel[r256]", "ksw_extd2_kernel[r512]", "ksw_extd2_kernel[r1k]", "ksw_extd2_kernel[r2k]", "ksw_extd2_kernel[r4k]", "ksw_extd2_kernel[r8k]", "ksw_extd2_kernel[hbm]" };
		for (int pass = 0; pass < 3; ++pass) // the side-stream groups first: their long jobs should start as early as possible
		for (int tier = 0; tier < kNTiers; ++tier) {
			const Plan &P = plan[tier];
			if (P.end == P.beg || group_of(tier) != (pass == 0 ? 1 : pass == 1 ? 2 : 0)) continue;
			const bool on_side = use_side && group_of(tier) == 1;
			hipStream_t stream_ = on_side ? side : use_side2 && group_of(tier) == 2 ? side2 : stream;
			KswLaunch L;
			L.jobs = d_jobs.p + P.beg, L.res = d_res.p + P.beg, L.n_jobs = (int32_t)(P.end - P.beg);
			L.qpool = d_qpool, L.tpool = d_tpool, L.S = d_S;
			L.cigar_pool = d_cigar.p, L.cigar_pool_cap = (uint32_t)pool_cap, L.cigar_cursor = d_cursor.p;
			L.cigar_tmp = tmp_g[group_of(tier)], L.cigar_tmp_cap = (uint32_t)P.tmp_cap; // concurrent groups have scratch of their own
			L.dir_pool = dir_g[group_of(tier)], L.slot_bytes = P.slot_bytes;
			L.counter = d_counter.p + tier;
			L.ring = P.ring, L.max_Q16 = P.max_Q16, L.sc = sc_dev;
			L.state_pool = P.hbm ? d_state.p : nullptr;
			L.single_affine = single_affine, L.splice = splice;
			const int band_sets = ksw_band_sets(tier);
			if (band_sets) { // what this launch cannot prove goes onto the lists (positions in the batch's launch order)
				L.widen_list = band_sets == 1 ? d_band_lists.p : nullptr, L.widen_count = d_band_ctl + 0, L.widen_W = 256, L.widen_slots = ksw_band_slots(2);
				L.retry_list = d_band_lists.p + n_band + 1, L.retry_count = d_band_ctl + 1, L.list_base = (uint32_t)P.beg, L.band_reject = band_reject;
				L.big_list = d_band_lists.p + 2 * n_band + 2, L.big_count = d_band_ctl + 8, L.retry_max = kBandMaxSmall;
				L.band_acc = (unsigned long long *)(d_band_ctl + 4);
			}
			if (prof) prof->begin(stream_);
			const int n_stream = tier < kFirstExact && stream_on ? stream_sets(tier) : 0;
			if (band_sets) ksw_band_launch(L, (int)P.n_slots, band_sets, stream_);
			else if (n_stream) ksw_stream_launch(L, (int)P.n_slots, n_stream, stream_);
			else if (tier < kFirstExact) ksw_gapfill_launch(L, (int)P.n_slots, kFastQCap[tier], stream_);
			else if (tier >= kFirstExt && !ext_by_target) ksw_extq_launch(L, (int)P.n_slots, ((tier - kFirstExt) & 1) != 0, 2 << ((tier - kFirstExt) >> 1), stream_);
			else if (tier >= kFirstExt) ksw_ext_launch(L, (int)P.n_slots, ((tier - kFirstExt) & 1) != 0, 4 * ((tier - kFirstExt) / 2 + 1), stream_);
			else if (tier >= kFirstSplice) ksw_splice_launch(L, (int)P.n_slots, kSpliceSets[(tier - kFirstSplice) / kDirClasses], kSpliceSelf[(tier - kFirstSplice) / kDirClasses], stream_);
			else ksw_extd2_launch(L, (int)P.n_slots, P.wpb, P.team, stream_);
			static const char *kSpliceNames[kSpliceClasses] = { "ksw_splice_kernel<2,pair>", "ksw_splice_kernel<4,pair>", "ksw_splice_kernel<4,strips>" };
			static const char *kStreamNames[2] = { "ksw_stream_kernel<4>[t256]", "ksw_stream_kernel<8>[t512]" };
			static const char *kBandNames[kBandClasses] = { "ksw_band_kernel<1>[w128]", "ksw_band_kernel<2>[w256]", "ksw_band_kernel<4>[w512]" };
			static const char *kBandCells[kBandClasses] = { "band_cells_computed<1>", "band_cells_computed<2>", "band_cells_computed<4>" }; // rows x lanes of the band: what the VALU roofline counts (units of the launch itself: the rectangles' cells)
			if (prof && band_sets) prof->add_units(kBandCells[ksw_band_class(band_sets)], cls[tier].sum_len * 64.0 * band_sets);
			if (prof) prof->end(stream_, band_sets ? kBandNames[ksw_band_class(band_sets)] : n_stream ? kStreamNames[tier] : tier >= kFirstExt ? (ext_by_target ? kExtNames[(tier - kFirstExt) & 3] : kExtqNames[tier - kFirstExt]) : tier >= kFirstSplice ? kSpliceNames[(tier - kFirstSplice) / kDirClasses] : tier < kFirstExact ? kFastNames[tier] : P.hbm ? kRingNames[kHbmRing] : kRingNames[(tier - kFirstExact) / kDirClasses], P.alg_bytes, P.cells);
		}
		// ---- the banded kernel's rejects (the lists are complete when the stream gets here) ----
		for (int which = 0; which < 3; ++which) {
			const ListPlan &lp = which == 2 ? big_plan : which ? retry_plan : widen_plan;
			if (lp.n_slots == 0) continue;
			KswLaunch L;
			L.jobs = d_jobs.p, L.res = d_res.p, L.n_jobs = 0;
			L.list = which == 2 ? d_band_lists.p + 2 * n_band + 2 : which ? d_band_lists.p + n_band + 1 : d_band_lists.p;
			L.n_list = which == 2 ? d_band_ctl + 8 : d_band_ctl + which, L.counter = which == 2 ? d_band_ctl + 9 : d_band_ctl + 2 + which;
			L.qpool = d_qpool, L.tpool = d_tpool, L.S = d_S;
			L.cigar_pool = d_cigar.p, L.cigar_pool_cap = (uint32_t)pool_cap, L.cigar_cursor = d_cursor.p;
			L.cigar_tmp = tmp_g[0], L.cigar_tmp_cap = (uint32_t)lp.tmp_cap, L.dir_pool = dir_g[0], L.slot_bytes = lp.slot_bytes;
			L.ring = 64, L.max_Q16 = 16, L.sc = sc_dev;
			if (!which) L.retry_list = d_band_lists.p + n_band + 1, L.retry_count = d_band_ctl + 1, L.list_base = 0, L.band_reject = band_reject > 1 ? band_reject : 0;
			if (prof) prof->begin(stream);
			if (which == 2) ksw_gapfill_launch(L, (int)lp.n_slots, 1024, stream);
			else if (which) ksw_stream_launch(L, (int)lp.n_slots, 8, stream);
			else ksw_band_launch(L, (int)lp.n_slots, 2, stream);
			if (prof) prof->end(stream, which == 2 ? "ksw_gapfill_kernel<1024>[band rejects]" : which ? "ksw_stream_kernel<8>[band rejects]" : "ksw_band_kernel<2>[widened]", 0.0, 0.0);
		}
		if (use_side) {
			HIP_CHECK(hipEventRecord(ev_side_done, side));
			HIP_CHECK(hipStreamWaitEvent(stream, ev_side_done, 0));
		}
		if (use_side2) {
			HIP_CHECK(hipEventRecord(ev_side2_done, side2));
			HIP_CHECK(hipStreamWaitEvent(stream, ev_side2_done, 0));
		}
		// (into PINNED memory: an asynchronous copy to pageable memory -- a stack array here until round 4 -- makes the runtime wait for the stream inside
		// the call, spinning: the lane drivers spent the whole duration of the DP kernels on a core each, 1.2 core-seconds per step)
		uint32_t *cursor = h_cursor.ensure(2 + 16);
		HIP_CHECK(hipMemcpyAsync(cursor, d_cursor.p, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
		const bool any_band = widen_plan.n_slots + retry_plan.n_slots + big_plan.n_slots != 0;
		if (any_band) HIP_CHECK(hipMemcpyAsync(cursor + 2, d_band_ctl, 16 * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
		if (!resident) HIP_CHECK(hipMemcpyAsync(tr, d_res.p, n * sizeof(KswRes), hipMemcpyDeviceToHost, stream));
		stream_wait(stream);
		Trace::get().add(lane, "gpu:ksw", tt, Trace::now()); tt = Trace::now();
		if (cursor[1] == 0 && any_band) { // how the band classes fared: counts for the caller, and the expected score share follows the accepted windows
			band_stats.n_band1 += n_band1, band_stats.n_band2 += n_band2, band_stats.n_band4 += n_band4, band_stats.n_widened += cursor[2], band_stats.n_retried += cursor[3], band_stats.n_retried_big += cursor[2 + 8];
			band_counters().n_band1 += n_band1, band_counters().n_band2 += n_band2, band_counters().n_band4 += n_band4, band_counters().n_widened += cursor[2], band_counters().n_retried += cursor[3], band_counters().n_retried_big += cursor[2 + 8];
			unsigned long long acc[2];
			memcpy(acc, cursor + 2 + 4, sizeof acc);
			const double got = (double)acc[0], best = (double)acc[1];
			static const bool band_debug = getenv("MM2AMD_BAND_DEBUG") != nullptr;
			if (band_debug) fprintf(stderr, "[mm2amd] band: %zu + %zu + %zu windows tried, %u widened, %u + %u to the rectangle, score share %.3f (expected %.3f)\n", n_band1, n_band2, n_band4, cursor[2], cursor[3], cursor[2 + 8], best > 0 ? got / best : 0.0, cctx.band_rho256 / 256.0);
			if (best >= 100000.0) band_rho = std::min(1.0, std::max(0.05, 0.75 * band_rho + 0.25 * (got / best - 0.03)));
		}
		if (cursor[1] == 0 && resident) { *cigar_out = d_cigar.p, *n_cigar_out = cursor[0]; break; }
		if (cursor[1] == 0) {
			uint32_t *hc = cigar_host.ensure((size_t)cursor[0] + 1);
			Trace::get().add(lane, "host:cigar-buffer", tt, Trace::now()); tt = Trace::now();
			if (cursor[0]) {
				HIP_CHECK(hipMemcpyAsync(hc, d_cigar.p, (size_t)cursor[0] * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
				stream_wait(stream);
			}
			*cigar_out = hc, *n_cigar_out = cursor[0];
			Trace::get().add(lane, "d2h:cigar", tt, Trace::now()); tt = Trace::now();
			break;
		}
		if (attempt > 0) throw std::runtime_error("[mm2amd] CIGAR pool overflow even at worst-case size");
		pool_cap = sum_len + 16;
	}
	if (!resident) {
		hostprof::Scope hp(hostprof::KSW_UNPERM);
		parallel_for(n_threads, (long)n, [&](long i, int) { res[i] = tr[perm[i]]; }, 4096);
	}
	Trace::get().add(lane, "host:ksw-unperm", tt, Trace::now());
}

} // namespace mm2amd

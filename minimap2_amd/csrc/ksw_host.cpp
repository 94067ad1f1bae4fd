#include <algorithm>
#include <numeric>
#include <cstring>
#include "ksw_host.hpp"
#include "kernel_prof.hpp"
#include "threads.hpp"
#include <atomic>
#include <cmath>

namespace mm2amd {

namespace {
struct Tier { int max_dim; int waves_per_block; };
// jobs whose 16-rounded max(qlen,tlen) is <= max_dim share a launch; LDS per wave = 13*T16 + Q16 + 16
const Tier kTiers[] = { {512, 4}, {2048, 1}, {11264, 1} };
constexpr int kMaxWavesPerCU = 20; // 81 VGPRs -> 5 waves/SIMD
}

void KswRunner::run(const std::vector<KswJob> &jobs, const uint8_t *d_qpool, const uint8_t *d_tpool, const uint32_t *d_S,
                    const KswScoring &sc, KswRes *res, const uint32_t **cigar_out, size_t *n_cigar_out, hipStream_t stream)
{
	const size_t n = jobs.size();
	*cigar_out = nullptr, *n_cigar_out = 0;
	if (n == 0) return;
	// launch order: tier ascending, then cost (rows * row width) roughly descending (longest-job-first for the persistent
	// waves).  An exact order is not needed, so a counting sort on sqrt(cost) does it in two parallel passes over the jobs.
	constexpr int NB = 4096; // cost buckets per tier
	auto r16 = [](int v) { return (v + 15) / 16 * 16; };
	bucket.resize(n), perm.resize(n);
	std::vector<size_t> sum_len_t((size_t)n_threads + 1, 0);
	std::atomic<bool> too_big(false);
	parallel_for(n_threads, (long)n, [&](long i, int tid) {
		const KswJob &j = jobs[i];
		int dim = std::max(r16(j.qlen), r16(j.tlen)), tier = 0;
		while (tier < 3 && dim > kTiers[tier].max_dim) ++tier;
		if (tier == 3) { too_big = true; tier = 2; }
		const double cost = (j.flag & KSWJ_SKIP) ? 0.0 : (double)(j.qlen + j.tlen) * (double)std::min(std::min(j.qlen, j.tlen), j.w < 0 ? INT32_MAX : j.w + 1);
		int cb = (int)std::sqrt(cost);
		if (cb >= NB) cb = NB - 1;
		bucket[i] = (uint32_t)(tier * NB + (NB - 1 - cb));
		if (!(j.flag & (KSWJ_SKIP | KSW_SCORE_ONLY)) && j.qlen > 0 && j.tlen > 0) sum_len_t[tid] += (size_t)j.qlen + j.tlen;
	}, 4096);
	if (too_big) throw std::runtime_error("[mm2amd] ksw job larger than the LDS-resident kernel supports (qlen/tlen > 11264)");
	size_t sum_len = 0;
	for (size_t v : sum_len_t) sum_len += v;
	std::vector<uint32_t> start(3 * NB + 1, 0);
	for (size_t i = 0; i < n; ++i) ++start[bucket[i] + 1];
	for (int k = 0; k < 3 * NB; ++k) start[k + 1] += start[k];
	size_t tier_beg[4] = { start[0], start[NB], start[2 * NB], start[3 * NB] };
	for (size_t i = 0; i < n; ++i) perm[i] = start[bucket[i]]++; // perm[i] = launch position of job i (stable within a bucket)
	KswJob *sj = sorted.ensure(n);
	parallel_for(n_threads, (long)n, [&](long i, int) { sj[perm[i]] = jobs[i]; }, 4096);

	d_jobs.ensure(n);
	d_res.ensure(n);
	d_counter.ensure(8);
	d_cursor.ensure(2);
	HIP_CHECK(hipMemcpyAsync(d_jobs.p, sj, n * sizeof(KswJob), hipMemcpyHostToDevice, stream));
	KswRes *tr = tmp_res.ensure(n);

	// CIGARs are much shorter than qlen+tlen; start with a quarter of the worst case and retry in full on overflow
	size_t pool_cap = std::min<size_t>(sum_len, sum_len / 4 + 64 * n) + 16;
	for (int attempt = 0;; ++attempt) {
		if (pool_cap >= (1ull << 32)) throw std::runtime_error("[mm2amd] ksw batch too large for a 32-bit CIGAR pool; split the batch");
		d_cigar.ensure(pool_cap);
		HIP_CHECK(hipMemsetAsync(d_counter.p, 0, 8 * sizeof(int32_t), stream));
		HIP_CHECK(hipMemsetAsync(d_cursor.p, 0, 2 * sizeof(uint32_t), stream));
		for (int tier = 0; tier < 3; ++tier) {
			const size_t beg = tier_beg[tier], end = tier_beg[tier + 1];
			if (end == beg) continue;
			int max_T16 = 16, max_Q16 = 16;
			size_t slot_bytes = 16, tmp_cap = 16;
			double alg_bytes = 0; // query bytes + packed target + result record; the 1 B/cell direction matrix only
			                      // counts when it cannot stay on chip (> 160 KB of LDS); CIGAR bytes are added after the launch
			for (size_t i = beg; i < end; ++i) {
				const KswJob &j = sj[i];
				alg_bytes += sizeof(KswJob) + sizeof(KswRes);
				if ((j.flag & KSWJ_SKIP) || j.qlen <= 0 || j.tlen <= 0) continue;
				alg_bytes += (double)j.qlen + ((j.flag & KSWJ_T_PACKED) ? 0.5 : 1.0) * j.tlen;
				if (!(j.flag & KSW_SCORE_ONLY)) { const size_t db = ksw_dir_bytes(j.qlen, j.tlen, j.w); if (db > 160 * 1024) alg_bytes += (double)db; }
				max_T16 = std::max(max_T16, r16(j.tlen)), max_Q16 = std::max(max_Q16, r16(j.qlen));
				if (!(j.flag & KSW_SCORE_ONLY)) {
					slot_bytes = std::max(slot_bytes, ksw_dir_bytes(j.qlen, j.tlen, j.w));
					tmp_cap = std::max(tmp_cap, (size_t)j.qlen + j.tlen);
				}
			}
			slot_bytes = (slot_bytes + 255) / 256 * 256;
			const int wpb = kTiers[tier].waves_per_block;
			const size_t region = (ksw_lds_per_wave(max_T16, max_Q16) + 15) / 16 * 16;
			int blocks_per_cu = (int)std::min<size_t>((160 * 1024) / (region * wpb), kMaxWavesPerCU / wpb);
			if (blocks_per_cu < 1) blocks_per_cu = 1;
			size_t n_slots = std::min<size_t>(end - beg, (size_t)n_cu * blocks_per_cu * wpb);
			n_slots = std::min<size_t>(n_slots, std::max<size_t>(1, dir_budget / slot_bytes));
			n_slots = (n_slots + wpb - 1) / wpb * wpb;
			d_dir.ensure(n_slots * slot_bytes, 1.0);
			d_cigar_tmp.ensure(n_slots * tmp_cap, 1.0);

			KswLaunch L;
			L.jobs = d_jobs.p + beg, L.res = d_res.p + beg, L.n_jobs = (int32_t)(end - beg);
			L.qpool = d_qpool, L.tpool = d_tpool, L.S = d_S;
			L.cigar_pool = d_cigar.p, L.cigar_pool_cap = (uint32_t)pool_cap, L.cigar_cursor = d_cursor.p;
			L.cigar_tmp = d_cigar_tmp.p, L.cigar_tmp_cap = (uint32_t)tmp_cap;
			L.dir_pool = d_dir.p, L.slot_bytes = slot_bytes;
			L.counter = d_counter.p + tier;
			L.max_T16 = max_T16, L.max_Q16 = max_Q16, L.sc = sc;
			if (prof) prof->begin(stream);
			ksw_extd2_launch(L, (int)n_slots, wpb, stream);
			if (prof) prof->end(stream, tier == 0 ? "ksw_extd2_kernel[t0]" : tier == 1 ? "ksw_extd2_kernel[t1]" : "ksw_extd2_kernel[t2]", alg_bytes);
		}
		uint32_t cursor[2];
		HIP_CHECK(hipMemcpyAsync(cursor, d_cursor.p, sizeof cursor, hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipMemcpyAsync(tr, d_res.p, n * sizeof(KswRes), hipMemcpyDeviceToHost, stream));
		HIP_CHECK(hipStreamSynchronize(stream));
		if (cursor[1] == 0) {
			uint32_t *hc = cigar_host.ensure((size_t)cursor[0] + 1);
			if (cursor[0]) {
				HIP_CHECK(hipMemcpyAsync(hc, d_cigar.p, (size_t)cursor[0] * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
				HIP_CHECK(hipStreamSynchronize(stream));
			}
			*cigar_out = hc, *n_cigar_out = cursor[0];
			break;
		}
		if (attempt > 0) throw std::runtime_error("[mm2amd] CIGAR pool overflow even at worst-case size");
		pool_cap = sum_len + 16;
	}
	parallel_for(n_threads, (long)n, [&](long i, int) { res[i] = tr[perm[i]]; }, 4096);
}

} // namespace mm2amd

// C ABI of libmm2amd.so (declared in include/mm2amd.h).
#include <mutex>
#include <string>
#include <vector>
#include <cstring>
#include "../../include/mm2amd.h"
#include "hip_util.hpp"
#include "device_ctx.hpp"
#include "ksw_host.hpp"
#include "kernel_prof.hpp"
#include "region_finish.hpp"
#include "device_sort.hpp"
#include <map>

namespace mm2amd { int capi_fail(int code, const std::string &msg); }
using namespace mm2amd;

namespace {

int fail(int code, const std::string &msg) { return capi_fail(code, msg); }

struct KernelApiState { // buffers of the kernel-level entry points
	KswRunner ksw;
	DevBuf<uint8_t> d_qpool, d_tpool;
};
KernelApiState &kstate() { static KernelApiState s; return s; }

template <typename F>
int guarded(F &&f)
{
	try {
		return f();
	} catch (const HipError &e) {
		std::string s = e.what();
		return fail(s.find("no HIP device") != std::string::npos ? MM2AMD_ENODEV : MM2AMD_EHIP, s);
	} catch (const std::exception &e) {
		return fail(MM2AMD_EINVAL, e.what());
	}
}

} // namespace

extern "C" {

int mm2amd_device_count(void)
{
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess) return fail(MM2AMD_ENODEV, std::string("[mm2amd] hipGetDeviceCount: ") + hipGetErrorString(e));
	return n;
}

static int ksw_batch(int mode, int8_t noncan, int n_jobs, const mm2amd_ksw_job_t *jobs, int8_t m, const int8_t *mat,
                     int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2,
                     mm2amd_ksw_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap)
{
	if (n_jobs < 0 || (n_jobs > 0 && (!jobs || !res)) || !mat || m != 5) return fail(MM2AMD_EINVAL, "[mm2amd] ksw_extd2_batch: bad arguments (m must be 5)");
	if (n_jobs == 0) return 0;
	return guarded([&]() -> int {
		DeviceCtx &dc = device_ctx();
		std::lock_guard<std::mutex> lk(dc.mu);
		ensure_device(dc);
		KernelApiState &d = kstate();
		d.ksw.n_cu = dc.n_cu;
		std::vector<KswJob> dj(n_jobs);
		size_t qtot = 0, ttot = 0, ctot = 0;
		for (int i = 0; i < n_jobs; ++i) {
			const mm2amd_ksw_job_t &j = jobs[i];
			KswJob &o = dj[i];
			o.q_off = qtot, o.t_off = ttot, o.qlen = j.qlen, o.tlen = j.tlen, o.w = j.w, o.zdrop = j.zdrop, o.end_bonus = j.end_bonus;
			o.flag = j.flag & 0x1fff;
			o.tag = (uint32_t)i, o.reserved = 0;
			qtot += j.qlen > 0 ? j.qlen : 0, ttot += j.tlen > 0 ? j.tlen : 0;
		}
		(void)ctot;
		std::vector<uint8_t> hq(qtot + 1), ht(ttot + 1);
		for (int i = 0; i < n_jobs; ++i) {
			if (jobs[i].qlen > 0) memcpy(&hq[dj[i].q_off], jobs[i].query, jobs[i].qlen);
			if (jobs[i].tlen > 0) memcpy(&ht[dj[i].t_off], jobs[i].target, jobs[i].tlen);
		}
		d.d_qpool.ensure(qtot + 1), d.d_tpool.ensure(ttot + 1);
		HIP_CHECK(hipMemcpyAsync(d.d_qpool.p, hq.data(), qtot + 1, hipMemcpyHostToDevice, dc.stream));
		HIP_CHECK(hipMemcpyAsync(d.d_tpool.p, ht.data(), ttot + 1, hipMemcpyHostToDevice, dc.stream));
		KswScoring sc;
		memcpy(sc.mat, mat, 25);
		sc.m = m, sc.q = gapo, sc.e = gape, sc.q2 = gapo2, sc.e2 = gape2, sc.single = (int8_t)mode, sc.noncan = noncan;
		std::vector<KswRes> r(n_jobs);
		const uint32_t *cig = nullptr;
		size_t n_cig = 0;
		d.ksw.prof = &kernel_profiler(0);
		d.ksw.disable_fast = getenv("MM2AMD_KSW_EXACT_ONLY") != nullptr;
		d.ksw.run(dj, d.d_qpool.p, d.d_tpool.p, nullptr, sc, r.data(), &cig, &n_cig, dc.stream);
		kernel_profiler().collect();
		if (n_cig > cigar_pool_cap) return fail(MM2AMD_ENOMEM, "[mm2amd] ksw_extd2_batch: cigar_pool too small (sum(qlen+tlen) always suffices)");
		if (n_cig) memcpy(cigar_pool, cig, n_cig * sizeof(uint32_t));
		for (int i = 0; i < n_jobs; ++i) {
			mm2amd_ksw_res_t &o = res[i];
			o.max = r[i].max, o.zdropped = r[i].zdropped, o.max_q = r[i].max_q, o.max_t = r[i].max_t;
			o.mqe = r[i].mqe, o.mqe_t = r[i].mqe_t, o.mte = r[i].mte, o.mte_q = r[i].mte_q;
			o.score = r[i].score, o.n_cigar = r[i].n_cigar, o.reach_end = r[i].reach_end, o.cigar_off = r[i].cigar_off;
		}
		return 0;
	});
}

int mm2amd_ksw_extd2_batch(int n_jobs, const mm2amd_ksw_job_t *jobs, int8_t m, const int8_t *mat,
                           int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2,
                           mm2amd_ksw_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap)
{
	return ksw_batch(0, 0, n_jobs, jobs, m, mat, gapo, gape, gapo2, gape2, res, cigar_pool, cigar_pool_cap);
}

int mm2amd_ksw_extz2_batch(int n_jobs, const mm2amd_ksw_job_t *jobs, int8_t m, const int8_t *mat, int8_t gapo, int8_t gape,
                           mm2amd_ksw_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap)
{
	return ksw_batch(1, 0, n_jobs, jobs, m, mat, gapo, gape, gapo, gape, res, cigar_pool, cigar_pool_cap);
}

int mm2amd_ksw_exts2_batch(int n_jobs, const mm2amd_ksw_job_t *jobs, int8_t m, const int8_t *mat, int8_t gapo, int8_t gape, int8_t gapo2, int8_t noncan,
                           mm2amd_ksw_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap)
{
	return ksw_batch(2, noncan, n_jobs, jobs, m, mat, gapo, gape, gapo2, 0, res, cigar_pool, cigar_pool_cap);
}

int mm2amd_update_extra_batch(int n_jobs, const mm2amd_fin_job_t *jobs, const int8_t *mat25, int8_t q, int8_t e, int log_gap,
                              mm2amd_fin_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap)
{
	if (n_jobs < 0 || (n_jobs > 0 && (!jobs || !res)) || !mat25) return fail(MM2AMD_EINVAL, "[mm2amd] update_extra_batch: bad arguments");
	if (n_jobs == 0) return 0;
	return guarded([&]() -> int {
		DeviceCtx &dc = device_ctx();
		std::lock_guard<std::mutex> lk(dc.mu);
		ensure_device(dc);
		std::vector<FinRegion> regs(n_jobs);
		std::vector<FinPiece> pieces;
		std::vector<uint32_t> cig;
		size_t qtot = 0, ttot = 0, out_words = 0;
		uint32_t longest = 1;
		for (int i = 0; i < n_jobs; ++i) {
			const mm2amd_fin_job_t &j = jobs[i];
			if (j.qlen < 0 || j.tlen < 0 || j.n_pieces < 0 || (j.n_pieces > 0 && (!j.piece || !j.piece_len))) return fail(MM2AMD_EINVAL, "[mm2amd] update_extra_batch: bad job");
			FinRegion &r = regs[i];
			r.q_pos = qtot, r.t_pos = ttot, r.piece0 = (uint32_t)pieces.size(), r.n_pieces = (uint32_t)j.n_pieces, r.out_off = (uint32_t)out_words;
			r.q_len = j.qlen, r.t_len = j.tlen;
			uint32_t sum = 0;
			for (int k = 0; k < j.n_pieces; ++k) {
				if (j.piece_len[k] < 0) return fail(MM2AMD_EINVAL, "[mm2amd] update_extra_batch: negative piece length");
				pieces.push_back(FinPiece{ (uint32_t)cig.size(), (uint32_t)j.piece_len[k] });
				cig.insert(cig.end(), j.piece[k], j.piece[k] + j.piece_len[k]);
				sum += (uint32_t)j.piece_len[k];
			}
			if (sum > (uint32_t)kFinMaxOps) return fail(MM2AMD_EINVAL, "[mm2amd] update_extra_batch: a region has more CIGAR operations than the kernel stages in LDS");
			longest = std::max(longest, sum);
			out_words += sum;
			qtot += ((size_t)j.qlen + 15) & ~(size_t)7, ttot += ((size_t)j.tlen + 15) & ~(size_t)7; // (the kernel reads aligned 8-byte / 8-code blocks)
		}
		if (out_words > cigar_pool_cap) return fail(MM2AMD_ENOMEM, "[mm2amd] update_extra_batch: cigar_pool too small (the sum of the piece lengths suffices)");
		std::vector<uint8_t> hq(qtot + 16, 0);
		std::vector<uint32_t> hS(ttot / 8 + 4, 0);
		for (int i = 0; i < n_jobs; ++i) {
			const mm2amd_fin_job_t &j = jobs[i];
			if (j.qlen) memcpy(&hq[regs[i].q_pos], j.query, (size_t)j.qlen);
			for (int32_t t = 0; t < j.tlen; ++t) { const uint64_t o = regs[i].t_pos + (uint64_t)t; hS[o >> 3] |= (uint32_t)(j.target[t] & 0xf) << ((o & 7) << 2); }
		}
		DevBuf<uint8_t> d_q;
		DevBuf<uint32_t> d_S, d_cig, d_out;
		DevBuf<FinRegion> d_regs;
		DevBuf<FinPiece> d_pieces;
		DevBuf<FinResult> d_res;
		d_q.ensure(hq.size()), d_S.ensure(hS.size()), d_cig.ensure(cig.size() + 1), d_out.ensure(out_words + 1), d_regs.ensure(n_jobs), d_pieces.ensure(pieces.size() + 1), d_res.ensure(n_jobs);
		HIP_CHECK(hipMemcpyAsync(d_q.p, hq.data(), hq.size(), hipMemcpyHostToDevice, dc.stream));
		HIP_CHECK(hipMemcpyAsync(d_S.p, hS.data(), hS.size() * 4, hipMemcpyHostToDevice, dc.stream));
		if (!cig.empty()) HIP_CHECK(hipMemcpyAsync(d_cig.p, cig.data(), cig.size() * 4, hipMemcpyHostToDevice, dc.stream));
		HIP_CHECK(hipMemcpyAsync(d_regs.p, regs.data(), (size_t)n_jobs * sizeof(FinRegion), hipMemcpyHostToDevice, dc.stream));
		if (!pieces.empty()) HIP_CHECK(hipMemcpyAsync(d_pieces.p, pieces.data(), pieces.size() * sizeof(FinPiece), hipMemcpyHostToDevice, dc.stream));
		FinParams P;
		P.regions = d_regs.p, P.n_regions = n_jobs, P.pieces = d_pieces.p, P.cigar_pool = d_cig.p, P.out_pool = d_out.p, P.results = d_res.p;
		P.qpool = d_q.p, P.S = d_S.p;
		memcpy(P.mat, mat25, 25);
		P.q = q, P.e = e, P.log_gap = log_gap ? 1 : 0;
		P.cap_ops = (int)std::min<uint32_t>((longest + 63) & ~63u, (uint32_t)kFinMaxOps);
		region_finish_launch(P, dc.stream);
		std::vector<FinResult> hr(n_jobs);
		std::vector<uint32_t> ho(out_words + 1);
		HIP_CHECK(hipMemcpyAsync(hr.data(), d_res.p, (size_t)n_jobs * sizeof(FinResult), hipMemcpyDeviceToHost, dc.stream));
		if (out_words) HIP_CHECK(hipMemcpyAsync(ho.data(), d_out.p, out_words * 4, hipMemcpyDeviceToHost, dc.stream));
		HIP_CHECK(hipStreamSynchronize(dc.stream));
		for (int i = 0; i < n_jobs; ++i) {
			const FinResult &f = hr[i];
			mm2amd_fin_res_t &o = res[i];
			o.n_cigar = f.n_cigar, o.blen = f.blen, o.mlen = f.mlen, o.n_ambi = f.n_ambi, o.dp_max = f.dp_max, o.qshift = f.qshift, o.tshift = f.tshift, o.is_spliced = f.is_spliced;
			o.cigar_off = regs[i].out_off;
		}
		if (out_words) memcpy(cigar_pool, ho.data(), out_words * 4);
		return 0;
	});
}

int mm2amd_sort_pairs_u64(uint64_t *keys, uint64_t *vals, uint64_t n, int bits)
{
	if ((n > 0 && (!keys || !vals)) || bits < 0 || bits > 64 || n >= (1ull << 32)) return fail(MM2AMD_EINVAL, "[mm2amd] sort_pairs_u64: bad arguments (n < 2^32, 0 <= bits <= 64)");
	if (n == 0) return 0;
	return guarded([&]() -> int {
		DeviceCtx &dc = device_ctx();
		std::lock_guard<std::mutex> lk(dc.mu);
		ensure_device(dc);
		DevBuf<uint64_t> k0, v0, k1, v1;
		k0.ensure(n, 1.0), v0.ensure(n, 1.0), k1.ensure(n, 1.0), v1.ensure(n, 1.0);
		HIP_CHECK(hipMemcpyAsync(k0.p, keys, n * 8, hipMemcpyHostToDevice, dc.stream));
		HIP_CHECK(hipMemcpyAsync(v0.p, vals, n * 8, hipMemcpyHostToDevice, dc.stream));
		const int where = device_sort_pairs_u64(k0.p, v0.p, k1.p, v1.p, n, bits, dc.stream);
		HIP_CHECK(hipMemcpyAsync(keys, where ? k1.p : k0.p, n * 8, hipMemcpyDeviceToHost, dc.stream));
		HIP_CHECK(hipMemcpyAsync(vals, where ? v1.p : v0.p, n * 8, hipMemcpyDeviceToHost, dc.stream));
		HIP_CHECK(hipStreamSynchronize(dc.stream));
		return 0;
	});
}

int mm2amd_exclusive_sum_u32(const uint32_t *in, uint32_t *out, uint64_t n)
{
	if (!out || (n > 0 && !in) || n >= (1ull << 32)) return fail(MM2AMD_EINVAL, "[mm2amd] exclusive_sum_u32: bad arguments (n < 2^32)");
	return guarded([&]() -> int {
		DeviceCtx &dc = device_ctx();
		std::lock_guard<std::mutex> lk(dc.mu);
		ensure_device(dc);
		DevBuf<uint32_t> d;
		d.ensure(n + 1, 1.0);
		if (n) HIP_CHECK(hipMemcpyAsync(d.p, in, n * 4, hipMemcpyHostToDevice, dc.stream));
		device_exclusive_sum_u32(d.p, d.p, n, dc.stream);
		HIP_CHECK(hipMemcpyAsync(out, d.p, (n + 1) * 4, hipMemcpyDeviceToHost, dc.stream));
		HIP_CHECK(hipStreamSynchronize(dc.stream));
		return 0;
	});
}

long long mm2amd_alloc_counter(int which)
{
	AllocStats &a = alloc_stats();
	BandCounters &b = band_counters();
	return which == 0 ? a.dev_allocs.load() : which == 1 ? a.pin_allocs.load() : which == 2 ? a.ns.load() : which == 3 ? (long long)b.n_band1.load() : which == 4 ? (long long)b.n_band2.load() :
	       which == 5 ? (long long)b.n_widened.load() : which == 6 ? (long long)b.n_retried.load() :
	       which == 7 ? (long long)dev_arena().held_bytes() : which == 8 ? (long long)pin_arena().held_bytes() : which == 9 ? (long long)dev_arena().used_bytes() : which == 10 ? (long long)pin_arena().used_bytes() :
	       which == 11 ? (long long)b.n_band4.load() : (long long)b.n_retried_big.load();
}

void mm2amd_profile_enable(int on)
{
	for (int r = 0; r < kMaxReplicas; ++r) for (int l = 0; l < kMaxProfLanes; ++l) kernel_profiler(l, r).reset();
	KernelProfiler::enabled_flag() = on != 0;
}

int mm2amd_profile_get(mm2amd_kernel_stat_t *out, int cap)
{
	std::map<std::string, KernelStat> all;
	for (int r = 0; r < kMaxReplicas; ++r) for (int l = 0; l < kMaxProfLanes; ++l)
		for (const auto &kv : kernel_profiler(l, r).stats()) {
			KernelStat &k = all[kv.first];
			k.ms += kv.second.ms, k.alg_bytes += kv.second.alg_bytes, k.units += kv.second.units, k.launches += kv.second.launches;
		}
	int n = 0;
	for (const auto &kv : all) {
		if (n >= cap) break;
		mm2amd_kernel_stat_t &o = out[n++];
		memset(&o, 0, sizeof o);
		strncpy(o.name, kv.first.c_str(), sizeof o.name - 1);
		o.ms = kv.second.ms, o.alg_bytes = kv.second.alg_bytes, o.launches = kv.second.launches, o.units = kv.second.units;
	}
	return n;
}

} // extern "C"

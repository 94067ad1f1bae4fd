// Small HIP runtime helpers shared by the product's translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <time.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <atomic>
#include <chrono>

// A kernel's dynamic LDS as an array `name` of T.  (MM2AMD_WAVE_EMU: tests/cpucheck/wave_emu builds the .hip sources for the host.)
// MM2_LOCKSTEP(): a point where the code relies on a wavefront executing in lock step -- every lane's loads above it happen before any
// lane's stores below it, because they are the same instructions.  Nothing on the hardware; the emulator, whose lanes run one after
// the other between cross-lane operations, makes them meet here.
#ifdef MM2AMD_WAVE_EMU
#define MM2_DYN_LDS(T, name) T *const name = (T *)wave_emu::dyn_shared()
#define MM2_LOCKSTEP() __builtin_amdgcn_wave_barrier()
#else
#define MM2_DYN_LDS(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#define MM2_LOCKSTEP() ((void)0)
#endif

namespace mm2amd {

struct HipError : std::runtime_error {
	using std::runtime_error::runtime_error;
};

inline void hip_check(hipError_t e, const char *what, const char *file, int line)
{
	if (e != hipSuccess) {
		char buf[512];
		snprintf(buf, sizeof buf, "[mm2amd] HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
		throw HipError(buf);
	}
}
#define HIP_CHECK(x) ::mm2amd::hip_check((x), #x, __FILE__, __LINE__)

// Wait for a stream WITHOUT spinning.  hipStreamSynchronize busy-waits, and so does hipEventSynchronize -- also on an event created with
// hipEventBlockingSync (round 3 relied on that flag; tools/wait_cost.hip, profiles/r04_wait_cost_v8.txt: 100 % of a core for the whole wait with
// either, on this ROCm).  Five lane drivers waiting that way cost 2.5 core-seconds per 1-Gbase step, paid out of the container's CPU quota
// with the time of the working threads.  So: record an event and POLL it (hipEventQuery) between short sleeps -- 1 % of a core; the
// sleeps start at 20 us and grow to 200 us, waits here are milliseconds long.
inline void stream_wait(hipStream_t s)
{
	struct PerDevice { int dev = -1; hipEvent_t ev = nullptr; ~PerDevice() { if (ev) (void)hipEventDestroy(ev); } }; // (destroyed when the thread ends)
	thread_local PerDevice cache[4]; // a host thread drives the lanes of one replica: one device, rarely more
	int dev = 0;
	HIP_CHECK(hipGetDevice(&dev));
	PerDevice *slot = nullptr;
	for (PerDevice &c : cache) if (c.dev == dev) { slot = &c; break; }
	if (!slot) {
		for (PerDevice &c : cache) if (c.dev < 0) { slot = &c; break; }
		if (!slot) { HIP_CHECK(hipStreamSynchronize(s)); return; }
		HIP_CHECK(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
		slot->dev = dev;
	}
	HIP_CHECK(hipEventRecord(slot->ev, s));
	// (round 5: the sleeps grow to 500 us -- MM2AMD_WAIT_MAX_US -- instead of 200: a lane waits 300-400 ms per sub-batch in half a dozen waits, and eight
	// lanes polling at 5 kHz were a measurable share of the lane threads' CPU seconds; half a millisecond late on a wait of tens of milliseconds is noise)
	static const long max_ns = [] { const char *e = getenv("MM2AMD_WAIT_MAX_US"); const long us = e ? atol(e) : 500; return (us < 20 ? 20 : us) * 1000L; }();
	long ns = 20000;
	for (int it = 0;; ++it) {
		const hipError_t e = hipEventQuery(slot->ev);
		if (e == hipSuccess) return;
		if (e != hipErrorNotReady) HIP_CHECK(e);
		if (it < 4) continue; // (work that is all but done: a few queries back to back)
		timespec ts = { 0, ns };
		nanosleep(&ts, nullptr);
		if (ns < max_ns) ns += ns / 2;
	}
}

// CPUs this process may actually use: the smaller of the hardware thread count and the container's CPU quota (cgroup v2 cpu.max, v1 cfs quota)
int effective_cpus(); // capi_common.cpp

struct AllocStats { std::atomic<long> dev_allocs{0}, pin_allocs{0}; std::atomic<double> dummy{0}; std::atomic<long long> dev_bytes{0}, pin_bytes{0}; std::atomic<long long> ns{0}; };
inline AllocStats &alloc_stats() { static AllocStats s; return s; }
// how the banded gap-fill kernel's launch classes fared since the process started (ksw_host.cpp): windows tried in 128 / 256 diagonals, sent on to the wider
// band, computed again as the full rectangle
struct BandCounters { std::atomic<unsigned long long> n_band1{0}, n_band2{0}, n_widened{0}, n_retried{0}; };
inline BandCounters &band_counters() { static BandCounters s; return s; }

// Grow-only device buffer; contents are NOT preserved across a grow.
template <typename T>
struct DevBuf {
	T *p = nullptr;
	size_t cap = 0;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { if (p) (void)hipFree(p); }
	T *ensure(size_t n, double slack = 1.25)
	{
		if (n > cap) {
			if (p) HIP_CHECK(hipFree(p));
			p = nullptr;
			cap = (size_t)(n * slack) + 64;
			const auto t0_ = std::chrono::steady_clock::now();
			HIP_CHECK(hipMalloc((void **)&p, cap * sizeof(T)));
			alloc_stats().dev_allocs++, alloc_stats().dev_bytes += (long long)(cap * sizeof(T));
			alloc_stats().ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
		}
		return p;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Grow-only pinned host buffer.
template <typename T>
struct PinBuf {
	T *p = nullptr;
	size_t cap = 0;
	PinBuf() = default;
	PinBuf(const PinBuf &) = delete;
	PinBuf &operator=(const PinBuf &) = delete;
	~PinBuf() { if (p) (void)hipHostFree(p); }
	T *ensure(size_t n, double slack = 1.25)
	{
		if (n > cap) {
			if (p) HIP_CHECK(hipHostFree(p));
			p = nullptr;
			cap = (size_t)(n * slack) + 64;
			const auto t0_ = std::chrono::steady_clock::now();
			HIP_CHECK(hipHostMalloc((void **)&p, cap * sizeof(T), hipHostMallocDefault));
			alloc_stats().pin_allocs++, alloc_stats().pin_bytes += (long long)(cap * sizeof(T));
			alloc_stats().ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
		}
		return p;
	}
};

} // namespace mm2amd

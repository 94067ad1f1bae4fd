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
#include <map>
#include <mutex>
#include <vector>
#include <iterator>

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
struct BandCounters { std::atomic<unsigned long long> n_band1{0}, n_band2{0}, n_band4{0}, n_widened{0}, n_retried{0}, n_retried_big{0}; };
inline BandCounters &band_counters() { static BandCounters s; return s; }

// ---------------------------------------------------------------------------------------------------------------------------------------
// Arenas behind DevBuf / PinBuf (round 6).  The work buffers of the lanes are grow-only vectors that find their sizes during the first batches: ~550 hipMalloc and
// ~240 hipHostMalloc calls, the pinned ones at 185 ms per GB (tools/alloc_cost.hip) -- the first batch of a process cost seconds.  Now the buffers are carved out
// of a few large chunks per device (first fit over an offset-ordered free list, neighbours merged on release): a buffer that grows hands its old block back
// and the next one reuses it -- pages are pinned once --, and mm_gpu_init reserves the first chunks before any batch arrives (MM2AMD_ARENA_DEV_GB /
// MM2AMD_ARENA_PIN_GB; 0 = nothing ahead of need).  Requests of more than half a chunk get an allocation of their own, as before.  MM2AMD_NO_ARENA=1 (and the
// emulator build, where AddressSanitizer should see every buffer's own bounds) allocates every buffer directly.
// ---------------------------------------------------------------------------------------------------------------------------------------
class MemArena {
public:
	MemArena(bool pinned, size_t chunk_bytes) : pinned_(pinned), chunk_bytes_(chunk_bytes) {}
	~MemArena() { for (Chunk &c : chunks_) raw_free(c.base); for (auto &kv : solo_) raw_free(kv.first); }
	static bool enabled()
	{
#ifdef MM2AMD_WAVE_EMU
		static const bool on = getenv("MM2AMD_ARENA") != nullptr;
#else
		static const bool on = getenv("MM2AMD_NO_ARENA") == nullptr;
#endif
		return on;
	}
	void *alloc(size_t bytes)
	{
		bytes = (bytes + 255) & ~(size_t)255;
		if (bytes == 0) bytes = 256;
		int dev = 0;
		(void)hipGetDevice(&dev);
		if (!enabled() || bytes > chunk_bytes_ / 2) { // an allocation of its own -- made OUTSIDE the lock: a multi-GB hipMalloc must not hold up the other lanes' small requests
			char *p = raw_alloc(bytes);
			std::lock_guard<std::mutex> lk(mu_);
			solo_[p] = bytes;
			return p;
		}
		for (int pass = 0;; ++pass) {
			{
				std::lock_guard<std::mutex> lk(mu_);
				for (Chunk &c : chunks_) {
					if (c.dev != dev) continue;
					for (auto it = c.free_.begin(); it != c.free_.end(); ++it) {
						if (it->second < bytes) continue;
						const size_t off = it->first, len = it->second;
						c.free_.erase(it);
						if (len > bytes) c.free_[off + bytes] = len - bytes;
						c.used[off] = bytes;
						return c.base + off;
					}
				}
			}
			if (pass >= 2) throw HipError("[mm2amd] MemArena: fresh chunks did not hold the request");
			add_chunk(dev); // (outside the lock as well; two lanes that run dry together each add one)
		}
	}
	void release(void *p)
	{
		if (!p) return;
		std::lock_guard<std::mutex> lk(mu_);
		auto so = solo_.find((char *)p);
		if (so != solo_.end()) { raw_free(so->first); solo_.erase(so); return; }
		for (Chunk &c : chunks_) {
			if ((char *)p < c.base || (char *)p >= c.base + c.size) continue;
			const size_t off = (size_t)((char *)p - c.base);
			auto u = c.used.find(off);
			if (u == c.used.end()) return; // (not ours, or released twice: leave it)
			size_t len = u->second, beg = off;
			c.used.erase(u);
			auto nx = c.free_.lower_bound(beg);
			if (nx != c.free_.end() && nx->first == beg + len) { len += nx->second; nx = c.free_.erase(nx); }
			if (nx != c.free_.begin()) { auto pv = std::prev(nx); if (pv->first + pv->second == beg) { beg = pv->first, len += pv->second; c.free_.erase(pv); } }
			c.free_[beg] = len;
			return;
		}
	}
	size_t held_bytes() { std::lock_guard<std::mutex> lk(mu_); size_t n = 0; for (const Chunk &c : chunks_) n += c.size; return n; }                 // in chunks (all devices)
	size_t used_bytes() { std::lock_guard<std::mutex> lk(mu_); size_t n = 0; for (const Chunk &c : chunks_) for (auto &u : c.used) n += u.second; return n; } // handed out of them
	// hold at least `bytes` in chunks on the current device (context set-up: the first batch then finds its memory there)
	void reserve(size_t bytes)
	{
		if (!enabled()) return;
		int dev = 0;
		(void)hipGetDevice(&dev);
		for (;;) {
			size_t have = 0;
			{ std::lock_guard<std::mutex> lk(mu_); for (const Chunk &c : chunks_) if (c.dev == dev) have += c.size; }
			if (have >= bytes) return;
			add_chunk(dev);
		}
	}
private:
	struct Chunk { char *base; size_t size; int dev; std::map<size_t, size_t> free_, used; };
	char *raw_alloc(size_t bytes)
	{
		void *p = nullptr;
		const auto t0_ = std::chrono::steady_clock::now();
		if (pinned_) { HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault)); alloc_stats().pin_allocs++, alloc_stats().pin_bytes += (long long)bytes; }
		else { HIP_CHECK(hipMalloc(&p, bytes)); alloc_stats().dev_allocs++, alloc_stats().dev_bytes += (long long)bytes; }
		alloc_stats().ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
		return (char *)p;
	}
	void raw_free(char *p) { if (pinned_) (void)hipHostFree(p); else (void)hipFree(p); }
	void add_chunk(int dev) // call WITHOUT the lock
	{
		Chunk c;
		c.base = raw_alloc(chunk_bytes_), c.size = chunk_bytes_, c.dev = dev;
		c.free_[0] = chunk_bytes_;
		std::lock_guard<std::mutex> lk(mu_);
		chunks_.push_back(std::move(c));
	}
	const bool pinned_;
	const size_t chunk_bytes_;
	std::mutex mu_;
	std::vector<Chunk> chunks_;
	std::map<char *, size_t> solo_;
};
inline size_t arena_env_gb(const char *name, size_t dflt) { const char *e = getenv(name); return e ? (size_t)atol(e) : dflt; }
inline MemArena &dev_arena() { static MemArena *a = new MemArena(false, (size_t)4 << 30); return *a; }  // (never destroyed: buffers in static objects may outlive any order of destruction; the process' exit frees the memory)
inline MemArena &pin_arena() { static MemArena *a = new MemArena(true, (size_t)1 << 30); return *a; }

// Grow-only device buffer; contents are NOT preserved across a grow.
template <typename T>
struct DevBuf {
	T *p = nullptr;
	size_t cap = 0;
	DevBuf() = default;
	DevBuf(const DevBuf &) = delete;
	DevBuf &operator=(const DevBuf &) = delete;
	~DevBuf() { if (p) dev_arena().release(p); }
	T *ensure(size_t n, double slack = 1.25)
	{
		if (n > cap) {
			if (p) dev_arena().release(p);
			p = nullptr;
			cap = (size_t)(n * slack) + 64;
			p = (T *)dev_arena().alloc(cap * sizeof(T));
		}
		return p;
	}
	void release() { if (p) dev_arena().release(p); p = nullptr; cap = 0; }
};

// Grow-only pinned host buffer.
template <typename T>
struct PinBuf {
	T *p = nullptr;
	size_t cap = 0;
	PinBuf() = default;
	PinBuf(const PinBuf &) = delete;
	PinBuf &operator=(const PinBuf &) = delete;
	~PinBuf() { if (p) pin_arena().release(p); }
	T *ensure(size_t n, double slack = 1.25)
	{
		if (n > cap) {
			if (p) pin_arena().release(p);
			p = nullptr;
			cap = (size_t)(n * slack) + 64;
			p = (T *)pin_arena().alloc(cap * sizeof(T));
		}
		return p;
	}
};

} // namespace mm2amd

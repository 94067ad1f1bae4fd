// Per-kernel timing with HIP events on the stream the kernels are launched on, plus the algorithmic byte count of each
// launch -- the numbers behind bench.py's "roofline" object.  Off by default; enabling it costs
// two hipEventRecord calls per launch.
#pragma once
#include <map>
#include <string>
#include <vector>
#include "hip_util.hpp"

namespace mm2amd {

struct KernelStat { double ms = 0, alg_bytes = 0, units = 0; long launches = 0; }; // units: DP cells for the DP kernels (0 elsewhere)

class KernelProfiler {
public:
	static bool &enabled_flag() { static bool on = false; return on; } // one switch for all lanes
	bool active_ = false; // latched at begin() so that an end() always matches its begin()
	void begin(hipStream_t s)
	{
		active_ = enabled_flag();
		if (!active_) return;
		Pending p;
		p.e0 = get_event(), p.e1 = get_event();
		HIP_CHECK(hipEventRecord(p.e0, s));
		pending_.push_back(p);
	}
	void end(hipStream_t s, const char *name, double alg_bytes, double units = 0)
	{
		if (!active_) return;
		Pending &p = pending_.back();
		p.name = name, p.bytes = alg_bytes, p.units = units;
		HIP_CHECK(hipEventRecord(p.e1, s));
	}
	// a count that goes with a kernel but is no launch of its own (the banded DP kernel: the cells it computed, beside the cells of the rectangles it replaced)
	void add_units(const char *name, double units) { if (enabled_flag()) stats_[name].units += units; }
	// call after the stream has been synchronised
	void collect()
	{
		for (Pending &p : pending_) {
			float ms = 0;
			if (p.name && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
				KernelStat &k = stats_[p.name];
				k.ms += ms, k.alg_bytes += p.bytes, k.units += p.units, ++k.launches;
			}
			free_.push_back(p.e0), free_.push_back(p.e1);
		}
		pending_.clear();
	}
	void reset() { collect(); stats_.clear(); }
	// the pooled events belong to the device that was current when they were created: a context being torn down drops them, so that a
	// later context on another device does not record them on its streams
	void drop_events() { collect(); for (hipEvent_t e : free_) (void)hipEventDestroy(e); free_.clear(); }
	const std::map<std::string, KernelStat> &stats() const { return stats_; }
private:
	struct Pending { hipEvent_t e0, e1; const char *name = nullptr; double bytes = 0, units = 0; };
	hipEvent_t get_event()
	{
		if (!free_.empty()) { hipEvent_t e = free_.back(); free_.pop_back(); return e; }
		hipEvent_t e;
		HIP_CHECK(hipEventCreate(&e));
		return e;
	}
	std::vector<Pending> pending_;
	std::vector<hipEvent_t> free_;
	std::map<std::string, KernelStat> stats_;
};

constexpr int kMaxProfLanes = 16, kMaxReplicas = 16;
KernelProfiler &kernel_profiler(int lane = 0, int replica = 0); // device_ctx.cpp; one per backend lane of every replica (each is used by one host thread at a time)

} // namespace mm2amd

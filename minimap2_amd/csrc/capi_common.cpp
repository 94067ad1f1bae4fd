#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include "../../include/mm2amd.h"
namespace mm2amd {
// The CPUs this process can really use: hardware threads, capped by the container's CPU quota (cgroup v2: /sys/fs/cgroup/cpu.max
// "quota period"; v1: cpu.cfs_quota_us / cpu.cfs_period_us).  A 256-thread host with a 16-CPU quota runs 64 pool threads at a quarter of
// their speed and spends the quota on their spinning.
int effective_cpus()
{
	static const int n = [] {
		int hw = (int)std::max(1u, std::thread::hardware_concurrency());
		long long quota = -1, period = 100000;
		if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
			char q[64] = { 0 };
			if (fscanf(f, "%63s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
			fclose(f);
		} else {
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
			if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 100000; fclose(g); }
		}
		if (quota > 0 && period > 0) hw = (int)std::min<long long>(hw, std::max<long long>(1, (quota + period - 1) / period));
		return hw;
	}();
	return n;
}
// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); launches that share a queue run one after
// the other.  A mapper drives 8 lanes x 2 streams + the hand-over's: with 4 queues a lane's seeding kernels wait behind another lane's DP kernels
// (bench.py: 1.85 -> 2.01 Gbases/s with 16).  The runtime reads the variable when it initialises, at the process's first HIP call -- so a
// process whose first HIP call is this library's gets the setting from here (never overriding the user's); one that has initialised HIP before
// loading the library (a Python process that imported torch and touched the device) sets it itself: INTEGRATION.md section 5.
// Round 5: set on the library's first way to the device (device_ctx / the device count), not by a static initializer at load time -- loading the
// library no longer touches the process environment; MM2AMD_KEEP_HW_QUEUES=1 leaves the variable alone altogether.
void apply_hw_queue_default()
{
	static const bool once = [] { if (!getenv("MM2AMD_KEEP_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "16", 0); return true; }();
	(void)once;
}
static thread_local std::string g_last_error;
void capi_set_error(const std::string &msg) { g_last_error = msg; }
int capi_fail(int code, const std::string &msg) { g_last_error = msg; return code; }
}
extern "C" {
const char *mm2amd_last_error(void) { return mm2amd::g_last_error.c_str(); }
int mm2amd_version(void) { return 1; }
int mm2amd_host_cpus(void) { return mm2amd::effective_cpus(); }
}

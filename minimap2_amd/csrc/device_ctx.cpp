#include <cstdlib>
#include "device_ctx.hpp"
#include "kernel_prof.hpp"

namespace mm2amd {

namespace {
int device_count() // asked once: the answer does not change while the process lives
{
	static const int n = [] { int v = 0; return hipGetDeviceCount(&v) == hipSuccess ? v : 0; }();
	if (n <= 0) throw HipError("[mm2amd] no HIP device visible: this library has no CPU path");
	return n;
}
}

int default_device()
{
	const int n = device_count();
	int id = 0;
	if (const char *s = getenv("MM2AMD_DEVICE")) id = atoi(s);
	else if (const char *s = getenv("LOCAL_RANK")) id = atoi(s) % n;
	if (id < 0 || id >= n) throw HipError("[mm2amd] MM2AMD_DEVICE names a device this process cannot see");
	return id;
}

void apply_hw_queue_default(); // capi_common.cpp

DeviceCtx &device_ctx(int id)
{
	apply_hw_queue_default(); // (before this process's first HIP call, when that call is ours)
	static DeviceCtx d[kMaxDevices];
	if (id < 0) id = default_device();
	if (id >= kMaxDevices) throw HipError("[mm2amd] device ordinal beyond the supported range");
	d[id].device_id = id;
	return d[id];
}

KernelProfiler &kernel_profiler(int lane, int replica)
{
	static KernelProfiler p[kMaxReplicas][kMaxProfLanes];
	return p[replica < 0 || replica >= kMaxReplicas ? 0 : replica][lane < 0 || lane >= kMaxProfLanes ? 0 : lane];
}

void ensure_device(DeviceCtx &d)
{
	if (d.device_id >= device_count()) throw HipError("[mm2amd] device ordinal beyond the devices this process can see");
	HIP_CHECK(hipSetDevice(d.device_id));
	if (d.ready) return;
	hipDeviceProp_t prop;
	HIP_CHECK(hipGetDeviceProperties(&prop, d.device_id));
	d.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	HIP_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
	d.ready = true;
}

} // namespace mm2amd

#include <cstdlib>
#include "device_ctx.hpp"
#include "kernel_prof.hpp"

namespace mm2amd {

DeviceCtx &device_ctx()
{
	static DeviceCtx d;
	return d;
}

KernelProfiler &kernel_profiler(int lane)
{
	static KernelProfiler p[kMaxProfLanes];
	return p[lane < 0 || lane >= kMaxProfLanes ? 0 : lane];
}

void ensure_device(DeviceCtx &d)
{
	if (d.ready) return;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) throw HipError("[mm2amd] no HIP device visible: this library has no CPU path");
	int id = 0;
	if (const char *s = getenv("MM2AMD_DEVICE")) id = atoi(s);
	else if (const char *s = getenv("LOCAL_RANK")) id = atoi(s) % n;
	HIP_CHECK(hipSetDevice(id));
	hipDeviceProp_t prop;
	HIP_CHECK(hipGetDeviceProperties(&prop, id));
	d.device_id = id;
	d.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
	HIP_CHECK(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
	d.ready = true;
}

} // namespace mm2amd

// HIP device contexts shared by the C ABI entry points and the HIP backend: one per device this process uses (a process maps on
// one device, or on several through mm_gpu_init_multi's replicas).
#pragma once
#include <mutex>
#include "hip_util.hpp"

namespace mm2amd {

struct DeviceCtx {
	std::mutex mu;
	bool ready = false;
	int device_id = 0, n_cu = 256;
	hipStream_t stream = nullptr;
};

constexpr int kMaxDevices = 16;
int default_device();                 // $MM2AMD_DEVICE, else $LOCAL_RANK modulo the device count, else 0
DeviceCtx &device_ctx(int id = -1);   // -1: default_device()
// Makes the context's device current for the calling thread (HIP's current device is per thread) and, the first time, brings it
// up; throws HipError("... no HIP device ...") when there is none.
void ensure_device(DeviceCtx &d);

} // namespace mm2amd

// Process-wide HIP device context shared by the C ABI entry points and the HIP backend.
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

DeviceCtx &device_ctx();
// Bring up $MM2AMD_DEVICE / $LOCAL_RANK / device 0; throws HipError("... no HIP device ...") when there is none.
void ensure_device(DeviceCtx &d);

} // namespace mm2amd

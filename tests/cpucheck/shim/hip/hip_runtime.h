/* TEST INFRASTRUCTURE: stands in for <hip/hip_runtime.h> when a device header is built for the host (tests/cpucheck/sketch_test.cpp):
 * the qualifiers vanish, nothing else is needed by the headers built this way. */
#pragma once
#define __device__
#define __host__
#define __forceinline__ inline

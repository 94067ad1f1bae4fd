/* TEST INFRASTRUCTURE -- never part of libmm2amd.so.
 *
 * Stand-in for <hip/hip_runtime.h> that lets g++ build the product's .hip sources FOR THE HOST, so that the kernels' own source --
 * not a restatement of it -- runs in a container without a GPU (tests/cpucheck/wave_emu/README.md).  A kernel launch runs the
 * blocks of the grid on host threads; inside a block every HIP thread is a fiber with its own stack, and the fibers of a
 * wavefront meet at every cross-lane operation (__shfl*, __ballot, readfirstlane, DPP moves, wave barriers and fences), which is
 * where the values are exchanged.  Between two meeting points a lane runs alone, i.e. with the LARGEST possible skew between lanes:
 * code that relies on lock-step execution without saying so (a fence or a wave barrier) fails here, although it may pass on the
 * hardware.  "Device memory" is host memory.  Only what the product's sources use is provided. */
#pragma once
#define MM2AMD_WAVE_EMU 1
#include <cstdint>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <functional>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __constant__ static
#define __shared__ static thread_local
#define __launch_bounds__(...)

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace wave_emu {

enum Op : int { OP_NONE = 0, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_BALLOT, OP_FIRST, OP_BARRIER, OP_DPP, OP_READLANE };

struct LaneView { dim3 tid, bid, bdim, gdim; };
LaneView &here();                      // the running HIP thread's coordinates
void *dyn_shared();                    // the block's dynamic shared memory
// all live lanes of the running lane's wavefront meet; each contributes (v, arg) and receives what `op` gives its lane
uint64_t collective(Op op, uint64_t v, int64_t arg, int width);
void block_barrier();                  // __syncthreads
void launch(dim3 grid, dim3 block, size_t dyn_shared_bytes, const std::function<void()> &body);
int host_threads();                    // blocks of a launch run on this many host threads (MM2AMD_EMU_THREADS, default 8)

template <typename T> inline uint64_t to_bits(T v) { uint64_t b = 0; static_assert(sizeof(T) <= 8, "cross-lane values are at most 64 bits"); memcpy(&b, &v, sizeof(T)); return b; }
template <typename T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

} // namespace wave_emu

#define threadIdx (wave_emu::here().tid)
#define blockIdx (wave_emu::here().bid)
#define blockDim (wave_emu::here().bdim)
#define gridDim (wave_emu::here().gdim)
#define warpSize 64

// ---- cross-lane operations ----
template <typename T> inline T __shfl(T v, int src, int width = 64) { return wave_emu::from_bits<T>(wave_emu::collective(wave_emu::OP_SHFL, wave_emu::to_bits(v), src, width)); }
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) { return wave_emu::from_bits<T>(wave_emu::collective(wave_emu::OP_SHFL_UP, wave_emu::to_bits(v), d, width)); }
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) { return wave_emu::from_bits<T>(wave_emu::collective(wave_emu::OP_SHFL_DOWN, wave_emu::to_bits(v), d, width)); }
template <typename T> inline T __shfl_xor(T v, int m, int width = 64) { return wave_emu::from_bits<T>(wave_emu::collective(wave_emu::OP_SHFL_XOR, wave_emu::to_bits(v), m, width)); }
inline unsigned long long __ballot(int pred) { return wave_emu::collective(wave_emu::OP_BALLOT, pred ? 1 : 0, 0, 64); }
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)wave_emu::collective(wave_emu::OP_FIRST, (uint32_t)v, 0, 64); }
inline int __builtin_amdgcn_readlane(int v, int lane) { return (int)(uint32_t)wave_emu::collective(wave_emu::OP_READLANE, (uint32_t)v, lane, 64); }
// v_mov_b32_dpp: only the controls the kernels use -- 0x138 wave_shr:1 (lane 0 keeps `old`), 0x13c wave_ror:1, 0x130 wave_shl:1 (lane 63 keeps `old`), 0x134 wave_rol:1, 0x111..0x11f row_shr:n,
// 0x142 / 0x143 row_bcast:15 / 31; a lane whose row is not in row_mask, or that has no source lane, keeps `old` (bound_ctrl clear)
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
	(void)bank_mask, (void)bound_ctrl;
	return (int)(uint32_t)wave_emu::collective(wave_emu::OP_DPP, (uint64_t)(uint32_t)src | (uint64_t)(uint32_t)old << 32, ctrl | (row_mask & 0xf) << 16, 64);
}
inline uint32_t __builtin_amdgcn_perm(uint32_t a, uint32_t b, uint32_t sel) // v_perm_b32: bytes 0-3 of b, 4-7 of a; selector >= 0x0c gives 0
{
	const uint64_t src = (uint64_t)a << 32 | b;
	uint32_t r = 0;
	for (int i = 0; i < 4; ++i) { const uint32_t s = sel >> (8 * i) & 0xff; const uint32_t byte = s <= 7 ? (uint32_t)(src >> (8 * s) & 0xff) : s == 0x0c ? 0u : s >= 0x0d ? 0xffu : 0u; r |= byte << (8 * i); }
	return r;
}
inline void __syncthreads() { wave_emu::block_barrier(); }
// Lock-step execution makes "all lanes' stores, then all lanes' loads" out of a fence between them; here the lanes have to meet.
inline void __threadfence_block() { wave_emu::collective(wave_emu::OP_BARRIER, 0, 0, 64); }
inline void __threadfence() { wave_emu::collective(wave_emu::OP_BARRIER, 0, 0, 64); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { wave_emu::collective(wave_emu::OP_BARRIER, 0, 0, 64); }
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }
inline unsigned long long wall_clock64() { return 0; }

// ---- scalar helpers ----
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clzll(long long v) { return v == 0 ? 64 : __builtin_clzll((unsigned long long)v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
template <typename T> inline T __ldg(const T *p) { return *p; }
template <typename T> inline void __builtin_nontemporal_store(T v, T *p) { *p = v; }

// ---- atomics (blocks of one launch run on several host threads) ----
inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicMax(unsigned *p, unsigned v) { unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = __atomic_load_n(p, __ATOMIC_RELAXED); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicExch(unsigned *p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }

// ---- the runtime API the product's host code calls; "device memory" is host memory, every stream is synchronous ----
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
typedef struct wave_emu_stream *hipStream_t;
typedef struct wave_emu_event { double t; } *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipEventBlockingSync = 1, hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[64]; char gcnArchName[64]; int multiProcessorCount; int clockRate; size_t totalGlobalMem; };
inline const char *hipGetErrorString(hipError_t) { return "wave_emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorInvalidValue; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyPeer(void *d, int, const void *s, int, size_t n) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
#define HIP_SYMBOL(x) ((void *)&(x))
inline hipError_t hipMemcpyToSymbolAsync(void *sym, const void *s, size_t n, size_t off, hipMemcpyKind, hipStream_t = nullptr) { memcpy((char *)sym + off, s, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = (hipStream_t)malloc(8); return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0, *hi = 0; return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, sizeof(wave_emu_event)); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
double wave_emu_now();
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = wave_emu_now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; } // (launches run to completion inside the launch call)
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t - a->t) * 1e3); return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "wave_emu"); strcpy(p->gcnArchName, "gfx950(emulated)"); p->multiProcessorCount = 8; p->clockRate = 2400000; p->totalGlobalMem = (size_t)8 << 30; return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void *, int, int) { return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int *ok, int, int) { *ok = 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *fr = *tot = (size_t)8 << 30; return hipSuccess; }

namespace wave_emu {
template <typename K, typename... A>
inline void launch_kernel(dim3 grid, dim3 block, size_t shmem, K kernel, A... args) // arguments are evaluated once, at the launch, and passed by value like kernel arguments
{
	launch(grid, block, shmem, [&]() { kernel(args...); });
}
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) wave_emu::launch_kernel(dim3(grid), dim3(block), (size_t)(shmem), kernel, ##__VA_ARGS__)

// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this part for the access patterns of OUR kernels (the guide calibrates only wide
// coalesced streaming reads: /opt/skills/guides/MI355X_MICROARCH.md "HBM").  Kernels with a KNOWN byte count, one per pattern; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- tools/build/pmc_calib     and again with WRITE_SIZE
// by tools/pmc_calib.py, which divides the reported KiB by the bytes printed here and keeps the factors under profiles/.
//   st_dword   : one dword per lane, 256 contiguous bytes per wave-instruction (the direction-ring stores of ksw_stream_kernel)
//   st_dwordx4 : 16 B per lane, 1 KiB per wave-instruction
//   st_byte    : one byte per lane, 64 contiguous bytes per wave-instruction (the lane-exact kernel's direction rows)
//   ld_dword / ld_dwordx4 / ld_qword : coalesced streaming reads of 4 / 16 / 8 B per lane
//   ld_random8 : one 8-byte load per lane at a random 8-aligned offset of a 4 GiB buffer (an index probe: bytes fetched per probe = sector size)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) st_dword(uint32_t *p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)i; }
__global__ void __launch_bounds__(256) st_dwordx4(u32x4 *p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { u32x4 v = { (uint32_t)i, 1, 2, 3 }; p[i] = v; } }
__global__ void __launch_bounds__(256) st_byte(uint8_t *p, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint8_t)i; }
__global__ void __launch_bounds__(256) ld_dword(const uint32_t *p, size_t n, uint32_t *out) { uint32_t a = 0; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a ^= p[i]; if (a == 0x12345u) out[0] = a; }
__global__ void __launch_bounds__(256) ld_qword(const uint64_t *p, size_t n, uint32_t *out) { uint64_t a = 0; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a ^= p[i]; if (a == 0x12345u) out[0] = (uint32_t)a; }
__global__ void __launch_bounds__(256) ld_dwordx4(const u32x4 *p, size_t n, uint32_t *out) { uint32_t a = 0; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const u32x4 v = p[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; } if (a == 0x12345u) out[0] = a; }
__global__ void __launch_bounds__(256) ld_random8(const uint64_t *p, size_t n_words, size_t n_probes, uint32_t *out)
{
	uint64_t a = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_probes; i += (size_t)gridDim.x * 256) {
		uint64_t h = (i + 1) * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
		a ^= p[h % n_words];
	}
	if (a == 0x12345u) out[0] = (uint32_t)a;
}

int main()
{
	const size_t GiB = (size_t)1 << 30, big = 4 * GiB;
	uint8_t *buf; uint32_t *out;
	CHECK(hipMalloc(&buf, big)); CHECK(hipMalloc(&out, 64));
	CHECK(hipMemset(buf, 1, big));
	CHECK(hipDeviceSynchronize());
	const int grid = 256 * 16;
	hipLaunchKernelGGL(st_dword, dim3(grid), dim3(256), 0, 0, (uint32_t *)buf, GiB / 4);
	hipLaunchKernelGGL(st_dwordx4, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, GiB / 16);
	hipLaunchKernelGGL(st_byte, dim3(grid), dim3(256), 0, 0, buf, GiB / 4);
	hipLaunchKernelGGL(ld_dword, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, GiB / 4, out);
	hipLaunchKernelGGL(ld_qword, dim3(grid), dim3(256), 0, 0, (const uint64_t *)buf, GiB / 8, out);
	hipLaunchKernelGGL(ld_dwordx4, dim3(grid), dim3(256), 0, 0, (const u32x4 *)buf, GiB / 16, out);
	const size_t n_probes = (size_t)1 << 26;
	hipLaunchKernelGGL(ld_random8, dim3(grid), dim3(256), 0, 0, (const uint64_t *)buf, big / 8, n_probes, out);
	CHECK(hipDeviceSynchronize());
	// what each kernel moved, for tools/pmc_calib.py: name, bytes written, bytes read (ld_random8: 8 useful bytes per probe; the sector the hardware moves is what we want to learn)
	printf("CALIB st_dword %zu 0\nCALIB st_dwordx4 %zu 0\nCALIB st_byte %zu 0\nCALIB ld_dword 0 %zu\nCALIB ld_qword 0 %zu\nCALIB ld_dwordx4 0 %zu\nCALIB ld_random8 0 %zu\n",
	       GiB, GiB, GiB / 4, GiB, GiB, GiB, n_probes * 8);
	return 0;
}

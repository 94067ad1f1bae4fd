// Does an asynchronous device-to-host copy on one stream wait for a long kernel on ANOTHER stream?  (round 6: the CIGAR copy-back of the host region path stalled for as long as the
// other lane's persistent splice kernels ran.)  A kernel that fills every CU (by wave slots, by LDS, or by registers) spins for ~300 ms on stream A; 20 ms later stream B copies
// 16 MB / 64 KB to pinned memory, with and without a small kernel of its own in front.   hipcc --offload-arch=gfx950 -O2 -o copy_behind_kernel tools/copy_behind_kernel.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int LDS_BYTES>
__global__ void __launch_bounds__(256) spin_kernel(long long cycles, int *sink)
{
	__shared__ int s[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
	s[threadIdx.x % (LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1)] = threadIdx.x;
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) { }
	if (sink && threadIdx.x == 0 && s[0] == 12345) *sink = 1;
}
__global__ void tiny_kernel(int *p) { if (p && threadIdx.x == 0) p[1] = 2; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
	setenv("GPU_MAX_HW_QUEUES", "16", 0); // as the library does
	hipStream_t a, b;
	CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
	int *d_sink; char *d_buf, *h_buf, *h_chunk;
	CK(hipMalloc(&d_sink, 64));
	CK(hipMalloc(&d_buf, 64 << 20));
	CK(hipHostMalloc(&h_buf, 64 << 20));
	CK(hipHostMalloc(&h_chunk, (size_t)1 << 30)); // a sub-range of a large pinned block, as the arenas hand out
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int n_cu = prop.multiProcessorCount;
	const long long cycles = 30000000; // wall_clock64 ticks at 100 MHz: 300 ms
	for (int fill = 0; fill < 3; ++fill) // 0: 8 blocks of 256 per CU (every wave slot), 1: 4 blocks with 40 KB of LDS each, 2: nothing running
		for (int tiny = 0; tiny < 2; ++tiny)
			for (int which = 0; which < 3; ++which) { // 0: 16 MB to h_buf, 1: 64 KB to h_buf, 2: 16 MB into the 1-GB block at an offset
				CK(hipDeviceSynchronize());
				const double t0 = now();
				if (fill == 0) hipLaunchKernelGGL(spin_kernel<0>, dim3(n_cu * 8), dim3(256), 0, a, cycles, d_sink);
				else if (fill == 1) hipLaunchKernelGGL(spin_kernel<40960>, dim3(n_cu * 4), dim3(256), 0, a, cycles, d_sink);
				std::this_thread::sleep_for(std::chrono::milliseconds(20));
				const double t1 = now();
				if (tiny) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, b, d_sink);
				const size_t bytes = which == 1 ? (64 << 10) : (16 << 20);
				CK(hipMemcpyAsync(which == 2 ? h_chunk + (300 << 20) : h_buf, d_buf, bytes, hipMemcpyDeviceToHost, b));
				const double t2 = now();
				CK(hipStreamSynchronize(b));
				const double t3 = now();
				CK(hipStreamSynchronize(a));
				const double t4 = now();
				printf("other stream: %-28s own kernel first: %d  copy %8zu B%s: call %.3f ms, done after %.3f ms (the other stream's kernel ends at %.1f ms)\n",
				       fill == 0 ? "all wave slots taken" : fill == 1 ? "all LDS taken" : "idle", tiny, bytes, which == 2 ? " (into a 1-GB pinned block)" : "", (t2 - t1) * 1e3, (t3 - t1) * 1e3, (t4 - t0) * 1e3);
			}
	// ---- which streams share a hardware queue?  stream 0 runs the long kernel; stream k copies, records an event and polls it (what stream_wait does) ----
	{
		hipStream_t st[24];
		for (int k = 0; k < 24; ++k) CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
		hipEvent_t ev;
		CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		for (int pass = 0; pass < 2; ++pass) // pass 1: every stream has been used before
			for (int k = 1; k < 24; ++k) {
				CK(hipDeviceSynchronize());
				hipLaunchKernelGGL(spin_kernel<0>, dim3(n_cu * 8), dim3(256), 0, st[0], (long long)10000000, d_sink); // 100 ms
				std::this_thread::sleep_for(std::chrono::milliseconds(10));
				const double t1 = now();
				CK(hipMemcpyAsync(h_buf, d_buf, 1 << 20, hipMemcpyDeviceToHost, st[k]));
				CK(hipEventRecord(ev, st[k]));
				while (hipEventQuery(ev) == hipErrorNotReady) { }
				const double t2 = now();
				printf("pass %d: long kernel on stream 0, copy + event on stream %2d: the event is reached after %7.3f ms%s\n", pass, k, (t2 - t1) * 1e3, (t2 - t1) > 0.02 ? "   <-- waited for the other stream's kernel" : "");
			}
	}
	// ---- as in the library: stream B has run kernels of its own (finished), THEN stream A starts its long kernel, then B copies and waits on an event ----
	{
		hipStream_t sa, sb;
		CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
		CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
		hipEvent_t ev;
		CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
		char *d_big;
		CK(hipMalloc(&d_big, (size_t)4 << 30));
		for (int fill = 0; fill < 2; ++fill)
			for (size_t bytes : { (size_t)8, (size_t)4 << 20, (size_t)40 << 20, (size_t)200 << 20 }) {
				CK(hipDeviceSynchronize());
				hipLaunchKernelGGL(tiny_kernel, dim3(64), dim3(64), 0, sb, d_sink);
				CK(hipMemcpyAsync(h_buf, d_sink, 8, hipMemcpyDeviceToHost, sb));
				CK(hipEventRecord(ev, sb));
				while (hipEventQuery(ev) == hipErrorNotReady) { }
				if (fill == 0) hipLaunchKernelGGL(spin_kernel<0>, dim3(n_cu * 8), dim3(256), 0, sa, (long long)20000000, d_sink); // 200 ms, every wave slot
				else hipLaunchKernelGGL(spin_kernel<40960>, dim3(n_cu * 4), dim3(256), 0, sa, (long long)20000000, d_sink);
				std::this_thread::sleep_for(std::chrono::milliseconds(10));
				const double t1 = now();
				CK(hipMemcpyAsync(h_chunk + (100 << 20), d_big + ((size_t)1 << 30), bytes, hipMemcpyDeviceToHost, sb));
				CK(hipEventRecord(ev, sb));
				while (hipEventQuery(ev) == hipErrorNotReady) { }
				const double t2 = now();
				printf("B ran kernels before; A's long kernel (%s) running; B copies %9zu B and waits on an event: %8.3f ms\n", fill == 0 ? "all wave slots" : "all LDS", bytes, (t2 - t1) * 1e3);
			}
	}
	return 0;
}

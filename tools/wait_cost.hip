// What does waiting for the GPU cost in CPU seconds?  A kernel that runs for ~T ms is launched N times; the host waits for each launch
//   (a) hipStreamSynchronize, (b) hipEventSynchronize on a hipEventBlockingSync event, (c) hipEventQuery polled with nanosleep(P us),
//   (d) the same from K threads at once on K streams (what five lane drivers do).
// Prints thread CPU seconds (CLOCK_THREAD_CPUTIME_ID) against wall seconds per method.  Measurement scaffolding.
//   hipcc --offload-arch=gfx950 -O2 -o wait_cost tools/wait_cost.hip -lpthread && ./wait_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin_kernel(long long cycles, int *out)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) {}
	if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}

static double thr_cpu() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
static double wall() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

enum Method { STREAM_SYNC, EVENT_BLOCKING, EVENT_DEFAULT, POLL };

static void run(Method m, int n, long long cycles, int poll_us, double *cpu, double *wl)
{
	hipStream_t s;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	hipEvent_t ev;
	CK(hipEventCreateWithFlags(&ev, (m == EVENT_BLOCKING ? hipEventBlockingSync : 0) | hipEventDisableTiming));
	hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, 1000, nullptr);
	CK(hipStreamSynchronize(s));
	const double c0 = thr_cpu(), w0 = wall();
	for (int i = 0; i < n; ++i) {
		hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, cycles, nullptr);
		if (m == STREAM_SYNC) CK(hipStreamSynchronize(s));
		else if (m == EVENT_BLOCKING || m == EVENT_DEFAULT) { CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); }
		else {
			CK(hipEventRecord(ev, s));
			while (hipEventQuery(ev) == hipErrorNotReady) { timespec ts = { 0, poll_us * 1000L }; nanosleep(&ts, nullptr); }
		}
	}
	*cpu = thr_cpu() - c0, *wl = wall() - w0;
	CK(hipEventDestroy(ev));
	CK(hipStreamDestroy(s));
}

int main(int argc, char **argv)
{
	const double ms = argc > 1 ? atof(argv[1]) : 20.0;
	const int n = argc > 2 ? atoi(argv[2]) : 25;
	const long long cycles = (long long)(ms * 1e-3 * 100e6); // wall_clock64 ticks at 100 MHz
	const char *names[] = { "hipStreamSynchronize", "hipEventSynchronize (blocking event)", "hipEventSynchronize (default event)", "hipEventQuery + nanosleep" };
	for (int threads : { 1, 5 }) {
		for (int m = 0; m < 4; ++m)
			for (int poll_us : { 50, 200, 1000 }) {
				if (m != POLL && poll_us != 50) continue;
				std::vector<double> cpu(threads), wl(threads);
				std::vector<std::thread> th;
				for (int t = 0; t < threads; ++t) th.emplace_back([&, t] { CK(hipSetDevice(0)); run((Method)m, n, cycles, poll_us, &cpu[t], &wl[t]); });
				for (auto &t : th) t.join();
				double c = 0, w = 0;
				for (int t = 0; t < threads; ++t) c += cpu[t], w += wl[t];
				printf("%d thread(s), %d kernels of %.0f ms each, %-40s%s: thread CPU %.3f s of %.3f s waited (%.1f %%)\n", threads, n, ms, names[m],
				       m == POLL ? (poll_us == 50 ? " 50 us" : poll_us == 200 ? " 200 us" : " 1 ms") : "", c, w, 100.0 * c / w);
			}
	}
	return 0;
}

// What does device / pinned memory cost to allocate on this box: per call or per byte?  (the first batches of a process spend
// ~2 s in ~260 hipMalloc + ~90 hipHostMalloc calls; an arena only helps if the cost is per call.)
//   hipcc --offload-arch=gfx950 -O2 -o tools/build/alloc_cost tools/alloc_cost.hip && tools/build/alloc_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
	(void)hipFree(nullptr);
	const size_t GB = (size_t)1 << 30;
	for (int pinned = 0; pinned < 2; ++pinned) {
		const size_t total = pinned ? 8 * GB : 32 * GB;
		for (size_t piece : { total, total / 16, total / 256, total / 2048 }) {
			std::vector<void *> p(total / piece);
			const double t0 = now();
			for (auto &q : p) { if ((pinned ? hipHostMalloc(&q, piece, hipHostMallocDefault) : hipMalloc(&q, piece)) != hipSuccess) { printf("allocation of %zu failed\n", piece); return 1; } }
			const double t1 = now();
			for (auto &q : p) { if (pinned) (void)hipHostFree(q); else (void)hipFree(q); }
			const double t2 = now();
			printf("%s: %5zu x %9.1f MB: alloc %8.1f ms (%7.3f ms per call, %6.2f ms per GB)   free %8.1f ms\n", pinned ? "hipHostMalloc" : "hipMalloc    ", p.size(), piece / 1048576.0,
			       (t1 - t0) * 1e3, (t1 - t0) * 1e3 / p.size(), (t1 - t0) * 1e3 / (total / (double)GB), (t2 - t1) * 1e3);
		}
	}
	return 0;
}

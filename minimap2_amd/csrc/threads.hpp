// Minimal fork-join helper for the host stages (the reference uses kt_for, kthread.c:54-72).
#pragma once
#include <atomic>
#include <functional>
#include <thread>
#include <vector>
#include <exception>
#include <mutex>

namespace mm2amd {

// Runs fn(i, tid) for i in [0,n) on n_threads threads with dynamic chunking; rethrows the first exception.
inline void parallel_for(int n_threads, long n, const std::function<void(long, int)> &fn, long chunk = 16)
{
	if (n <= 0) return;
	if (n_threads <= 1 || n <= chunk) { for (long i = 0; i < n; ++i) fn(i, 0); return; }
	std::atomic<long> next(0);
	std::exception_ptr err;
	std::mutex mu;
	auto worker = [&](int tid) {
		try {
			for (;;) {
				const long b = next.fetch_add(chunk);
				if (b >= n) break;
				const long e = b + chunk < n ? b + chunk : n;
				for (long i = b; i < e; ++i) fn(i, tid);
			}
		} catch (...) {
			std::lock_guard<std::mutex> lk(mu);
			if (!err) err = std::current_exception();
			next.store(n);
		}
	};
	const int nt = (int)std::min<long>(n_threads, (n + chunk - 1) / chunk);
	std::vector<std::thread> th;
	for (int t = 1; t < nt; ++t) th.emplace_back(worker, t);
	worker(0);
	for (auto &t : th) t.join();
	if (err) std::rethrow_exception(err);
}

} // namespace mm2amd

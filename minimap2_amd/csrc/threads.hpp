// Fork-join helper for the host stages (the reference uses kt_for, kthread.c:54-72), backed by a process-wide pool of
// persistent worker threads.  The mapper issues a dozen short parallel loops per sub-batch from several driver threads at
// once, so dispatch has to be cheap: jobs live in a small fixed table of slots that workers poll with plain atomics; a worker
// spins for a short while after running out of work and only then sleeps on a condition variable (one wake-up per idle period,
// not one mutex hand-off per loop per thread).
#pragma once
#include <atomic>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <pthread.h>
#include <time.h>

namespace mm2amd {

// names this thread for /proc/<pid>/task/*/comm (bench.py attributes the process's CPU seconds to thread groups by these names)
inline void name_thread(const char *name) { pthread_setname_np(pthread_self(), name); }

class ThreadPool {
public:
	// pool 0: the mapper's host stages; pool 1: the pipeline's other steps (hand-over packing, output formatting), which run BESIDE
	// the mapping of another batch and must not take its workers (a mapping loop waiting behind millisecond-long formatting chunks
	// leaves the GPU idle)
	static ThreadPool &instance(int which = 0)
	{
		static ThreadPool main_pool(false), side_pool(true);
		return which ? side_pool : main_pool;
	}
	explicit ThreadPool(bool side) : side_(side) {}
	// Runs fn(i, tid) for i in [0,n) on up to n_threads threads (the caller is one of them); tid < n_threads.
	void run(int n_threads, long n, const std::function<void(long, int)> &fn, long chunk)
	{
		if (n <= 0) return;
		if (n_threads <= 1 || n <= chunk) { for (long i = 0; i < n; ++i) fn(i, 0); return; }
		const int want = (int)std::min<long>(n_threads, (n + chunk - 1) / chunk);
		ensure_workers(want - 1);
		Slot *s = nullptr;
		for (;;) { // claim a free slot (there are more slots than driver threads)
			for (Slot &c : slots_) { bool f = false; if (c.in_use.compare_exchange_strong(f, true, std::memory_order_acquire)) { s = &c; break; } }
			if (s) break;
			std::this_thread::yield();
		}
		s->fn = &fn, s->n = n, s->chunk = chunk, s->err = nullptr;
		s->next.store(0, std::memory_order_relaxed), s->next_tid.store(1, std::memory_order_relaxed);
		const int helpers = std::min<int>(want - 1, n_workers_.load());
		s->helpers_wanted.store(helpers, std::memory_order_relaxed);
		s->open.store(true, std::memory_order_release);
		n_wanted_.fetch_add(helpers, std::memory_order_release);
		if (n_sleeping_.load(std::memory_order_acquire) > 0) { std::lock_guard<std::mutex> lk(mu_); if (helpers >= n_sleeping_.load()) cv_.notify_all(); else for (int k = 0; k < helpers; ++k) cv_.notify_one(); }
		work(*s, 0);
		// the range is exhausted: no helper can claim another chunk; wait for those still inside their last one
		s->open.store(false, std::memory_order_release);
		{ // helpers that never came are no longer wanted
			const int left = s->helpers_wanted.exchange(0, std::memory_order_acq_rel);
			if (left > 0) n_wanted_.fetch_sub(left, std::memory_order_release);
		}
		for (int spin = 0; s->active.load(std::memory_order_acquire) != 0; ++spin) { // (a helper's last chunk can be milliseconds long: do not pay for the wait out of the CPU quota)
			if (spin < 2000) cpu_relax();
			else { timespec ts = { 0, 20000 }; nanosleep(&ts, nullptr); }
		}
		std::exception_ptr err = s->err;
		s->err = nullptr;
		s->in_use.store(false, std::memory_order_release);
		if (err) std::rethrow_exception(err);
	}
	~ThreadPool()
	{
		{ std::lock_guard<std::mutex> lk(mu_); stop_.store(true); }
		cv_.notify_all();
		for (auto &t : workers_) t.join();
	}
private:
	struct alignas(128) Slot {
		std::atomic<bool> in_use{false}, open{false};
		const std::function<void(long, int)> *fn = nullptr;
		long n = 0, chunk = 1;
		alignas(64) std::atomic<long> next{0};
		alignas(64) std::atomic<int> next_tid{1};
		std::atomic<int> helpers_wanted{0}, active{0};
		std::exception_ptr err;
		std::mutex err_mu;
	};
	static void cpu_relax()
	{
#if defined(__x86_64__)
		_mm_pause();
#else
		std::this_thread::yield();
#endif
	}
	void work(Slot &j, int tid)
	{
		try {
			const long n = j.n, chunk = j.chunk;
			for (;;) {
				const long b = j.next.fetch_add(chunk, std::memory_order_relaxed);
				if (b >= n) break;
				const long e = b + chunk < n ? b + chunk : n;
				for (long i = b; i < e; ++i) (*j.fn)(i, tid);
			}
		} catch (...) {
			std::lock_guard<std::mutex> lk(j.err_mu);
			if (!j.err) j.err = std::current_exception();
			j.next.store(j.n);
		}
	}
	bool try_help()
	{
		bool helped = false;
		for (Slot &s : slots_) {
			if (!s.open.load(std::memory_order_acquire) || s.helpers_wanted.load(std::memory_order_relaxed) <= 0) continue;
			s.active.fetch_add(1, std::memory_order_acq_rel); // announce first, then re-check: run() closes before it waits for active == 0
			if (s.open.load(std::memory_order_acquire)) {
				// take one of the places still wanted (run() zeroes the count when it closes; never below zero, so that n_wanted_ stays the sum)
				int w = s.helpers_wanted.load(std::memory_order_relaxed);
				while (w > 0 && !s.helpers_wanted.compare_exchange_weak(w, w - 1, std::memory_order_acq_rel)) {}
				if (w > 0) {
					n_wanted_.fetch_sub(1, std::memory_order_release);
					work(s, s.next_tid.fetch_add(1, std::memory_order_relaxed));
					helped = true;
				}
			}
			s.active.fetch_sub(1, std::memory_order_acq_rel);
		}
		return helped;
	}
	void ensure_workers(int n)
	{
		if (n_workers_.load(std::memory_order_acquire) >= n) return;
		std::lock_guard<std::mutex> lk(mu_);
		const int cap = std::max(1u, std::thread::hardware_concurrency());
		if (n > cap) n = cap;
		while ((int)workers_.size() < n) workers_.emplace_back([this] { loop(); });
		n_workers_.store((int)workers_.size(), std::memory_order_release);
	}
	void loop()
	{
		name_thread(side_ ? "mm2side" : "mm2pool");
		for (;;) {
			if (try_help()) continue;
			// nothing to do: spin briefly (the next loop of the same sub-batch is usually microseconds away), then sleep.  "Something to do" = a
			// loop that still wants helpers (round 4: it used to be "a loop is open", and idle workers polled and yielded for as long as any
			// driver was inside a loop's last chunks -- CPU seconds out of the quota that the host stages needed)
			bool found = false;
			for (int spin = 0; spin < 400 && !found; ++spin) { // (a few microseconds: spinning is paid for out of the process's CPU quota)
				if (stop_.load(std::memory_order_relaxed)) return;
				if (n_wanted_.load(std::memory_order_acquire) > 0) found = true; else cpu_relax();
			}
			if (found) continue;
			std::unique_lock<std::mutex> lk(mu_);
			n_sleeping_.fetch_add(1, std::memory_order_acq_rel);
			cv_.wait(lk, [&] { return stop_.load() || n_wanted_.load(std::memory_order_acquire) > 0; });
			n_sleeping_.fetch_sub(1, std::memory_order_acq_rel);
			if (stop_.load()) return;
		}
	}
	Slot slots_[64]; // more than the driver threads a context can have at once (16 replicas x 5 lanes would queue; 8 x 5 fit)
	std::atomic<int> n_wanted_{0}, n_sleeping_{0}, n_workers_{0}; // n_wanted_: helpers the open loops still want, over all slots
	const bool side_;
	std::atomic<bool> stop_{false};
	std::mutex mu_;
	std::condition_variable cv_;
	std::vector<std::thread> workers_;
};

// Runs fn(i, tid) for i in [0,n) on n_threads threads with dynamic chunking; rethrows the first exception.
inline void parallel_for(int n_threads, long n, const std::function<void(long, int)> &fn, long chunk = 16)
{
	ThreadPool::instance(0).run(n_threads, n, fn, chunk);
}
// the same on the pool of the pipeline's side steps (hand-over, output stage)
inline void parallel_for_side(int n_threads, long n, const std::function<void(long, int)> &fn, long chunk = 16)
{
	ThreadPool::instance(1).run(n_threads, n, fn, chunk);
}

} // namespace mm2amd

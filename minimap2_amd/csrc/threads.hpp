// Fork-join helper for the host stages (the reference uses kt_for, kthread.c:54-72), backed by a process-wide pool of
// persistent worker threads: the mapper issues a dozen parallel loops per sub-batch and spawning hundreds of threads for each
// costs more than some of the loops themselves.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace mm2amd {

class ThreadPool {
public:
	static ThreadPool &instance()
	{
		static ThreadPool p;
		return p;
	}
	// Runs fn(i, tid) for i in [0,n) on up to n_threads threads (the caller is one of them); tid < n_threads.
	void run(int n_threads, long n, const std::function<void(long, int)> &fn, long chunk)
	{
		if (n <= 0) return;
		if (n_threads <= 1 || n <= chunk) { for (long i = 0; i < n; ++i) fn(i, 0); return; }
		const int want = (int)std::min<long>(n_threads, (n + chunk - 1) / chunk);
		auto job = std::make_shared<Job>();
		job->fn = &fn, job->n = n, job->chunk = chunk, job->next_tid.store(1);
		{
			std::lock_guard<std::mutex> lk(mu_);
			ensure_workers(want - 1);
			job->helpers_wanted = std::min<int>(want - 1, (int)workers_.size());
			if (job->helpers_wanted > 0) queue_.push_back(job);
		}
		cv_.notify_all();
		work(*job, 0);
		{ // no new helper may join once the caller has drained the range; wait for the ones still inside
			std::unique_lock<std::mutex> lk(mu_);
			job->closed = true;
			for (auto it = queue_.begin(); it != queue_.end(); ++it) if (it->get() == job.get()) { queue_.erase(it); break; }
			done_cv_.wait(lk, [&] { return job->active == 0; });
		}
		if (job->err) std::rethrow_exception(job->err);
	}
	~ThreadPool()
	{
		{ std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
		cv_.notify_all();
		for (auto &t : workers_) t.join();
	}
private:
	struct Job {
		const std::function<void(long, int)> *fn = nullptr;
		long n = 0, chunk = 1;
		std::atomic<long> next{0};
		std::atomic<int> next_tid{1};
		int helpers_wanted = 0, active = 0; // guarded by mu_
		bool closed = false;                // guarded by mu_
		std::exception_ptr err;
		std::mutex err_mu;
	};
	void work(Job &j, int tid)
	{
		try {
			for (;;) {
				const long b = j.next.fetch_add(j.chunk);
				if (b >= j.n) break;
				const long e = b + j.chunk < j.n ? b + j.chunk : j.n;
				for (long i = b; i < e; ++i) (*j.fn)(i, tid);
			}
		} catch (...) {
			std::lock_guard<std::mutex> lk(j.err_mu);
			if (!j.err) j.err = std::current_exception();
			j.next.store(j.n);
		}
	}
	void ensure_workers(int n) // mu_ held
	{
		const int cap = std::max(1u, std::thread::hardware_concurrency());
		if (n > cap) n = cap;
		while ((int)workers_.size() < n) workers_.emplace_back([this] { loop(); });
	}
	void loop()
	{
		std::unique_lock<std::mutex> lk(mu_);
		for (;;) {
			cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
			if (stop_) return;
			std::shared_ptr<Job> j = queue_.front();
			if (j->closed || j->helpers_wanted <= 0) { queue_.pop_front(); continue; }
			if (--j->helpers_wanted == 0) queue_.pop_front();
			++j->active;
			const int tid = j->next_tid.fetch_add(1);
			lk.unlock();
			work(*j, tid);
			lk.lock();
			if (--j->active == 0) done_cv_.notify_all();
		}
	}
	std::mutex mu_;
	std::condition_variable cv_, done_cv_;
	std::deque<std::shared_ptr<Job>> queue_;
	std::vector<std::thread> workers_;
	bool stop_ = false;
};

// Runs fn(i, tid) for i in [0,n) on n_threads threads with dynamic chunking; rethrows the first exception.
inline void parallel_for(int n_threads, long n, const std::function<void(long, int)> &fn, long chunk = 16)
{
	ThreadPool::instance().run(n_threads, n, fn, chunk);
}

} // namespace mm2amd

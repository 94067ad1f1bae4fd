// Optional host-side stage trace (MM2AMD_TRACE=<file>): (lane, stage, begin, end) records for pipeline debugging.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace mm2amd {

struct TraceRec { int lane; const char *stage; double t0, t1; };

class Trace {
public:
	static Trace &get() { static Trace t; return t; }
	bool on() const { return path_ != nullptr; }
	static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
	void add(int lane, const char *stage, double t0, double t1)
	{
		if (!path_) return;
		std::lock_guard<std::mutex> lk(mu_);
		recs_.push_back(TraceRec{lane, stage, t0, t1});
	}
	void flush()
	{
		if (!path_) return;
		std::lock_guard<std::mutex> lk(mu_);
		if (FILE *f = fopen(path_, "a")) {
			for (const TraceRec &r : recs_) fprintf(f, "%d\t%s\t%.6f\t%.6f\n", r.lane, r.stage, r.t0, r.t1);
			fclose(f);
		}
		recs_.clear();
	}
private:
	Trace() : path_(getenv("MM2AMD_TRACE")) {}
	const char *path_;
	std::mutex mu_;
	std::vector<TraceRec> recs_;
};

struct TraceScope {
	int lane; const char *stage; double t0;
	TraceScope(int l, const char *s) : lane(l), stage(s), t0(Trace::get().on() ? Trace::now() : 0) {}
	~TraceScope() { if (Trace::get().on()) Trace::get().add(lane, stage, t0, Trace::now()); }
};

} // namespace mm2amd

// Cycle accounting of the host stages' inner pieces (MM2AMD_HOST_PROF=1 prints the table when the mapping context is dropped): the GPU box
// gives the process a CPU quota, so a batch's wall time is bounded below by its host CPU seconds -- this is how they are attributed.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#if defined(__x86_64__)
#include <x86intrin.h>
#endif

namespace mm2amd {
namespace hostprof {
enum Piece { Q4_ENCODE, CHAINS_TO_HITS, PLAN_REGION, ADD_JOBS, CONSUME_WINDOWS, GETSEQ, FIX_CIGAR, EXTRA_SCAN, FINISH_READ, HAND_OVER, KSW_CLASSIFY, KSW_SCATTER, KSW_UNPERM, FORMAT_RANGE, GEN_REGS, PARENT_SELECT, EST_ERR, BEGIN_READ, COMPLETE_FINISHED, N_PIECES };
inline const char *name(int k)
{
	static const char *n[N_PIECES] = {"q4_encode", "chains_to_hits", "plan_region", "add_jobs", "consume_windows", "getseq", "fix_cigar", "extra_scan", "finish_read", "hand_over",
	                                  "ksw_classify", "ksw_scatter", "ksw_unperm", "format_range", "gen_regs", "parent_select", "est_err", "begin_read", "complete_finished"};
	return n[k];
}
struct Table {
	std::atomic<uint64_t> cyc[N_PIECES], cnt[N_PIECES];
	bool on;
	Table() : on(getenv("MM2AMD_HOST_PROF") != nullptr) { for (int k = 0; k < N_PIECES; ++k) cyc[k] = 0, cnt[k] = 0; }
	~Table() { report(); }
	void report()
	{
		if (!on) return;
		for (int k = 0; k < N_PIECES; ++k)
			if (cnt[k]) fprintf(stderr, "[mm2amd] host piece %-16s %12.0f kcycles  %10llu calls  %8.0f cycles/call\n", name(k), cyc[k] * 1e-3, (unsigned long long)cnt[k].load(), (double)cyc[k] / cnt[k]);
	}
};
inline Table &table() { static Table t; return t; }
// Every thread adds into counters of its own (a shared table's cache lines bounced between 16 threads and cost more CPU than some of the pieces
// it measured); a thread's counters are folded into the table when the thread ends and when the report is printed from it.
struct Local {
	uint64_t cyc[N_PIECES] = {0}, cnt[N_PIECES] = {0};
	void flush() { for (int k = 0; k < N_PIECES; ++k) if (cnt[k]) table().cyc[k].fetch_add(cyc[k], std::memory_order_relaxed), table().cnt[k].fetch_add(cnt[k], std::memory_order_relaxed), cyc[k] = cnt[k] = 0; }
	~Local() { flush(); }
};
inline Local &local() { static thread_local Local l; return l; }
struct Scope {
	int k; uint64_t t0 = 0; bool on;
	explicit Scope(int piece) : k(piece), on(table().on)
	{
#if defined(__x86_64__)
		if (on) t0 = __rdtsc();
#endif
	}
	void stop()
	{
#if defined(__x86_64__)
		if (on) {
			Local &l = local();
			l.cyc[k] += __rdtsc() - t0, ++l.cnt[k];
			if ((l.cnt[k] & 1023) == 0) l.flush(); // (pool threads live as long as the process: fold now and then)
		}
#endif
		on = false;
	}
	~Scope() { stop(); }
};
} // namespace hostprof
} // namespace mm2amd

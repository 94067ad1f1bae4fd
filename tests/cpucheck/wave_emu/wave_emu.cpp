// TEST INFRASTRUCTURE -- the runtime behind tests/cpucheck/wave_emu/hip/hip_runtime.h: HIP threads as fibers, wavefronts as groups
// of 64 fibers that meet at cross-lane operations, blocks of a launch spread over host threads.  x86-64 only (the context switch
// is 12 instructions of assembly; ucontext would make a system call per switch and the kernels switch millions of times).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <vector>
#include <sys/mman.h>

extern "C" void wave_emu_switch(void **save_sp, void *to_sp);
asm(R"(
	.text
	.globl wave_emu_switch
	.type wave_emu_switch,@function
wave_emu_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq %rsi, %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
	.size wave_emu_switch,.-wave_emu_switch
)");

double wave_emu_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

namespace wave_emu {

namespace {

constexpr size_t kStack = 256 * 1024;

struct Wave {
	uint64_t live = 0;            // lanes that have not returned
	uint64_t arrived = 0;         // lanes waiting in the pending operation
	int op = OP_NONE, width = 64;
	int64_t arg[64];
	uint64_t val[64];
	uint64_t res[2][64];          // results of the last two operations (a lane can be one operation ahead of the slowest)
	uint64_t gen = 0;             // completed operations
};

struct Lane {
	void *sp = nullptr;
	char *stack = nullptr;
	LaneView view;
	int index = 0;                // thread index within the block
	bool done = false;
	// what the lane waits for: a wave operation (gen to exceed wait_gen) or the block barrier (bar_gen to exceed wait_bar)
	bool wait_wave = false, wait_block = false;
	uint64_t wait_gen = 0, wait_bar = 0;
};

struct Worker { // one per host thread: the fibers of the block it is running
	std::vector<Lane> lanes;
	std::vector<Wave> waves;
	void *sched_sp = nullptr;
	Lane *cur = nullptr;
	const std::function<void()> *body = nullptr;
	std::vector<char> dyn;
	int n_threads = 0, n_live = 0, bar_arrived = 0;
	uint64_t bar_gen = 0;
};
thread_local Worker *tl_worker = nullptr;
thread_local LaneView tl_host_view; // outside a kernel (threadIdx read on the host): zeros

void yield_to_scheduler()
{
	Worker &w = *tl_worker;
	Lane *me = w.cur;
	wave_emu_switch(&me->sp, w.sched_sp);
}

void complete(Wave &wv)
{
	uint64_t *out = wv.res[wv.gen & 1];
	const uint64_t m = wv.arrived;
	switch (wv.op) {
	case OP_BARRIER: break;
	case OP_BALLOT: { uint64_t b = 0; for (int l = 0; l < 64; ++l) if ((m >> l & 1) && wv.val[l]) b |= 1ull << l; for (int l = 0; l < 64; ++l) out[l] = b; break; }
	case OP_FIRST: { const int f = __builtin_ctzll(m); for (int l = 0; l < 64; ++l) out[l] = wv.val[f]; break; }
	case OP_READLANE: for (int l = 0; l < 64; ++l) if (m >> l & 1) out[l] = wv.val[wv.arg[l] & 63]; break;
	case OP_SHFL: case OP_SHFL_UP: case OP_SHFL_DOWN: case OP_SHFL_XOR:
		for (int l = 0; l < 64; ++l) {
			if (!(m >> l & 1)) continue;
			const int wd = wv.width, base = l & ~(wd - 1), rel = l & (wd - 1);
			int src;
			if (wv.op == OP_SHFL) src = base + (int)(wv.arg[l] & (wd - 1));
			else if (wv.op == OP_SHFL_UP) src = rel - (int)wv.arg[l] >= 0 ? l - (int)wv.arg[l] : l;
			else if (wv.op == OP_SHFL_DOWN) src = rel + (int)wv.arg[l] < wd ? l + (int)wv.arg[l] : l;
			else src = (rel ^ (int)wv.arg[l]) < wd ? base + (rel ^ (int)wv.arg[l]) : l;
			out[l] = (m >> src & 1) ? wv.val[src] : wv.val[l]; // reading an inactive lane is undefined on the hardware; keep the lane's own value
		}
		break;
	case OP_DPP:
		for (int l = 0; l < 64; ++l) {
			if (!(m >> l & 1)) continue;
			const int ctrl = (int)wv.arg[l] & 0xffff, row_mask = (int)wv.arg[l] >> 16 & 0xf;
			const uint32_t old = (uint32_t)(wv.val[l] >> 32);
			int src = -1;
			if (!(row_mask >> (l >> 4) & 1)) { out[l] = old; continue; }
			if (ctrl == 0x138) src = l - 1;                                   // wave_shr:1
			else if (ctrl == 0x13c) src = (l + 63) & 63;                       // wave_ror:1
			else if (ctrl == 0x130) src = l + 1 < 64 ? l + 1 : -1;             // wave_shl:1
			else if (ctrl == 0x134) src = (l + 1) & 63;                        // wave_rol:1
			else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl & 15; src = (l & 15) >= n ? l - n : -1; } // row_shr:n
			else if (ctrl >= 0x101 && ctrl <= 0x10f) { const int n = ctrl & 15; src = (l & 15) + n < 16 ? l + n : -1; } // row_shl:n
			else if (ctrl == 0x142) src = l >= 16 ? (l & ~15) - 1 : -1;        // row_bcast:15: lane 15 of the row before
			else if (ctrl == 0x143) src = l >= 32 ? (l & ~31) - 1 : -1;        // row_bcast:31: lane 31 to the upper half
			else { fprintf(stderr, "[wave_emu] DPP control 0x%x is not emulated\n", ctrl); abort(); }
			out[l] = src >= 0 && (m >> src & 1) ? (uint32_t)wv.val[src] : old;
		}
		break;
	default: fprintf(stderr, "[wave_emu] unknown operation %d\n", wv.op); abort();
	}
	wv.arrived = 0, wv.op = OP_NONE;
	++wv.gen;
}

void lane_exit()
{
	Worker &w = *tl_worker;
	Lane *me = w.cur;
	me->done = true;
	--w.n_live;
	Wave &wv = w.waves[me->index >> 6];
	wv.live &= ~(1ull << (me->index & 63));
	if (wv.op != OP_NONE && wv.arrived == wv.live && wv.live) complete(wv); // the others were only waiting for this lane
	if (w.bar_arrived > 0 && w.bar_arrived == w.n_live) w.bar_arrived = 0, ++w.bar_gen;
	yield_to_scheduler();
	abort(); // never resumed
}

extern "C" void wave_emu_fiber_main()
{
	(*tl_worker->body)();
	lane_exit();
}

void prepare(Lane &ln)
{
	if (!ln.stack) {
		ln.stack = (char *)mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		if (ln.stack == MAP_FAILED) { perror("[wave_emu] mmap"); abort(); }
	}
	void **sp = (void **)(ln.stack + kStack);
	*--sp = nullptr;                          // return address of the entry function (never used); keeps rsp = 16n + 8 at its first instruction
	*--sp = (void *)&wave_emu_fiber_main;     // what wave_emu_switch's ret jumps to
	for (int i = 0; i < 6; ++i) *--sp = nullptr; // rbp rbx r12..r15
	ln.sp = sp;
	ln.done = false, ln.wait_wave = ln.wait_block = false;
}

void run_block(Worker &w, dim3 grid, dim3 block, dim3 bid, size_t dyn_bytes)
{
	const int n = (int)(block.x * block.y * block.z);
	if ((int)w.lanes.size() < n) w.lanes.resize(n);
	w.waves.assign((n + 63) / 64, Wave());
	if (w.dyn.size() < dyn_bytes + 64) w.dyn.resize(dyn_bytes + 64);
	w.n_threads = w.n_live = n, w.bar_arrived = 0, w.bar_gen = 0;
	for (int t = 0; t < n; ++t) {
		Lane &ln = w.lanes[t];
		prepare(ln);
		ln.index = t;
		ln.view.tid = dim3(t % block.x, t / block.x % block.y, t / (block.x * block.y));
		ln.view.bid = bid, ln.view.bdim = block, ln.view.gdim = grid;
		w.waves[t >> 6].live |= 1ull << (t & 63);
	}
	while (w.n_live > 0) {
		bool progress = false;
		for (int t = 0; t < n; ++t) {
			Lane &ln = w.lanes[t];
			if (ln.done) continue;
			if (ln.wait_wave) { if (w.waves[t >> 6].gen <= ln.wait_gen) continue; ln.wait_wave = false; }
			if (ln.wait_block) { if (w.bar_gen <= ln.wait_bar) continue; ln.wait_block = false; }
			w.cur = &ln;
			wave_emu_switch(&w.sched_sp, ln.sp);
			w.cur = nullptr;
			progress = true;
		}
		if (!progress) {
			fprintf(stderr, "[wave_emu] deadlock in block (%u,%u,%u): %d threads alive\n", bid.x, bid.y, bid.z, w.n_live);
			for (size_t v = 0; v < w.waves.size(); ++v)
				fprintf(stderr, "  wave %zu: live %016llx arrived %016llx pending op %d; block barrier: %d arrived\n", v, (unsigned long long)w.waves[v].live, (unsigned long long)w.waves[v].arrived, w.waves[v].op, w.bar_arrived);
			abort();
		}
	}
}

} // namespace

LaneView &here() { Worker *w = tl_worker; return w && w->cur ? w->cur->view : tl_host_view; }
void *dyn_shared() { Worker *w = tl_worker; return (void *)(((uintptr_t)w->dyn.data() + 63) & ~(uintptr_t)63); }

uint64_t collective(Op op, uint64_t v, int64_t arg, int width)
{
	Worker &w = *tl_worker;
	Lane *me = w.cur;
	const int l = me->index & 63;
	Wave &wv = w.waves[me->index >> 6];
	if (wv.op != OP_NONE && wv.op != op) {
		fprintf(stderr, "[wave_emu] divergent cross-lane operations in one wavefront: lane %d is in op %d while others wait in op %d (thread %d of block %u)\n", l, op, wv.op, me->index, me->view.bid.x);
		abort();
	}
	wv.op = op, wv.width = width, wv.val[l] = v, wv.arg[l] = arg, wv.arrived |= 1ull << l;
	const uint64_t g = wv.gen;
	if (wv.arrived == wv.live) complete(wv);
	else {
		me->wait_wave = true, me->wait_gen = g;
		yield_to_scheduler();
	}
	return wv.res[g & 1][l];
}

void block_barrier()
{
	Worker &w = *tl_worker;
	Lane *me = w.cur;
	const uint64_t g = w.bar_gen;
	if (++w.bar_arrived == w.n_live) { w.bar_arrived = 0, ++w.bar_gen; return; }
	me->wait_block = true, me->wait_bar = g;
	yield_to_scheduler();
}

int host_threads()
{
	static const int n = [] { const char *e = getenv("MM2AMD_EMU_THREADS"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : v; }();
	return n;
}

void launch(dim3 grid, dim3 block, size_t dyn_bytes, const std::function<void()> &body)
{
	const size_t n_blocks = (size_t)grid.x * grid.y * grid.z;
	if (n_blocks == 0) return;
	if (tl_worker && tl_worker->cur) { fprintf(stderr, "[wave_emu] kernel launch from inside a kernel\n"); abort(); }
	std::atomic<size_t> next(0);
	auto work = [&]() {
		static thread_local Worker me; // fiber stacks persist per host thread
		Worker *saved = tl_worker;
		tl_worker = &me;
		me.body = &body;
		for (;;) {
			const size_t b = next.fetch_add(1);
			if (b >= n_blocks) break;
			run_block(me, grid, block, dim3((unsigned)(b % grid.x), (unsigned)(b / grid.x % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y))), dyn_bytes);
		}
		tl_worker = saved;
	};
	const int nt = (int)std::min<size_t>((size_t)host_threads(), n_blocks);
	if (nt <= 1) { work(); return; }
	std::vector<std::thread> th;
	for (int t = 1; t < nt; ++t) th.emplace_back(work);
	work();
	for (auto &t : th) t.join();
}

} // namespace wave_emu

// MM2AMD_EMU_BACKTRACE=1: a crash inside the emulated product prints its C++ stack (development aid; run pytest with -p no:faulthandler)
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
void crash_backtrace(int sig)
{
	void *frames[64];
	const int n = backtrace(frames, 64);
	const char msg[] = "[wave_emu] fatal signal, stack:\n";
	(void)!write(2, msg, sizeof msg - 1);
	backtrace_symbols_fd(frames, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
struct CrashHandlers {
	CrashHandlers()
	{
		if (!getenv("MM2AMD_EMU_BACKTRACE")) return;
		static char alt[1 << 16]; // the faulting thread may be on a small fiber stack
		stack_t ss; ss.ss_sp = alt, ss.ss_size = sizeof alt, ss.ss_flags = 0;
		sigaltstack(&ss, nullptr);
		struct sigaction sa;
		sa.sa_handler = crash_backtrace, sigemptyset(&sa.sa_mask), sa.sa_flags = SA_ONSTACK;
		sigaction(SIGSEGV, &sa, nullptr), sigaction(SIGABRT, &sa, nullptr), sigaction(SIGBUS, &sa, nullptr);
	}
} crash_handlers;
}

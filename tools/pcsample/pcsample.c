// A minimal PC-sampling profiler for the host side (development aid; this image has no perf / gprof for shared objects): LD_PRELOAD it, every
// PCSAMPLE_US microseconds of process CPU time (ITIMER_PROF; default 1000) the interrupted thread's program counter is recorded; at exit the samples and
// /proc/self/maps go to $PCSAMPLE_OUT (default /tmp/pcsample.out).  tools/pcsample/report.py attributes them to functions (nm) and lines (addr2line).
#define _GNU_SOURCE
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <ucontext.h>
#include <stdint.h>

#define CAP (1u << 22)
static uint64_t *g_pc;
static volatile uint32_t g_n;

static void on_prof(int sig, siginfo_t *si, void *uc_)
{
	(void)sig, (void)si;
	const ucontext_t *uc = (const ucontext_t *)uc_;
	const uint32_t i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
	if (i < CAP) g_pc[i] = (uint64_t)uc->uc_mcontext.gregs[REG_RIP];
}

__attribute__((constructor)) static void start(void)
{
	g_pc = (uint64_t *)calloc(CAP, sizeof(uint64_t));
	struct sigaction sa;
	memset(&sa, 0, sizeof sa);
	sa.sa_sigaction = on_prof, sa.sa_flags = SA_SIGINFO | SA_RESTART;
	sigaction(SIGPROF, &sa, NULL);
	const long us = getenv("PCSAMPLE_US") ? atol(getenv("PCSAMPLE_US")) : 1000;
	struct itimerval it = { { 0, us }, { 0, us } };
	setitimer(ITIMER_PROF, &it, NULL);
}

__attribute__((destructor)) static void stop(void)
{
	struct itimerval it = { { 0, 0 }, { 0, 0 } };
	setitimer(ITIMER_PROF, &it, NULL);
	const char *path = getenv("PCSAMPLE_OUT") ? getenv("PCSAMPLE_OUT") : "/tmp/pcsample.out";
	FILE *fp = fopen(path, "w");
	if (!fp) return;
	FILE *maps = fopen("/proc/self/maps", "r");
	char line[1024];
	if (maps) { while (fgets(line, sizeof line, maps)) if (strstr(line, " r-xp ") || strstr(line, "r-xp")) fprintf(fp, "M %s", line); fclose(maps); }
	const uint32_t n = g_n < CAP ? g_n : CAP;
	for (uint32_t i = 0; i < n; ++i) fprintf(fp, "S %llx\n", (unsigned long long)g_pc[i]);
	fclose(fp);
}

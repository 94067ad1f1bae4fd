// Who allocates?  LD_PRELOAD: counts malloc / calloc / realloc calls by call site (return address) and writes the busiest to $MCOUNT_OUT (default
// /tmp/mcount.out) with /proc/self/maps, for tools/pcsample/report.py-style attribution.  Development aid.
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
extern void *__libc_malloc(size_t);
extern void *__libc_calloc(size_t, size_t);
extern void *__libc_realloc(void *, size_t);
#define NB (1u << 16)
static struct { uint64_t pc; uint64_t n, bytes; } g_tab[NB];
static void note(void *ra, size_t sz)
{
	uint64_t pc = (uint64_t)ra;
	uint32_t h = (uint32_t)((pc * 0x9E3779B97F4A7C15ull) >> 48);
	for (int k = 0; k < 64; ++k, h = (h + 1) & (NB - 1)) {
		uint64_t cur = __atomic_load_n(&g_tab[h].pc, __ATOMIC_RELAXED);
		if (cur == 0) { uint64_t z = 0; if (__atomic_compare_exchange_n(&g_tab[h].pc, &z, pc, 0, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) cur = pc; else cur = z; }
		if (cur == pc) { __atomic_fetch_add(&g_tab[h].n, 1, __ATOMIC_RELAXED); __atomic_fetch_add(&g_tab[h].bytes, sz, __ATOMIC_RELAXED); return; }
	}
}
void *malloc(size_t n) { note(__builtin_return_address(0), n); return __libc_malloc(n); }
void *calloc(size_t a, size_t b) { note(__builtin_return_address(0), a * b); return __libc_calloc(a, b); }
void *realloc(void *p, size_t n) { note(__builtin_return_address(0), n); return __libc_realloc(p, n); }
__attribute__((destructor)) static void stop(void)
{
	const char *path = getenv("MCOUNT_OUT") ? getenv("MCOUNT_OUT") : "/tmp/mcount.out";
	FILE *fp = fopen(path, "w");
	if (!fp) return;
	FILE *maps = fopen("/proc/self/maps", "r");
	char line[1024];
	if (maps) { while (fgets(line, sizeof line, maps)) if (strstr(line, "r-xp")) fprintf(fp, "M %s", line); fclose(maps); }
	for (uint32_t i = 0; i < NB; ++i) if (g_tab[i].n) fprintf(fp, "C %llx %llu %llu\n", (unsigned long long)g_tab[i].pc, (unsigned long long)g_tab[i].n, (unsigned long long)g_tab[i].bytes);
	fclose(fp);
}
// operator new / new[] (the C++ library's call malloc from inside libstdc++: count the caller instead)
void *_Znwm(size_t n) { note(__builtin_return_address(0), n); void *p = __libc_malloc(n ? n : 1); if (!p) abort(); return p; }
void *_Znam(size_t n) { note(__builtin_return_address(0), n); void *p = __libc_malloc(n ? n : 1); if (!p) abort(); return p; }

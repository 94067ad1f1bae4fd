// Device helpers shared by the register-resident DP kernels (ksw_fast.hip, ksw_splice.hip): DPP lane shifts, the packed
// 16-bit VOP3P instructions through inline asm (written as C++ vector code the optimiser rewrites the min/mul idioms back into
// compares and de-vectorises them), and the CIGAR run-length pusher.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mm2amd {

__device__ __forceinline__ int dpp_shr1(int carry_in, int v) // lane i <- v[i-1], lane 0 <- carry_in
{
	return __builtin_amdgcn_update_dpp(carry_in, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// inclusive prefix maximum over the lanes (lane 63 holds the wave's maximum): gfx9 DPP row shifts and row broadcasts, no LDS crossbar
__device__ __forceinline__ int32_t wave_prefix_max_i32_ext(int32_t v)
{
	int32_t o;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;
	o = __builtin_amdgcn_update_dpp(INT32_MIN, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;
	return v;
}

struct FastCig { uint32_t *c; int n; uint32_t last; };
__device__ __forceinline__ void fast_cig_push(FastCig &g, uint32_t op, int len) // ksw_push_cigar (ksw2.h:114-124)
{
	if (g.n == 0 || op != (g.last & 0xf)) {
		if (g.n > 0) g.c[g.n - 1] = g.last;
		g.last = (uint32_t)len << 4 | op;
		++g.n;
	} else g.last += (uint32_t)len << 4;
}

#ifdef MM2AMD_WAVE_EMU // tests/cpucheck/wave_emu: the instructions below as plain C++ on the two halves
} // namespace mm2amd
#include "ksw_pk_emu.hpp"
namespace mm2amd {
#else
#define MM2_PK2(name, ins) \
	__device__ __forceinline__ uint32_t name(uint32_t a, uint32_t b) { uint32_t r; asm(ins " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
MM2_PK2(pk_add, "v_pk_add_u16")
MM2_PK2(pk_sub, "v_pk_sub_u16")
MM2_PK2(pk_max, "v_pk_max_i16")
MM2_PK2(pk_min, "v_pk_min_i16")
MM2_PK2(pk_minu, "v_pk_min_u16")
MM2_PK2(pk_mul, "v_pk_mul_lo_u16")
__device__ __forceinline__ uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ uint32_t pk_shr2(uint32_t a) { uint32_t r; asm("v_pk_lshrrev_b16 %0, 2, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; } // the inline constant feeds both halves
__device__ __forceinline__ uint32_t pk2(int v) { return ((uint32_t)v & 0xffffu) | (uint32_t)v << 16; }
// a uniform constant pinned in a VGPR (the asm wrappers take VGPR operands; without this every use re-copies it from an SGPR)
__device__ __forceinline__ uint32_t pk2v(int v) { uint32_t r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(pk2(v))); return r; }
#endif
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); } // v_bfi_b32
__device__ __forceinline__ uint32_t dpp_shr1u(uint32_t carry_in, uint32_t v) { return (uint32_t)dpp_shr1((int)carry_in, (int)v); }


} // namespace mm2amd

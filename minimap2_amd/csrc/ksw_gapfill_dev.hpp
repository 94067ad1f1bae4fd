// Device helpers of the register-resident gap-fill kernels (ksw_gapfill.hip, ksw_stream.hip): packed 16-bit VOP3P with the operand
// kinds that keep VGPRs free, and the hand-ordered arithmetic of one register set and anti-diagonal (gf_cell).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "ksw_pk.hpp"

namespace mm2amd {

#ifndef MM2AMD_WAVE_EMU // (the emulator build takes these from tests/cpucheck/wave_emu/ksw_pk_emu.hpp, through ksw_pk.hpp)
// ---------------------------------------------------------------------------------------------------------
// Packed 16-bit VOP3P with the operand kinds that keep VGPRs free: small constants are inline constants (op_sel_hi clear on
// that operand: the low half feeds both lanes of the pair), launch-uniform scores sit in SGPRs (one constant-bus operand per
// instruction on gfx9).  Only the match score needs a VGPR copy (it meets a second SGPR operand in one instruction).
// ---------------------------------------------------------------------------------------------------------
#define GF_V_C(name, ins, cst) \
	__device__ __forceinline__ uint32_t name(uint32_t a) { uint32_t r; asm(ins " %0, %1, " #cst " op_sel_hi:[1,0]" : "=v"(r) : "v"(a)); return r; }
GF_V_C(gf_minu1, "v_pk_min_u16", 1)
GF_V_C(gf_max0, "v_pk_max_i16", 0)
GF_V_C(gf_add1, "v_pk_add_u16", 1)
#define GF_V_S(name, ins) \
	__device__ __forceinline__ uint32_t name(uint32_t a, uint32_t s) { uint32_t r; asm(ins " %0, %1, %2" : "=v"(r) : "v"(a), "s"(s)); return r; }
GF_V_S(gf_sub_s, "v_pk_sub_u16")
__device__ __forceinline__ uint32_t gf_rsub_s(uint32_t s, uint32_t a) { uint32_t r; asm("v_pk_sub_u16 %0, %1, %2" : "=v"(r) : "s"(s), "v"(a)); return r; }
__device__ __forceinline__ uint32_t gf_mad_vsv(uint32_t a, uint32_t s, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(s), "v"(c)); return r; }
__device__ __forceinline__ uint32_t gf_mad_vv1(uint32_t a, uint32_t b) { uint32_t r; asm("v_pk_mad_u16 %0, %1, %2, 1 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define GF_MADC(name, cst) \
	__device__ __forceinline__ uint32_t name(uint32_t a, uint32_t c) { uint32_t r; asm("v_pk_mad_u16 %0, %1, " #cst ", %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(c)); return r; }
GF_MADC(gf_mad8, 8)
GF_MADC(gf_mad16, 16)
GF_MADC(gf_mad32, 32)
GF_MADC(gf_mad64, 64)
__device__ __forceinline__ uint32_t gf_ror1(uint32_t v) { uint32_t r; asm("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v)); return r; } // every lane has a source: no old operand
// sign-extend the low byte of each 16-bit half
__device__ __forceinline__ uint32_t gf_sext8(uint32_t v)
{
	uint32_t r;
	asm("v_pk_lshlrev_b16 %0, 8, %1 op_sel_hi:[0,1]\n\tv_pk_ashrrev_i16 %0, 8, %0 op_sel_hi:[0,1]" : "=&v"(r) : "v"(v));
	return r;
}

__device__ __forceinline__ uint32_t gf_asr3(uint32_t v) { uint32_t r; asm("v_pk_ashrrev_i16 %0, 3, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(v)); return r; } // key -> difference

// ---------------------------------------------------------------------------------------------------------
// The arithmetic of one register set and anti-diagonal (128 cells), in an instruction order of our own.  gfx950 needs one wait
// state between a packed op and a dependent one; the compiler covers it with an s_nop after nearly every instruction of a
// dependent chain (19 per row when the body is written as separate statements).  Here the 50 packed operations are ordered so
// that no instruction reads its predecessor's result -- the four gap candidates, the substitution score, the direction index and
// the continuation flags are independent strands woven together, the maximum is a tree -- and emitted as three blocks the
// compiler cannot reorder; two wait states remain (around the maximum every later value depends on).
//   in : x1 = tv ^ qv, o1 = tv | qv (base codes), xp / vp / x2p (left neighbour's x, v, x2), u / y / y2 (this column's)
//   out: u, v, x, y, x2, y2 updated; d = direction byte per half
// Same operations as ksw2_extd2_sse.c:165-272 on the valid cells; see the kernel header for the encoding of d.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gf_cell(uint32_t x1, uint32_t o1, uint32_t xp, uint32_t vp, uint32_t x2p, uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y,
                                        uint32_t &x2, uint32_t &y2, uint32_t &d, uint32_t P_MCH, uint32_t S_MISD, uint32_t S_SCN, uint32_t S_Q, uint32_t S_Q2,
                                        uint32_t S_QE, uint32_t S_QE2)
{
	uint32_t a, b, a2, b2, z, z4, m, n, w, tA, tB;
	asm volatile(
		"v_pk_add_u16 %[a], %[xp], %[vp]\n\t"
		"v_pk_min_u16 %[m], %[x1], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_add_u16 %[b], %[y], %[u]\n\t"
		"v_pk_add_u16 %[a2], %[x2p], %[vp]\n\t"
		"v_pk_mad_u16 %[z], %[m], %[misd], %[mch]\n\t"
		"v_pk_add_u16 %[b2], %[y2], %[u]\n\t"
		"v_pk_max_i16 %[tA], %[a], %[b]\n\t"
		"v_pk_lshrrev_b16 %[n], 2, %[o1] op_sel_hi:[0,1]\n\t"
		"v_pk_max_i16 %[tB], %[a2], %[b2]\n\t"
		"v_pk_sub_u16 %[w], %[scn], %[z]\n\t"
		"v_pk_max_i16 %[tA], %[tA], %[tB]\n\t"
		"v_pk_mad_u16 %[z], %[n], %[w], %[z]\n\t"
		"s_nop 0\n\t"
		"v_pk_max_i16 %[z4], %[z], %[tA]\n\t"
		"s_nop 0"
		: [a] "=&v"(a), [b] "=&v"(b), [a2] "=&v"(a2), [b2] "=&v"(b2), [z] "=&v"(z), [z4] "=&v"(z4), [m] "=&v"(m), [n] "=&v"(n), [w] "=&v"(w), [tA] "=&v"(tA), [tB] "=&v"(tB)
		: [xp] "v"(xp), [vp] "v"(vp), [x2p] "v"(x2p), [x1] "v"(x1), [o1] "v"(o1), [u] "v"(u), [y] "v"(y), [y2] "v"(y2), [mch] "v"(P_MCH), [misd] "s"(S_MISD), [scn] "s"(S_SCN));
	uint32_t zc, d0, d1, d2, d3, t1, t2, e, un, vn;
	asm volatile(
		"v_pk_sub_u16 %[d0], %[z4], %[z]\n\t"
		"v_pk_sub_u16 %[d1], %[z4], %[a]\n\t"
		"v_pk_sub_u16 %[d2], %[z4], %[b]\n\t"
		"v_pk_sub_u16 %[d3], %[z4], %[a2]\n\t"
		"v_pk_min_i16 %[zc], %[z4], %[mch]\n\t"
		"v_pk_min_u16 %[d0], %[d0], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_min_u16 %[d1], %[d1], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_min_u16 %[d2], %[d2], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_min_u16 %[d3], %[d3], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_sub_u16 %[un], %[zc], %[vp]\n\t"
		"v_pk_sub_u16 %[vn], %[zc], %[u]\n\t"
		"v_pk_sub_u16 %[t1], %[zc], %[q]\n\t"
		"v_pk_sub_u16 %[t2], %[zc], %[q2]\n\t"
		"v_pk_add_u16 %[e], %[d3], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_sub_u16 %[a], %[a], %[t1]\n\t"
		"v_pk_sub_u16 %[b], %[b], %[t1]\n\t"
		"v_pk_mad_u16 %[e], %[d2], %[e], 1 op_sel_hi:[1,1,0]\n\t"
		"v_pk_sub_u16 %[a2], %[a2], %[t2]\n\t"
		"v_pk_sub_u16 %[b2], %[b2], %[t2]\n\t"
		"v_pk_mad_u16 %[e], %[d1], %[e], 1 op_sel_hi:[1,1,0]\n\t"
		"v_pk_max_i16 %[a], %[a], 0 op_sel_hi:[1,0]\n\t"
		"v_pk_max_i16 %[b], %[b], 0 op_sel_hi:[1,0]\n\t"
		"v_pk_mul_lo_u16 %[e], %[d0], %[e]\n\t"
		"v_pk_max_i16 %[a2], %[a2], 0 op_sel_hi:[1,0]\n\t"
		"v_pk_max_i16 %[b2], %[b2], 0 op_sel_hi:[1,0]"
		: [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3), [zc] "=&v"(zc), [t1] "=&v"(t1), [t2] "=&v"(t2), [e] "=&v"(e), [un] "=&v"(un), [vn] "=&v"(vn),
		  [a] "+v"(a), [b] "+v"(b), [a2] "+v"(a2), [b2] "+v"(b2)
		: [z4] "v"(z4), [z] "v"(z), [vp] "v"(vp), [u] "v"(u), [mch] "v"(P_MCH), [q] "s"(S_Q), [q2] "s"(S_Q2));
	uint32_t fa, fb, fa2, fb2, xn, yn, x2n, y2n;
	asm volatile(
		"v_pk_min_u16 %[fa], %[a], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_sub_u16 %[xn], %[a], %[qe]\n\t"
		"v_pk_min_u16 %[fb], %[b], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_mad_u16 %[e], %[fa], 8, %[e] op_sel_hi:[1,0,1]\n\t"
		"v_pk_sub_u16 %[yn], %[b], %[qe]\n\t"
		"v_pk_min_u16 %[fa2], %[a2], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_mad_u16 %[e], %[fb], 16, %[e] op_sel_hi:[1,0,1]\n\t"
		"v_pk_sub_u16 %[x2n], %[a2], %[qe2]\n\t"
		"v_pk_min_u16 %[fb2], %[b2], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_mad_u16 %[e], %[fa2], 32, %[e] op_sel_hi:[1,0,1]\n\t"
		"v_pk_sub_u16 %[y2n], %[b2], %[qe2]\n\t"
		"v_pk_mad_u16 %[e], %[fb2], 64, %[e] op_sel_hi:[1,0,1]"
		: [fa] "=&v"(fa), [fb] "=&v"(fb), [fa2] "=&v"(fa2), [fb2] "=&v"(fb2), [xn] "=&v"(xn), [yn] "=&v"(yn), [x2n] "=&v"(x2n), [y2n] "=&v"(y2n), [e] "+v"(e)
		: [a] "v"(a), [b] "v"(b), [a2] "v"(a2), [b2] "v"(b2), [qe] "s"(S_QE), [qe2] "s"(S_QE2));
	u = un, v = vn, x = xn, y = yn, x2 = x2n, y2 = y2n, d = e;
}

#endif // MM2AMD_WAVE_EMU (the instruction wrappers above)

// ---------------------------------------------------------------------------------------------------------
// gf_cell_k: the same cell with KEYED candidates (round 4).  A third of gf_cell's packed operations only name the direction: four
// differences against the maximum, four normalisations to 0/1, and a multiply-add chain that turns them into "the first of (s, a, b,
// a2, b2) that reaches the maximum" (ksw2_extd2_sse.c:235-243) -- packed compares do not exist.  Here every score difference is kept
// times 8, and the low three bits of a candidate carry a tag that falls with its rank in that order (s 7, a 6, b 5, a2 4, b2 3): the
// signed maximum of the five KEYS is the maximum value and, among equals, the first candidate, so the direction is the key's low bits
// (one 32-bit AND on both halves, VOP2: half the issue cost of a packed op) and the twelve operations are gone.  The tags ride in the
// stored gap states: x carries 6, y 5, x2 4, y2 3 (a = x + v, so the sum has x's tag; u and v carry none), the substitution score gets
// its 7 from the constants.  The gap states are stored as the reference stores them, x = max(a - (z - q), 0) - (q + e), in one step:
// max(a - (z + e), tag - 8 (q + e)) -- the clamp and the subtraction folded into the constant ka (kb, ka2, kb2) -- and "the gap
// continues" (a - (z - q) > 0) is x >= ka + 8: min_i16(x, ka + 8) is ka or ka + 8, summed with weights 1, 2, 4, 8.  The byte stored per
// cell is therefore   (7 - d) + 8 fa + 16 fb + 32 fa2 + 64 fb2 + bias (mod 256),   bias = ka + 2 kb + 4 ka2 + 8 kb2,
// and the traceback -- the only reader -- takes the bias off and flips the low bits (gf_k_decode).  No value leaves the range of a
// 16-bit half: the reference's 8-bit quantities times 8.  34 packed + 2 VOP2 operations per register set and row instead of 50.
//   in : x1, o1 as gf_cell; xp (tag 6), vp, x2p (tag 4), u, y (tag 5), y2 (tag 3) -- all times 8;  u is updated in place
// ---------------------------------------------------------------------------------------------------------
constexpr int GF_K_TS = 7, GF_K_TA = 6, GF_K_TB = 5, GF_K_TA2 = 4, GF_K_TB2 = 3;
// the launch's constants of the keyed cell, all as both halves of a dword (pk2): built once per kernel by gf_k_consts
struct GfK {
	uint32_t mcht;                  // 8 mch + 7 (VGPR: it meets an SGPR operand in its instruction)
	uint32_t misd8, scnt, mch8;     // 8 (mis - mch), 8 sc_N + 7, 8 mch
	uint32_t e8, e28;               // 8 e, 8 e2: z - q + (q + e) = z + e is what a gap candidate is measured against when the state is stored as x = max(., 0) - (q + e)
	uint32_t ka, kb, ka2, kb2;      // the clamp of a stored gap state: tag - 8 (q + e)  (x, y) / tag - 8 (q2 + e2)  (x2, y2)
	uint32_t ka8, kb8, ka28, kb28;  // the same plus 8: "the gap continues" is state >= clamp + 8
	uint32_t nqe_x, nqe_y, nqe2_x, nqe2_y; // border values of x / y / x2 / y2: -(q + e), -(q2 + e2) times 8, tagged (= ka, kb, ka2, kb2)
	int bias;                       // what the stored byte carries on top of (7 - d) + 8 fa + 16 fb + 32 fa2 + 64 fb2, mod 256
};
__device__ __forceinline__ uint32_t gf_pk2(int v) { return ((uint32_t)v & 0xffffu) | (uint32_t)v << 16; }
__device__ __forceinline__ GfK gf_k_consts(int sc_mch, int sc_mis, int sc_N, int q, int e, int q2, int e2) // (q, e) = the cheaper pair to open, as the kernels swap them
{
	GfK k;
	const int qe8 = 8 * (q + e), qe28 = 8 * (q2 + e2);
	k.mcht = 0, k.misd8 = gf_pk2(8 * (sc_mis - sc_mch)), k.scnt = gf_pk2(8 * sc_N + GF_K_TS), k.mch8 = gf_pk2(8 * sc_mch);
	k.e8 = gf_pk2(8 * e), k.e28 = gf_pk2(8 * e2);
	const int ka = GF_K_TA - qe8, kb = GF_K_TB - qe8, ka2 = GF_K_TA2 - qe28, kb2 = GF_K_TB2 - qe28;
	k.ka = gf_pk2(ka), k.kb = gf_pk2(kb), k.ka2 = gf_pk2(ka2), k.kb2 = gf_pk2(kb2);
	k.ka8 = gf_pk2(ka + 8), k.kb8 = gf_pk2(kb + 8), k.ka28 = gf_pk2(ka2 + 8), k.kb28 = gf_pk2(kb2 + 8);
	k.nqe_x = k.ka, k.nqe_y = k.kb, k.nqe2_x = k.ka2, k.nqe2_y = k.kb2;
	k.bias = (ka + 2 * kb + 4 * ka2 + 8 * kb2) & 0xff;
	return k;
}
__device__ __forceinline__ int gf_k_decode(int byte, int bias) { return ((byte - bias) ^ 7) & 0xff; } // -> the reference's direction byte (d | flags << 3)
#ifndef MM2AMD_WAVE_EMU
__device__ __forceinline__ void gf_cell_k(uint32_t x1, uint32_t o1, uint32_t xp, uint32_t vp, uint32_t x2p, uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y,
                                          uint32_t &x2, uint32_t &y2, uint32_t &d, uint32_t P_MCHT, const GfK &K)
{
	uint32_t a, b, a2, b2, z, z4, m, n, w, tA, tB;
	asm volatile(
		"v_pk_add_u16 %[a], %[xp], %[vp]\n\t"
		"v_pk_min_u16 %[m], %[x1], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_add_u16 %[b], %[y], %[u]\n\t"
		"v_pk_add_u16 %[a2], %[x2p], %[vp]\n\t"
		"v_pk_mad_u16 %[z], %[m], %[misd], %[mch]\n\t"
		"v_pk_add_u16 %[b2], %[y2], %[u]\n\t"
		"v_pk_max_i16 %[tA], %[a], %[b]\n\t"
		"v_pk_lshrrev_b16 %[n], 2, %[o1] op_sel_hi:[0,1]\n\t"
		"v_pk_max_i16 %[tB], %[a2], %[b2]\n\t"
		"v_pk_sub_u16 %[w], %[scn], %[z]\n\t"
		"v_pk_max_i16 %[tA], %[tA], %[tB]\n\t"
		"v_pk_mad_u16 %[z], %[n], %[w], %[z]\n\t"
		"s_nop 0\n\t"
		"v_pk_max_i16 %[z4], %[z], %[tA]\n\t"
		"s_nop 0"
		: [a] "=&v"(a), [b] "=&v"(b), [a2] "=&v"(a2), [b2] "=&v"(b2), [z] "=&v"(z), [z4] "=&v"(z4), [m] "=&v"(m), [n] "=&v"(n), [w] "=&v"(w), [tA] "=&v"(tA), [tB] "=&v"(tB)
		: [xp] "v"(xp), [vp] "v"(vp), [x2p] "v"(x2p), [x1] "v"(x1), [o1] "v"(o1), [u] "v"(u), [y] "v"(y), [y2] "v"(y2), [mch] "v"(P_MCHT), [misd] "s"(K.misd8), [scn] "s"(K.scnt));
	uint32_t zv, zc, t1, t2, e, vn, xn, yn, x2n, y2n;
	asm volatile(
		"v_and_b32 %[zv], 0xfff8fff8, %[z4]\n\t"
		"v_and_b32 %[e], 0x70007, %[z4]\n\t"
		"v_pk_min_i16 %[zc], %[zv], %[mch8]\n\t"
		"s_nop 0\n\t"
		"v_pk_sub_u16 %[vn], %[zc], %[u]\n\t"
		"v_pk_add_u16 %[t1], %[zc], %[e8]\n\t"
		"v_pk_add_u16 %[t2], %[zc], %[e28]\n\t"
		"v_pk_sub_u16 %[u], %[zc], %[vp]\n\t"
		"v_pk_sub_u16 %[a], %[a], %[t1]\n\t"
		"v_pk_sub_u16 %[b], %[b], %[t1]\n\t"
		"v_pk_sub_u16 %[a2], %[a2], %[t2]\n\t"
		"v_pk_sub_u16 %[b2], %[b2], %[t2]\n\t"
		"v_pk_max_i16 %[xn], %[a], %[ka]\n\t"
		"v_pk_max_i16 %[yn], %[b], %[kb]\n\t"
		"v_pk_max_i16 %[x2n], %[a2], %[ka2]\n\t"
		"v_pk_max_i16 %[y2n], %[b2], %[kb2]"
		: [zv] "=&v"(zv), [zc] "=&v"(zc), [t1] "=&v"(t1), [t2] "=&v"(t2), [e] "=&v"(e), [vn] "=&v"(vn), [xn] "=&v"(xn), [yn] "=&v"(yn), [x2n] "=&v"(x2n), [y2n] "=&v"(y2n),
		  [u] "+v"(u), [a] "+v"(a), [b] "+v"(b), [a2] "+v"(a2), [b2] "+v"(b2)
		: [z4] "v"(z4), [vp] "v"(vp), [mch8] "s"(K.mch8), [e8] "s"(K.e8), [e28] "s"(K.e28), [ka] "s"(K.ka), [kb] "s"(K.kb), [ka2] "s"(K.ka2), [kb2] "s"(K.kb2));
	uint32_t fa, fb, fa2, fb2;
	asm volatile(
		"v_pk_min_i16 %[fa], %[xn], %[ka8]\n\t"
		"v_pk_min_i16 %[fb], %[yn], %[kb8]\n\t"
		"v_pk_min_i16 %[fa2], %[x2n], %[ka28]\n\t"
		"v_pk_mad_u16 %[fa], %[fb], 2, %[fa] op_sel_hi:[1,0,1]\n\t"
		"v_pk_min_i16 %[fb2], %[y2n], %[kb28]\n\t"
		"v_pk_add_u16 %[e], %[e], %[fa]\n\t"
		"v_pk_mad_u16 %[fa2], %[fb2], 2, %[fa2] op_sel_hi:[1,0,1]\n\t"
		"s_nop 0\n\t"
		"v_pk_mad_u16 %[e], %[fa2], 4, %[e] op_sel_hi:[1,0,1]"
		: [fa] "=&v"(fa), [fb] "=&v"(fb), [fa2] "=&v"(fa2), [fb2] "=&v"(fb2), [e] "+v"(e)
		: [xn] "v"(xn), [yn] "v"(yn), [x2n] "v"(x2n), [y2n] "v"(y2n), [ka8] "s"(K.ka8), [kb8] "s"(K.kb8), [ka28] "s"(K.ka28), [kb28] "s"(K.kb28));
	v = vn, x = xn, y = yn, x2 = x2n, y2 = y2n, d = e;
}
// gf_cell_k2: gf_cell_k with every input by value and every output separate (round 6, the banded kernel: on odd rows a lane's (u, y, y2) arrive in registers that must
// survive the cell -- the DPP moves' destination, whose last lane keeps the band's edge constant from row to row -- and go out into the lane's own).  Same
// instructions in the same order; u's new value gets a register of its own instead of replacing the old one.
__device__ __forceinline__ void gf_cell_k2(uint32_t x1, uint32_t o1, uint32_t xp, uint32_t vp, uint32_t x2p, uint32_t u, uint32_t y, uint32_t y2, uint32_t &un, uint32_t &v, uint32_t &x, uint32_t &yo,
                                           uint32_t &x2, uint32_t &y2o, uint32_t &d, uint32_t P_MCHT, const GfK &K)
{
	uint32_t a, b, a2, b2, z, z4, m, n, w, tA, tB;
	asm volatile(
		"v_pk_add_u16 %[a], %[xp], %[vp]\n\t"
		"v_pk_min_u16 %[m], %[x1], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_add_u16 %[b], %[y], %[u]\n\t"
		"v_pk_add_u16 %[a2], %[x2p], %[vp]\n\t"
		"v_pk_mad_u16 %[z], %[m], %[misd], %[mch]\n\t"
		"v_pk_add_u16 %[b2], %[y2], %[u]\n\t"
		"v_pk_max_i16 %[tA], %[a], %[b]\n\t"
		"v_pk_lshrrev_b16 %[n], 2, %[o1] op_sel_hi:[0,1]\n\t"
		"v_pk_max_i16 %[tB], %[a2], %[b2]\n\t"
		"v_pk_sub_u16 %[w], %[scn], %[z]\n\t"
		"v_pk_max_i16 %[tA], %[tA], %[tB]\n\t"
		"v_pk_mad_u16 %[z], %[n], %[w], %[z]\n\t"
		"s_nop 0\n\t"
		"v_pk_max_i16 %[z4], %[z], %[tA]\n\t"
		"s_nop 0"
		: [a] "=&v"(a), [b] "=&v"(b), [a2] "=&v"(a2), [b2] "=&v"(b2), [z] "=&v"(z), [z4] "=&v"(z4), [m] "=&v"(m), [n] "=&v"(n), [w] "=&v"(w), [tA] "=&v"(tA), [tB] "=&v"(tB)
		: [xp] "v"(xp), [vp] "v"(vp), [x2p] "v"(x2p), [x1] "v"(x1), [o1] "v"(o1), [u] "v"(u), [y] "v"(y), [y2] "v"(y2), [mch] "v"(P_MCHT), [misd] "s"(K.misd8), [scn] "s"(K.scnt));
	uint32_t zv, zc, t1, t2, e, vn, xn, yn, x2n, y2n, u_new;
	asm volatile(
		"v_and_b32 %[zv], 0xfff8fff8, %[z4]\n\t"
		"v_and_b32 %[e], 0x70007, %[z4]\n\t"
		"v_pk_min_i16 %[zc], %[zv], %[mch8]\n\t"
		"s_nop 0\n\t"
		"v_pk_sub_u16 %[vn], %[zc], %[u]\n\t"
		"v_pk_add_u16 %[t1], %[zc], %[e8]\n\t"
		"v_pk_add_u16 %[t2], %[zc], %[e28]\n\t"
		"v_pk_sub_u16 %[un], %[zc], %[vp]\n\t"
		"v_pk_sub_u16 %[a], %[a], %[t1]\n\t"
		"v_pk_sub_u16 %[b], %[b], %[t1]\n\t"
		"v_pk_sub_u16 %[a2], %[a2], %[t2]\n\t"
		"v_pk_sub_u16 %[b2], %[b2], %[t2]\n\t"
		"v_pk_max_i16 %[xn], %[a], %[ka]\n\t"
		"v_pk_max_i16 %[yn], %[b], %[kb]\n\t"
		"v_pk_max_i16 %[x2n], %[a2], %[ka2]\n\t"
		"v_pk_max_i16 %[y2n], %[b2], %[kb2]"
		: [zv] "=&v"(zv), [zc] "=&v"(zc), [t1] "=&v"(t1), [t2] "=&v"(t2), [e] "=&v"(e), [vn] "=&v"(vn), [xn] "=&v"(xn), [yn] "=&v"(yn), [x2n] "=&v"(x2n), [y2n] "=&v"(y2n),
		  [un] "=&v"(u_new), [a] "+v"(a), [b] "+v"(b), [a2] "+v"(a2), [b2] "+v"(b2)
		: [z4] "v"(z4), [vp] "v"(vp), [u] "v"(u), [mch8] "s"(K.mch8), [e8] "s"(K.e8), [e28] "s"(K.e28), [ka] "s"(K.ka), [kb] "s"(K.kb), [ka2] "s"(K.ka2), [kb2] "s"(K.kb2));
	uint32_t fa, fb, fa2, fb2;
	asm volatile(
		"v_pk_min_i16 %[fa], %[xn], %[ka8]\n\t"
		"v_pk_min_i16 %[fb], %[yn], %[kb8]\n\t"
		"v_pk_min_i16 %[fa2], %[x2n], %[ka28]\n\t"
		"v_pk_mad_u16 %[fa], %[fb], 2, %[fa] op_sel_hi:[1,0,1]\n\t"
		"v_pk_min_i16 %[fb2], %[y2n], %[kb28]\n\t"
		"v_pk_add_u16 %[e], %[e], %[fa]\n\t"
		"v_pk_mad_u16 %[fa2], %[fb2], 2, %[fa2] op_sel_hi:[1,0,1]\n\t"
		"s_nop 0\n\t"
		"v_pk_mad_u16 %[e], %[fa2], 4, %[e] op_sel_hi:[1,0,1]"
		: [fa] "=&v"(fa), [fb] "=&v"(fb), [fa2] "=&v"(fa2), [fb2] "=&v"(fb2), [e] "+v"(e)
		: [xn] "v"(xn), [yn] "v"(yn), [x2n] "v"(x2n), [y2n] "v"(y2n), [ka8] "s"(K.ka8), [kb8] "s"(K.kb8), [ka28] "s"(K.ka28), [kb28] "s"(K.kb28));
	un = u_new, v = vn, x = xn, yo = yn, x2 = x2n, y2o = y2n, d = e;
}
#endif // MM2AMD_WAVE_EMU (the emulator's twin: ksw_pk_emu.hpp)

// One register set and anti-diagonal with RIGHT-aligned gaps (ksw2_extd2_sse.c:282-320): the LAST of (s, a, b, a2, b2) that reaches the
// maximum names the state (ties go to the gap states), and a gap continues when its value is >= 0, not > 0.  Same operands and results
// as gf_cell; written with the one-instruction helpers (this variant serves 1.5 % of the cells).  Used by the two extension kernels (ksw_ext.hip, ksw_extq.hip).
__device__ __forceinline__ void gf_cell_right(uint32_t x1, uint32_t o1, uint32_t xp, uint32_t vp, uint32_t x2p, uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y,
                                              uint32_t &x2, uint32_t &y2, uint32_t &d, uint32_t P_MCH, uint32_t P_MISD, uint32_t P_SCN, uint32_t P_Q, uint32_t P_Q2,
                                              uint32_t P_QE, uint32_t P_QE2)
{
	const uint32_t ONE = pk2(1);
	uint32_t a = pk_add(xp, vp), b = pk_add(y, u), a2 = pk_add(x2p, vp), b2 = pk_add(y2, u);
	uint32_t z = pk_mad(pk_minu(x1, ONE), P_MISD, P_MCH);           // match / mismatch
	z = pk_mad(pk_shr2(o1), pk_sub(P_SCN, z), z);                   // sc_N when either base is ambiguous (code 4 = bit 2)
	const uint32_t z4 = pk_max(pk_max(pk_max(z, a), pk_max(b, a2)), b2);
	// state = 4 - d4 * (1 + d3 * (1 + d2 * (1 + d1))), d_i = 1 when candidate i is below the maximum: the last candidate equal to it
	const uint32_t d1 = pk_minu(pk_sub(z4, a), ONE), d2 = pk_minu(pk_sub(z4, b), ONE), d3 = pk_minu(pk_sub(z4, a2), ONE), d4 = pk_minu(pk_sub(z4, b2), ONE);
	uint32_t e = pk_add(d1, ONE);
	e = pk_mad(d2, e, ONE);
	e = pk_mad(d3, e, ONE);
	e = pk_sub(pk2(4), pk_mul(d4, e));
	const uint32_t zc = pk_min(z4, P_MCH);
	const uint32_t un = pk_sub(zc, vp), vn = pk_sub(zc, u), t1 = pk_sub(zc, P_Q), t2 = pk_sub(zc, P_Q2);
	a = pk_sub(a, t1), b = pk_sub(b, t1), a2 = pk_sub(a2, t2), b2 = pk_sub(b2, t2);
	auto ge0 = [&](uint32_t w) { return pk_minu(pk_max(pk_add(w, ONE), 0u), ONE); }; // 1 where the signed half is >= 0
	e = pk_mad(ge0(a), pk2(8), e), e = pk_mad(ge0(b), pk2(16), e), e = pk_mad(ge0(a2), pk2(32), e), e = pk_mad(ge0(b2), pk2(64), e);
	x = pk_sub(pk_max(a, 0u), P_QE), y = pk_sub(pk_max(b, 0u), P_QE), x2 = pk_sub(pk_max(a2, 0u), P_QE2), y2 = pk_sub(pk_max(b2, 0u), P_QE2);
	u = un, v = vn, d = e;
}

// ---- ksw_backtrack (ksw2.h:130-162) with every cell inside the matrix, by the 32 lanes of a half-wave; both halves of a wave at once ----
// Lane k of the half looks k cells ahead along the run of the current state (the match diagonal, or a gap) and one ballot tells how far the
// run goes, so a read's typical 8-base match runs cost one load round, not eight.  Round 4: the cell that ENDS the run was loaded by lane
// `run` in the same round -- it names the next state (:141-144) and is consumed in the same iteration, so a one-base indel between two match
// runs costs two rounds of dependent loads instead of three; and the CIGAR pusher is written without branches on the operation (every lane of
// the half keeps the same (i, j, state, CIGAR tail); lane 0 of the half stores).  A fifth of the gap-fill kernels' instructions were this
// loop, its four inlined pushers diverging between the two halves.
//   start: the half has a job and (i, j) is inside the matrix;  dir_at(ii, jj) -> the reference's direction byte of cell (target ii, query jj)
__device__ __forceinline__ void gf_cig_push(FastCig &g, uint32_t op, int len, bool writer) // ksw_push_cigar (ksw2.h:114-124); len may be 0
{
	const bool same = g.n > 0 && op == (g.last & 0xf);
	const bool fresh = len > 0 && !same;
	if (fresh && g.n > 0 && writer) g.c[g.n - 1] = g.last;
	g.last = fresh ? (uint32_t)len << 4 | op : g.last + ((uint32_t)len << 4);
	g.n += fresh ? 1 : 0;
}
template <class DIR>
__device__ __forceinline__ void gf_traceback(bool start, int i, int j, DIR dir_at, FastCig &g)
{
	const int lane = (int)(threadIdx.x & 63), hl = lane & 31, hbase = lane & 32;
	const bool writer = hl == 0;
	int state = 0;
	bool live = start && i >= 0 && j >= 0; // uniform within a half
	while (__ballot(live) != 0ull) {
		const int di = (state == 2 || state == 4) ? 0 : 1, dj = (state == 1 || state == 3) ? 0 : 1;
		const int ii = i - hl * di, jj = j - hl * dj;
		const bool valid = live && ii >= 0 && jj >= 0;
		const int tmp = valid ? dir_at(ii, jj) : 0;
		const bool cont = valid && (state == 0 ? (tmp & 7) == 0 : (tmp >> (state + 2) & 1) != 0);
		const unsigned long long bal = __ballot(cont);
		const uint32_t mine = hbase ? (uint32_t)(bal >> 32) : (uint32_t)bal;
		const int run = mine == 0xffffffffu ? 32 : __builtin_ctz(~mine);
		const int stop = __shfl(tmp, hbase + (run & 31), 64); // the cell that ended the run (meaningful when run < 32 and it lies inside the matrix)
		if (live) {
			int ni = i - run * di, nj = j - run * dj;
			const uint32_t op1 = state == 0 ? 0u : dj == 0 ? 2u : 1u;
			const bool step = run < 32 && ni >= 0 && nj >= 0;
			const int ns = step ? (stop & 7) : state;
			const int si = (ns == 2 || ns == 4) ? 0 : 1, sj = (ns == 1 || ns == 3) ? 0 : 1;
			const uint32_t op2 = ns == 0 ? 0u : sj == 0 ? 2u : 1u;
			gf_cig_push(g, op1, run, writer);
			gf_cig_push(g, op2, step ? 1 : 0, writer);
			i = step ? ni - si : ni, j = step ? nj - sj : nj, state = ns;
			live = i >= 0 && j >= 0;
		}
	}
	if (start) {
		if (i >= 0) gf_cig_push(g, 2u, i + 1, writer);
		if (j >= 0) gf_cig_push(g, 1u, j + 1, writer);
	}
}

// ---- mm_test_zdrop's walk over a finished alignment (align.c:46-84), by the 32 lanes of a half-wave ----
// The reference walks the CIGAR from the start: a running score (substitution scores base by base, -(q + e * len) per gap), the
// running maximum with the position of its LAST occurrence, and, at every step below the maximum, the drop
// max - score - |advance on the target - advance on the query| * e; it reports the FIRST step with the largest positive drop
// (update_max_zdrop, align.c:46-59).  A "step" is one aligned base pair or one whole gap.  Here the steps are cut into 32 consecutive
// segments, one per lane: (1) prefix sums over the operations (steps, target and query advance) give every operation its first
// step and position; (2) each lane walks its segment once for the segment's score sum and its highest prefix, a scan over the lanes
// turns these into the score and the running maximum (value, position; the later of equals) each segment starts from; (3) a second
// walk applies the reference's test step by step, and the largest drop of the lowest lane wins.  The same walks sum the alignment's
// score under the DP's own costs (dual affine: the cheaper of the two gap costs; ambiguous bases by sc_N, ksw2_extd2_sse.c:71) =
// the corner cell the reference reports (:366-383).  Both half-waves of a wave run it at once, each on its own job (have = the
// half has one); ops = the half's operations LAST FIRST (n_ops of them), pA / pB = n_ops + 1 words of scratch each.
struct GfZdrop { int32_t zd_max, t0, t1, q0, q1, dp_sum; };

template <class TB, class QB>
__device__ __forceinline__ GfZdrop gf_zdrop_scan(bool have, int n_ops, const uint32_t *ops, uint32_t *pA, uint32_t *pB, TB tb_at, QB qb_at,
                                                 const int8_t *s_mat, int gq, int ge, int gq2, int ge2, int sc_N)
{
	const int lane = (int)(threadIdx.x & 63), hl = lane & 31;
	if (!have) n_ops = 0;
	const int n_other = __shfl_xor(n_ops, 32, 64), n_max = n_ops > n_other ? n_ops : n_other; // the loops with shuffles run the same count in both halves
	// (1) first step and position of every operation
	int S = 0, tot_i = 0, tot_j = 0;
	for (int f0 = 0; f0 < n_max; f0 += 32) {
		const int f = f0 + hl;
		const bool ok = f < n_ops;
		const uint32_t w = ok ? ops[n_ops - 1 - f] : 0u, op = w & 0xf;
		const int len = (int)(w >> 4);
		int st = ok ? (op == 0 ? len : 1) : 0, di = ok && op != 1 ? len : 0, dj = ok && op != 2 ? len : 0;
		const int st0 = st, di0 = di, dj0 = dj;
#pragma unroll
		for (int d = 1; d < 32; d <<= 1) {
			const int a = __shfl_up(st, d, 32), b = __shfl_up(di, d, 32), c = __shfl_up(dj, d, 32);
			if (hl >= d) st += a, di += b, dj += c;
		}
		if (ok) pA[f] = (uint32_t)(S + st - st0), pB[f] = (uint32_t)(tot_i + di - di0) | (uint32_t)(tot_j + dj - dj0) << 16;
		S += __shfl(st, 31, 32), tot_i += __shfl(di, 31, 32), tot_j += __shfl(dj, 31, 32);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	// the lane's segment of steps, and the operation its first step lies in
	const int seg = (S + 31) >> 5, lo = hl * seg < S ? hl * seg : S, hi = lo + seg < S ? lo + seg : S;
	int f_lo = 0;
	if (lo < hi) {
		int l = 0, r = n_ops - 1;
		while (l < r) {
			const int mid = (l + r + 1) >> 1;
			if ((int)pA[mid] <= lo) l = mid; else r = mid - 1;
		}
		f_lo = l;
	}
	// one step after the other: body(score change, DP-score change, target position, query position)
	auto walk = [&](auto body) {
		if (lo >= hi) return;
		int f = f_lo;
		uint32_t w = ops[n_ops - 1 - f], op = w & 0xf;
		int len = (int)(w >> 4), first = (int)pA[f], cnt = op == 0 ? len : 1;
		const uint32_t b0 = pB[f];
		int ci = (int)(b0 & 0xffffu), cj = (int)(b0 >> 16);
		for (int s = lo; s < hi; ++s) {
			if (s >= first + cnt) { // next operation
				if (op != 1) ci += len;
				if (op != 2) cj += len;
				first += cnt, ++f;
				w = ops[n_ops - 1 - f], op = w & 0xf, len = (int)(w >> 4), cnt = op == 0 ? len : 1;
			}
			if (op == 0) {
				const int o = s - first, tb = tb_at(ci + o), qb = qb_at(cj + o), sm = s_mat[tb * 5 + qb];
				body(sm, ((tb | qb) & 4) ? sc_N : sm, ci + o, cj + o);
			} else {
				const int c1 = gq + ge * len, c2 = gq2 + ge2 * len;
				body(-c1, -(c1 < c2 ? c1 : c2), op == 2 ? ci + len : ci, op == 1 ? cj + len : cj);
			}
		}
	};
	// (2) per segment: score sum, DP-score sum, highest prefix (the later of equals) relative to the segment's start
	int sum = 0, dp = 0, rel = INT32_MIN, rel_i = -1, rel_j = -1;
	walk([&](int d, int dd, int pi, int pj) {
		sum += d, dp += dd;
		if (sum >= rel) rel = sum, rel_i = pi, rel_j = pj;
	});
	int incl = sum;
#pragma unroll
	for (int d = 1; d < 32; d <<= 1) {
		const int a = __shfl_up(incl, d, 32);
		if (hl >= d) incl += a;
	}
#pragma unroll
	for (int d = 16; d >= 1; d >>= 1) dp += __shfl_xor(dp, d, 32);
	const int start = incl - sum;
	int mx = rel == INT32_MIN ? INT32_MIN : start + rel, mx_i = rel_i, mx_j = rel_j; // inclusive running maximum over the lanes: the later of equals
#pragma unroll
	for (int d = 1; d < 32; d <<= 1) {
		const int a = __shfl_up(mx, d, 32), b = __shfl_up(mx_i, d, 32), c = __shfl_up(mx_j, d, 32);
		if (hl >= d && a > mx) mx = a, mx_i = b, mx_j = c;
	}
	{ // what the segment starts from: the maximum over the lanes before it
		const int a = __shfl_up(mx, 1, 32), b = __shfl_up(mx_i, 1, 32), c = __shfl_up(mx_j, 1, 32);
		mx = hl ? a : INT32_MIN, mx_i = hl ? b : -1, mx_j = hl ? c : -1;
	}
	// (3) update_max_zdrop over the segment
	GfZdrop z = { 0, -1, -1, -1, -1, dp };
	int score = start;
	walk([&](int d, int, int pi, int pj) {
		score += d;
		if (score < mx) {
			const int li = pi - mx_i, lj = pj - mx_j, diff = li > lj ? li - lj : lj - li, zz = mx - score - diff * ge;
			if (zz > z.zd_max) z.zd_max = zz, z.t0 = mx_i, z.t1 = pi, z.q0 = mx_j, z.q1 = pj;
		} else mx = score, mx_i = pi, mx_j = pj;
	});
	int best = z.zd_max;
#pragma unroll
	for (int d = 16; d >= 1; d >>= 1) { const int a = __shfl_xor(best, d, 32); best = a > best ? a : best; }
	const unsigned long long bal = __ballot(z.zd_max == best);
	const uint32_t mine = lane >= 32 ? (uint32_t)(bal >> 32) : (uint32_t)bal;
	const int src = __builtin_ctz(mine); // the first step with the largest drop lies in the lowest such lane
	z.zd_max = best, z.t0 = __shfl(z.t0, src, 32), z.t1 = __shfl(z.t1, src, 32), z.q0 = __shfl(z.q0, src, 32), z.q1 = __shfl(z.q1, src, 32);
	return z;
}

} // namespace mm2amd

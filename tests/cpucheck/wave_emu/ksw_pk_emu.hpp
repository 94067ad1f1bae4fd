/* TEST INFRASTRUCTURE: host versions of the gfx950 inline-assembly helpers of ksw_pk.hpp and ksw_gapfill_dev.hpp -- each is the operation its
 * instruction performs on the two 16-bit halves of a register (VOP3P v_pk_*), so that the register-resident DP kernels' own source (control
 * flow, carries, rings, traceback, Z-drop walk) runs under the wave emulator.  The instruction sequences themselves are only exercised on the
 * hardware (tests -m gpu). */
#pragma once
#include <cstdint>
namespace mm2amd {
namespace pk_emu {
template <class F> inline uint32_t halves(uint32_t a, uint32_t b, F f) { return ((uint32_t)f((uint16_t)a, (uint16_t)b) & 0xffffu) | (uint32_t)f((uint16_t)(a >> 16), (uint16_t)(b >> 16)) << 16; }
inline uint32_t both(uint32_t c) { return (c & 0xffffu) | c << 16; } // an inline constant or the low half of an SGPR feeding both halves (op_sel_hi clear)
}
inline uint32_t pk_add(uint32_t a, uint32_t b) { return pk_emu::halves(a, b, [](uint16_t x, uint16_t y) { return (uint16_t)(x + y); }); }
inline uint32_t pk_sub(uint32_t a, uint32_t b) { return pk_emu::halves(a, b, [](uint16_t x, uint16_t y) { return (uint16_t)(x - y); }); }
inline uint32_t pk_max(uint32_t a, uint32_t b) { return pk_emu::halves(a, b, [](uint16_t x, uint16_t y) { return (uint16_t)((int16_t)x > (int16_t)y ? x : y); }); }
inline uint32_t pk_min(uint32_t a, uint32_t b) { return pk_emu::halves(a, b, [](uint16_t x, uint16_t y) { return (uint16_t)((int16_t)x < (int16_t)y ? x : y); }); }
inline uint32_t pk_minu(uint32_t a, uint32_t b) { return pk_emu::halves(a, b, [](uint16_t x, uint16_t y) { return (uint16_t)(x < y ? x : y); }); }
inline uint32_t pk_mul(uint32_t a, uint32_t b) { return pk_emu::halves(a, b, [](uint16_t x, uint16_t y) { return (uint16_t)((uint32_t)x * y); }); }
inline uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { return pk_add(pk_mul(a, b), c); }
inline uint32_t pk_shr2(uint32_t a) { return ((a & 0xffffu) >> 2) | ((a >> 16) >> 2) << 16; }
inline uint32_t pk2(int v) { return ((uint32_t)v & 0xffffu) | (uint32_t)v << 16; }
inline uint32_t pk2v(int v) { return pk2(v); }
// ---- ksw_gapfill_dev.hpp ----
inline uint32_t gf_minu1(uint32_t a) { return pk_minu(a, 0x00010001u); }
inline uint32_t gf_max0(uint32_t a) { return pk_max(a, 0u); }
inline uint32_t gf_add1(uint32_t a) { return pk_add(a, 0x00010001u); }
inline uint32_t gf_sub_s(uint32_t a, uint32_t s) { return pk_sub(a, s); }
inline uint32_t gf_rsub_s(uint32_t s, uint32_t a) { return pk_sub(s, a); }
inline uint32_t gf_mad_vsv(uint32_t a, uint32_t s, uint32_t c) { return pk_mad(a, s, c); }
inline uint32_t gf_mad_vv1(uint32_t a, uint32_t b) { return pk_mad(a, b, 0x00010001u); }
inline uint32_t gf_mad8(uint32_t a, uint32_t c) { return pk_mad(a, pk_emu::both(8), c); }
inline uint32_t gf_mad16(uint32_t a, uint32_t c) { return pk_mad(a, pk_emu::both(16), c); }
inline uint32_t gf_mad32(uint32_t a, uint32_t c) { return pk_mad(a, pk_emu::both(32), c); }
inline uint32_t gf_mad64(uint32_t a, uint32_t c) { return pk_mad(a, pk_emu::both(64), c); }
inline uint32_t gf_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13c, 0xf, 0xf, false); } // wave_ror:1
inline uint32_t gf_sext8(uint32_t v) { return ((uint32_t)(uint16_t)(int16_t)(int8_t)v) | (uint32_t)(uint16_t)(int16_t)(int8_t)(v >> 16) << 16; }
inline uint32_t gf_asr3(uint32_t v) { return ((uint32_t)(uint16_t)((int16_t)(uint16_t)v >> 3)) | (uint32_t)(uint16_t)((int16_t)(uint16_t)(v >> 16) >> 3) << 16; }
// gf_cell: the same operations as the three assembly blocks, in their order
inline void gf_cell(uint32_t x1, uint32_t o1, uint32_t xp, uint32_t vp, uint32_t x2p, uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y, uint32_t &x2, uint32_t &y2, uint32_t &d,
                    uint32_t P_MCH, uint32_t S_MISD, uint32_t S_SCN, uint32_t S_Q, uint32_t S_Q2, uint32_t S_QE, uint32_t S_QE2)
{
	const uint32_t ONE = 0x00010001u;
	uint32_t a = pk_add(xp, vp), m = pk_minu(x1, ONE), b = pk_add(y, u), a2 = pk_add(x2p, vp);
	uint32_t z = pk_mad(m, S_MISD, P_MCH);
	uint32_t b2 = pk_add(y2, u), tA = pk_max(a, b), n = pk_shr2(o1), tB = pk_max(a2, b2), w = pk_sub(S_SCN, z);
	tA = pk_max(tA, tB);
	z = pk_mad(n, w, z);
	const uint32_t z4 = pk_max(z, tA);
	uint32_t d0 = pk_sub(z4, z), d1 = pk_sub(z4, a), d2 = pk_sub(z4, b), d3 = pk_sub(z4, a2);
	const uint32_t zc = pk_min(z4, P_MCH);
	d0 = pk_minu(d0, ONE), d1 = pk_minu(d1, ONE), d2 = pk_minu(d2, ONE), d3 = pk_minu(d3, ONE);
	const uint32_t un = pk_sub(zc, vp), vn = pk_sub(zc, u), t1 = pk_sub(zc, S_Q), t2 = pk_sub(zc, S_Q2);
	uint32_t e = pk_add(d3, ONE);
	a = pk_sub(a, t1), b = pk_sub(b, t1);
	e = pk_mad(d2, e, ONE);
	a2 = pk_sub(a2, t2), b2 = pk_sub(b2, t2);
	e = pk_mad(d1, e, ONE);
	a = pk_max(a, 0u), b = pk_max(b, 0u);
	e = pk_mul(d0, e);
	a2 = pk_max(a2, 0u), b2 = pk_max(b2, 0u);
	const uint32_t fa = pk_minu(a, ONE), xn = pk_sub(a, S_QE), fb = pk_minu(b, ONE);
	e = pk_mad(fa, pk_emu::both(8), e);
	const uint32_t yn = pk_sub(b, S_QE), fa2 = pk_minu(a2, ONE);
	e = pk_mad(fb, pk_emu::both(16), e);
	const uint32_t x2n = pk_sub(a2, S_QE2), fb2 = pk_minu(b2, ONE);
	e = pk_mad(fa2, pk_emu::both(32), e);
	const uint32_t y2n = pk_sub(b2, S_QE2);
	e = pk_mad(fb2, pk_emu::both(64), e);
	u = un, v = vn, x = xn, y = yn, x2 = x2n, y2 = y2n, d = e;
}
// gf_cell_k: the keyed cell (values times 8, candidate tags in the low three bits, the clamp and -(q + e) folded into one constant), the same
// operations as its three assembly blocks; K = GfK (ksw_gapfill_dev.hpp, declared after this header is read)
template <class KT>
inline void gf_cell_k(uint32_t x1, uint32_t o1, uint32_t xp, uint32_t vp, uint32_t x2p, uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y, uint32_t &x2, uint32_t &y2, uint32_t &d,
                      uint32_t P_MCHT, const KT &K)
{
	using pk_emu::both;
	uint32_t a = pk_add(xp, vp), m = pk_minu(x1, both(1)), b = pk_add(y, u), a2 = pk_add(x2p, vp);
	uint32_t z = pk_mad(m, K.misd8, P_MCHT);
	uint32_t b2 = pk_add(y2, u), tA = pk_max(a, b), n = pk_shr2(o1), tB = pk_max(a2, b2), w = pk_sub(K.scnt, z);
	tA = pk_max(tA, tB);
	z = pk_mad(n, w, z);
	const uint32_t z4 = pk_max(z, tA);
	const uint32_t zv = z4 & 0xfff8fff8u;
	uint32_t e = z4 & 0x00070007u;
	const uint32_t zc = pk_min(zv, K.mch8);
	const uint32_t vn = pk_sub(zc, u), t1 = pk_add(zc, K.e8), t2 = pk_add(zc, K.e28);
	u = pk_sub(zc, vp);
	a = pk_sub(a, t1), b = pk_sub(b, t1), a2 = pk_sub(a2, t2), b2 = pk_sub(b2, t2);
	const uint32_t xn = pk_max(a, K.ka), yn = pk_max(b, K.kb), x2n = pk_max(a2, K.ka2), y2n = pk_max(b2, K.kb2);
	uint32_t fa = pk_min(xn, K.ka8), fa2 = pk_min(x2n, K.ka28);
	const uint32_t fb = pk_min(yn, K.kb8);
	fa = pk_mad(fb, both(2), fa);
	const uint32_t fb2 = pk_min(y2n, K.kb28);
	e = pk_add(e, fa);
	fa2 = pk_mad(fb2, both(2), fa2);
	e = pk_mad(fa2, both(4), e);
	v = vn, x = xn, y = yn, x2 = x2n, y2 = y2n, d = e;
}
template <class KT>
inline void gf_cell_k2(uint32_t x1, uint32_t o1, uint32_t xp, uint32_t vp, uint32_t x2p, uint32_t u, uint32_t y, uint32_t y2, uint32_t &un, uint32_t &v, uint32_t &x, uint32_t &yo, uint32_t &x2, uint32_t &y2o,
                       uint32_t &d, uint32_t P_MCHT, const KT &K)
{
	uint32_t uu = u, yy = y, yy2 = y2, vv, xx, xx2;
	gf_cell_k(x1, o1, xp, vp, x2p, uu, vv, xx, yy, xx2, yy2, d, P_MCHT, K);
	un = uu, v = vv, x = xx, yo = yy, x2 = xx2, y2o = yy2;
}
} // namespace mm2amd

// Register-resident splice gap-fill DP for gfx950: the device counterpart of ksw_exts2_sse (ksw2_exts2_sse.c:33-465) +
// ksw_backtrack (ksw2.h:130-162) for the calls that hold most of the DP cells of spliced alignment -- the global alignments
// between adjacent anchors across introns (align.c:840: flag KSW_EZ_APPROX_MAX): a query stretch of a few hundred bases against
// a target window as long as the intron (up to 200 kb with -x splice).
//
// ksw_exts2 has no band, so the valid cells of anti-diagonal r are exactly t in [max(0,r-qlen+1), min(tlen-1,r)], their
// neighbours are valid cells or the documented boundary values, and (under the scoring limits the host checks) no 8-bit
// overflow happens in a valid cell.  Only valid cells matter, so the layout is ours (the lane-exact SPLICE mode of
// ksw_extd2.hip remains the kernel for every other splice call and the one this kernel is tested against):
//
//   * lane = QUERY position: query base j lives in lane j%64 of register set j/64 for the whole job; the five difference
//     states never leave VGPRs.  Cell (r, j) sits on target position t = r - j: what the reference reads at t-1 (x, v, x2) was
//     written by the same lane one row earlier, what it reads at t (u, y) by lane j-1 -- one DPP wave shift each.
//   * the target streams through: every 64 rows the wave admits the next 64 target positions into an LDS ring, each entry
//     holding the base and the donor / acceptor cost of its position (ksw2_exts2_sse.c:120-194) ready for packed use.
//   * two jobs per wavefront in the halves of packed 16-bit registers, as in ksw_fast.hip (the launch is ordered by row count);
//     or, for queries longer than 256 (SELF): ONE job whose query positions j and j + 64*NC share a register -- the high
//     half is then a continuation of the low half (its left neighbour at lane 0 is the low half's last lane, its target
//     position lags 64*NC behind).  Queries beyond those 128*NC positions are cut into strips that sweep the target one
//     after the other, the (u, y) column between two strips going through HBM; register use stays at 4 waves per SIMD.
//   * the 1 B/cell direction matrix is the only HBM traffic (row-major by query position: 64 B coalesced per register set);
//     the traceback is done by the whole wave: lane k looks k cells ahead along the current run (match diagonal, gap, intron)
//     and one ballot tells how long the run lasts, so an intron of 50 000 bases costs 800 loads in sequence, not 50 000.
#include <hip/hip_runtime.h>
#include "hip_util.hpp"
#include "ksw_dev.hpp"
#include "ksw_pk.hpp"
#include <type_traits>

namespace mm2amd {

namespace {
struct SpliceParams { int is_for, has_strand, sp0, sp1, sp2, sp3; };

// donor cost of position t from the three bases after it, acceptor cost from the base at t and the two before it
// (ksw2_exts2_sse.c:133-170, the variants for the forward CIGAR order: gap fills are never reversed)
__device__ __forceinline__ int site_cost(const SpliceParams &P, int z) { return z < 0 ? 0 : z == 0 ? -P.sp0 : z == 1 ? -P.sp1 : z == 2 ? -P.sp2 : -P.sp3; }
__device__ __forceinline__ int donor_class(const SpliceParams &P, int c1, int c2, int c3)
{
	int z = 3;
	if (P.is_for) {
		if (c1 == 2 && c2 == 3) z = (c3 == 0 || c3 == 2) ? -1 : 0;
		else if (c1 == 2 && c2 == 1) z = 1;
		else if (c1 == 0 && c2 == 3) z = 2;
	} else {
		if (c1 == 1 && c2 == 3) z = (c3 == 0 || c3 == 2) ? -1 : 0;
		else if (c1 == 2 && c2 == 3) z = 2;
	}
	return z;
}
__device__ __forceinline__ int acceptor_class(const SpliceParams &P, int c0, int c1, int c2) // c0 = base at t, c1 = t-1, c2 = t-2
{
	int z = 3;
	if (P.is_for) {
		if (c1 == 0 && c0 == 2) z = (c2 == 1 || c2 == 3) ? -1 : 0;
		else if (c1 == 0 && c0 == 1) z = 2;
	} else {
		if (c1 == 0 && c0 == 1) z = (c2 == 1 || c2 == 3) ? -1 : 0;
		else if (c1 == 2 && c0 == 1) z = 1;
		else if (c1 == 0 && c0 == 3) z = 2;
	}
	return z;
}
}

// One register set and anti-diagonal of the recurrence (ksw2_exts2_sse.c:249-348, the left-aligned variant gap fills use), both packed halves: target entry (tv, dn, ac),
// query bases qv, (u, y) of the left neighbour one row up (up, yp); v, x, x2 in / out, u, y, d out.  The interior rows' loop (splice_lean_rows below) uses it: the general
// row body further down writes the same operations with the one-instruction helpers.  As in ksw_gapfill_dev.hpp no instruction reads the result of the packed instruction
// right before it, and an instruction takes at most one scalar operand.
struct SpliceK { uint32_t misd, scn, q, q2, qe; }; // packed launch constants that stay in SGPRs
#ifndef MM2AMD_WAVE_EMU
__device__ __forceinline__ void splice_cell(uint32_t tv, uint32_t dn, uint32_t ac, uint32_t qv, uint32_t up, uint32_t yp, uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y, uint32_t &x2,
                                            uint32_t &d, uint32_t P_MCH, const SpliceK &K)
{
	uint32_t m, n, a, b, a2, t1, z, z3; // m: tv ^ qv, then "bases differ", then N score - z;  n: tv | qv, then "a base is N";  t1: a2 + acceptor, then the gap candidates' maximum
	asm volatile(
		"v_xor_b32 %[m], %[tv], %[qv]\n\t"
		"v_or_b32 %[n], %[tv], %[qv]\n\t"
		"v_pk_add_u16 %[a], %[x], %[v]\n\t"
		"v_pk_min_u16 %[m], %[m], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_add_u16 %[b], %[yp], %[up]\n\t"
		"v_pk_lshrrev_b16 %[n], 2, %[n] op_sel_hi:[0,1]\n\t"
		"v_pk_mad_u16 %[z], %[m], %[misd], %[mch]\n\t"
		"v_pk_add_u16 %[a2], %[x2], %[v]\n\t"
		"v_pk_max_i16 %[z3], %[a], %[b]\n\t"
		"v_pk_sub_u16 %[m], %[scn], %[z]\n\t"
		"v_pk_add_u16 %[t1], %[a2], %[ac]\n\t"
		"v_pk_mad_u16 %[z], %[n], %[m], %[z]\n\t"
		"v_pk_max_i16 %[t1], %[z3], %[t1]\n\t"
		"s_nop 0\n\t"
		"v_pk_max_i16 %[z3], %[z], %[t1]\n\t"
		"s_nop 0"
		: [m] "=&v"(m), [n] "=&v"(n), [a] "=&v"(a), [b] "=&v"(b), [a2] "=&v"(a2), [t1] "=&v"(t1), [z] "=&v"(z), [z3] "=&v"(z3)
		: [tv] "v"(tv), [qv] "v"(qv), [x] "v"(x), [v] "v"(v), [x2] "v"(x2), [yp] "v"(yp), [up] "v"(up), [ac] "v"(ac), [mch] "v"(P_MCH), [misd] "s"(K.misd), [scn] "s"(K.scn));
	uint32_t es, ea, eb, un, vn, tq, tq2, ma, mb, m2; // ea ends as the direction byte; tq / tq2 / es end as the three "goes on" flags; a / b / a2 end as the new x / y / x2
	asm volatile(
		"v_pk_sub_u16 %[es], %[z3], %[z]\n\t"
		"v_pk_sub_u16 %[ea], %[z3], %[a]\n\t"
		"v_pk_sub_u16 %[eb], %[z3], %[b]\n\t"
		"v_pk_sub_u16 %[un], %[z3], %[v]\n\t"
		"v_pk_min_u16 %[es], %[es], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_min_u16 %[ea], %[ea], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_min_u16 %[eb], %[eb], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_sub_u16 %[vn], %[z3], %[up]\n\t"
		"v_pk_sub_u16 %[tq], %[z3], %[pq]\n\t"
		"v_pk_add_u16 %[eb], %[eb], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_sub_u16 %[tq2], %[z3], %[pq2]\n\t"
		"v_pk_sub_u16 %[a], %[a], %[tq]\n\t"
		"v_pk_mad_u16 %[ea], %[ea], %[eb], 1 op_sel_hi:[1,1,0]\n\t"
		"v_pk_sub_u16 %[b], %[b], %[tq]\n\t"
		"v_pk_sub_u16 %[a2], %[a2], %[tq2]\n\t"
		"v_pk_mul_lo_u16 %[ea], %[es], %[ea]\n\t"
		"v_pk_max_i16 %[ma], %[a], 0\n\t"
		"v_pk_max_i16 %[mb], %[b], 0\n\t"
		"v_pk_max_i16 %[m2], %[a2], %[dn]\n\t"
		"v_pk_min_u16 %[tq], %[ma], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_min_u16 %[tq2], %[mb], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_sub_u16 %[es], %[m2], %[dn]\n\t"
		"v_pk_sub_u16 %[a], %[ma], %[pqe]\n\t"
		"v_pk_mad_u16 %[ea], %[tq], 8, %[ea] op_sel_hi:[1,0,1]\n\t"
		"v_pk_min_u16 %[es], %[es], 1 op_sel_hi:[1,0]\n\t"
		"v_pk_sub_u16 %[b], %[mb], %[pqe]\n\t"
		"v_pk_mad_u16 %[ea], %[tq2], 16, %[ea] op_sel_hi:[1,0,1]\n\t"
		"v_pk_sub_u16 %[a2], %[m2], %[pq2]\n\t"
		"v_pk_mad_u16 %[ea], %[es], 32, %[ea] op_sel_hi:[1,0,1]"
		: [es] "=&v"(es), [ea] "=&v"(ea), [eb] "=&v"(eb), [un] "=&v"(un), [vn] "=&v"(vn), [tq] "=&v"(tq), [tq2] "=&v"(tq2), [ma] "=&v"(ma), [mb] "=&v"(mb), [m2] "=&v"(m2),
		  [a] "+v"(a), [b] "+v"(b), [a2] "+v"(a2)
		: [z3] "v"(z3), [z] "v"(z), [v] "v"(v), [up] "v"(up), [dn] "v"(dn), [pq] "s"(K.q), [pq2] "s"(K.q2), [pqe] "s"(K.qe));
	u = un, v = vn, x = a, y = b, x2 = a2, d = ea;
}
__device__ __forceinline__ uint32_t splice_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13c /* wave_ror:1 */, 0xf, 0xf, false); }
#else // the emulator's twin: the same operations through the helpers
inline void splice_cell(uint32_t tv, uint32_t dn, uint32_t ac, uint32_t qv, uint32_t up, uint32_t yp, uint32_t &u, uint32_t &v, uint32_t &x, uint32_t &y, uint32_t &x2, uint32_t &d, uint32_t P_MCH, const SpliceK &K)
{
	const uint32_t ONE = 0x00010001u;
	uint32_t z = pk_mad(pk_minu(tv ^ qv, ONE), K.misd, P_MCH);
	z = pk_mad(pk_shr2(tv | qv), pk_sub(K.scn, z), z);
	const uint32_t vt = v;
	uint32_t a = pk_add(x, vt), b = pk_add(yp, up), a2 = pk_add(x2, vt);
	const uint32_t a2a = pk_add(a2, ac);
	const uint32_t z3 = pk_max(pk_max(pk_max(z, a), b), a2a);
	const uint32_t ne_s = pk_minu(pk_sub(z3, z), ONE), ne_a = pk_minu(pk_sub(z3, a), ONE), ne_b = pk_minu(pk_sub(z3, b), ONE);
	uint32_t dd = pk_mul(ne_s, pk_mad(ne_a, pk_add(ne_b, ONE), ONE));
	u = pk_sub(z3, vt), v = pk_sub(z3, up);
	const uint32_t tmp = pk_sub(z3, K.q);
	a = pk_sub(a, tmp), b = pk_sub(b, tmp), a2 = pk_sub(a2, pk_sub(z3, K.q2));
	const uint32_t ma = pk_max(a, 0u), mb = pk_max(b, 0u), m2 = pk_max(a2, dn);
	dd = pk_mad(pk_minu(ma, ONE), pk_emu::both(8), dd), dd = pk_mad(pk_minu(mb, ONE), pk_emu::both(16), dd), dd = pk_mad(pk_minu(pk_sub(m2, dn), ONE), pk_emu::both(32), dd);
	x = pk_sub(ma, K.qe), y = pk_sub(mb, K.qe), x2 = pk_sub(m2, K.q2), d = dd;
}
inline uint32_t splice_ror1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13c, 0xf, 0xf, false); }
#endif

constexpr int splice_ring(int nc, bool self) { int need = (self ? 128 : 64) * nc + 64, p = 128; while (p < need) p <<= 1; return p; }
constexpr int splice_wpb(int nc, bool self) { const int b = splice_ring(nc, self) * (self ? 8 : 16); return b * 4 <= 65536 ? 4 : b * 2 <= 65536 ? 2 : 1; }

template <int NC, bool SELF>
__global__ void __launch_bounds__(64 * splice_wpb(NC, SELF), (NC <= 4 ? 4 : 2) * 4 / splice_wpb(NC, SELF)) ksw_splice_kernel(KswLaunch L)
{
	constexpr int RING = splice_ring(NC, SELF), RM = RING - 1, WPB = splice_wpb(NC, SELF), QOFF = 64 * NC;
	// per wave and target position: {base, donor, acceptor}; paired: 16 B of packed halves (job A low, job B high);
	// SELF: 8 B {base | donor << 16, acceptor} of the one job, read at two positions and merged
	__shared__ __attribute__((aligned(16))) uint8_t s_raw[WPB * RING * (SELF ? 8 : 16)];
	const int lane = threadIdx.x & 63, wave_in_block = threadIdx.x >> 6;
	const int slot = blockIdx.x * WPB + wave_in_block;
	uint4 *ring = (uint4 *)s_raw + (SELF ? 0 : wave_in_block * RING);
	uint2 *ring2 = (uint2 *)s_raw + (SELF ? wave_in_block * RING : 0);
	const int m = L.sc.m;
	const int q = L.sc.q, e = L.sc.e, q2 = L.sc.q2, qe = q + e;
	const int sc_mch = L.sc.mat[0], sc_mis = L.sc.mat[1];
	const int sc_N = L.sc.mat[m * m - 1] == 0 ? -e : L.sc.mat[m * m - 1];
	int long_thres = (q2 - q) / e - 1; // ksw2_exts2_sse.c:98-101
	if (q2 > q + e + long_thres * e) ++long_thres;
	const int long_diff = long_thres * e - (q2 - q);
	// (only the match score is pinned in a VGPR: the interior rows' cell takes its other constants as scalar or inline operands; the general row body, which runs on
	// the rows near the matrix' corners only, lets the compiler copy them where its one-instruction helpers want a VGPR)
	const uint32_t P_ONE = pk2(1), P_ZERO = pk2(0), P_MCH = pk2v(sc_mch), P_MISD = pk2(sc_mis - sc_mch), P_SCN = pk2(sc_N);
	const uint32_t P_Q = pk2(q), P_Q2 = pk2(q2), P_QE = pk2(qe), P_NQE = pk2(-qe), P_NQ2 = pk2(-q2);
	const uint32_t P_8 = pk2(8), P_16 = pk2(16), P_32 = pk2(32);

	for (;;) {
		int pid = 0;
		if (lane == 0) pid = atomicAdd(L.counter, 1);
		pid = __builtin_amdgcn_readfirstlane(pid);
		if ((SELF ? pid : 2 * pid) >= L.n_jobs) break;
		const int jidA = SELF ? pid : 2 * pid, jidB = SELF ? pid : 2 * pid + 1;
		const bool hasB = !SELF && jidB < L.n_jobs; // a second JOB in the high halves
		const KswJob JA = L.jobs[jidA], JB = L.jobs[hasB ? jidB : jidA];
		const int qlenA = JA.qlen, tlenA = JA.tlen, qlenB = hasB ? JB.qlen : 0, tlenB = hasB ? JB.tlen : 0;
		const int qsA = (qlenA + 63) & ~63, qsB = SELF ? qsA : (qlenB + 63) & ~63; // row stride of the direction matrices
		uint8_t *dirA = L.dir_pool + (size_t)(SELF ? slot : 2 * slot) * L.slot_bytes, *dirB = SELF ? dirA + QOFF : dirA + L.slot_bytes;
		auto splice_params = [&](int flag) {
			SpliceParams P;
			P.is_for = (flag & KSW_SPLICE_FOR) ? 1 : 0, P.has_strand = (flag & (KSW_SPLICE_FOR | KSW_SPLICE_REV)) ? 1 : 0;
			if (flag & KSW_SPLICE_CMPLX) P.sp0 = 3, P.sp1 = 5, P.sp2 = 7, P.sp3 = 10; // (int)({8,15,21,30} / 3. + .499)
			else P.sp0 = (flag & KSW_SPLICE_FLANK) ? L.sc.noncan / 2 : 0, P.sp1 = P.sp2 = P.sp3 = L.sc.noncan;
			return P;
		};
		const SpliceParams SA = splice_params(JA.flag), SB = splice_params(JB.flag);
		auto tfetch = [&](const KswJob &J, int t) -> int {
			const uint64_t pos = (J.flag & KSWJ_T_REVERSED) ? J.t_off - (uint64_t)t : J.t_off + (uint64_t)t;
			return (J.flag & KSWJ_T_PACKED) ? (int)(L.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : (int)L.tpool[pos];
		};
		// base, donor and acceptor cost of target position t of one job (0 costs when no transcript strand is assumed)
		auto target_entry = [&](const KswJob &J, const SpliceParams &P, int tlen, int t, int &dn, int &ac) -> int {
			dn = ac = 0;
			if (t >= tlen) return 4;
			const int c0 = tfetch(J, t);
			if (P.has_strand) {
				int zd = 3, za = 3;
				if (t < tlen - 4) zd = donor_class(P, tfetch(J, t + 1), tfetch(J, t + 2), tfetch(J, t + 3));
				if (t >= 2) za = acceptor_class(P, c0, tfetch(J, t - 1), tfetch(J, t - 2));
				dn = site_cost(P, zd), ac = site_cost(P, za);
			}
			return c0;
		};
		// Corner score H(tlen-1, qlen-1) of each job, summed along the matrix border (path-independent; see ksw_fast.hip): u of the
		// first query row's cells while the anti-diagonal still starts a new target column, then v down the last column.
		int H0A = -qe, H0B = -qe;
		// SELF: a query longer than the 2*QOFF positions the registers hold is processed in STRIPS of that many positions, one full
		// sweep over the target per strip.  What a strip's first query position reads from its left neighbour -- (u, y) of the
		// previous strip's last position, one pair per target position -- is handed over through a per-slot array in HBM (the
		// slot's CIGAR scratch, free until the traceback), written and read 64 positions at a time.
		constexpr int STRIP = 2 * QOFF;
		const int qlen_all = qlenA, n_pass = SELF ? (qlen_all + STRIP - 1) / STRIP : 1;
		uint32_t *handover = L.cigar_tmp + (size_t)(SELF ? slot : 2 * slot) * L.cigar_tmp_cap;
		for (int pass = 0; pass < n_pass; ++pass) {
		const int QB = pass * STRIP;                                             // first query position of this strip
		const int qlenA = SELF ? (qlen_all - QB < STRIP ? qlen_all - QB : STRIP) : qlen_all; // shadows the job's query length inside the strip
		const bool hand_on = SELF && pass + 1 < n_pass;
		// ---- operands: one packed query base pair per (register set, lane); the states start at the values of :108-109 ----
		uint32_t Q[NC], U[NC], V[NC], X[NC], Y[NC], X2[NC];
#pragma unroll
		for (int c = 0; c < NC; ++c) {
			const int j = c * 64 + lane;
			uint32_t bA = 4, bB = 4;
			if (j < qlenA) bA = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)(QB + j) : JA.q_off + (uint64_t)(QB + j)];
			if (SELF) { if (j + QOFF < qlenA) bB = L.qpool[(JA.flag & KSWJ_Q_REVERSED) ? JA.q_off - (uint64_t)(QB + j + QOFF) : JA.q_off + (uint64_t)(QB + j + QOFF)]; }
			else if (j < qlenB) bB = L.qpool[(JB.flag & KSWJ_Q_REVERSED) ? JB.q_off - (uint64_t)j : JB.q_off + (uint64_t)j];
			Q[c] = bA | bB << 16;
			U[c] = V[c] = X[c] = Y[c] = P_NQE, X2[c] = P_NQ2;
		}
		int frontier = -1; // target positions <= frontier are in the ring
		uint32_t hand_in = 0, hand_out = 0;
		const int n_rowsA = qlenA + tlenA - 1, n_rowsB = hasB ? qlenB + tlenB - 1 : 0, n_rows = n_rowsA > n_rowsB ? n_rowsA : n_rowsB;
		// ---- the interior rows [R0, R1): every query position of both jobs has a cell (the anti-diagonal has left the first query row's start and not reached the last
		//      target column), no column starts, the border terms are at their constants (ksw2_exts2_sse.c:234-247 with r > long_thres).  With a target as long as an intron
		//      that is nearly every row, and none of what the general row body below derives per row -- ranges, masks, edge patches, the sets in use -- changes there.
		const int qwide = SELF ? (qlenA < QOFF ? qlenA : QOFF) : (qlenA > qlenB ? qlenA : qlenB);
		const int n_act = (qwide - 1) / 64 + 1;                                                     // register sets in use
		const int nsA = SELF ? n_act : qsA >> 6, nsB = SELF ? (qlenA > QOFF ? (qlenA - QOFF + 63) >> 6 : 0) : hasB ? qsB >> 6 : 0; // sets whose bytes a job's matrix has columns for
		int R0 = qlenA > qlenB ? qlenA : qlenB, R1 = (hasB && tlenB < tlenA ? tlenB : tlenA) - 1;
		if (R0 < long_thres + 1 - QB) R0 = long_thres + 1 - QB;
		const SpliceK KC = { pk2(sc_mis - sc_mch), pk2(sc_N), pk2(q), pk2(q2), pk2(qe) };
		auto lean_rows = [&](auto na_tag, int r0, int r1) {
			constexpr int NA = decltype(na_tag)::value;
			uint32_t hacc = 0;              // (u, u) of query position 0 summed over the block's rows (at most 64 rows of a few units each: no 16-bit overflow)
			uint32_t up0 = 0, yp0 = P_NQE;  // paired jobs: lane 0 keeps the border's (u, y) through the shifts (wave_shr:1 never writes it)
			uint32_t ra = (uint32_t)(r0 - lane) & RM;
			uint8_t *pA = dirA + (size_t)(r0 + QB) * qsA + QB + lane, *pB = SELF ? pA + QOFF : dirB + (size_t)r0 * qsB + lane;
			for (int r = r0; r < r1; ++r) {
				uint32_t sU = 0, sY = 0;
				if (SELF) { // query position QOFF continues the low halves' last lane; position 0 has the border or the previous strip to its left
					uint32_t lU = 0, lY = P_NQE & 0xffffu;
					if (pass > 0) { const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hand_in, r & 63); lU = h & 0xffffu, lY = h >> 16; }
					sU = splice_ror1(U[NC - 1]) << 16 | lU, sY = splice_ror1(Y[NC - 1]) << 16 | lY;
				}
				// set c's left neighbour at lane 0 is the last lane of set c - 1, one row up: all of them rotated into place before any set computes (a DPP move may not
				// follow the write of its source by less than two instructions)
				uint32_t cU[NA], cY[NA];
#pragma unroll
				for (int c = 1; c < NA; ++c) cU[c] = splice_ror1(U[c - 1]), cY[c] = splice_ror1(Y[c - 1]);
				// a set's target entries are read while the set before it computes (the LDS round trip is as long as a cell)
				uint4 te_n = make_uint4(0u, 0u, 0u, 0u);
				uint2 el_n = make_uint2(0u, 0u), eh_n = make_uint2(0u, 0u);
				if (SELF) el_n = ring2[(ra - 64u * (NA - 1)) & RM], eh_n = ring2[(ra - 64u * (NA - 1) - QOFF) & RM];
				else te_n = ring[(ra - 64u * (NA - 1)) & RM];
#pragma unroll
				for (int c = NA - 1; c >= 0; --c) { // highest set first: set c - 1 still holds row r - 1
					uint32_t up, yp;
					if (c > 0) up = dpp_shr1u(cU[c], U[c]), yp = dpp_shr1u(cY[c], Y[c]);
					else if (SELF) up = dpp_shr1u(sU, U[0]), yp = dpp_shr1u(sY, Y[0]);
					else up = up0 = dpp_shr1u(up0, U[0]), yp = yp0 = dpp_shr1u(yp0, Y[0]);
					uint32_t tv, dn, ac, d;
					if (SELF) {
						const uint2 el = el_n, eh = eh_n;
						if (c > 0) el_n = ring2[(ra - 64u * (c - 1)) & RM], eh_n = ring2[(ra - 64u * (c - 1) - QOFF) & RM];
						tv = __builtin_amdgcn_perm(eh.x, el.x, 0x05040100u), dn = __builtin_amdgcn_perm(eh.x, el.x, 0x07060302u), ac = __builtin_amdgcn_perm(eh.y, el.y, 0x05040100u);
					} else {
						const uint4 te = te_n;
						if (c > 0) te_n = ring[(ra - 64u * (c - 1)) & RM];
						tv = te.x, dn = te.y, ac = te.z;
					}
					splice_cell(tv, dn, ac, Q[c], up, yp, U[c], V[c], X[c], Y[c], X2[c], d, P_MCH, KC);
					if (c < nsA) pA[c * 64] = (uint8_t)d;          // (a set's lanes beyond the query fall into the row's padding)
					if (c < nsB) pB[c * 64] = (uint8_t)(d >> 16);
				}
				if (!SELF || pass == 0) hacc = pk_add(hacc, U[0]);
				if (hand_on) { // as in the general row body
					const int th = r - (STRIP - 1);
					const uint32_t hv = (U[NC - 1] >> 16) | (Y[NC - 1] & 0xffff0000u);
					const uint32_t h = (uint32_t)__builtin_amdgcn_readlane((int)hv, 63);
					if (lane == (th & 63)) hand_out = h;
					if ((th & 63) == 63) handover[(th & ~63) + lane] = hand_out;
				}
				ra = (ra + 1) & RM, pA += qsA, pB += SELF ? qsA : qsB;
			}
			const uint32_t hs = (uint32_t)__builtin_amdgcn_readlane((int)hacc, 0);
			H0A += (int16_t)hs;
			if (!SELF) H0B += (int16_t)(hs >> 16);
		};
		for (int r = 0; r < n_rows; ++r) { // r: anti-diagonal within the strip; the matrix's anti-diagonal is r + QB
			if (r > frontier) { // admit the next 64 target positions (wave-uniform)
				const int t = frontier + 1 + lane;
				int dnA, acA, dnB = 0, acB = 0;
				const int cA = target_entry(JA, SA, tlenA, t, dnA, acA), cB = SELF ? 0 : target_entry(JB, SB, tlenB, t, dnB, acB);
				if (SELF) ring2[t & RM] = make_uint2((uint32_t)cA | (uint32_t)dnA << 16, (uint32_t)acA & 0xffffu);
				else ring[t & RM] = make_uint4((uint32_t)cA | (uint32_t)cB << 16, ((uint32_t)dnA & 0xffffu) | (uint32_t)dnB << 16, ((uint32_t)acA & 0xffffu) | (uint32_t)acB << 16, 0u);
				frontier += 64;
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
				if (SELF && pass > 0) hand_in = t < tlenA ? handover[t] : 0u; // (u | y << 16) left of this strip, for target positions r..r+63
			}
			if (r >= R0 && r < R1) { // interior rows up to the next admission
				const int stop = R1 < frontier + 1 ? R1 : frontier + 1;
				if (NC >= 4 && n_act == 4) lean_rows(std::integral_constant<int, 4>(), r, stop);
				else if (NC >= 4 && n_act == 3) lean_rows(std::integral_constant<int, 3>(), r, stop);
				else if (n_act == 2) lean_rows(std::integral_constant<int, 2>(), r, stop);
				else lean_rows(std::integral_constant<int, 1>(), r, stop);
				r = stop - 1;
				continue;
			}
			// query positions with a valid cell on this anti-diagonal, per job: j in [max(0, r-tlen+1), min(qlen-1, r)]
			int jloA = r - tlenA + 1 > 0 ? r - tlenA + 1 : 0, jhiA = r < qlenA - 1 ? r : qlenA - 1;
			int jloB = r - tlenB + 1 > 0 ? r - tlenB + 1 : 0, jhiB = r < qlenB - 1 ? r : qlenB - 1;
			if (r >= n_rowsA) jloA = 1, jhiA = 0;
			if (r >= n_rowsB) jloB = 1, jhiB = 0;
			if (SELF) jloB = jloA - QOFF, jhiB = jhiA - QOFF; // the high halves hold query positions lane + QOFF of the same job
			const uint32_t wA = (uint32_t)(jhiA - jloA + 1), wB = (uint32_t)(jhiB - jloB + 1);
			int lo = jloA <= jhiA ? (jloB <= jhiB && jloB < jloA ? jloB : jloA) : jloB, hi = jhiA > jhiB ? jhiA : jhiB;
			if (SELF) lo = 0, hi = jhiA < QOFF ? jhiA : QOFF - 1; // register sets 0..: the low halves' range covers the high halves' (both start at their lane 0 region)
			// v[-1] / u[r] on the matrix border (ksw2_exts2_sse.c:234-247): depends on r only
			const int rg = r + QB;
			const int bnd = rg == 0 ? -qe : rg < long_thres ? -e : rg == long_thres ? long_diff : 0;
			const uint32_t P_BND = pk2(bnd);
			// query position r starts its column on this row (t = 0): x, v, x2 of "t = -1" are border values
			const bool newA = r < qlenA && r < n_rowsA, newB = r < qlenB && r < n_rowsB;
			const int edge_j = SELF && r >= QOFF ? r - QOFF : r;
			const int edge_set = edge_j >> 6, edge_lane = edge_j & 63;
			const uint32_t edge_halves = SELF ? (newA ? (r >= QOFF ? 0xffff0000u : 0xffffu) : 0u) : (newA ? 0xffffu : 0u) | (newB ? 0xffff0000u : 0u);
			// SELF: the high halves continue the low halves -- query position QOFF's left neighbour is the last lane of the last set
			uint32_t selfU = 0, selfY = 0;
			if (SELF) selfU = (uint32_t)__builtin_amdgcn_readlane(U[NC - 1], 63) << 16, selfY = (uint32_t)__builtin_amdgcn_readlane(Y[NC - 1], 63) << 16;
			const bool topA = pass == 0 && r < tlenA && r < n_rowsA, topB = r < tlenB && r < n_rowsB; // query position 0 still has a cell (t = r)
			const int lastA = r - tlenA + 1, lastB = r - tlenB + 1;                    // query position on the last target column
			uint8_t *prA = dirA + (size_t)rg * qsA + QB, *prB = dirB + (size_t)rg * qsB + QB;
			uint32_t leftU = P_BND & 0xffffu, leftY = P_NQE & 0xffffu; // SELF: what query position QB - 1 hands to position QB
			if (SELF && pass > 0) { const uint32_t h = (uint32_t)__builtin_amdgcn_readlane(hand_in, r & 63); leftU = h & 0xffffu, leftY = h >> 16; }
			// register sets from the highest down, so that set c still sees row r-1 in set c-1 when it fetches its carry-ins
#pragma unroll
			for (int c = NC - 1; c >= 0; --c) {
				if (c * 64 > hi || c * 64 + 63 < lo) continue; // register set outside both anti-diagonals (uniform)
				const int j = c * 64 + lane;
				const bool actA = (uint32_t)(j - jloA) < wA, actB = (uint32_t)(j - jloB) < wB;
				uint32_t cU = P_BND, cY = P_NQE; // query position -1: the matrix border (u[r], y[r], :241-247)
				if (c > 0) cU = __builtin_amdgcn_readlane(U[c - 1], 63), cY = __builtin_amdgcn_readlane(Y[c - 1], 63);
				else if (SELF) cU = leftU | selfU, cY = leftY | selfY;
				const uint32_t up = dpp_shr1u(cU, U[c]), yp = dpp_shr1u(cY, Y[c]);
				if (edge_halves && edge_set == c) {
					const uint32_t em = lane == edge_lane ? edge_halves : 0u;
					V[c] = bfi(em, P_BND, V[c]), X[c] = bfi(em, P_NQE, X[c]), X2[c] = bfi(em, P_NQ2, X2[c]);
				}
				{
					// every lane computes, active or not (see ksw_fast.hip: a lane's registers are only read while its cell, or the
					// next cell of its right neighbour, is valid); only the stores are guarded
					uint32_t tv, dn, ac;
					if (SELF) {
						const uint2 el = ring2[(r - j) & RM], eh = ring2[(r - j - QOFF) & RM];
						tv = __builtin_amdgcn_perm(eh.x, el.x, 0x05040100u), dn = __builtin_amdgcn_perm(eh.x, el.x, 0x07060302u), ac = __builtin_amdgcn_perm(eh.y, el.y, 0x05040100u);
					} else {
						const uint4 te = ring[(r - j) & RM];
						tv = te.x, dn = te.y, ac = te.z;
					}
					const uint32_t qv = Q[c];
					uint32_t z = pk_mad(pk_minu(tv ^ qv, P_ONE), P_MISD, P_MCH);
					z = pk_mad(pk_shr2(tv | qv), pk_sub(P_SCN, z), z);
					const uint32_t vt = V[c];
					uint32_t a = pk_add(X[c], vt), b = pk_add(yp, up), a2 = pk_add(X2[c], vt);
					const uint32_t a2a = pk_add(a2, ac);
					const uint32_t z1 = pk_max(z, a), z2 = pk_max(z1, b), z3 = pk_max(z2, a2a);
					// d = index of the first of (s, a, b, a2a) equal to the maximum (the strictly-greater chain of :312-318)
					const uint32_t ne_s = pk_minu(pk_sub(z3, z), P_ONE), ne_a = pk_minu(pk_sub(z3, a), P_ONE), ne_b = pk_minu(pk_sub(z3, b), P_ONE);
					uint32_t d = pk_mul(ne_s, pk_mad(ne_a, pk_add(ne_b, P_ONE), P_ONE));
					U[c] = pk_sub(z3, vt), V[c] = pk_sub(z3, up); // no clamp in this recurrence
					uint32_t tmp = pk_sub(z3, P_Q);
					a = pk_sub(a, tmp), b = pk_sub(b, tmp);
					a2 = pk_sub(a2, pk_sub(z3, P_Q2));
					const uint32_t ma = pk_max(a, P_ZERO), mb = pk_max(b, P_ZERO), m2 = pk_max(a2, dn);
					d = pk_mad(pk_minu(ma, P_ONE), P_8, d);              // a > 0, b > 0: the gap can be extended (:333-338)
					d = pk_mad(pk_minu(mb, P_ONE), P_16, d);
					d = pk_mad(pk_minu(pk_sub(m2, dn), P_ONE), P_32, d); // a2 > donor: the intron goes on (:340-348)
					X[c] = pk_sub(ma, P_QE), Y[c] = pk_sub(mb, P_QE), X2[c] = pk_sub(m2, P_Q2);
					if (actA) prA[(uint32_t)j] = (uint8_t)d;
					if (actB) prB[(uint32_t)j] = (uint8_t)(d >> 16);
				}
				if (topA) { if (c == 0) H0A += (int16_t)__builtin_amdgcn_readlane(U[c], 0); }
				else if (SELF) {
					if (r < n_rowsA && ((lastA >= QOFF ? lastA - QOFF : lastA) >> 6) == c) {
						const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(V[c], lastA & 63);
						H0A += lastA >= QOFF ? (int16_t)(v >> 16) : (int16_t)v;
					}
				} else if (r < n_rowsA && (lastA >> 6) == c) H0A += (int16_t)__builtin_amdgcn_readlane(V[c], lastA & 63);
				if (!SELF) {
					if (topB) { if (c == 0) H0B += (int16_t)(__builtin_amdgcn_readlane(U[c], 0) >> 16); }
					else if (r < n_rowsB && (lastB >> 6) == c) H0B += (int16_t)(__builtin_amdgcn_readlane(V[c], lastB & 63) >> 16);
				}
			}
			if (hand_on) { // (u, y) of the strip's last query position (high half of the last lane) at target position r - (STRIP - 1)
				const int th = r - (STRIP - 1);
				if (th >= 0) {
					const uint32_t h = ((uint32_t)__builtin_amdgcn_readlane(U[NC - 1], 63) >> 16) | ((uint32_t)__builtin_amdgcn_readlane(Y[NC - 1], 63) & 0xffff0000u);
					if (lane == (th & 63)) hand_out = h;
					if ((th & 63) == 63 || r == n_rows - 1) { // in place: these positions were consumed STRIP - 1 rows ago
						const int tw = (th & ~63) + lane;
						if (tw <= th) handover[tw] = hand_out;
					}
				}
			}
		}
		if (hand_on) __threadfence_block(); // the next strip reads what other lanes of this wave wrote
		} // strips
		// ---- tracebacks from (tlen-1, qlen-1) (ksw2_exts2_sse.c:459-461; every cell on the way is inside the matrix), one job
		//      after the other, by the whole wave: a run is followed 64 cells at a time ----
		__threadfence_block();
		int n_cigA = 0, n_cigB = 0;
		uint32_t cig_offA = 0, cig_offB = 0;
#pragma unroll
		for (int which = 0; which < 2; ++which) {
			if (which == 1 && !hasB) break;
			const uint8_t *dir = which ? dirB : dirA;
			const int qs = which ? qsB : qsA;
			FastCig g = { L.cigar_tmp + (size_t)((SELF ? slot : 2 * slot) + which) * L.cigar_tmp_cap, 0, 0u };
			int i = (which ? tlenB : tlenA) - 1, j = (which ? qlenB : qlenA) - 1, state = 0;
			const uint32_t op3 = long_thres > 0 ? 3u : 2u; // min_intron_len = long_thres (ksw2.h:147-148)
			while (i >= 0 && j >= 0) {
				const int di = state == 2 ? 0 : 1, dj = (state == 0 || state == 2) ? 1 : 0;
				const int ii = i - lane * di, jj = j - lane * dj;
				const bool valid = ii >= 0 && jj >= 0;
				const int tmp = valid ? dir[(size_t)(ii + jj) * qs + jj] : 0;
				const bool cont = valid && (state == 0 ? (tmp & 7) == 0 : (tmp >> (state + 2) & 1) != 0);
				const unsigned long long stop = ~__ballot(cont);
				const int run = stop ? __builtin_ctzll(stop) : 64;
				if (run > 0) {
					fast_cig_push(g, state == 0 ? 0u : state == 2 ? 1u : state == 3 ? op3 : 2u, run);
					i -= run * di, j -= run * dj;
					continue;
				}
				state = __builtin_amdgcn_readfirstlane(tmp) & 7; // the run ends on this cell: it names the next state (ksw2.h:141-144)
				if (state == 0) fast_cig_push(g, 0, 1), --i, --j;
				else if (state == 1) fast_cig_push(g, 2, 1), --i;
				else if (state == 3) fast_cig_push(g, op3, 1), --i;
				else fast_cig_push(g, 1, 1), --j;
			}
			if (i >= 0) fast_cig_push(g, long_thres > 0 && i >= long_thres ? 3u : 2u, i + 1);
			if (j >= 0) fast_cig_push(g, 1, j + 1);
			if (g.n > 0 && lane == 0) g.c[g.n - 1] = g.last;
			uint32_t off = 0;
			if (g.n > 0 && lane == 0) off = atomicAdd(&L.cigar_cursor[0], (uint32_t)g.n);
			off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
			__threadfence_block();
			if (g.n > 0) { // forward order into the pool, all lanes copying
				if ((unsigned long long)off + (unsigned)g.n > L.cigar_pool_cap) { if (lane == 0) L.cigar_cursor[1] = 1; }
				else for (int k = lane; k < g.n; k += 64) L.cigar_pool[off + k] = g.c[g.n - 1 - k];
			}
			if (which) n_cigB = g.n, cig_offB = off; else n_cigA = g.n, cig_offA = off;
		}
		if (lane == 0 || (lane == 32 && hasB)) {
			const bool isB = lane >= 32;
			KswRes R;
			R.max = 0, R.zdropped = 0, R.max_q = R.max_t = -1, R.mqe = R.mte = KSW_NEG_INF, R.mqe_t = R.mte_q = -1;
			R.score = isB ? H0B : H0A, R.n_cigar = isB ? n_cigB : n_cigA, R.reach_end = 0, R.cigar_off = isB ? cig_offB : cig_offA;
			R.zd_max = KSW_ZD_NONE, R.zd_t0 = R.zd_t1 = R.zd_q0 = R.zd_q1 = -1; // the host scans the alignment (mm_test_zdrop)
			L.res[isB ? jidB : jidA] = R;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}

namespace {
template <int NC, bool SELF>
void launch_splice(const KswLaunch &L, int n_slots, hipStream_t s)
{
	constexpr int WPB = splice_wpb(NC, SELF);
	hipLaunchKernelGGL((ksw_splice_kernel<NC, SELF>), dim3((n_slots + WPB - 1) / WPB), dim3(64 * WPB), 0, s, L);
}
}

// n_sets register sets of 64 lanes; self: one job per wave using both register halves (queries up to 128 * n_sets), else two
void ksw_splice_launch(const KswLaunch &L, int n_slots, int n_sets, bool self, void *stream)
{
	if (L.n_jobs <= 0) return;
	hipStream_t s = (hipStream_t)stream;
	if (!self && n_sets == 2) launch_splice<2, false>(L, n_slots, s);
	else if (!self && n_sets == 4) launch_splice<4, false>(L, n_slots, s);
	else if (self && n_sets == 4) launch_splice<4, true>(L, n_slots, s);
	else throw std::runtime_error("[mm2amd] ksw_splice_launch: unsupported register-set count");
	HIP_CHECK(hipGetLastError());
}

} // namespace mm2amd

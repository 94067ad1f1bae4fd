// Issue-rate microbenchmark for the instructions the register-resident DP kernels are made of (ksw_fast.hip, ksw_splice.hip):
// packed 16-bit integer VOP3P, DPP wave shifts, v_readlane, LDS byte reads, byte / dword global stores.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/build/valu_issue_bench tools/valu_issue_bench.hip && tools/build/valu_issue_bench
//
// Every wave runs ITERS x 16 copies of one instruction on 8 independent registers (no dependency stalls with >= 2 waves per
// SIMD) between two s_memtime reads and records where it ran (HW_ID, XCC_ID); the table gives shader cycles per wave-instruction
// PER SIMD from per-SIMD accounting -- the waves that actually shared a SIMD, not an assumed count -- at 1..16 blocks per CU.  A full-rate wave64 VALU instruction on CDNA4 is 2
// cycles (SIMD-32, /opt/skills/guides/MI355X_MICROARCH.md "Wave scheduling").  The result is what DESIGN.md section 4 prices the
// DP kernels' VALU-issue roof with; the output of a run is kept under profiles/.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
#include <map>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 32768; // x16 instructions: milliseconds per launch, long enough for the clocks to settle after the warm-up

#define REP16_2OP(ins) \
	asm volatile(ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" \
	             ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n" \
	             ins " %0, %0, %8\n" ins " %1, %1, %8\n" ins " %2, %2, %8\n" ins " %3, %3, %8\n" \
	             ins " %4, %4, %8\n" ins " %5, %5, %8\n" ins " %6, %6, %8\n" ins " %7, %7, %8\n" \
	             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
#define REP16_3OP(ins) \
	asm volatile(ins " %0, %0, %8, %8\n" ins " %1, %1, %8, %8\n" ins " %2, %2, %8, %8\n" ins " %3, %3, %8, %8\n" \
	             ins " %4, %4, %8, %8\n" ins " %5, %5, %8, %8\n" ins " %6, %6, %8, %8\n" ins " %7, %7, %8, %8\n" \
	             ins " %0, %0, %8, %8\n" ins " %1, %1, %8, %8\n" ins " %2, %2, %8, %8\n" ins " %3, %3, %8, %8\n" \
	             ins " %4, %4, %8, %8\n" ins " %5, %5, %8, %8\n" ins " %6, %6, %8, %8\n" ins " %7, %7, %8, %8\n" \
	             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
// dst <- op(src) with a modifier string (DPP)
#define REP16_DPP(mod) \
	asm volatile("v_mov_b32_dpp %0, %1 " mod "\nv_mov_b32_dpp %1, %2 " mod "\nv_mov_b32_dpp %2, %3 " mod "\nv_mov_b32_dpp %3, %4 " mod "\n" \
	             "v_mov_b32_dpp %4, %5 " mod "\nv_mov_b32_dpp %5, %6 " mod "\nv_mov_b32_dpp %6, %7 " mod "\nv_mov_b32_dpp %7, %0 " mod "\n" \
	             "v_mov_b32_dpp %0, %1 " mod "\nv_mov_b32_dpp %1, %2 " mod "\nv_mov_b32_dpp %2, %3 " mod "\nv_mov_b32_dpp %3, %4 " mod "\n" \
	             "v_mov_b32_dpp %4, %5 " mod "\nv_mov_b32_dpp %5, %6 " mod "\nv_mov_b32_dpp %6, %7 " mod "\nv_mov_b32_dpp %7, %0 " mod "\n" \
	             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))

enum Kind { K_ADD_U32, K_ADD_U32_E64, K_MAX_I32, K_ADD_U16, K_MAX_I16, K_ADD3, K_AND_OR, K_CNDMASK, K_CMP, K_SDWA, K_ALT_VOP2_PK, K_PK_ADD, K_PK_SUB, K_PK_MAX, K_PK_MINU, K_PK_MUL, K_PK_MAD, K_PK_SHR, K_BFI, K_XOR, K_PERM, K_DPP_WAVE_SHR, K_DPP_ROW_SHR,
            K_READLANE, K_READLANE_DPP_PAIR, K_LDS_U8, K_LDS_B32, K_ST_BYTE, K_ST_DWORD, K_ST_DWORDX4, K_MIX_DP, K_MIX_NOP, K_MIX_SALU, K_N };
static const char *kNames[K_N] = { "v_add_u32", "v_add_u32_e64 (VOP3 encoding)", "v_max_i32", "v_add_u16", "v_max_i16", "v_add3_u32", "v_and_or_b32", "v_cndmask_b32 (vcc)", "v_cmp_lt_i32 (-> vcc)", "v_add_u32_sdwa", "alternating v_add_u32 / v_pk_add_u16", "v_pk_add_u16", "v_pk_sub_u16", "v_pk_max_i16", "v_pk_min_u16", "v_pk_mul_lo_u16", "v_pk_mad_u16",
	"v_pk_lshrrev_b16", "v_bfi_b32", "v_xor_b32", "v_perm_b32", "v_mov_b32_dpp wave_shr:1", "v_mov_b32_dpp row_shr:1", "v_readlane_b32",
	"v_readlane_b32 + v_mov_dpp wave_shr (carry idiom)", "ds_read_u8", "ds_read_b32", "global_store_byte (64 B / wave-instr)",
	"global_store_dword (256 B / wave-instr)", "global_store_dwordx4 (1 KiB / wave-instr)", "DP cell body of ksw_fast (52 pk ops)", "DP cell body + s_nop 0 after every 2-operand op", "DP cell body + 2 SALU after every 2-operand op" };

template <int KIND>
__global__ void __launch_bounds__(256, 2) bench_kernel(uint32_t *out, unsigned long long *cyc, uint8_t *scratch, int iters)
{
	__shared__ uint32_t lds[4096]; // 16 KiB: eight blocks per CU fit
	const int tid = blockIdx.x * 256 + threadIdx.x;
	for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;
	__syncthreads();
	uint32_t a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3, a4 = tid + 4, a5 = tid + 5, a6 = tid + 6, a7 = tid + 7, b = 0x00010003u;
	uint8_t *wbase = scratch + (size_t)(tid >> 6) * 65536; // 64 KiB of scratch per wave, rewritten over and over (stays in L2)
	const unsigned long long w0 = wall_clock64();
	const unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
		if (KIND == K_ADD_U32) REP16_2OP("v_add_u32");
		else if (KIND == K_ADD_U32_E64) REP16_2OP("v_add_u32_e64");
		else if (KIND == K_MAX_I32) REP16_2OP("v_max_i32");
		else if (KIND == K_ADD_U16) REP16_2OP("v_add_u16");
		else if (KIND == K_MAX_I16) REP16_2OP("v_max_i16");
		else if (KIND == K_ADD3) REP16_3OP("v_add3_u32");
		else if (KIND == K_AND_OR) REP16_3OP("v_and_or_b32");
		else if (KIND == K_CNDMASK) {
			asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
			             "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
			             "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
			             "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
		} else if (KIND == K_CMP) {
			asm volatile("v_cmp_lt_i32 vcc, %0, %8\nv_cmp_lt_i32 vcc, %1, %8\nv_cmp_lt_i32 vcc, %2, %8\nv_cmp_lt_i32 vcc, %3, %8\n"
			             "v_cmp_lt_i32 vcc, %4, %8\nv_cmp_lt_i32 vcc, %5, %8\nv_cmp_lt_i32 vcc, %6, %8\nv_cmp_lt_i32 vcc, %7, %8\n"
			             "v_cmp_lt_i32 vcc, %0, %8\nv_cmp_lt_i32 vcc, %1, %8\nv_cmp_lt_i32 vcc, %2, %8\nv_cmp_lt_i32 vcc, %3, %8\n"
			             "v_cmp_lt_i32 vcc, %4, %8\nv_cmp_lt_i32 vcc, %5, %8\nv_cmp_lt_i32 vcc, %6, %8\nv_cmp_lt_i32 vcc, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
		} else if (KIND == K_SDWA) {
#define SD(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
			asm volatile(SD("%0") SD("%1") SD("%2") SD("%3") SD("%4") SD("%5") SD("%6") SD("%7") SD("%0") SD("%1") SD("%2") SD("%3") SD("%4") SD("%5") SD("%6") SD("%7")
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#undef SD
		} else if (KIND == K_ALT_VOP2_PK) {
			asm volatile("v_add_u32 %0, %0, %8\nv_pk_add_u16 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_pk_add_u16 %3, %3, %8\n"
			             "v_add_u32 %4, %4, %8\nv_pk_add_u16 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_pk_add_u16 %7, %7, %8\n"
			             "v_add_u32 %0, %0, %8\nv_pk_add_u16 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_pk_add_u16 %3, %3, %8\n"
			             "v_add_u32 %4, %4, %8\nv_pk_add_u16 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_pk_add_u16 %7, %7, %8\n"
			             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
		}
		else if (KIND == K_PK_ADD) REP16_2OP("v_pk_add_u16");
		else if (KIND == K_PK_SUB) REP16_2OP("v_pk_sub_u16");
		else if (KIND == K_PK_MAX) REP16_2OP("v_pk_max_i16");
		else if (KIND == K_PK_MINU) REP16_2OP("v_pk_min_u16");
		else if (KIND == K_PK_MUL) REP16_2OP("v_pk_mul_lo_u16");
		else if (KIND == K_PK_MAD) REP16_3OP("v_pk_mad_u16");
		else if (KIND == K_PK_SHR) REP16_2OP("v_pk_lshrrev_b16");
		else if (KIND == K_BFI) REP16_3OP("v_bfi_b32");
		else if (KIND == K_XOR) REP16_2OP("v_xor_b32");
		else if (KIND == K_PERM) REP16_3OP("v_perm_b32");
		else if (KIND == K_DPP_WAVE_SHR) REP16_DPP("wave_shr:1 row_mask:0xf bank_mask:0xf");
		else if (KIND == K_DPP_ROW_SHR) REP16_DPP("row_shr:1 row_mask:0xf bank_mask:0xf");
		else if (KIND == K_READLANE) {
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				uint32_t s0, s1, s2, s3, s4, s5, s6, s7;
				asm volatile("v_readlane_b32 %0, %8, 63\nv_readlane_b32 %1, %9, 63\nv_readlane_b32 %2, %10, 63\nv_readlane_b32 %3, %11, 63\n"
				             "v_readlane_b32 %4, %12, 63\nv_readlane_b32 %5, %13, 63\nv_readlane_b32 %6, %14, 63\nv_readlane_b32 %7, %15, 63\n"
				             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5), "=s"(s6), "=s"(s7)
				             : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
				b += s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7; // scalar
			}
		} else if (KIND == K_READLANE_DPP_PAIR) { // what one carried state costs per register set: readlane of the previous set's lane 63 + DPP shift with it as lane 0's value
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				const uint32_t c0 = __builtin_amdgcn_readlane(a0, 63), c1 = __builtin_amdgcn_readlane(a2, 63);
				a1 = (uint32_t)__builtin_amdgcn_update_dpp((int)c0, (int)a1, 0x138, 0xf, 0xf, false);
				a3 = (uint32_t)__builtin_amdgcn_update_dpp((int)c1, (int)a3, 0x138, 0xf, 0xf, false);
				a0 ^= a1, a2 ^= a3; // keep a dependence (2 more VALU: subtracted in the report? no -- reported as is: 2 readlane + 2 dpp + 2 xor)
			}
		} else if (KIND == K_LDS_U8) {
#pragma unroll
			for (int k = 0; k < 16; ++k) a0 += ((volatile uint8_t *)lds)[(a1 + k * 67 + threadIdx.x) & 16383];
		} else if (KIND == K_LDS_B32) {
#pragma unroll
			for (int k = 0; k < 16; ++k) a0 += ((volatile uint32_t *)lds)[(a1 + k * 67 + threadIdx.x) & 4095];
		} else if (KIND == K_ST_BYTE) {
#pragma unroll
			for (int k = 0; k < 16; ++k) ((volatile uint8_t *)wbase)[((it * 16 + k) * 64 & 65535) + (threadIdx.x & 63)] = (uint8_t)a0;
		} else if (KIND == K_ST_DWORD) {
#pragma unroll
			for (int k = 0; k < 16; ++k) ((volatile uint32_t *)wbase)[((it * 16 + k) * 64 & 16383) + (threadIdx.x & 63)] = a0;
		} else if (KIND == K_ST_DWORDX4) {
#pragma unroll
			for (int k = 0; k < 16; ++k) {
				typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
				u32x4 v = { a0, a1, a2, a3 };
				__builtin_nontemporal_store(v, (u32x4 *)wbase + (((it * 16 + k) * 64 & 4095) + (threadIdx.x & 63)));
			}
		} else if (KIND == K_MIX_DP || KIND == K_MIX_NOP || KIND == K_MIX_SALU) { // the arithmetic of one DP cell pair (ksw_fast.hip row body without operand fetch / stores): 52 packed ops
#pragma unroll
			for (int k = 0; k < 2; ++k) {
				uint32_t sacc = b;
				uint32_t z, a, bb, a2_, b2_, z1, z2, z3, z4, d, tmp;
				const uint32_t ONE = 0x00010001u;
#define XTRA (KIND == K_MIX_NOP ? "\n\ts_nop 0" : KIND == K_MIX_SALU ? "\n\ts_add_u32 %3, %3, 1\n\ts_nop 0" : "")
#define P2(r, ins, x, y) do { if (KIND == K_MIX_DP) asm volatile(ins " %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); else if (KIND == K_MIX_NOP) asm volatile(ins " %0, %1, %2\n\ts_nop 0" : "=v"(r) : "v"(x), "v"(y)); else asm volatile(ins " %0, %1, %2\n\ts_add_u32 %3, %3, 1\n\ts_and_b32 %3, %3, 0xffff" : "=v"(r) : "v"(x), "v"(y), "s"(sacc)); } while (0)
#define P3(r, ins, x, y, w) asm volatile(ins " %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(w))
				uint32_t tq; P2(tq, "v_xor_b32", a0, a1);
				P2(z, "v_pk_min_u16", tq, ONE); P3(z, "v_pk_mad_u16", z, b, b);
				uint32_t oq; P2(oq, "v_or_b32", a0, a1); P2(oq, "v_pk_lshrrev_b16", ONE, oq); P2(tmp, "v_pk_sub_u16", b, z); P3(z, "v_pk_mad_u16", oq, tmp, z);
				P2(a, "v_pk_add_u16", a2, a3); P2(bb, "v_pk_add_u16", a4, a5); P2(a2_, "v_pk_add_u16", a6, a3); P2(b2_, "v_pk_add_u16", a7, a5);
				P2(z1, "v_pk_max_i16", z, a); P2(z2, "v_pk_max_i16", z1, bb); P2(z3, "v_pk_max_i16", z2, a2_); P2(z4, "v_pk_max_i16", z3, b2_);
				uint32_t n0, n1, n2, n3;
				P2(n0, "v_pk_sub_u16", z4, z); P2(n0, "v_pk_min_u16", n0, ONE); P2(n1, "v_pk_sub_u16", z4, a); P2(n1, "v_pk_min_u16", n1, ONE);
				P2(n2, "v_pk_sub_u16", z4, bb); P2(n2, "v_pk_min_u16", n2, ONE); P2(n3, "v_pk_sub_u16", z4, a2_); P2(n3, "v_pk_min_u16", n3, ONE);
				P2(d, "v_pk_add_u16", n3, ONE); P3(d, "v_pk_mad_u16", n2, d, ONE); P3(d, "v_pk_mad_u16", n1, d, ONE); P2(d, "v_pk_mul_lo_u16", n0, d);
				P2(z, "v_pk_min_i16", z4, b);
				P2(a1, "v_pk_sub_u16", z, a3); P2(a3, "v_pk_sub_u16", z, a5);
				P2(tmp, "v_pk_sub_u16", z, b); P2(a, "v_pk_sub_u16", a, tmp); P2(bb, "v_pk_sub_u16", bb, tmp);
				P2(tmp, "v_pk_sub_u16", z, ONE); P2(a2_, "v_pk_sub_u16", a2_, tmp); P2(b2_, "v_pk_sub_u16", b2_, tmp);
				uint32_t m0, m1, m2, m3, f;
				P2(m0, "v_pk_max_i16", a, b); P2(m1, "v_pk_max_i16", bb, b); P2(m2, "v_pk_max_i16", a2_, b); P2(m3, "v_pk_max_i16", b2_, b);
				P2(f, "v_pk_min_u16", m0, ONE); P3(d, "v_pk_mad_u16", f, b, d); P2(f, "v_pk_min_u16", m1, ONE); P3(d, "v_pk_mad_u16", f, b, d);
				P2(f, "v_pk_min_u16", m2, ONE); P3(d, "v_pk_mad_u16", f, b, d); P2(f, "v_pk_min_u16", m3, ONE); P3(d, "v_pk_mad_u16", f, b, d);
				P2(a2, "v_pk_sub_u16", m0, b); P2(a4, "v_pk_sub_u16", m1, b); P2(a6, "v_pk_sub_u16", m2, b); P2(a7, "v_pk_sub_u16", m3, b);
				a0 ^= d;
#undef XTRA
#undef P2
#undef P3
			}
		}
	}
	__builtin_amdgcn_s_waitcnt(0);
	const unsigned long long t1 = __builtin_amdgcn_s_memtime();
	out[tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b;
	const unsigned long long w1 = wall_clock64();
	// where the wave ran: HW_ID (gfx9: wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13) and the XCD it is on
	uint32_t hw_id, xcc_id;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
	if ((threadIdx.x & 63) == 0) {
		unsigned long long *rec = cyc + (size_t)(tid >> 6) * 5;
		rec[0] = t0, rec[1] = t1, rec[2] = w0, rec[3] = w1, rec[4] = (unsigned long long)(xcc_id & 0xf) << 32 | (hw_id & 0xffffu);
	}
}

static double g_wall_hz = 1e8;

// Per-SIMD accounting (no assumption about how many waves share a SIMD): every wave records its s_memtime interval, its wall_clock64
// interval and where it ran (HW_ID + XCC_ID).  For each SIMD that ran waves: window = latest end - earliest start of its waves (s_memtime
// ticks of that XCD), issued = waves x instructions per wave, co-residency = sum of the waves' durations / window.  Reported: the MEDIAN
// SIMD's ticks per wave-instruction, the median measured co-residency, and the shader clock the ticks ran at (ticks / wall ns).
struct SimdStat { double tk_per_instr, ns_per_instr, resid, ghz; int simds; double waves_min, waves_med, waves_max; };

static SimdStat per_simd(const std::vector<unsigned long long> &rec, size_t n_waves, double instr_per_wave)
{
	struct Acc { unsigned long long t_lo = ~0ull, t_hi = 0, w_lo = ~0ull, w_hi = 0; double dur = 0; int n = 0; };
	std::map<unsigned long long, Acc> simd;
	for (size_t w = 0; w < n_waves; ++w) {
		const unsigned long long *r = &rec[w * 5];
		const unsigned long long key = (r[4] >> 32) << 16 | (r[4] & 0xfff0ull); // XCD | SE, SH, CU, pipe, SIMD (the wave slot, bits 3:0, dropped)
		Acc &a = simd[key];
		a.t_lo = std::min(a.t_lo, r[0]), a.t_hi = std::max(a.t_hi, r[1]), a.w_lo = std::min(a.w_lo, r[2]), a.w_hi = std::max(a.w_hi, r[3]);
		a.dur += (double)(r[1] - r[0]), ++a.n;
	}
	std::vector<double> tk, ns, resid, ghz, nw;
	for (auto &kv : simd) {
		const Acc &a = kv.second;
		const double win = (double)(a.t_hi - a.t_lo), wall_ns = (double)(a.w_hi - a.w_lo) / g_wall_hz * 1e9;
		tk.push_back(win / (a.n * instr_per_wave)), ns.push_back(wall_ns / (a.n * instr_per_wave)), resid.push_back(a.dur / win), ghz.push_back(win / wall_ns), nw.push_back(a.n);
	}
	auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
	SimdStat st;
	st.tk_per_instr = med(tk), st.ns_per_instr = med(ns), st.resid = med(resid), st.ghz = med(ghz), st.simds = (int)simd.size();
	std::sort(nw.begin(), nw.end());
	st.waves_min = nw.front(), st.waves_med = nw[nw.size() / 2], st.waves_max = nw.back();
	return st;
}

template <int KIND>
static void run(int n_cu, uint32_t *d_out, unsigned long long *d_cyc, uint8_t *d_scratch, double per_iter, FILE *fp)
{
	const int wps[5] = { 1, 2, 4, 8, 16 }; // blocks per CU launched; 16 = more than can be resident: the queued ("saturated") case
	const bool slow = KIND == K_ST_BYTE || KIND == K_ST_DWORD || KIND == K_ST_DWORDX4 || KIND == K_LDS_U8 || KIND == K_LDS_B32 || KIND == K_MIX_DP || KIND == K_MIX_NOP || KIND == K_MIX_SALU || KIND == K_READLANE_DPP_PAIR;
	const int iters = slow ? ITERS / 16 : ITERS;
	fprintf(fp, "%-50s", kNames[KIND]);
	for (int wi = 0; wi < 5; ++wi) {
		const int blocks = n_cu * wps[wi];
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL((bench_kernel<KIND>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_scratch, iters / 8); // warm-up
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL((bench_kernel<KIND>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, d_scratch, iters);
		CHECK(hipEventRecord(e1));
		CHECK(hipDeviceSynchronize());
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		std::vector<unsigned long long> c((size_t)blocks * 4 * 5);
		CHECK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
		const SimdStat st = per_simd(c, (size_t)blocks * 4, (double)iters * per_iter);
		const double ev_ns = ms * 1e6 / ((double)blocks * 4 / (n_cu * 4.0) * (double)iters * per_iter); // event wall time x SIMDs / total wave-instructions
		fprintf(fp, " | B=%-2d %6.3f tk %6.3f ns (event %6.3f ns) resid %4.1f waves/SIMD %g/%g/%g @%.2f GHz", wps[wi], st.tk_per_instr, st.ns_per_instr, ev_ns, st.resid,
		        st.waves_min, st.waves_med, st.waves_max, st.ghz);
		CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
	}
	fprintf(fp, "\n");
	fflush(fp);
}

int main()
{
	hipDeviceProp_t p;
	CHECK(hipGetDeviceProperties(&p, 0));
	const int n_cu = p.multiProcessorCount;
	printf("device %s, %d CUs, clockRate %d kHz; ITERS %d; blocks of 256 threads (one wave per SIMD), W blocks per CU\n", p.gcnArchName, n_cu, p.clockRate, ITERS);
	uint32_t *d_out; unsigned long long *d_cyc; uint8_t *d_scratch;
	const size_t n_waves = (size_t)n_cu * 4 * 16;
	CHECK(hipMalloc(&d_out, n_waves * 64 * 4)); CHECK(hipMalloc(&d_cyc, n_waves * 5 * 8)); CHECK(hipMalloc(&d_scratch, n_waves * 65536));
	FILE *fp = stdout;
	int wall_khz = 100000;
	if (hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0) == hipSuccess && wall_khz > 0) g_wall_hz = wall_khz * 1e3;
	printf("wall_clock64 rate %d kHz.  Columns: B blocks of 256 threads launched per CU (B=16 exceeds what can be resident: queued) -> s_memtime ticks and wall ns per wave64 instruction PER SIMD\n"
	       "(median SIMD; window of the SIMD's waves / instructions they issued), the same from the HIP-event time of the launch, measured co-residency (sum of wave durations / window), waves that\n"
	       "ran on a SIMD (min/median/max over SIMDs, from HW_ID + XCC_ID), shader clock during the launch (s_memtime ticks per wall ns)\n", wall_khz);
	for (int k = 0; k < 40; ++k) hipLaunchKernelGGL((bench_kernel<K_PK_ADD>), dim3(n_cu * 4), dim3(256), 0, 0, d_out, d_cyc, d_scratch, ITERS); // clocks up
	CHECK(hipDeviceSynchronize());
	run<K_ADD_U32>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_ADD_U32_E64>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_MAX_I32>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_ADD_U16>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_MAX_I16>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_XOR>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_CNDMASK>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_CMP>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_SDWA>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_ADD3>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_AND_OR>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_BFI>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PERM>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PK_ADD>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PK_SUB>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PK_MAX>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PK_MINU>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PK_MUL>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PK_MAD>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_PK_SHR>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_ALT_VOP2_PK>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_DPP_WAVE_SHR>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_DPP_ROW_SHR>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_READLANE>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_READLANE_DPP_PAIR>(n_cu, d_out, d_cyc, d_scratch, 8, fp); // per (2 readlane + 2 dpp + 2 xor) group
	run<K_LDS_U8>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_LDS_B32>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_ST_BYTE>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_ST_DWORD>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_ST_DWORDX4>(n_cu, d_out, d_cyc, d_scratch, 16, fp);
	run<K_MIX_DP>(n_cu, d_out, d_cyc, d_scratch, 2, fp); // per cell-pair body (52 packed ops + 3 plain)
	run<K_MIX_NOP>(n_cu, d_out, d_cyc, d_scratch, 2, fp);
	return 0;
}

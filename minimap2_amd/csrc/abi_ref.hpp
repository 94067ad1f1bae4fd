// Layout-compatible views of the reference's public structs (lh3/minimap2 v2.30).  The drop-in boundary
// exchanges these by pointer with host code compiled against the reference's own minimap.h, so the field
// order, widths and bit-field packing below must match it exactly (x86-64 SysV ABI).  Citations are to
// /root/reference; tests/test_capi.py::test_struct_layouts_match_reference_headers compares sizeof and every offsetof (the ctypes
// mirrors and these C++ ones) against a probe compiled with the real headers.
#pragma once
#include <cstdint>
#include <cstddef>

namespace mm2amd {
namespace ref {

struct mm128 { uint64_t x, y; };                       // minimap.h:77

struct IdxSeq {                                        // mm_idx_seq_t, minimap.h:81-86
	char *name;
	uint64_t offset;
	uint32_t len;
	uint32_t is_alt;
};

struct KhashIdx {                                      // khash_t(idx): khash.h:182-188 with index.c:19-22
	uint32_t n_buckets, size, n_occupied, upper_bound;
	uint32_t *flags;                                   // 2 bits per slot: bit1 = empty, bit0 = deleted
	uint64_t *keys;                                    // (minimizer >> b) << 1 | is_singleton
	uint64_t *vals;                                    // singleton: the position; else start<<32 | n into Bucket::p
};

struct Bucket {                                        // mm_idx_bucket_t, index.c:28-33
	struct { size_t n, m; mm128 *a; } a;
	int32_t n;
	uint64_t *p;
	KhashIdx *h;
};

struct Idx {                                           // mm_idx_t, minimap.h:88-100
	int32_t b, w, k, flag;
	uint32_t n_seq;
	int32_t index;
	int32_t n_alt;
	IdxSeq *seq;
	uint32_t *S;
	Bucket *B;
	void *I, *spsc, *J;
	void *km, *h;
};

struct Intv1 { int32_t st, en, cnt; int32_t score : 30, strand : 2; };   // mm_idx_intv1_t, index.c:35-38
struct IntvList { int32_t n, m; Intv1 *a; };                              // mm_idx_intv_t, index.c:40-43 (mm_idx_t::I: one per sequence)

struct JJump1 { int32_t off, off2, cnt; int16_t strand; uint16_t flag; };  // mm_idx_jjump1_t, mmpriv.h:58-62
struct JJumpList { int32_t n, m; JJump1 *a; };                            // mm_idx_jjump_t, index.c:45-48 (mm_idx_t::J: one per sequence)
struct SpscList { uint32_t n, m; uint64_t *a; };                          // mm_idx_spsc_t, index.c:965-968 (mm_idx_t::spsc: two per sequence, + then - strand)
constexpr uint16_t JUNC_ANNO = 0x1;                                       // MM_JUNC_ANNO, mmpriv.h:27

struct Extra {                                         // mm_extra_t, minimap.h:103-110
	uint32_t capacity;
	int32_t dp_score, dp_max, dp_max2;
	int32_t dp_max0;
	uint32_t n_ambi : 30, trans_strand : 2;
	uint32_t n_cigar;
	uint32_t cigar[];
};

struct Reg1 {                                          // mm_reg1_t, minimap.h:112-127
	int32_t id;
	int32_t cnt;
	int32_t rid;
	int32_t score;
	int32_t qs, qe, rs, re;
	int32_t parent, subsc;
	int32_t as;
	int32_t mlen, blen;
	int32_t n_sub;
	int32_t score0;
	uint32_t mapq : 8, split : 2, rev : 1, inv : 1, sam_pri : 1, proper_frag : 1, pe_thru : 1, seg_split : 1, seg_id : 8,
	         split_inv : 1, is_alt : 1, strand_retained : 1, is_spliced : 1, dummy : 4;
	uint32_t hash;
	float div;
	Extra *p;
};

struct MapOpt {                                        // mm_mapopt_t, minimap.h:136-192
	int64_t flag;
	int seed;
	int sdust_thres;
	int max_qlen;
	int bw, bw_long;
	int max_gap, max_gap_ref;
	int max_frag_len;
	int max_chain_skip, max_chain_iter;
	int min_cnt;
	int min_chain_score;
	float chain_gap_scale;
	float chain_skip_scale;
	int rmq_size_cap, rmq_inner_dist;
	int rmq_rescue_size;
	float rmq_rescue_ratio;
	float mask_level;
	int mask_len;
	float pri_ratio;
	int best_n;
	float alt_drop;
	int a, b, q, e, q2, e2;
	int transition;
	int sc_ambi;
	int noncan;
	int junc_bonus;
	int junc_pen;
	int zdrop, zdrop_inv;
	int end_bonus;
	int min_dp_max;
	int min_ksw_len;
	int anchor_ext_len, anchor_ext_shift;
	float max_clip_ratio;
	int rank_min_len;
	float rank_frac;
	int pe_ori, pe_bonus;
	int32_t jump_min_match;
	float mid_occ_frac;
	float q_occ_frac;
	int32_t min_mid_occ, max_mid_occ;
	int32_t mid_occ;
	int32_t max_occ, max_max_occ, occ_dist;
	int64_t mini_batch_size;
	int64_t max_sw_mat;
	int64_t cap_kalloc;
	const char *split_prefix;
};

struct IdxOpt {                                        // mm_idxopt_t, minimap.h:130-134
	short k, w, flag, bucket_bits;
	int64_t mini_batch_size;
	uint64_t batch_size;
};

struct Bseq1 {                                         // mm_bseq1_t, bseq.h:14-17
	int l_seq, rid;
	char *name, *seq, *qual, *comment;
};

static_assert(sizeof(Reg1) == 80, "mm_reg1_t is 80 bytes");
static_assert(sizeof(Extra) == 28, "mm_extra_t header is 28 bytes");
static_assert(sizeof(IdxSeq) == 24, "mm_idx_seq_t is 24 bytes");
static_assert(sizeof(Bseq1) == 40, "mm_bseq1_t is 40 bytes");
static_assert(sizeof(IdxOpt) == 24, "mm_idxopt_t is 24 bytes");
static_assert(sizeof(MapOpt) == 264, "mm_mapopt_t is 264 bytes");

// option flags, minimap.h:10-50
constexpr int64_t F_NO_DIAG = 0x001, F_NO_DUAL = 0x002, F_CIGAR = 0x004, F_OUT_SAM = 0x008, F_NO_PRINT_2ND = 0x4000, F_2_IO_THREADS = 0x8000, F_SPLICE = 0x080,
	F_SPLICE_FOR = 0x100, F_SPLICE_REV = 0x200, F_NO_LJOIN = 0x400, F_SR = 0x1000, F_FRAG_MODE = 0x2000,
	F_INDEPEND_SEG = 0x20000, F_SPLICE_FLANK = 0x40000, F_FOR_ONLY = 0x100000, F_REV_ONLY = 0x200000,
	F_HEAP_SORT = 0x400000, F_ALL_CHAINS = 0x800000, F_EQX = 0x4000000, F_NO_END_FLT = 0x10000000,
	F_HARD_MLEVEL = 0x20000000, F_RMQ = 0x80000000LL, F_QSTRAND = 0x100000000LL, F_NO_INV = 0x200000000LL,
	F_NO_HASH_NAME = 0x400000000LL, F_SPLICE_OLD = 0x800000000LL, F_WEAK_PAIRING = 0x4000000000LL, F_SR_RNA = 0x8000000000LL;
constexpr int I_HPC = 0x1, I_NO_SEQ = 0x2, I_NO_NAME = 0x4;

// anchor y-field flags, mmpriv.h:19-25
constexpr uint64_t SEED_LONG_JOIN = 1ULL << 40, SEED_IGNORE = 1ULL << 41, SEED_TANDEM = 1ULL << 42, SEED_SELF = 1ULL << 43;
constexpr int SEED_SEG_SHIFT = 48;
constexpr uint64_t SEED_SEG_MASK = 0xffULL << SEED_SEG_SHIFT;
constexpr int PARENT_UNSET = -1, PARENT_TMP_PRI = -2;  // mmpriv.h:9-10

} // namespace ref
} // namespace mm2amd

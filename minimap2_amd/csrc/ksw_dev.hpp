// Device-side job/result records of the batched extension-DP kernels (host <-> device contract).
#pragma once
#include <cstdint>
#include <cstddef>
#ifndef __HIPCC__
#define __host__
#define __device__
#endif

namespace mm2amd {

// flag bits 0..12 are the reference's KSW_EZ_* (ksw2.h:8-19); the bits below are ours
enum : int32_t {
	KSW_SCORE_ONLY = 0x01, KSW_RIGHT = 0x02, KSW_GENERIC_SC = 0x04, KSW_APPROX_MAX = 0x08, KSW_APPROX_DROP = 0x10,
	KSW_EXTZ_ONLY = 0x40, KSW_REV_CIGAR = 0x80,
	KSW_SPLICE_FOR = 0x100, KSW_SPLICE_REV = 0x200, KSW_SPLICE_FLANK = 0x400, KSW_SPLICE_CMPLX = 0x800, KSW_SPLICE_SCORE = 0x1000,
	KSWJ_Q_REVERSED = 1 << 16,  // query bytes are read backwards from q_off (left extension, align.c:787)
	KSWJ_T_PACKED   = 1 << 17,  // target comes from the 4-bit packed reference (mi->S); t_off is a base index
	KSWJ_T_REVERSED = 1 << 18,  // target is read backwards (left extension, align.c:788)
	KSWJ_SKIP       = 1 << 19,  // max_sw_mat guard hit (align.c:349-351): reset + zdropped=1, no DP
};
constexpr int32_t KSW_NEG_INF = -0x40000000;

struct KswJob {             // 48 B
	uint64_t q_off;         // byte offset into the query pool (nt4 codes, 1 B/base); first base read (last if Q_REVERSED)
	uint64_t t_off;         // byte offset into the target pool, or base index into packed S; first base read (last if T_REVERSED)
	int32_t qlen, tlen;
	int32_t w, zdrop, end_bonus, flag;
	uint32_t tag;           // free for the host (e.g. owner of the job)
	uint32_t reserved;
};

struct KswRes {             // 68 B; the first 12 fields have the meaning of ksw_extz_t (ksw2.h:34-43)
	int32_t max, zdropped;
	int32_t max_q, max_t;
	int32_t mqe, mqe_t;
	int32_t mte, mte_q;
	int32_t score;
	int32_t n_cigar;
	int32_t reach_end;
	uint32_t cigar_off;     // where the CIGAR was placed in the cigar pool (allocated by the kernel)
	// mm_test_zdrop's scan of the finished alignment (align.c:61-84), done by the kernel that produced it: the largest
	// diagonal-adjusted score drop and where it happened (target/query offsets inside the window).  zd_max == KSW_ZD_NONE: not
	// computed (the host scans the CIGAR itself).
	int32_t zd_max, zd_t0, zd_t1, zd_q0, zd_q1;
};
constexpr int32_t KSW_ZD_NONE = INT32_MIN;

struct KswScoring {         // uniform over a launch
	int8_t mat[25];
	int8_t m;
	int8_t q, e, q2, e2;    // as passed by the caller, BEFORE the swap at ksw2_extd2_sse.c:78
	int8_t single;          // 1: single-affine recurrences (ksw_extz2_sse; q2/e2 unused), the rule of mm_align_pair (align.c:353-356); 2: splice (ksw_exts2_sse)
	int8_t noncan;          // splice mode: cost of a non-canonical splice site (ksw2_exts2_sse.c:33, opt->noncan); gapo2 = q2, e2 unused
	// splice mode with junction annotation (ksw2_exts2_sse.c:201-217, the junc[] of mm_idx_bed_junc): a job with reserved > 0 owns
	// juncs[tag .. tag + reserved), entries  t << 4 | bits  ascending in t, t = offset of the base in the window as stored in the
	// index (NOT reversed for a T_REVERSED job), bits as in junc[]: 1 / 8 first base of a + / - strand intron, 2 / 4 its last base
	// Jobs with KSW_SPLICE_SCORE (--spsc, ksw2_exts2_sse.c:196-200) use the same pool with entries  t << 8 | score byte  (the junc[]
	// of mm_idx_spsc_get: (score + 64) << 1 | is_acceptor); positions without an entry cost junc_pen.
	int8_t junc_bonus = 0, junc_pen = 0;
	const uint32_t *juncs = nullptr;
	size_t n_juncs = 0;
	// targets that are not windows of the reference (jobs without KSWJ_T_PACKED): t_off indexes this pool of nt4 bytes
	const uint8_t *tbytes = nullptr;
	size_t n_tbytes = 0;
};

struct KswLaunch {
	const KswJob *jobs;     // device, sorted by decreasing cost
	KswRes *res;            // device
	int32_t n_jobs;
	const uint8_t *qpool;   // device nt4 bytes
	const uint8_t *tpool;   // device nt4 bytes (may be null when every job is T_PACKED)
	const uint32_t *S;      // device 4-bit packed reference (may be null)
	uint32_t *cigar_pool;   // device; CIGARs are packed back to back, space handed out by an atomic cursor
	uint32_t cigar_pool_cap;
	uint32_t *cigar_cursor; // device: [0] = next free entry, [1] = set to 1 when the pool overflowed
	uint32_t *cigar_tmp;    // device scratch: n_slots * cigar_tmp_cap (traceback output before it is packed)
	uint32_t cigar_tmp_cap;
	uint8_t *dir_pool;      // device scratch for direction matrices: n_slots * slot_bytes
	size_t slot_bytes;
	int32_t *counter;       // device, zeroed before launch: persistent-wave job queue head
	int32_t ring, max_Q16;    // LDS sizing: slots in the per-position state rings (a power of two), largest 16-rounded qlen in the launch
	bool single_affine = false;    // ksw_extz2 recurrences (q2/e2 ignored) instead of ksw_extd2
	bool splice = false;           // ksw_exts2 recurrences (no band, intron state, N operations)
	uint8_t *state_pool = nullptr; // when set: per-slot state slabs in HBM (ksw_lds_per_wave bytes each) instead of LDS
	// Launches fed by a list made on the device (round 6; ksw_band.hip, ksw_stream.hip): position k of the queue is job list[k] of `jobs` / `res`, and
	// *n_list (read by the kernel) says how many there are -- the banded kernel's rejects, re-run without a host round trip.
	const uint32_t *list = nullptr;
	const int32_t *n_list = nullptr;
	// The banded kernel's ways out for a window whose band it could not prove sufficient: a wider band (widen_W diagonals; widen_slots = the base slots a wave of
	// the launch that takes the list holds per sequence) or the full rectangle -- retry_list for windows of at most retry_max x retry_max (the streaming kernel's
	// largest class), big_list for the others (the strip kernel).  Entries are list_base + the job's index in this launch, i.e. positions in the batch's launch order.
	uint32_t *widen_list = nullptr, *retry_list = nullptr, *big_list = nullptr;
	int32_t *widen_count = nullptr, *retry_count = nullptr, *big_count = nullptr;
	int32_t widen_W = 0, widen_slots = 0, retry_max = 512;
	uint32_t list_base = 0;
	unsigned long long *band_acc = nullptr; // [0] += score found + the corners' unavoidable gap, [1] += best possible score, over the windows a first attempt computed
	int32_t band_reject = 0;       // tests: 1 = no result of this launch is accepted, 2 = ... and none goes to the wider band
	KswScoring sc;
};

// LDS bytes one wave needs for a job class (A,B,H int32 rings + target ring + reversed query)
__host__ __device__ inline size_t ksw_lds_per_wave(int ring, int max_Q16) { return (size_t)13 * ring + max_Q16 + 16; }

// rows * bytes-per-row of the direction matrix of one job (ksw2_extd2_sse.c:94-95,122)
__host__ __device__ inline size_t ksw_dir_bytes(int qlen, int tlen, int w)
{
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	long n = qlen < tlen ? qlen : tlen;
	n = ((n < (long)w + 1 ? n : (long)w + 1) + 15) / 16 + 1;
	return (size_t)(qlen + tlen - 1) * (size_t)n * 16;
}

// host launcher (ksw_extd2.hip); n_slots persistent waves, waves_per_block in {1,4}
void ksw_extd2_launch(const KswLaunch &L, int n_slots, int waves_per_block, int team, void *stream /* hipStream_t */);

} // namespace mm2amd

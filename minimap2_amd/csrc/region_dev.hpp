// Chains -> hits -> DP windows -> consumed DP results, on the device (round 5).
//
// Between the chaining kernels and the DP kernels, and again between the DP kernels and region_finish_kernel, the reference runs scalar
// per-read code on the anchors and on the DP results (map.c:283-336: mm_gen_regs, mm_set_parent, mm_select_sub, mm_est_err; align.c:645-914:
// the seed clean-up, the extension limits, the window walk of mm_align1 and what it does with every ksw_extz_t).  Rounds 1-4 ran that on host
// threads: the chained anchors and every DP result crossed PCIe and were walked cache-cold, 1.7 host core-seconds per 1-Gbase step.  Here the
// same rules run where the data is:
//
//   chain_regs_kernel    one wavefront per read: the read's chains become hit records (sort by score and the read's hash, coordinates, fuzzy
//                        match / block lengths by a wave reduction over the chain's anchors), parent / secondary marking, secondary selection,
//                        the minimizer counts mm_est_err needs (a merge of the chain's anchors with the read's minimizer positions, done as
//                        one binary search per anchor), and the anchors of the hits that survive squeezed together (hit.c:322-340);
//   region_plan_kernel   one wavefront per region: seed clean-up (align.c:435-561), extension limits (:695-767), the window walk (:779-890);
//                        emits one KswJob per window, densely, in the order the reference would have called ksw2;
//   region_consume_kernel one region: walks its windows' KswRes in the reference's order (extension end points, the Z-drop test on the
//                        kernels' own scan, score bookkeeping, mm_extra_t's capacity rule) and writes the FinRegion / FinPiece records
//                        region_finish_kernel takes.
//
// What the device does NOT decide is handed back: a read any of whose regions needs a second DP round (a gap fill that trips the Z-drop test,
// a truncated alignment with its split, the inversion rescue), equal sort keys whose order the reference's unstable sort decides, more chains
// than the kernel keeps in LDS, a strand-retained secondary (its filter needs libm's pow).  Such a read is flagged and goes through the host
// path (hits.cpp / align.cpp) from its chains, exactly as before; every other read reaches the host as finished hit records.
#pragma once
#include <cstdint>
#include "types.hpp"
#include "ksw_dev.hpp"
#include "region_finish.hpp"

namespace mm2amd {

struct RgnOpts {              // uniform over a launch: what the per-read code reads from mm_mapopt_t / mm_idx_t
	int64_t flag;             // mm_mapopt_t::flag
	int64_t max_sw_mat;
	int k;                    // mm_idx_t::k
	float mask_level, pri_ratio;
	int mask_len, best_n, sub_diff, min_strand_sc; // sub_diff = a * 2 + b (map.c:209); min_strand_sc = (int)(max_gap * 0.8) (map.c:210)
	int max_gap, min_cnt, min_chain_score, bw;
	int bw_ext, bw_gap;       // (int)(bw * 1.5 + 1.), and the same of bw_long, at least bw_ext (align.c:676-678)
	int a, b, q, e, zdrop, zdrop_inv, end_bonus, min_ksw_len, transition;
	int hpc;                  // mm_idx_t::flag & MM_I_HPC: window boundaries sit at the start of a homopolymer run (mm_adjust_minier, align.c:418-428), minimizer spans vary
	int sc_ambi;              // short reads: the ungapped score of the one gap window (align.c:823-833)
	int8_t mat[25];           // ... and the scoring matrix its Z-drop test walks (mm_test_zdrop, align.c:59-84)
};

enum : uint32_t {             // RgnRead::src
	RGN_SRC_LJ = 1,           // the read's chains are the long-join re-chain's (second backtrack's arrays)
	RGN_SRC_SKIP = 2,         // the host keeps this read (not chained on the device, or still to be re-chained by the host's tree)
};
enum : uint32_t {             // RgnReadOut::flags: why a read goes back to the host path (0: it does not)
	RGN_F_SKIPPED = 1, RGN_F_MANY_CHAINS = 2, RGN_F_SORT_TIE = 4, RGN_F_STRAND_RETAINED = 8, RGN_F_MULTI_ROUND = 16, RGN_F_LONG_CIGAR = 32, RGN_F_NO_CIGAR = 64,
	RGN_F_SKIP_JOB = 128,
};

struct RgnRead {              // per read of the sub-batch, made by the host from what it knows anyway (offsets, lengths, the read's hash)
	uint64_t a_off, u_off;    // the read's chained anchors / chain records in the arrays `src` names
	uint64_t sq_off;          // where its squeezed anchors go (a slice as long as its chained anchors)
	uint64_t mp_off;          // its minimizer positions (seed.c:124)
	uint64_t qpool_fwd;       // its forward nt4 block in the query pool; the reverse complement follows at + qlen
	int32_t n_u, n_a, n_mp, qlen;
	uint32_t hash, src;       // map.c:246-248; RGN_SRC_*
	// a two-segment fragment (a read pair; round 6): the chains were found on the concatenation of the segments (qlen + qlen2 bases) and are cut per segment
	// (mm_seg_gen, hit.c:342-396); segment 1's query block follows segment 0's (at + 2 qlen), its squeezed anchors at sq_off + n_a.  qlen2 = 0: a single read
	int32_t qlen2, gap_ref;   // gap_ref: max_chain_gap_ref of the fragment (map.c:264-270), what mm_select_sub_multi calls "beside"
};

// One per read -- per SEGMENT when the launch holds fragments (RgnBuffers::rout_stride == 2: entry 2 * read + segment; a hand-back's flags in the read's first entry)
struct RgnReadOut { uint32_t reg0; int32_t n_regs, n_a_sq; uint32_t flags; float avg_k; uint32_t pad; }; // avg_k: the mean minimizer span mm_est_err divides by (esterr.c:37-40)

struct RgnAux { int32_t n_match, n_tot; };   // mm_est_err's counts (esterr.c:46-60); n_tot < 0: the hit keeps div = -1

struct RgnPlan {              // one per hit record (same index): what region_plan_kernel decided, what region_consume_kernel found
	uint32_t read, job0;      // the region's read; its first window = its first DP job
	int32_t n_win;
	int32_t as1, cnt1;        // the anchors the windows were cut from (after the end trimming)
	int32_t rid, rev;
	int32_t rs, qs;           // where the first gap window starts (align.c:800-801)
	int32_t has_left, has_right;
	int32_t dp_score;         // consume: mm_extra_t::dp_score
	uint32_t capacity;        // consume: what mm_extra_t::capacity would be had the windows' CIGARs been appended one by one (align.c:305-334)
	int32_t status;           // consume: 0 = finished by region_finish_kernel; > 0: RGN_F_* (the read goes back to the host)
	uint32_t piece0;          // where the region's pieces go (cursors[RGN_CUR_PIECES]): one per window, plus one for a short read's ungapped window
	int32_t seg;              // which segment of its fragment the region belongs to
	int32_t ug_len, ug_score; // short reads: the one gap window lies on one diagonal and its ungapped alignment beats any gapped one (align.c:823-833): no DP job, a single M
};

struct RgnWin { int32_t qs, qe, rs, re, anchor_i, kind; }; // one DP window = one job (same index); kind: WindowKind

enum { RGN_CUR_REGS = 0, RGN_CUR_JOBS = 1, RGN_CUR_PIECES = 2, RGN_CUR_OUT = 3, RGN_CUR_MAX_OPS = 4, RGN_CUR_N_FIN = 5, RGN_CUR_N = 8 };

struct RgnBuffers {           // device pointers of one sub-batch
	int n_reads;
	const RgnRead *reads;
	const Anchor *a_src[2];   // chained anchors: first backtrack / long-join backtrack
	const uint64_t *u_src[2]; // chain records (score << 32 | anchors)
	const uint64_t *mini_pos;
	Anchor *sq_a;             // squeezed anchors of the surviving hits, read by read at RgnRead::sq_off
	ref::Reg1 *regs;          // hit records, handed out by cursors[RGN_CUR_REGS]
	RgnAux *aux;
	RgnReadOut *rout;
	unsigned int *cursors;    // RGN_CUR_*
	const uint32_t *ref_len;  // reference sequence lengths and offsets (bases) in the packed sequence
	const uint64_t *ref_off;
	const uint8_t *qpool;     // the reads' nt4 codes (forward | reverse complement) and the packed reference: what an HPC window boundary looks at
	const uint32_t *S;
	uint32_t max_regs;        // capacity of regs / aux / plan / fin
	int rout_stride;          // 1, or 2 when the launch holds two-segment fragments (rout per segment)
	int lds_chains;           // chains per read the kernel keeps in LDS (multiple of 64)
	// planning
	RgnPlan *plan;
	RgnWin *win;
	KswJob *jobs;
	int32_t *gap_sites;       // scratch: per region a slice as long as its anchors (at sq_off + as), for the long-gap site lists
	uint32_t max_jobs;
	// consuming
	const KswRes *res;        // the DP kernels' results in LAUNCH order
	const uint32_t *perm;     // perm[job] = launch position of job
	FinRegion *fin;           // one per hit record (n_pieces = 0: not finished on the device)
	FinPiece *pieces;         // a region's pieces at [job0, job0 + n_win)
};

// span[2 r], span[2 r + 1]: query positions of the first and the last anchor of read r's first chain (what the long-join question of map.c:283-292 asks)
void launch_first_chain_span(int n_reads, const Anchor *a, const uint64_t *u, const uint64_t *a_off, const uint64_t *u_off, const int32_t *n_u, int32_t *span, void *stream);
// the listed slices of two source arrays packed back to back (a hand-back's anchors and minimizer positions on their way to the host)
struct RgnGather { uint64_t a_src, a_dst, mp_src, mp_dst; int32_t n_a, n_mp; uint32_t src, pad; };
void launch_gather_chains(int n, const RgnGather *g, const Anchor *a0, const Anchor *a1, const uint64_t *mp, Anchor *a_out, uint64_t *mp_out, void *stream);
void launch_chain_regs(const RgnBuffers &B, const RgnOpts &O, void *stream);
void launch_region_plan(const RgnBuffers &B, const RgnOpts &O, void *stream);
void launch_region_consume(const RgnBuffers &B, const RgnOpts &O, const uint32_t *cigar_pool, void *stream); // cigar_pool: the DP batch's CIGARs (KswRes::cigar_off)

// Is the whole chains -> hits -> windows -> consume path of a mapper with these options and this index the device's?  (single-segment
// reads are a property of the batch: the caller checks)
bool region_path_supported(const ref::MapOpt &opt, int idx_flag, int n_alt, bool has_junc_or_jump);

} // namespace mm2amd

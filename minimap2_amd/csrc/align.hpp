// Base-level alignment of chained regions, restructured for batched DP: the reference's mm_align1
// (align.c:645-914) interleaves window selection, ksw2 calls and CIGAR stitching per region; here every region is
// PLANNED first (pure integer work on the anchors: which (query,target) windows need a DP and with which flags),
// all planned windows of all reads go to the GPU as one job list, and the results are CONSUMED in the reference's
// order (Z-drop test, append, split, extension end points).  Work that the results themselves trigger (second-pass
// re-alignment, split-off regions, inversion rescue) is queued for the next round.
#pragma once
#include <vector>
#include "types.hpp"
#include "hits.hpp"
#include "flat_index.hpp"
#include "ksw_dev.hpp"
#include "region_finish.hpp"

namespace mm2amd {

void gen_score_matrix(const ref::MapOpt &opt, int8_t mat[25]);       // ksw_gen_ts_mat, align.c:26-36

// ksw_ll_qinit + ksw_ll_i16 (ksw2_ll_sse.c:37-152): striped local SW, score and end coordinates only
int ll_local_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t mat[25], int gapo, int gape, int *qe, int *te);

enum WindowKind : uint8_t { W_LEFT = 0, W_GAP = 1, W_RIGHT = 2, W_INV = 3 };

struct Window {            // one planned DP problem of a region
	int32_t qs, qe, rs, re; // half-open windows on the (strand-adjusted) query and on the reference
	int32_t bw;
	int32_t anchor_i;       // gap fills: index i of the anchor closing the window (relative to as1)
	int32_t job = -1;       // index into the current round's job list (-1 = none in this round)
	int32_t saved = -1;     // index into RegionTask::saved when the result was computed in an earlier round
	WindowKind kind;
	bool pass2 = false;     // (to be) re-aligned exactly after the approximate pass tripped the Z-drop test
	int32_t zdrop_code = 0;
	int8_t pre = 0;         // splice:sr (mm_align_sr_rna, align.c:370-400): 1 = the flank-only alignment is tried first, 2 = it has been tried
};

struct SavedResult { KswRes res; std::vector<uint32_t> cigar; };

struct RegionTask {
	Reg r{}, r2{};
	int32_t as1 = 0, cnt1 = 0;
	int32_t rid = 0, rev = 0;
	int32_t rs = 0, qs = 0, re = 0, qe = 0;      // running window ends (as in mm_align1)
	int32_t rs0 = 0, qs0 = 0, re0 = 0, qe0 = 0;  // extension limits
	int32_t rs1 = 0, qs1 = 0, re1 = 0, qe1 = 0;  // final alignment ends
	std::vector<Window> win;                     // [left?] gap... [right?]
	size_t next_win = 0;                         // consumption cursor
	bool has_left = false, has_right = false, dropped = false, done = false, planned = false;
	// spliced alignment (align.c:1068-1096): with both transcript strands enabled a region is aligned twice, once per assumed
	// strand, by two tasks that point at each other; the lead task sits in ReadAlign::order and receives the winner
	int32_t splice_flag = 0;                     // F_SPLICE_FOR / F_SPLICE_REV as passed to mm_align1
	int32_t ksw_flag = 0;                        // KSW_SPLICE_* bits every DP job of this task carries (align.c:684-689, :354)
	int32_t twin = -1;
	bool lead = true;
	bool chain_ungapped = false;                 // the chain spans equally many query and reference bases (splice:sr rule, align.c:1072)
	std::vector<SavedResult> saved;              // results carried over a round boundary (only when a region stalls)
	// A region whose windows all come back in one round is finished on the device (region_finish.hip): consuming its windows only RECORDS
	// their CIGARs (position in the round's pool) and sums their scores; a region that stalls appends what was recorded and goes on in the
	// reference's way (host_mode).
	std::vector<FinPiece> pieces;
	uint32_t sim_cap = 0, sim_n = 0, sim_last_op = 0; // what mm_extra_t::capacity and n_cigar would be had the pieces been appended one by one (align.c:305-334): the hand-over carries `capacity`
	int32_t dp_acc = 0;
	bool host_mode = false, awaiting_finish = false;
	// inversion-rescue tasks only (mm_align1_inv): where the extension starts and what it is anchored to
	int32_t inv_q0 = 0, inv_t0 = 0, inv_r2_qs = 0, inv_r2_qe = 0, inv_r1_re = 0, inv_qoff = 0, inv_toff = 0;
};

// A read's host-side nt4 codes: forward block, then reverse complement, each q4_stride(qlen) bytes -- the strand's codes plus >= 15 bytes of
// its own, so that update_extra's 16-byte loads past a run's end never touch a block another thread may be encoding.
inline uint64_t q4_stride(int qlen) { return ((uint64_t)qlen + 15 + 15) & ~(uint64_t)15; }

struct ReadAlign {        // per-read alignment state
	int qlen = 0;
	uint64_t qpool_off = 0, qpool_rev = 0;       // where this read's nt4 forward / reverse-complement bytes start in the device query pool
	uint8_t *q4 = nullptr;                       // fwd then reverse complement, nt4 codes, q4_stride(qlen) bytes each; owned by the caller
	const char *seq = nullptr;                   // the read as given (owned by the caller, alive for the read's rounds)
	bool q4_ready[2] = {false, false};           // a strand is encoded when something first looks at it (strand_codes): most reads use one
	Anchor *a = nullptr;             // the read's chained anchors (modified in place; owned by the caller)
	int n_a = 0;
	std::vector<RegionTask> tasks;               // in creation order
	std::vector<int> order;                      // output order: indices into tasks (inversions included)
	std::vector<uint8_t> tbytes;                 // composed targets of this round's jobs without KSWJ_T_PACKED (KswScoring::tbytes; a job's t_off indexes it)
	std::vector<uint32_t> juncs;                 // annotated splice sites inside this round's DP windows (KswScoring::juncs entries; a job's tag indexes it)
	std::vector<int> finish_queue;               // tasks of this round waiting for the device's region_finish (Aligner::consume fills, complete_finished drains)
};

// nt4 codes of one strand of the read (0 forward, 1 reverse complement), encoded on first use
const uint8_t *strand_codes(ReadAlign &ra, int strand);

class Aligner {
public:
	Aligner(const ref::MapOpt &opt, const FlatIndex &fi);
	// Prepare a read: encode, squeeze anchors, create one task per region (mm_align_skeleton, align.c:1048-1066).
	// q4: 2*qlen bytes of caller-owned storage that must outlive the read's rounds
	void begin_read(ReadAlign &ra, const char *seq, int qlen, RegVec &regs, Anchor *a, uint64_t qpool_fwd, uint64_t qpool_rev, uint8_t *q4);
	// Plan everything plannable and append the DP jobs of this round to `jobs`.
	void schedule(ReadAlign &ra, std::vector<KswJob> &jobs);
	// Consume results; returns true when the read still has unfinished work (another round needed).
	bool consume(ReadAlign &ra, const KswRes *res, const uint32_t *cigar_pool);
	// With device_finish(true), consume() leaves the regions it could finish in ra.finish_queue; the caller sends them through
	// Backend::finish_regions (describe_finish fills one record per queued task) and hands the results back: complete_finished() does what
	// follows a region's completion (strand choice, split-off tails, inversion rescue) and returns whether the read still has unfinished work.
	void device_finish(bool on) { device_finish_ = on; }
	void describe_finish(const ReadAlign &ra, int ti, FinRegion &fr) const; // everything but piece0 / out_off
	bool complete_finished(ReadAlign &ra, const FinResult *results, const FinRegion *regions, const uint32_t *cigars);
	// Collect the regions in output order and run the post-alignment steps of mm_align_skeleton (:1110-1118).
	void finish_read(ReadAlign &ra, RegVec &out);
	void finish_regs(int qlen, RegVec &out) const; // ... its second half: `out` holds the read's aligned regions in output order

	const int8_t *mat() const { return mat_; }
private:
	void add_region(ReadAlign &ra, const Reg &r, int order_pos);
	void join_strands(ReadAlign &ra, int lead_ti);
	void plan_region(ReadAlign &ra, RegionTask &t);
	void add_job(ReadAlign &ra, RegionTask &t, Window &w, int flag, int zdrop, int end_bonus, std::vector<KswJob> &jobs);
	bool qstrand_ = false; // MM_F_QSTRAND: reverse-strand hits keep the query as given and reverse-complement the reference
	void add_flank_job(ReadAlign &ra, RegionTask &t, Window &w, std::vector<KswJob> &jobs);
	bool consume_region(ReadAlign &ra, int ti, const KswRes *res, const uint32_t *cigar_pool);
	void consume_inversion(ReadAlign &ra, int ti, const KswRes *res, const uint32_t *cigar_pool);
	void finalize_region(ReadAlign &ra, RegionTask &t);
	bool after_finalize(ReadAlign &ra, int ti);
	void take_piece(RegionTask &t, const Window &w, const KswRes &ez, const uint32_t *cg, const uint32_t *cigar_pool);
	void materialize(RegionTask &t, const uint32_t *cigar_pool);
	bool device_finish_ = false;
	void try_inversion(ReadAlign &ra, int prev_ti, int ti, int pos_in_order);

	const ref::MapOpt &opt_;
	const FlatIndex &fi_;
	int8_t mat_[25];
	int bw_, bw_long_;
	std::vector<uint8_t> tbuf_;
	std::vector<int32_t> gap_sites_; // the chain's long-gap anchors (region_rules.hpp), reused from read to read
};

// mm_extra_t management (align.c:305-334)
void append_cigar(Reg &r, uint32_t n_cigar, const uint32_t *cigar);
// mm_jump_split (jump.c): extend clipped alignment ends across annotated junctions (FlatIndex::jump); qseq is the read as mapped (ASCII)
void jump_split(const FlatIndex &fi, const ref::MapOpt &opt, int32_t qlen, const char *qseq, Reg &r, int32_t ts_strand);
// mm_update_extra (align.c:254-303) incl. mm_fix_cigar (:105-181)
void update_extra(Reg &r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, bool is_eqx, bool log_gap);
// mm_test_zdrop (align.c:61-103)
int test_zdrop(const ref::MapOpt &opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, const uint32_t *cigar, const int8_t *mat);

} // namespace mm2amd

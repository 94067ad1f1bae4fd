// The device-side stages of the mapper behind one interface.  The product library links exactly one
// implementation (HipBackend: hand-written gfx950 kernels).  tests/ may build the same host pipeline against a
// checker implementation made of the oracle's C restatement to validate the host logic without a GPU; that
// checker is never part of libmm2amd.so.
#pragma once
#include <vector>
#include "types.hpp"
#include "flat_index.hpp"
#include "ksw_dev.hpp"
#include "region_finish.hpp"
#include "region_dev.hpp"

namespace mm2amd {

struct SeedChainParams {          // what mm_map_frag_core passes to seeding and chaining (map.c:250-281)
	int k, w, is_hpc;
	int mid_occ, max_max_occ, occ_dist;
	int sdust_thres = 0;          // > 0: minimizers lying mostly in SDUST-masked regions are dropped (mm_dust_minier, map.c:34-57,68)
	int q_mid_occ;                // threshold of the query-side filter mm_seed_mz_flt (map.c:251): mm_mapopt_t::mid_occ also in the max_occ pass (map.c:311)
	float q_occ_frac;
	int64_t flag;                 // mm_mapopt_t::flag (FOR_ONLY / REV_ONLY / NO_DIAG ... for skip_seed)
	int max_gap, max_gap_ref, max_frag_len, is_sr; // mm_mapopt_t::max_gap / max_gap_ref / max_frag_len as given, MM_F_SR: see chain_gaps()
	int bw, max_chain_skip, max_chain_iter, min_cnt, min_chain_score;
	float chn_pen_gap, chn_pen_skip;
	int is_cdna;
	int anchors_only = 0;         // 1: stop after the anchor sort and return every read's sorted anchors (n_u = 0): the caller chains them (a backend without an RMQ chainer)
	int rmq = 0;                  // 1: chain with mg_lchain_rmq's rules (MM_F_RMQ, map.c:275-277) instead of mg_lchain_dp's
	int rmq_inner_dist = 0, rmq_size_cap = 0; // mm_mapopt_t::rmq_inner_dist / rmq_size_cap
	int rmq_dev_max_anchors = 1 << 17; // reads with more anchors are handed back to the host's RMQ chainer (one wavefront walks a read's anchors one by one, microseconds each: whole contigs are faster on a host thread); MM2AMD_RMQ_DEV_MAX_ANCHORS lowers it for tests
	// long-join re-chaining of long reads (map.c:283-292): a read with more than one chain whose first chain leaves much of the read uncovered
	// (or covers a tenth of it) has its chained anchors sorted by reference position again and chained by mg_lchain_rmq with bw_long.
	// long_join = 1: seed_chain() does it for single-segment reads and says so in ReadChains::long_join_done; reads its RMQ kernel hands
	// back keep their first chains for the caller's host chainer.
	// 1: the caller takes the chains through align_regions(): seed_chain() leaves the chained anchors and the minimizer positions on the device
	// (ReadChains::a_p / mp_p null, the counts set); fetch_chains() brings the reads the host has to see
	int lazy_chains = 0;
	int long_join = 0, bw_long = 0, rmq_rescue_size = 0;
	float rmq_rescue_ratio = 0;
};

// the two chaining distance limits of a read of qlen bases (map.c:262-271): the query-side limit grows with the read for short
// reads, the reference-side limit follows max_frag_len when no explicit max_gap_ref is set
MM2_HD inline void chain_gaps(const SeedChainParams &p, int qlen, int *gap_ref, int *gap_qry)
{
	*gap_qry = p.is_sr && qlen > p.max_gap ? qlen : p.max_gap;
	if (p.max_gap_ref > 0) *gap_ref = p.max_gap_ref;
	else if (p.max_frag_len > 0) { const int g = p.max_frag_len - qlen; *gap_ref = g < p.max_gap ? p.max_gap : g; }
	else *gap_ref = p.max_gap;
}

class Backend {
public:
	virtual ~Backend() {}
	// Make the batch's sequences resident; the nt4 forward|reverse-complement pool places read i at
	// qpool_off[i] (2*len bytes).  Must be called before seed_chain()/ksw() of that batch.
	// begin_batch() prepares the NEXT batch; activate_batch() makes it the one seed_chain()/ksw() work on.  A backend that says
	// stages_beside_mapping() keeps two resident sets, and begin_batch() of batch k+1 may then run (on another thread) while batch k is
	// being mapped -- the hand-over costs no mapping time (pipeline step 0 beside step 1, map.c:541-577).
	virtual void begin_batch(const std::vector<ReadView> &reads, std::vector<uint64_t> &qpool_off) = 0;
	virtual void activate_batch() {}
	// Two batches can be under way at once (round 4: the lanes start on the staged batch while the last sub-batches of the current one finish):
	// current_set() names the resident set activate_batch() made current, bind_lane() tells a lane which set its next seed_chain() / ksw() /
	// finish_regions() calls work on.  A backend with one resident set has nothing to tell apart.
	virtual int current_set() const { return 0; }
	virtual void bind_lane(int /*lane*/, int /*set*/) {}
	virtual bool stages_beside_mapping() const { return false; }
	// sketch -> seed lookup -> anchor sort -> chaining DP -> chains, for reads [lo, hi) of the batch (out[i] is read lo+i)
	// `lane` selects one of n_lanes() independent sets of work buffers (one host thread per lane at a time); host-side parts use
	// up to n_threads pool threads.
	virtual void seed_chain(const SeedChainParams &p, long lo, long hi, int lane, int n_threads, std::vector<ReadChains> &out) = 0;
	virtual int n_lanes() const { return 1; }
	// how many lanes the caller is about to drive concurrently (<= n_lanes()): shared budgets are split among these only
	virtual void set_active_lanes(int /*n*/) {}
	// all-vs-all mapping (MM_F_NO_DIAG / MM_F_NO_DUAL): seed_chain() then applies skip_seed's read-name rules (map.c:81-91), which
	// need the reads' names in begin_batch().  Call once, before the first batch.
	virtual void enable_name_rules() {}
	virtual void enable_seq_len() {} // --qstrand: seed_chain() needs the reference sequence lengths (reverse-strand anchors in query-strand coordinates)
	virtual bool supports_junctions() const { return true; } // KswScoring::juncs honoured by ksw()
	virtual bool supports_sdust() const { return true; }
	virtual bool supports_long_join() const { return false; } // SeedChainParams::long_join honoured by seed_chain()
	virtual bool supports_rmq() const { return false; }    // SeedChainParams::rmq honoured by seed_chain(); otherwise the mapper asks for anchors_only and chains on the host
	virtual bool supports_byte_targets() const { return true; } // KswScoring::tbytes honoured by ksw() (splice:sr)     // SeedChainParams::sdust_thres honoured by seed_chain()
	virtual long max_reads_per_call() const { return 1L << 30; } // upper bound on hi - lo the backend accepts in seed_chain()
	// batched extension DP (ksw_extd2 semantics); *cigar points at the batch's packed CIGARs (backend-owned, valid until the next
	// call), addressed by res[i].cigar_off
	virtual void ksw(const std::vector<KswJob> &jobs, const KswScoring &sc, int lane, int n_threads, std::vector<KswRes> &res, const uint32_t **cigar) = 0;
	// The regions' last step on the device (region_finish.hpp): stitches the windows' CIGARs of this lane's LAST ksw() call (pieces address its
	// pool), left-aligns and counts as mm_update_extra does.  results[i] belongs to regions[i]; its CIGAR is at *cigars + regions[i].out_off
	// (backend-owned, valid until the lane's next call).  out_words = room needed in the output pool.
	// The whole base-level alignment of a sub-batch's reads (round 6: and of short-read pairs, segment by segment) on the device (region_dev.hpp): the chains of this lane's LAST
	// seed_chain() call become hit records, DP windows, DP results and finished regions without crossing PCIe; what comes back are the finished
	// hit records (host views, valid until the lane's next align_regions()).  in[i].skip: the host keeps read i.  A read whose RgnReadOut::flags
	// is non-zero goes through the host path from its chains.
	struct RegionReadIn { uint32_t hash; bool skip; int32_t gap_ref = 0; }; // gap_ref: the fragment's max_chain_gap_ref (map.c:264-270; two-segment fragments only)
	struct RegionBatchOut {
		const RgnReadOut *reads = nullptr; const ref::Reg1 *regs = nullptr; const RgnAux *aux = nullptr; const RgnPlan *plan = nullptr;
		const FinRegion *fin = nullptr; const FinResult *fin_res = nullptr; const uint32_t *cigars = nullptr;
		uint32_t n_regs = 0; size_t n_jobs = 0; double dp_cells = 0;
		int rout_stride = 1; // reads[] holds one entry per read, or (2: the sub-batch has two-segment fragments) two, one per segment
	};
	virtual bool aligns_regions() const { return false; }
	virtual void align_regions(int /*lane*/, const RgnOpts & /*O*/, const KswScoring & /*sc*/, bool /*log_gap*/, const std::vector<ReadChains> & /*chains*/, const std::vector<RegionReadIn> & /*in*/,
	                           int /*n_threads*/, RegionBatchOut & /*out*/) {}
	// the chained anchors and minimizer positions of the listed reads of this lane's last seed_chain(.., lazy_chains) come to the host
	// (one gather kernel, two copies); chains[i].a_p / mp_p point at them until the lane's next seed_chain()
	virtual void fetch_chains(int /*lane*/, const std::vector<long> & /*reads*/, std::vector<ReadChains> & /*chains*/) {}
	virtual bool finishes_regions() const { return false; }
	virtual void finish_regions(int /*lane*/, const std::vector<FinRegion> & /*regions*/, const std::vector<FinPiece> & /*pieces*/, size_t /*out_words*/, const int8_t * /*mat25*/,
	                            int /*q*/, int /*e*/, bool /*log_gap*/, std::vector<FinResult> & /*results*/, const uint32_t ** /*cigars*/) {}
};

} // namespace mm2amd

// Drop-in boundary: the entry points a minimap2 build calls instead of kt_for(worker_for) (map.c:576).
// Declared in include/mm2amd.h; the backend (HIP in the product) is supplied by make_backend().
#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include "../../include/mm2amd.h"
#include "mapper.hpp"
#include "format.hpp"
#include "index_handle.hpp"
#include "threads.hpp"

namespace mm2amd {
int effective_cpus(); // capi_common.cpp
// backend_hip.cpp in the product: `device` < 0 = the process's default device; `replica` numbers the backends of one context;
// `tables_device` is where `device_tables` live (a backend on another device copies them)
Backend *make_backend(const FlatIndex &fi, void *device_tables, int n_threads, int device, int replica, int tables_device);
int backend_device_count();
// The minimizer tables of `fi` (sequence table and packed sequence set) built by the backend on `device` (< 0: the default one);
// returns the tables (for make_backend's device_tables; *on_device = their device) or nullptr when this backend wants host tables.
void *backend_build_index_tables(FlatIndex &fi, int device, int *on_device);
void backend_free_index_tables(void *tables);
const char *backend_name();
void capi_set_error(const std::string &msg);              // capi_common.cpp
int capi_fail(int code, const std::string &msg);
}

using namespace mm2amd;

extern "C" long long mm2amd_alloc_counter(int which); // device allocations, pinned allocations, nanoseconds spent in them; 3..6: the banded gap fill's windows tried in 128 / 256 diagonals, widened, recomputed as rectangles (process-wide)

namespace {
// Where the results of one mapper fragment go: reads o .. o+n_out-1 of the caller's arrays; flip_len[j] >= 0 when read o+j was
// mapped reverse-complemented and its hits have to be turned back (length of that read), see hand_over().
struct OutSlot { int o, n_out, flip_len[2]; int weak = 0, len0 = 0; }; // weak: 1 / 2 = first / second mate of a pair mapped separately and paired afterwards (MM_F_WEAK_PAIRING)

// One mapper per GPU the context uses (mm_gpu_init_multi): every replica holds a full copy of the index in its device's HBM and
// maps a contiguous share of each batch, cut by cumulative bases; the shares are independent, so there is no exchange between
// devices -- results land in the caller's arrays, which are host memory (the hit gather is only needed across processes).
struct Replica {
	int device = -1;
	std::unique_ptr<Backend> be;
	std::unique_ptr<Mapper> mapper;
};

// A batch as handed over by mm_gpu_batch_stage*: the mapper-side fragments (views into the caller's sequences and into `flipped`), where
// their results go, and the replicas' shares (replica r maps fragments [cut[r], cut[r+1])).
struct StagedBatch {
	std::vector<ReadView> views;
	std::vector<OutSlot> slots;
	std::string flipped; // the mates handed over reverse-complemented (pe_ori), back to back
	std::vector<long> cut;
};

struct MapContext {
	FlatIndex fi_own;                 // index flattened from a reference mm_idx_t (mm_gpu_init)
	const FlatIndex *fi = nullptr;    // the index in use (fi_own, or the one inside an mm2amd_index_t)
	ref::MapOpt opt;
	// Two batches: the one being (or last) mapped and the one staged next.  stage_mu guards which is which; a staging call fills
	// batch[1 - cur] and sets `pending`, the next mapping call takes it over (cur flips, pending clears, stage_cv wakes a queued stager).
	StagedBatch batch[2];
	int cur = 0;
	bool pending = false, has_current = false;
	std::mutex stage_mu;
	std::condition_variable stage_cv;
	bool stages_beside_mapping = false; // every replica's backend keeps two resident sets
	std::mutex stats_mu;
	FormatScratch fmt;                // mm_gpu_format_batch_view's reusable text buffers
	std::mutex fmt_mu;
	int n_threads = 1;
	uint64_t generation = 0;
	const void *mi_ptr = nullptr;     // the reference index this context mirrors (mm_gpu_init; the batch-of-one calls compare it ...
	uint64_t mi_stamp[4] = { 0, 0, 0, 0 }; // ... and these fields of it: an index changed in place or another one at a recycled address is not mistaken for it)
	void *built_tables = nullptr;     // minimizer tables the backend built for fi_own (mm_gpu_init); freed with the context
	std::vector<Replica> reps;
	~MapContext() { reps.clear(); if (built_tables) backend_free_index_tables(built_tables); }
	MapperStats stats;                // summed over the replicas of the last run
};
// g_ctx_mu guards the context's lifetime (init / destroy take it exclusively, every other call shared); g_map_mu serialises the
// calls that stage or map (one batch at a time, like the reference's pipeline step 1).  mm_gpu_format_batch only reads the
// context, so the output stage of batch k runs beside the mapping of batch k+1 (map.c:541-643: steps 1 and 2 of worker_pipeline).
std::shared_mutex g_ctx_mu;
std::mutex g_map_mu;
std::unique_ptr<MapContext> g_ctx;
uint64_t g_generation = 0;

// fragments [0, n) -> one contiguous range per replica with about the same number of bases (shard.py: split_by_bases); a pair
// of mates mapped separately and paired afterwards (OutSlot::weak) stays in one share
void shard_by_bases(const std::vector<ReadView> &reads, const std::vector<OutSlot> &slots, int n_parts, std::vector<long> &cut)
{
	const long n = (long)reads.size();
	std::vector<uint64_t> acc((size_t)n + 1, 0);
	for (long i = 0; i < n; ++i) acc[i + 1] = acc[i] + (uint64_t)reads[i].total();
	cut.assign((size_t)n_parts + 1, n);
	cut[0] = 0;
	for (int p = 1; p < n_parts; ++p) {
		const uint64_t want = acc[n] * (uint64_t)p / (uint64_t)n_parts;
		long c = (long)(std::lower_bound(acc.begin(), acc.end(), want) - acc.begin());
		c = std::max(c, cut[p - 1]);
		if (c < n && slots[c].weak == 2) ++c;
		cut[p] = std::min(c, n);
	}
}

// every replica maps its share on a host thread of its own
template <typename F>
void for_each_replica(MapContext &c, F &&f)
{
	std::vector<std::thread> th;
	std::vector<std::exception_ptr> err(c.reps.size());
	for (size_t r = 1; r < c.reps.size(); ++r) th.emplace_back([&, r] { name_thread("mm2replica"); try { f(c.reps[r]); } catch (...) { err[r] = std::current_exception(); } });
	try { f(c.reps[0]); } catch (...) { err[0] = std::current_exception(); }
	for (auto &t : th) t.join();
	for (auto &e : err) if (e) std::rethrow_exception(e);
}

void run_replicas(MapContext &c, const StagedBatch &bt, std::vector<ReadResult> &out)
{
	std::vector<std::vector<ReadResult>> part(c.reps.size());
	for_each_replica(c, [&](Replica &rp) { rp.mapper->run(part[&rp - c.reps.data()]); });
	out.clear();
	out.resize(bt.views.size());
	MapperStats sum;
	for (size_t r = 0; r < c.reps.size(); ++r) {
		for (long i = bt.cut[r]; i < bt.cut[r + 1]; ++i) out[i] = std::move(part[r][i - bt.cut[r]]);
		const MapperStats &s = c.reps[r].mapper->stats;
		sum.t_seed_chain += s.t_seed_chain, sum.t_host_pre += s.t_host_pre, sum.t_plan += s.t_plan, sum.t_ksw += s.t_ksw, sum.t_consume += s.t_consume;
		sum.t_finish += s.t_finish, sum.n_jobs += s.n_jobs, sum.n_rounds += s.n_rounds, sum.dp_cells += s.dp_cells;
		sum.c_seed_chain += s.c_seed_chain, sum.c_host_pre += s.c_host_pre, sum.c_plan += s.c_plan, sum.c_ksw += s.c_ksw, sum.c_consume += s.c_consume, sum.c_finish += s.c_finish;
		sum.n_early_sub += s.n_early_sub;
		sum.d_seed_chain += s.d_seed_chain, sum.d_host_pre += s.d_host_pre, sum.d_plan += s.d_plan, sum.d_ksw += s.d_ksw, sum.d_consume += s.d_consume, sum.d_finish += s.d_finish;
		sum.n_long_join_dev += s.n_long_join_dev, sum.n_long_join_host += s.n_long_join_host;
		sum.n_region_reads_dev += s.n_region_reads_dev, sum.n_region_reads_host += s.n_region_reads_host;
	}
	std::lock_guard<std::mutex> lk(c.stats_mu);
	c.stats = sum;
}

// hand the batch's shares to the replicas' mappers (which copy the sequences to their devices); call with stage_mu held
void stage_replicas(MapContext &c, StagedBatch &bt, bool queued = false)
{
	shard_by_bases(bt.views, bt.slots, (int)c.reps.size(), bt.cut);
	for_each_replica(c, [&](Replica &rp) {
		const size_t r = (size_t)(&rp - c.reps.data());
		std::vector<ReadView> share(bt.views.begin() + bt.cut[r], bt.views.begin() + bt.cut[r + 1]);
		rp.mapper->stage(share, queued); // queued: a pipeline's hand-over -- mapped exactly once, next; the lanes may start on it early
	});
}

// the staged batch becomes the current one; call with stage_mu held
void take_staged(MapContext &c)
{
	for (Replica &rp : c.reps) rp.mapper->take();
	c.cur = 1 - c.cur, c.pending = false, c.has_current = true;
	c.stage_cv.notify_all();
}

int build_context(std::unique_ptr<MapContext> &c, void *device_tables, int tables_device, int n_threads, int n_gpus, const int *device_ids)
{
	if (n_gpus <= 0 && device_ids) return capi_fail(MM2AMD_EINVAL, "[mm2amd] device_ids given without n_gpus: how many entries does the array hold?");
	if (n_gpus <= 0) { n_gpus = 1; if (const char *e = getenv("MM2AMD_GPUS")) n_gpus = std::max(1, atoi(e)); }
	std::vector<int> env_ids;
	if (!device_ids && getenv("MM2AMD_DEVICE_IDS")) { // "0,0" or "3,2,1,0": ordinals for the replicas when the caller names none
		for (const char *p = getenv("MM2AMD_DEVICE_IDS"); *p;) { env_ids.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
		if ((int)env_ids.size() < n_gpus) return capi_fail(MM2AMD_EINVAL, "[mm2amd] MM2AMD_DEVICE_IDS names fewer devices than replicas were asked for");
		device_ids = env_ids.data();
	}
	if (n_gpus > 16) return capi_fail(MM2AMD_EINVAL, "[mm2amd] at most 16 replicas per context");
	if (!device_ids && n_gpus > 1 && n_gpus > backend_device_count()) return capi_fail(MM2AMD_ENODEV, "[mm2amd] more GPUs requested than this process can see");
	if (n_threads <= 0) n_threads = std::min(64 * n_gpus, effective_cpus()); // the host stages stop scaling (and start contending) beyond ~64 threads per GPU; a CPU quota counts
	c->n_threads = n_threads;
	const int per = std::max(1, n_threads / n_gpus);
	c->reps.resize(n_gpus);
	for (int r = 0; r < n_gpus; ++r) {
		Replica &rp = c->reps[r];
		rp.device = device_ids ? device_ids[r] : n_gpus > 1 ? r : -1;
		rp.be.reset(make_backend(*c->fi, device_tables, per, rp.device, r, tables_device));
		rp.mapper.reset(new Mapper(*c->fi, c->opt, *rp.be, per));
	}
	c->stages_beside_mapping = true;
	for (Replica &rp : c->reps) c->stages_beside_mapping &= rp.mapper->stages_beside_mapping();
	c->generation = ++g_generation;
	return 0;
}

uint64_t g_stamp_hash(const ref::Idx *mi) { return (uint64_t)mi->k | (uint64_t)mi->w << 8 | (uint64_t)(uint32_t)mi->flag << 16 | (uint64_t)mi->n_seq << 32; }
}

extern "C" {

int mm_gpu_init_multi(const void *mi, const void *opt, int n_threads, int n_gpus, const int *device_ids)
{
	if (!mi || !opt) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_init: null index or options");
	std::unique_lock<std::shared_mutex> lk(g_ctx_mu);
	try {
		std::unique_ptr<MapContext> c(new MapContext);
		c->opt = *(const ref::MapOpt *)opt;
		// the index's minimizer tables: rebuilt on the device from its packed sequence (about a second for 3 Gb) rather than collected
		// from the reference's hash tables on the host (most of a minute); MM2AMD_INIT_FROM_HASH=1 or an index without sequence
		// (MM_I_NO_SEQ) takes the host route
		const ref::Idx *rmi = (const ref::Idx *)mi;
		const bool on_device = rmi->S && !(rmi->flag & ref::I_NO_SEQ) && !getenv("MM2AMD_INIT_FROM_HASH");
		c->fi_own.from_reference(rmi, !on_device);
		c->fi = &c->fi_own;
		int tables_device = -1;
		if (on_device) {
			int first = -1; // the device the first replica will run on: the tables are built where they are used
			if (device_ids && n_gpus > 0) first = device_ids[0];
			else if (!device_ids && getenv("MM2AMD_DEVICE_IDS")) first = atoi(getenv("MM2AMD_DEVICE_IDS"));
			else if (n_gpus > 1 || (n_gpus <= 0 && getenv("MM2AMD_GPUS") && atoi(getenv("MM2AMD_GPUS")) > 1)) first = 0;
			c->built_tables = backend_build_index_tables(c->fi_own, first, &tables_device);
			if (!c->built_tables) c->fi_own.from_reference(rmi, true); // a backend that works from host tables
		}
		if (int rc = build_context(c, c->built_tables, tables_device, n_threads, n_gpus, device_ids)) return rc;
		c->mi_ptr = mi;
		c->mi_stamp[0] = g_stamp_hash(rmi), c->mi_stamp[1] = (uint64_t)(uintptr_t)rmi->S, c->mi_stamp[2] = (uint64_t)(uintptr_t)rmi->seq, c->mi_stamp[3] = (uint64_t)rmi->n_alt;
		g_ctx = std::move(c);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		const std::string s = e.what();
		return capi_fail(s.find("no HIP device") != std::string::npos ? MM2AMD_ENODEV : MM2AMD_EHIP, s);
	}
}

int mm_gpu_init_index_multi(const mm2amd_index_t *idx, const void *opt, int n_threads, int n_gpus, const int *device_ids)
{
	if (!idx || !opt) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_init_index: null index or options");
	std::unique_lock<std::shared_mutex> lk(g_ctx_mu);
	try {
		std::unique_ptr<MapContext> c(new MapContext);
		c->opt = *(const ref::MapOpt *)opt;
		c->fi = &index_flat((const IndexHandle *)idx);
		if (int rc = build_context(c, index_device_tables((const IndexHandle *)idx), index_device((const IndexHandle *)idx), n_threads, n_gpus, device_ids)) return rc;
		g_ctx = std::move(c);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		const std::string s = e.what();
		return capi_fail(s.find("no HIP device") != std::string::npos ? MM2AMD_ENODEV : MM2AMD_EHIP, s);
	}
}

int mm_gpu_init(const void *mi, const void *opt, int n_threads) { return mm_gpu_init_multi(mi, opt, n_threads, 0, nullptr); }
int mm_gpu_init_index(const mm2amd_index_t *idx, const void *opt, int n_threads) { return mm_gpu_init_index_multi(idx, opt, n_threads, 0, nullptr); }

uint64_t mm_gpu_context_generation(void)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	return g_ctx ? g_ctx->generation : 0;
}

int mm_gpu_n_replicas(void)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	return g_ctx ? (int)g_ctx->reps.size() : 0;
}

// mm_revcomp_bseq (mmpriv.h) on a copy: complement table of bseq.c:11-28 (IUPAC codes, case kept, other bytes unchanged)
static void revcomp_into(const char *seq, int len, char *out)
{
	static const struct Table {
		unsigned char t[256];
		Table()
		{
			for (int i = 0; i < 256; ++i) t[i] = (unsigned char)i;
			const char *from = "ACGTUMRWSYKVHDBN", *to = "TGCAAKYWSRMBDHVN";
			for (int i = 0; from[i]; ++i) t[(unsigned char)from[i]] = (unsigned char)to[i], t[(unsigned char)(from[i] + 32)] = (unsigned char)(to[i] + 32);
		}
	} tab;
	for (int i = 0; i < len; ++i) out[len - 1 - i] = (char)tab.t[(unsigned char)seq[i]];
}

// The fragments of a batch as the mapper sees them.  A two-segment fragment (paired-end reads) is handed over in mapping
// orientation: worker_for reverse-complements a mate in place according to pe_ori before mapping and back afterwards
// (map.c:436-442, 457-473); here the flipped copy lives in `flipped` and the caller's buffers are left alone.  With
// MM_F_INDEPEND_SEG the two reads of a pair are mapped as two single reads (map.c:443-448), still in flipped orientation.
static int collect_views(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, const ref::MapOpt &opt, int n_threads, std::vector<ReadView> &reads,
                         std::vector<OutSlot> &slots, std::string &flipped)
{
	const ref::Bseq1 *seq = (const ref::Bseq1 *)seq_;
	const int pe_ori = opt.pe_ori;
	const bool independent = (opt.flag & ref::F_INDEPEND_SEG) != 0;
	reads.clear(), slots.clear();
	// the flipped mates in one pool (round 6: a million pairs were a million strings, filled one after the other by the hand-over's own thread -- 0.15 s of a
	// 0.25 s step); where each goes is known from the lengths, the copies themselves are made by the side pool below
	struct Flip { const char *src; int len; size_t at; };
	std::vector<Flip> flips;
	size_t flip_bytes = 0;
	for (int i = 0; i < n_frag; ++i) {
		if (n_seg[i] != 1 && n_seg[i] != 2) return capi_fail(MM2AMD_EINVAL, "[mm2amd] fragments of more than two segments are not implemented");
		if (n_seg[i] == 2) {
			if (pe_ori >> 1 & 1) flip_bytes += (size_t)seq[seg_off[i]].l_seq;
			if (pe_ori & 1) flip_bytes += (size_t)seq[seg_off[i] + 1].l_seq;
		}
	}
	flipped.resize(flip_bytes + 1); // sized first: the views point into it
	char *const pool = &flipped[0];
	if (flip_bytes) flips.reserve((size_t)n_frag);
	size_t at = 0;
	auto flip = [&](const char *src, int len) { const char *dst = pool + at; flips.push_back(Flip{src, len, at}); at += (size_t)len; return dst; };
	reads.reserve((size_t)n_frag), slots.reserve((size_t)n_frag);
	for (int i = 0; i < n_frag; ++i) {
		const int o = seg_off[i];
		const ref::Bseq1 &s = seq[o];
		ReadView v;
		OutSlot sl = { o, 1, { -1, -1 } };
		v.seq = s.seq, v.len = s.l_seq, v.name = s.name;
		if (n_seg[i] == 2) {
			const ref::Bseq1 &s2 = seq[o + 1];
			const char *q2 = s2.seq;
			if (pe_ori >> 1 & 1) { v.seq = flip(s.seq, s.l_seq); sl.flip_len[0] = s.l_seq; }
			if (pe_ori & 1) { q2 = flip(s2.seq, s2.l_seq); sl.flip_len[1] = s2.l_seq; }
			const bool weak = !independent && (opt.flag & ref::F_WEAK_PAIRING) && pe_ori >= 0 && (opt.flag & ref::F_CIGAR); // mm_map_frag, map.c:382-387
			if (independent || weak) {
				ReadView v2;
				OutSlot sl2 = { o + 1, 1, { sl.flip_len[1], -1 } };
				v2.seq = q2, v2.len = s2.l_seq, v2.name = weak ? s.name : s2.name; // mm_map_frag hands the fragment's (first) name to both calls
				sl.flip_len[1] = -1;
				if (weak) sl.weak = 1, sl2.weak = 2;
				sl.len0 = s.l_seq, sl2.len0 = s2.l_seq;
				reads.push_back(v), slots.push_back(sl);
				reads.push_back(v2), slots.push_back(sl2);
				continue;
			}
			v.seq2 = q2, v.len2 = s2.l_seq, sl.n_out = 2;
		}
		reads.push_back(v), slots.push_back(sl);
	}
	parallel_for_side(n_threads, (long)flips.size(), [&](long k, int) { revcomp_into(flips[k].src, flips[k].len, pool + flips[k].at); }, 1024);
	return 0;
}

static void *regs_block(const RegVec &v)
{
	if (v.empty()) return nullptr;
	void *p = malloc(v.size() * sizeof(ref::Reg1)); // handed over as one libc block, like the reference's realloc'd array (map.c:340)
	memcpy(p, v.data(), v.size() * sizeof(ref::Reg1));
	return p;
}

static void hand_over(const std::vector<OutSlot> &slots, std::vector<ReadResult> &out, int *n_reg, void **reg, int *rep_len, int *frag_gap)
{
	const ref::MapOpt &opt = g_ctx->opt;
	parallel_for(g_ctx ? g_ctx->n_threads : 1, (long)slots.size(), [&](long i, int) {
		const OutSlot &sl = slots[i];
		if (sl.weak == 1) { // the two mates were mapped on their own: pair them now (map.c:386), in mapping orientation
			const int qlens[2] = { slots[i].len0, slots[i + 1].len0 };
			RegVec both[2];
			both[0].swap(out[i].regs), both[1].swap(out[i + 1].regs);
			pair_hits(opt.max_gap_ref, opt.pe_bonus, opt.a * 2 + opt.b, opt.a, qlens, both);
			both[0].swap(out[i].regs), both[1].swap(out[i + 1].regs);
			out[i].rep_len = out[i + 1].rep_len, out[i].frag_gap = out[i + 1].frag_gap; // mm_tbuf_t keeps the last call's values (map.c:450-453)
		}
	}, 256);
	parallel_for(g_ctx ? g_ctx->n_threads : 1, (long)slots.size(), [&](long i, int) {
		const OutSlot &sl = slots[i];
		for (int j = 0; j < sl.n_out; ++j) {
			RegVec &regs = j == 0 ? out[i].regs : out[i].regs2;
			if (sl.flip_len[j] >= 0) { // back to the strand the read was given in (map.c:457-473)
				const int qlen = sl.flip_len[j];
				for (ref::Reg1 &r : regs) {
					const int t = r.qs;
					r.qs = qlen - r.qe, r.qe = qlen - t;
					r.rev = !r.rev;
					if (r.p) {
						if (r.p->trans_strand == 1) r.p->trans_strand = 2;
						else if (r.p->trans_strand == 2) r.p->trans_strand = 1;
					}
				}
			}
			n_reg[sl.o + j] = (int)regs.size();
			reg[sl.o + j] = regs_block(regs);
			if (rep_len) rep_len[sl.o + j] = out[i].rep_len;
			if (frag_gap) frag_gap[sl.o + j] = out[i].frag_gap;
		}
	}, 256);
}

// The hand-over of a batch (pipeline step 0: the reads are copied to the GPU).  `queued`: wait until the previously staged batch has been
// taken over by a mapping call instead of replacing it -- the form a pipeline uses, where staging batch k+1 runs beside the mapping
// of batch k and every staged batch is mapped exactly once, in order (kt_pipeline's ordering rule, kthread.c:107-112).
static int stage_batch(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, bool queued)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_batch_stage called before mm_gpu_init");
	if (n_frag < 0 || (n_frag > 0 && (!seg_off || !n_seg || !seq_))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_batch_stage: bad arguments");
	MapContext &c = *g_ctx;
	std::unique_lock<std::mutex> lk_map(g_map_mu, std::defer_lock);
	std::unique_lock<std::mutex> lk_st(c.stage_mu, std::defer_lock);
	for (;;) { // lock order: mapping before staging; a queued hand-over waits for the staged batch to be taken, holding neither
		if (!c.stages_beside_mapping) lk_map.lock(); // a backend with one resident set: the hand-over waits for the mapping call in flight
		lk_st.lock();
		if (!queued || !c.pending) break;
		if (lk_map.owns_lock()) lk_map.unlock();
		c.stage_cv.wait(lk_st, [&] { return !c.pending; });
		lk_st.unlock();
	}
	try {
		c.pending = false;
		StagedBatch &bt = c.batch[1 - c.cur];
		if (int rc = collect_views(n_frag, seg_off, n_seg, seq_, c.opt, c.n_threads, bt.views, bt.slots, bt.flipped)) return rc;
		stage_replicas(c, bt, queued);
		c.pending = true;
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EHIP, e.what());
	}
}

int mm_gpu_batch_stage(int n_frag, const int *seg_off, const int *n_seg, const void *seq_) { return stage_batch(n_frag, seg_off, n_seg, seq_, false); }
int mm_gpu_batch_stage_queued(int n_frag, const int *seg_off, const int *n_seg, const void *seq_) { return stage_batch(n_frag, seg_off, n_seg, seq_, true); }

// drops a staged batch that will not be mapped (a pipeline shutting down after an error), so that a queued hand-over is not left waiting
void mm_gpu_batch_discard(void)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	if (!g_ctx) return;
	std::lock_guard<std::mutex> lk_st(g_ctx->stage_mu);
	for (Replica &rp : g_ctx->reps) rp.mapper->discard(); // (sub-batches of it the lanes had started on are awaited and dropped)
	g_ctx->pending = false;
	g_ctx->stage_cv.notify_all();
}

namespace {
// MM2AMD_INJECT_BATCH_FAILURE=k (test knob): the k-th mapping call of the process (1-based) fails AFTER its batch has been mapped, the way a device
// fault reported at the end of a batch would -- the outputs stay untouched, MM2AMD_EHIP comes back, the context stays usable: what the hook's
// per-batch fallback to the reference's own kt_for(worker_for) relies on (tests/test_gpu_dropin.py).
bool inject_batch_failure()
{
	static const long k = getenv("MM2AMD_INJECT_BATCH_FAILURE") ? atol(getenv("MM2AMD_INJECT_BATCH_FAILURE")) : 0;
	static std::atomic<long> calls{0};
	return k > 0 && ++calls == k;
}
void free_results(std::vector<ReadResult> &out)
{
	for (ReadResult &r : out) { for (Reg &x : r.regs) free(x.p); for (Reg &x : r.regs2) free(x.p); }
	out.clear();
}
}

int mm_gpu_map_staged(int *n_reg, void **reg, int *rep_len, int *frag_gap)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	std::lock_guard<std::mutex> lk_map(g_map_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_map_staged: no staged batch");
	if (!n_reg || !reg) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_map_staged: bad arguments");
	MapContext &c = *g_ctx;
	try {
		{
			std::lock_guard<std::mutex> lk_st(c.stage_mu);
			if (c.pending) take_staged(c);                       // the batch staged last ...
			else if (!c.has_current) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_map_staged: no staged batch");
		}                                                        // ... or, when nothing new was staged, the current one again
		const StagedBatch &bt = c.batch[c.cur];
		std::vector<ReadResult> out;
		run_replicas(c, bt, out);
		if (inject_batch_failure()) { free_results(out); throw std::runtime_error("[mm2amd] MM2AMD_INJECT_BATCH_FAILURE: this batch is reported as failed (test knob)"); }
		hand_over(bt.slots, out, n_reg, reg, rep_len, frag_gap);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EHIP, e.what());
	}
}

int mm_gpu_format_batch(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, const int *n_reg, void *const *reg, const int *rep_len, char **out, size_t *out_len)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_format_batch called before mm_gpu_init");
	if (n_frag < 0 || !out || !out_len || (n_frag > 0 && (!seq_ || !n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_format_batch: bad arguments");
	for (int i = 0; i < n_frag; ++i)
		if (n_seg && n_seg[i] != 1 && n_seg[i] != 2) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_format_batch: fragments of one or two segments only");
	const std::string why = format_check(g_ctx->opt);
	if (!why.empty()) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_format_batch: " + why);
	try {
		*out = format_batch(*g_ctx->fi, g_ctx->opt, g_ctx->n_threads, n_frag, seg_off, n_seg, (const ref::Bseq1 *)seq_, n_reg, reg, rep_len, out_len);
		if (!*out) return capi_fail(MM2AMD_ENOMEM, "[mm2amd] mm_gpu_format_batch: out of memory");
		return 0;
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	}
}

int mm_gpu_format_batch_view(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, const int *n_reg, void *const *reg, const int *rep_len, const char **out, size_t *out_len)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_format_batch_view called before mm_gpu_init");
	if (n_frag < 0 || !out || !out_len || (n_frag > 0 && (!seq_ || !n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_format_batch_view: bad arguments");
	for (int i = 0; i < n_frag; ++i)
		if (n_seg && n_seg[i] != 1 && n_seg[i] != 2) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_format_batch_view: fragments of one or two segments only");
	const std::string why = format_check(g_ctx->opt);
	if (!why.empty()) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_format_batch_view: " + why);
	std::lock_guard<std::mutex> lk_fmt(g_ctx->fmt_mu); // one formatting call at a time uses the buffers (pipeline step 2)
	try {
		// (this step runs beside the mapping of the next batch: half the threads)
		*out = format_batch_view(*g_ctx->fi, g_ctx->opt, std::max(1, g_ctx->n_threads / 2), n_frag, seg_off, n_seg, (const ref::Bseq1 *)seq_, n_reg, reg, rep_len, g_ctx->fmt, out_len);
		if (!*out) return capi_fail(MM2AMD_ENOMEM, "[mm2amd] mm_gpu_format_batch_view: out of memory");
		return 0;
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	}
}

void mm2amd_free_regs(int n_frag, int *n_reg, void **reg)
{
	if (!n_reg || !reg) return;
	int nt = 1;
	{ std::shared_lock<std::shared_mutex> lk(g_ctx_mu); if (g_ctx) nt = std::min(g_ctx->n_threads, 16); }
	parallel_for_side(nt, n_frag, [&](long i, int) { // (a mini-batch holds a few hundred thousand libc blocks: tenths of a second on one thread)
		ref::Reg1 *r = (ref::Reg1 *)reg[i];
		for (int j = 0; j < n_reg[i]; ++j) free(r[j].p);
		free(r);
		reg[i] = nullptr, n_reg[i] = 0;
	}, 2048);
}

// ---- hit records as one flat byte payload: the unit of the multi-GPU gather ----
// per fragment: int32 n_reg, then per hit the 80-byte mm_reg1_t (pointer field zeroed), a uint32 "has extra", and when
// set the 28-byte mm_extra_t header followed by n_cigar uint32 CIGAR words.
int64_t mm2amd_pack_regs(int n_frag, const int *n_reg, void *const *reg, uint8_t *buf, int64_t cap)
{
	if (n_frag < 0 || (n_frag > 0 && (!n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_pack_regs: bad arguments");
	// sizes first (prefix sums give every fragment its slot), then the copies, both on the pool threads
	int nt = 1;
	{ std::shared_lock<std::shared_mutex> lk(g_ctx_mu); if (g_ctx) nt = g_ctx->n_threads; }
	std::vector<int64_t> off((size_t)n_frag + 1, 0);
	parallel_for(nt, n_frag, [&](long i, int) {
		int64_t sz = 4;
		const ref::Reg1 *r = (const ref::Reg1 *)reg[i];
		for (int j = 0; j < n_reg[i]; ++j) sz += (int64_t)sizeof(ref::Reg1) + 4 + (r[j].p ? (int64_t)sizeof(ref::Extra) + 4ll * r[j].p->n_cigar : 0);
		off[i + 1] = sz;
	}, 1024);
	for (int i = 0; i < n_frag; ++i) off[i + 1] += off[i];
	const int64_t need = off[n_frag];
	if (!buf) return need;
	if (cap < need) return capi_fail(MM2AMD_ENOMEM, "[mm2amd] mm2amd_pack_regs: buffer too small");
	parallel_for(nt, n_frag, [&](long i, int) {
		uint8_t *o = buf + off[i];
		const int32_t n = n_reg[i];
		memcpy(o, &n, 4), o += 4;
		const ref::Reg1 *r = (const ref::Reg1 *)reg[i];
		for (int j = 0; j < n; ++j) {
			ref::Reg1 t = r[j];
			const ref::Extra *ex = t.p;
			t.p = nullptr;
			memcpy(o, &t, sizeof t), o += sizeof t;
			const uint32_t has = ex ? 1u : 0u;
			memcpy(o, &has, 4), o += 4;
			if (ex) { const size_t nb = sizeof(ref::Extra) + 4ull * ex->n_cigar; memcpy(o, ex, nb), o += nb; }
		}
	}, 1024);
	return need;
}

int mm2amd_unpack_regs(const uint8_t *buf, int64_t size, int n_frag, int *n_reg, void **reg)
{
	if (!buf || size < 0 || n_frag < 0 || (n_frag > 0 && (!n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: bad arguments");
	for (int i = 0; i < n_frag; ++i) n_reg[i] = 0, reg[i] = nullptr; // whatever happens below, the caller can hand the arrays to mm2amd_free_regs
	auto bail = [&](int code, const char *msg) { mm2amd_free_regs(n_frag, n_reg, reg); return capi_fail(code, msg); };
	const uint8_t *o = buf, *end = buf + size;
	for (int i = 0; i < n_frag; ++i) {
		int32_t n;
		if (end - o < 4) return bail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
		memcpy(&n, o, 4), o += 4;
		if (n < 0 || (int64_t)n > (end - o) / (int64_t)(sizeof(ref::Reg1) + 4)) return bail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: hit count beyond the payload");
		if (n == 0) continue;
		ref::Reg1 *r = (ref::Reg1 *)calloc(n, sizeof(ref::Reg1));
		if (!r) return bail(MM2AMD_ENOMEM, "[mm2amd] mm2amd_unpack_regs: out of memory");
		reg[i] = r, n_reg[i] = n; // the array is zeroed: unfilled entries have no extra block
		for (int j = 0; j < n; ++j) {
			uint32_t has;
			if (end - o < (int64_t)sizeof(ref::Reg1) + 4) return bail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
			memcpy(&r[j], o, sizeof(ref::Reg1)), o += sizeof(ref::Reg1);
			memcpy(&has, o, 4), o += 4;
			r[j].p = nullptr;
			if (has) {
				ref::Extra hd;
				if (end - o < (int64_t)sizeof hd) return bail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
				memcpy(&hd, o, sizeof hd);
				const size_t nb = sizeof(ref::Extra) + 4ull * hd.n_cigar;
				if ((size_t)(end - o) < nb) return bail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
				size_t words = (nb + 3) / 4; // what the CIGAR needs; the sender's capacity (the reference rounds it up to a power of two, align.c:199-207) is kept only while it is plausible
				if ((size_t)hd.capacity >= words && (size_t)hd.capacity <= 2 * words + 16) words = hd.capacity;
				r[j].p = (ref::Extra *)calloc(words, 4);
				if (!r[j].p) return bail(MM2AMD_ENOMEM, "[mm2amd] mm2amd_unpack_regs: out of memory");
				memcpy(r[j].p, o, nb), o += nb;
				r[j].p->capacity = (uint32_t)words;
			}
		}
	}
	return 0;
}

int mm_gpu_map_batch(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, int *n_reg, void **reg, int *rep_len, int *frag_gap)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	std::lock_guard<std::mutex> lk_map(g_map_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_map_batch called before mm_gpu_init");
	if (n_frag < 0 || (n_frag > 0 && (!seg_off || !n_seg || !seq_ || !n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_map_batch: bad arguments");
	MapContext &c = *g_ctx;
	try {
		StagedBatch *bt;
		{ // hand-over and take-over in one go (a batch staged but not yet mapped is replaced)
			std::lock_guard<std::mutex> lk_st(c.stage_mu);
			c.pending = false;
			bt = &c.batch[1 - c.cur];
			if (int rc = collect_views(n_frag, seg_off, n_seg, seq_, c.opt, c.n_threads, bt->views, bt->slots, bt->flipped)) return rc;
			stage_replicas(c, *bt);
			c.pending = true;
			take_staged(c);
		}
		std::vector<ReadResult> out;
		run_replicas(c, *bt, out);
		if (inject_batch_failure()) { free_results(out); throw std::runtime_error("[mm2amd] MM2AMD_INJECT_BATCH_FAILURE: this batch is reported as failed (test knob)"); }
		hand_over(bt->slots, out, n_reg, reg, rep_len, frag_gap);
		c.has_current = false; // the views point into the caller's buffers, which this call does not own beyond its return
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EHIP, e.what());
	}
}

// ---- batch-of-one calls with the reference's own signatures (mm_map, map.c:380-392; mm_map_frag, map.c:227-378 / :394-397): what a
// binding that maps read by read calls (python/cmappy.h:74-102 wraps mm_map).  They go through mm_gpu_map_batch with one fragment:
// correct and convenient, slow (a GPU pipeline pass per read) -- batch wherever the caller can.  The context is (re)built when
// (mi, opt) differ from the live one.  b, when given, receives rep_len / frag_gap like the reference's mm_tbuf_t (minimap.h:207-210).
struct TbufView { void *km; int rep_len, frag_gap; };

// These calls use the ONE process-wide context: g_single_mu is held from the check of (mi, opt) to the end of the mapping, so two threads
// that pass different options or indexes cannot map with each other's context (they rebuild it in turn, which costs seconds: batch-of-one
// callers should stick to one (mi, opt)).  The context is matched on the index's address AND on fields of it (k, w, flag, n_seq, n_alt,
// the sequence arrays), so an index freed and another allocated at the same address, or one changed in place, is not mistaken for it.
static std::mutex g_single_mu;

static int ensure_context_for(const void *mi, const void *opt)
{
	{
		std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
		const ref::Idx *rmi = (const ref::Idx *)mi;
		if (g_ctx && g_ctx->mi_ptr == mi && memcmp(&g_ctx->opt, opt, sizeof(ref::MapOpt)) == 0 && g_ctx->mi_stamp[0] == g_stamp_hash(rmi) &&
		    g_ctx->mi_stamp[1] == (uint64_t)(uintptr_t)rmi->S && g_ctx->mi_stamp[2] == (uint64_t)(uintptr_t)rmi->seq && g_ctx->mi_stamp[3] == (uint64_t)rmi->n_alt) return 0;
	}
	return mm_gpu_init(mi, opt, 0);
}

void mm_gpu_map_frag(const void *mi, int n_segs, const int *qlens, const char **seqs, int *n_regs, void **regs, void *b, const void *opt, const char *qname)
{
	for (int s = 0; s < n_segs; ++s) n_regs[s] = 0, regs[s] = nullptr;
	if (!mi || !opt || n_segs < 1 || n_segs > 2) { capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_map_frag: one or two segments, non-null index and options"); return; }
	std::lock_guard<std::mutex> lk_single(g_single_mu);
	if (ensure_context_for(mi, opt) != 0) return;
	ref::Bseq1 rec[2];
	for (int s = 0; s < n_segs; ++s) {
		rec[s].l_seq = qlens[s], rec[s].rid = s;
		rec[s].name = const_cast<char *>(qname), rec[s].seq = const_cast<char *>(seqs[s]), rec[s].qual = rec[s].comment = nullptr;
	}
	const int seg_off = 0;
	int rep_len[2] = { 0, 0 }, frag_gap[2] = { 0, 0 };
	if (mm_gpu_map_batch(1, &seg_off, &n_segs, rec, n_regs, regs, rep_len, frag_gap) != 0) {
		for (int s = 0; s < n_segs; ++s) n_regs[s] = 0, regs[s] = nullptr;
		return;
	}
	if (b) ((TbufView *)b)->rep_len = rep_len[0], ((TbufView *)b)->frag_gap = frag_gap[0];
}

// SURVEY.md 8(b)(2) as written: the batch call that names its index and options.  The context for (mi, *opt) is built on first use and rebuilt
// when either changes (as mm_gpu_map does); then this is mm_gpu_map_batch.
int mm_gpu_map_batch_with(const void *mi, const void *opt, int n_frag, const int *seg_off, const int *n_seg, const void *seq, int *n_reg, void **reg, int *rep_len, int *frag_gap)
{
	if (!mi || !opt) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_map_batch_with: non-null index and options");
	//  the lock is held until the batch is mapped, as in mm_gpu_map_frag: a second thread naming another (mi, opt) rebuilds the ONE context only
	// after this batch has come back -- otherwise this batch could be mapped against the other thread's index and options
	std::lock_guard<std::mutex> lk_single(g_single_mu);
	if (int rc = ensure_context_for(mi, opt)) return rc;
	return mm_gpu_map_batch(n_frag, seg_off, n_seg, seq, n_reg, reg, rep_len, frag_gap);
}

void *mm_gpu_map(const void *mi, int qlen, const char *seq, int *n_regs, void *b, const void *opt, const char *qname)
{
	void *regs = nullptr;
	mm_gpu_map_frag(mi, 1, &qlen, &seq, n_regs, &regs, b, opt, qname);
	return regs;
}

void mm_gpu_destroy(void)
{
	std::unique_lock<std::shared_mutex> lk(g_ctx_mu);
	g_ctx.reset();
}

// for bindings with several owners (two Python Aligner objects): tears the context down only if it is still the one the
// caller installed; returns 1 when it did
int mm_gpu_destroy_if(uint64_t generation)
{
	std::unique_lock<std::shared_mutex> lk(g_ctx_mu);
	if (!g_ctx || g_ctx->generation != generation) return 0;
	g_ctx.reset();
	return 1;
}

const char *mm2amd_backend_name(void) { return backend_name(); }
int mm2amd_format_fraction(double v, char *buf) { return buf ? format_fraction_for_test(v, buf) : 0; }

int mm2amd_last_stats(double *v, int n)
{
	std::shared_lock<std::shared_mutex> lk(g_ctx_mu);
	if (!g_ctx) return 0;
	std::lock_guard<std::mutex> lk_stats(g_ctx->stats_mu);
	const MapperStats &s = g_ctx->stats;
	const double a[] = { s.t_seed_chain, s.t_host_pre, s.t_plan, s.t_ksw, s.t_consume, s.t_finish, (double)s.n_jobs, (double)s.n_rounds, s.dp_cells,
	                     (double)mm2amd_alloc_counter(0), (double)mm2amd_alloc_counter(1), (double)mm2amd_alloc_counter(2),
	                     s.c_seed_chain, s.c_host_pre, s.c_plan, s.c_ksw, s.c_consume, s.c_finish, (double)s.n_long_join_dev, (double)s.n_long_join_host,
	                     s.d_seed_chain, s.d_host_pre, s.d_plan, s.d_ksw, s.d_consume, s.d_finish, (double)s.n_early_sub,
	                     (double)s.n_region_reads_dev, (double)s.n_region_reads_host,
	                     (double)mm2amd_alloc_counter(3), (double)mm2amd_alloc_counter(4), (double)mm2amd_alloc_counter(5), (double)mm2amd_alloc_counter(6),
	                     (double)mm2amd_alloc_counter(7), (double)mm2amd_alloc_counter(8), (double)mm2amd_alloc_counter(9), (double)mm2amd_alloc_counter(10),
	                     (double)mm2amd_alloc_counter(11), (double)mm2amd_alloc_counter(12) }; // (the arenas behind the work buffers: bytes held in chunks, device / pinned; bytes handed out of them) // (process CPU seconds while a lane was in the stage: lanes overlap, so these attribute, they do not add up)
	int k = 0;
	for (; k < n && k < (int)(sizeof a / sizeof a[0]); ++k) v[k] = a[k];
	return k;
}

} // extern "C"

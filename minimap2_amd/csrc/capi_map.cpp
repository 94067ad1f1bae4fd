// Drop-in boundary: the entry points a minimap2 build calls instead of kt_for(worker_for) (map.c:576).
// Declared in include/mm2amd.h; the backend (HIP in the product) is supplied by make_backend().
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include "../../include/mm2amd.h"
#include "mapper.hpp"
#include "format.hpp"
#include "index_handle.hpp"
#include "threads.hpp"

namespace mm2amd {
Backend *make_backend(const FlatIndex &fi, void *device_tables, int n_threads); // backend_hip.cpp in the product
const char *backend_name();
void capi_set_error(const std::string &msg);              // capi_common.cpp
int capi_fail(int code, const std::string &msg);
}

using namespace mm2amd;

extern "C" long long mm2amd_alloc_counter(int which); // device allocations, pinned allocations, nanoseconds spent in them

namespace {
// Where the results of one mapper fragment go: reads o .. o+n_out-1 of the caller's arrays; flip_len[j] >= 0 when read o+j was
// mapped reverse-complemented and its hits have to be turned back (length of that read), see hand_over().
struct OutSlot { int o, n_out, flip_len[2]; int weak = 0, len0 = 0; }; // weak: 1 / 2 = first / second mate of a pair mapped separately and paired afterwards (MM_F_WEAK_PAIRING)

struct MapContext {
	FlatIndex fi_own;                 // index flattened from a reference mm_idx_t (mm_gpu_init)
	const FlatIndex *fi = nullptr;    // the index in use (fi_own, or the one inside an mm2amd_index_t)
	ref::MapOpt opt;
	std::vector<ReadView> staged;
	std::vector<OutSlot> staged_slots;
	std::vector<std::string> staged_flipped;
	bool has_staged = false;
	int n_threads = 1;
	std::unique_ptr<Backend> be;
	std::unique_ptr<Mapper> mapper;
};
std::mutex g_mu;
std::unique_ptr<MapContext> g_ctx;
}

extern "C" {

int mm_gpu_init(const void *mi, const void *opt, int n_threads)
{
	if (!mi || !opt) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_init: null index or options");
	std::lock_guard<std::mutex> lk(g_mu);
	try {
		std::unique_ptr<MapContext> c(new MapContext);
		c->opt = *(const ref::MapOpt *)opt;
		c->fi_own.from_reference((const ref::Idx *)mi);
		c->fi = &c->fi_own;
		if (n_threads <= 0) n_threads = std::min(64, (int)std::thread::hardware_concurrency()); // the host stages stop scaling (and start contending) beyond ~64 threads per GPU
		c->be.reset(make_backend(*c->fi, nullptr, n_threads));
		c->mapper.reset(new Mapper(*c->fi, c->opt, *c->be, n_threads));
		c->n_threads = n_threads;
		g_ctx = std::move(c);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		const std::string s = e.what();
		return capi_fail(s.find("no HIP device") != std::string::npos ? MM2AMD_ENODEV : MM2AMD_EHIP, s);
	}
}

int mm_gpu_init_index(const mm2amd_index_t *idx, const void *opt, int n_threads)
{
	if (!idx || !opt) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_init_index: null index or options");
	std::lock_guard<std::mutex> lk(g_mu);
	try {
		std::unique_ptr<MapContext> c(new MapContext);
		c->opt = *(const ref::MapOpt *)opt;
		c->fi = &index_flat((const IndexHandle *)idx);
		if (n_threads <= 0) n_threads = std::min(64, (int)std::thread::hardware_concurrency()); // the host stages stop scaling (and start contending) beyond ~64 threads per GPU
		c->be.reset(make_backend(*c->fi, index_device_tables((const IndexHandle *)idx), n_threads));
		c->mapper.reset(new Mapper(*c->fi, c->opt, *c->be, n_threads));
		c->n_threads = n_threads;
		g_ctx = std::move(c);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		const std::string s = e.what();
		return capi_fail(s.find("no HIP device") != std::string::npos ? MM2AMD_ENODEV : MM2AMD_EHIP, s);
	}
}

// mm_revcomp_bseq (mmpriv.h) on a copy: complement table of bseq.c:11-28 (IUPAC codes, case kept, other bytes unchanged)
static void revcomp_into(const char *seq, int len, std::string &out)
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
	out.resize(len);
	for (int i = 0; i < len; ++i) out[len - 1 - i] = (char)tab.t[(unsigned char)seq[i]];
}

// The fragments of a batch as the mapper sees them.  A two-segment fragment (paired-end reads) is handed over in mapping
// orientation: worker_for reverse-complements a mate in place according to pe_ori before mapping and back afterwards
// (map.c:436-442, 457-473); here the flipped copy lives in `flipped` and the caller's buffers are left alone.  With
// MM_F_INDEPEND_SEG the two reads of a pair are mapped as two single reads (map.c:443-448), still in flipped orientation.
static int collect_views(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, const ref::MapOpt &opt, std::vector<ReadView> &reads,
                         std::vector<OutSlot> &slots, std::vector<std::string> &flipped)
{
	const ref::Bseq1 *seq = (const ref::Bseq1 *)seq_;
	const int pe_ori = opt.pe_ori;
	const bool independent = (opt.flag & ref::F_INDEPEND_SEG) != 0;
	reads.clear(), slots.clear(), flipped.clear();
	size_t n_flip = 0;
	for (int i = 0; i < n_frag; ++i) {
		if (n_seg[i] != 1 && n_seg[i] != 2) return capi_fail(MM2AMD_EINVAL, "[mm2amd] fragments of more than two segments are not implemented");
		if (n_seg[i] == 2) n_flip += (pe_ori >> 1 & 1) + (pe_ori & 1);
	}
	flipped.resize(n_flip); // sized first: the views point into it
	n_flip = 0;
	for (int i = 0; i < n_frag; ++i) {
		const int o = seg_off[i];
		const ref::Bseq1 &s = seq[o];
		ReadView v;
		OutSlot sl = { o, 1, { -1, -1 } };
		v.seq = s.seq, v.len = s.l_seq, v.name = s.name;
		if (n_seg[i] == 2) {
			const ref::Bseq1 &s2 = seq[o + 1];
			const char *q2 = s2.seq;
			if (pe_ori >> 1 & 1) { revcomp_into(s.seq, s.l_seq, flipped[n_flip]); v.seq = flipped[n_flip++].data(); sl.flip_len[0] = s.l_seq; }
			if (pe_ori & 1) { revcomp_into(s2.seq, s2.l_seq, flipped[n_flip]); q2 = flipped[n_flip++].data(); sl.flip_len[1] = s2.l_seq; }
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

int mm_gpu_batch_stage(int n_frag, const int *seg_off, const int *n_seg, const void *seq_)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_batch_stage called before mm_gpu_init");
	if (n_frag < 0 || (n_frag > 0 && (!seg_off || !n_seg || !seq_))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_batch_stage: bad arguments");
	try {
		g_ctx->has_staged = false;
		if (int rc = collect_views(n_frag, seg_off, n_seg, seq_, g_ctx->opt, g_ctx->staged, g_ctx->staged_slots, g_ctx->staged_flipped)) return rc;
		g_ctx->mapper->stage(g_ctx->staged);
		g_ctx->has_staged = true;
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EHIP, e.what());
	}
}

int mm_gpu_map_staged(int *n_reg, void **reg, int *rep_len, int *frag_gap)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_ctx || !g_ctx->has_staged) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_map_staged: no staged batch");
	if (!n_reg || !reg) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_map_staged: bad arguments");
	try {
		std::vector<ReadResult> out;
		g_ctx->mapper->run(out);
		hand_over(g_ctx->staged_slots, out, n_reg, reg, rep_len, frag_gap);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EHIP, e.what());
	}
}

int mm_gpu_format_batch(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, const int *n_reg, void *const *reg, const int *rep_len, char **out, size_t *out_len)
{
	std::lock_guard<std::mutex> lk(g_mu);
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

void mm2amd_free_regs(int n_frag, int *n_reg, void **reg)
{
	if (!n_reg || !reg) return;
	for (int i = 0; i < n_frag; ++i) {
		ref::Reg1 *r = (ref::Reg1 *)reg[i];
		for (int j = 0; j < n_reg[i]; ++j) free(r[j].p);
		free(r);
		reg[i] = nullptr, n_reg[i] = 0;
	}
}

// ---- hit records as one flat byte payload: the unit of the multi-GPU gather ----
// per fragment: int32 n_reg, then per hit the 80-byte mm_reg1_t (pointer field zeroed), a uint32 "has extra", and when
// set the 28-byte mm_extra_t header followed by n_cigar uint32 CIGAR words.
int64_t mm2amd_pack_regs(int n_frag, const int *n_reg, void *const *reg, uint8_t *buf, int64_t cap)
{
	if (n_frag < 0 || (n_frag > 0 && (!n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_pack_regs: bad arguments");
	// sizes first (prefix sums give every fragment its slot), then the copies, both on the pool threads
	const int nt = g_ctx ? g_ctx->n_threads : 1;
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
	if (!buf || n_frag < 0 || (n_frag > 0 && (!n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: bad arguments");
	const uint8_t *o = buf, *end = buf + size;
	for (int i = 0; i < n_frag; ++i) {
		int32_t n;
		if (end - o < 4) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
		memcpy(&n, o, 4), o += 4;
		n_reg[i] = n, reg[i] = nullptr;
		if (n <= 0) continue;
		ref::Reg1 *r = (ref::Reg1 *)calloc(n, sizeof(ref::Reg1));
		reg[i] = r;
		for (int j = 0; j < n; ++j) {
			uint32_t has;
			if (end - o < (int64_t)sizeof(ref::Reg1) + 4) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
			memcpy(&r[j], o, sizeof(ref::Reg1)), o += sizeof(ref::Reg1);
			memcpy(&has, o, 4), o += 4;
			r[j].p = nullptr;
			if (has) {
				ref::Extra hd;
				if (end - o < (int64_t)sizeof hd) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
				memcpy(&hd, o, sizeof hd);
				const size_t nb = sizeof(ref::Extra) + 4ull * hd.n_cigar;
				if ((size_t)(end - o) < nb) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_unpack_regs: truncated payload");
				size_t words = hd.capacity;
				if (words * 4 < nb) words = (nb + 3) / 4;
				r[j].p = (ref::Extra *)calloc(words, 4);
				memcpy(r[j].p, o, nb), o += nb;
				r[j].p->capacity = (uint32_t)words;
			}
		}
	}
	return 0;
}

int mm_gpu_map_batch(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, int *n_reg, void **reg, int *rep_len, int *frag_gap)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_map_batch called before mm_gpu_init");
	if (n_frag < 0 || (n_frag > 0 && (!seg_off || !n_seg || !seq_ || !n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_map_batch: bad arguments");
	try {
		std::vector<ReadView> reads;
		std::vector<OutSlot> slots;
		std::vector<std::string> flipped;
		if (int rc = collect_views(n_frag, seg_off, n_seg, seq_, g_ctx->opt, reads, slots, flipped)) return rc;
		std::vector<ReadResult> out;
		g_ctx->mapper->map_batch(reads, out);
		hand_over(slots, out, n_reg, reg, rep_len, frag_gap);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EHIP, e.what());
	}
}

void mm_gpu_destroy(void)
{
	std::lock_guard<std::mutex> lk(g_mu);
	g_ctx.reset();
}

const char *mm2amd_backend_name(void) { return backend_name(); }

int mm2amd_last_stats(double *v, int n)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_ctx) return 0;
	const MapperStats &s = g_ctx->mapper->stats;
	const double a[] = { s.t_seed_chain, s.t_host_pre, s.t_plan, s.t_ksw, s.t_consume, s.t_finish, (double)s.n_jobs, (double)s.n_rounds, s.dp_cells,
	                     (double)mm2amd_alloc_counter(0), (double)mm2amd_alloc_counter(1), (double)mm2amd_alloc_counter(2) };
	int k = 0;
	for (; k < n && k < (int)(sizeof a / sizeof a[0]); ++k) v[k] = a[k];
	return k;
}

} // extern "C"

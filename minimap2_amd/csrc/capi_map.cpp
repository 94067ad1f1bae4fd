// Drop-in boundary: the entry points a minimap2 build calls instead of kt_for(worker_for) (map.c:576).
// Declared in include/mm2amd.h; the backend (HIP in the product) is supplied by make_backend().
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include "../../include/mm2amd.h"
#include "mapper.hpp"

namespace mm2amd {
Backend *make_backend(const FlatIndex &fi, int device); // backend_hip.cpp in the product
const char *backend_name();
void capi_set_error(const std::string &msg);              // capi_common.cpp
int capi_fail(int code, const std::string &msg);
}

using namespace mm2amd;

namespace {
struct MapContext {
	FlatIndex fi;
	ref::MapOpt opt;
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
		c->fi.from_reference((const ref::Idx *)mi);
		if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
		c->be.reset(make_backend(c->fi, -1));
		c->mapper.reset(new Mapper(c->fi, c->opt, *c->be, n_threads));
		g_ctx = std::move(c);
		return 0;
	} catch (const std::invalid_argument &e) {
		return capi_fail(MM2AMD_EINVAL, e.what());
	} catch (const std::exception &e) {
		const std::string s = e.what();
		return capi_fail(s.find("no HIP device") != std::string::npos ? MM2AMD_ENODEV : MM2AMD_EHIP, s);
	}
}

int mm_gpu_map_batch(int n_frag, const int *seg_off, const int *n_seg, const void *seq_, int *n_reg, void **reg, int *rep_len, int *frag_gap)
{
	std::lock_guard<std::mutex> lk(g_mu);
	if (!g_ctx) return capi_fail(MM2AMD_ESTATE, "[mm2amd] mm_gpu_map_batch called before mm_gpu_init");
	if (n_frag < 0 || (n_frag > 0 && (!seg_off || !n_seg || !seq_ || !n_reg || !reg))) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm_gpu_map_batch: bad arguments");
	const ref::Bseq1 *seq = (const ref::Bseq1 *)seq_;
	try {
		std::vector<ReadView> reads(n_frag);
		for (int i = 0; i < n_frag; ++i) {
			if (n_seg[i] != 1) return capi_fail(MM2AMD_EINVAL, "[mm2amd] multi-segment fragments (paired-end) are not implemented");
			const ref::Bseq1 &s = seq[seg_off[i]];
			reads[i].seq = s.seq, reads[i].len = s.l_seq, reads[i].name = s.name;
		}
		std::vector<ReadResult> out;
		g_ctx->mapper->map_batch(reads, out);
		for (int i = 0; i < n_frag; ++i) {
			const int o = seg_off[i];
			const size_t n = out[i].regs.size();
			n_reg[o] = (int)n;
			reg[o] = nullptr;
			if (n) { // handed over as one libc block, like the reference's realloc'd array (map.c:340)
				reg[o] = malloc(n * sizeof(ref::Reg1));
				memcpy(reg[o], out[i].regs.data(), n * sizeof(ref::Reg1));
			}
			if (rep_len) rep_len[o] = out[i].rep_len;
			if (frag_gap) frag_gap[o] = out[i].frag_gap;
		}
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
	const double a[] = { s.t_seed_chain, s.t_host_pre, s.t_plan, s.t_ksw, s.t_consume, s.t_finish, (double)s.n_jobs, (double)s.n_rounds, s.dp_cells };
	int k = 0;
	for (; k < n && k < (int)(sizeof a / sizeof a[0]); ++k) v[k] = a[k];
	return k;
}

} // extern "C"

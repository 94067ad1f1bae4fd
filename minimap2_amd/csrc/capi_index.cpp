// Index and option entry points of the C ABI (include/mm2amd.h): in-memory index construction on the device
// (mm_idx_str, index.c:421-470) and the option presets (options.c).
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/mm2amd.h"
#include "device_ctx.hpp"
#include "index_build.hpp"
#include "index_handle.hpp"
#include "options.hpp"

namespace mm2amd {
int capi_fail(int code, const std::string &msg);
struct IndexHandle {
	FlatIndex fi;
	DeviceIndexTables T;
	int device = 0;
};
const FlatIndex &index_flat(const IndexHandle *h) { return h->fi; }
void *index_device_tables(const IndexHandle *h) { return (void *)&h->T; }
int index_device(const IndexHandle *h) { return h->device; }
}

using namespace mm2amd;

namespace {
int32_t handle_max_occ(const void *h, float f) { return ((const IndexHandle *)h)->T.cal_max_occ(f); }
}

extern "C" {

mm2amd_index_t *mm2amd_idx_str(int w, int k, int is_hpc, int bucket_bits, int n, const char **seq, const char **name)
{
	(void)bucket_bits; // the flat device table sizes its direct-address level from the number of distinct minimizers
	if (n <= 0 || !seq) { capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_idx_str: need n > 0 sequences"); return nullptr; }
	try {
		DeviceCtx &dc = device_ctx();
		std::lock_guard<std::mutex> lk(dc.mu);
		ensure_device(dc);
		std::unique_ptr<IndexHandle> h(new IndexHandle);
		h->device = dc.device_id;
		std::vector<uint64_t> lens(n);
		for (int i = 0; i < n; ++i) lens[i] = seq[i] ? strlen(seq[i]) : 0;
		int flag = 0;
		if (is_hpc) flag |= ref::I_HPC;
		if (!name) flag |= ref::I_NO_NAME;
		DeviceIndexBuilder::build(h->fi, h->T, k, w, flag, n, seq, lens.data(), name, dc.stream);
		return (mm2amd_index_t *)h.release();
	} catch (const HipError &e) {
		const std::string s = e.what();
		capi_fail(s.find("no HIP device") != std::string::npos ? MM2AMD_ENODEV : MM2AMD_EHIP, s);
	} catch (const std::exception &e) {
		capi_fail(MM2AMD_EINVAL, e.what());
	}
	return nullptr;
}

void mm2amd_idx_destroy(mm2amd_index_t *idx) { delete (IndexHandle *)idx; }

int mm2amd_idx_stat(const mm2amd_index_t *idx, int *k, int *w, int *flag, uint32_t *n_seq, uint64_t *sum_len, uint64_t *n_distinct, uint64_t *n_minimizers)
{
	if (!idx) return capi_fail(MM2AMD_EINVAL, "[mm2amd] null index");
	const IndexHandle *h = (const IndexHandle *)idx;
	if (k) *k = h->fi.k;
	if (w) *w = h->fi.w;
	if (flag) *flag = h->fi.flag;
	if (n_seq) *n_seq = h->fi.n_seq;
	if (sum_len) *sum_len = h->fi.sum_len;
	if (n_distinct) *n_distinct = h->T.n_keys;
	if (n_minimizers) *n_minimizers = h->T.n_pos;
	return 0;
}

int32_t mm2amd_idx_cal_max_occ(const mm2amd_index_t *idx, float f)
{
	if (!idx) return capi_fail(MM2AMD_EINVAL, "[mm2amd] null index");
	try { return handle_max_occ(idx, f); } catch (const std::exception &e) { return capi_fail(MM2AMD_EINVAL, e.what()); }
}

// Copies the flat tables to host memory (any pointer may be null to skip it); sizes come from mm2amd_idx_stat and
// mm2amd_idx_table_shape.  Used by tests to compare the device-built index with the reference's mm_idx_t.
int mm2amd_idx_table_shape(const mm2amd_index_t *idx, int *bucket_bits, int *key_shift)
{
	if (!idx) return capi_fail(MM2AMD_EINVAL, "[mm2amd] null index");
	const IndexHandle *h = (const IndexHandle *)idx;
	if (bucket_bits) *bucket_bits = h->T.bucket_bits;
	if (key_shift) *key_shift = h->T.key_shift;
	return 0;
}

int mm2amd_idx_export(const mm2amd_index_t *idx, uint32_t *bucket_start, uint64_t *keys, uint32_t *val_off, uint64_t *pos, uint32_t *S)
{
	if (!idx) return capi_fail(MM2AMD_EINVAL, "[mm2amd] null index");
	const IndexHandle *h = (const IndexHandle *)idx;
	try {
		DeviceCtx &dc = device_ctx(h->device);
		std::lock_guard<std::mutex> lk(dc.mu);
		ensure_device(dc);
		if (bucket_start) HIP_CHECK(hipMemcpy(bucket_start, h->T.bucket_start.p, ((1ull << h->T.bucket_bits) + 1) * 4, hipMemcpyDeviceToHost));
		if (keys && h->T.n_keys) HIP_CHECK(hipMemcpy(keys, h->T.keys.p, h->T.n_keys * 8, hipMemcpyDeviceToHost));
		if (val_off) HIP_CHECK(hipMemcpy(val_off, h->T.val_off.p, (h->T.n_keys + 1) * 4, hipMemcpyDeviceToHost));
		if (pos && h->T.n_pos) HIP_CHECK(hipMemcpy(pos, h->T.pos.p, h->T.n_pos * 8, hipMemcpyDeviceToHost));
		if (S) memcpy(S, h->fi.S, (h->fi.sum_len + 7) / 8 * 4);
		return 0;
	} catch (const std::exception &e) {
		return capi_fail(MM2AMD_EHIP, e.what());
	}
}

void mm2amd_idxopt_init(void *io) { idxopt_init((ref::IdxOpt *)io); }
void mm2amd_mapopt_init(void *mo) { mapopt_init((ref::MapOpt *)mo); }
int mm2amd_set_opt(const char *preset, void *io, void *mo)
{
	if (!io || !mo) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_set_opt: null option struct");
	return set_opt(preset, (ref::IdxOpt *)io, (ref::MapOpt *)mo);
}
int mm2amd_check_opt(const void *io, const void *mo)
{
	std::string why;
	const int rc = check_opt((const ref::IdxOpt *)io, (const ref::MapOpt *)mo, &why);
	if (rc != 0) capi_fail(rc, "[mm2amd] " + why);
	return rc;
}
int mm2amd_mapopt_update(void *mo, const mm2amd_index_t *idx)
{
	if (!mo || !idx) return capi_fail(MM2AMD_EINVAL, "[mm2amd] mm2amd_mapopt_update: null argument");
	try { mapopt_update((ref::MapOpt *)mo, handle_max_occ, idx); return 0; } catch (const std::exception &e) { return capi_fail(MM2AMD_EINVAL, e.what()); }
}

} // extern "C"

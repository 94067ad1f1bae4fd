/* mm2amd.h -- C ABI of libmm2amd.so, the MI355X-native seed-chain-extend engine behind minimap2's API.
 *
 * Everything here is plain C: pointers, sizes, PODs.  Each entry point names the reference interface it
 * replaces (file:line under lh3/minimap2 v2.30).  INTEGRATION.md shows the binding a minimap2 maintainer
 * would add.  All functions return 0 on success and a negative MM2AMD_E* code on failure; the text of the
 * last failure on the calling thread is available from mm2amd_last_error().  There is NO CPU fallback: if
 * no gfx950 device is usable, every compute entry point fails with MM2AMD_ENODEV.
 */
#ifndef MM2AMD_H
#define MM2AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MM2AMD_EINVAL  (-1)   /* bad argument */
#define MM2AMD_ENODEV  (-2)   /* no usable HIP device / kernel image */
#define MM2AMD_EHIP    (-3)   /* HIP runtime error */
#define MM2AMD_ENOMEM  (-4)   /* a caller-provided pool was too small */
#define MM2AMD_ESTATE  (-5)   /* called in the wrong state (e.g. map before init) */

const char *mm2amd_last_error(void);
int mm2amd_version(void);                 /* ABI version, currently 1 */
int mm2amd_device_count(void);            /* number of visible HIP devices, or negative error */

/* ------------------------------------------------------------------------------------------------
 * Kernel-level entry points: batched forms of the reference's per-call kernels.
 * ------------------------------------------------------------------------------------------------ */

/* One extension/global DP problem; same argument meaning as ksw_extd2_sse (ksw2.h:72-73). */
typedef struct {
	const uint8_t *query;    /* qlen codes in 0..m-1 (host memory) */
	const uint8_t *target;   /* tlen codes in 0..m-1 (host memory) */
	int32_t qlen, tlen;
	int32_t w;               /* band width, <0 to disable */
	int32_t zdrop, end_bonus;
	int32_t flag;            /* KSW_EZ_* bits of ksw2.h:8-19 */
} mm2amd_ksw_job_t;

/* Result of one DP problem; field meaning as ksw_extz_t (ksw2.h:34-43).  The CIGAR (BAM encoding,
 * len<<4|op) is at cigar_pool[cigar_off .. cigar_off+n_cigar). */
typedef struct {
	int32_t max, zdropped;
	int32_t max_q, max_t;
	int32_t mqe, mqe_t;
	int32_t mte, mte_q;
	int32_t score;
	int32_t n_cigar;
	int32_t reach_end;
	uint32_t cigar_off;
} mm2amd_ksw_res_t;

/* Batched ksw_extd2_sse (ksw2_extd2_sse.c:34): dual-affine gap cost, results bit-identical to the
 * reference for every job, including jobs whose band binds.  cigar_pool must hold at least
 * sum(qlen+tlen) entries over the batch (MM2AMD_ENOMEM otherwise). */
int mm2amd_ksw_extd2_batch(int n_jobs, const mm2amd_ksw_job_t *jobs, int8_t m, const int8_t *mat,
                           int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2,
                           mm2amd_ksw_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap);

/* ------------------------------------------------------------------------------------------------
 * Drop-in boundary: the batched replacement of kt_for(n_threads, worker_for, step, n_frag) (map.c:576).
 *
 * The pointer arguments are the reference's own objects (their layouts are mirrored in
 * minimap2_amd/csrc/abi_ref.hpp): mi = const mm_idx_t* (minimap.h:88-100), opt = const mm_mapopt_t*
 * (minimap.h:136-192), seq = const mm_bseq1_t* (bseq.h:14-17), reg = mm_reg1_t** (minimap.h:112-127).
 * When this header is included after minimap.h the real types are used in the prototypes.
 * ------------------------------------------------------------------------------------------------ */
#ifdef MINIMAP2_H
typedef mm_idx_t mm2amd_idx_t; typedef mm_mapopt_t mm2amd_mapopt_t; typedef mm_reg1_t mm2amd_reg1_t;
struct mm2amd_bseq1_s; /* mm_bseq1_t lives in bseq.h; pass it as-is */
#define MM2AMD_BSEQ_PTR const void *
#define MM2AMD_REG_PP   void **
#else
typedef void mm2amd_idx_t; typedef void mm2amd_mapopt_t;
#define MM2AMD_BSEQ_PTR const void *
#define MM2AMD_REG_PP   void **
#endif

/* Call once per index part after mm_mapopt_update() (main.c:465): builds the device mirror of the index
 * (flat minimizer table + 4-bit packed reference) and captures the mapping options.  n_threads sizes the host
 * worker pool (<=0: all hardware threads).  Replaces nothing in the reference; it is the set-up the GPU path needs. */
int mm_gpu_init(const mm2amd_idx_t *mi, const mm2amd_mapopt_t *opt, int n_threads);

/* Same contract as calling worker_for(step, i, tid) for i in [0, n_frag) (map.c:425-474): for fragment i with
 * segments seq[seg_off[i] .. seg_off[i]+n_seg[i]) fills n_reg[], reg[] (libc-allocated, caller frees reg[k] and each
 * reg[k][j].p), rep_len[] and frag_gap[] at the segment's index.  Output order == input order. */
int mm_gpu_map_batch(int n_frag, const int *seg_off, const int *n_seg, MM2AMD_BSEQ_PTR seq,
                     int *n_reg, MM2AMD_REG_PP reg, int *rep_len, int *frag_gap);

/* Releases the device mirror; call before mm_idx_destroy (main.c:501). */
void mm_gpu_destroy(void);

const char *mm2amd_backend_name(void);           /* "hip:gfx950" in the product library */
int mm2amd_last_stats(double *v, int n);         /* per-stage wall times of the last batch (diagnostics) */

#ifdef __cplusplus
}
#endif
#endif

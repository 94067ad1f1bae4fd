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

#ifdef __cplusplus
}
#endif
#endif

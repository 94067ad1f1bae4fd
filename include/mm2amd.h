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
int mm2amd_host_cpus(void);                     /* CPUs this process may use: hardware threads capped by the container's CPU quota (cgroup cpu.max); what n_threads <= 0 resolves against */
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

/* Batched mm_update_extra (align.c:254-303, with mm_append_cigar :320-334 and mm_fix_cigar :105-181): a region's window CIGARs are stitched,
 * indels left-aligned, I/D clusters merged, empty operations and a leading gap dropped, and the block / match lengths, ambiguous bases and
 * the best-scoring segment (dp_max) counted -- region_finish_kernel, one wavefront per region.  query / target: nt4 codes (0-3, 4 = N) of
 * the aligned stretches; piece[i] / piece_len[i]: the windows' CIGARs in alignment order.  cigar_pool must hold the sum of all piece lengths
 * (MM2AMD_ENOMEM otherwise).  Results are those of the reference for every input the reference accepts (its asserts hold: the operations
 * cover exactly qlen and tlen); n_cigar < 0 marks a region whose operations do not.  A region of more than MM2AMD_FIN_MAX_OPS operations (the
 * kernel stages a region's CIGAR in LDS) is refused with MM2AMD_EINVAL: the mapper finishes such regions with its host routine. */
#define MM2AMD_FIN_MAX_OPS 7168
typedef struct {
	const uint8_t *query, *target;
	int32_t qlen, tlen;
	int32_t n_pieces;
	const uint32_t *const *piece;
	const int32_t *piece_len;
} mm2amd_fin_job_t;
typedef struct { int32_t n_cigar, blen, mlen, n_ambi, dp_max, qshift, tshift, is_spliced; uint32_t cigar_off; } mm2amd_fin_res_t;
int mm2amd_update_extra_batch(int n_jobs, const mm2amd_fin_job_t *jobs, const int8_t *mat25, int8_t q, int8_t e, int log_gap,
                              mm2amd_fin_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap);

/* The two device-wide primitives of the index build (device_sort.hip), exposed for testing.  mm2amd_sort_pairs_u64: n (key, value) pairs
 * sorted in place by key bits [0, bits), stably -- what radix_sort_128x (ksort.h:101-151, instantiated at sketch.c:13 and called per bucket
 * at index.c:236) yields for pairs whose input order is ascending in the value: rs_hist / rs_chunk_scan / rs_block_offsets / rs_scatter
 * kernels, one LSD pass per 8 bits.  mm2amd_exclusive_sum_u32: out[i] = in[0] + ... + in[i-1] mod 2^32 for i in [0, n], i.e. out holds
 * n + 1 entries (the running offsets of index.c:249-268).  n < 2^32 for both. */
int mm2amd_sort_pairs_u64(uint64_t *keys, uint64_t *vals, uint64_t n, int bits);
int mm2amd_exclusive_sum_u32(const uint32_t *in, uint32_t *out, uint64_t n);

/* Batched ksw_extz2_sse (ksw2_extz2_sse.c:25, ksw2.h:70-71): single-affine gap cost; same contract as above. */
int mm2amd_ksw_extz2_batch(int n_jobs, const mm2amd_ksw_job_t *jobs, int8_t m, const int8_t *mat, int8_t gapo, int8_t gape,
                           mm2amd_ksw_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap);

/* Batched ksw_exts2_sse (ksw2_exts2_sse.c:33, ksw2.h:77-79): splice-aware alignment (intron state on the target, N operations;
 * flag carries KSW_EZ_SPLICE_FOR/REV/FLANK/CMPLX as in the reference; the job's w is ignored: this DP has no band).  Junction
 * annotation (the reference's junc/junc_bonus/junc_pen arguments) is not taken at this boundary: the call equals the
 * reference's with junc == NULL.  Same contract as above otherwise. */
int mm2amd_ksw_exts2_batch(int n_jobs, const mm2amd_ksw_job_t *jobs, int8_t m, const int8_t *mat, int8_t gapo, int8_t gape, int8_t gapo2, int8_t noncan,
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

/* ------------------------------------------------------------------------------------------------
 * Options (options.c).  io = mm_idxopt_t* (minimap.h:130-134), mo = mm_mapopt_t* (minimap.h:136-192).
 * ------------------------------------------------------------------------------------------------ */
void mm2amd_idxopt_init(void *io);                                /* mm_idxopt_init, options.c:5 */
void mm2amd_mapopt_init(void *mo);                                /* mm_mapopt_init, options.c:14 */
int mm2amd_set_opt(const char *preset, void *io, void *mo);       /* mm_set_opt, options.c:91; -1 for an unknown preset */
int mm2amd_check_opt(const void *io, const void *mo);             /* mm_check_opt, options.c:202 (per-read-path subset) */

/* ------------------------------------------------------------------------------------------------
 * Index built on the device from in-memory sequences: the counterpart of mm_idx_str (index.c:421,
 * minimap.h:324).  seq[i] are NUL-terminated; name may be NULL.  Returns NULL on failure.  A lookup in
 * this index yields the same (count, ascending position list) as mm_idx_get on the reference's index.
 * ------------------------------------------------------------------------------------------------ */
typedef struct mm2amd_index_s mm2amd_index_t;
mm2amd_index_t *mm2amd_idx_str(int w, int k, int is_hpc, int bucket_bits, int n, const char **seq, const char **name);
void mm2amd_idx_destroy(mm2amd_index_t *idx);                     /* mm_idx_destroy, index.c:50 */
int mm2amd_idx_stat(const mm2amd_index_t *idx, int *k, int *w, int *flag, uint32_t *n_seq, uint64_t *sum_len,
                    uint64_t *n_distinct, uint64_t *n_minimizers); /* the figures mm_idx_stat prints, index.c:112-134 */
int32_t mm2amd_idx_cal_max_occ(const mm2amd_index_t *idx, float f); /* mm_idx_cal_max_occ, index.c:198 */
int mm2amd_mapopt_update(void *mo, const mm2amd_index_t *idx);    /* mm_mapopt_update, options.c:69 */
int mm2amd_idx_table_shape(const mm2amd_index_t *idx, int *bucket_bits, int *key_shift);
int mm2amd_idx_export(const mm2amd_index_t *idx, uint32_t *bucket_start, uint64_t *keys, uint32_t *val_off, uint64_t *pos, uint32_t *S);

/* Call once per index part after mm_mapopt_update() (main.c:465): builds the device mirror of the index
 * (flat minimizer table + 4-bit packed reference) and captures the mapping options.  n_threads sizes the host
 * worker pool (<=0: all hardware threads).  Replaces nothing in the reference; it is the set-up the GPU path needs. */
int mm_gpu_init(const mm2amd_idx_t *mi, const mm2amd_mapopt_t *opt, int n_threads);

/* Same contract as calling worker_for(step, i, tid) for i in [0, n_frag) (map.c:425-474): for fragment i with
 * segments seq[seg_off[i] .. seg_off[i]+n_seg[i]) fills n_reg[], reg[] (libc-allocated, caller frees reg[k] and each
 * reg[k][j].p), rep_len[] and frag_gap[] at the segment's index.  Output order == input order.
 * n_seg[i] is 1 or 2.  For a read pair the library does what worker_for does around mm_map_frag (map.c:436-473): the mates are
 * mapped in the orientation mm_mapopt_t::pe_ori prescribes (on copies; seq is not modified), jointly, or each on its own with
 * MM_F_INDEPEND_SEG / MM_F_WEAK_PAIRING, paired by mm_pair's rules, and their hits are turned back to the given orientation.
 * mm_gpu_init refuses (MM2AMD_EINVAL) the configurations the library does not handle; see INTEGRATION.md section 2. */
int mm_gpu_map_batch(int n_frag, const int *seg_off, const int *n_seg, MM2AMD_BSEQ_PTR seq,
                     int *n_reg, MM2AMD_REG_PP reg, int *rep_len, int *frag_gap);

/* As mm_gpu_init, for an index built by mm2amd_idx_str (which must outlive the mapper). */
int mm_gpu_init_index(const mm2amd_index_t *idx, const mm2amd_mapopt_t *opt, int n_threads);

/* Several GPUs behind the same hook (the reference has ONE kt_for call per mini-batch, map.c:576, so the
 * dispatcher lives below it): n_gpus replicas, each with a full copy of the index in its device's HBM, map contiguous shares of
 * every batch cut by cumulative bases -- independent shares, no exchange between devices, results straight into the caller's
 * host arrays.  device_ids: HIP ordinals (NULL: 0 .. n_gpus-1); an ordinal may repeat (several replicas on one device: tests on
 * a single-GPU machine).  n_gpus <= 0: $MM2AMD_GPUS, else 1 -- which is what mm_gpu_init / mm_gpu_init_index pass.  n_threads
 * is the host pool of the whole context (<= 0: 64 per GPU, capped by the hardware threads), divided among the replicas. */
int mm_gpu_init_multi(const mm2amd_idx_t *mi, const mm2amd_mapopt_t *opt, int n_threads, int n_gpus, const int *device_ids);
int mm_gpu_init_index_multi(const mm2amd_index_t *idx, const mm2amd_mapopt_t *opt, int n_threads, int n_gpus, const int *device_ids);
int mm_gpu_n_replicas(void);                     /* replicas of the live context (0: none) */

/* SURVEY.md 8(b)(2) as written -- the batch call that names its index and options: the device context for (mi, *opt) is built on first use
 * and rebuilt when either changes (as mm_gpu_map does), then this is mm_gpu_map_batch.  Calls of mm_gpu_map_batch_with, mm_gpu_map and mm_gpu_map_frag
 * are SERIALISED among themselves from the (mi, opt) check to the end of the mapping: there is one context per process, and a thread naming another
 * index or other options rebuilds it only after the batch under way has come back (a rebuild costs seconds -- callers should stick to one pair). */
int mm_gpu_map_batch_with(const mm2amd_idx_t *mi, const mm2amd_mapopt_t *opt, int n_frag, const int *seg_off, const int *n_seg, MM2AMD_BSEQ_PTR seq,
                          int *n_reg, MM2AMD_REG_PP reg, int *rep_len, int *frag_gap);

/* Batch-of-one calls with the reference's signatures: mm_gpu_map for mm_map (map.c:380-392, minimap.h:375-376), mm_gpu_map_frag for
 * mm_map_frag (map.c:227-378, minimap.h:378-379; n_segs 1 or 2).  Results as the reference returns them (libc blocks; NULL / 0 when
 * nothing maps or on failure -- mm2amd_last_error() tells which); b, when not NULL, receives rep_len and frag_gap (mm_tbuf_t,
 * minimap.h:207-210).  The device context is built on first use and rebuilt when (mi, *opt) change.  One GPU pipeline pass per
 * call: the convenience path of a binding that maps read by read (python/cmappy.h:74-102), not the fast path. */
void *mm_gpu_map(const mm2amd_idx_t *mi, int qlen, const char *seq, int *n_regs, void *b, const mm2amd_mapopt_t *opt, const char *qname);
void mm_gpu_map_frag(const mm2amd_idx_t *mi, int n_segs, const int *qlens, const char **seqs, int *n_regs, MM2AMD_REG_PP regs, void *b,
                     const mm2amd_mapopt_t *opt, const char *qname);

/* mm_gpu_map_batch in two halves: mm_gpu_batch_stage copies the batch's sequences to the device (the hand-over the
 * reference's pipeline step 0 makes, map.c:543-575) and returns when they are resident; mm_gpu_map_staged runs the
 * hot path on the staged batch (results placed as mm_gpu_map_batch places them: read seg_off[i] + j of fragment i).  seq, and the
 * seg_off / n_seg the batch was staged with, describe the result arrays; seq must stay valid in between. */
int mm_gpu_batch_stage(int n_frag, const int *seg_off, const int *n_seg, MM2AMD_BSEQ_PTR seq);
int mm_gpu_map_staged(int *n_reg, MM2AMD_REG_PP reg, int *rep_len, int *frag_gap);
/* The pipeline form of the hand-over (kt_pipeline, map.c:541-577: step 0 of batch k+1 runs beside step 1 of batch k): the library keeps two
 * resident batches, so this call may run on another thread WHILE mm_gpu_map_staged maps the previous batch -- pinned-memory packing and H2D
 * then cost no mapping time.  It waits (instead of replacing, as mm_gpu_batch_stage does) while an earlier staged batch has not been taken
 * over by a mapping call yet: every staged batch is mapped exactly once, in staging order.  seq must stay valid until its batch is mapped. */
int mm_gpu_batch_stage_queued(int n_frag, const int *seg_off, const int *n_seg, MM2AMD_BSEQ_PTR seq);
void mm_gpu_batch_discard(void);                 /* drop a staged batch that will not be mapped (a pipeline shutting down) */

/* Host output stage: replaces the record-writing loop of the reference's pipeline step 2
 * (map.c:585-623: mm_write_sam3, format.c:522, or mm_write_paf4, format.c:425, per hit, then mm_err_puts) for one mini-batch.
 * The text is byte-identical to what that loop prints -- SAM or PAF by MM_F_OUT_SAM, cg/cs/ds/MD/ts/SA tags, unmapped records
 * by MM_F_PAF_NO_HIT / MM_F_SAM_HIT_ONLY, secondaries by MM_F_NO_PRINT_2ND -- but produced on the host thread pool.
 * Arguments as for mm_gpu_map_batch (results as it returned them); *out receives ONE malloc'd block of '\n'-terminated
 * records in input order (free() it), *out_len its length.  Fragments of one or two segments (mate fields of mm_write_sam3,
 * /1 /2 names of mm_write_paf4); no RG tag; uses the index and
 * options given to mm_gpu_init. */
int mm_gpu_format_batch(int n_frag, const int *seg_off, const int *n_seg, MM2AMD_BSEQ_PTR seq, const int *n_reg, void *const *reg,
                        const int *rep_len, char **out, size_t *out_len);
/* The same text in a buffer the library owns and reuses: *out stays valid until the next mm_gpu_format_batch_view call (or mm_gpu_destroy);
 * the caller writes it out (fwrite / write) and does not free it.  For pipeline step 2, which one thread runs at a time: a gigabyte of
 * records per mini-batch is then formatted into memory that is already mapped, instead of into a fresh block whose page faults cost more
 * than the formatting. */
int mm_gpu_format_batch_view(int n_frag, const int *seg_off, const int *n_seg, MM2AMD_BSEQ_PTR seq, const int *n_reg, void *const *reg,
                             const int *rep_len, const char **out, size_t *out_len);

/* free() every reg[i][j].p and reg[i] (what the reference's step 2 does, map.c:629-631); for non-C callers. */
void mm2amd_free_regs(int n_frag, int *n_reg, MM2AMD_REG_PP reg);

/* Hit records of a (shard of a) batch as one flat payload -- the unit of the multi-GPU hit gather.  pack returns the
 * number of bytes needed/written (call with buf == NULL to size it); unpack rebuilds libc-allocated reg[] arrays. */
int64_t mm2amd_pack_regs(int n_frag, const int *n_reg, void *const *reg, uint8_t *buf, int64_t cap);
int mm2amd_unpack_regs(const uint8_t *buf, int64_t size, int n_frag, int *n_reg, MM2AMD_REG_PP reg);

/* Releases the device mirror; call before mm_idx_destroy (main.c:501). */
void mm_gpu_destroy(void);
/* The mapping context is process-wide, like the reference's pipeline: a later mm_gpu_init* replaces it.  Bindings whose objects
 * can outlive each other (two Python Aligner objects) remember the generation their init produced and tear down only that one. */
uint64_t mm_gpu_context_generation(void);        /* of the live context; 0: none */
int mm_gpu_destroy_if(uint64_t generation);      /* 1: it was the live context and is gone now; 0: left alone */

const char *mm2amd_backend_name(void);           /* "hip:gfx950" in the product library */
int mm2amd_format_fraction(double v, char *buf); /* diagnostics: the output stage's "%.4f" (de:f / dv:f tags; no printf: exact integer arithmetic) into buf[>= 16]; returns the length */
int mm2amd_last_stats(double *v, int n);         /* per-stage wall times of the last batch (diagnostics) */

/* Per-kernel timing (HIP events on the launch stream) and algorithmic bytes, accumulated while enabled. */
typedef struct { char name[48]; double ms; double alg_bytes; int64_t launches; double units; } mm2amd_kernel_stat_t; /* units: DP cells (DP kernels) */
void mm2amd_profile_enable(int on);              /* also clears the accumulated statistics */
int mm2amd_profile_get(mm2amd_kernel_stat_t *out, int cap); /* returns the number of kernels written */

#ifdef __cplusplus
}
#endif
#endif

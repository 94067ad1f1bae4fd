/* tests/dropin/dropin_pipeline.c -- TEST / MEASUREMENT INFRASTRUCTURE (not part of the product library).
 *
 * The reference's three-step mini-batch pipeline (worker_pipeline, map.c:541-643, run by kt_pipeline) with its step 1 replaced by
 * the GPU dispatcher and its step 2 by the library's parallel output stage -- INTEGRATION.md section 1 as a program:
 *   step 0  mm_bseq_read3: the reference's own FASTA/FASTQ reader (map.c:543-575)
 *           + mm_gpu_batch_stage_queued: the batch just read is handed to the library (pinned-memory packing + H2D beside the mapping
 *             of the batch before)
 *   step 1  mm_gpu_map_staged instead of kt_for(worker_for) (map.c:576)
 *   step 2  mm_gpu_format_batch_view instead of the mm_write_sam3 / mm_write_paf4 loop (map.c:585-623), then the frees of :624-636
 * -- the same three entry points, in the same places, as minimap2_amd.Aligner.pipeline() and therefore as bench.py's clock
 * (--one-call: step 1 = mm_gpu_map_batch, step 2 = mm_gpu_format_batch, the un-pipelined pair, for A/B).
 * kt_pipeline (the reference's, kthread.c:130) runs the steps of consecutive mini-batches on three threads, so parsing batch k+1,
 * mapping batch k and formatting batch k-1 overlap exactly as in the reference.  It prints the reference's own progress stamps
 * ("[M::worker_pipeline::<real>*<cpu/real>] mapped <n> sequences", map.c:638-639; "[M::main::...] loaded/built the index" after the
 * index is in memory, main.c:456-459), so tools/e2e_wall.py reads the same interval off both programs.
 *
 * usage: dropin_pipeline [-x preset] [-a|-c] [-t threads] [-K batch_bases] [--one-call] ref.fa|ref.mmi reads.fa     (single-end reads) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#include "bseq.h"
#include "kthread.h"
#include "mm2amd.h"

static int64_t parse_num(const char *str) /* main.c:103-111: 500M, 4g, 200k */
{
	char *q;
	double x = strtod(str, &q);
	if (*q == 'G' || *q == 'g') x *= 1e9;
	else if (*q == 'M' || *q == 'm') x *= 1e6;
	else if (*q == 'K' || *q == 'k') x *= 1e3;
	return (int64_t)(x + .499);
}

typedef struct {
	mm_bseq_file_t *fp[2];
	int n_fp;
	const mm_idx_t *mi;
	const mm_mapopt_t *opt;
	int64_t batch;
	int n_processed, failed, one_call, n_threads;
} pipeline_t;

typedef struct {
	pipeline_t *p;
	int n_seq, n_frag; /* fragments (map.c:556-563): a read pair is one fragment of two segments, a plain read a fragment of one */
	mm_bseq1_t *seq;
	int *n_reg, *seg_off, *n_seg, *rep_len, *frag_gap;
	mm_reg1_t **reg;
} step_t;

/* SURVEY.md 8(b) "error conventions": a mini-batch the GPU dispatcher fails on (its outputs are untouched) is mapped by the reference's own per-read
 * path -- what kt_for(worker_for) does for single-segment reads (map.c:425-474): mm_map_frag per read with a thread-local buffer. */
typedef struct { void *km; int rep_len, frag_gap; } tbuf_view_t; /* struct mm_tbuf_s (map.c:24-27): worker_for reads rep_len / frag_gap out of it (map.c:448-449) */
typedef struct { step_t *s; mm_tbuf_t **buf; } cpu_fallback_t;
static void cpu_fallback_one(void *data, long i, int tid)
{
	cpu_fallback_t *f = (cpu_fallback_t*)data;
	step_t *s = f->s;
	const int off = s->seg_off[i], n = s->n_seg[i];
	const char *seqs[2];
	int qlens[2], j;
	for (j = 0; j < n; ++j) seqs[j] = s->seq[off + j].seq, qlens[j] = s->seq[off + j].l_seq;
	mm_map_frag(s->p->mi, n, qlens, seqs, &s->n_reg[off], &s->reg[off], f->buf[tid], s->p->opt, s->seq[off].name);
	for (j = 0; j < n; ++j) s->rep_len[off + j] = ((tbuf_view_t*)f->buf[tid])->rep_len, s->frag_gap[off + j] = ((tbuf_view_t*)f->buf[tid])->frag_gap;
}
static void cpu_fallback(step_t *s, int n_threads)
{
	cpu_fallback_t f;
	int t;
	f.s = s, f.buf = (mm_tbuf_t**)calloc(n_threads, sizeof(mm_tbuf_t*));
	for (t = 0; t < n_threads; ++t) f.buf[t] = mm_tbuf_init();
	kt_for(n_threads, cpu_fallback_one, &f, s->n_frag);
	for (t = 0; t < n_threads; ++t) mm_tbuf_destroy(f.buf[t]);
	free(f.buf);
}

static void *worker(void *shared, int step, void *in)
{
	pipeline_t *p = (pipeline_t*)shared;
	int i, j;
	if (step == 0) {
		step_t *s = (step_t*)calloc(1, sizeof(step_t));
		const int with_qual = (p->opt->flag & MM_F_OUT_SAM) && !(p->opt->flag & MM_F_NO_QUAL);
		const int frag_mode = p->n_fp > 1 || !!(p->opt->flag & MM_F_FRAG_MODE); /* map.c:549-554 */
		if (p->n_fp > 1) s->seq = mm_bseq_read_frag2(p->n_fp, p->fp, p->batch, with_qual, !!(p->opt->flag & MM_F_COPY_COMMENT), &s->n_seq);
		else s->seq = mm_bseq_read3(p->fp[0], p->batch, with_qual, !!(p->opt->flag & MM_F_COPY_COMMENT), frag_mode, &s->n_seq);
		if (s->seq == 0) { free(s); return 0; }
		s->p = p;
		for (i = 0; i < s->n_seq; ++i) s->seq[i].rid = p->n_processed++;
		s->n_reg = (int*)calloc(5 * (size_t)s->n_seq, sizeof(int));
		s->seg_off = s->n_reg + s->n_seq, s->n_seg = s->seg_off + s->n_seq, s->rep_len = s->n_seg + s->n_seq, s->frag_gap = s->rep_len + s->n_seq;
		s->reg = (mm_reg1_t**)calloc(s->n_seq, sizeof(mm_reg1_t*));
		for (i = 1, j = 0; i <= s->n_seq; ++i) /* map.c:556-563: consecutive records of one name are one fragment */
			if (i == s->n_seq || !frag_mode || !mm_qname_same(s->seq[i - 1].name, s->seq[i].name)) {
				s->n_seg[s->n_frag] = i - j, s->seg_off[s->n_frag++] = j;
				j = i;
			}
		if (!p->one_call && !p->failed && mm_gpu_batch_stage_queued(s->n_frag, s->seg_off, s->n_seg, s->seq) != 0) { /* end of step 0 */
			fprintf(stderr, "mm_gpu_batch_stage_queued: %s\n", mm2amd_last_error());
			p->failed = 1;
		}
		return s;
	} else if (step == 1) {
		step_t *s = (step_t*)in;
		if (p->failed) { if (!p->one_call) mm_gpu_batch_discard(); return s; }
		if ((p->one_call? mm_gpu_map_batch(s->n_frag, s->seg_off, s->n_seg, s->seq, s->n_reg, (void**)s->reg, s->rep_len, s->frag_gap)
		                : mm_gpu_map_staged(s->n_reg, (void**)s->reg, s->rep_len, s->frag_gap)) != 0) {
			fprintf(stderr, "[WARNING] %s: %s; this mini-batch is mapped by the reference's own path\n", p->one_call? "mm_gpu_map_batch" : "mm_gpu_map_staged", mm2amd_last_error());
			cpu_fallback(s, p->n_threads);
		}
		return s;
	} else {
		step_t *s = (step_t*)in;
		char *text = 0;
		const char *view = 0;
		size_t text_len = 0;
		if (p->failed) ;
		else if (p->one_call) {
			if (mm_gpu_format_batch(s->n_frag, s->seg_off, s->n_seg, s->seq, s->n_reg, (void *const*)s->reg, s->rep_len, &text, &text_len) != 0) {
				fprintf(stderr, "mm_gpu_format_batch: %s\n", mm2amd_last_error());
				p->failed = 1;
			}
			if (text) fwrite(text, 1, text_len, stdout), free(text);
		} else { /* the library owns and reuses the buffer: nothing to free */
			if (mm_gpu_format_batch_view(s->n_frag, s->seg_off, s->n_seg, s->seq, s->n_reg, (void *const*)s->reg, s->rep_len, &view, &text_len) != 0) {
				fprintf(stderr, "mm_gpu_format_batch_view: %s\n", mm2amd_last_error());
				p->failed = 1;
			} else fwrite(view, 1, text_len, stdout);
		}
		for (i = 0; i < s->n_seq; ++i) {
			mm_bseq1_t *t = &s->seq[i];
			for (j = 0; j < s->n_reg[i]; ++j) free(s->reg[i][j].p);
			free(s->reg[i]);
			free(t->seq); free(t->name);
			if (t->qual) free(t->qual);
			if (t->comment) free(t->comment);
		}
		fprintf(stderr, "[M::worker_pipeline::%.3f*%.2f] mapped %d sequences\n", realtime() - mm_realtime0, cputime() / (realtime() - mm_realtime0), s->n_seq);
		free(s->reg); free(s->n_reg); free(s->seq); free(s);
	}
	return 0;
}

int main(int argc, char *argv[])
{
	mm_idxopt_t iopt;
	mm_mapopt_t mopt;
	mm_idx_reader_t *rd;
	mm_idx_t *mi;
	int n_threads = 3, k = 1, rc = 0, one_call = 0;
	int64_t batch = 500000000;
	mm_realtime0 = realtime();
	mm_set_opt(0, &iopt, &mopt);
	for (; k < argc && argv[k][0] == '-' && argv[k][1]; ++k) {
		if (strcmp(argv[k], "-x") == 0) { if (mm_set_opt(argv[++k], &iopt, &mopt) < 0) { fprintf(stderr, "unknown preset\n"); return 1; } }
		else if (strcmp(argv[k], "-a") == 0) mopt.flag |= MM_F_OUT_SAM | MM_F_CIGAR;
		else if (strcmp(argv[k], "-c") == 0) mopt.flag |= MM_F_OUT_CG | MM_F_CIGAR; /* main.c:238 */
		else if (strcmp(argv[k], "-t") == 0) n_threads = atoi(argv[++k]);
		else if (strcmp(argv[k], "-K") == 0) batch = parse_num(argv[++k]);
		else if (strcmp(argv[k], "--one-call") == 0) one_call = 1;
		else { fprintf(stderr, "unknown option %s\n", argv[k]); return 1; }
	}
	if (argc - k < 2) { fprintf(stderr, "usage: dropin_pipeline [-x preset] [-a|-c] [-t threads] [-K batch] ref reads [mates]\n"); return 1; }
	if (mm_check_opt(&iopt, &mopt) < 0) return 1;
	rd = mm_idx_reader_open(argv[k], &iopt, 0);
	if (rd == 0) { fprintf(stderr, "failed to open %s\n", argv[k]); return 1; }
	while ((mi = mm_idx_reader_read(rd, n_threads)) != 0) {
		pipeline_t pl;
		if (mopt.flag & MM_F_OUT_SAM) mm_write_sam_hdr(mm_idx_reader_eof(rd) ? mi : 0, 0, MM_VERSION, 0, 0);
		fprintf(stderr, "[M::main::%.3f*%.2f] loaded/built the index for %d target sequence(s)\n", realtime() - mm_realtime0, cputime() / (realtime() - mm_realtime0), mi->n_seq);
		mm_mapopt_update(&mopt, mi);
		setenv("MM2AMD_MALLOPT", "1", 0);
		if (mm_gpu_init(mi, &mopt, n_threads) != 0) { fprintf(stderr, "mm_gpu_init: %s\n", mm2amd_last_error()); return 2; }
		fprintf(stderr, "[M::main::%.3f*%.2f] device mirror of the index ready (%d replica(s))\n", realtime() - mm_realtime0, cputime() / (realtime() - mm_realtime0), mm_gpu_n_replicas());
		memset(&pl, 0, sizeof pl);
		for (pl.n_fp = 0; pl.n_fp < 2 && k + 1 + pl.n_fp < argc; ++pl.n_fp) { /* one file, or the two files of paired-end reads (map.c:650-667) */
			pl.fp[pl.n_fp] = mm_bseq_open(argv[k + 1 + pl.n_fp]);
			if (pl.fp[pl.n_fp] == 0) { fprintf(stderr, "failed to open %s\n", argv[k + 1 + pl.n_fp]); return 1; }
		}
		pl.mi = mi, pl.opt = &mopt, pl.batch = batch, pl.one_call = one_call, pl.n_threads = n_threads;
		kt_pipeline(3, worker, &pl, 3); /* map.c:669: pl_threads = n_threads == 1 ? 1 : 3 (2 with --2-io-threads off) */
		rc |= pl.failed;
		{ int f; for (f = 0; f < pl.n_fp; ++f) mm_bseq_close(pl.fp[f]); }
		mm_gpu_destroy();
		mm_idx_destroy(mi);
	}
	mm_idx_reader_close(rd);
	if (fflush(stdout) == EOF) return 1;
	fprintf(stderr, "[M::main] Real time: %.3f sec; CPU: %.3f sec\n", realtime() - mm_realtime0, cputime());
	return rc ? 2 : 0;
}

/* tests/dropin/dropin_main.c -- TEST HARNESS.
 *
 * A minimal minimap2-like front end that keeps the reference's own host code for everything outside the hot path
 * (option presets, index reader, FASTA reader, SAM/PAF writer -- linked from oracle/_ref/libminimap2_ref.a) and
 * calls OUR mapper through the drop-in C ABI (include/mm2amd.h) where the reference runs kt_for(worker_for)
 * (map.c:576).  Its output is diffed against oracle/_ref/minimap2_ref: this is the "SAM diff == 0" gate, and it is
 * also exactly the binding INTEGRATION.md proposes.  Compiled against the reference headers where they lie
 * (/root/reference), so it is built in the dev container and travels to the GPU box as a prebuilt binary.
 *
 * usage: dropin [-x preset] [-a|-c] [-t threads] [-K batch_bases] [-s seed] [-z zdrop[,inv]] [--stats] ref.fa|ref.mmi reads.fa
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "minimap.h"
#include "mmpriv.h"
#include "bseq.h"
#include "mm2amd.h"

int main(int argc, char *argv[])
{
	mm_idxopt_t iopt;
	mm_mapopt_t mopt;
	const char *preset = 0;
	int n_threads = 3, i, k = 1, print_stats = 0, format_lib = 0, staged = 0, one_by_one = 0, batch_with = 0, old_best_n = -1;
	const char *alt_fn = 0, *junc_fn = 0, *jump_fn = 0, *pass1_fn = 0, *spsc_fn = 0;
	float spsc_scale = 0.7f;
	int64_t batch = 500000000;
	kstring_t str = {0, 0, 0};

	mm_verbose = 2;
	mm_set_opt(0, &iopt, &mopt);
	for (i = 1; i < argc; ++i) /* presets first, like main.c:150-163 */
		if (strcmp(argv[i], "-x") == 0 && i + 1 < argc) preset = argv[i + 1];
	if (preset && mm_set_opt(preset, &iopt, &mopt) < 0) { fprintf(stderr, "unknown preset %s\n", preset); return 1; }
	for (k = 1; k < argc && argv[k][0] == '-'; ++k) {
		if (strcmp(argv[k], "-x") == 0) ++k;
		else if (strcmp(argv[k], "-a") == 0) mopt.flag |= MM_F_OUT_SAM | MM_F_CIGAR;
		else if (strcmp(argv[k], "-c") == 0) mopt.flag |= MM_F_OUT_CG | MM_F_CIGAR;
		else if (strcmp(argv[k], "-t") == 0) n_threads = atoi(argv[++k]);
		else if (strcmp(argv[k], "-K") == 0) batch = atoll(argv[++k]);
		else if (strcmp(argv[k], "-I") == 0) iopt.batch_size = atoll(argv[++k]); /* main.c:196: a multi-part index, one mm_gpu_init / mm_gpu_destroy per part */
		else if (strcmp(argv[k], "-s") == 0) mopt.min_dp_max = atoi(argv[++k]);
		else if (strcmp(argv[k], "--seed") == 0) mopt.seed = atoi(argv[++k]);
		else if (strcmp(argv[k], "--stats") == 0) print_stats = 1;
		else if (strcmp(argv[k], "--batch-with") == 0) batch_with = 1; /* no mm_gpu_init: every batch through mm_gpu_map_batch_with(mi, &mopt, ...), which builds the context on first use */
		else if (strcmp(argv[k], "--one-by-one") == 0) one_by_one = 1; /* mm_gpu_map / mm_gpu_map_frag per fragment instead of one mm_gpu_map_batch */
		else if (strcmp(argv[k], "-O") == 0) { char *s; mopt.q = mopt.q2 = strtol(argv[++k], &s, 10); if (*s == ',') mopt.q2 = strtol(s + 1, &s, 10); }
		else if (strcmp(argv[k], "-E") == 0) { char *s; mopt.e = mopt.e2 = strtol(argv[++k], &s, 10); if (*s == ',') mopt.e2 = strtol(s + 1, &s, 10); }
		else if (strcmp(argv[k], "-A") == 0) mopt.a = atoi(argv[++k]);
		else if (strcmp(argv[k], "-B") == 0) mopt.b = atoi(argv[++k]);
		else if (strcmp(argv[k], "-u") == 0) { /* main.c:332-336 */
			const char c = argv[++k][0];
			if (c == 'b') mopt.flag |= MM_F_SPLICE_FOR | MM_F_SPLICE_REV;
			else if (c == 'f') mopt.flag |= MM_F_SPLICE_FOR, mopt.flag &= ~MM_F_SPLICE_REV;
			else if (c == 'r') mopt.flag |= MM_F_SPLICE_REV, mopt.flag &= ~MM_F_SPLICE_FOR;
			else mopt.flag &= ~(MM_F_SPLICE_FOR | MM_F_SPLICE_REV);
		} else if (strcmp(argv[k], "-G") == 0) mm_mapopt_max_intron_len(&mopt, atoi(argv[++k]));
		else if (strcmp(argv[k], "-C") == 0) mopt.noncan = atoi(argv[++k]);
		else if (strcmp(argv[k], "-z") == 0) { char *s; mopt.zdrop = mopt.zdrop_inv = strtol(argv[++k], &s, 10); if (*s == ',') mopt.zdrop_inv = strtol(s + 1, &s, 10); }
		else if (strcmp(argv[k], "--splice-flank=no") == 0) mopt.flag &= ~MM_F_SPLICE_FLANK;
		else if (strcmp(argv[k], "-J") == 0) { if (atoi(argv[++k]) == 0) mopt.flag |= MM_F_SPLICE_OLD; else mopt.flag &= ~MM_F_SPLICE_OLD; }
		else if (strcmp(argv[k], "--cs") == 0) mopt.flag |= MM_F_OUT_CS | MM_F_CIGAR, mopt.flag &= ~MM_F_OUT_CS_LONG; /* main.c:284-287 */
		else if (strcmp(argv[k], "--MD") == 0) mopt.flag |= MM_F_OUT_MD;
		else if (strcmp(argv[k], "--eqx") == 0) mopt.flag |= MM_F_EQX;
		else if (strcmp(argv[k], "--ds") == 0) mopt.flag |= MM_F_OUT_DS; /* main.c:258 */
		else if (strcmp(argv[k], "--cs=long") == 0) mopt.flag |= MM_F_OUT_CS | MM_F_OUT_CS_LONG | MM_F_CIGAR;
		else if (strcmp(argv[k], "-Y") == 0) mopt.flag |= MM_F_SOFTCLIP;
		else if (strcmp(argv[k], "-L") == 0) mopt.flag |= MM_F_LONG_CIGAR;
		else if (strcmp(argv[k], "-y") == 0) mopt.flag |= MM_F_COPY_COMMENT;
		else if (strcmp(argv[k], "--secondary=no") == 0) mopt.flag |= MM_F_NO_PRINT_2ND;
		else if (strcmp(argv[k], "--secondary-seq") == 0) mopt.flag |= MM_F_SECONDARY_SEQ;
		else if (strcmp(argv[k], "--paf-no-hit") == 0) mopt.flag |= MM_F_PAF_NO_HIT;
		else if (strcmp(argv[k], "--sam-hit-only") == 0) mopt.flag |= MM_F_SAM_HIT_ONLY;
		/* more of main.c's mapping options (main.c:168-352), for the option-fuzzing parity test */
		else if (strcmp(argv[k], "-k") == 0) iopt.k = atoi(argv[++k]);
		else if (strcmp(argv[k], "-w") == 0) iopt.w = atoi(argv[++k]);
		else if (strcmp(argv[k], "-H") == 0) iopt.flag |= MM_I_HPC;
		else if (strcmp(argv[k], "-g") == 0) mopt.max_gap = atoi(argv[++k]);
		else if (strcmp(argv[k], "-r") == 0) { char *e; mopt.bw = (int)strtol(argv[++k], &e, 10); if (*e == ',') mopt.bw_long = (int)strtol(e + 1, &e, 10); }
		else if (strcmp(argv[k], "-U") == 0) { char *e; mopt.min_mid_occ = (int)strtol(argv[++k], &e, 10); if (*e == ',') mopt.max_mid_occ = (int)strtol(e + 1, &e, 10); }
		else if (strcmp(argv[k], "-f") == 0) { char *e; double x = strtod(argv[++k], &e); if (x < 1.0) mopt.mid_occ_frac = (float)x, mopt.mid_occ = 0; else mopt.mid_occ = (int)(x + .499); if (*e == ',') mopt.max_occ = (int)(strtod(e + 1, &e) + .499); }
		else if (strcmp(argv[k], "-N") == 0) old_best_n = mopt.best_n, mopt.best_n = atoi(argv[++k]);
		else if (strcmp(argv[k], "-p") == 0) mopt.pri_ratio = (float)atof(argv[++k]);
		else if (strcmp(argv[k], "-M") == 0) mopt.mask_level = (float)atof(argv[++k]);
		else if (strcmp(argv[k], "-n") == 0) mopt.min_cnt = atoi(argv[++k]);
		else if (strcmp(argv[k], "-m") == 0) mopt.min_chain_score = atoi(argv[++k]);
		else if (strcmp(argv[k], "-b") == 0) mopt.transition = atoi(argv[++k]);
		else if (strcmp(argv[k], "-P") == 0) mopt.flag |= MM_F_ALL_CHAINS;
		else if (strcmp(argv[k], "-e") == 0) mopt.occ_dist = atoi(argv[++k]);
		else if (strcmp(argv[k], "--max-chain-skip") == 0) mopt.max_chain_skip = atoi(argv[++k]);
		else if (strcmp(argv[k], "--max-chain-iter") == 0) mopt.max_chain_iter = atoi(argv[++k]);
		else if (strcmp(argv[k], "--min-dp-len") == 0) mopt.min_ksw_len = atoi(argv[++k]);
		else if (strcmp(argv[k], "--no-long-join") == 0) mopt.flag |= MM_F_NO_LJOIN;
		else if (strcmp(argv[k], "--end-bonus") == 0) mopt.end_bonus = atoi(argv[++k]);
		else if (strcmp(argv[k], "--end-seed-pen") == 0) mopt.anchor_ext_shift = atoi(argv[++k]);
		else if (strcmp(argv[k], "--min-occ-floor") == 0) mopt.min_mid_occ = atoi(argv[++k]);
		else if (strcmp(argv[k], "--score-N") == 0) mopt.sc_ambi = atoi(argv[++k]);
		else if (strcmp(argv[k], "--no-end-flt") == 0) mopt.flag |= MM_F_NO_END_FLT;
		else if (strcmp(argv[k], "--hard-mask-level") == 0) mopt.flag |= MM_F_HARD_MLEVEL;
		else if (strcmp(argv[k], "--cap-sw-mat") == 0) mopt.max_sw_mat = atoll(argv[++k]);
		else if (strcmp(argv[k], "--max-qlen") == 0) mopt.max_qlen = atoi(argv[++k]);
		else if (strcmp(argv[k], "--chain-gap-scale") == 0) mopt.chain_gap_scale = (float)atof(argv[++k]);
		else if (strcmp(argv[k], "--chain-skip-scale") == 0) mopt.chain_skip_scale = (float)atof(argv[++k]);
		else if (strcmp(argv[k], "--alt-drop") == 0) mopt.alt_drop = (float)atof(argv[++k]);
		else if (strcmp(argv[k], "--mask-len") == 0) mopt.mask_len = atoi(argv[++k]);
		else if (strcmp(argv[k], "--q-occ-frac") == 0) mopt.q_occ_frac = (float)atof(argv[++k]);
		else if (strcmp(argv[k], "--no-hash-name") == 0) mopt.flag |= MM_F_NO_HASH_NAME;
		else if (strcmp(argv[k], "--for-only") == 0) mopt.flag |= MM_F_FOR_ONLY;
		else if (strcmp(argv[k], "--rev-only") == 0) mopt.flag |= MM_F_REV_ONLY;
		else if (strcmp(argv[k], "--heap-sort=yes") == 0) mopt.flag |= MM_F_HEAP_SORT; /* main.c:298 */
		else if (strcmp(argv[k], "--heap-sort=no") == 0) mopt.flag &= ~(int64_t)MM_F_HEAP_SORT;
		else if (strcmp(argv[k], "--sr") == 0) mopt.flag |= MM_F_SR;
		else if (strcmp(argv[k], "--qstrand") == 0) mopt.flag |= MM_F_QSTRAND | MM_F_NO_INV; /* main.c:252 */
		else if (strcmp(argv[k], "--no-pairing") == 0) mopt.flag |= MM_F_INDEPEND_SEG; /* main.c:228 */
		else if (strcmp(argv[k], "-F") == 0) mopt.max_frag_len = atoi(argv[++k]);
		else if (strcmp(argv[k], "-T") == 0) mopt.sdust_thres = atoi(argv[++k]); /* main.c:171 */
		else if (strcmp(argv[k], "-D") == 0) mopt.flag |= MM_F_NO_DIAG; /* main.c:180 */
		else if (strcmp(argv[k], "-X") == 0) mopt.flag |= MM_F_ALL_CHAINS | MM_F_NO_DIAG | MM_F_NO_DUAL | MM_F_NO_LJOIN; /* main.c:182 */
		else if (strcmp(argv[k], "--dual=no") == 0) mopt.flag |= MM_F_NO_DUAL; /* main.c:300 */
		else if (strcmp(argv[k], "--dual=yes") == 0) mopt.flag &= ~(int64_t)MM_F_NO_DUAL;
		else if (strcmp(argv[k], "--alt") == 0) alt_fn = argv[++k];
		else if (strcmp(argv[k], "--junc-bed") == 0) junc_fn = argv[++k]; /* main.c:244,468 */
		else if (strcmp(argv[k], "-j") == 0) jump_fn = argv[++k]; /* main.c:202,473 */
		else if (strcmp(argv[k], "--pass1") == 0) pass1_fn = argv[++k]; /* main.c:478 */
		else if (strcmp(argv[k], "--spsc") == 0) spsc_fn = argv[++k]; /* main.c:260 */
		else if (strcmp(argv[k], "--spsc-scale") == 0) spsc_scale = (float)atof(argv[++k]); /* main.c:265 */
		else if (strcmp(argv[k], "--spsc0") == 0 || strcmp(argv[k], "--junc-pen") == 0) mopt.junc_pen = atoi(argv[++k]); /* main.c:266 */
		else if (strcmp(argv[k], "--junc-bonus") == 0) mopt.junc_bonus = atoi(argv[++k]); /* main.c:245 */
		else if (strcmp(argv[k], "--staged") == 0) staged = 1; /* mm_gpu_batch_stage + mm_gpu_map_staged instead of mm_gpu_map_batch */
		else if (strcmp(argv[k], "--format-lib") == 0) format_lib = 1; /* records written by mm_gpu_format_batch instead of the reference's writers */
		else { fprintf(stderr, "unknown option %s\n", argv[k]); return 1; }
	}
	if (argc - k < 2) { fprintf(stderr, "usage: dropin [options] ref reads [mates]\n"); return 1; }
	if (!(mopt.flag & MM_F_CIGAR)) iopt.flag |= MM_I_NO_SEQ; /* main.c:352-353 */
	if (mm_check_opt(&iopt, &mopt) < 0) return 1;
	if (mopt.best_n == 0) mopt.best_n = old_best_n, mopt.flag |= MM_F_NO_PRINT_2ND; /* main.c:356-359: '-N 0' becomes '-N <preset> --secondary=no' */

	mm_idx_reader_t *rd = mm_idx_reader_open(argv[k], &iopt, 0);
	if (rd == 0) { fprintf(stderr, "failed to open %s\n", argv[k]); return 1; }
	mm_idx_t *mi;
	int n_parts = 0;
	while ((mi = mm_idx_reader_read(rd, n_threads)) != 0) {
		if ((mopt.flag & MM_F_OUT_SAM) && n_parts++ == 0) /* main.c:443-455: one header, with @SQ lines only for a single-part index */
			mm_write_sam_hdr(mm_idx_reader_eof(rd) ? mi : 0, 0, MM_VERSION, 0, 0);
		mm_mapopt_update(&mopt, mi);
		if (junc_fn) mm_idx_bed_read(mi, junc_fn, 1); /* main.c:468 */
		if (jump_fn) mm_idx_jjump_read(mi, jump_fn, MM_JUNC_ANNO, -1); /* main.c:473 */
		if (pass1_fn) mm_idx_jjump_read(mi, pass1_fn, MM_JUNC_MISC, 5); /* main.c:478 */
		if (spsc_fn) mm_idx_spsc_read2(mi, spsc_fn, mm_max_spsc_bonus(&mopt), spsc_scale); /* main.c:483 */
		if (alt_fn) mm_idx_alt_read(mi, alt_fn); /* main.c:480 */
		setenv("MM2AMD_MALLOPT", "1", 0); /* this driver owns its process: let the library tune glibc malloc */
		if (batch_with) {
			/* the context comes from the first mm_gpu_map_batch_with call (default host pool) */
		} else if (mm_gpu_init(mi, &mopt, n_threads) != 0) { fprintf(stderr, "mm_gpu_init: %s\n", mm2amd_last_error()); return 2; }
		/* one read file, or two for paired-end reads (worker_pipeline step 0, map.c:545-569) */
		/* without MM_F_FRAG_MODE several query files are mapped one after the other (main.c:493-500) */
		int n_files = argc - (k + 1) >= 2 ? 2 : 1, file0;
		for (file0 = 0; file0 < n_files; file0 += (mopt.flag & MM_F_FRAG_MODE) ? n_files : 1) {
		int n_fp = (mopt.flag & MM_F_FRAG_MODE) ? n_files : 1, n_frag;
		int frag_mode = (n_fp > 1 || !!(mopt.flag & MM_F_FRAG_MODE));
		mm_bseq_file_t *fp, *fps[2];
		for (i = 0; i < n_fp; ++i) {
			fps[i] = mm_bseq_open(argv[k + 1 + file0 + i]);
			if (fps[i] == 0) { fprintf(stderr, "failed to open %s\n", argv[k + 1 + file0 + i]); return 1; }
		}
		fp = fps[0];
		int with_qual = (!!(mopt.flag & MM_F_OUT_SAM) && !(mopt.flag & MM_F_NO_QUAL)), n_seq;
		mm_bseq1_t *seq;
		while ((seq = n_fp > 1 ? mm_bseq_read_frag2(n_fp, fps, batch, with_qual, !!(mopt.flag & MM_F_COPY_COMMENT), &n_seq)
		                       : mm_bseq_read3(fp, batch, with_qual, !!(mopt.flag & MM_F_COPY_COMMENT), frag_mode, &n_seq)) != 0) {
			int *n_reg = (int*)calloc(5 * (size_t)n_seq, sizeof(int));
			int *seg_off = n_reg + n_seq, *n_seg = seg_off + n_seq, *rep_len = n_seg + n_seq, *frag_gap = rep_len + n_seq;
			mm_reg1_t **reg = (mm_reg1_t**)calloc(n_seq, sizeof(mm_reg1_t*));
			int j, f;
			for (i = 1, j = 0, n_frag = 0; i <= n_seq; ++i)
				if (i == n_seq || !frag_mode || !mm_qname_same(seq[i-1].name, seq[i].name)) n_seg[n_frag] = i - j, seg_off[n_frag++] = j, j = i;
			if (one_by_one) {
				for (f = 0; f < n_frag; ++f) {
					struct mm_tbuf_s tb = { 0, 0, 0 };
					int o = seg_off[f], qlens[2];
					const char *seqs[2];
					for (i = 0; i < n_seg[f]; ++i) qlens[i] = seq[o + i].l_seq, seqs[i] = seq[o + i].seq;
					if (n_seg[f] == 1) reg[o] = (mm_reg1_t*)mm_gpu_map(mi, qlens[0], seqs[0], &n_reg[o], &tb, &mopt, seq[o].name);
					else mm_gpu_map_frag(mi, n_seg[f], qlens, seqs, &n_reg[o], (void**)&reg[o], &tb, &mopt, seq[o].name);
					for (i = 0; i < n_seg[f]; ++i) rep_len[o + i] = tb.rep_len, frag_gap[o + i] = tb.frag_gap;
				}
			} else if (staged) {
				if (mm_gpu_batch_stage(n_frag, seg_off, n_seg, seq) != 0 || mm_gpu_map_staged(n_reg, (void**)reg, rep_len, frag_gap) != 0) {
					fprintf(stderr, "staged mapping: %s\n", mm2amd_last_error());
					return 2;
				}
			} else if (batch_with) {
				if (mm_gpu_map_batch_with(mi, &mopt, n_frag, seg_off, n_seg, seq, n_reg, (void**)reg, rep_len, frag_gap) != 0) {
					fprintf(stderr, "mm_gpu_map_batch_with: %s\n", mm2amd_last_error());
					return 2;
				}
			} else if (mm_gpu_map_batch(n_frag, seg_off, n_seg, seq, n_reg, (void**)reg, rep_len, frag_gap) != 0) {
				fprintf(stderr, "mm_gpu_map_batch: %s\n", mm2amd_last_error());
				return 2;
			}
			if (print_stats) {
				double v[16];
				int nv = mm2amd_last_stats(v, 16);
				fprintf(stderr, "[dropin] backend=%s replicas=%d reads=%d", mm2amd_backend_name(), mm_gpu_n_replicas(), n_seq);
				for (i = 0; i < nv; ++i) fprintf(stderr, " %.4g", v[i]);
				fputc('\n', stderr);
			}
			if (format_lib) {
				char *text = 0;
				size_t text_len = 0;
				if (mm_gpu_format_batch(n_frag, seg_off, n_seg, seq, n_reg, (void *const*)reg, rep_len, &text, &text_len) != 0) {
					fprintf(stderr, "mm_gpu_format_batch: %s\n", mm2amd_last_error());
					return 2;
				}
				fwrite(text, 1, text_len, stdout);
				free(text);
			}
			for (f = 0; f < n_frag; ++f) { /* output, as step 2 of worker_pipeline (map.c:585-636) */
				int seg_st = seg_off[f], seg_en = seg_off[f] + n_seg[f];
				for (i = seg_st; i < seg_en; ++i) {
					mm_bseq1_t *t = &seq[i];
					if (format_lib) {
					} else if (n_reg[i] > 0) {
						for (j = 0; j < n_reg[i]; ++j) {
							const mm_reg1_t *r = &reg[i][j];
							if ((mopt.flag & MM_F_NO_PRINT_2ND) && r->id != r->parent) continue;
							if (mopt.flag & MM_F_OUT_SAM) mm_write_sam3(&str, mi, t, i - seg_st, j, n_seg[f], &n_reg[seg_st], (const mm_reg1_t*const*)&reg[seg_st], 0, mopt.flag, rep_len[i]);
							else mm_write_paf4(&str, mi, t, r, 0, mopt.flag, rep_len[i], n_seg[f], i - seg_st);
							mm_err_puts(str.s);
						}
					} else if ((mopt.flag & MM_F_PAF_NO_HIT) || ((mopt.flag & MM_F_OUT_SAM) && !(mopt.flag & MM_F_SAM_HIT_ONLY))) {
						if (mopt.flag & MM_F_OUT_SAM) mm_write_sam3(&str, mi, t, i - seg_st, -1, n_seg[f], &n_reg[seg_st], (const mm_reg1_t*const*)&reg[seg_st], 0, mopt.flag, rep_len[i]);
						else mm_write_paf4(&str, mi, t, 0, 0, mopt.flag, rep_len[i], n_seg[f], i - seg_st);
						mm_err_puts(str.s);
					}
				}
				for (i = seg_st; i < seg_en; ++i) {
					mm_bseq1_t *t = &seq[i];
					for (j = 0; j < n_reg[i]; ++j) free(reg[i][j].p);
					free(reg[i]);
					free(t->seq); free(t->name);
					if (t->qual) free(t->qual);
					if (t->comment) free(t->comment);
				}
			}
			free(reg); free(n_reg); free(seq);
		}
		for (i = 0; i < n_fp; ++i) mm_bseq_close(fps[i]);
		}
		mm_gpu_destroy();
		mm_idx_destroy(mi);
	}
	mm_idx_reader_close(rd);
	free(str.s);
	if (fflush(stdout) == EOF) return 1;
	return 0;
}

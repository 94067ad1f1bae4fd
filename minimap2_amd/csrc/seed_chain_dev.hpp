// Device-side contract of the seeding + chaining stages (seed_chain.hip).
#pragma once
#include <cstdint>
#include <cstddef>
#include "types.hpp"
#include "backend.hpp"

namespace mm2amd {

struct DevIndex {             // device mirror of FlatIndex
	const uint32_t *bucket_start;
	const uint64_t *keys;
	const uint32_t *val_off;
	const struct IdxSlot *slots; // (key, first position, count) per distinct minimizer: what a probe reads (index_build.hpp)
	const struct IdxSlot *first; // the same per BUCKET: its first key inline (round 6); null: probes go through bucket_start only
	const uint64_t *pos;
	const uint32_t *S;
	int bucket_bits, key_shift;
	// all-vs-all rules (skip_seed, map.c:81-91): rank of each sequence's name among the distinct sorted names, and its length
	const int32_t *name_rank;
	const uint32_t *seq_len;
};

struct SeedChainBuffers {     // all device pointers; per-read slices addressed through the offset arrays
	int n_reads;
	const uint64_t *seq_off;  // n_reads+1: base offset of read r in the ASCII pool; the nt4 pool places it at 2*seq_off[r]
	const char *ascii;
	uint8_t *qpool;           // nt4 forward | reverse complement per read
	// per read, for the all-vs-all rules: number of distinct reference names that sort before the read's name, and the rank of
	// the reference name equal to it (-1: none).  Null when the batch carries no read names.
	const int32_t *name_lb, *name_eq;
	// pairs (null otherwise): fragment r consists of units unit_first[r] .. unit_first[r+1]-1 (one or two reads), unit u starts at
	// base unit_off[u] of the batch and has unit_cnt[u] minimizers, written from slot unit_off[u]
	const int32_t *unit_first;
	const uint64_t *unit_off;
	const uint32_t *unit_cnt;
	// minimizers
	uint32_t *mz_cnt;         // n_reads
	const uint64_t *mz_off;   // n_reads (+1): first minimizer slot of each read (the read's base offset: at most one minimizer per base)
	uint64_t *mz_x, *mz_y;
	// per-minimizer seed info (same indexing as mz_*)
	uint32_t *sd_n, *sd_off, *sd_aoff, *sd_qpos, *sd_info; // info: bits0-7 span, bit8 tandem, bit9 filtered
	// per-read seed results
	uint32_t *n_anchor, *n_minipos, *n_seedhit;
	int32_t *rep_len;
	const uint64_t *a_off;    // n_reads+1 anchor offsets
	const uint64_t *mp_off;   // n_reads+1 mini_pos offsets
	Anchor *anchors;
	uint64_t *sort_key_in, *sort_val_in; // anchors as (compact key: strand | rid | rpos, y) pairs in seed order, before the per-read sort
	uint64_t *sort_key_out, *sort_val_out; // per-read scratch of the same size: the sort of reads beyond the LDS classes, RMQ priorities, the backtrack's sorts
	uint32_t *tie_flag;       // n_reads: set when a read has two anchors with equal x
	uint32_t *tie_list, *tie_count; // those reads (of the LDS classes), in the order their sorting workgroups found out; how many: anchor_sort_ties_kernel's work list
	int rid_bits;             // bits needed for a reference sequence id in the compact sort key
	uint64_t *mini_pos;
	int32_t *f, *p, *t;       // chaining DP arrays, indexed like anchors
	// the chaining kernels' work list (null: one wavefront per read): (read, piece number) pairs, a piece = piece_len anchors moved to the next cluster
	// heads (seed_chain.hip: chain_piece_bounds); made by the host when a read of the launch has more than piece_len anchors, heaviest reads first
	const uint32_t *pieces = nullptr;
	int n_pieces = 0, piece_len = 0;
	int rmq_rank_max = 256;   // chain_rmq_kernel: neighbourhoods up to this many anchors are sorted by counting larger keys (at most 256; 0: always the bitonic network -- MM2AMD_RMQ_RANK_MAX)
	int piece_dense = 0;      // chain_rmq_kernel: a piece of this many anchors or more (a cluster that long) is chain_rmq_wide_kernel's; 0 = none is
	// chain backtrack results: dense outputs handed out by two atomic cursors ([0] anchors, [1] chains)
	unsigned long long *bt_cursor;
	Anchor *bt_out_a;         // compacted anchors of all chains, chain by chain, read by read (in completion order)
	uint64_t *bt_out_u;       // score<<32 | n_anchors per chain
	int32_t *bt_nu, *bt_nv;   // per read: chains, anchors
	uint64_t *bt_aoff, *bt_uoff; // per read: where its results start in bt_out_a / bt_out_u
};

void launch_encode(const SeedChainBuffers &B, void *stream);
void launch_chain_rmq(const SeedChainBuffers &B, const SeedChainParams &P, void *stream); // mg_lchain_rmq's fill; tie_flag[r] = 1: read r is left to the host
void launch_sketch(const SeedChainBuffers &B, const SeedChainParams &P, int max_len, void *stream); // max_len: the longest read (unit) of the launch
void launch_dust_filter(const SeedChainBuffers &B, void *stream); // regions in sd_n / sd_off / sd_aoff (see seed_chain.hip)
void launch_seed_collect(const SeedChainBuffers &B, const DevIndex &I, const SeedChainParams &P, void *stream);
void launch_seed_expand(const SeedChainBuffers &B, const DevIndex &I, const SeedChainParams &P, void *stream);
class KernelProfiler;
// the per-read anchor sort's launch classes (by anchors per read; the last one sorts on global scratch): `list` holds the reads grouped
// by class, n_class[c] of them in class c, carrying anchors_in_class[c] anchors
constexpr int kAnchorSortClasses = 6;
int anchor_sort_class(uint64_t n_anchors, int rid_bits);
void launch_anchor_sort(const SeedChainBuffers &B, const DevIndex &I, const SeedChainParams &P, const uint32_t *d_list, const int *n_class, const double *anchors_in_class, void *stream,
                        KernelProfiler *kp);
void launch_chain_fill(const SeedChainBuffers &B, const SeedChainParams &P, void *stream);
void launch_chain_backtrack(const SeedChainBuffers &B, const SeedChainParams &P, void *stream);
void launch_rechain_gather(const SeedChainBuffers &B, const Anchor *src, const uint64_t *src_off, void *stream); // long-join: chained anchors -> the per-read sort's input

} // namespace mm2amd

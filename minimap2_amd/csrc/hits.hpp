// Chain -> hit bookkeeping on the host: the scalar per-read post-processing between chaining and base-level
// alignment and after it (the reference's hit.c and esterr.c).  Tens of records per read; stays on the host.
#pragma once
#include <vector>
#include "types.hpp"
#include "flat_index.hpp"

namespace mm2amd {

using Reg = ref::Reg1;
using RegVec = std::vector<Reg>;

void reg_set_coor(Reg &r, int32_t qlen, const Anchor *a, bool is_qstrand);                         // hit.c:24-38
void gen_regs(uint32_t hash, int qlen, const uint64_t *u, int n_u, const Anchor *a, bool is_qstrand, RegVec &out); // hit.c:52-88
void split_reg(Reg &r, Reg &r2, int n, int qlen, const Anchor *a, bool is_qstrand);                // hit.c:106-123
void set_parent(float mask_level, int mask_len, RegVec &r, int sub_diff, bool hard_mask_level, float alt_diff_frac); // hit.c:125-186
void hit_sort(RegVec &r, float alt_diff_frac);                                                      // hit.c:188-218
int set_sam_pri(RegVec &r);                                                                         // hit.c:220-229
void sync_regs(RegVec &r);                                                                          // hit.c:231-253
void select_sub(float pri_ratio, int min_diff, int best_n, bool check_strand, int min_strand_sc, RegVec &r); // hit.c:255-281
void filter_strand_retained(RegVec &r);                                                             // hit.c:283-299
void filter_regs(const ref::MapOpt &opt, int qlen, RegVec &r);                                      // hit.c:301-320
int squeeze_anchors(RegVec &r, Anchor *a);                                                          // hit.c:322-340
void set_mapq(RegVec &r, int min_chain_sc, int match_sc, int rep_len, bool is_sr, bool is_splice);  // hit.c:432-485
void est_err(const FlatIndex &fi, int qlen, RegVec &r, const Anchor *a, const uint64_t *mini_pos, int32_t n_mini_pos); // esterr.c:30-64
// two-segment fragments (paired-end reads)
void seg_gen(uint32_t hash, int n_segs, const int *qlens, const RegVec &regs0, const Anchor *a, RegVec *regs, std::vector<Anchor> *seg_a); // hit.c:342-396
void select_sub_multi(float pri_ratio, float pri1, float pri2, int max_gap_ref, int min_diff, int best_n, int n_segs, const int *qlens, RegVec &r); // pe.c:6-50
void pair_hits(int max_gap_ref, int pe_bonus, int sub_diff, int match_sc, const int *qlens, RegVec *regs); // mm_pair, pe.c:81-182
void update_dp_max(int qlen, RegVec &r, float frac, int a, int b);                                  // align.c:1022-1046

} // namespace mm2amd

// Option presets of the per-read path: the host-side mirror of mm_idxopt_init / mm_mapopt_init / mm_set_opt /
// mm_mapopt_update / mm_check_opt (options.c:5-67, :69-83, :91-193, :202-260).  A preset is data: a list of
// (field, value) overrides applied on top of whatever the structs hold, exactly like the reference applies
// `-x` on top of the defaults; the numbers are the reference's, the mechanism is ours.
#include <climits>
#include <cstring>
#include <string>
#include "options.hpp"

namespace mm2amd {

using namespace ref;

namespace {

struct IntSet { int MapOpt::*f; int v; };
struct FltSet { float MapOpt::*f; float v; };

struct Preset {
	const char *names[4];          // aliases
	int k, w, idx_flag;            // -1: leave untouched; idx_flag: -1 untouched, else value; idx_or ORs
	int idx_or;
	int64_t map_or;                // bits OR-ed into mm_mapopt_t::flag
	IntSet ints[20];
	FltSet flts[4];
	int64_t mini_batch, max_sw_mat; // -1: leave untouched
};

#define I(f, v) { &MapOpt::f, v }
const int64_t kAva = F_ALL_CHAINS | F_NO_DIAG | F_NO_DUAL | F_NO_LJOIN;
const int64_t kSplice = F_SPLICE | F_SPLICE_FOR | F_SPLICE_REV | F_SPLICE_FLANK;
const int64_t kSrIo = F_NO_PRINT_2ND | F_2_IO_THREADS | F_HEAP_SORT | F_FRAG_MODE;

const Preset kPresets[] = {
	{ {"lr", "map-ont"}, -1, -1, -1, 0, 0, {}, {}, -1, -1 },
	{ {"ava-ont"}, 15, 5, 0, 0, kAva, { I(min_chain_score, 100), I(max_chain_skip, 25), I(bw, 2000), I(bw_long, 2000), I(occ_dist, 0) }, { I(pri_ratio, 0.0f) }, -1, -1 },
	{ {"map10k", "map-pb"}, 19, -1, -1, I_HPC, 0, {}, {}, -1, -1 },
	{ {"ava-pb"}, 19, 5, -1, I_HPC, kAva, { I(min_chain_score, 100), I(max_chain_skip, 25), I(occ_dist, 0) }, { I(pri_ratio, 0.0f) }, -1, -1 }, // bw_long = bw: see apply()
	{ {"lr:hq"}, 19, 19, 0, 0, 0, { I(max_gap, 10000), I(min_mid_occ, 50), I(max_mid_occ, 500) }, {}, -1, -1 },
	{ {"map-hifi", "map-ccs"}, 19, 19, 0, 0, 0, { I(max_gap, 10000), I(min_mid_occ, 50), I(max_mid_occ, 500), I(a, 1), I(b, 4), I(q, 6), I(q2, 26), I(e, 2), I(e2, 1), I(min_dp_max, 200) }, {}, -1, -1 },
	{ {"lr:hqae"}, 25, 51, 0, 0, F_RMQ, { I(min_mid_occ, 50), I(max_mid_occ, 500), I(rmq_inner_dist, 5000), I(occ_dist, 200), I(best_n, 100) }, { I(chain_gap_scale, 5.0f) }, -1, -1 },
	{ {"map-iclr-prerender"}, 15, -1, 0, 0, 0, { I(b, 6), I(transition, 1), I(q, 10), I(q2, 50) }, {}, -1, -1 },
	{ {"map-iclr"}, 19, -1, 0, 0, 0, { I(b, 6), I(transition, 4), I(q, 10), I(q2, 50) }, {}, -1, -1 },
	{ {"asm5"}, 19, 19, 0, 0, F_RMQ, { I(bw, 1000), I(bw_long, 100000), I(max_gap, 10000), I(min_mid_occ, 50), I(max_mid_occ, 500), I(min_dp_max, 200), I(best_n, 50),
	               I(a, 1), I(b, 19), I(q, 39), I(q2, 81), I(e, 3), I(e2, 1), I(zdrop, 200), I(zdrop_inv, 200) }, {}, -1, -1 },
	{ {"asm10"}, 19, 19, 0, 0, F_RMQ, { I(bw, 1000), I(bw_long, 100000), I(max_gap, 10000), I(min_mid_occ, 50), I(max_mid_occ, 500), I(min_dp_max, 200), I(best_n, 50),
	               I(a, 1), I(b, 9), I(q, 16), I(q2, 41), I(e, 2), I(e2, 1), I(zdrop, 200), I(zdrop_inv, 200) }, {}, -1, -1 },
	{ {"asm20"}, 19, 10, 0, 0, F_RMQ, { I(bw, 1000), I(bw_long, 100000), I(max_gap, 10000), I(min_mid_occ, 50), I(max_mid_occ, 500), I(min_dp_max, 200), I(best_n, 50),
	               I(a, 1), I(b, 4), I(q, 6), I(q2, 26), I(e, 2), I(e2, 1), I(zdrop, 200), I(zdrop_inv, 200) }, {}, -1, -1 },
	{ {"short", "sr"}, 21, 11, 0, 0, F_SR | kSrIo, { I(pe_ori, 1), I(a, 2), I(b, 8), I(q, 12), I(e, 2), I(q2, 24), I(e2, 1), I(zdrop, 100), I(zdrop_inv, 100), I(end_bonus, 10),
	               I(max_frag_len, 800), I(max_gap, 100), I(bw, 100), I(bw_long, 100), I(min_cnt, 2), I(min_chain_score, 25), I(min_dp_max, 40), I(best_n, 20), I(mid_occ, 1000), I(max_occ, 5000) },
	               { I(pri_ratio, 0.5f) }, 50000000, -1 },
	{ {"splice", "cdna"}, 15, 5, 0, 0, kSplice, { I(max_gap, 2000), I(max_gap_ref, 200000), I(bw, 200000), I(bw_long, 200000), I(a, 1), I(b, 2), I(q, 2), I(e, 1), I(q2, 32), I(e2, 0),
	               I(noncan, 9), I(junc_bonus, 9), I(junc_pen, 5), I(zdrop, 200), I(zdrop_inv, 100) }, {}, -1, 0 },
	{ {"splice:hq"}, 15, 5, 0, 0, kSplice, { I(max_gap, 2000), I(max_gap_ref, 200000), I(bw, 200000), I(bw_long, 200000), I(a, 1), I(b, 4), I(q, 6), I(e, 1), I(q2, 24), I(e2, 0),
	               I(noncan, 5), I(junc_bonus, 9), I(junc_pen, 5), I(zdrop, 200), I(zdrop_inv, 100) }, {}, -1, 0 },
	{ {"splice:sr"}, 15, 5, 0, 0, kSplice | kSrIo | F_WEAK_PAIRING | F_SR_RNA, { I(max_gap, 2000), I(max_gap_ref, 200000), I(bw, 200000), I(bw_long, 200000), I(a, 1), I(b, 4), I(q, 6), I(e, 1),
	               I(q2, 24), I(e2, 0), I(noncan, 5), I(junc_bonus, 9), I(junc_pen, 5), I(zdrop, 200), I(zdrop_inv, 100), I(min_chain_score, 25), I(min_dp_max, 40), I(min_ksw_len, 20),
	               I(pe_ori, 1), I(best_n, 10) }, {}, 100000000, 0 },
};
#undef I

void apply(const Preset &p, IdxOpt *io, MapOpt *mo)
{
	if (p.k >= 0) io->k = (short)p.k;
	if (p.w >= 0) io->w = (short)p.w;
	if (p.idx_flag >= 0) io->flag = (short)p.idx_flag;
	io->flag |= (short)p.idx_or;
	mo->flag |= p.map_or;
	for (const IntSet &s : p.ints) if (s.f) mo->*(s.f) = s.v;
	for (const FltSet &s : p.flts) if (s.f) mo->*(s.f) = s.v;
	if (p.mini_batch >= 0) mo->mini_batch_size = p.mini_batch;
	if (p.max_sw_mat >= 0) mo->max_sw_mat = p.max_sw_mat;
	if (strcmp(p.names[0], "ava-pb") == 0) mo->bw_long = mo->bw;
}

} // namespace

void idxopt_init(IdxOpt *io) // options.c:5-12
{
	memset(io, 0, sizeof *io);
	io->k = 15, io->w = 10, io->bucket_bits = 14;
	io->mini_batch_size = 50000000, io->batch_size = 8000000000ULL;
}

void mapopt_init(MapOpt *o) // options.c:14-67
{
	memset(o, 0, sizeof *o);
	o->seed = 11;
	o->mid_occ_frac = 2e-4f, o->min_mid_occ = 10, o->max_mid_occ = 1000000, o->q_occ_frac = 0.01f;
	o->max_max_occ = 4095, o->occ_dist = 500;
	o->min_cnt = 3, o->min_chain_score = 40;
	o->bw = 500, o->bw_long = 20000, o->max_gap = 5000, o->max_gap_ref = -1;
	o->max_chain_skip = 25, o->max_chain_iter = 5000;
	o->rmq_inner_dist = 1000, o->rmq_size_cap = 100000, o->rmq_rescue_size = 1000, o->rmq_rescue_ratio = 0.1f;
	o->chain_gap_scale = 0.8f, o->chain_skip_scale = 0.0f;
	o->mask_level = 0.5f, o->mask_len = INT_MAX, o->pri_ratio = 0.8f, o->best_n = 5, o->alt_drop = 0.15f;
	o->a = 2, o->b = 4, o->q = 4, o->e = 2, o->q2 = 24, o->e2 = 1, o->sc_ambi = 1;
	o->zdrop = 400, o->zdrop_inv = 200, o->end_bonus = -1;
	o->min_dp_max = o->min_chain_score * o->a, o->min_ksw_len = 200;
	o->anchor_ext_len = 20, o->anchor_ext_shift = 6, o->max_clip_ratio = 1.0f;
	o->mini_batch_size = 500000000, o->max_sw_mat = 100000000, o->cap_kalloc = 500000000;
	o->rank_min_len = 500, o->rank_frac = 0.9f;
	o->pe_ori = 0, o->pe_bonus = 33, o->jump_min_match = 3;
}

int set_opt(const char *preset, IdxOpt *io, MapOpt *mo) // options.c:91-193
{
	if (!preset) { idxopt_init(io); mapopt_init(mo); return 0; }
	for (const Preset &p : kPresets)
		for (const char *nm : p.names)
			if (nm && strcmp(nm, preset) == 0) { apply(p, io, mo); return 0; }
	return -1;
}

void mapopt_update(MapOpt *o, int32_t (*cal_max_occ)(const void *, float), const void *idx) // options.c:69-83
{
	if (o->flag & (F_SPLICE_FOR | F_SPLICE_REV)) o->flag |= F_SPLICE;
	if (o->mid_occ <= 0) {
		o->mid_occ = cal_max_occ(idx, o->mid_occ_frac);
		if (o->mid_occ < o->min_mid_occ) o->mid_occ = o->min_mid_occ;
		if (o->max_mid_occ > o->min_mid_occ && o->mid_occ > o->max_mid_occ) o->mid_occ = o->max_mid_occ;
	}
	if (o->bw_long < o->bw) o->bw_long = o->bw;
}

// The subset of mm_check_opt (options.c:202-260) that concerns the per-read path; returns 0 or the reference's negative code.
int check_opt(const IdxOpt *io, const MapOpt *mo, std::string *why)
{
	auto bad = [&](int code, const char *msg) { if (why) *why = msg; return code; };
	if (mo->bw > mo->bw_long) return bad(-8, "with '-rNUM1,NUM2', NUM1 (chaining bandwidth) should be NO larger than NUM2 (long-join bandwidth)");
	if ((mo->flag & F_RMQ) && (mo->flag & (F_SR | F_SPLICE))) return bad(-7, "--rmq doesn't work with --sr or --splice");
	if (io->k <= 0 || io->w <= 0) return bad(-5, "-k and -w must be positive");
	if (mo->best_n < 0) return bad(-4, "-N must be no less than 0");
	if (mo->pri_ratio < 0.0f || mo->pri_ratio > 1.0f) return bad(-4, "-p must be within 0 and 1 (including 0 and 1)");
	if ((mo->flag & F_FOR_ONLY) && (mo->flag & F_REV_ONLY)) return bad(-3, "--for-only and --rev-only can't be applied at the same time");
	if (mo->e <= 0 || mo->q <= 0) return bad(-1, "-O and -E must be positive");
	if ((mo->q != mo->q2 || mo->e != mo->e2) && !(mo->e > mo->e2 && mo->q + mo->e < mo->q2 + mo->e2)) return bad(-2, "dual gap penalties violating E1>E2 and O1+E1<O2+E2");
	if ((mo->q + mo->e) + (mo->q2 + mo->e2) > 127) return bad(-1, "scoring system violating ({-O}+{-E})+({-O2}+{-E2}) <= 127");
	if (mo->sc_ambi < 0 || mo->sc_ambi >= mo->b) return bad(-1, "--score-N should be within [0,{-B})");
	if (mo->zdrop < mo->zdrop_inv) return bad(-5, "Z-drop should not be less than inversion-Z-drop");
	return 0;
}

} // namespace mm2amd

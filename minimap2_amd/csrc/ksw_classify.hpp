// Launch classes of the batched DP (ksw_host.cpp) as one function of a job record, shared by the host's job ordering and by the device's
// (ksw_order.hip: when the jobs are born on the device -- region_plan_kernel -- they are classed, counted and put in launch order there,
// and only the per-class sizing figures come to the host, which plans the launches).
#pragma once
#include <cstdint>
#include <cstddef>
#include <cmath>
#include "ksw_dev.hpp"
#include "exact_rsort.hpp" // MM2_HD
#include "ksw_band.hpp"

namespace mm2amd {

// Launch classes.  0..5: the register-resident gap-fill kernels (ksw_gapfill.hip / ksw_stream.hip), classed by query capacity (512 / 1024 bytes of
// LDS per job: eight / four waves per SIMD) and by target length (one strip of 256 columns, up to 2-4 strips, more), so that the
// two jobs of a wave have the same strip count and a class's direction-matrix slots are not sized by its rare giants.
// 6..: the lane-exact kernel (ksw_extd2.hip), classed by (a) the size of its state window -- rings of 256..8192 positions in
// LDS, 13 B per position, or any size in HBM; a job needs min(qlen, tlen, band) + 64 positions -- and (b) the size of its
// direction matrix, because every persistent wave owns a scratch slot as large as the biggest matrix of its class.
constexpr int kFirstExact = 6, kRingClasses = 7, kDirClasses = 11, kFirstSplice = kFirstExact + kRingClasses * kDirClasses;
// kFirstSplice..: the register-resident splice gap-fill kernel (ksw_splice.hip): two jobs per wave with 2 or 4 register sets of
// 64 QUERY positions (queries up to 128 / 256), or one job per wave using both register halves of 4 sets (512 positions per
// sweep over the target, longer queries in several sweeps); classed by direction-matrix size like the exact kernel.
constexpr int kSpliceClasses = 3, kFirstExt = kFirstSplice + kSpliceClasses * kDirClasses;
// kFirstExt..: the register-resident extension kernels.  Round 6 (ksw_extq.hip, the query across the lanes): + 0/1: queries up to 128 (left- / right-aligned gaps),
// + 2/3: up to 256, + 4/5: up to 512, targets up to kExtqMaxT.  MM2AMD_EXT_BY_TARGET=1 (A/B): round 3's ksw_ext.hip, the target across the lanes: + 0/1: targets up to 256,
// + 2/3: up to 512, queries up to 512.
constexpr int kExtClasses = 6, kFirstBand = kFirstExt + kExtClasses, kExtMaxQ = 512, kExtMaxT = 512, kExtqMaxT = 2048;
// kFirstBand..: the banded gap-fill kernel (ksw_band.hip, round 6): + 0: a band of 128 diagonals (one register set), + 1: 256 diagonals (two), both for windows up to
// 512 x 512; + 2: 512 diagonals (four sets), windows up to 1024 x 1024.  A gap fill goes here when the score its length lets one expect would prove the band sufficient (ksw_band.hpp); what the
// kernel cannot prove is computed again in the wider band or as the full rectangle, so the choice is a matter of speed only.
constexpr int kBandClasses = 3, kNTiers = kFirstBand + kBandClasses, kBandMaxSmall = 512, kBandMaxBig = 1024;
constexpr int kHbmRing = kRingClasses - 1; // the last ring class keeps its state in HBM and takes any width
constexpr int kFastMaxQ = 1024, kFastMaxTAny = 3072;
constexpr int kOrderBuckets = 256; // cost buckets per class

struct KswClassCtx { // uniform over a batch
	int scoring_ok, splice_ok, splice, stream_on, ext_on, ext_max_t;
	int ext_by_target = 0; // 1: the extension classes of ksw_ext.hip (A/B)
	int merge_rings = 1;   // the lane-exact kernel's ring classes 512 / 1024 / 2048 as ONE launch class (ring 2048): see ksw_classify()
	int ext_max_q = 512;   // longer queries go to the lane-exact kernel (see ksw_host.cpp for why 256 is the default with ksw_extq.hip)
	// the banded kernel: on / off; the scores the acceptance test works with; the share of the best possible score (sc_max per base of the shorter side, in
	// 1/256) a window is EXPECTED to reach -- the classes are chosen with it, the kernel's test uses the score actually found
	int band_on = 0, sc_max = 0, gq = 0, ge = 0, gq2 = 0, ge2 = 0, band_rho256 = 128;
	int band_max = kBandMaxBig; // windows up to this on either side may try a band (512: the four-set class is off, A/B)
};
struct KswClassOut {
	int tier, cb;            // launch class, cost bucket (higher = launched earlier)
	size_t db;               // direction-matrix bytes of the job
	int ring_need;
	bool live, fast, xfast, sfast;
};

MM2_HD inline int ksw_ring_size(int rc) { return rc == 0 ? 256 : rc == 1 ? 512 : rc == 2 ? 1024 : rc == 3 ? 2048 : rc == 4 ? 4096 : rc == 5 ? 8192 : 0; }
MM2_HD inline int ksw_splice_max_q(int nc) { return nc == 0 ? 128 : nc == 1 ? 256 : 1 << 30; }
MM2_HD inline int ksw_fast_max_t(int t) { return t == 0 ? 256 : t == 1 ? 512 : t == 2 ? 1536 : t == 3 ? 256 : t == 4 ? 1024 : 3072; } // <= 3 * query capacity: the kernel's LDS holds the target bytes for the Z-drop scan
MM2_HD inline size_t ksw_dir_limit(int dc) { return dc == kDirClasses - 1 ? SIZE_MAX : (size_t)256 << (10 + dc); } // 256 KB, 512 KB, ... 128 MB, any
MM2_HD inline int ksw_stream_sets(int tier) { return tier == 0 ? 4 : tier == 1 ? 8 : 0; } // classes the streaming kernel takes (query <= 512, target <= 64 * sets); 0: the strip kernel
MM2_HD inline int ksw_fast_tier(const KswJob &j) { int t = j.qlen <= 512 && j.tlen <= 1536 ? 0 : 3; while (j.tlen > ksw_fast_max_t(t)) ++t; return t; }

// ksw_extd2_sse limits anti-diagonal r to t in [max(0, r - qlen + 1, (r - w + 1) >> 1), min(tlen - 1, r, (r + w) >> 1)] (ksw2_extd2_sse.c:139-146).
// The band terms never decide when (r - w + 1) >> 1 <= max(0, r - qlen + 1) and (r + w) >> 1 >= min(r, tlen - 1) for every r, i.e. when
// w >= qlen - 1 and w >= tlen - 1: the row limits, and with them every boundary value the reference picks (:148-163), are then those of an
// unbanded call.  (Rounds 1-2 used the sufficient w >= qlen + tlen, which sent every extension longer than 375 + 376 to the lane-exact kernel.)
MM2_HD inline bool ksw_band_cannot_bind(const KswJob &j) { return j.w < 0 || ((int64_t)j.w + 1 >= j.qlen && (int64_t)j.w + 1 >= j.tlen); }
// A job may take the register-resident kernel when nothing but valid cells can matter: global alignment with the approximate
// score (the gap-fill call, align.c:838), default substitution scores, and a band that cannot bind.
MM2_HD inline bool ksw_fast_eligible(const KswJob &j, bool scoring_ok)
{
	if (!scoring_ok || (j.flag & 0x1fff) != KSW_APPROX_MAX || (j.flag & KSWJ_SKIP)) return false;
	if (j.qlen <= 0 || j.tlen <= 0 || j.qlen > kFastMaxQ || j.tlen > kFastMaxTAny) return false;
	return ksw_band_cannot_bind(j);
}
// The splice gap fill (align.c:840 with -x splice) may take the register-resident splice kernel: global alignment with the
// approximate score, default substitution scores, forward CIGAR, no junction scores; the scoring must
// keep every intermediate of a valid cell inside 8 bits (what the reference's int8 lanes assume).
MM2_HD inline bool ksw_splice_fast_eligible(const KswJob &j, bool scoring_ok)
{
	constexpr int kSpliceBits = KSW_SPLICE_FOR | KSW_SPLICE_REV | KSW_SPLICE_FLANK | KSW_SPLICE_CMPLX;
	if (!scoring_ok || ((j.flag & 0x1fff) & ~kSpliceBits) != KSW_APPROX_MAX || (j.flag & KSWJ_SKIP)) return false;
	if (j.reserved) return false; // windows with annotated splice sites (KswScoring::juncs) are priced by the lane-exact kernel only
	return j.qlen > 0 && j.tlen > 0;
}
// An extension (align.c:791, :883: KSW_EZ_EXTZ_ONLY; left extensions also KSW_EZ_RIGHT | KSW_EZ_REV_CIGAR) may take the register-resident
// extension kernel when its band cannot bind, with default substitution scores and dual-affine costs: exact row maxima, Z-drop and end
// bonus are computed there (ksw_ext.hip).
MM2_HD inline bool ksw_ext_eligible(const KswJob &j, bool scoring_ok, int max_t, int max_q = kExtMaxQ)
{
	const int f = j.flag & 0x1fff;
	if (!scoring_ok || (j.flag & KSWJ_SKIP) || (f != KSW_EXTZ_ONLY && f != (KSW_EXTZ_ONLY | KSW_RIGHT | KSW_REV_CIGAR))) return false;
	if (j.qlen <= 0 || j.tlen <= 0 || j.qlen > kExtMaxQ || j.qlen > max_q || j.tlen > max_t) return false;
	return ksw_band_cannot_bind(j);
}
MM2_HD inline int ksw_pow2ceil(int v) { int p = 64; while (p < v) p <<= 1; return p; }
MM2_HD inline int ksw_band_sets(int tier) { return tier == kFirstBand ? 1 : tier == kFirstBand + 1 ? 2 : tier == kFirstBand + 2 ? 4 : 0; }
MM2_HD inline int ksw_band_class(int sets) { return sets == 4 ? 2 : sets - 1; } // register sets -> class index
// the narrowest band class whose acceptance test the window's expected score passes: 1, 2 or 4 register sets, 0 = none (the rectangle is the better bet)
MM2_HD inline int ksw_band_choice(const KswJob &j, const KswClassCtx &C)
{
	if (!C.band_on || j.qlen > C.band_max || j.tlen > C.band_max) return 0;
	const bool big = j.qlen > kBandMaxSmall || j.tlen > kBandMaxSmall;
	const int mn = j.qlen < j.tlen ? j.qlen : j.tlen, D = j.tlen - j.qlen;
	const int expect = (int)(((int64_t)C.band_rho256 * C.sc_max * mn) >> 8) - band_gap_cost(D < 0 ? -D : D, C.gq, C.ge, C.gq2, C.ge2);
	for (int nb = big ? 4 : 1; nb <= (big ? 4 : 2); nb *= 2) // (a window of at most 512 x 512 that 256 diagonals are not expected to do for: the rectangle there is two sweeps of the four sets)
		if (band_holds_corners(j.qlen, j.tlen, 128 * nb) && expect > band_outside_bound(j.qlen, j.tlen, 128 * nb, C.sc_max, C.gq, C.ge, C.gq2, C.ge2)) return nb;
	return 0;
}

MM2_HD inline void ksw_classify(const KswJob &j, const KswClassCtx &C, KswClassOut &o)
{
	const bool splice = C.splice != 0;
	o.ring_need = 64;
	o.fast = ksw_fast_eligible(j, C.scoring_ok != 0), o.sfast = ksw_splice_fast_eligible(j, C.splice_ok != 0), o.xfast = C.ext_on && ksw_ext_eligible(j, C.scoring_ok != 0, C.ext_max_t, C.ext_max_q);
	o.live = !(j.flag & KSWJ_SKIP) && j.qlen > 0 && j.tlen > 0;
	o.db = !o.live || (j.flag & KSW_SCORE_ONLY) ? 0 : o.xfast && !C.ext_by_target ? (size_t)(j.qlen + j.tlen - 1) * (size_t)(j.qlen > 256 ? 512 : j.qlen > 128 ? 256 : 128) :
	       o.fast || o.xfast ? (size_t)(j.qlen + j.tlen - 1) * (size_t)((j.tlen + 63) & ~63) :
	       o.sfast ? (size_t)(j.qlen + j.tlen - 1) * (size_t)((j.qlen + 63) & ~63) : ksw_dir_bytes(j.qlen, j.tlen, splice ? -1 : j.w);
	const int band_sets = o.fast ? ksw_band_choice(j, C) : 0;
	if (band_sets) o.tier = kFirstBand + ksw_band_class(band_sets), o.db = (size_t)(j.qlen + j.tlen - 1) * (size_t)(64 * band_sets);
	else if (o.fast) o.tier = ksw_fast_tier(j);
	else if (o.xfast && C.ext_by_target) o.tier = kFirstExt + (j.tlen > 256 ? 2 : 0) + ((j.flag & KSW_RIGHT) ? 1 : 0);
	else if (o.xfast) o.tier = kFirstExt + (j.qlen > 256 ? 4 : j.qlen > 128 ? 2 : 0) + ((j.flag & KSW_RIGHT) ? 1 : 0);
	else if (o.sfast) {
		int nc = 0, dc = 0;
		while (j.qlen > ksw_splice_max_q(nc)) ++nc;
		while (o.db > ksw_dir_limit(dc)) ++dc;
		o.tier = kFirstSplice + nc * kDirClasses + dc;
	} else {
		int width = j.qlen < j.tlen ? j.qlen : j.tlen; // widest anti-diagonal
		if (!splice && j.w >= 0 && j.w + 2 < width) width = j.w + 2;
		o.ring_need = ksw_pow2ceil((o.live ? width : 0) + 64);
		int rc = 0, dc = 0;
		while (rc < kHbmRing && o.ring_need > ksw_ring_size(rc)) ++rc;
		// a launch of this kernel holds a few hundred long extensions and lasts as long as its longest job, microseconds per row: three ring classes were three such
		// launches one after the other per sub-batch.  One class at the largest of the three rings (26 KB of LDS per job: five workgroups per CU, far more than a launch
		// has jobs) is one launch (MM2AMD_KSW_SPLIT_RINGS=1: apart, A/B).
		if (C.merge_rings && !splice && rc >= 1 && rc <= 3) rc = 3;
		while (o.db > ksw_dir_limit(dc)) ++dc;
		// banded matrices vary little, and a launch of this kernel lasts as long as its longest job whatever it holds (a few hundred long extensions, microseconds per
		// row): ONE launch per ring class for everything up to 16 MB (round 5: 256 KB / 2 MB / 16 MB apart -- four launches per sub-batch and ring class, each ~3 ms of
		// latency; the slots are sized by the class's largest matrix, 1-2 MB on long-read batches), the rare giants on their own
		if (!splice) dc = dc <= 6 ? 6 : kDirClasses - 1;
		o.tier = kFirstExact + rc * kDirClasses + dc;
	}
	// launch order inside a class: cost (rows * row width) roughly descending -- longest-job-first for the persistent waves
	const int mn = j.qlen < j.tlen ? j.qlen : j.tlen, bwid = j.w < 0 || splice ? INT32_MAX : j.w + 1;
	const double cost = (j.flag & KSWJ_SKIP) ? 0.0 : (double)(j.qlen + j.tlen) * (double)(mn < bwid ? mn : bwid);
	int cb = o.fast && j.tlen <= 512 && j.qlen <= 512 ? (int)(sqrt(cost) * 0.25) : (int)(8.0 * log2(cost + 1.0)); // small gap fills: cost <= 1024*512; other classes: any (9 % steps)
	if (o.sfast) cb = (int)(12.0 * log2((double)(j.qlen + j.tlen))); // the two jobs of a wave advance row by row: order by row count (6 % steps)
	// the streaming kernel computes, row by row, the register sets its jobs in flight reach: jobs of one width class (64-column
	// sets) together, the widest first; within a class the longest queries first (short ones fill the launch's tail)
	if (o.fast && C.stream_on && ksw_stream_sets(o.tier)) cb = (((j.tlen + 63) / 64 - 1) & 3) * 64 + (j.qlen / 8 < 63 ? j.qlen / 8 : 63);
	if (band_sets) cb = (j.qlen + j.tlen) >> 2; // the two jobs of a wave advance row by row: order by row count
	if (cb >= kOrderBuckets) cb = kOrderBuckets - 1;
	if (cb < 0) cb = 0;
	o.cb = cb;
}

// per-class sizing and accounting figures, gathered over the jobs of a class (query bytes + packed target + job/result
// records; the 1 B/cell direction matrix only counts when it cannot stay on chip, i.e. exceeds 160 KB of LDS)
struct KswClassStat { unsigned long long slot_bytes, tmp_cap, alg_bytes, cells, sum_len; unsigned int max_ring, max_Q16, max_rows, max_ncol, n_jobs, pad; };

// what the host needs to plan the launches when the ordering ran on the device
struct KswOrderResult { KswClassStat cls[kNTiers]; unsigned int tier_beg[kNTiers + 1]; unsigned int ticket; };
// d_jobs (n records, device) -> d_sorted in launch order, d_perm[i] = launch position of job i; *h_out (pinned) is valid after the stream has been waited for.
// d_work: scratch of ksw_order_work_words(n) 32-bit words.
size_t ksw_order_work_words(size_t n);
void ksw_order_device(const KswJob *d_jobs, size_t n, const KswClassCtx &C, KswJob *d_sorted, uint32_t *d_perm, uint32_t *d_work, KswOrderResult *d_out, void *stream);

} // namespace mm2amd

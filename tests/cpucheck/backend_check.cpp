// TEST INFRASTRUCTURE ONLY -- never linked into libmm2amd.so.
//
// A Backend made of the oracle's plain-C restatement (oracle/*.c).  It lets the CPU-only test suite drive the
// product's host pipeline (chains -> hits -> window planning -> CIGAR stitching -> MAPQ) end to end and compare
// its SAM output with the reference binary, in a container that has no GPU.  The GPU tests exercise the same host
// pipeline with the real HipBackend.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include "../../minimap2_amd/csrc/backend.hpp"
#include "../../oracle/oracle.h"
#include "../../minimap2_amd/csrc/sdust_core.hpp"

extern "C" const uint8_t *ora_nt4_table(void);

namespace mm2amd {

namespace {

const uint64_t *flat_get(const void *idx, uint64_t minier, int *n) { return ((const FlatIndex *)idx)->get(minier, n); }
const char *flat_name(const void *idx, uint32_t rid, uint32_t *len)
{
	const FlatIndex *fi = (const FlatIndex *)idx;
	*len = fi->seq_len[rid];
	return fi->names[rid].c_str();
}

class CheckBackend : public Backend {
public:
	explicit CheckBackend(const FlatIndex &fi) : fi_(fi) {}
	void begin_batch(const std::vector<ReadView> &reads, std::vector<uint64_t> &qpool_off) override
	{
		reads_ = reads;
		qpool_off.resize(reads.size());
		size_t tot = 0;
		for (size_t i = 0; i < reads.size(); ++i) qpool_off[i] = tot, tot += 2 * (size_t)reads[i].total();
		qpool_off_ = qpool_off;
		qpool_.assign(tot + 1, 0);
		const uint8_t *nt4 = ora_nt4_table();
		for (size_t i = 0; i < reads.size(); ++i) { // every read of a pair gets its own  forward | reverse-complement  block
			for (int s = 0; s < (reads[i].paired() ? 2 : 1); ++s) {
				uint8_t *f = &qpool_[qpool_off[i] + (s ? 2 * (size_t)reads[i].len : 0)];
				const char *seq = s ? reads[i].seq2 : reads[i].seq;
				const int len = s ? reads[i].len2 : reads[i].len;
				for (int j = 0; j < len; ++j) {
					const uint8_t c = nt4[(uint8_t)seq[j]];
					f[j] = c, f[2 * (size_t)len - 1 - j] = c < 4 ? 3 - c : 4;
				}
			}
		}
	}
	void seed_chain(const SeedChainParams &p, long lo, long hi, int /*lane*/, int /*n_threads*/, std::vector<ReadChains> &out) override
	{
		out.clear();
		out.resize((size_t)(hi - lo));
		std::vector<ora128_t> mv;
		for (size_t i = (size_t)lo; i < (size_t)hi; ++i) {
			const int len = reads_[i].total(), n_seg = reads_[i].paired() ? 2 : 1;
			mv.resize((size_t)len + 2);
			int64_t n_mv = ora_sketch(reads_[i].seq, reads_[i].len, p.w, p.k, 0, p.is_hpc, mv.data(), (int64_t)mv.size());
			if (n_seg == 2) { // collect_minimizers (map.c:59-72): segment id in the rid field, positions offset by the earlier segments
				const int64_t n1 = ora_sketch(reads_[i].seq2, reads_[i].len2, p.w, p.k, 1, p.is_hpc, mv.data() + n_mv, (int64_t)mv.size() - n_mv);
				for (int64_t j = n_mv; j < n_mv + n1; ++j) mv[j].y += (uint64_t)reads_[i].len << 1;
				n_mv += n1;
			}
			if (p.sdust_thres > 0) { // mm_dust_minier per read, after the second read's positions have been shifted (map.c:64-69)
				int64_t base = 0, kept = 0;
				for (int sgm = 0; sgm < n_seg; ++sgm) {
					const int slen = sgm ? reads_[i].len2 : reads_[i].len;
					const uint8_t *codes = &qpool_[qpool_off_[i] + (sgm ? 2 * (size_t)reads_[i].len : 0)];
					int64_t n_this = 0;
					while (base + n_this < n_mv && (int)(mv[base + n_this].y >> 32) == sgm) ++n_this;
					std::vector<int32_t> rs, re;
					std::vector<SdustState::Perf> pbuf(SdustState::PCAP);
					SdustState S;
					S.P = pbuf.data();
					sdust_scan(codes, slen, p.sdust_thres, S, [&](int st, int en) { rs.push_back(st), re.push_back(en); });
					std::vector<uint64_t> x(n_this), y(n_this);
					for (int64_t j = 0; j < n_this; ++j) x[j] = mv[base + j].x, y[j] = mv[base + j].y;
					const int k = dust_filter_minimizers((int)n_this, x.data(), y.data(), (int)rs.size(),
						[&](int u, int32_t *st, int32_t *en) { *st = rs[u], *en = re[u]; }, 0);
					for (int j = 0; j < k; ++j) mv[kept + j].x = x[j], mv[kept + j].y = y[j];
					kept += k, base += n_this;
				}
				n_mv = kept;
			}
			ora128_t *a = nullptr;
			uint64_t *mp = nullptr;
			int64_t n_a = 0;
			int n_mp = 0, rep_len = 0;
			ora_collect_seed_hits_named(&fi_, flat_get, reads_[i].name, fi_.names.empty() ? nullptr : flat_name, p.flag, len, p.q_mid_occ, p.mid_occ, p.max_max_occ, p.occ_dist, p.q_occ_frac, mv.data(), n_mv, &a, &n_a, &mp, &n_mp, &rep_len);
			if (const char *dump = getenv("MM2AMD_CHECK_SEED_DUMP")) { // the anchors as the reference's --print-seeds shows them (map.c:255-260)
				FILE *fp = fopen(dump, "a");
				fprintf(fp, "QR\t%s\t%d\nRS\t%d\n", reads_[i].name ? reads_[i].name : "*", p.mid_occ, rep_len);
				for (int64_t j = 0; j < n_a; ++j)
					fprintf(fp, "SD\t%s\t%d\t%c\t%d\t%d\t%d\n", fi_.names[a[j].x << 1 >> 33].c_str(), (int32_t)a[j].x, "+-"[a[j].x >> 63], (int32_t)a[j].y, (int32_t)(a[j].y >> 32 & 0xff),
					        j == 0 ? 0 : ((int32_t)a[j].y - (int32_t)a[j - 1].y) - ((int32_t)a[j].x - (int32_t)a[j - 1].x));
				fclose(fp);
			}
			ReadChains &c = out[i - (size_t)lo];
			c.rep_len = rep_len;
			c.mini_pos.assign(mp, mp + n_mp);
			if (p.anchors_only) {
				c.u.clear(), c.chained = false;
				c.a.resize(n_a);
				if (n_a) memcpy(c.a.data(), a, n_a * sizeof(ora128_t));
				c.view_own();
				free(a); free(mp);
				continue;
			}
			c.u.resize(n_a > 0 ? n_a : 1);
			int64_t n_kept = 0;
			int gap_ref, gap_qry;
			chain_gaps(p, len, &gap_ref, &gap_qry);
			const int n_u = ora_lchain_dp(gap_ref, gap_qry, p.bw, p.max_chain_skip, p.max_chain_iter, p.min_cnt, p.min_chain_score,
			                              p.chn_pen_gap, p.chn_pen_skip, p.is_cdna, n_seg, n_a, a, c.u.data(), &n_kept);
			c.u.resize(n_u);
			c.a.resize(n_kept);
			if (n_kept) memcpy(c.a.data(), a, n_kept * sizeof(ora128_t));
			c.view_own();
			free(a); free(mp);
		}
	}
	// MM2AMD_CHECK_BAND_STATS=file (measurement): for every gap-fill window, how many cells of its q x t rectangle can NO alignment
	// at least as good as a known one pass through?  The known one is the best global alignment inside a narrow band (16 + the length difference: what a
	// cheap first pass would give); a path through cell (i, j) scores at most  a * (min(i, j) + min(q - i, t - j)) - gap(|i - j|) - gap(|(q - i) - (t - j)|)
	// (all other columns matching, the unavoidable gaps at the dual affine cost).  One line per window: qlen tlen w score narrow_score cells_below_narrow
	// cells_below_optimum, then the symmetric band that holds the remaining cells, its cells, and whether the reference's routine gives the same result in it.
	const char *band_stats_ = getenv("MM2AMD_CHECK_BAND_STATS");
	void band_stats(const KswJob &j, const KswScoring &sc, const uint8_t *q, const uint8_t *t, int score)
	{
		const int ql = j.qlen, tl = j.tlen, diff = ql > tl ? ql - tl : tl - ql;
		ora_ez_t ezn;
		std::vector<uint32_t> cg((size_t)ql + tl + 8);
		ora_ksw_extd2(ql, q, tl, t, sc.m, sc.mat, sc.q, sc.e, sc.q2, sc.e2, 16 + diff, -1, -1, KSW_APPROX_MAX, &ezn, cg.data(), (int)cg.size());
		const int a = sc.mat[0];
		auto gap = [&](int d) { if (d == 0) return 0; const int g1 = sc.q + sc.e * d, g2 = sc.q2 + sc.e2 * d; return g1 < g2 ? g1 : g2; };
		long below_narrow = 0, below_opt = 0;
		int reach = 0; // the widest |i - j| among the cells that are not excluded
		for (int i = 0; i <= ql; ++i)
			for (int k = 0; k <= tl; ++k) {
				const int before = i < k ? i : k, after = ql - i < tl - k ? ql - i : tl - k;
				const int d0 = i > k ? i - k : k - i, d1 = (ql - i) > (tl - k) ? (ql - i) - (tl - k) : (tl - k) - (ql - i);
				const int ub = a * (before + after) - gap(d0) - gap(d1);
				below_narrow += ub < ezn.score, below_opt += ub < score;
				if (ub >= ezn.score && d0 > reach) reach = d0;
			}
		// the reference's own routine with the band that holds those cells (+ 2): the same score and CIGAR as with the window's band?  cells of that band
		const int w2 = reach + 2;
		ora_ez_t ez2, ez1;
		std::vector<uint32_t> cg2((size_t)ql + tl + 8);
		ora_ksw_extd2(ql, q, tl, t, sc.m, sc.mat, sc.q, sc.e, sc.q2, sc.e2, j.w, j.zdrop, j.end_bonus, j.flag & 0x1fff, &ez1, cg.data(), (int)cg.size());
		ora_ksw_extd2(ql, q, tl, t, sc.m, sc.mat, sc.q, sc.e, sc.q2, sc.e2, w2, j.zdrop, j.end_bonus, j.flag & 0x1fff, &ez2, cg2.data(), (int)cg2.size());
		const bool same = ez1.score == ez2.score && ez1.n_cigar == ez2.n_cigar && ez1.zdropped == ez2.zdropped && memcmp(cg.data(), cg2.data(), (size_t)ez1.n_cigar * 4) == 0;
		long band_cells = 0;
		for (int i = 0; i < ql; ++i) { const int lo = i - w2 > 0 ? i - w2 : 0, hi = i + w2 < tl - 1 ? i + w2 : tl - 1; if (hi >= lo) band_cells += hi - lo + 1; }
		static std::mutex mu;
		std::lock_guard<std::mutex> lk(mu);
		if (FILE *fp = fopen(band_stats_, "a")) {
			fprintf(fp, "%d\t%d\t%d\t%d\t%d\t%ld\t%ld\t%d\t%ld\t%d\n", ql, tl, j.w, score, ezn.score, below_narrow, below_opt, w2, band_cells, same ? 1 : 0);
			fclose(fp);
		}
	}
	void ksw(const std::vector<KswJob> &jobs, const KswScoring &sc, int /*lane*/, int /*n_threads*/, std::vector<KswRes> &res, const uint32_t **cigar_out) override
	{
		std::vector<uint32_t> &cigar = cigar_store_;
		res.resize(jobs.size());
		cigar.clear();
		std::vector<uint8_t> q, t, jbuf;
		std::vector<uint32_t> cg;
		for (size_t k = 0; k < jobs.size(); ++k) {
			const KswJob &j = jobs[k];
			KswRes &r = res[k];
			q.resize(j.qlen > 0 ? j.qlen : 0), t.resize(j.tlen > 0 ? j.tlen : 0);
			for (int i = 0; i < j.qlen; ++i) q[i] = qpool_[(j.flag & KSWJ_Q_REVERSED) ? j.q_off - i : j.q_off + i];
			for (int i = 0; i < j.tlen; ++i) {
				const uint64_t pos = (j.flag & KSWJ_T_REVERSED) ? j.t_off - i : j.t_off + i;
				t[i] = (j.flag & KSWJ_T_PACKED) ? (uint8_t)(fi_.S[pos >> 3] >> ((pos & 7) << 2) & 0xf) : sc.tbytes[pos];
			}
			ora_ez_t ez;
			if (j.flag & KSWJ_SKIP) {
				memset(&ez, 0, sizeof ez);
				ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1, ez.score = ez.mqe = ez.mte = ORA_NEG_INF, ez.zdropped = 1;
			} else {
				cg.resize((size_t)j.qlen + j.tlen + 8);
				if (sc.single == 2) {
					const uint8_t *junc = nullptr;
					if (j.flag & KSW_SPLICE_SCORE) { // the window's junc[] as mm_idx_spsc_get fills it: 0xff where no score is known
						jbuf.assign((size_t)j.tlen, 0xff);
						for (uint32_t e = 0; e < j.reserved; ++e) {
							const uint32_t v = sc.juncs[j.tag + e], pos = v >> 8;
							jbuf[(j.flag & KSWJ_T_REVERSED) ? (uint32_t)j.tlen - 1 - pos : pos] = (uint8_t)(v & 0xff);
						}
						junc = jbuf.data();
					} else if (j.reserved) { // the window's junc[] as mm_idx_bed_junc fills it, in the order the job reads the target
						jbuf.assign((size_t)j.tlen, 0);
						for (uint32_t e = 0; e < j.reserved; ++e) {
							const uint32_t v = sc.juncs[j.tag + e], pos = v >> 4;
							jbuf[(j.flag & KSWJ_T_REVERSED) ? (uint32_t)j.tlen - 1 - pos : pos] |= (uint8_t)(v & 15u);
						}
						junc = jbuf.data();
					}
					ora_ksw_exts2(j.qlen, q.data(), j.tlen, t.data(), sc.m, sc.mat, sc.q, sc.e, sc.q2, sc.noncan, j.zdrop, j.end_bonus, sc.junc_bonus, sc.junc_pen, j.flag & 0x1fff, junc,
					              &ez, cg.data(), (int)cg.size());
				}
				else if (sc.single) ora_ksw_extz2(j.qlen, q.data(), j.tlen, t.data(), sc.m, sc.mat, sc.q, sc.e, j.w, j.zdrop, j.end_bonus, j.flag & 0x1fff, &ez, cg.data(), (int)cg.size());
				else ora_ksw_extd2(j.qlen, q.data(), j.tlen, t.data(), sc.m, sc.mat, sc.q, sc.e, sc.q2, sc.e2, j.w, j.zdrop, j.end_bonus, j.flag & 0x1fff,
				                   &ez, cg.data(), (int)cg.size());
			}
			if (band_stats_ && !sc.single && !(j.flag & KSWJ_SKIP) && (j.flag & 0x1fff) == KSW_APPROX_MAX && j.qlen > 0 && j.tlen > 0) band_stats(j, sc, q.data(), t.data(), ez.score);
			r.max = ez.max, r.zdropped = ez.zdropped, r.max_q = ez.max_q, r.max_t = ez.max_t, r.mqe = ez.mqe, r.mqe_t = ez.mqe_t;
			r.mte = ez.mte, r.mte_q = ez.mte_q, r.score = ez.score, r.n_cigar = ez.n_cigar, r.reach_end = ez.reach_end;
			r.cigar_off = (uint32_t)cigar.size();
			r.zd_max = KSW_ZD_NONE, r.zd_t0 = r.zd_t1 = r.zd_q0 = r.zd_q1 = -1;
			cigar.insert(cigar.end(), cg.begin(), cg.begin() + ez.n_cigar);
		}
		*cigar_out = cigar.data();
	}
private:
	const FlatIndex &fi_;
	std::vector<ReadView> reads_;
	std::vector<uint8_t> qpool_;
	std::vector<uint64_t> qpool_off_;
	std::vector<uint32_t> cigar_store_;
};

} // namespace

Backend *make_backend(const FlatIndex &fi, void * /*device_tables*/, int /*n_threads*/, int /*device*/, int /*replica*/, int /*tables_device*/) { return new CheckBackend(fi); }
void *backend_build_index_tables(FlatIndex &, int, int *) { return nullptr; } // the check backend looks minimizers up in the host tables
void backend_free_index_tables(void *) {}
int backend_device_count() { return 16; } // replicas of the check backend are plain objects: any count goes
const char *backend_name() { return "cpu-check(oracle)"; }
// the check library has no device-built index objects
struct IndexHandle;
const FlatIndex &index_flat(const IndexHandle *) { throw std::runtime_error("check backend: no device index"); }
void *index_device_tables(const IndexHandle *) { return nullptr; }
int index_device(const IndexHandle *) { return 0; }
}
extern "C" long long mm2amd_alloc_counter(int) { return 0; }
namespace mm2amd {

} // namespace mm2amd

// TEST HOOK (not part of the product's ABI): the HOST's append_cigar + update_extra (align.cpp: the twin of mm_append_cigar, mm_fix_cigar and
// mm_update_extra the host path uses where region_finish_kernel does not run) on jobs in mm2amd_update_extra_batch's layout, so that the adversarial
// CIGARs of tests/test_gpu_update_extra.py can be put to it and to the reference's own routine (oracle/ref_align_shim.c) side by side.
#include "../../minimap2_amd/csrc/align.hpp"
#include "mm2amd.h"
extern "C" int check_update_extra_host(int n_jobs, const mm2amd_fin_job_t *jobs, const int8_t *mat25, int8_t q, int8_t e, int log_gap, int eqx,
                                       mm2amd_fin_res_t *res, uint32_t *cigar_pool, size_t cigar_pool_cap)
{
	using namespace mm2amd;
	size_t used = 0;
	for (int i = 0; i < n_jobs; ++i) {
		const mm2amd_fin_job_t &j = jobs[i];
		mm2amd_fin_res_t &o = res[i];
		memset(&o, 0, sizeof o);
		Reg r{};
		r.qs = 0, r.qe = j.qlen, r.rs = 0, r.re = j.tlen, r.rev = 0;
		int64_t qsum = 0, tsum = 0;
		for (int k = 0; k < j.n_pieces; ++k) {
			for (int w = 0; w < j.piece_len[k]; ++w) {
				const uint32_t c = j.piece[k][w], op = c & 0xf, len = c >> 4;
				if (op == 0 || op == 7 || op == 8) qsum += len, tsum += len; else if (op == 1) qsum += len; else if (op == 2 || op == 3) tsum += len;
			}
			if (j.piece_len[k] > 0) append_cigar(r, (uint32_t)j.piece_len[k], j.piece[k]);
		}
		if (qsum != j.qlen || tsum != j.tlen || !r.p) { o.n_cigar = -1; free(r.p); continue; }
		std::vector<uint8_t> qb((size_t)j.qlen + 32, 4), tb((size_t)j.tlen + 32, 4); // (update_extra compares 16 columns per load)
		if (j.qlen) memcpy(qb.data(), j.query, (size_t)j.qlen);
		if (j.tlen) memcpy(tb.data(), j.target, (size_t)j.tlen);
		update_extra(r, qb.data(), tb.data(), mat25, q, e, eqx != 0, log_gap != 0);
		o.n_cigar = (int32_t)r.p->n_cigar, o.blen = r.blen, o.mlen = r.mlen, o.n_ambi = (int32_t)r.p->n_ambi, o.dp_max = r.p->dp_max;
		o.qshift = r.qs, o.tshift = r.rs, o.is_spliced = r.is_spliced, o.cigar_off = (uint32_t)used;
		if (used + r.p->n_cigar > cigar_pool_cap) { free(r.p); return -1; }
		memcpy(cigar_pool + used, r.p->cigar, (size_t)r.p->n_cigar * 4);
		used += r.p->n_cigar;
		free(r.p);
	}
	return 0;
}

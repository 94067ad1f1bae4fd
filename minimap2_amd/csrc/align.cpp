#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include "align.hpp"
#include "region_rules.hpp"
#include "host_prof.hpp"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "chain_host.hpp"

namespace mm2amd {

using namespace ref;

extern const uint8_t kNt4Table[256];

namespace {
inline int span_of(const Anchor &a) { return (int)(a.y >> 32 & 0xff); }
inline uint32_t roundup32(uint32_t x) { --x; x |= x >> 1; x |= x >> 2; x |= x >> 4; x |= x >> 8; x |= x >> 16; return ++x; }
constexpr uint32_t kExtraWords = sizeof(Extra) / 4; // 7
}

void gen_score_matrix(const MapOpt &opt, int8_t mat[25])
{
	int8_t a = (int8_t)opt.a, b = (int8_t)opt.b, amb = (int8_t)opt.sc_ambi, ts = (int8_t)opt.transition;
	a = a < 0 ? -a : a, b = b > 0 ? -b : b, amb = amb > 0 ? -amb : amb;
	for (int i = 0; i < 4; ++i) {
		for (int j = 0; j < 4; ++j) mat[i * 5 + j] = i == j ? a : b;
		mat[i * 5 + 4] = amb;
	}
	for (int j = 0; j < 5; ++j) mat[20 + j] = amb;
	if (ts == 0 || ts == b) return; // NB: compared with the negated b, as in the reference (align.c:30)
	ts = ts > 0 ? -ts : ts;
	mat[0 * 5 + 2] = mat[1 * 5 + 3] = mat[2 * 5 + 0] = mat[3 * 5 + 1] = ts;
}

// ---------------------------------------------------------------------------------------------------------
// CIGAR container (mm_extra_t) handling; storage comes from libc because the caller frees it (map.c:629-630)
// ---------------------------------------------------------------------------------------------------------
static void enlarge_cigar(Reg &r, uint32_t n_cigar) // align.c:305-318
{
	if (n_cigar == 0) return;
	if (r.p == nullptr) {
		uint32_t cap = roundup32(n_cigar + kExtraWords);
		r.p = (Extra *)calloc(cap, 4);
		r.p->capacity = cap;
	} else if (r.p->n_cigar + n_cigar + kExtraWords > r.p->capacity) {
		r.p->capacity = roundup32(r.p->n_cigar + n_cigar + kExtraWords);
		r.p = (Extra *)realloc(r.p, (size_t)r.p->capacity * 4);
	}
}

// ---------------------------------------------------------------------------------------------------------
// mm_jump_split (jump.c): a clipped end of a spliced alignment is carried across an annotated junction when the clipped bases
// match the other side of the junction exactly (at most one mismatch on the near side).  No DP: byte comparisons only.
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int kMinExonLen = 20; // MM_MIN_EXON_LEN

// mm_idx_jump_get (index.c:932-959): the jumps whose off lies in (st, en]
const ref::JJump1 *jumps_between(const FlatIndex &fi, int32_t cid, int32_t st, int32_t en, int32_t *n)
{
	*n = 0;
	if (cid < 0 || cid >= (int32_t)fi.n_seq || !fi.has_jump) return nullptr;
	if (en < 0 || en > (int32_t)fi.seq_len[cid]) en = (int32_t)fi.seq_len[cid];
	const std::vector<ref::JJump1> &a = fi.jump[cid];
	if (a.empty()) return nullptr;
	auto last_le = [&](int32_t x) -> int32_t { // index of the last element with off <= x, -1 if none
		int32_t lo = 0, hi = (int32_t)a.size();
		while (lo < hi) { const int32_t mid = lo + (hi - lo) / 2; if (a[mid].off <= x) lo = mid + 1; else hi = mid; }
		return lo - 1;
	};
	const int32_t l = last_le(st), r = last_le(en);
	*n = r - l;
	return a.data() + (l + 1);
}

bool jump_applicable(const FlatIndex &fi, int32_t qlen, const Reg &r, int32_t ext, bool is_left) // mm_jump_check (jump.c:7-23)
{
	const int e = !r.rev ^ !is_left; // 0: the left end of the alignment is the read's start
	if (!r.p || r.p->n_cigar <= 0) return false;
	const int32_t clip = e == 0 ? r.qs : qlen - r.qe;
	const uint32_t cigar = r.p->cigar[is_left ? 0 : r.p->n_cigar - 1];
	const int32_t clen = (cigar & 0xf) == 0 ? (int32_t)(cigar >> 4) : 0;
	if (clen <= ext) return false;
	if (is_left) { if (clip >= r.rs) return false; }
	else if (clip >= (int32_t)fi.seq_len[r.rid] - r.re) return false;
	return true;
}

// the first (is_left) or last ql0 bases of the read in alignment orientation (mm_jump_get_qseq_seq, jump.c:25-51)
void jump_query(int32_t qlen, const char *qseq0, const Reg &r, bool is_left, int32_t ql0, uint8_t *qseq)
{
	int32_t k = 0;
	if (!r.rev) {
		if (is_left) for (int32_t i = 0; i < ql0; ++i) qseq[k++] = kNt4Table[(uint8_t)qseq0[i]];
		else for (int32_t i = qlen - ql0; i < qlen; ++i) qseq[k++] = kNt4Table[(uint8_t)qseq0[i]];
	} else {
		if (is_left) for (int32_t i = qlen - 1; i >= qlen - ql0; --i) { const uint8_t c = kNt4Table[(uint8_t)qseq0[i]]; qseq[k++] = c >= 4 ? c : 3 - c; }
		else for (int32_t i = ql0 - 1; i >= 0; --i) { const uint8_t c = kNt4Table[(uint8_t)qseq0[i]]; qseq[k++] = c >= 4 ? c : 3 - c; }
	}
}

void jump_left(const FlatIndex &fi, const MapOpt &opt, int32_t qlen, const char *qseq0, Reg &r, int32_t ts_strand) // mm_jump_split_left (jump.c:53-122)
{
	int32_t n, i0_anno = -1, n_anno = 0, mm0_anno = 0, i0_misc = -1, n_misc = 0, mm0_misc = 0, m, i0, mm0;
	const int32_t ext = 1 + (opt.b + opt.a - 1) / opt.a + 1;
	const int32_t clip = !r.rev ? r.qs : qlen - r.qe, extt = clip < ext ? clip : ext;
	if (!jump_applicable(fi, qlen, r, ext + kMinExonLen, true)) return;
	const ref::JJump1 *a = jumps_between(fi, r.rid, r.rs - extt, r.rs + ext, &n);
	if (n == 0) return;
	std::vector<uint8_t> buf;
	uint8_t *tseq = nullptr, *qseq = nullptr;
	for (int32_t i = 0; i < n; ++i) {
		const ref::JJump1 &ai = a[i];
		if (ts_strand * ai.strand < 0) continue; // wrong strand
		if (ai.off2 >= ai.off) continue;         // wrong direction
		if (ai.off - ai.off2 < 6) continue;      // intron too small
		if (ai.off2 < clip + ext) continue;      // not long enough
		if (!tseq) {
			buf.assign((size_t)(clip + ext) * 2, 0);
			tseq = buf.data(), qseq = tseq + clip + ext;
			jump_query(qlen, qseq0, r, true, clip + ext, qseq);
		}
		const int32_t tl1 = clip + (ai.off - r.rs);
		fi.getseq(r.rid, ai.off, r.rs + ext, &tseq[tl1]);
		fi.getseq(r.rid, ai.off2 - tl1, ai.off2, tseq);
		int32_t j, mm1 = 0, mm2 = 0;
		for (j = 0; j < tl1; ++j) if (qseq[j] != tseq[j] || qseq[j] > 3 || tseq[j] > 3) ++mm1;
		for (; j < clip + ext; ++j) if (qseq[j] != tseq[j] || qseq[j] > 3 || tseq[j] > 3) ++mm2;
		if (mm1 == 0 && mm2 <= 1) {
			if (ai.flag & ref::JUNC_ANNO) i0_anno = i, mm0_anno = mm1 + mm2, ++n_anno; // the rightmost one
			else i0_misc = i, mm0_misc = mm1 + mm2, ++n_misc;
		}
	}
	if (n_anno > 0) m = n_anno, i0 = i0_anno, mm0 = mm0_anno;
	else m = n_misc, i0 = i0_misc, mm0 = mm0_misc;
	const int32_t l = m > 0 ? a[i0].off - r.rs : 0; // may be negative
	if (m == 1 && clip + l >= opt.jump_min_match) { // one more exon
		enlarge_cigar(r, 2);
		memmove(r.p->cigar + 2, r.p->cigar, (size_t)r.p->n_cigar * 4);
		r.p->cigar[0] = (uint32_t)(clip + l) << 4 | 0u;
		r.p->cigar[1] = (uint32_t)(a[i0].off - a[i0].off2) << 4 | 3u; // N
		r.p->cigar[2] = ((r.p->cigar[2] >> 4) - (uint32_t)l) << 4 | 0u;
		r.p->n_cigar += 2;
		r.rs = a[i0].off2 - (clip + l);
		if (!r.rev) r.qs = 0; else r.qe = qlen;
		r.blen += clip, r.mlen += clip - mm0;
		r.p->dp_max0 += (clip - mm0) * opt.a - mm0 * opt.b;
		r.p->dp_max += (clip - mm0) * opt.a - mm0 * opt.b;
		if (!r.is_spliced) r.is_spliced = 1, r.p->dp_max += (opt.a + opt.b) + ((opt.a + opt.b) >> 1);
	} else if (m > 0 && a[i0].off > r.rs) { // trim by l (positive here)
		r.p->cigar[0] -= (uint32_t)l << 4 | 0u;
		r.rs += l;
		if (!r.rev) r.qs += l; else r.qe -= l;
	}
}

void jump_right(const FlatIndex &fi, const MapOpt &opt, int32_t qlen, const char *qseq0, Reg &r, int32_t ts_strand) // mm_jump_split_right (jump.c:124-193)
{
	int32_t n, i0_anno = -1, n_anno = 0, mm0_anno = 0, i0_misc = -1, n_misc = 0, mm0_misc = 0, m, i0, mm0;
	const int32_t ext = 1 + (opt.b + opt.a - 1) / opt.a + 1;
	const int32_t clip = !r.rev ? qlen - r.qe : r.qs, extt = clip < ext ? clip : ext;
	if (!jump_applicable(fi, qlen, r, ext + kMinExonLen, false)) return;
	const ref::JJump1 *a = jumps_between(fi, r.rid, r.re - ext, r.re + extt, &n);
	if (n == 0) return;
	std::vector<uint8_t> buf;
	uint8_t *tseq = nullptr, *qseq = nullptr;
	for (int32_t i = 0; i < n; ++i) {
		const ref::JJump1 &ai = a[i];
		if (ts_strand * ai.strand < 0) continue;
		if (ai.off2 <= ai.off) continue;
		if (ai.off2 - ai.off < 6) continue;
		if (ai.off2 + clip + ext > (int32_t)fi.seq_len[r.rid]) continue;
		if (!tseq) {
			buf.assign((size_t)(clip + ext) * 2, 0);
			tseq = buf.data(), qseq = tseq + clip + ext;
			jump_query(qlen, qseq0, r, false, clip + ext, qseq);
		}
		const int32_t tl1 = clip + (r.re - ai.off);
		fi.getseq(r.rid, r.re - ext, ai.off, tseq);
		fi.getseq(r.rid, ai.off2, ai.off2 + tl1, &tseq[clip + ext - tl1]);
		int32_t j, mm1 = 0, mm2 = 0;
		for (j = 0; j < clip + ext - tl1; ++j) if (qseq[j] != tseq[j] || qseq[j] > 3 || tseq[j] > 3) ++mm2;
		for (; j < clip + ext; ++j) if (qseq[j] != tseq[j] || qseq[j] > 3 || tseq[j] > 3) ++mm1;
		if (mm1 == 0 && mm2 <= 1) {
			if (ai.flag & ref::JUNC_ANNO) { if (i0_anno < 0) i0_anno = i, mm0_anno = mm1 + mm2; ++n_anno; } // the leftmost one
			else { if (i0_misc < 0) i0_misc = i, mm0_misc = mm1 + mm2; ++n_misc; }
		}
	}
	if (n_anno > 0) m = n_anno, i0 = i0_anno, mm0 = mm0_anno;
	else m = n_misc, i0 = i0_misc, mm0 = mm0_misc;
	const int32_t l = m > 0 ? r.re - a[i0].off : 0;
	if (m == 1 && clip + l >= opt.jump_min_match) {
		enlarge_cigar(r, 2);
		r.p->cigar[r.p->n_cigar - 1] = ((r.p->cigar[r.p->n_cigar - 1] >> 4) - (uint32_t)l) << 4 | 0u;
		r.p->cigar[r.p->n_cigar] = (uint32_t)(a[i0].off2 - a[i0].off) << 4 | 3u;
		r.p->cigar[r.p->n_cigar + 1] = (uint32_t)(clip + l) << 4 | 0u;
		r.p->n_cigar += 2;
		r.re = a[i0].off2 + (clip + l);
		if (!r.rev) r.qe = qlen; else r.qs = 0;
		r.blen += clip, r.mlen += clip - mm0;
		r.p->dp_max0 += (clip - mm0) * opt.a - mm0 * opt.b;
		r.p->dp_max += (clip - mm0) * opt.a - mm0 * opt.b;
		if (!r.is_spliced) r.is_spliced = 1, r.p->dp_max += (opt.a + opt.b) + ((opt.a + opt.b) >> 1);
	} else if (m > 0 && r.re > a[i0].off) {
		r.p->cigar[r.p->n_cigar - 1] -= (uint32_t)l << 4 | 0u;
		r.re -= l;
		if (!r.rev) r.qe -= l; else r.qs += l;
	}
}
} // namespace

void jump_split(const FlatIndex &fi, const MapOpt &opt, int32_t qlen, const char *qseq, Reg &r, int32_t ts_strand) // mm_jump_split (jump.c:195-200)
{
	jump_left(fi, opt, qlen, qseq, r, ts_strand);
	jump_right(fi, opt, qlen, qseq, r, ts_strand);
}

void append_cigar(Reg &r, uint32_t n_cigar, const uint32_t *cigar) // what mm_append_cigar (align.c:320-334) leaves: a first operation of the tail's kind is absorbed by the tail
{
	if (n_cigar == 0) return;
	enlarge_cigar(r, n_cigar);
	Extra &x = *r.p;
	const uint32_t absorbed = x.n_cigar > 0 && ((x.cigar[x.n_cigar - 1] ^ cigar[0]) & 0xf) == 0 ? 1u : 0u;
	if (absorbed) x.cigar[x.n_cigar - 1] += cigar[0] & ~0xfu;
	std::copy(cigar + absorbed, cigar + n_cigar, x.cigar + x.n_cigar);
	x.n_cigar += n_cigar - absorbed;
}

// What mm_fix_cigar (align.c:105-181) leaves of a CIGAR, formulated the way region_finish_kernel does it (region_finish.hip, step B) rather than as one
// in-place walk: (1) how far each gap slides is a property of the sequence it skips and of the room in the match before it, worked out against the
// CIGAR as it came and applied afterwards; (2) clusters of insertions and deletions; (3) packing; (4) the leading gap.  qshift / tshift: bases a
// dropped leading insertion / deletion took off the query / target start.
namespace {
inline uint32_t cg_op(uint32_t c) { return c & 0xfu; }
inline uint32_t cg_len(uint32_t c) { return c >> 4; }
inline bool cg_is_gap(uint32_t c) { return cg_op(c) == 1 || cg_op(c) == 2; }

// a gap of g bases that starts at seq[at] can move one base to the left whenever the base that enters it on the left equals the base that leaves it on
// the right: how many such moves in a row, at most `room` (the bases of the match before it)
inline uint32_t gap_slide(const uint8_t *seq, uint32_t at, uint32_t g, uint32_t room)
{
	uint32_t moved = 0;
	while (moved < room && seq[at - 1 - moved] == seq[at + g - 1 - moved]) ++moved;
	return moved;
}
} // namespace

static void fix_cigar(Reg &r, const uint8_t *qseq, const uint8_t *tseq, int *qshift, int *tshift)
{
	Extra *const p = r.p;
	uint32_t *const cg = p->cigar;
	*qshift = *tshift = 0;
	if (p->n_cigar <= 1) return;
	uint32_t n = p->n_cigar;
	bool repack = false; // an empty operation came in or was made: step (3) runs
	{ // (1) a gap between two matches gives bases of the match before it to the match after it; that match may in turn lend them to the next gap.
	  //     Operation k starts where the incoming CIGAR puts it: slides before it move bases between matches that both lie before it.
		thread_local std::vector<uint32_t> slide;
		slide.assign(n, 0);
		uint32_t q_at = 0, t_at = 0;
		for (uint32_t k = 0; k < n; ++k) {
			const uint32_t op = cg_op(cg[k]), len = cg_len(cg[k]);
			if (len + (k > 0 ? slide[k - 1] : 0) == 0) repack = true; // empty as the reference's walk meets it: after the gap before it has slid (a match it lent bases to is no longer empty)
			if (cg_is_gap(cg[k]) && k > 0 && k + 1 < n && cg_op(cg[k - 1]) == 0 && cg_op(cg[k + 1]) == 0) {
				const uint32_t room = cg_len(cg[k - 1]) + (k >= 2 ? slide[k - 2] : 0);
				slide[k] = op == 1 ? gap_slide(qseq, q_at, len, room) : gap_slide(tseq, t_at, len, room);
				if (slide[k] == room) repack = true; // the match before it is used up
			}
			if (op == 0) q_at += len, t_at += len;
			else if (op == 1) q_at += len;
			else if (op == 2 || op == 3) t_at += len;
		}
		assert((int32_t)q_at == r.qe - r.qs && (int32_t)t_at == r.re - r.rs);
		for (uint32_t k = 1; k + 1 < n; ++k)
			if (slide[k]) cg[k - 1] -= slide[k] << 4, cg[k + 1] += slide[k] << 4;
	}
	// (2) a run of insertions and deletions (empty operations inside it do not end it) that opens with one of each, has both kinds and more than two
	//     members becomes one insertion and one deletion
	for (uint32_t k = 0; k + 2 < n;) {
		if (!(cg_is_gap(cg[k]) && cg_is_gap(cg[k + 1]) && cg_op(cg[k]) != cg_op(cg[k + 1]))) { ++k; continue; }
		uint32_t end = k, bases[3] = {0, 0, 0};
		while (end < n && (cg_is_gap(cg[end]) || cg_len(cg[end]) == 0)) {
			if (cg_is_gap(cg[end])) bases[cg_op(cg[end])] += cg_len(cg[end]);
			++end;
		}
		if (bases[1] > 0 && bases[2] > 0 && end - k > 2) {
			cg[k] = bases[1] << 4 | 1, cg[k + 1] = bases[2] << 4 | 2;
			for (uint32_t j = k + 2; j < end; ++j) cg[j] &= 0xfu;
			repack = true;
		}
		k = end + 1; // (the operation that ended the run is neither a gap nor empty: no run opens there)
	}
	if (repack) { // (3) empty operations go, then neighbours of one kind join
		uint32_t kept = 0;
		for (uint32_t k = 0; k < n; ++k)
			if (cg_len(cg[k]) != 0) cg[kept++] = cg[k];
		n = kept, kept = 0;
		for (uint32_t k = 0; k < n; ++k) {
			if (k + 1 < n && cg_op(cg[k]) == cg_op(cg[k + 1])) cg[k + 1] += cg_len(cg[k]) << 4;
			else cg[kept++] = cg[k];
		}
		n = kept;
	}
	if (cg_is_gap(cg[0])) { // (4) an alignment does not open with a gap: its bases leave the aligned interval instead
		const int32_t bases = (int32_t)cg_len(cg[0]);
		if (cg_op(cg[0]) == 1) {
			if (r.rev) r.qe -= bases; else r.qs += bases;
			*qshift = bases;
		} else r.rs += bases, *tshift = bases;
		--n;
		memmove(cg, cg + 1, (size_t)n * 4);
	}
	p->n_cigar = n;
}

// MM_F_EQX (mm_update_cigar_eqx, align.c:183-252): every match operation cut into stretches of equal and of unequal columns.  The stretches are
// visited by one routine, once to count them and once to write them.  (As in the reference, a CIGAR whose matches are one stretch each is relabelled
// in place as all-equal.)
namespace {
template <class Visit> // visit(stretch length, equal?) for the stretches of a match of len columns that starts at q / t
inline void match_stretches(const uint8_t *q, const uint8_t *t, uint32_t len, Visit visit)
{
	for (uint32_t at = 0; at < len;) {
		const bool equal = q[at] == t[at];
		uint32_t end = at + 1;
		while (end < len && (q[end] == t[end]) == equal) ++end;
		visit(end - at, equal);
		at = end;
	}
}
template <class OnMatch, class OnOther> // the CIGAR's operations with their places in the two sequences
inline void walk_cigar(const Extra *p, OnMatch on_match, OnOther on_other)
{
	uint32_t q_at = 0, t_at = 0;
	for (uint32_t k = 0; k < p->n_cigar; ++k) {
		const uint32_t op = cg_op(p->cigar[k]), len = cg_len(p->cigar[k]);
		if (op == 0) { on_match(q_at, t_at, len); q_at += len, t_at += len; continue; }
		on_other(p->cigar[k]);
		if (op == 1) q_at += len;
		else if (op == 2 || op == 3) t_at += len;
	}
}
} // namespace

static void cigar_to_eqx(Reg &r, const uint8_t *qseq, const uint8_t *tseq)
{
	if (!r.p) return;
	uint32_t n_stretch = 0, n_match = 0;
	walk_cigar(r.p, [&](uint32_t q, uint32_t t, uint32_t len) { ++n_match; match_stretches(qseq + q, tseq + t, len, [&](uint32_t, bool) { ++n_stretch; }); }, [](uint32_t) {});
	if (n_stretch == n_match) {
		for (uint32_t k = 0; k < r.p->n_cigar; ++k)
			if (cg_op(r.p->cigar[k]) == 0) r.p->cigar[k] = cg_len(r.p->cigar[k]) << 4 | 7;
		return;
	}
	const uint32_t cap = roundup32(r.p->n_cigar + (n_stretch - n_match) + (uint32_t)sizeof(Extra));
	Extra *const grown = (Extra *)calloc(cap, 4);
	memcpy(grown, r.p, sizeof(Extra));
	grown->capacity = cap;
	uint32_t m = 0;
	walk_cigar(r.p, [&](uint32_t q, uint32_t t, uint32_t len) { match_stretches(qseq + q, tseq + t, len, [&](uint32_t l, bool equal) { grown->cigar[m++] = l << 4 | (equal ? 7u : 8u); }); },
	           [&](uint32_t c) { grown->cigar[m++] = c; });
	grown->n_cigar = m;
	free(r.p);
	r.p = grown;
}

void update_extra(Reg &r, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat, int8_t q, int8_t e, bool is_eqx, bool log_gap)
{
	Extra *p = r.p;
	if (!p) return;
	int qshift, tshift;
	int32_t toff = 0, qoff = 0;
	double s = 0.0, max = 0.0;
	{ hostprof::Scope hp(hostprof::FIX_CIGAR); fix_cigar(r, qseq, tseq, &qshift, &tshift); }
	hostprof::Scope hp(hostprof::EXTRA_SCAN);
	const int32_t match_sc = mat[0];
	const bool uniform_match = match_sc >= 0 && mat[6] == match_sc && mat[12] == match_sc && mat[18] == match_sc; // every A/C/G/T match scores the same and not below zero
	qseq += qshift, tseq += tshift;
	// (running totals in locals: `r` and `p` may alias the CIGAR as far as the compiler knows, which would put five stores into every step)
	int32_t blen = 0, mlen = 0, n_ambi_all = 0, spliced = 0;
	const uint32_t n_cigar = p->n_cigar, *const cigar = p->cigar;
	static thread_local struct GapCost { int q = INT32_MIN, e = INT32_MIN; double c[64]; } gc; // q + e * log2(1 + len) of the short gaps, by the reference's own expression
	if (log_gap && (gc.q != q || gc.e != e)) {
		gc.q = q, gc.e = e;
		for (int len = 0; len < 64; ++len) gc.c[len] = q + (double)e * fast_log2(1.0 + len);
	}
	for (uint32_t k = 0; k < n_cigar; ++k) {
		const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			// The reference walks the run base by base in doubles: s += score; clamp at 0; track the maximum (align.c:268-277).  Every value
			// here is an integer plus gap costs that are multiples of 2^-23 (fast_log2 returns a float >= 1), far below 2^29 in size, so
			// each double addition is exact and the walk can be done on integer prefix sums: with `lo`/`hi` the smallest / largest prefix
			// of the run, no clamp happens iff s + lo >= 0, and then the maximum candidate is s + hi.  A run that does clamp (start of an
			// alignment) is walked again the reference's way.
			int n_ambi = 0, n_diff = 0;
			const uint8_t *qr = qseq + qoff, *tr = tseq + toff;
			int32_t acc = 0, lo = INT32_MAX, hi = INT32_MIN;
#if defined(__SSE2__)
			if (uniform_match) { // 16 columns at a time: between two columns that are not plain matches the prefix only rises
				for (uint32_t l0 = 0; l0 < len; l0 += 16) {
					const uint32_t m = len - l0 < 16 ? len - l0 : 16, lane_mask = m == 16 ? 0xffffu : (1u << m) - 1;
					const __m128i vq = _mm_loadu_si128((const __m128i *)(qr + l0)), vt = _mm_loadu_si128((const __m128i *)(tr + l0)); // reads up to 15 bytes past the run: both buffers are padded
					const uint32_t eq = (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(vq, vt));
					const uint32_t amb = (uint32_t)_mm_movemask_epi8(_mm_cmpgt_epi8(_mm_or_si128(vq, vt), _mm_set1_epi8(3))) & lane_mask;
					uint32_t special = (~eq | amb) & lane_mask; // mismatches and ambiguous columns
					n_ambi += __builtin_popcount(amb), n_diff += __builtin_popcount(special & ~amb);
					uint32_t done = 0; // columns of this block already added
					while (special) {
						const uint32_t c = (uint32_t)__builtin_ctz(special);
						special &= special - 1;
						if (c > done) { // plain matches before it: the first is the lowest, the last the highest
							lo = lo < acc + match_sc ? lo : acc + match_sc;
							acc += (int32_t)(c - done) * match_sc;
							hi = hi > acc ? hi : acc;
						}
						acc += mat[tr[l0 + c] * 5 + qr[l0 + c]];
						lo = lo < acc ? lo : acc, hi = hi > acc ? hi : acc;
						done = c + 1;
					}
					if (m > done) {
						lo = lo < acc + match_sc ? lo : acc + match_sc;
						acc += (int32_t)(m - done) * match_sc;
						hi = hi > acc ? hi : acc;
					}
				}
			} else
#endif
			for (uint32_t l = 0; l < len; ++l) {
				const int cq = qr[l], ct = tr[l], amb = (ct > 3) | (cq > 3);
				n_ambi += amb, n_diff += (ct != cq) & !amb;
				acc += mat[ct * 5 + cq];
				lo = lo < acc ? lo : acc, hi = hi > acc ? hi : acc;
			}
			if (len > 0) {
				if (s + (double)lo >= 0) {
					const double top = s + (double)hi;
					max = max > top ? max : top;
					s += (double)acc;
				} else {
					for (uint32_t l = 0; l < len; ++l) {
						s += mat[tr[l] * 5 + qr[l]];
						if (s < 0) s = 0;
						else max = max > s ? max : s;
					}
				}
			}
			blen += len - n_ambi, mlen += len - (n_ambi + n_diff), n_ambi_all += n_ambi;
			toff += len, qoff += len;
		} else if (op == 1 || op == 2) {
			int n_ambi = 0;
			const uint8_t *sq = op == 1 ? qseq + qoff : tseq + toff;
			for (uint32_t l = 0; l < len; ++l) n_ambi += sq[l] > 3;
			blen += len - n_ambi, n_ambi_all += n_ambi;
			if (log_gap) s -= len < 64 ? gc.c[len] : q + (double)e * fast_log2(1.0 + len);
			else s -= q + e;
			if (s < 0) s = 0;
			if (op == 1) qoff += len; else toff += len;
		} else if (op == 3) spliced = 1, toff += len;
	}
	r.blen = blen, r.mlen = mlen, r.is_spliced = spliced, p->n_ambi += n_ambi_all;
	p->dp_max = p->dp_max0 = (int32_t)(max + .499);
	assert(qoff == r.qe - r.qs && toff == r.re - r.rs);
	if (is_eqx) cigar_to_eqx(r, qseq, tseq);
}

// ---------------------------------------------------------------------------------------------------------
// Z-drop / inversion test on a finished gap-fill alignment (mm_test_zdrop, align.c:61-103)
// ---------------------------------------------------------------------------------------------------------
struct ZdropScan { int32_t max_zdrop = 0; int pos[2][2] = {{-1, -1}, {-1, -1}}; }; // pos[0] target, pos[1] query: [peak, trough]

// the scan half of mm_test_zdrop (align.c:61-84): the same loop runs inside ksw_fast.hip for the jobs that kernel handles
static ZdropScan zdrop_scan(const MapOpt &opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, const uint32_t *cigar, const int8_t *mat)
{
	ZdropScan z;
	int32_t score = 0, max = INT32_MIN, max_i = -1, max_j = -1, i = 0, j = 0;
	auto track = [&](int32_t sc, int ci, int cj) { // update_max_zdrop, align.c:46-59
		if (sc < max) {
			const int li = ci - max_i, lj = cj - max_j, diff = li > lj ? li - lj : lj - li, zd = max - sc - diff * opt.e;
			if (zd > z.max_zdrop) z.max_zdrop = zd, z.pos[0][0] = max_i, z.pos[0][1] = ci, z.pos[1][0] = max_j, z.pos[1][1] = cj;
		} else max = sc, max_i = ci, max_j = cj;
	};
	for (uint32_t k = 0; k < n_cigar; ++k) {
		const uint32_t op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			for (uint32_t l = 0; l < len; ++l) {
				score += mat[tseq[i + l] * 5 + qseq[j + l]];
				track(score, i + l, j + l);
			}
			i += len, j += len;
		} else if (op == 1 || op == 2 || op == 3) {
			score -= opt.q + opt.e * len;
			if (op == 1) j += len; else i += len;
			track(score, i, j);
		}
	}
	return z;
}

// the decision half (align.c:86-103): 2 = the dropped stretch aligns on the reverse strand (inversion), 1 = plain Z-drop
static int zdrop_decide(const MapOpt &opt, const ZdropScan &z, const uint8_t *qseq, const uint8_t *tseq, const int8_t *mat)
{
	const int q_len = z.pos[1][1] - z.pos[1][0], t_len = z.pos[0][1] - z.pos[0][0];
	if (!(opt.flag & (F_SPLICE | F_SR | F_FOR_ONLY | F_REV_ONLY)) && z.max_zdrop > opt.zdrop_inv && q_len < opt.max_gap && t_len < opt.max_gap) {
		std::vector<uint8_t> rc(q_len > 0 ? q_len : 0); // reverse complement of the dropped query segment
		for (int x = 0; x < q_len; ++x) { const int c = qseq[z.pos[1][1] - x - 1]; rc[x] = c >= 4 ? 4 : 3 - c; }
		int q_off, t_off;
		const int sc = ll_local_score(q_len, rc.data(), t_len, tseq + z.pos[0][0], mat, opt.q, opt.e, &q_off, &t_off);
		if (sc >= opt.min_chain_score * opt.a && sc >= opt.min_dp_max) return 2;
	}
	return z.max_zdrop > opt.zdrop ? 1 : 0;
}

int test_zdrop(const MapOpt &opt, const uint8_t *qseq, const uint8_t *tseq, uint32_t n_cigar, const uint32_t *cigar, const int8_t *mat)
{
	return zdrop_decide(opt, zdrop_scan(opt, qseq, tseq, n_cigar, cigar, mat), qseq, tseq, mat);
}

// ---------------------------------------------------------------------------------------------------------
// Seed clean-up before window selection (align.c:435-561)
// ---------------------------------------------------------------------------------------------------------
// (the rules themselves: region_rules.hpp, shared with region_plan_kernel)
static void long_gap_sites(const Anchor *chain, int cnt1, int min_gap, std::vector<int32_t> &K) // collect_long_gaps, align.c:435-452: nothing unless there are two
{
	K.clear();
	for (int i = 1; i < cnt1; ++i) if (rr_is_long_gap(chain, i, min_gap)) K.push_back(i);
	if (K.size() <= 1) K.clear();
}

// ---------------------------------------------------------------------------------------------------------
// Aligner
// ---------------------------------------------------------------------------------------------------------
Aligner::Aligner(const MapOpt &opt, const FlatIndex &fi) : opt_(opt), fi_(fi)
{
	gen_score_matrix(opt, mat_);
	bw_ = (int)(opt.bw * 1.5 + 1.);
	bw_long_ = (int)(opt.bw_long * 1.5 + 1.);
	if (bw_long_ < bw_) bw_long_ = bw_;
	qstrand_ = (opt.flag & F_QSTRAND) != 0;
}

void Aligner::begin_read(ReadAlign &ra, const char *seq, int qlen, RegVec &regs, Anchor *a, uint64_t qpool_fwd, uint64_t qpool_rev, uint8_t *q4)
{
	ra.qlen = qlen, ra.qpool_off = qpool_fwd, ra.qpool_rev = qpool_rev, ra.a = a;
	ra.q4 = q4, ra.seq = seq, ra.q4_ready[0] = ra.q4_ready[1] = false;
	ra.n_a = squeeze_anchors(regs, a);
	ra.tasks.clear(); ra.order.clear(); ra.finish_queue.clear();
	for (size_t i = 0; i < regs.size(); ++i) add_region(ra, regs[i], -1);
}

#if defined(__x86_64__)
// nt4 codes of 16 bases per step (kNt4Table: A/a 0, C/c 1, G/g 2, T/t/U/u 3, the bytes 0..3 themselves, anything else 4): the letters' low
// nibbles differ (A 1, C 3, G 7, T 4, U 5), so one byte shuffle proposes the code and another names the letter the nibble stands for, which
// the byte (case bit cleared) must equal.  `rev`: the block is the reverse complement of the 16 bases ENDING at src + 16.
__attribute__((target("ssse3"))) static inline __m128i nt4_block(__m128i c, bool rev)
{
	// (unused nibbles expect 0x20, a value `up` -- the byte with its case bit 0x20 cleared -- can never take: with 0 there, the bytes 0x00 and 0x20
	// passed for letters of code 4)
	const __m128i expect = _mm_setr_epi8(0x20, 'A', 0x20, 'C', 'T', 'U', 0x20, 'G', 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20, 0x20);
	const __m128i code = _mm_setr_epi8(4, 0, 4, 1, 3, 3, 4, 2, 4, 4, 4, 4, 4, 4, 4, 4);
	if (rev) c = _mm_shuffle_epi8(c, _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0));
	const __m128i nib = _mm_and_si128(c, _mm_set1_epi8(0x0f)), up = _mm_and_si128(c, _mm_set1_epi8((char)0xdf));
	const __m128i letter = _mm_cmpeq_epi8(_mm_shuffle_epi8(expect, nib), up);                     // a letter of the alphabet, either case
	const __m128i raw = _mm_cmpeq_epi8(_mm_and_si128(c, _mm_set1_epi8((char)0xfc)), _mm_setzero_si128()); // the bytes 0..3
	__m128i v = _mm_or_si128(_mm_and_si128(letter, _mm_shuffle_epi8(code, nib)), _mm_and_si128(raw, c));
	const __m128i known = _mm_or_si128(letter, raw);
	if (rev) v = _mm_xor_si128(v, _mm_and_si128(_mm_cmplt_epi8(v, _mm_set1_epi8(4)), _mm_set1_epi8(3))); // complement: 3 - code, where there is a code
	return _mm_or_si128(_mm_and_si128(known, v), _mm_andnot_si128(known, _mm_set1_epi8(4)));
}
__attribute__((target("ssse3"))) static void nt4_encode_ssse3(const uint8_t *src, int n, uint8_t *dst, bool rev)
{
	int i = 0;
	if (!rev) for (; i + 16 <= n; i += 16) _mm_storeu_si128((__m128i *)(dst + i), nt4_block(_mm_loadu_si128((const __m128i *)(src + i)), false));
	else for (; i + 16 <= n; i += 16) _mm_storeu_si128((__m128i *)(dst + i), nt4_block(_mm_loadu_si128((const __m128i *)(src + n - 16 - i)), true));
	for (; i < n; ++i) { const uint8_t c = kNt4Table[rev ? src[n - 1 - i] : src[i]]; dst[i] = !rev ? c : c < 4 ? 3 - c : 4; }
}
static bool nt4_ssse3_ok() // the CPU has it, and it agrees with the table on every byte value (checked once)
{
	static const bool ok = [] {
		if (!__builtin_cpu_supports("ssse3")) return false;
		uint8_t src[272], a[272], b[272];
		for (int i = 0; i < 272; ++i) src[i] = (uint8_t)(i * 7 + (i >> 4));
		for (int i = 0; i < 256; ++i) src[i] = (uint8_t)i;
		for (int rev = 0; rev < 2; ++rev) {
			nt4_encode_ssse3(src, 272, a, rev != 0);
			for (int i = 0; i < 272; ++i) { const uint8_t c = kNt4Table[rev ? src[271 - i] : src[i]]; b[i] = !rev ? c : c < 4 ? 3 - c : 4; }
			if (memcmp(a, b, 272) != 0) {
				fprintf(stderr, "[mm2amd] the SSSE3 read encoder disagrees with the nt4 table: using the scalar encoder\n");
				return false;
			}
		}
		return true;
	}();
	return ok;
}
#endif

const uint8_t *strand_codes(ReadAlign &ra, int strand)
{
	uint8_t *dst = ra.q4 + (size_t)strand * q4_stride(ra.qlen);
	if (ra.q4_ready[strand]) return dst;
	hostprof::Scope hp(hostprof::Q4_ENCODE);
	const int n = ra.qlen;
	const uint8_t *src = (const uint8_t *)ra.seq;
#if defined(__x86_64__)
	if (nt4_ssse3_ok()) { nt4_encode_ssse3(src, n, dst, strand != 0); ra.q4_ready[strand] = true; return dst; }
#endif
	static const struct RcTable { uint8_t t[256]; RcTable() { for (int c = 0; c < 256; ++c) t[c] = kNt4Table[c] < 4 ? 3 - kNt4Table[c] : 4; } } rc;
	if (strand == 0) for (int i = 0; i < n; ++i) dst[i] = kNt4Table[src[i]];
	else for (int i = 0; i < n; ++i) dst[i] = rc.t[src[n - 1 - i]];
	ra.q4_ready[strand] = true;
	return dst;
}

// One region to align: one task, or one task per assumed transcript strand (align.c:1068-1077).  order_pos < 0 appends.
void Aligner::add_region(ReadAlign &ra, const Reg &r, int order_pos)
{
	const bool splice = opt_.flag & F_SPLICE, both = splice && (opt_.flag & F_SPLICE_FOR) && (opt_.flag & F_SPLICE_REV);
	const int ti = (int)ra.tasks.size();
	RegionTask t;
	t.r = r;
	t.chain_ungapped = r.qe - r.qs == r.re - r.rs;
	if (both) t.splice_flag = (int32_t)F_SPLICE_FOR, t.twin = ti + 1;
	else if (splice) t.splice_flag = (int32_t)(opt_.flag & (F_SPLICE_FOR | F_SPLICE_REV));
	ra.tasks.push_back(t);
	if (both) {
		t.splice_flag = (int32_t)F_SPLICE_REV, t.twin = ti, t.lead = false;
		ra.tasks.push_back(std::move(t));
	}
	if (order_pos < 0) ra.order.push_back(ti);
	else ra.order.insert(ra.order.begin() + order_pos, ti);
}

// Both strand attempts of a region are finished: keep the better one (align.c:1078-1096)
void Aligner::join_strands(ReadAlign &ra, int lead_ti)
{
	RegionTask &s0 = ra.tasks[lead_ti], &s1 = ra.tasks[s0.twin];
	if ((opt_.flag & F_SR_RNA) && s0.r.p && s0.chain_ungapped && s0.r.qe - s0.r.qs == s0.r.re - s0.r.rs && s0.r.qs == 0 && s0.r.qe == ra.qlen) {
		// splice:sr (align.c:1072-1074): an end-to-end ungapped alignment says nothing about the transcript strand; the reference does
		// not even try the other one
		free(s1.r.p), s1.r.p = nullptr;
		s0.r.p->trans_strand = 0;
		return;
	}
	if (!s0.r.p || !s1.r.p) throw std::runtime_error("[mm2amd] spliced alignment produced no CIGAR for a region (the reference dereferences a null pointer here)");
	int which, trans_strand;
	if (s0.r.p->dp_score > s1.r.p->dp_score) which = 0, trans_strand = 1;
	else if (s0.r.p->dp_score < s1.r.p->dp_score) which = 1, trans_strand = 2;
	else trans_strand = 3, which = (ra.qlen + s0.r.p->dp_score) & 1;
	if (which == 1) std::swap(s0.r, s1.r), std::swap(s0.r2, s1.r2);
	free(s1.r.p), s1.r.p = nullptr;
	Reg &r = s0.r;
	r.p->trans_strand = trans_strand;
	if (r.is_spliced) {
		if (trans_strand == 1 || trans_strand == 2) r.p->dp_max += (opt_.a + opt_.b) + ((opt_.a + opt_.b) >> 1);
		else r.p->dp_max -= opt_.a + opt_.b;
	}
}

// ksw_ll_i16 score of an anchor's k-mer extended by anchor_ext_len on both sides (mm_seed_ext_score, align.c:591-616)
static int seed_ext_score(const MapOpt &opt, const FlatIndex &fi, const int8_t *mat, int qlen, ReadAlign &ra, const Anchor &a, std::vector<uint8_t> &tbuf)
{
	const int q_span = span_of(a), ext_len = opt.anchor_ext_len;
	const uint32_t rid = (uint32_t)(a.x << 1 >> 33);
	int re = (int)((uint32_t)a.x + 1), rs = re - q_span, qe = (int)((uint32_t)a.y + 1), qs = qe - q_span, q_off, t_off;
	rs = rs - ext_len > 0 ? rs - ext_len : 0;
	qs = qs - ext_len > 0 ? qs - ext_len : 0;
	re = re + ext_len < (int32_t)fi.seq_len[rid] ? re + ext_len : (int32_t)fi.seq_len[rid];
	qe = qe + ext_len < qlen ? qe + ext_len : qlen;
	tbuf.resize(re - rs);
	fi.getseq(rid, rs, re, tbuf.data());
	return ll_local_score(qe - qs, strand_codes(ra, (int)(a.x >> 63)) + qs, re - rs, tbuf.data(), mat, opt.q, opt.e, &q_off, &t_off);
}

// boundary exons held by one weak anchor far from the rest are dropped (mm_fix_bad_ends_splice, align.c:618-636)
static void trim_bad_ends_splice(const MapOpt &opt, const FlatIndex &fi, const Reg &r, const int8_t *mat, int qlen, ReadAlign &ra, const Anchor *a,
                                 std::vector<uint8_t> &tbuf, int32_t *as1, int32_t *cnt1)
{
	*as1 = r.as, *cnt1 = r.cnt;
	if (r.cnt < 3) return;
	double log_gap = log((double)((int32_t)a[r.as + 1].x - (int32_t)a[r.as].x));
	if (span_of(a[r.as]) < log_gap + opt.anchor_ext_shift) {
		const int score = seed_ext_score(opt, fi, mat, qlen, ra, a[r.as], tbuf);
		if ((double)score / mat[0] < log_gap + opt.anchor_ext_shift) ++(*as1), --(*cnt1);
	}
	log_gap = log((double)((int32_t)a[r.as + r.cnt - 1].x - (int32_t)a[r.as + r.cnt - 2].x));
	if (span_of(a[r.as + r.cnt - 1]) < log_gap + opt.anchor_ext_shift) {
		const int score = seed_ext_score(opt, fi, mat, qlen, ra, a[r.as + r.cnt - 1], tbuf);
		if ((double)score / mat[0] < log_gap + opt.anchor_ext_shift) --(*cnt1);
	}
}

// where an anchor's window boundary sits: the middle of the k-mer, or for HPC indices the start of the
// homopolymer run holding the anchor's last base (mm_adjust_minier, align.c:418-433)
static void anchor_boundary(const FlatIndex &fi, ReadAlign &ra, int qlen, const Anchor &a, int32_t *r, int32_t *q)
{
	if (fi.flag & I_HPC) {
		const uint8_t *qseq = strand_codes(ra, (int)(a.x >> 63));
		int i, c;
		*q = (int32_t)a.y;
		for (i = *q - 1, c = qseq[*q]; i > 0; --i) if (qseq[i] != c) break;
		*q = i + 1;
		const uint32_t rid = (uint32_t)(a.x << 1 >> 33);
		const int64_t x = (int32_t)a.x;
		const int cb = fi.base(rid, (uint32_t)x);
		int64_t j;
		for (j = x - 1; j >= 0; --j) if (fi.base(rid, (uint32_t)j) != cb) break;
		*r = (int32_t)a.x + 1 - (int)(x - j);
	} else {
		*r = (int32_t)a.x - (fi.k >> 1);
		*q = (int32_t)a.y - (fi.k >> 1);
	}
}

void Aligner::plan_region(ReadAlign &ra, RegionTask &t)
{
	Reg &r = t.r;
	Anchor *a = ra.a;
	const int qlen = ra.qlen;
	t.planned = true;
	t.r2.cnt = 0;
	if (r.cnt == 0) { t.done = true; return; }
	const bool is_splice = opt_.flag & F_SPLICE, is_sr = opt_.flag & F_SR;
	const int32_t rid = (int32_t)(a[r.as].x << 1 >> 33), rev = (int32_t)(a[r.as].x >> 63);
	const int32_t ref_len = (int32_t)fi_.seq_len[rid];
	t.rid = rid, t.rev = rev;
	int32_t as1, cnt1, rs, qs, re, qe, rs0, qs0, re0, qe0, rs1, qs1, re1, qe1, l;

	if (is_sr) { // align.c:664-669: a short read is aligned from its best run of seeds on one diagonal
		rr_best_diagonal_run(r, a, &as1, &cnt1); // mm_max_stretch, align.c:563-589
		rs = (int32_t)a[as1].x + 1 - span_of(a[as1]), qs = (int32_t)a[as1].y + 1 - span_of(a[as1]);
		re = (int32_t)a[as1 + cnt1 - 1].x + 1, qe = (int32_t)a[as1 + cnt1 - 1].y + 1;
	} else {
		if (!(opt_.flag & F_NO_END_FLT)) {
			if (is_splice) trim_bad_ends_splice(opt_, fi_, r, mat_, qlen, ra, a, tbuf_, &as1, &cnt1);
			else rr_trim_ends(r, a, opt_.bw, opt_.min_chain_score * 2, &as1, &cnt1); // mm_fix_bad_ends, align.c:527-561
		} else as1 = r.as, cnt1 = r.cnt;
		std::vector<int32_t> &K = gap_sites_;
		long_gap_sites(a + as1, cnt1, 10, K);
		rr_drop_compensating_gaps(a + as1, K.data(), (int)K.size(), 40, opt_.max_gap >> 1, 10); // mm_filter_bad_seeds, align.c:454-489
		long_gap_sites(a + as1, cnt1, 30, K);
		rr_join_gap_clusters(a + as1, K.data(), (int)K.size(), opt_.max_gap >> 1);               // mm_filter_bad_seeds_alt, align.c:491-525
		anchor_boundary(fi_, ra, qlen, a[as1], &rs, &qs);
		anchor_boundary(fi_, ra, qlen, a[as1 + cnt1 - 1], &re, &qe);
	}
	assert(cnt1 > 0);
	t.ksw_flag = 0;
	if (is_splice) { // align.c:684-689 and :354
		if (t.splice_flag & F_SPLICE_FOR) t.ksw_flag |= rev ? KSW_SPLICE_REV : KSW_SPLICE_FOR;
		if (t.splice_flag & F_SPLICE_REV) t.ksw_flag |= rev ? KSW_SPLICE_FOR : KSW_SPLICE_REV;
		if (opt_.flag & F_SPLICE_FLANK) t.ksw_flag |= KSW_SPLICE_FLANK;
		if (!(opt_.flag & F_SPLICE_OLD)) t.ksw_flag |= KSW_SPLICE_CMPLX;
		if (fi_.has_spsc) t.ksw_flag |= KSW_SPLICE_SCORE; // align.c:688
	}

	// how far the two extensions may reach (align.c:695-767): region_rules.hpp, one routine for both ends on coordinates that face the end
	if (is_sr) { // the whole read, and as much reference as its unaligned ends could span with gaps
		qs0 = 0, qe0 = qlen;
		rs0 = std::max<int32_t>(rs - rr_sr_reach(qs, opt_.a, opt_.q, opt_.e, opt_.end_bonus), 0);
		re0 = std::min<int32_t>(re + rr_sr_reach(qlen - qe, opt_.a, opt_.q, opt_.e, opt_.end_bonus), ref_len);
	} else {
		const RrExtScoring S = { opt_.a, opt_.q, opt_.e, opt_.max_gap, opt_.min_cnt };
		const Anchor &first = a[r.as], &last = a[r.as + r.cnt - 1];
		assert(rr_y(first) + 1 - rr_span(first) >= 0);
		rr_extension_limit(rr_x(first) + 1 - rr_span(first), rr_y(first) + 1 - rr_span(first), rs, qs,
			[&](int k, int32_t *nt, int32_t *nq) { // earlier seeds of the read on the same target and strand
				const int32_t i = r.as - 1 - k;
				if (i < 0 || !rr_same_target(a[i], first)) return false;
				*nt = rr_x(a[i]) + 1 - rr_span(a[i]), *nq = rr_y(a[i]) + 1 - rr_span(a[i]);
				return true;
			}, S, true, &rs0, &qs0);
		int32_t far_t, far_q; // the right end, in distances from the sequences' ends
		rr_extension_limit(ref_len - (rr_x(last) + 1), qlen - (rr_y(last) + 1), ref_len - re, qlen - qe,
			[&](int k, int32_t *nt, int32_t *nq) { // later seeds
				const int32_t i = r.as + r.cnt + k;
				if (i >= ra.n_a || !rr_same_target(a[i], first)) return false;
				*nt = ref_len - (rr_x(a[i]) + 1), *nq = qlen - (rr_y(a[i]) + 1);
				return true;
			}, S, false, &far_t, &far_q);
		re0 = ref_len - far_t, qe0 = qlen - far_q;
	}
	if (a[r.as].y & SEED_SELF) { // a hit overlapping itself stays on its side of the diagonal (align.c:760-767)
		const int32_t room_l = std::abs(r.qs - r.rs), room_r = std::abs(r.qe - r.re);
		rs0 = rr_self_limit(rs0, r.rs, room_l), qs0 = rr_self_limit(qs0, r.qs, room_l);
		re0 = ref_len - rr_self_limit(ref_len - re0, ref_len - r.re, room_r), qe0 = qlen - rr_self_limit(qlen - qe0, qlen - r.qe, room_r);
	}
	assert(re0 > rs0);
	t.as1 = as1, t.cnt1 = cnt1;
	t.rs0 = rs0, t.qs0 = qs0, t.re0 = re0, t.qe0 = qe0;

	// the windows, in the order the reference aligns them (align.c:779-890)
	t.win.clear();
	t.win.reserve(4); // (left extension, a gap fill or two, right extension: one allocation instead of the growth steps 1, 2, 4 -- a short read's whole plan)
	t.has_left = t.has_right = false;
	if (qs > 0 && rs > 0) {
		Window w; w.kind = W_LEFT, w.qs = qs0, w.qe = qs, w.rs = rs0, w.re = rs, w.bw = bw_, w.anchor_i = 0;
		t.win.push_back(w), t.has_left = true;
	}
	t.rs = rs, t.qs = qs; // start of the first gap window
	for (int32_t i = is_sr ? cnt1 - 1 : 1; i < cnt1; ++i) { // a short read has one window, from its first seed to its last (align.c:803)
		if ((a[as1 + i].y & (SEED_IGNORE | SEED_TANDEM)) && i != cnt1 - 1) continue;
		if (is_sr) re = (int32_t)a[as1 + i].x + 1, qe = (int32_t)a[as1 + i].y + 1;
		else anchor_boundary(fi_, ra, qlen, a[as1 + i], &re, &qe);
		if (i == cnt1 - 1 || (a[as1 + i].y & SEED_LONG_JOIN) || (qe - qs >= opt_.min_ksw_len && re - rs >= opt_.min_ksw_len)) {
			Window w; w.kind = W_GAP, w.qs = qs, w.qe = qe, w.rs = rs, w.re = re, w.anchor_i = i;
			w.bw = bw_long_;
			if (a[as1 + i].y & SEED_LONG_JOIN) w.bw = qe - qs > re - rs ? qe - qs : re - rs;
			w.job = w.saved = -1;
			const bool is_sr_rna = (opt_.flag & F_SR_RNA) && is_splice;
			if (is_sr_rna && qe - qs != re - rs) { // mm_align_sr_rna's own preconditions (align.c:376-384): a short query whose two ends match the window's ends
				const int32_t ql = qe - qs, tl = re - rs, ilen = opt_.q2 * 2;
				if (ql <= 100 && ql * 2 + ilen <= tl) {
					const uint8_t *qseq = strand_codes(ra, qstrand_ ? 0 : rev) + qs;
					tbuf_.resize(tl);
					fi_.getseq2(qstrand_ && rev, rid, rs, re, tbuf_.data());
					int32_t ll = 0, lr = 0;
					for (int32_t j = 0; j < ql; ++j) if (qseq[j] == tbuf_[j] && qseq[j] < 4) ++ll;
					for (int32_t j = 0; j < ql; ++j) if (qseq[ql - 1 - j] == tbuf_[tl - 1 - j] && qseq[ql - 1 - j] < 4) ++lr;
					if (ql - (ll + lr) <= 9) w.pre = 1;
				}
			}
			if (is_sr || (is_sr_rna && qe - qs == re - rs)) { // align.c:823-833: the seeds lie on one diagonal; if the ungapped alignment beats any gapped one, it is the result
				assert(qe - qs == re - rs);
				const int32_t len = qe - qs, max_gapped_score = (len - 2) * opt_.a - 2 * (opt_.q + opt_.e);
				const uint8_t *qseq = strand_codes(ra, qstrand_ ? 0 : rev) + qs;
				tbuf_.resize(len);
				fi_.getseq2(qstrand_ && rev, rid, rs, re, tbuf_.data());
				int32_t score = 0;
				for (int32_t j = 0; j < len; ++j) {
					if (qseq[j] >= 4 || tbuf_[j] >= 4) score += opt_.sc_ambi > 0 ? -opt_.sc_ambi : opt_.sc_ambi;
					else score += qseq[j] == tbuf_[j] ? opt_.a : -opt_.b;
				}
				if (score > max_gapped_score) {
					SavedResult sr;
					memset(&sr.res, 0, sizeof sr.res);
					sr.res.max_q = sr.res.max_t = sr.res.mqe_t = sr.res.mte_q = -1; // ksw_reset_extz (ksw2.h:164-169)
					sr.res.mqe = sr.res.mte = KSW_NEG_INF;
					sr.res.score = score, sr.res.n_cigar = 1;
					sr.res.zd_max = KSW_ZD_NONE, sr.res.zd_t0 = sr.res.zd_t1 = sr.res.zd_q0 = sr.res.zd_q1 = -1;
					sr.cigar.assign(1, (uint32_t)len << 4 | 0u);
					t.saved.push_back(std::move(sr));
					w.saved = (int32_t)t.saved.size() - 1;
				}
			}
			t.win.push_back(w);
			rs = re, qs = qe;
		}
	}
	t.re = re, t.qe = qe; // end of the last gap window == where the right extension starts
	if (qe < qe0 && re < re0) {
		Window w; w.kind = W_RIGHT, w.qs = qe, w.qe = qe0, w.rs = re, w.re = re0, w.bw = bw_, w.anchor_i = 0;
		t.win.push_back(w), t.has_right = true;
	}
	t.next_win = 0, t.dropped = false;
	t.rs1 = t.rs, t.qs1 = t.qs;  // no left extension: the alignment starts at the first anchor (align.c:800)
	t.re1 = t.rs, t.qe1 = t.qs;  // (align.c:801)
}

void Aligner::add_job(ReadAlign &ra, RegionTask &t, Window &w, int flag, int zdrop, int end_bonus, std::vector<KswJob> &jobs)
{
	KswJob j;
	const bool reversed = w.kind == W_LEFT;
	j.qlen = w.qe - w.qs, j.tlen = w.re - w.rs;
	j.w = w.bw, j.zdrop = zdrop, j.end_bonus = end_bonus;
	if (opt_.transition != 0 && opt_.b != opt_.transition) flag |= KSW_GENERIC_SC;             // align.c:347-348
	if (opt_.max_sw_mat > 0 && (int64_t)j.tlen * j.qlen > opt_.max_sw_mat) flag |= KSWJ_SKIP;   // align.c:349-351
	const uint64_t qbase = (t.rev && (!qstrand_ || w.kind == W_INV)) ? ra.qpool_rev : ra.qpool_off, tbase = fi_.seq_off[t.rid]; // the inversion rescue is not query-strand aware in the reference either
	j.q_off = reversed ? qbase + w.qe - 1 : qbase + w.qs;
	j.t_off = reversed ? tbase + w.re - 1 : tbase + w.rs;
	j.flag = flag | t.ksw_flag | KSWJ_T_PACKED | (reversed ? (KSWJ_Q_REVERSED | KSWJ_T_REVERSED) : 0);
	if (qstrand_ && t.rev && w.kind != W_INV) { // --qstrand: the query stays as given and the TARGET is reverse-complemented (mm_idx_getseq2, align.c:780-786):
		const size_t o = ra.tbytes.size();      // such a window is not a run of the packed reference, it is composed into the byte pool
		ra.tbytes.resize(o + (size_t)j.tlen);
		fi_.getseq2(true, t.rid, w.rs, w.re, &ra.tbytes[o]);
		j.t_off = reversed ? o + (uint64_t)j.tlen - 1 : o;
		j.flag &= ~KSWJ_T_PACKED;
	}
	j.tag = 0, j.reserved = 0;
	if ((opt_.flag & F_SPLICE) && fi_.has_spsc && w.kind != W_INV) { // mm_get_junc -> mm_idx_spsc_get (align.c:640, index.c:1045-1066): the scored sites strictly inside the window
		const std::vector<uint64_t> &S = fi_.spsc[(size_t)t.rid << 1 | ((t.ksw_flag & KSW_SPLICE_REV) ? 1 : 0)];
		const size_t first = ra.juncs.size();
		auto last_le = [&](int64_t x) -> int64_t { // index of the last entry at a position <= x, -1 if none (mm_idx_find_intv)
			size_t lo = 0, hi = S.size();
			while (lo < hi) { const size_t mid = lo + ((hi - lo) >> 1); if ((int64_t)(S[mid] >> 8) <= x) lo = mid + 1; else hi = mid; }
			return (int64_t)lo - 1;
		};
		if (!S.empty()) {
			const int64_t l = last_le(w.rs), r = last_le(w.re);
			for (int64_t k = l + 1; k <= r; ++k) {
				const int64_t x = (int64_t)(S[k] >> 8) - w.rs;
				if (x == w.re - w.rs) continue;
				const uint32_t e = (uint32_t)x << 8 | (uint32_t)(S[k] & 0xff);
				if (ra.juncs.size() > first && ra.juncs.back() >> 8 == (uint32_t)x) { if ((ra.juncs.back() & 0xff) < (e & 0xff)) ra.juncs.back() = e; } // the best score of a position
				else ra.juncs.push_back(e);
			}
		}
		j.tag = (uint32_t)first, j.reserved = (uint32_t)(ra.juncs.size() - first);
	} else if ((opt_.flag & F_SPLICE) && fi_.has_junc && w.kind != W_INV) { // mm_get_junc -> mm_idx_bed_junc (align.c:638-643, index.c:803-826): introns lying entirely inside the window
		const std::vector<FlatIndex::Junc> &J = fi_.junc[t.rid];
		const size_t first = ra.juncs.size();
		size_t lo = 0, hi = J.size();
		while (hi > lo) { const size_t mid = lo + ((hi - lo) >> 1); if (J[mid].st >= w.rs) hi = mid; else lo = mid + 1; }
		for (size_t i = lo; i < J.size() && J[i].st < w.re; ++i) {
			if (J[i].en > w.re || J[i].strand == 0) continue;
			ra.juncs.push_back((uint32_t)(J[i].st - w.rs) << 4 | (J[i].strand > 0 ? 1u : 8u));
			ra.juncs.push_back((uint32_t)(J[i].en - 1 - w.rs) << 4 | (J[i].strand > 0 ? 2u : 4u));
		}
		if (ra.juncs.size() > first) { // ascending positions, one entry per position
			std::sort(ra.juncs.begin() + first, ra.juncs.end());
			size_t k = first;
			for (size_t i = first + 1; i < ra.juncs.size(); ++i) {
				if (ra.juncs[i] >> 4 == ra.juncs[k] >> 4) ra.juncs[k] |= ra.juncs[i] & 15u;
				else ra.juncs[++k] = ra.juncs[i];
			}
			ra.juncs.resize(k + 1);
			j.tag = (uint32_t)first, j.reserved = (uint32_t)(k + 1 - first);
		}
	}
	w.job = (int32_t)jobs.size(), w.saved = -1;
	jobs.push_back(j);
}

// mm_align_sr_rna (align.c:370-400): a short query across a long window is first aligned to the window's two ends only -- qlen bases
// from each, 2*q2 Ns in between -- which answers "one clean intron?" with a tiny DP instead of a window-sized one.
void Aligner::add_flank_job(ReadAlign &ra, RegionTask &t, Window &w, std::vector<KswJob> &jobs)
{
	const int32_t ql = w.qe - w.qs, tl = w.re - w.rs, ilen = opt_.q2 * 2, tl2 = ql * 2 + ilen;
	KswJob j;
	j.qlen = ql, j.tlen = tl2;
	j.w = w.bw, j.zdrop = opt_.zdrop, j.end_bonus = -1;
	const int flag = KSW_APPROX_MAX; // ksw_exts2_sse is called directly here (align.c:393): neither KSW_EZ_GENERIC_SC nor the max_sw_mat guard of mm_align_pair
	j.q_off = (t.rev ? ra.qpool_rev : ra.qpool_off) + w.qs;
	j.t_off = ra.tbytes.size();
	ra.tbytes.resize(ra.tbytes.size() + tl2);
	uint8_t *dst = &ra.tbytes[j.t_off];
	tbuf_.resize(tl);
	fi_.getseq2(qstrand_ && t.rev, t.rid, w.rs, w.re, tbuf_.data());
	memcpy(dst, tbuf_.data(), ql);
	memset(dst + ql, 4, ilen);
	memcpy(dst + ql + ilen, tbuf_.data() + tl - ql, ql);
	j.flag = flag | t.ksw_flag; // a byte target: no KSWJ_T_PACKED
	j.tag = 0, j.reserved = 0;
	w.job = (int32_t)jobs.size(), w.saved = -1;
	jobs.push_back(j);
}

void Aligner::schedule(ReadAlign &ra, std::vector<KswJob> &jobs)
{
	ra.juncs.clear(), ra.tbytes.clear();
	for (size_t ti = 0; ti < ra.tasks.size(); ++ti) {
		RegionTask &t = ra.tasks[ti];
		if (t.done) continue;
		if (t.r.inv) { // inversion rescue: one forward extension from the local-alignment start (align.c:944)
			Window &w = t.win[0];
			if (w.job < 0 && w.saved < 0) add_job(ra, t, w, KSW_EXTZ_ONLY, opt_.zdrop, -1, jobs);
			continue;
		}
		if (!t.planned) {
			{ hostprof::Scope hp(hostprof::PLAN_REGION); plan_region(ra, t); }
			if (t.done) continue;
			hostprof::Scope hp(hostprof::ADD_JOBS);
			for (Window &w : t.win) {
				if (w.saved >= 0) continue; // resolved while planning (the ungapped short-read case)
				if (w.kind == W_LEFT) add_job(ra, t, w, KSW_EXTZ_ONLY | KSW_RIGHT | KSW_REV_CIGAR, t.r.split_inv ? opt_.zdrop_inv : opt_.zdrop, opt_.end_bonus, jobs);
				else if (w.kind == W_GAP && w.pre == 1) add_flank_job(ra, t, w, jobs);
				else if (w.kind == W_GAP) add_job(ra, t, w, KSW_APPROX_MAX, opt_.zdrop, -1, jobs);
				else add_job(ra, t, w, KSW_EXTZ_ONLY, opt_.zdrop, opt_.end_bonus, jobs);
			}
		} else if (t.next_win < t.win.size()) { // stalled on a second pass (align.c:843-844)
			Window &w = t.win[t.next_win];
			if (w.pass2 && w.job < 0 && w.saved < 0) add_job(ra, t, w, 0, w.zdrop_code == 2 ? opt_.zdrop_inv : opt_.zdrop, -1, jobs);
			else if (!w.pass2 && w.pre == 2 && w.job < 0 && w.saved < 0) add_job(ra, t, w, KSW_APPROX_MAX, opt_.zdrop, -1, jobs); // the flank-only attempt failed (align.c:839-840)
		}
	}
}

bool Aligner::consume(ReadAlign &ra, const KswRes *res, const uint32_t *cigar_pool)
{
	bool pending = false;
	const size_t n0 = ra.tasks.size(); // tasks appended while consuming are picked up by the next schedule()
	for (size_t ti = 0; ti < n0; ++ti) {
		if (ra.tasks[ti].done) continue;
		if (ra.tasks[ti].r.inv) consume_inversion(ra, (int)ti, res, cigar_pool);
		else pending |= consume_region(ra, (int)ti, res, cigar_pool);
	}
	for (size_t ti = n0; ti < ra.tasks.size(); ++ti) pending |= !ra.tasks[ti].done;
	return pending;
}

// A consumed window's CIGAR: appended right away (host mode), or recorded for the device's region_finish
void Aligner::take_piece(RegionTask &t, const Window &w, const KswRes &ez, const uint32_t *cg, const uint32_t *cigar_pool)
{
	if (!t.host_mode && w.saved >= 0) materialize(t, cigar_pool); // a result kept from an earlier step is not in the round's pool
	if (t.host_mode) { append_cigar(t.r, (uint32_t)ez.n_cigar, cg); return; }
	const uint32_t n = (uint32_t)ez.n_cigar;
	t.pieces.push_back(FinPiece{ ez.cigar_off, n });
	// the container's growth, as enlarge_cigar / append_cigar would have done it
	if (t.sim_cap == 0) t.sim_cap = roundup32(n + kExtraWords);
	else if (t.sim_n + n + kExtraWords > t.sim_cap) t.sim_cap = roundup32(t.sim_n + n + kExtraWords);
	t.sim_n += t.sim_n > 0 && t.sim_last_op == (cg[0] & 0xf) ? n - 1 : n;
	t.sim_last_op = cg[n - 1] & 0xf;
}

// The region goes on in the reference's way: append what was recorded (the pieces address the CURRENT round's pool) and carry the score over
void Aligner::materialize(RegionTask &t, const uint32_t *cigar_pool)
{
	if (t.host_mode) return;
	t.host_mode = true;
	for (const FinPiece &pc : t.pieces) append_cigar(t.r, pc.n, cigar_pool + pc.off);
	t.pieces.clear();
	if (!t.r.p && (t.dropped || t.dp_acc != 0)) { // (the reference creates the container when a Z-drop cuts a region that has no CIGAR yet, align.c:849)
		const uint32_t cap = roundup32(kExtraWords);
		t.r.p = (Extra *)calloc(cap, 4);
		t.r.p->capacity = cap;
	}
	if (t.r.p) t.r.p->dp_score += t.dp_acc;
	t.dp_acc = 0;
}

void Aligner::describe_finish(const ReadAlign &ra, int ti, FinRegion &fr) const
{
	const RegionTask &t = ra.tasks[ti];
	fr.q_pos = (t.r.rev ? ra.qpool_rev : ra.qpool_off) + (uint64_t)t.qs1;
	fr.t_pos = fi_.seq_off[t.rid] + (uint64_t)t.rs1;
	fr.n_pieces = (uint32_t)t.pieces.size();
	fr.q_len = t.qe1 - t.qs1, fr.t_len = t.re1 - t.rs1;
}

bool Aligner::complete_finished(ReadAlign &ra, const FinResult *results, const FinRegion *regions, const uint32_t *cigars)
{
	hostprof::Scope hp(hostprof::COMPLETE_FINISHED);
	const std::vector<int> queue = ra.finish_queue; // (after_finalize may queue nothing, but it does touch ra.tasks)
	ra.finish_queue.clear();
	for (size_t k = 0; k < queue.size(); ++k) {
		const int ti = queue[k];
		const FinResult &f = results[k];
		if (f.n_cigar < 0) throw std::runtime_error("[mm2amd] region_finish: a stitched CIGAR does not cover its windows");
		{
			RegionTask &t = ra.tasks[ti];
			Reg &r = t.r;
			const uint32_t cap = t.sim_cap; // >= the stitched length + header, and mm_fix_cigar only shortens
			if ((uint32_t)f.n_cigar + kExtraWords > cap) throw std::runtime_error("[mm2amd] region_finish: CIGAR longer than its windows' CIGARs");
			r.p = (Extra *)calloc(cap, 4);
			r.p->capacity = cap;
			r.p->n_cigar = (uint32_t)f.n_cigar;
			memcpy(r.p->cigar, cigars + regions[k].out_off, (size_t)f.n_cigar * 4);
			r.p->dp_score = t.dp_acc, r.p->dp_max = r.p->dp_max0 = f.dp_max, r.p->n_ambi = (uint32_t)f.n_ambi;
			r.blen = f.blen, r.mlen = f.mlen, r.is_spliced = f.is_spliced;
			if (f.qshift) { if (r.rev) r.qe -= f.qshift; else r.qs += f.qshift; } // mm_fix_cigar's dropped leading gap (align.c:171-180)
			r.rs += f.tshift;
			static const bool fin_check = getenv("MM2AMD_FIN_CHECK") != nullptr; // (once, not per region)
			if (fin_check && (r.qs < 0 || r.qe > ra.qlen || r.qs >= r.qe))
				fprintf(stderr, "[mm2amd] FIN_CHECK bad query range: qs %d qe %d qlen %d rev %d | qs1 %d qe1 %d rs1 %d re1 %d | qshift %d tshift %d n_cigar %d | has_left %d win0 kind %d qs %d qe %d job %d\n", r.qs, r.qe, ra.qlen,
				        (int)r.rev, t.qs1, t.qe1, t.rs1, t.re1, f.qshift, f.tshift, f.n_cigar, (int)t.has_left, t.win.empty() ? -1 : (int)t.win[0].kind, t.win.empty() ? 0 : t.win[0].qs, t.win.empty() ? 0 : t.win[0].qe, t.win.empty() ? 0 : t.win[0].job);
			t.pieces.clear(), t.dp_acc = 0, t.awaiting_finish = false;
			t.saved.clear();
			t.done = true;
		}
		after_finalize(ra, ti);
	}
	for (const RegionTask &t : ra.tasks) if (!t.done) return true;
	return false;
}

bool Aligner::consume_region(ReadAlign &ra, int ti, const KswRes *res, const uint32_t *cigar_pool)
{
	Anchor *a = ra.a;
	const int qlen = ra.qlen;
	{
		RegionTask &t = ra.tasks[ti];
		Reg &r = t.r;
		hostprof::Scope hp(hostprof::CONSUME_WINDOWS);
		if (!device_finish_) t.host_mode = true;
		auto add_score = [&](int32_t v) { if (t.host_mode) r.p->dp_score += v; else t.dp_acc += v; };
		while (t.next_win < t.win.size()) {
			Window &w = t.win[t.next_win];
			if (w.job < 0 && w.saved < 0) { materialize(t, cigar_pool); return true; } // its job has not run yet
			const KswRes &ez = w.saved >= 0 ? t.saved[w.saved].res : res[w.job];
			const uint32_t *cg = w.saved >= 0 ? t.saved[w.saved].cigar.data() : cigar_pool + ez.cigar_off;
			if (w.kind == W_LEFT) {
				if (ez.n_cigar > 0) { take_piece(t, w, ez, cg, cigar_pool); add_score(ez.max); }
				t.rs1 = w.re - (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
				t.qs1 = w.qe - (ez.reach_end ? w.qe - w.qs : ez.max_q + 1);
				++t.next_win;
			} else if (w.kind == W_GAP) {
				if (w.pre == 1) { // the flank-only attempt came back (align.c:394-399): it counts only as one clean intron between two matches
					bool ok = !ez.zdropped && ez.n_cigar > 0 && (cg[0] & 0xf) == 0 && (cg[ez.n_cigar - 1] & 0xf) == 0;
					int nn = 0, n_ins = 0;
					for (int32_t k = 0; ok && k < ez.n_cigar; ++k) nn += (cg[k] & 0xf) == 3, n_ins += (cg[k] & 0xf) == 1;
					ok = ok && nn == 1 && n_ins == 0;
					w.pre = 2;
					if (ok) { // becomes the window's first-pass result, the intron stretched back to the window's length
						SavedResult sr;
						sr.res = ez;
						sr.res.zd_max = KSW_ZD_NONE; // the kernel scanned the composed target; the host rescans the real window
						sr.cigar.assign(cg, cg + ez.n_cigar);
						const int32_t tl = w.re - w.rs, tl2 = (w.qe - w.qs) * 2 + opt_.q2 * 2;
						for (uint32_t &c : sr.cigar) if ((c & 0xf) == 3) c += (uint32_t)(tl - tl2) << 4;
						t.saved.push_back(std::move(sr));
						w.saved = (int32_t)t.saved.size() - 1, w.job = -1;
						continue; // re-enter with the saved result
					}
					for (size_t k = t.next_win + 1; k < t.win.size(); ++k) { // keep the later windows' results of this round
						Window &wk = t.win[k];
						if (wk.job >= 0 && wk.saved < 0) {
							SavedResult sr;
							sr.res = res[wk.job];
							sr.cigar.assign(cigar_pool + sr.res.cigar_off, cigar_pool + sr.res.cigar_off + sr.res.n_cigar);
							t.saved.push_back(std::move(sr));
							wk.saved = (int32_t)t.saved.size() - 1, wk.job = -1;
						}
					}
					w.job = -1, w.saved = -1;
					materialize(t, cigar_pool);
					return true;
				}
				if (!w.pass2) { // the approximate pass: test it (align.c:843)
					auto qseq_of = [&]() { return strand_codes(ra, qstrand_ ? 0 : t.rev) + w.qs; }; // (encodes the strand: only when the test needs the bases)
					int code;
					if (ez.zd_max != KSW_ZD_NONE) { // the kernel scanned its own alignment; sequences are only needed for the rare inversion check
						ZdropScan z;
						z.max_zdrop = ez.zd_max, z.pos[0][0] = ez.zd_t0, z.pos[0][1] = ez.zd_t1, z.pos[1][0] = ez.zd_q0, z.pos[1][1] = ez.zd_q1;
						if (z.max_zdrop > opt_.zdrop_inv) {
							tbuf_.resize(w.re - w.rs);
							fi_.getseq2(qstrand_ && t.rev, t.rid, w.rs, w.re, tbuf_.data());
							code = zdrop_decide(opt_, z, qseq_of(), tbuf_.data(), mat_);
						} else code = z.max_zdrop > opt_.zdrop ? 1 : 0;
					} else {
						tbuf_.resize(w.re - w.rs);
						fi_.getseq2(qstrand_ && t.rev, t.rid, w.rs, w.re, tbuf_.data());
						code = test_zdrop(opt_, qseq_of(), tbuf_.data(), ez.n_cigar, cg, mat_);
					}
					if (code != 0) {
						// keep the results of the later windows of this region: they belong to this round
						for (size_t k = t.next_win + 1; k < t.win.size(); ++k) {
							Window &wk = t.win[k];
							if (wk.job >= 0 && wk.saved < 0) {
								SavedResult sr;
								sr.res = res[wk.job];
								sr.cigar.assign(cigar_pool + sr.res.cigar_off, cigar_pool + sr.res.cigar_off + sr.res.n_cigar);
								t.saved.push_back(std::move(sr));
								wk.saved = (int32_t)t.saved.size() - 1, wk.job = -1;
							}
						}
						w.pass2 = true, w.zdrop_code = code, w.job = -1, w.saved = -1;
						materialize(t, cigar_pool);
						return true;
					}
				}
				if (ez.n_cigar > 0) take_piece(t, w, ez, cg, cigar_pool);
				if (ez.zdropped) { // truncated: cut the region here, maybe split off the rest (align.c:848-868)
					if (t.host_mode && !r.p) {
						const uint32_t cap = roundup32(kExtraWords);
						r.p = (Extra *)calloc(cap, 4);
						r.p->capacity = cap;
					}
					int j;
					for (j = w.anchor_i - 1; j >= 0; --j)
						if ((int32_t)a[t.as1 + j].x <= w.rs + ez.max_t) break;
					t.dropped = true;
					if (j < 0) j = 0;
					add_score(ez.max);
					t.re1 = w.rs + (ez.max_t + 1);
					t.qe1 = w.qs + (ez.max_q + 1);
					if (t.cnt1 - (j + 1) >= opt_.min_cnt) {
						split_reg(r, t.r2, t.as1 + j + 1 - r.as, qlen, a, qstrand_);
						if (w.zdrop_code == 2) t.r2.split_inv = 1;
					}
					t.next_win = t.win.size();
					break;
				}
				add_score(ez.score);
				t.re1 = w.re, t.qe1 = w.qe;
				++t.next_win;
			} else { // W_RIGHT; only reached when nothing was dropped (align.c:874)
				if (ez.n_cigar > 0) { take_piece(t, w, ez, cg, cigar_pool); add_score(ez.max); }
				t.re1 = w.rs + (ez.reach_end ? ez.mqe_t + 1 : ez.max_t + 1);
				t.qe1 = w.qs + (ez.reach_end ? w.qe - w.qs : ez.max_q + 1);
				++t.next_win;
			}
		}
		hp.stop();
		if (!t.host_mode && device_finish_ && !qstrand_ && !(opt_.flag & F_EQX)) { // every window came back in this round: the device stitches and finishes it
			size_t n_ops = 0;
			for (const FinPiece &pc : t.pieces) n_ops += pc.n;
			if (n_ops > 0 && n_ops <= (size_t)kFinMaxOps) {
				const int qlen = ra.qlen;
				r.rs = t.rs1, r.re = t.re1;
				if (!t.rev) r.qs = t.qs1, r.qe = t.qe1; // align.c:894
				else r.qs = qlen - t.qe1, r.qe = qlen - t.qs1;
				t.awaiting_finish = true;
				ra.finish_queue.push_back(ti);
				return true;
			}
		}
		materialize(t, cigar_pool);
		finalize_region(ra, t);
	}
	return after_finalize(ra, ti);
}

// what follows a region's completion in mm_align_skeleton (align.c:1078-1108); returns whether it created more work
bool Aligner::after_finalize(ReadAlign &ra, int ti)
{
	if (opt_.flag & F_SPLICE) {
		const int twin = ra.tasks[ti].twin;
		if (twin >= 0) { // two strand attempts: the second one to finish picks the winner (align.c:1078-1096)
			if (!ra.tasks[twin].done) return false;
			if (!ra.tasks[ti].lead) ti = twin;
			join_strands(ra, ti);
		} else ra.tasks[ti].r.p->trans_strand = (opt_.flag & F_SPLICE_FOR) ? 1 : 2; // align.c:1099-1100
	}
	// follow-up work.  NB: ra.tasks may reallocate below, so no references are held across push_back.
	const int self_pos = (int)(std::find(ra.order.begin(), ra.order.end(), ti) - ra.order.begin());
	bool more = false;
	if (ra.tasks[ti].r2.cnt > 0) { // the split-off tail becomes a region right after this one (align.c:1102)
		const Reg tail = ra.tasks[ti].r2;
		add_region(ra, tail, self_pos + 1);
		more = true;
	}
	if (self_pos > 0 && ra.tasks[ti].r.split_inv && !(opt_.flag & F_NO_INV)) { // inversion rescue (align.c:1103-1108)
		const size_t before = ra.tasks.size();
		try_inversion(ra, ra.order[self_pos - 1], ti, self_pos);
		more |= ra.tasks.size() > before;
	}
	return more;
}

void Aligner::finalize_region(ReadAlign &ra, RegionTask &t)
{
	Reg &r = t.r;
	const int qlen = ra.qlen;
	assert(t.qe1 <= qlen);
	r.rs = t.rs1, r.re = t.re1;
	if (!t.rev || qstrand_) r.qs = t.qs1, r.qe = t.qe1; // align.c:894
	else r.qs = qlen - t.qe1, r.qe = qlen - t.qs1;
	if (r.p) {
		{
			hostprof::Scope hp(hostprof::GETSEQ);
			tbuf_.resize((size_t)(t.re1 - t.rs1) + 16); // update_extra compares 16 columns per load
			fi_.getseq2(qstrand_ && t.rev, t.rid, t.rs1, t.re1, tbuf_.data());
		}
		const uint8_t *qseq = strand_codes(ra, qstrand_ ? 0 : r.rev) + t.qs1;
		update_extra(r, qseq, tbuf_.data(), mat_, (int8_t)opt_.q, (int8_t)opt_.e, opt_.flag & F_EQX, !(opt_.flag & (F_SR | F_SR_RNA)));
		if (t.rev && r.p->trans_strand) r.p->trans_strand ^= 3; // align.c:907-908
	}
	t.saved.clear();
	t.done = true;
}

// mm_align1_inv (align.c:916-971), first half: decide whether the gap between a region and its split-off
// successor looks like an inversion (local alignment of the reverse strand), and if so queue the extension.
void Aligner::try_inversion(ReadAlign &ra, int prev_ti, int ti, int pos_in_order)
{
	const Reg r1 = ra.tasks[prev_ti].r, r2 = ra.tasks[ti].r;
	const int qlen = ra.qlen;
	if (!(r1.split & 1) || !(r2.split & 2)) return;
	if (r1.id != r1.parent && r1.parent != PARENT_TMP_PRI) return;
	if (r2.id != r2.parent && r2.parent != PARENT_TMP_PRI) return;
	if (r1.rid != r2.rid || r1.rev != r2.rev) return;
	const int ql = r1.rev ? r1.qs - r2.qe : r2.qs - r1.qe, tl = r2.rs - r1.re;
	if (ql < opt_.min_chain_score || ql > opt_.max_gap) return;
	if (tl < opt_.min_chain_score || tl > opt_.max_gap) return;
	const int strand = r1.rev ? 0 : 1;                      // the query is read on the strand opposite to the flanks
	const int q0 = r1.rev ? r2.qe : qlen - r2.qs;
	std::vector<uint8_t> qrev(ql), trev(tl);
	const uint8_t *qseq = strand_codes(ra, strand) + q0;
	for (int i = 0; i < ql; ++i) qrev[i] = qseq[ql - 1 - i];
	fi_.getseq(r1.rid, r1.re, r2.rs, trev.data());
	std::reverse(trev.begin(), trev.end());
	int q_off, t_off;
	const int score = ll_local_score(ql, qrev.data(), tl, trev.data(), mat_, opt_.q, opt_.e, &q_off, &t_off);
	if (score < opt_.min_dp_max) return;
	q_off = ql - (q_off + 1), t_off = tl - (t_off + 1);
	RegionTask nt;
	memset(&nt.r, 0, sizeof(Reg));
	nt.r.inv = 1; // marks the task kind; the remaining fields are filled when the extension comes back
	nt.planned = true;
	nt.rev = strand, nt.rid = r1.rid;
	nt.inv_q0 = q0, nt.inv_qoff = q_off, nt.inv_toff = t_off;
	nt.inv_r2_qs = r2.qs, nt.inv_r2_qe = r2.qe, nt.inv_r1_re = r1.re;
	nt.r.rev = !r1.rev, nt.r.rid = r1.rid;
	Window w;
	w.kind = W_INV, w.qs = q0 + q_off, w.qe = q0 + ql, w.rs = r1.re + t_off, w.re = r2.rs, w.bw = (int)(opt_.bw * 1.5), w.anchor_i = 0;
	nt.win.push_back(w);
	ra.tasks.push_back(std::move(nt));
	// if it succeeds it is reported right after region `ti`, before anything split off from `ti` (align.c:1105-1106)
	ra.order.insert(ra.order.begin() + pos_in_order + 1, -(int)ra.tasks.size()); // negative = provisional, see consume_inversion
}

// mm_align1_inv, second half: turn the extension result into an inversion hit, or drop it.
void Aligner::consume_inversion(ReadAlign &ra, int ti, const KswRes *res, const uint32_t *cigar_pool)
{
	RegionTask &t = ra.tasks[ti];
	Window &w = t.win[0];
	if (w.job < 0) return;
	const KswRes &ez = res[w.job];
	auto slot = std::find(ra.order.begin(), ra.order.end(), -(ti + 1));
	t.done = true;
	if (ez.n_cigar == 0) { if (slot != ra.order.end()) ra.order.erase(slot); return; }
	Reg &ri = t.r;
	const bool rev = ri.rev;
	const int32_t rid = ri.rid;
	memset(&ri, 0, sizeof(Reg));
	append_cigar(ri, ez.n_cigar, cigar_pool + ez.cigar_off);
	ri.p->dp_score = ez.max;
	ri.id = -1, ri.parent = PARENT_UNSET, ri.inv = 1, ri.rev = rev, ri.rid = rid, ri.div = -1.0f;
	if (ri.rev == 0) ri.qs = t.inv_r2_qe + t.inv_qoff, ri.qe = ri.qs + ez.max_q + 1;
	else ri.qe = t.inv_r2_qs - t.inv_qoff, ri.qs = ri.qe - (ez.max_q + 1);
	ri.rs = t.inv_r1_re + t.inv_toff, ri.re = ri.rs + ez.max_t + 1;
	tbuf_.resize((size_t)(w.re - w.rs) + 16);
	fi_.getseq(rid, w.rs, w.re, tbuf_.data());
	update_extra(ri, strand_codes(ra, t.rev) + w.qs, tbuf_.data(), mat_, (int8_t)opt_.q, (int8_t)opt_.e, opt_.flag & F_EQX,
	             !(opt_.flag & (F_SR | F_SR_RNA)));
	if (slot != ra.order.end()) *slot = ti;
}

void Aligner::finish_read(ReadAlign &ra, RegVec &out)
{
	hostprof::Scope hp(hostprof::FINISH_READ);
	out.clear();
	for (int ti : ra.order)
		if (ti >= 0) out.push_back(ra.tasks[ti].r);
	finish_regs(ra.qlen, out);
}

void Aligner::finish_regs(int qlen, RegVec &out) const // align.c:1110-1118
{
	filter_regs(opt_, qlen, out);
	if (!(opt_.flag & (F_SR | F_SR_RNA | F_ALL_CHAINS)) && !opt_.split_prefix && qlen >= opt_.rank_min_len) {
		update_dp_max(qlen, out, opt_.rank_frac, opt_.a, opt_.b);
		filter_regs(opt_, qlen, out);
	}
	hit_sort(out, opt_.alt_drop);
}

} // namespace mm2amd

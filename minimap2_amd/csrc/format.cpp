// Host output stage: the text the reference writes in step 2 of its pipeline (map.c:585-623) -- SAM
// records as mm_write_sam3 (format.c:522-679) and PAF records as mm_write_paf4 (format.c:425-458) produce them, with the tag
// block of write_tags (:397-423) and the cs / ds / MD strings of write_cs_ds_core / write_MD_core (:171-254, :302-331) --
// formatted for a whole mini-batch on the host thread pool instead of by the single pipeline thread.  Once mapping is an
// order of magnitude faster, that single thread is the Amdahl term; records are independent, so reads are cut into chunks,
// every chunk is formatted into its own buffer by one pool thread, and the buffers are concatenated in input order.
// Single-segment fragments only (the mapper's scope); no read-group tag (-R is CLI state of the reference).
#include <array>
#include <atomic>
#include <chrono>
#include <cassert>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>
#include "format.hpp"
#include "threads.hpp"
#include "host_prof.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace mm2amd {

using namespace ref;

extern const uint8_t kNt4Table[256];

namespace {

constexpr int64_t F_OUT_CG = 0x020, F_OUT_CS = 0x040, F_OUT_CS_LONG = 0x800, F_LONG_CIGAR = 0x10000, F_SOFTCLIP = 0x80000, F_OUT_MD = 0x1000000,
	F_COPY_COMMENT = 0x2000000, F_PAF_NO_HIT = 0x8000000, F_SAM_HIT_ONLY = 0x40000000, F_SECONDARY_SEQ = 0x1000000000LL, F_OUT_DS = 0x2000000000LL,
	F_OUT_JUNC = 0x10000000000LL;
const char kCigarOps[] = "MIDNSHP=XB";

// append-only byte buffer.  One per chunk of fragments, in a vector, each filled by a different pool thread: the object is padded to
// two cache lines, because std::string updates its length field on EVERY appended byte and neighbouring objects sharing a line made 64
// threads on two sockets ping-pong it -- 20-40 ns per byte instead of 1.3 (profiles/r03: 25-50 core-seconds per Gbase of reads).
struct alignas(128) Text {
	char *p = nullptr;
	size_t n = 0, cap = 0;
	Text() = default;
	Text(Text &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr, o.n = o.cap = 0; }
	Text(const Text &) = delete;
	Text &operator=(const Text &) = delete;
	~Text() { free(p); }
	void clear() { n = 0; }
	size_t size() const { return n; }
	const char *data() const { return p; }
	char *need(size_t k) // room for k more bytes; returns where they go (the caller advances n)
	{
		if (n + k > cap) {
			size_t c = cap ? cap : 4096;
			while (c < n + k) c += c / 2 + 64;
			char *q = (char *)realloc(p, c);
			if (!q) throw std::bad_alloc();
			p = q, cap = c;
		}
		return p + n;
	}
	void ch(char c) { *need(1) = c, ++n; }
	void str(const char *s) { str(s, strlen(s)); }
	void str(const char *s, size_t l) { memcpy(need(l), s, l), n += l; }
	static char *put_u64(char *w, uint64_t x) // decimal digits of x at w; returns the end
	{
		char buf[20];
		int l = 0;
		do buf[l++] = (char)('0' + x % 10), x /= 10; while (x);
		while (l) *w++ = buf[--l];
		return w;
	}
	static char *put_u32(char *w, uint32_t x) // the common case: CIGAR lengths and flags, one to three digits
	{
		if (x < 10) { *w++ = (char)('0' + x); return w; }
		if (x < 100) { w[0] = (char)('0' + x / 10), w[1] = (char)('0' + x % 10); return w + 2; }
		if (x < 1000) { w[0] = (char)('0' + x / 100), w[1] = (char)('0' + x / 10 % 10), w[2] = (char)('0' + x % 10); return w + 3; }
		return put_u64(w, x);
	}
	void num(int64_t v)
	{
		char *w = need(21);
		if (v < 0) *w++ = '-';
		const uint64_t x = v < 0 ? (uint64_t)(-v) : (uint64_t)v;
		w = x <= UINT32_MAX ? put_u32(w, (uint32_t)x) : put_u64(w, x);
		n = (size_t)(w - p);
	}
	void tag(const char *name, int64_t v) { ch('\t'), str(name), num(v); } // "\tNM:i:" + value
	// (round 5) the writer of a CIGAR's text also counts its gaps -- what the de:f tag of the same record needs (mm_count_gaps): the record's second and third
	// walk over the 1400 operations of a 10 kb read were a third of the formatting time
	// (round 6) the counts belong to the Text they were made for, not to the thread: whoever writes a record's tags into this object finds what the
	// same object's cigar() counted, keyed by the CIGAR's address and length; a caller with a fresh Text finds nothing and counts itself
	struct GapCount { const uint32_t *c = nullptr; uint32_t n = 0; int n_gap = 0, n_gapo = 0; } gaps;
	void cigar(const uint32_t *c, uint32_t n_cigar) // <len><op> per entry
	{
		int n_gap = 0, n_gapo = 0;
		// lengths below 1000 (all but a handful per read) from a table: four bytes stored, the digit count added -- no branch on the number of digits
		static const struct Small { uint32_t chars[1000]; uint8_t len[1000]; Small() { for (uint32_t x = 0; x < 1000; ++x) { char b[8] = {0}; len[x] = (uint8_t)(put_u32(b, x) - b); memcpy(&chars[x], b, 4); } } } small;
		char *w = need((size_t)n_cigar * 11 + 4);
		for (uint32_t k = 0; k < n_cigar; ++k) {
			const uint32_t x = c[k] >> 4, op = c[k] & 0xf;
			if (x < 1000) memcpy(w, &small.chars[x], 4), w += small.len[x];
			else w = put_u32(w, x);
			*w++ = kCigarOps[op];
			const uint32_t is_gap = (op - 1u) < 2u; // I or D
			n_gapo += (int)is_gap, n_gap += (int)(is_gap ? x : 0u);
		}
		n = (size_t)(w - p);
		gaps.c = c, gaps.n = n_cigar, gaps.n_gap = n_gap, gaps.n_gapo = n_gapo;
	}
};

struct Seqs { std::vector<uint8_t> q, t; std::string tmp; }; // per-thread scratch for cs/ds/MD

void count_gaps(const Text &o, const Reg1 &r, int *n_gap, int *n_gapo) // mm_count_gaps (align.c:985-995)
{
	*n_gap = *n_gapo = 0;
	if (!r.p) return;
	const Text::GapCount &g = o.gaps;
	if (g.c == r.p->cigar && g.n == r.p->n_cigar) { *n_gap = g.n_gap, *n_gapo = g.n_gapo; return; } // counted while this record's CIGAR text was written
	for (uint32_t i = 0; i < r.p->n_cigar; ++i) {
		const int op = r.p->cigar[i] & 0xf, len = r.p->cigar[i] >> 4;
		if (op == 1 || op == 2) ++*n_gapo, *n_gap += len;
	}
}

double event_identity(const Text &o, const Reg1 &r) // mm_event_identity (align.c:997-1003)
{
	if (!r.p) return -1.0f;
	int n_gap, n_gapo;
	count_gaps(o, r, &n_gap, &n_gapo);
	return (double)r.mlen / (r.blen + r.p->n_ambi - n_gap + n_gapo);
}

// "%.4f" of v in [0, 1] exactly as printf rounds it: the double is M x 2^-k, so v x 10^4 = M x 10^4 / 2^k is an exact 67-bit quotient and remainder -- round to
// nearest, ties to even on the exact value (what glibc's printf does with its big-number arithmetic).  Anything else goes to snprintf.
bool put_fraction_exact(Text &o, double v)
{
	if (!(v >= 0.0 && v <= 1.0)) return false;
	uint64_t bits;
	memcpy(&bits, &v, 8);
	const int be = (int)(bits >> 52 & 0x7ff);
	uint64_t M = bits & ((1ull << 52) - 1);
	int k; // v = M x 2^-k
	if (be == 0) k = 1074; else M |= 1ull << 52, k = 1075 - be;
	unsigned q;
	if (k <= 0) q = (unsigned)(M << -k) * 10000u; // (v == 1.0: M = 2^52, k = 52 -- never here; kept for completeness)
	else if (k >= 120) q = 0; // below 2^-67: rounds to 0.0000
	else {
		const unsigned __int128 N = (unsigned __int128)M * 10000u, one = (unsigned __int128)1 << k;
		const unsigned __int128 quo = N >> k, rem = N & (one - 1), half = one >> 1;
		q = (unsigned)quo;
		if (rem > half || (rem == half && (q & 1u))) ++q;
	}
	char *w = o.need(8);
	w[0] = (char)('0' + q / 10000), w[1] = '.';
	w[2] = (char)('0' + q / 1000 % 10), w[3] = (char)('0' + q / 100 % 10), w[4] = (char)('0' + q / 10 % 10), w[5] = (char)('0' + q % 10);
	o.n += 6;
	return true;
}
void put_fraction(Text &o, double v) // "0" or %.4f (format.c:413-414, :418-419)
{
	if (v == 0.0) { o.ch('0'); return; }
	if (put_fraction_exact(o, v)) return;
	char buf[16];
	snprintf(buf, 16, "%.4f", v);
	o.str(buf);
}

void put_tags(Text &o, const Reg1 &r) // write_tags
{
	const char type = r.id == r.parent ? (r.inv ? 'I' : 'P') : (r.inv ? 'i' : 'S');
	if (r.p) {
		o.tag("NM:i:", r.blen - r.mlen + (int)r.p->n_ambi), o.tag("ms:i:", r.p->dp_max0), o.tag("AS:i:", r.p->dp_score), o.tag("nn:i:", (int)r.p->n_ambi);
		if (r.p->trans_strand == 1 || r.p->trans_strand == 2) o.str("\tts:A:"), o.ch("?+-?"[r.p->trans_strand]);
	}
	o.str("\ttp:A:"), o.ch(type), o.tag("cm:i:", r.cnt), o.tag("s1:i:", r.score);
	if (r.parent == r.id) o.tag("s2:i:", r.subsc);
	if (r.p) {
		o.str("\tde:f:");
		const double div = 1.0 - event_identity(o, r);
		if (div == 0.0) o.ch('0'); else put_fraction(o, div);
	} else if (r.div >= 0.0f && r.div <= 1.0f) {
		o.str("\tdv:f:");
		if (r.div == 0.0f) o.ch('0'); else put_fraction(o, r.div);
	}
	if (r.split) o.tag("zd:i:", r.split);
}

// the aligned stretches of query and target as nt4 codes, query on the alignment strand (write_cs_ds_or_MD, format.c:333-362)
void fetch_pair(const FlatIndex &fi, const Bseq1 &t, const Reg1 &r, Seqs &sq, bool is_qstrand)
{
	sq.q.resize(r.qe - r.qs), sq.t.resize(r.re - r.rs);
	fi.getseq2(is_qstrand && r.rev, r.rid, r.rs, r.re, sq.t.data()); // format.c:343-346: query as given, reference reverse-complemented
	if (!r.rev || is_qstrand) for (int i = r.qs; i < r.qe; ++i) sq.q[i - r.qs] = kNt4Table[(uint8_t)t.seq[i]];
	else for (int i = r.qs; i < r.qe; ++i) { const uint8_t c = kNt4Table[(uint8_t)t.seq[i]]; sq.q[r.qe - i - 1] = c >= 4 ? 4 : 3 - c; }
}

void put_bases(Text &o, const char *alphabet, const uint8_t *s, int64_t n) { for (int64_t i = 0; i < n; ++i) o.ch(alphabet[s[i]]); }

void put_ds_indel(Text &o, int64_t len, const uint8_t *seq, int64_t ll, int64_t lr) // write_indel_ds (format.c:142-169)
{
	if (ll + lr >= len) { o.ch('['), put_bases(o, "acgtn", seq, len), o.ch(']'); return; }
	int64_t k = 0;
	if (ll > 0) o.ch('['), put_bases(o, "acgtn", seq, ll), o.ch(']'), k += ll;
	put_bases(o, "acgtn", seq + k, len - lr - ll), k += len - lr - ll;
	if (lr > 0) o.ch('['), put_bases(o, "acgtn", seq + k, lr), o.ch(']');
}

void put_cs(Text &o, const Reg1 &r, const Seqs &sq, bool no_iden, bool is_ds) // write_cs_ds_core
{
	const uint8_t *qs = sq.q.data(), *ts = sq.t.data();
	o.str(is_ds ? "\tds:Z:" : "\tcs:Z:");
	int q_len = 0, t_len = 0;
	for (uint32_t i = 0; i < r.p->n_cigar; ++i) {
		const int op = r.p->cigar[i] & 0xf, len = r.p->cigar[i] >> 4;
		if (op == 0 || op == 7 || op == 8) q_len += len, t_len += len;
		else if (op == 1) q_len += len;
		else if (op == 2 || op == 3) t_len += len;
	}
	int q_off = 0, t_off = 0;
	for (uint32_t i = 0; i < r.p->n_cigar; ++i) {
		const int op = r.p->cigar[i] & 0xf, len = r.p->cigar[i] >> 4;
		if (op == 0 || op == 7 || op == 8) {
			int run = 0; // identical bases seen since the last difference
			auto flush = [&](int end) {
				if (run == 0) return;
				if (no_iden) o.ch(':'), o.num(run);
				else o.ch('='), put_bases(o, "ACGTN", qs + q_off + end - run, run);
				run = 0;
			};
			for (int j = 0; j < len; ++j) {
				if (qs[q_off + j] != ts[t_off + j]) flush(j), o.ch('*'), o.ch("acgtn"[ts[t_off + j]]), o.ch("acgtn"[qs[q_off + j]]);
				else ++run;
			}
			flush(len);
			q_off += len, t_off += len;
		} else if (op == 1) {
			o.ch('+');
			if (is_ds) { // how far the inserted bases could be shifted right / left (format.c:214-222)
				int z, y = q_off;
				for (z = 1; z <= len; ++z) if (y - z < 0 || qs[y + len - z] != qs[y - z]) break;
				const int lr = z - 1;
				for (z = 0; z < len; ++z) if (y + len + z >= q_len || qs[y + len + z] != qs[y + z]) break;
				put_ds_indel(o, len, qs + y, z, lr);
			} else put_bases(o, "acgtn", qs + q_off, len);
			q_off += len;
		} else if (op == 2) {
			o.ch('-');
			if (is_ds) {
				int z, x = t_off;
				for (z = 1; z <= len; ++z) if (x - z < 0 || ts[x + len - z] != ts[x - z]) break;
				const int lr = z - 1;
				for (z = 0; z < len; ++z) if (x + len + z >= t_len || ts[x + z] != ts[x + len + z]) break;
				put_ds_indel(o, len, ts + x, z, lr);
			} else put_bases(o, "acgtn", ts + t_off, len);
			t_off += len;
		} else { // intron: its first and last two bases
			o.ch('~'), o.ch("acgtn"[ts[t_off]]), o.ch("acgtn"[ts[t_off + 1]]), o.num(len), o.ch("acgtn"[ts[t_off + len - 2]]), o.ch("acgtn"[ts[t_off + len - 1]]);
			t_off += len;
		}
	}
}

void put_md(Text &o, const Reg1 &r, const Seqs &sq) // write_MD_core
{
	const uint8_t *qs = sq.q.data(), *ts = sq.t.data();
	o.str("\tMD:Z:");
	int q_off = 0, t_off = 0, l_md = 0;
	for (uint32_t i = 0; i < r.p->n_cigar; ++i) {
		const int op = r.p->cigar[i] & 0xf, len = r.p->cigar[i] >> 4;
		if (op == 0 || op == 7 || op == 8) {
			for (int j = 0; j < len; ++j) {
				if (qs[q_off + j] != ts[t_off + j]) o.num(l_md), o.ch("ACGTN"[ts[t_off + j]]), l_md = 0;
				else ++l_md;
			}
			q_off += len, t_off += len;
		} else if (op == 1) q_off += len;
		else if (op == 2) o.num(l_md), o.ch('^'), put_bases(o, "ACGTN", ts + t_off, len), l_md = 0, t_off += len;
		else if (op == 3) t_off += len;
	}
	if (l_md > 0) o.num(l_md);
}

void put_cs_or_md(Text &o, const FlatIndex &fi, const Bseq1 &t, const Reg1 &r, int64_t flag, Seqs &sq)
{
	if (!r.p) return;
	fetch_pair(fi, t, r, sq, (flag & F_QSTRAND) != 0);
	if (flag & F_OUT_MD) put_md(o, r, sq);
	else put_cs(o, r, sq, !(flag & F_OUT_CS_LONG), flag & F_OUT_DS);
}

void put_paf(Text &o, const FlatIndex &fi, const Bseq1 &t, const Reg1 *r, int64_t flag, int rep_len, int n_seg, int seg_idx, Seqs &sq) // mm_write_paf4
{
	o.str(t.name);
	if ((flag & F_FRAG_MODE) && n_seg >= 2 && seg_idx >= 0) o.ch('/'), o.num(seg_idx + 1);
	if (!r) {
		o.ch('\t'), o.num(t.l_seq), o.str("\t0\t0\t*\t*\t0\t0\t0\t0\t0\t0");
		if (rep_len >= 0) o.tag("rl:i:", rep_len);
		return;
	}
	o.ch('\t'), o.num(t.l_seq), o.ch('\t'), o.num(r->qs), o.ch('\t'), o.num(r->qe), o.ch('\t'), o.ch("+-"[r->rev]), o.ch('\t');
	if (!fi.names[r->rid].empty()) o.str(fi.names[r->rid].c_str()); else o.num(r->rid);
	o.ch('\t'), o.num(fi.seq_len[r->rid]), o.ch('\t');
	if ((flag & F_QSTRAND) && r->rev) o.num((int)fi.seq_len[r->rid] - r->re), o.ch('\t'), o.num((int)fi.seq_len[r->rid] - r->rs); // format.c:440-443
	else o.num(r->rs), o.ch('\t'), o.num(r->re);
	o.ch('\t'), o.num(r->mlen), o.ch('\t'), o.num(r->blen), o.ch('\t'), o.num(r->mapq);
	put_tags(o, *r);
	if (rep_len >= 0) o.tag("rl:i:", rep_len);
	if (r->p && (flag & F_OUT_CG)) {
		o.str("\tcg:Z:");
		o.cigar(r->p->cigar, r->p->n_cigar);
	}
	if (r->p && (flag & (F_OUT_CS | F_OUT_DS | F_OUT_MD))) put_cs_or_md(o, fi, t, *r, flag, sq);
	if ((flag & F_COPY_COMMENT) && t.comment) o.ch('\t'), o.str(t.comment);
}

#if defined(__x86_64__)
// Reverse complement, 16 bases per step, for blocks made of A/C/G/T/N in either case: the low nibbles of these five letters differ
// (1, 3, 7, 4, 14), so one byte shuffle maps a letter to its complement and another checks that the block holds nothing else.
static bool have_ssse3() { static const bool v = __builtin_cpu_supports("ssse3"); return v; }
__attribute__((target("ssse3"))) static int revcomp_blocks(const char *seq, int l, char *w)
{
	const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
	const __m128i self = _mm_setr_epi8(0, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 'N', 0);
	const __m128i other = _mm_setr_epi8(0, 'T', 0, 'G', 'A', 0, 0, 'C', 0, 0, 0, 0, 0, 0, 'N', 0);
	const __m128i low = _mm_set1_epi8(0x0f), upper = _mm_set1_epi8((char)0xdf), lower_bit = _mm_set1_epi8(0x20);
	int i = 0;
	for (; i + 16 <= l; i += 16) {
		const __m128i v = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(seq + l - 16 - i)), rev);
		const __m128i nib = _mm_and_si128(v, low), up = _mm_and_si128(v, upper);
		if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_shuffle_epi8(self, nib), up)) != 0xffff) break;
		_mm_storeu_si128((__m128i *)(w + i), _mm_or_si128(_mm_shuffle_epi8(other, nib), _mm_and_si128(v, lower_bit)));
	}
	return i;
}
#endif

void put_seq(Text &o, const char *seq, int l, bool rev, bool comp) // sam_write_sq (format.c:470-482)
{
	static const std::array<char, 128> kComp = [] { // seq_comp_table (bseq.c:11-28) for ASCII: IUPAC complements, case preserved
		std::array<char, 128> t;
		for (int c = 0; c < 128; ++c) t[c] = (char)c;
		const char *a = "ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", *b = "TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn";
		for (int i = 0; a[i]; ++i) t[(int)a[i]] = b[i];
		return t;
	}();
	if (!rev) { o.str(seq, l); return; }
	char *w = o.need((size_t)l);
	if (comp) {
		int i = 0;
#if defined(__x86_64__)
		if (have_ssse3()) i = revcomp_blocks(seq, l, w); // whole 16-base blocks of plain A/C/G/T/N (either case); stops at the first block holding anything else
#endif
		for (; i < l; ++i) { const int c = (unsigned char)seq[l - 1 - i]; w[i] = c < 128 ? kComp[c] : (char)c; }
	}
	else for (int i = 0; i < l; ++i) w[i] = seq[l - 1 - i];
	o.n += (size_t)l;
}

void put_sam_cigar(Text &o, int sam_flag, bool in_tag, int qlen, const Reg1 &r, int64_t flag) // write_sam_cigar (format.c:494-520)
{
	if (!r.p) { o.ch('*'); return; }
	const uint32_t clip0 = r.rev ? qlen - r.qe : r.qs, clip1 = r.rev ? r.qs : qlen - r.qe;
	const bool hard = ((sam_flag & 0x800) || ((sam_flag & 0x100) && (flag & F_SECONDARY_SEQ))) && !(flag & F_SOFTCLIP);
	if (in_tag) {
		const uint32_t op = hard ? 5 : 4;
		o.str("\tCG:B:I");
		if (clip0) o.ch(','), o.num(clip0 << 4 | op);
		for (uint32_t k = 0; k < r.p->n_cigar; ++k) o.ch(','), o.num(r.p->cigar[k]);
		if (clip1) o.ch(','), o.num(clip1 << 4 | op);
	} else {
		const char c = hard ? 'H' : 'S';
		if (clip0) o.num(clip0), o.ch(c);
		o.cigar(r.p->cigar, r.p->n_cigar);
		if (clip1) o.num(clip1), o.ch(c);
	}
}

// mm_write_sam3 for a single-segment read: reg_idx < 0 writes the unmapped record
const Reg1 *sam_primary(int n_regs, const Reg1 *regs) // get_sam_pri (format.c:484-492)
{
	for (int i = 0; i < n_regs; ++i) if (regs[i].sam_pri) return &regs[i];
	return nullptr;
}

// mm_write_sam3 (format.c:522-700) for fragments of one or two segments
void put_sam(Text &o, const FlatIndex &fi, const Bseq1 &t, int seg_idx, int reg_idx, int n_seg, const int *n_regss, void *const *regss, int64_t flag, int rep_len, Seqs &sq)
{
	const int n_regs = n_regss[seg_idx];
	const Reg1 *regs = (const Reg1 *)regss[seg_idx];
	const Reg1 *r = n_regs > 0 && reg_idx >= 0 && reg_idx < n_regs ? &regs[reg_idx] : nullptr;
	const Reg1 *r_next = nullptr, *r_prev = nullptr; // the mate's primary, if it is mapped
	if (n_seg > 1) {
		const int next_sid = (seg_idx + 1) % n_seg;
		r_prev = r_next = sam_primary(n_regss[next_sid], (const Reg1 *)regss[next_sid]);
	}
	if (n_seg > 1) { // the name without a /1 or /2 suffix (mm_qname_len, bseq.h:31-36)
		size_t l = strlen(t.name);
		if (l >= 3 && t.name[l - 1] >= '0' && t.name[l - 1] <= '9' && t.name[l - 2] == '/') l -= 2;
		for (size_t i = 0; i < l; ++i) o.ch(t.name[i]);
	} else o.str(t.name);
	int sf = n_seg > 1 ? 0x1 : 0x0;
	if (!r) sf |= 0x4;
	else {
		if (r->rev) sf |= 0x10;
		if (r->parent != r->id) sf |= 0x100;
		else if (!r->sam_pri) sf |= 0x800;
	}
	if (n_seg > 1) {
		if (r && r->proper_frag) sf |= 0x2;
		if (seg_idx == 0) sf |= 0x40;
		else if (seg_idx == n_seg - 1) sf |= 0x80;
		if (!r_next) sf |= 0x8;
		else if (r_next->rev) sf |= 0x20;
	}
	int this_rid = -1, this_pos = -1;
	o.ch('\t'), o.num(sf);
	bool cigar_in_tag = false;
	if (!r) {
		if (r_prev) { // an unmapped read is placed at its mate's position
			this_rid = r_prev->rid, this_pos = r_prev->rs;
			o.ch('\t'), o.str(fi.names[this_rid].c_str()), o.ch('\t'), o.num(this_pos + 1), o.str("\t0\t*");
		} else o.str("\t*\t0\t0\t*");
	} else {
		this_rid = r->rid, this_pos = r->rs;
		o.ch('\t'), o.str(fi.names[r->rid].c_str()), o.ch('\t'), o.num(r->rs + 1), o.ch('\t'), o.num(r->mapq), o.ch('\t');
		if ((flag & F_LONG_CIGAR) && r->p && r->p->n_cigar > 65535 - 2) {
			int n_cigar = (int)r->p->n_cigar;
			if (r->qs != 0) ++n_cigar;
			if (r->qe != t.l_seq) ++n_cigar;
			cigar_in_tag = n_cigar > 65535;
		}
		if (cigar_in_tag) {
			int slen;
			if ((sf & 0x900) == 0 || (flag & F_SOFTCLIP)) slen = t.l_seq;
			else if ((sf & 0x100) && !(flag & F_SECONDARY_SEQ)) slen = 0;
			else slen = r->qe - r->qs;
			o.num(slen), o.ch('S'), o.num(r->re - r->rs), o.ch('N');
		} else put_sam_cigar(o, sf, false, t.l_seq, *r, flag);
	}
	if (n_seg > 1) { // mate position and template length (format.c:592-613)
		int tlen = 0;
		if (this_rid >= 0 && r_next) {
			if (this_rid == r_next->rid) {
				if (r) {
					const int this_pos5 = r->rev ? r->re - 1 : this_pos, next_pos5 = r_next->rev ? r_next->re - 1 : r_next->rs;
					tlen = next_pos5 - this_pos5;
				}
				o.str("\t=\t");
			} else o.ch('\t'), o.str(fi.names[r_next->rid].c_str()), o.ch('\t');
			o.num(r_next->rs + 1), o.ch('\t');
		} else if (r_next) o.ch('\t'), o.str(fi.names[r_next->rid].c_str()), o.ch('\t'), o.num(r_next->rs + 1), o.ch('\t');
		else if (this_rid >= 0) o.str("\t=\t"), o.num(this_pos + 1), o.ch('\t');
		else o.str("\t*\t0\t");
		if (tlen > 0) ++tlen;
		else if (tlen < 0) --tlen;
		o.num(tlen), o.ch('\t');
	} else o.str("\t*\t0\t0\t"); // no mate
	if (!r) {
		put_seq(o, t.seq, t.l_seq, false, false), o.ch('\t');
		if (t.qual) put_seq(o, t.qual, t.l_seq, false, false); else o.ch('*');
	} else if ((sf & 0x900) == 0 || (flag & F_SOFTCLIP)) {
		put_seq(o, t.seq, t.l_seq, r->rev, r->rev), o.ch('\t');
		if (t.qual) put_seq(o, t.qual, t.l_seq, r->rev, false); else o.ch('*');
	} else if ((sf & 0x100) && !(flag & F_SECONDARY_SEQ)) o.str("*\t*");
	else {
		put_seq(o, t.seq + r->qs, r->qe - r->qs, r->rev, r->rev), o.ch('\t');
		if (t.qual) put_seq(o, t.qual + r->qs, r->qe - r->qs, r->rev, false); else o.ch('*');
	}
	if (r) {
		put_tags(o, *r);
		if (r->parent == r->id && r->p && n_regs > 1) { // supplementary alignments of the same read (format.c:638-664)
			int n_sa = 0;
			for (int i = 0; i < n_regs; ++i) if (i != reg_idx && regs[i].parent == regs[i].id && regs[i].p) ++n_sa;
			if (n_sa > 0) {
				o.str("\tSA:Z:");
				for (int i = 0; i < n_regs; ++i) {
					const Reg1 &q = regs[i];
					if (i == reg_idx || q.parent != q.id || !q.p) continue;
					int l_M, l_I = 0, l_D = 0;
					if (q.qe - q.qs < q.re - q.rs) l_M = q.qe - q.qs, l_D = (q.re - q.rs) - l_M;
					else l_M = q.re - q.rs, l_I = (q.qe - q.qs) - l_M;
					const int clip5 = q.rev ? t.l_seq - q.qe : q.qs, clip3 = q.rev ? q.qs : t.l_seq - q.qe;
					o.str(fi.names[q.rid].c_str()), o.ch(','), o.num(q.rs + 1), o.ch(','), o.ch("+-"[q.rev]), o.ch(',');
					if (clip5) o.num(clip5), o.ch('S');
					if (l_M) o.num(l_M), o.ch('M');
					if (l_I) o.num(l_I), o.ch('I');
					if (l_D) o.num(l_D), o.ch('D');
					if (clip3) o.num(clip3), o.ch('S');
					o.ch(','), o.num(q.mapq), o.ch(','), o.num(q.blen - q.mlen + (int)q.p->n_ambi), o.ch(';');
				}
			}
		}
		if (r->p && (flag & (F_OUT_CS | F_OUT_DS | F_OUT_MD))) put_cs_or_md(o, fi, t, *r, flag, sq);
		if (cigar_in_tag) put_sam_cigar(o, sf, true, t.l_seq, *r, flag);
	}
	if (rep_len >= 0) o.tag("rl:i:", rep_len);
	if ((flag & F_COPY_COMMENT) && t.comment) o.ch('\t'), o.str(t.comment);
}

} // namespace

std::string format_check(const MapOpt &opt)
{
	if (opt.flag & F_OUT_JUNC) return "--write-junc output is not implemented";
	if (opt.split_prefix) return "split-index output is not implemented";
	return "";
}

// the records of fragments [lo, hi), in order (map.c:585-623)
// mm2amd_format_fraction (diagnostics, include/mm2amd.h): "%.4f" as the output stage writes it
int format_fraction_for_test(double v, char *buf)
{
	Text t;
	put_fraction(t, v);
	memcpy(buf, t.data(), t.size());
	buf[t.size()] = 0;
	return (int)t.size();
}

static void format_range(const FlatIndex &fi, const MapOpt &opt, const int *seg_off, const int *n_seg, const Bseq1 *seq, const int *n_reg, void *const *reg,
                         const int *rep_len, long lo, long hi, Text &o)
{
	Seqs sq;
	const int64_t flag = opt.flag;
	hostprof::Scope hp(hostprof::FORMAT_RANGE);
	o.gaps = Text::GapCount(); // (the gap counts kept from a record's CIGAR text are only good while the batch's blocks are alive)
	for (long f = lo; f < hi; ++f) {
		const int seg_st = seg_off ? seg_off[f] : (int)f, ns = n_seg ? n_seg[f] : 1;
		for (int i = seg_st; i < seg_st + ns; ++i) {
			const Bseq1 &t = seq[i];
			const Reg1 *regs = (const Reg1 *)reg[i];
			const int rl = rep_len ? rep_len[i] : -1;
			if (n_reg[i] > 0) {
				for (int j = 0; j < n_reg[i]; ++j) {
					if ((flag & F_NO_PRINT_2ND) && regs[j].id != regs[j].parent) continue;
					if (flag & F_OUT_SAM) put_sam(o, fi, t, i - seg_st, j, ns, &n_reg[seg_st], &reg[seg_st], flag, rl, sq);
					else put_paf(o, fi, t, &regs[j], flag, rl, ns, i - seg_st, sq);
					o.ch('\n');
				}
			} else if ((flag & F_PAF_NO_HIT) || ((flag & F_OUT_SAM) && !(flag & F_SAM_HIT_ONLY))) {
				if (flag & F_OUT_SAM) put_sam(o, fi, t, i - seg_st, -1, ns, &n_reg[seg_st], &reg[seg_st], flag, rl, sq);
				else put_paf(o, fi, t, nullptr, flag, rl, ns, i - seg_st, sq);
				o.ch('\n');
			}
		}
	}
}

struct FormatScratch::Impl { std::vector<Text> parts; };
FormatScratch::FormatScratch() : impl(new Impl) {}
FormatScratch::~FormatScratch() { delete impl; free(buf); }

// the chunks' texts (64 fragments each, formatted by one pool thread each) and where each starts in the concatenation
static size_t format_parts(const FlatIndex &fi, const MapOpt &opt, int n_threads, long n, const int *seg_off, const int *n_seg, const Bseq1 *seq, const int *n_reg,
                           void *const *reg, const int *rep_len, std::vector<Text> &parts, std::vector<size_t> &off)
{
	const long chunk = 64, n_chunks = (n + chunk - 1) / chunk;
	if ((long)parts.size() < n_chunks) parts.resize(n_chunks);
	static const bool trace = getenv("MM2AMD_FMT_TRACE") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	std::atomic<long long> busy_ns{0};
	parallel_for_side(n_threads, n_chunks, [&](long c, int) {
		const auto c0 = std::chrono::steady_clock::now();
		const long lo = c * chunk, hi = std::min(n, lo + chunk);
		parts[c].clear(); // keeps its capacity: a reused scratch formats into memory it already owns
		format_range(fi, opt, seg_off, n_seg, seq, n_reg, reg, rep_len, lo, hi, parts[c]);
		if (trace) busy_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - c0).count();
	}, 1);
	if (trace) fprintf(stderr, "[mm2amd] format: %ld chunks, wall %.3f s, summed chunk time %.3f s on %d threads\n", n_chunks,
	                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), busy_ns.load() * 1e-9, n_threads);
	off.assign(n_chunks + 1, 0);
	for (long c = 0; c < n_chunks; ++c) off[c + 1] = off[c] + parts[c].size();
	return off[n_chunks];
}

char *format_batch(const FlatIndex &fi, const MapOpt &opt, int n_threads, long n_frag, const int *seg_off, const int *n_seg, const Bseq1 *seq, const int *n_reg,
                   void *const *reg, const int *rep_len, size_t *out_len)
{
	std::vector<Text> parts;
	std::vector<size_t> off;
	const size_t total = format_parts(fi, opt, n_threads, n_frag, seg_off, n_seg, seq, n_reg, reg, rep_len, parts, off);
	char *out = (char *)malloc(total + 1);
	if (!out) return nullptr;
	parallel_for_side(n_threads, (long)off.size() - 1, [&](long c, int) { memcpy(out + off[c], parts[c].data(), parts[c].size()); }, 8);
	out[total] = 0;
	*out_len = total;
	return out;
}

const char *format_batch_view(const FlatIndex &fi, const MapOpt &opt, int n_threads, long n_frag, const int *seg_off, const int *n_seg, const Bseq1 *seq, const int *n_reg,
                              void *const *reg, const int *rep_len, FormatScratch &fs, size_t *out_len)
{
	std::vector<size_t> off;
	const size_t total = format_parts(fi, opt, n_threads, n_frag, seg_off, n_seg, seq, n_reg, reg, rep_len, fs.impl->parts, off);
	if (fs.cap < total + 1) { // grow-only: in the steady state of a pipeline no page of this buffer is new
		free(fs.buf);
		fs.cap = (total + 1) + (total + 1) / 4;
		fs.buf = (char *)malloc(fs.cap);
		if (!fs.buf) { fs.cap = 0; return nullptr; }
	}
	parallel_for_side(n_threads, (long)off.size() - 1, [&](long c, int) { memcpy(fs.buf + off[c], fs.impl->parts[c].data(), fs.impl->parts[c].size()); }, 8);
	fs.buf[total] = 0;
	*out_len = total;
	return fs.buf;
}

} // namespace mm2amd

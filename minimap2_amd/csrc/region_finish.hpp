// The last step of a region's base-level alignment on the device: stitch the DP windows' CIGARs in order, then mm_update_extra
// (align.c:254-303) with mm_fix_cigar (:105-181) -- indels left-aligned, I/D clusters merged, a leading gap dropped, the matching /
// aligned column counts, ambiguous bases and the gap-compressed best score (dp_max).  On the host this is the largest per-read cost
// of the alignment stages (a walk over ~1400 CIGAR operations and 10 000 base pairs per 10 kb read); the device has the windows'
// CIGARs, the reads and the packed reference already.
#pragma once
#include <cstdint>

namespace mm2amd {

constexpr int kFinMaxOps = 7168; // CIGAR operations of a region the kernel can stage in LDS (two regions per 64 KB workgroup; a 10 kb ONT read has ~1400); longer ones are finished on the host

struct FinRegion {        // one region whose windows all came back in the current round
	uint64_t q_pos;       // first aligned query base in the device query pool (strand block of the read + qs1)
	uint64_t t_pos;       // first aligned reference base in the packed reference (sequence offset + rs1)
	uint32_t piece0, n_pieces; // its windows' CIGARs in the piece list, in alignment order
	uint32_t out_off;     // where the stitched CIGAR goes in the output pool (room for the sum of the pieces)
	int32_t q_len, t_len; // query / reference bases the CIGAR must cover
};
struct FinPiece { uint32_t off, n; }; // a window's CIGAR in the DP batch's pool (KswRes::cigar_off, n_cigar); n == kFinLiteral: ONE operation, and `off` is the CIGAR word itself (a short read's ungapped window)
constexpr uint32_t kFinLiteral = 0x80000000u;
struct FinResult { int32_t n_cigar, blen, mlen, n_ambi, dp_max, qshift, tshift, is_spliced; }; // n_cigar < 0: the CIGAR does not cover the windows (a bug: the host throws)
struct FinParams {
	const FinRegion *regions; int n_regions;
	const FinPiece *pieces;
	const uint32_t *cigar_pool;   // the DP batch's CIGARs
	uint32_t *out_pool;
	FinResult *results;
	const uint8_t *qpool;
	const uint32_t *S;            // packed reference, eight 4-bit codes per word
	int8_t mat[25];
	int8_t q, e;
	int log_gap;                  // gap cost q + e * log2(1 + len) (long reads) or q + e
	int cap_ops;                  // LDS slots per region: >= the longest stitched CIGAR of the launch (sum of its pieces), <= kFinMaxOps
};

void region_finish_launch(const FinParams &P, void *stream);

} // namespace mm2amd

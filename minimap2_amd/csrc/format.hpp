// Host output stage: SAM / PAF text of a mini-batch's hits, byte-identical to the reference's writers (format.cpp).
#pragma once
#include <string>
#include "abi_ref.hpp"
#include "flat_index.hpp"

namespace mm2amd {

// "" when the options can be formatted here, else what is missing
std::string format_check(const ref::MapOpt &opt);
// One malloc'd block ('\n'-terminated records of fragments 0..n_frag-1 in order, NUL after the last byte), or nullptr when out of
// memory.  Fragment f consists of reads seg_off[f] .. seg_off[f]+n_seg[f]-1 of seq[] (null arrays: one read per fragment).
char *format_batch(const FlatIndex &fi, const ref::MapOpt &opt, int n_threads, long n_frag, const int *seg_off, const int *n_seg, const ref::Bseq1 *seq, const int *n_reg,
                   void *const *reg, const int *rep_len, size_t *out_len);

// The same into memory that is reused from call to call (owned by `fs`; the returned pointer is valid until the next call with it)
struct FormatScratch {
	struct Impl;
	Impl *impl;
	char *buf = nullptr;
	size_t cap = 0;
	FormatScratch();
	~FormatScratch();
	FormatScratch(const FormatScratch &) = delete;
	FormatScratch &operator=(const FormatScratch &) = delete;
};
const char *format_batch_view(const FlatIndex &fi, const ref::MapOpt &opt, int n_threads, long n_frag, const int *seg_off, const int *n_seg, const ref::Bseq1 *seq, const int *n_reg,
                              void *const *reg, const int *rep_len, FormatScratch &fs, size_t *out_len);

int format_fraction_for_test(double v, char *buf); // "%.4f" as the output stage writes it (mm2amd_format_fraction)

} // namespace mm2amd

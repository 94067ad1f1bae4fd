// Host output stage: SAM / PAF text of a mini-batch's hits, byte-identical to the reference's writers (format.cpp).
#pragma once
#include <string>
#include "abi_ref.hpp"
#include "flat_index.hpp"

namespace mm2amd {

// "" when the options can be formatted here, else what is missing
std::string format_check(const ref::MapOpt &opt);
// One malloc'd block ('\n'-terminated records of reads 0..n-1 in order, NUL after the last byte), or nullptr when out of memory.
char *format_batch(const FlatIndex &fi, const ref::MapOpt &opt, int n_threads, long n, const ref::Bseq1 *seq, const int *n_reg, void *const *reg, const int *rep_len, size_t *out_len);

} // namespace mm2amd

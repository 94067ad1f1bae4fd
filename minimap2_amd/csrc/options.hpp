// Host-side mirror of the reference's option API for the per-read path (options.c).
#pragma once
#include <string>
#include "abi_ref.hpp"

namespace mm2amd {
void idxopt_init(ref::IdxOpt *io);
void mapopt_init(ref::MapOpt *mo);
int set_opt(const char *preset, ref::IdxOpt *io, ref::MapOpt *mo);           // 0, or -1 for an unknown preset
void mapopt_update(ref::MapOpt *mo, int32_t (*cal_max_occ)(const void *, float), const void *idx);
int check_opt(const ref::IdxOpt *io, const ref::MapOpt *mo, std::string *why);
}

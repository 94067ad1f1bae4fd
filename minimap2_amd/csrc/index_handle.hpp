// Opaque index object behind mm2amd_index_t (include/mm2amd.h): host-side sequence store + device-resident tables.
#pragma once
#include "flat_index.hpp"

namespace mm2amd {
struct IndexHandle;                                   // defined in capi_index.cpp (HIP product only)
const FlatIndex &index_flat(const IndexHandle *h);
void *index_device_tables(const IndexHandle *h);      // DeviceIndexTables*, passed through to make_backend()
int index_device(const IndexHandle *h);               // the HIP device those tables live on
}

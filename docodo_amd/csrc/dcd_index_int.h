// dcd_index_int.h -- the dcd_index handle shared by dcd_index.hip and dcd_query.hip.
#pragma once
#include <memory>

#include "dcd_build_int.h"
#include "dcd_prims.h"

struct dcd_build;

struct dcd_index {
  dcd::DeviceIndex ix;
  hipStream_t st = nullptr;
  // decoded postings (built lazily on the first query): segment-relative values per term
  dcd::DevBuf<u64> values;   // [n_postings], term t at [term_item_start[t], term_item_start[t+1])
  bool values_ready = false;
};

// dcd_index_int.h -- the dcd_index handle shared by dcd_index.hip and dcd_query.hip.
#pragma once
#include <memory>
#include <mutex>

#include "dcd_build_int.h"
#include "dcd_prims.h"

struct dcd_build;

struct dcd_index {
  dcd::DeviceIndex ix;
  hipStream_t st = nullptr;
  // Search is issued from many pool threads (server.cs:16-30) and serialised by lock(DoSearchLock) in the reference
  // (Search.cs:448); batches on ONE index handle are serialised here the same way, different handles run concurrently
  std::mutex mu;
  // decoded postings (built lazily on the first query): segment-relative values per term
  dcd::DevBuf<u64> values;   // [n_postings], term t at [term_item_start[t], term_item_start[t+1])
  bool values_ready = false;
  // cell directories of the longer posting lists (auxiliary, built with `values`): for term t with shift s,
  // dir[dir_off[t] + c] = index (inside the term) of its first posting >= c << s; one extra entry = posting count.
  // A galloping '*' finds the neighbourhood of a coordinate with two loads instead of a ~20-step binary search.
  dcd::DevBuf<u32> dir;
  std::vector<u64> h_dir_off;    // [n_terms + 1]
  std::vector<u8> h_dir_shift;   // [n_terms]; 255 = no directory
  // ".index" bytes of this index, generated on demand (dcd_index_image / dcd_index_save)
  dcd::DevBuf<u8> image;
  u64 image_bytes = 0;
};

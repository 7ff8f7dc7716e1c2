// dcd_index_int.h -- the dcd_index handle shared by dcd_index.hip and dcd_query.hip.
#pragma once
#include <memory>
#include <mutex>

#include "dcd_build_int.h"
#include "dcd_prims.h"

struct dcd_build;

// Index.Load with InMemory == false (Index.cs:133-136, 346-348): the posting lists stay in the ".index" file, one stub per key
// = {n_units, file position of the payload}; this[key] reads the list when it is asked for (LoadSequence, Index.cs:198-216).
struct LazyPostings {
  int fd = -1;
  std::vector<u64> file_off;     // [n_terms] byte offset of the term's first unit in ".index"
  std::vector<u32> n_units;      // [n_terms]
  u64 resident_budget = 0;       // device bytes the fetched lists may occupy (the subset of the last batches is kept)
  // the resident subset: a small in-memory index over `terms` (ascending global ordinals), built by the query path
  std::unique_ptr<struct dcd_index> sub;
  std::vector<u32> terms;
  u64 fetched_bytes = 0, fetches = 0;   // statistics: bytes read from the file, lists read
  ~LazyPostings();
};

struct dcd_index {
  dcd::DeviceIndex ix;
  hipStream_t st = nullptr;
  std::unique_ptr<LazyPostings> lazy;   // non-null: on-disk postings (dcd_index_open_lazy)
  // Search is issued from many pool threads (server.cs:16-30) and serialised by lock(DoSearchLock) in the reference
  // (Search.cs:448); batches on ONE index handle are serialised here the same way, different handles run concurrently
  std::mutex mu;
  // decoded postings (built lazily on the first query): segment-relative values per term
  dcd::DevBuf<u64> values;   // [n_postings], term t at [term_item_start[t], term_item_start[t+1]) -- only when not packed
  // ... packed (DOCODO_QUERY_PACKED=1 and MaxCoord < 2^40): two planes, 5 bytes per posting instead of 8 (dcd_query.hip, IdxVals)
  dcd::DevBuf<u32> val_lo;   // low 32 bits
  dcd::DevBuf<u8> val_hi;    // bits 32..39
  bool values_packed = false;
  bool values_ready = false;
  // cell directories of the longer posting lists (auxiliary, built with `values`): for term t with shift s,
  // dir[dir_off[t] + c] = index (inside the term) of its first posting >= c << s; one extra entry = posting count.
  // A galloping '*' finds the neighbourhood of a coordinate with two loads instead of a ~20-step binary search.
  dcd::DevBuf<u32> dir;
  std::vector<u64> h_dir_off;    // [n_terms + 1]
  std::vector<u8> h_dir_shift;   // [n_terms]; 255 = no directory
  // work pool + scratch arena of the query batches (dcd_query.hip run_batch): grown on demand, kept until the handle is freed
  dcd::DevBuf<u64> work_pool;       // (pool first, scratch behind it)
  u64 work_pool_elems = 0;
  dcd::DevBuf<u8> res_arena;        // result arrays of the chunk in flight (pages, positions, ranks, documents): grown on demand
  // ".index" bytes of this index, generated on demand (dcd_index_image / dcd_index_save)
  dcd::DevBuf<u8> image;
  u64 image_bytes = 0;
};

namespace dcd {
// device layout of an index from host-side units + directory (h_term_unit_start, h_term_chars / h_term_str_off): dcd_index.hip
int32_t index_from_host_units(DeviceIndex& ix, const std::vector<u16>& units, int device);
int32_t lazy_read_units(const LazyPostings& lz, u32 term, u16* dst);
}  // namespace dcd

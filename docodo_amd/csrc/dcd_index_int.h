// dcd_index_int.h -- the dcd_index handle shared by dcd_index.hip and dcd_query.hip.
#pragma once
#include <memory>
#include <mutex>
#include <vector>

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

// Table of the common sub-expressions of a sub-batch (dcd_query.hip, plan_chunk): operator (operand keys, R values, opcode) -> number of
// its result.  It lives in the handle and is emptied by advancing a generation number: clearing 2 x the operators of a 100 k-query
// batch (20 MB) took 2 ms of every call.
struct CseTable {
  struct Ent { uint64_t ka, kb; int32_t Ra, Rb; uint32_t op, ref, gen, pad; };
  std::vector<Ent> t;
  uint64_t mask = 0;
  uint32_t gen = 0;
  void init(uint64_t n_ops) {
    uint64_t c = 64;
    while (c < 2 * n_ops + 16) c <<= 1;
    if (c > t.size() || gen == 0xFFFFFFFFu) { t.assign(c, Ent{0, 0, 0, 0, 0, 0xFFFFFFFFu, 0, 0}); gen = 0; }
    mask = t.size() - 1;
    gen++;                                                     // (entries of earlier generations are empty)
  }
  static uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
  // the entry of the key; ref == 0xFFFFFFFF: new (the caller sets ref)
  Ent& slot(uint64_t ka, uint64_t kb, int32_t Ra, int32_t Rb, uint32_t op) {
    const uint64_t h = mix(mix(ka) + 0x9E3779B97F4A7C15ULL * mix(kb ^ 0x632BE59BD9B4E019ULL) + ((((uint64_t)(uint32_t)Ra) << 32) | (uint32_t)Rb) * 0xD1B54A32D192ED03ULL + op);
    for (uint64_t i = h & mask;; i = (i + 1) & mask) {
      Ent& e = t[i];
      if (e.gen != gen) { e.ka = ka; e.kb = kb; e.Ra = Ra; e.Rb = Rb; e.op = op; e.ref = 0xFFFFFFFFu; e.gen = gen; return e; }
      if (e.ka == ka && e.kb == kb && e.Ra == Ra && e.Rb == Rb && e.op == op) return e;
    }
  }
};

struct dcd_index {
  dcd::DeviceIndex ix;
  CseTable cse;                         // (used under `mu`, by one sub-batch at a time)
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

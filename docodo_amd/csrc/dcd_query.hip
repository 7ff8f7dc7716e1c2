// dcd_query.hip -- batched query evaluation on the device.
//
// Replaces, per query: this[key] + Count() (Index.cs:255-263, Search.cs:307), the IndexSequence
// operators * / + (IndexSequence.cs:218-322), PrepareSearchResult (Search.cs:365-420) and the page /
// document ranks (Search.cs:99-111, 552-557, 599).
//
// Shape of the computation
//   * the posting lists are decoded once per index into absolute coordinates (HBM-resident u64,
//     built lazily on the first query: an auxiliary structure, the .index image stays canonical);
//   * a query is a postfix program; programs advance in rounds, one binary operator per query and
//     round, all operators of a round (ordered by their longer index term, so that queries probing
//     the same list run together) in ONE launch of k_pair_parts: a work item is one (operator,
//     partition of 8192 union elements) or (operator, 1024 elements of the short operand when it
//     gallops), a fixed grid of wavefronts takes items from a device counter, see that kernel;
//   * a work item streams both operands through LDS rings in 256-element tiles: merge path -> union
//     with source tags -> gap chains -> "contains both operands" (and the ordered-mode prefix rule)
//     -> compaction.  A chain that is still open at a tile border is written optimistically and
//     rolled back if it closes without the other operand (exact restatement of the group logic of
//     IndexSequence.cs:218-284); skewed sparse pairs gallop from the
//     short operand (GAL_RATIO = 2 measured best on MI355X): cell directory -> position in the long
//     list -> nearest neighbours; the few elements that have one within absR are compacted and their
//     chains resolved in registers with full lanes, the rest of them walks (part_gallop);
//   * k_result<false/true> walk the final sequence of each query (one workgroup per query):
//     coordinate -> page by binary search over the page END table (IndexPageList.GetPage,
//     Build.cs:74-97), page-relative positions, the 30000-page / 500-document caps, ranks.
//
// Operand streams must be strictly ascending (true for every posting list and every operator
// result); equal values in both operands are merged as the reference does.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <functional>
#include <memory>
#include <mutex>
#include <chrono>
#include <thread>

#include "dcd_codec.h"
#include "dcd_index_int.h"
#include "dcd_scan.h"

namespace dcd {

constexpr u32 NONE32 = 0xFFFFFFFFu;
// part_src[item] = (operator of the item within the pass) << 36 | first scratch slot of its output: k_pair_gather gets the operator
// without a search over the items' prefix (round 4: the 14 dependent loads of that search were most of its 0.12 ms per pass)
constexpr int PSRC_OP_SHIFT = 36;
constexpr u64 PSRC_SLOT_MASK = (1ull << PSRC_OP_SHIFT) - 1ull;
constexpr u32 PSRC_MAX_OPS = 1u << 28;

struct Seg {       // a sequence of absolute coordinates somewhere on the device
  u64 off;         // kind 0: element offset into the index values (kind 1: unused -- the pool offset of a result is assigned on the device)
  u64 cnt_imm;     // count when cnt_ref == NONE32
  u32 kind;        // 0 = posting list of an index term, 1 = result of an earlier operator (work pool)
  u32 cnt_ref;     // kind 1: number of the operator result -- index into the device arrays `counts` and `res_off`
  u64 dir_off;     // cell directory of an index term (see dcd_index), valid when dir_shift != NONE32
  u32 dir_shift;
  u32 pad;
};
struct PairOp {
  Seg a, b;
  int32_t Ra, Rb;
  u32 op;          // DCD_OP_AND / DCD_OP_OR
  u32 out_cnt_ref; // number of this operator's result: counts[ref] values at pool + res_off[ref], both written on the device
  u64 cap;         // host-side bound of the result length = cap(a) + cap(b)
  u64 pad2;
};
// The work pool as the kernels see it.  Round 4: results are placed by the DEVICE, back to back in the order the slices complete
// (res_off[ref] = pool elements used so far + exclusive scan of the slice's result counts), so the pool holds what the operators
// really produced instead of cap(a) + cap(b) per operator -- a phrase over two head terms has a capacity of 3.7e8 elements and a
// result of a few 1e7 -- and a sub-batch is bounded by its results, not by its worst case.
struct Pool {
  const u64* base;      // pool elements
  const u64* res_off;   // [n results] offset of result `ref`
  const u64* counts;    // [n results] length of result `ref`
  const u32* overflow;  // != 0: the pool ran full -- nothing behind that point may be read (every kernel leaves at once)
};
struct FinalSeq {
  Seg s;
};

// Resident coordinates of the index terms: plain u64 (default), or PACKED into two planes -- the low 32 bits and bits 32..39 of
// every coordinate, 5 bytes per posting, for any index whose MaxCoord is below 2^40 (DOCODO_QUERY_PACKED=1; see ensure_values
// for the measurement that keeps it optional).  Operator results (the work pool) and external operands are always plain u64.
struct IdxVals {
  const u64* v64;    // plain (external operands, MaxCoord >= 2^40), or nullptr
  const u32* lo;     // packed planes (v64 == nullptr)
  const u8* hi;
};
// A coordinate array seen through either representation; which one is wave-uniform (it depends on the operand only).
struct CV {
  const u64* v;
  const u32* lo;
  const u8* hi;
  __device__ __forceinline__ u64 operator[](u64 i) const { return v ? v[i] : (((u64)hi[i] << 32) | lo[i]); }
  __device__ __forceinline__ CV operator+(u64 d) const { CV r; r.v = v ? v + d : nullptr; r.lo = lo ? lo + d : nullptr; r.hi = hi ? hi + d : nullptr; return r; }
};
__device__ __forceinline__ CV seg_cv(const Seg& s, const IdxVals& iv, const Pool& pl) {
  CV r;
  if (s.kind != 0) { r.v = pl.base + pl.res_off[s.cnt_ref]; r.lo = nullptr; r.hi = nullptr; }
  else if (iv.v64) { r.v = iv.v64 + s.off; r.lo = nullptr; r.hi = nullptr; }
  else { r.v = nullptr; r.lo = iv.lo + s.off; r.hi = iv.hi + s.off; }
  return r;
}
__device__ __forceinline__ u64 seg_cnt(const Seg& s, const Pool& pl) { return s.kind == 0 ? s.cnt_imm : pl.counts[s.cnt_ref]; }

// ------------------------------------------------------------------------------------------------
// whole-index decode: values[i] = absolute coordinate of posting i (term-relative prefix sum)
// ------------------------------------------------------------------------------------------------
__global__ void k_term_relative(const u64* __restrict__ excl, const u64* __restrict__ deltas, u64 n, const u64* __restrict__ item_start,
                                u64 n_terms, u64* __restrict__ values, u32* __restrict__ lo, u8* __restrict__ hi) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 a = 0, b = n_terms;  // last t with item_start[t] <= i
  while (a + 1 < b) { u64 m = (a + b) >> 1; if (item_start[m] <= i) a = m; else b = m; }
  const u64 v = excl[i] + deltas[i] - excl[item_start[a]];
  if (values) values[i] = v;
  else { lo[i] = (u32)v; hi[i] = (u8)(v >> 32); }
}

// cell directory: posting i of a term writes the directory entries of the cells between its predecessor's cell
// (exclusive) and its own (inclusive); the last posting also fills the cells behind it with the posting count
__global__ void k_build_dir(CV values, u64 n, const u64* __restrict__ item_start, u64 n_terms,
                            const u64* __restrict__ dir_off, const u8* __restrict__ dir_shift, u32* __restrict__ dir) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 a = 0, b = n_terms;  // last t with item_start[t] <= i
  while (a + 1 < b) { u64 m = (a + b) >> 1; if (item_start[m] <= i) a = m; else b = m; }
  const u32 sh = dir_shift[a];
  if (sh == 255u) return;
  const u64 li = i - item_start[a], cnt = item_start[a + 1] - item_start[a];
  const u64 n_cells = dir_off[a + 1] - dir_off[a] - 1;
  u32* d = dir + dir_off[a];
  const u64 c = min(values[i] >> sh, n_cells - 1);          // (a coordinate beyond MaxCoord lands in the last cell)
  u64 c0 = li ? min(values[i - 1] >> sh, n_cells - 1) + 1 : 0;
  for (u64 x = c0; x <= c; x++) d[x] = (u32)li;
  if (li + 1 == cnt) for (u64 x = c + 1; x <= n_cells; x++) d[x] = (u32)cnt;
}

// ------------------------------------------------------------------------------------------------
// Round 4: adoption in TWO passes over the units.  The resident coordinates and the cell directories used to be made by five
// kernels and two scans over 8-byte arrays -- terminator count, per-value group gather, a prefix sum over all values, a binary
// search over the term table per posting (twice) -- 60 ms for the 10 M-doc index, 1.6 x the build it follows.  Now:
//   k_adopt_reduce   per tile of 2048 units: values that end in the tile, sum of their deltas                  (reads 2 U)
//   two scans over the tile totals; k_adopt_terms: running delta sum in front of every term's first value      (small)
//   k_adopt_apply    per tile: decode again, block scan, term of every value from a slice of the term table staged in LDS,
//                    coordinate = running sum - the term's start sum, directory cells from the coordinate and its delta
//                    (the previous posting of the term is coordinate - delta)                       (reads 2 U, writes 8 N + cells)
// ------------------------------------------------------------------------------------------------
constexpr u32 AD_T = 256, AD_ITEMS = 8, AD_TILE = AD_T * AD_ITEMS;
// delta of the value whose LAST unit is u[i] (low group first; C# shifts a ulong by (15 k) & 63, IndexSequence.cs:126)
__device__ __forceinline__ u64 ad_delta_at(const u16* __restrict__ u, u64 i) {
  u64 s = i;
  while (s > 0 && (u[s - 1] & 0x8000u)) s--;
  u64 d = 0;
  u32 sh = 0;
  for (u64 k = s; k <= i; k++) { d += ((u64)(u[k] & 0x7FFFu)) << (sh & 63u); sh += 15; }
  return d;
}
// the lane's eight units: number of values that end among them, sum of their deltas (dl[j] = delta of a value ending at unit j)
__device__ __forceinline__ void ad_lane(const u16* __restrict__ u, u64 n, u64 base, u16 (&v)[AD_ITEMS], u64 (&dl)[AD_ITEMS], u32& cnt, u64& sum) {
  if (base + AD_ITEMS <= n) {
    const uint4 q = *reinterpret_cast<const uint4*>(u + base);                 // (tiles start at multiples of 2048 units: 16-byte aligned)
    v[0] = (u16)q.x; v[1] = (u16)(q.x >> 16); v[2] = (u16)q.y; v[3] = (u16)(q.y >> 16);
    v[4] = (u16)q.z; v[5] = (u16)(q.z >> 16); v[6] = (u16)q.w; v[7] = (u16)(q.w >> 16);
  } else {
#pragma unroll
    for (int j = 0; j < (int)AD_ITEMS; j++) v[j] = base + j < n ? u[base + j] : (u16)0x8000;
  }
  cnt = 0; sum = 0;
  // a value of one unit (every delta below 2^15: the dense lists) needs no look-back beyond the lane's own registers
  u64 d = 0;
  u32 sh = 0;
  bool whole = !(base > 0 && (u[base - 1] & 0x8000u));                          // the lane's first unit starts a value
#pragma unroll
  for (int j = 0; j < (int)AD_ITEMS; j++) {
    dl[j] = 0;
    const bool in = base + j < n;
    if (!in) continue;
    d += ((u64)(v[j] & 0x7FFFu)) << (sh & 63u);
    sh += 15;
    if (!(v[j] & 0x8000u)) {
      const u64 dv = whole ? d : ad_delta_at(u, base + j);                     // (a value that began before the lane: walk back)
      dl[j] = dv; sum += dv; cnt++;
      d = 0; sh = 0; whole = true;
    }
  }
}
__global__ __launch_bounds__(AD_T) void k_adopt_reduce(const u16* __restrict__ u, u64 n, u64* __restrict__ tile_cnt, u64* __restrict__ tile_sum) {
  __shared__ u64 lds[2 * (AD_T / 64)];
  u16 v[AD_ITEMS];
  u64 dl[AD_ITEMS];
  u32 cnt; u64 sum;
  ad_lane(u, n, (u64)blockIdx.x * AD_TILE + (u64)threadIdx.x * AD_ITEMS, v, dl, cnt, sum);
  const u64 c = wave_sum((u64)cnt), sm = wave_sum(sum);
  if (lane_id() == 0) { lds[threadIdx.x >> 6] = c; lds[AD_T / 64 + (threadIdx.x >> 6)] = sm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 a = 0, b = 0;
    for (u32 w = 0; w < AD_T / 64; w++) { a += lds[w]; b += lds[AD_T / 64 + w]; }
    tile_cnt[blockIdx.x] = a; tile_sum[blockIdx.x] = b;
  }
}
// g0[t] = sum of the deltas of all values in front of term t's first value (running over the whole unit stream)
__global__ void k_adopt_terms(const u16* __restrict__ u, u64 n, const u64* __restrict__ unit_start, u64 n_terms, const u64* __restrict__ tile_sum_excl,
                              u64* __restrict__ g0) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const u64 us = unit_start[t];
  if (us >= n) { g0[t] = 0; return; }                                          // (no unit: no value will ask)
  const u64 b = us / AD_TILE, t0 = b * AD_TILE;
  u64 acc = tile_sum_excl[b];
  // values that END in [t0, us): the first of them may have begun in front of the tile
  u64 s = t0;
  while (s > 0 && (u[s - 1] & 0x8000u)) s--;
  u64 d = 0;
  u32 sh = 0;
  for (u64 k = s; k < us; k++) {
    const u16 x = u[k];
    d += ((u64)(x & 0x7FFFu)) << (sh & 63u);
    sh += 15;
    if (!(x & 0x8000u)) { acc += d; d = 0; sh = 0; }
  }
  g0[t] = acc;
}
__global__ __launch_bounds__(AD_T) void k_adopt_apply(const u16* __restrict__ u, u64 n, const u64* __restrict__ tile_cnt_excl, const u64* __restrict__ tile_sum_excl,
                                                     const u64* __restrict__ item_start, u64 n_terms, const u64* __restrict__ g0,
                                                     const u64* __restrict__ dir_off, const u8* __restrict__ dir_shift, u64* __restrict__ values,
                                                     u32* __restrict__ dir) {
  __shared__ u32 s_cnt[AD_T / 64 + 1];
  __shared__ u64 s_sum[AD_T / 64 + 1];
  __shared__ u32 s_rel[AD_TILE + 2];                     // first value (relative to the tile's first) of the terms the tile touches
  __shared__ u64 s_t[2];
  u16 v[AD_ITEMS];
  u64 dl[AD_ITEMS];
  u32 cnt; u64 sum;
  ad_lane(u, n, (u64)blockIdx.x * AD_TILE + (u64)threadIdx.x * AD_ITEMS, v, dl, cnt, sum);
  u32 tot_c; u64 tot_s;
  const u32 ex_c = block_scan_excl<u32, (int)(AD_T / 64)>(cnt, s_cnt, &tot_c);
  const u64 ex_s = block_scan_excl<u64, (int)(AD_T / 64)>(sum, s_sum, &tot_s);
  if (!tot_c) return;
  const u64 v0 = tile_cnt_excl[blockIdx.x];              // index of the tile's first value
  // terms of the tile's first and last value: last t with item_start[t] <= value (empty terms share their successor's start)
  if (threadIdx.x < 2) {
    const u64 x = threadIdx.x ? v0 + tot_c - 1 : v0;
    u64 a = 0, b = n_terms;
    while (a + 1 < b) { const u64 m = (a + b) >> 1; if (item_start[m] <= x) a = m; else b = m; }
    s_t[threadIdx.x] = a;
  }
  __syncthreads();
  const u64 t_first = s_t[0], t_last = s_t[1];
  const bool staged = t_last - t_first <= AD_TILE;
  if (staged) {
    for (u64 k = threadIdx.x; k <= t_last - t_first + 1; k += AD_T) {
      const u64 st = t_first + k <= n_terms ? item_start[t_first + k] : ~0ull;
      s_rel[k] = st <= v0 ? 0u : (st - v0 > (u64)AD_TILE ? AD_TILE + 1u : (u32)(st - v0));
    }
  }
  __syncthreads();
  if (!cnt) return;
  // term of the lane's first value
  u64 t;
  if (staged) {
    u32 a = 0, b = (u32)(t_last - t_first) + 1u;        // last k with s_rel[k] <= ex_c
    while (a + 1 < b) { const u32 m = (a + b) >> 1; if (s_rel[m] <= ex_c) a = m; else b = m; }
    t = t_first + a;
  } else {
    u64 a = t_first, b = t_last + 1;
    while (a + 1 < b) { const u64 m = (a + b) >> 1; if (item_start[m] <= v0 + ex_c) a = m; else b = m; }
    t = a;
  }
  u64 vi = v0 + ex_c, G = tile_sum_excl[blockIdx.x] + ex_s;
  u64 t_end = item_start[t + 1], t_beg = item_start[t], tg0 = g0[t];
  u32 sh = dir_shift[t];
  u64 doff = dir_off[t], n_cells = sh != 255u ? dir_off[t + 1] - doff - 1 : 0;
#pragma unroll
  for (int j = 0; j < (int)AD_ITEMS; j++) {
    if ((v[j] & 0x8000u) || (u64)blockIdx.x * AD_TILE + (u64)threadIdx.x * AD_ITEMS + j >= n) continue;
    while (vi >= t_end) {                                // (skips empty terms)
      t++;
      t_beg = t_end; t_end = item_start[t + 1]; tg0 = g0[t];
      sh = dir_shift[t]; doff = dir_off[t]; n_cells = sh != 255u ? dir_off[t + 1] - doff - 1 : 0;
    }
    G += dl[j];
    const u64 c = G - tg0;
    values[vi] = c;
    if (sh != 255u) {
      // posting li of the term writes the directory entries of the cells between its predecessor's cell (exclusive) and its own
      // (inclusive); the last posting also fills the cells behind it with the posting count
      const u64 li = vi - t_beg, cntt = t_end - t_beg;
      u32* dd = dir + doff;
      const u64 cell = min(c >> sh, n_cells - 1);
      const u64 c0 = li ? min((c - dl[j]) >> sh, n_cells - 1) + 1 : 0;
      for (u64 x = c0; x <= cell; x++) dd[x] = (u32)li;
      if (li + 1 == cntt) for (u64 x = cell + 1; x <= n_cells; x++) dd[x] = (u32)cntt;
    }
    vi++;
  }
}

__global__ void k_unpack(const u32* __restrict__ lo, const u8* __restrict__ hi, u64 n, u64* __restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = ((u64)hi[i] << 32) | lo[i];
}

// ------------------------------------------------------------------------------------------------
// operator * / + on operand pairs
// ------------------------------------------------------------------------------------------------
// ---- sparse operands: galloping ----------------------------------------------------------------------
// When one operand is much shorter than the other and the long one is sparse relative to absR, almost
// every element of the long list lies in a chain of its own that cannot contain both operands.  Instead of
// merging, every element s of the SHORT list finds its neighbourhood in the long list by binary search and
// walks its gap chain outwards (the chain is the same set the merge would build: maximal runs of the union
// with consecutive gaps <= absR); the first short-list element of a chain ("leader") emits the chain if it
// satisfies the both-operands / ordered-prefix rule.  Exactly the same output as the streaming merge.
#ifndef DCD_GAL_RATIO
#define DCD_GAL_RATIO 3
#endif
// gallop when RATIO * |short| <= |long|.  Measured with the mask form of k_pair_stream (round 4, 100 k queries on the 10 M-document index,
// configs[3] / configs[4] device ms): ratio 2: 85.2 / 2410, 3: 85.4 / 2370, 4: 86.9 / 2356, 6: 89.8 / 2353, 8: 91.6 / 2356, 16: 97.7 / 2414
constexpr u32 GAL_RATIO = DCD_GAL_RATIO;

__device__ __forceinline__ u64 lower_bound_u64(const CV& a, u64 n, u64 v) {
  u64 lo = 0, hi = n;
  while (lo < hi) { u64 m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
  return lo;
}

// merged, de-duplicated walk over S[si..se] and L[li..le); calls f(value) in ascending order
template <typename F>
__device__ __forceinline__ u32 gal_merge(const CV& Sv, u64 si, u64 se_incl, const CV& Lv, u64 li, u64 le, F f) {
  u32 n = 0;
  u64 i = si, j = li;
  while (i <= se_incl || j < le) {
    bool hs = i <= se_incl, hl = j < le;
    u64 v;
    if (hs && hl) {
      u64 a = Sv[i], b = Lv[j];
      if (a == b) { v = a; i++; j++; }
      else if (a < b) { v = a; i++; }
      else { v = b; j++; }
    } else if (hs) { v = Sv[i++]; }
    else { v = Lv[j++]; }
    f(v);
    n++;
  }
  return n;
}

// ------------------------------------------------------------------------------------------------
// partitioned operators: one WAVEFRONT per (operator, partition)
// ------------------------------------------------------------------------------------------------
// The union of both operands is cut at fixed diagonals (every PART union elements, found with a 64-ary
// merge-path search over the operands in HBM).  Gap chains are independent of each other, so the work
// item of partition p evaluates exactly the chains that START inside [p*PART, (p+1)*PART): it skips the
// elements before its first chain break and FOLLOWS its last chain beyond the partition end until the
// chain closes.  No state crosses work items.  Emitted coordinates are written compactly into a scratch
// arena at the union index of the first owned element (owned ranges are disjoint and an item never emits
// more than it owns, so the writes cannot collide), then one scan over the per-item counts and a gather
// pass produce the contiguous operator result.  Every wave runs the same tile logic as one operand pair
// did before (merge path, chain ids, "contains both operands", ordered prefix rule, optimistic write +
// roll-back at tile borders), on 256-element tiles with wave-level scans; '+' uses the same ownership
// rule with every element its own chain.  Skewed sparse pairs gallop from the short operand, the short
// list spread evenly over the operator's work items.
constexpr u32 PW_T = 64;
#ifndef DCD_PW_ITEMS
#define DCD_PW_ITEMS 4
#endif
constexpr u32 PW_ITEMS = DCD_PW_ITEMS;
static_assert(PW_ITEMS == 4, "the in-lane stages of the bitonic tile merge are written for 4 elements per lane");
constexpr u32 PW_TILE = PW_T * PW_ITEMS;
constexpr u64 PART_DEFAULT = 16384;  // union elements per streaming work item (round 2: 2048 / 4096 / 8192 / 16384 -> 51.5 / 43.6 / 41.7 / 41.9 ms for 20 k AND3 queries
                                     // on the 10 M-doc index; end of round 4, 100 k queries / 100 k phrases: 4096 / 8192 / 16384 / 32768 -> 83.2 / 78.2 / 76.9 / 79.8 ms and 2.49 / 2.33 / 2.30 / 2.27 s)

// orders a wave's LDS writes before its later LDS reads of other lanes' data: the work items are single
// wavefronts, LDS executes a wave's accesses in order, so no s_barrier / s_waitcnt is needed (and the
// outstanding HBM loads and stores are not waited for, unlike __syncthreads())
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ u32 wave_min(u32 v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { u32 t = __shfl_xor(v, d, DCD_WAVE); v = t < v ? t : v; }
  return v;
}
// wave-uniform results as SCALARS: the compiler treats shuffle / DPP results as divergent, which would drag the
// whole per-work-item state (cursors, previous element, chain flags) into vector registers
__device__ __forceinline__ u32 rfl(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 rfl64(u64 v) { return ((u64)rfl((u32)(v >> 32)) << 32) | rfl((u32)v); }
__device__ __forceinline__ u32 rdl(u32 v, u32 uniform_lane) { return (u32)__builtin_amdgcn_readlane((int)v, (int)uniform_lane); }
__device__ __forceinline__ u64 rdl64(u64 v, u32 uniform_lane) { return ((u64)rdl((u32)(v >> 32), uniform_lane) << 32) | rdl((u32)v, uniform_lane); }
__device__ __forceinline__ u64 shfl_xor64(u64 v, int mask) {
  u32 lo = __shfl_xor((u32)v, mask, DCD_WAVE), hi = __shfl_xor((u32)(v >> 32), mask, DCD_WAVE);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_up64(u64 v, u32 d) {
  u32 lo = __shfl_up((u32)v, d, DCD_WAVE), hi = __shfl_up((u32)(v >> 32), d, DCD_WAVE);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl64(u64 v, u32 src_lane) {
  u32 lo = __shfl((u32)v, (int)src_lane, DCD_WAVE), hi = __shfl((u32)(v >> 32), (int)src_lane, DCD_WAVE);
  return ((u64)hi << 32) | lo;
}

// number of A elements among the first d elements of the union of A and B (ties: A first); all 64 lanes
// probe one candidate each, so the search needs log64 rounds of dependent HBM accesses
__device__ __forceinline__ u64 wave_split(const CV& A, u64 na, const CV& B, u64 nb, u64 d) {
  u64 lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  const u32 lane = lane_id();
  while (lo < hi) {
    const u64 step = (hi - lo + 63) >> 6;
    const u64 p = lo + (u64)lane * step + (step - 1);
    bool pr = false;
    if (p < hi) pr = A[p] <= B[d - 1 - p];
    const u32 c = popc64(ballot64(pr));          // the predicate is monotone: the true lanes form a prefix
    const u64 nlo = lo + (u64)c * step;
    const u64 nhi = nlo + step - 1;
    lo = nlo;
    hi = nhi < hi ? nhi : hi;
  }
  return lo;
}

#ifdef DCD_PROF
// debug build only: wave-residency cycles per phase of k_pair_parts (s_memtime deltas, summed over all waves)
__device__ unsigned long long g_prof[24];   // [0, 16): cycles per phase; [16, 24): event counts
#define PROF_T() unsigned long long _pt = __builtin_readcyclecounter()
// (accumulated per wavefront in LDS and added to g_prof once, at the end of the work item: one global atomic per probe made the
// profiled kernel 17x slower and charged the atomics' latency to whichever phase waited for memory next)
__shared__ unsigned long long s_prof[24];
#define PROF(i) do { unsigned long long _n = __builtin_readcyclecounter(); if (threadIdx.x == 0) s_prof[i] += _n - _pt; _pt = __builtin_readcyclecounter(); } while (0)
#define PROF_BEGIN() do { if (threadIdx.x < 24) s_prof[threadIdx.x] = 0; } while (0)
#define PROF_CNT(i, n) do { const unsigned long long _c = (n); if (threadIdx.x == 0) s_prof[i] += _c; } while (0)
#define PROF_END() do { __builtin_amdgcn_s_waitcnt(0xc07f); if (threadIdx.x < 24 && s_prof[threadIdx.x]) atomicAdd(&g_prof[threadIdx.x], s_prof[threadIdx.x]); } while (0)
#else
#define PROF_T()
#define PROF(i)
#define PROF_BEGIN()
#define PROF_END()
#define PROF_CNT(i, n)
#endif
// ---- galloping, one slice of the short operand per work item ------------------------------------------------
// gmode: GAL_SCRATCH = the item's output goes compactly into the operator's scratch region, at the union index of its first
// emitted element (slot_out); GAL_COUNT = nothing is written, only counted; GAL_WRITE = written from tmp_op[0] on (the item's
// final place in the pool).  Round 4: an operator whose long list is many times its short one is evaluated TWICE (count, then
// write): its scratch region would be as long as the long list -- 1.8e8 elements for every phrase with the most frequent term --
// and slices of a few such operators each ran at a tenth of the device (one launch per slice, a handful of wavefronts).
constexpr int GAL_SCRATCH = 0, GAL_COUNT = 1, GAL_WRITE = 2;
__device__ void part_gallop(const CV& Sv, u64 ns, const CV& Lv, u64 nl, bool s_is_a, u32 absR, bool ordered,
                            u64 i_begin, u64 i_end, u64* __restrict__ tmp_op, u32* rec_lds, u64& count_out, u64& slot_out,
                            const u32* __restrict__ ldir, u32 ldir_shift, u64 ldir_last, int gmode) {
  const u32 lane = threadIdx.x;
  // records of the LISTED elements, by position in the list: the chains the dense passes resolved (finished records, only their
  // emission is left) and the ones that have to walk (r_n = NONE32 until flush() has walked them); an element without a neighbour
  // within absR emits nothing and needs no record
  u64* r_b = reinterpret_cast<u64*>(rec_lds);                 // [PW_TILE] first L index of the emitted part
  u32* r_d = rec_lds + 2 * PW_TILE;                           // [PW_TILE] number of L elements emitted from r_b
  u32* r_s0 = rec_lds + 3 * PW_TILE;                          // [PW_TILE] first S index emitted (relative to i_begin)
  u32* r_s1 = rec_lds + 4 * PW_TILE;                          // [PW_TILE] last S index emitted (inclusive, relative to i_begin)
  u32* r_n = rec_lds + 5 * PW_TILE;                           // [PW_TILE] emitted values (0 = not a leader / nothing)
  u16* g_list = reinterpret_cast<u16*>(rec_lds + 6 * PW_TILE);  // [PW_TILE] the walk list: element index relative to i_begin (< GAL_PER)
  u16* c_k = g_list + PW_TILE;                                  // [320] pending candidates (< 64 carried over + a round's 256): element index relative to i_begin ...
  u32* c_il = rec_lds + 6 * PW_TILE + PW_TILE / 2 + 160;         // [320] ... and position in the long list relative to the item's first element's
  u64 cursor = 0, slot = 0;
  bool have_slot = false;
  u32 n_list = 0;                                             // (wave-uniform)
  PROF_T();
  auto general = [&](u64 i, u32 k, u64 il) {
    const u64 s = Sv[i];
    u64 a = i, b = il, cur = s;                              // left walk over the union
    for (;;) {
      bool hs = a > 0, hl = b > 0;
      if (!hs && !hl) break;
      u64 cs = hs ? Sv[a - 1] : 0, cl = hl ? Lv[b - 1] : 0;
      bool takeS = hs && (!hl || cs >= cl);
      u64 cand = takeS ? cs : cl;
      if (cur - cand > (u64)absR) break;
      if (takeS) a--; else b--;
      cur = cand;
      if (a < i) break;   // an earlier short-list element is in this chain: not the leader
    }
    u32 n = 0;
    u64 eb = 0, es0 = i, es1 = i;
    u32 ed = 0;
    if (a == i) {
      u64 c = i, d = il;                                     // right walk
      cur = s;
      for (;;) {
        bool hs = c + 1 < ns, hl = d < nl;
        if (!hs && !hl) break;
        u64 cs = hs ? Sv[c + 1] : 0, cl = hl ? Lv[d] : 0;
        bool takeS = hs && (!hl || cs <= cl);
        u64 cand = takeS ? cs : cl;
        if (cand - cur > (u64)absR) break;
        if (takeS) c++; else d++;
        cur = cand;
      }
      // chain = S[i..c] u L[b..d)
      u64 lb = b, ld = d, sb = i;
      bool ok = d > b;                                      // needs an element of the long operand
      if (ok && ordered) {
        if (s_is_a) {                                       // first A element = s: drop L elements before it
          lb = il;
          ok = d > il;                                      // a B element at or after the first A
        } else {                                            // A is the long operand: first A = L[b]
          u64 fa = Lv[b];
          while (sb <= c && Sv[sb] < fa) sb++;             // drop B-only prefix
          ok = sb <= c;
        }
      }
      if (ok) {
        eb = lb; ed = (u32)(ld - lb); es0 = sb; es1 = c;
        n = gal_merge(Sv, sb, c, Lv, lb, ld, [](u64) {});
      }
    }
    r_b[k] = eb; r_d[k] = ed; r_s0[k] = (u32)(es0 - i_begin); r_s1[k] = (u32)(es1 - i_begin); r_n[k] = n;
  };
  // Round 3: the chain walks and the emission are BATCHED over the whole work item.  They are serial, latency-bound code -- a
  // lane follows its chain load by load, three traversals per emitting chain (walk, count, emit) -- and used to run once per
  // round of 256 short-list elements with the two or three elements of that round that had a neighbour: 56 % of the wave
  // residency of configs[3] (in-kernel phase profile) with a handful of lanes active.  Now the rounds only APPEND to the walk
  // list; one walk + emit phase serves the item's 1024 elements (or runs earlier when the list is about to overflow).
  auto flush = [&]() {
    wsync();
    PROF(13);  // gallop: (a pass that ran behind the last round)
    for (u32 b0 = 0; b0 < n_list; b0 += PW_T) {
      if (b0 + lane < n_list && r_n[b0 + lane] == NONE32) {       // (the others were resolved by the round that listed them)
        const u32 p = b0 + lane;
        const u64 i = i_begin + g_list[p];
        const u64 s = Sv[i];
        u64 il;                                                // first L >= s, located again (cheaper than keeping it in LDS)
        if (ldir) {
          const u64 c = min(s >> ldir_shift, ldir_last);
          u64 lo = ldir[c], hi = ldir[c + 1];
          while (lo < hi) { u64 m = (lo + hi) >> 1; if (Lv[m] < s) lo = m + 1; else hi = m; }
          il = lo;
        } else il = lower_bound_u64(Lv, nl, s);
        general(i, p, il);
      }
    }
    wsync();
    PROF(14);  // gallop: walks of the unresolved chains
    // entry p = q * 64 + lane: a list of 40 entries keeps 40 lanes busy for one serial merge each (with four consecutive entries
    // per lane it was ten lanes for four merges)
    u32 nq_[PW_ITEMS], exq[PW_ITEMS];
    u32 total = 0;
#pragma unroll
    for (int q = 0; q < (int)PW_ITEMS; q++) {
      const u32 p = (u32)q * PW_T + lane;
      nq_[q] = p < n_list ? r_n[p] : 0u;
      const u32 incl = wave_scan_incl(nq_[q]);
      exq[q] = total + incl - nq_[q];
      total += rdl(incl, 63);
    }
    if (gmode == GAL_SCRATCH && !have_slot && total) {
      // scratch slot of this work item = union index (S index + L index) of its first emitted element
      u64 cand = 0;
      bool found = false;
#pragma unroll
      for (int q = 0; q < (int)PW_ITEMS; q++) {
        const u64 m = ballot64(nq_[q] != 0u);
        if (!found && m) {
          const u32 fl = (u32)__builtin_ctzll(m), p = (u32)q * PW_T + fl;
          cand = r_b[p] + i_begin + r_s0[p];                   // (LDS reads of one address: every lane gets the value)
          found = true;
        }
      }
      slot = cand;
      have_slot = true;
    }
    u64* out = tmp_op + slot;
    if (gmode != GAL_COUNT) {
#pragma unroll
      for (int q = 0; q < (int)PW_ITEMS; q++) {
        const u32 p = (u32)q * PW_T + lane;
        if (nq_[q]) {
          u64 o = cursor + exq[q];
          gal_merge(Sv, i_begin + r_s0[p], i_begin + r_s1[p], Lv, r_b[p], r_b[p] + r_d[p], [&](u64 v) { out[o++] = v; });
        }
      }
    }
    cursor += total;
    wsync();                                                   // (the records are free again; the stores stay in flight)
    n_list = 0;
    PROF(15);  // gallop: scan + emission
  };
  u32 n_c = 0;                                                // pending candidates (wave-uniform)
  u64 il_base = 0;                                            // position of the item's first element in the long list
  const u64 s_last = ns - 1, l_last = nl - 1;
  auto resolve_pass = [&](u32 c0, u32 cnt) {                  // candidates c0 .. c0 + cnt (cnt <= 64), one per lane
    // Round 3: the chain of a candidate is RESOLVED IN REGISTERS from a window of both lists around it -- S[i-1 .. i+2] and
    // L[il-2 .. il+1], fetched together in one round trip (cache hits: the round has just touched these lines) -- whenever it
    // closes inside the window: the same decisions as the walk (general()), without its dependent loads.  Phrases and
    // Dist-sized windows over sparse lists give chains of two or three elements, so nearly every candidate ends here: as "emits
    // nothing" (not the leader of its chain, no element of the long operand, the ordered rule fails: dropped -- two thirds of
    // them on configs[4]) or as a finished record on the list (only its emission is left for flush()).  The rare longer chain
    // goes on the list unresolved (r_n = NONE32) and walks.
    {
      const bool on = lane < cnt;
      const u32 k = on ? c_k[c0 + lane] : 0u;                  // element number relative to i_begin
      const u32 il32 = on ? c_il[c0 + lane] : 0u;
      const u64 i = i_begin + k, il = il_base + (il32 == 0xFFFFFFFFu ? 0u : il32);
      const u64 ic = min(i, s_last);
      const u64 s = Sv[ic];
      const u64 sp = Sv[ic > 0 ? ic - 1 : 0], sn = Sv[min(ic + 1, s_last)], snn = Sv[min(ic + 2, s_last)];
      const u64 lpp = Lv[il > 1 ? min(il - 2, l_last) : 0], lp = Lv[il > 0 ? min(il - 1, l_last) : 0];
      const u64 ln = Lv[min(il, l_last)], lnn = Lv[min(il + 1, l_last)];
      const bool hsp = i > 0, hlp = il > 0, hlpp = il > 1;
      const u64 lft = (hsp || hlp) ? max(hsp ? sp : 0ull, hlp ? lp : 0ull) : ~0ull;
      const bool left_far = lft == ~0ull || s - lft > (u64)absR;
      bool listed = false;
      u64 e_b = 0;
      u32 e_d = 0, e_s0 = 0, e_s1 = 0, e_n = NONE32;
      if (on) {
        bool leader = true, unres = il32 == 0xFFFFFFFFu;       // (a round that spans 2^32 postings: the walk locates the element)
        u32 bl = 0;                                            // left walk: long-list elements taken (b = il - bl)
        if (!unres && !left_far) {
          bool takeS = hsp && (!hlp || sp >= lp);
          if (takeS) leader = false;                           // (the nearest left neighbour is within absR: !left_far)
          else {
            bl = 1;
            u64 cur = lp;
            if (hsp || hlpp) {
              takeS = hsp && (!hlpp || sp >= lpp);
              const u64 cv = takeS ? sp : lpp;
              if (cur - cv <= (u64)absR) {
                if (takeS) leader = false;
                else {
                  bl = 2;
                  cur = lpp;
                  if (il > 2) unres = true;                    // L[il - 3] is outside the window
                  else if (hsp && cur - sp <= (u64)absR) leader = false;
                }
              }
            }
          }
        }
        if (unres) listed = true;
        else if (leader) {
          u32 cr = 0, dr = 0;                                  // right walk: elements taken (c = i + cr, d = il + dr)
          u64 cur = s;
          bool open = true;
#pragma unroll
          for (int it = 0; it < 5; it++) {
            if (open) {
              const bool hs = i + cr + 1 < ns, hl = il + dr < nl;
              if ((hs && cr >= 2u) || (hl && dr >= 2u)) { unres = true; open = false; }   // S[i + 3] / L[il + 2]: outside the window
              else if (!hs && !hl) open = false;
              else {
                const u64 cs = cr == 0u ? sn : snn, cl = dr == 0u ? ln : lnn;
                const bool takeS = hs && (!hl || cs <= cl);
                const u64 cv = takeS ? cs : cl;
                if (cv - cur > (u64)absR) open = false;
                else { if (takeS) cr++; else dr++; cur = cv; }
              }
            }
          }
          if (unres) listed = true;
          else {
            // chain = S[i .. i + cr] u L[il - bl .. il + dr): the rules of general()
            u32 lb_back = bl, k0 = 0;
            bool ok = bl + dr > 0u;                            // needs an element of the long operand
            if (ok && ordered) {
              if (s_is_a) { lb_back = 0; ok = dr > 0u; }       // first A = s: long-list elements before it are dropped
              else {                                          // first A = L[il - bl]: the B-only prefix is dropped
                const u64 fa = bl == 2u ? lpp : (bl == 1u ? lp : ln);
                k0 = (s < fa ? 1u : 0u) + ((cr >= 1u && sn < fa) ? 1u : 0u) + ((cr >= 2u && snn < fa) ? 1u : 0u);
                ok = k0 <= cr;
              }
            }
            if (ok) {
              // a coordinate held by both operands is emitted once (gal_merge): only L[il], L[il + 1] can equal a short-list member
              const bool m0 = dr >= 1u, m1 = dr >= 2u;
              u32 dups = 0;
              if (k0 == 0u) dups += ((m0 && s == ln) ? 1u : 0u) + ((m1 && s == lnn) ? 1u : 0u);
              if (k0 <= 1u && cr >= 1u) dups += ((m0 && sn == ln) ? 1u : 0u) + ((m1 && sn == lnn) ? 1u : 0u);
              if (cr >= 2u) dups += ((m0 && snn == ln) ? 1u : 0u) + ((m1 && snn == lnn) ? 1u : 0u);
              e_b = il - lb_back;
              e_d = lb_back + dr;
              e_s0 = (u32)(i - i_begin) + k0;
              e_s1 = (u32)(i - i_begin) + cr;
              e_n = (cr + 1u - k0) + lb_back + dr - dups;
              listed = true;
            }
          }
        }
      }
      const u64 wm = ballot64(listed);
      PROF_CNT(18, popc64(ballot64(listed && e_n != NONE32)));   // ... resolved, emitting
      PROF_CNT(19, popc64(ballot64(listed && e_n == NONE32)));   // ... unresolved (walk)
      if (n_list + popc64(wm) > PW_TILE) flush();
      // (entries in element order: the emission relies on it)
      if (listed) {
        const u32 p = n_list + mbcnt(wm);
        g_list[p] = (u16)k;
        r_b[p] = e_b; r_d[p] = e_d; r_s0[p] = e_s0; r_s1[p] = e_s1; r_n[p] = e_n;
      }
      n_list += popc64(wm);
    }
  };
  for (u64 i0 = i_begin; i0 < i_end; i0 += PW_TILE) {
    const u32 cn = (u32)min((u64)PW_TILE, i_end - i0);
    // Every lane looks after PW_ITEMS short-list elements.  Their loads are issued together, phase by phase (the
    // chain of dependent accesses -- directory cell, long-list position, neighbours -- is latency bound, so the
    // elements of a lane overlap each other): position in the long list, then the four union neighbours.  An
    // element without a neighbour within absR on either side is a chain of its own and emits nothing; only the
    // others go on the walk list.
    {
      // (all loads below are unconditional on clamped, always-valid indices: a load under a lane predicate would put
      // a branch and a wait between the loads of the PW_ITEMS elements)
      u64 sv[PW_ITEMS], ilv[PW_ITEMS], hiv[PW_ITEMS];
      bool act[PW_ITEMS];
#pragma unroll
      for (int q = 0; q < (int)PW_ITEMS; q++) {
        const u32 k = lane + (u32)q * PW_T;
        act[q] = k < cn;
        sv[q] = Sv[min(i0 + k, s_last)];
      }
      if (ldir) {                                              // cell directory: the search starts inside the cell of s
#pragma unroll
        for (int q = 0; q < (int)PW_ITEMS; q++) {
          const u64 c = min(sv[q] >> ldir_shift, ldir_last);
          struct E2 { u32 a, b; } e;                             // both ends of the cell in ONE 8-byte request
          __builtin_memcpy(&e, ldir + c, sizeof(e));
          const u32 e0 = e.a, e1 = e.b;
          ilv[q] = e0;
          hiv[q] = act[q] ? e1 : e0;
        }
        bool more = true;
        while (more) {                                         // the PW_ITEMS binary searches advance in lock step
          more = false;
          u64 mv[PW_ITEMS], lv[PW_ITEMS];
#pragma unroll
          for (int q = 0; q < (int)PW_ITEMS; q++) { mv[q] = (ilv[q] + hiv[q]) >> 1; lv[q] = ilv[q] < hiv[q] ? Lv[min(mv[q], l_last)] : 0ull; }   // (finished searches load nothing)
#pragma unroll
          for (int q = 0; q < (int)PW_ITEMS; q++) {
            if (ilv[q] < hiv[q]) {
              if (lv[q] < sv[q]) ilv[q] = mv[q] + 1; else hiv[q] = mv[q];
              more |= ilv[q] < hiv[q];
            }
          }
          more = ballot64(more) != 0;
        }
      } else {
#pragma unroll
        for (int q = 0; q < (int)PW_ITEMS; q++) ilv[q] = act[q] ? lower_bound_u64(Lv, nl, sv[q]) : 0;
      }
      PROF(11);  // gallop: short elements, directory cells, position in the long list
      // An element without a neighbour within absR on either side is a chain of its own and emits nothing: the nearest union
      // neighbours S[i - 1], S[i + 1], L[il - 1], L[il] decide that for ALL elements of the round (four loads each).  The few that do
      // have one -- 10 % on configs[3], a third on configs[4] -- are COMPACTED (element number + position, in element order) and
      // looked at again below with full lanes: the window loads and the in-register resolution run once or twice per round over
      // the candidates instead of four times over every element slot with a tenth of the lanes active.
      if (i0 == i_begin) il_base = rfl64(ilv[0]);                // (positions ascend with the elements: the item's first is the smallest)
      {
        u64 sp[PW_ITEMS], sn[PW_ITEMS], lp[PW_ITEMS], ln[PW_ITEMS];
#pragma unroll
        for (int q = 0; q < (int)PW_ITEMS; q++) {
          const u64 i = min(i0 + lane + (u64)q * PW_T, s_last);
          const u64 il = ilv[q];
          sp[q] = Sv[i > 0 ? i - 1 : 0];
          sn[q] = Sv[min(i + 1, s_last)];
          lp[q] = Lv[il > 0 ? min(il - 1, l_last) : 0];
          ln[q] = Lv[min(il, l_last)];
        }
#pragma unroll
        for (int q = 0; q < (int)PW_ITEMS; q++) {
          const u64 i = i0 + lane + (u64)q * PW_T, il = ilv[q], s = sv[q];
          const bool hsp = i > 0, hlp = il > 0, hsn = i + 1 < ns, hln = il < nl;
          const u64 lft = (hsp || hlp) ? max(hsp ? sp[q] : 0ull, hlp ? lp[q] : 0ull) : ~0ull;   // nearest union neighbours (~0: none)
          const u64 rgt = min(hsn ? sn[q] : ~0ull, hln ? ln[q] : ~0ull);
          const bool left_far = lft == ~0ull || s - lft > (u64)absR;
          const bool right_far = rgt == ~0ull || rgt - s > (u64)absR;
          const bool cand = act[q] && !(left_far && right_far);
          const u64 cm = ballot64(cand);
          if (cand) {
            const u32 pos = n_c + mbcnt(cm);
            c_k[pos] = (u16)((u32)(i0 - i_begin) + lane + (u32)q * PW_T);
            c_il[pos] = il - il_base < 0xFFFFFFFFull ? (u32)(il - il_base) : 0xFFFFFFFFu;
          }
          n_c += popc64(cm);
          PROF_CNT(16, popc64(ballot64(act[q])));                // short-list elements
          PROF_CNT(17, popc64(cm));                              // ... with a neighbour within absR
        }
      }
      wsync();
      PROF(12);  // gallop: neighbours, candidates
      // full passes of 64 candidates now; the rest waits for the next round (a round of configs[3] leaves ~25: a pass per round
      // would run the resolution at 40 % of the lanes)
      {
        u32 c_head = 0;
        while (n_c - c_head >= PW_T) { resolve_pass(c_head, PW_T); c_head += PW_T; }
        if (c_head) {
          const u32 rem = n_c - c_head;
          const u16 tk = lane < rem ? c_k[c_head + lane] : (u16)0;
          const u32 ti = lane < rem ? c_il[c_head + lane] : 0u;
          wsync();
          if (lane < rem) { c_k[lane] = tk; c_il[lane] = ti; }
          n_c = rem;
          wsync();
        }
      }
      PROF(13);  // gallop: dense passes over the candidates (window loads, chains resolved in registers, list append)
    }
  }
  if (n_c) resolve_pass(0, n_c);
  if (n_list) flush();
  count_out = cursor;
  slot_out = slot;
}

#ifndef DCD_PW_WAVES
#define DCD_PW_WAVES 5
#endif
// Work plan of a round, computed on the device once the operand counts are known: how an operator is evaluated
// (mode) and how many work items it needs -- ceil(union / PART) for the streaming merge, ceil(|short| / GAL_PER) when
// it gallops, none when a '*' has an empty operand.  An exclusive scan of the item counts gives the operator of
// every work item; the launch grid is sized on the host for the capacities, surplus wavefronts leave at once.
#ifndef DCD_GAL_PER
#define DCD_GAL_PER 1024
#endif
constexpr u32 GAL_PER = DCD_GAL_PER;
static_assert(DCD_GAL_PER <= 65536, "part_gallop's walk-list entries are 16-bit element indices");      // short-list elements per galloping work item (256 / 512 / 1024 / 2048 -> 26.5 / 24.3 / 23.8 / 23.8 ms per 20 k C4 queries)
constexpr u8 MODE_NONE = 0, MODE_STREAM = 1, MODE_GALLOP = 2, MODE_STREAM32 = 3, MODE_GALLOP2 = 4;
#ifndef DCD_GAL2_RATIO
#define DCD_GAL2_RATIO 256
#endif
// Work plan of the operators [0, n_ops) of a round that are still to be evaluated, computed on the device once the operand counts
// are known.  Per operator: mode, work items, and the scratch it NEEDS -- |a| + |b| REAL elements (not the host's capacity bound)
// for a streaming or galloping operator, nothing for a two-pass galloping one (long list >= DCD_GAL2_RATIO x short list and the
// two together at least 1 / 128 of the scratch arena; configs[4], 100 k phrases: ratio 16 / 64 / 256 / never -> 2.67 / 2.56 / 2.51 /
// 2.76 s in 309 / 352 / 442 / 1713 passes).
// items_all: work items of the operator; items_old / items_s32 / items_g2: the same number in the array of the kernel launch that
// evaluates them (k_pair_parts: galloping items and, with stream32 == 0, the streaming ones; k_pair_stream: the streaming items;
// the write pass of k_pair_parts: the two-pass items again) -- each launch numbers its own items through the scan of its array
// and records an item at the operator's place in the scan of items_all.
struct PlanOut { u64* need; u32* items_all; u32* items_old; u32* items_s32; u32* items_g2; u8* mode; };
__global__ void k_pair_plan(const PairOp* __restrict__ ops, u32 n_ops, IdxVals idx_values, Pool pool,
                            u64 PART, PlanOut o, u32 gal_ratio, u32 stream32, u64 g2_min_need, u32 g2_ratio) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_ops) return;
  if (i == n_ops || *pool.overflow) {                     // (overflow: the pool ran full, the sub-batch is abandoned)
    o.need[i] = 0; o.items_all[i] = 0; o.items_old[i] = 0; o.items_s32[i] = 0; o.items_g2[i] = 0;
    if (i < n_ops) o.mode[i] = MODE_NONE;
    return;
  }
  const PairOp op = ops[i];
  const u64 na = seg_cnt(op.a, pool), nb = seg_cnt(op.b, pool);
  const bool is_or = op.op == DCD_OP_OR;
  const u32 absR = (u32)max(abs(op.Ra), abs(op.Rb));
  u8 m = stream32 ? MODE_STREAM32 : MODE_STREAM;
  u64 n = (na + nb + PART - 1) / PART;
  if (!is_or && (na == 0 || nb == 0)) { m = MODE_NONE; n = 0; }          // '*' with an empty operand is empty
  else if (!is_or && absR != 0) {
    const bool a_short = na <= nb;
    const u64 ns = a_short ? na : nb, nl = a_short ? nb : na;
    const CV Lv = seg_cv(a_short ? op.b : op.a, idx_values, pool);
    if (ns * (u64)gal_ratio <= nl && (Lv[nl - 1] - Lv[0]) / nl > 4ull * absR) {
      // two passes (twice the searches) only where the scratch region would be a noticeable part of the arena: conjunctions of
      // ordinary terms (configs[3]: lists of at most ~1e6 postings) keep the single pass
      m = (ns * (u64)g2_ratio <= nl && na + nb >= g2_min_need) ? MODE_GALLOP2 : MODE_GALLOP;
      n = (ns + GAL_PER - 1) / GAL_PER;
    }
  }
  o.need[i] = (m == MODE_NONE || m == MODE_GALLOP2) ? 0ull : na + nb;
  o.items_all[i] = (u32)n;
  o.items_old[i] = m == MODE_STREAM32 ? 0u : (u32)n;
  o.items_s32[i] = m == MODE_STREAM32 ? (u32)n : 0u;
  o.items_g2[i] = m == MODE_GALLOP2 ? (u32)n : 0u;
  o.mode[i] = m;
}
// how many of the planned operators run in this pass: the longest prefix whose scratch needs and work items fit (at least one:
// the arena holds the largest single operator by construction)
__global__ void k_pair_admit(u32 n_ops, const u64* __restrict__ need_off, const u64* __restrict__ part_base, u64 scratch_cap, u64 item_cap,
                             u64* __restrict__ n_adm) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u32 lo = 0, hi = n_ops;                                  // largest m with need_off[m] <= scratch_cap && part_base[m] <= item_cap
  while (lo < hi) { const u32 m = (lo + hi + 1) >> 1; if (need_off[m] <= scratch_cap && part_base[m] <= item_cap) lo = m; else hi = m - 1; }
  *n_adm = lo ? lo : 1u;
}

__global__ __launch_bounds__(PW_T, DCD_PW_WAVES) void k_pair_parts(const PairOp* __restrict__ ops, u32 n_ops, const u64* __restrict__ kind_base,
                                                    const u64* __restrict__ part_base,
                                                    IdxVals idx_values, Pool pool, u64* __restrict__ tmp, u64* __restrict__ part_count,
                                                    u64* __restrict__ part_src, u64 PART, const u32* __restrict__ dirs,
                                                    u64 dir_max_coord, const u8* __restrict__ op_mode, u64 W_cap, unsigned long long* __restrict__ work_ctr,
                                                    const u64* __restrict__ tmp_off, int pass, u64* __restrict__ pool_w, const u64* __restrict__ excl) {
  // pass 0: every item of this launch's kind; two-pass galloping operators are only COUNTED.  pass 1 (kind_base = the scan of the
  // two-pass items): those operators again, written to their final place in the pool (res_off is known by now).
  if (pass != 0 && *pool.overflow) return;                 // (the results do not fit the pool: nothing is written, the host retries)
  constexpr u32 RAW_WORDS = 6u * PW_TILE + PW_TILE / 2u + 160u + 320u + 16u;   // rings + chain flags, or the gallop records + walk list + candidate list
  __shared__ __attribute__((aligned(16))) u32 s_raw[RAW_WORDS];
  u64* sA = reinterpret_cast<u64*>(s_raw);
  u64* sB = sA + PW_TILE;
  u32* s_cm = s_raw + 4 * PW_TILE;                 // [PW_TILE + 1] chain flags of the general path
  PROF_BEGIN();
  const u32 lane = threadIdx.x;
  // Round 3: a FIXED grid of wavefronts works through the items the device plan really made (part_base[n_ops]), each taking
  // its next item from a counter.  One wavefront per item of the host's capacity bound meant up to 1.5 M wavefronts that left at
  // once in the second round of a configs[3] batch (the bound of an intermediate result is the sum of its operands): 0.7 ns
  // each, 1 - 2 ms of a 20 ms batch.  (A static stride over the items instead of the counter was 10 % slower on configs[3]:
  // the items differ too much.)  The counter is read one item ahead, so its round trip hides behind the item's work.
  // (kind_base: exclusive scan of THIS kernel's items per operator; part_base: of all items of the slice -- the streaming items
  // may belong to k_pair_stream, which runs beside this kernel)
  const u64 n_items = kind_base[n_ops];
  if (pass == 0)
    for (u64 x = part_base[n_ops] + (u64)blockIdx.x * PW_T + lane; x < W_cap; x += (u64)gridDim.x * PW_T) { part_count[x] = 0; part_src[x] = 0; }   // (the scans run over the capacity)
  u64 w = blockIdx.x;
  while (w < n_items) {
  unsigned long long w_next = 0;
  if (lane == 0) w_next = atomicAdd(work_ctr, 1ull) + gridDim.x;
  PROF_T();
  u32 oi;
  {   // operator of this work item: last op with kind_base[op] <= w
    u32 lo = 0, hi = n_ops;
    while (lo + 1 < hi) { u32 m = (lo + hi) >> 1; if (kind_base[m] <= w) lo = m; else hi = m; }
    oi = lo;
  }
  PROF(0);   // operator lookup
  const u64 p = w - kind_base[oi];
  const u64 wi = part_base[oi] + p;              // the item's place among all items of the slice
  const PairOp op = ops[oi];
  const CV A = seg_cv(op.a, idx_values, pool);
  const CV B = seg_cv(op.b, idx_values, pool);
  const u64 na = seg_cnt(op.a, pool), nb = seg_cnt(op.b, pool);
  const u8 omode = op_mode[oi];
  const int gmode = omode == MODE_GALLOP2 ? (pass == 0 ? GAL_COUNT : GAL_WRITE) : GAL_SCRATCH;
  u64* tmp_op = gmode == GAL_WRITE ? pool_w + pool.res_off[op.out_cnt_ref] + (excl[wi] - excl[part_base[oi]]) : tmp + tmp_off[oi];
  const u32 absR = (u32)max(abs(op.Ra), abs(op.Rb));
  const bool ordered = op.Ra < 0 && op.Rb < 0 && absR != 0;   // R = -absR < 0  (IndexSequence.cs:221-222)
  const bool is_or = op.op == DCD_OP_OR;
  const bool gallop = omode == MODE_GALLOP || omode == MODE_GALLOP2;
  u64 count = 0, slot = 0;
  {
    if (gallop) {
      const bool a_short = na <= nb;
      const u64 ns = a_short ? na : nb, nl = a_short ? nb : na;
      const CV Sv = a_short ? A : B;
      const CV Lv = a_short ? B : A;
      const u64 i_begin = min(ns, p * (u64)GAL_PER), i_end = min(ns, i_begin + GAL_PER);
      const Seg& ls = a_short ? op.b : op.a;
      const u32* ldir = (ls.kind == 0 && ls.dir_shift != NONE32) ? dirs + ls.dir_off : nullptr;
      if (i_begin < i_end) part_gallop(Sv, ns, Lv, nl, a_short, absR, ordered, i_begin, i_end, tmp_op, s_raw, count, slot, ldir, ls.dir_shift,
                                       ldir ? dir_max_coord >> ls.dir_shift : 0, gmode);
    }
    PROF(1);   // operands, mode (gallop: everything)
    if (!gallop) {
      const u64 d_begin = p * PART, d_end = d_begin + PART;
      u64 ia = wave_split(A, na, B, nb, d_begin), ib = d_begin - ia;
      u64 prev_val = 0;
      u32 prev_src = 0, have_prev = 0;
      if (d_begin > 0) {
        have_prev = 1;
        const bool ua = ia > 0, ub = ib > 0;
        const u64 pa = ua ? A[ia - 1] : 0, pb = ub ? B[ib - 1] : 0;
        if (ua && ub) { if (pb >= pa) { prev_val = pb; prev_src = 1; } else { prev_val = pa; prev_src = 0; } }
        else if (ua) { prev_val = pa; prev_src = 0; }
        else { prev_val = pb; prev_src = 1; }
      }
      u64 g = d_begin;                 // union index of the next unconsumed element
      u64 cursor = 0, out_start = 0;
      u32 open = 0, committed = 0, seenA = 0;
      u32 hA = 0, hB = 0;
      bool skipping = !is_or;          // '*': nothing is owned before the first chain break of the partition
      u64* out = tmp_op + d_begin;
      slot = d_begin;
      // Operands are staged in two LDS rings of PW_TILE elements.  A tile consumes ca + cb = PW_TILE elements; the
      // refill of the freed slots is requested from HBM as soon as ca is known (a wave-wide merge-path probe) and is
      // stored into the rings after the tile's merge has read them, so the load latency overlaps the tile's work.
      u32 ra = (u32)min((u64)PW_TILE, na - ia), rb = (u32)min((u64)PW_TILE, nb - ib);
      for (u32 i = lane; i < ra; i += PW_T) sA[i] = A[ia + i];
      for (u32 i = lane; i < rb; i += PW_T) sB[i] = B[ib + i];
      wsync();
      PROF(2);   // split search over HBM, first fill
      while (ra + rb != 0) {
        if (g >= d_end && (skipping || is_or)) break;
        auto Aat = [&](u32 x) -> u64 { return sA[(hA + x) & (PW_TILE - 1)]; };
        auto Bat = [&](u32 x) -> u64 { return sB[(hB + x) & (PW_TILE - 1)]; };
        const u32 cnt = min((u32)PW_TILE, ra + rb);
        // ---- tile front end: bitonic merge in registers ----------------------------------------------------------
        // The tile is the cnt smallest of the staged elements.  Keys = (coordinate << 1) | operand, so that a left
        // element sorts before an equal right element (ties take A first) and all keys are distinct; slots past the
        // staged counts hold +inf.  A ascending and B REVERSED form a bitonic sequence of 512 keys; one half-cleaner
        // step, min(A[t], B[255 - t]), leaves the 256 smallest as a bitonic sequence (and tells how many elements
        // of each operand the tile consumes), eight compare-exchange stages sort them: distances 128 .. 4 cross lanes
        // (shuffles), 2 and 1 stay inside a lane (element t = 4 * lane + k).  No data-dependent LDS addressing: the
        // merge-path search per lane and the serial merge were 3/4 of the kernel's time in bank-conflicted,
        // dependent LDS round trips.
        constexpr u64 INF = ~0ull;
        u64 x[PW_ITEMS];
        u32 tookA = 0;
#pragma unroll
        for (int k = 0; k < (int)PW_ITEMS; k++) {
          const u32 t = lane * PW_ITEMS + (u32)k, u = PW_TILE - 1u - t;
          const u64 av = Aat(t), bv = Bat(u);
          const u64 ka = t < ra ? (av << 1) : INF;
          const u64 kb = u < rb ? ((bv << 1) | 1ull) : INF;
          const bool ta = ka <= kb;                            // (inf <= inf only beyond the staged elements: never consumed)
          x[k] = ta ? ka : kb;
          tookA += (ta && t < ra) ? 1u : 0u;
        }
        const u32 ca = rfl(wave_sum(tookA)), cb = cnt - ca;    // A / B elements among the tile's cnt union elements
        // request the refill
        const u32 keepA = ra - ca, keepB = rb - cb;
        const u32 nra = (u32)min((u64)PW_TILE, na - ia - ca), nrb = (u32)min((u64)PW_TILE, nb - ib - cb);
        const u32 newA = nra - keepA, newB = nrb - keepB;
        // newA + newB <= PW_TILE (a tile frees ca + cb slots): ONE stream of PW_ITEMS loads per lane covers both operands
        u64 pf[PW_ITEMS];
        {
          // (per lane the element comes from ONE operand: plane pointers are selected per lane, so that a packed operand costs
          // its two plane loads and a plain one its single load -- never the loads of both operands)
          const u64 offA = ia + ra, offB = ib + rb - newA;
#pragma unroll
          for (int q = 0; q < (int)PW_ITEMS; q++) {
            const u32 xx = lane + (u32)q * PW_T;
            const bool fa = xx < newA, act = xx < newA + newB;
            const u64 ix = (fa ? offA : offB) + xx;
            const u64* pv = fa ? A.v : B.v;
            const u32* pl = fa ? A.lo : B.lo;
            const u8* ph = fa ? A.hi : B.hi;
            u64 val = 0;
            if (act) {
              if (pv) val = pv[ix];
              else val = ((u64)ph[ix] << 32) | pl[ix];
            }
            pf[q] = val;
          }
        }
        PROF(3);   // half-cleaner, refill request
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {                    // element distance 4 * m: partner lane = lane ^ m
          const bool keep_min = (lane & (u32)m) == 0;
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) {
            const u64 y = shfl_xor64(x[k], m);
            const bool lt = x[k] < y;
            x[k] = (lt == keep_min) ? x[k] : y;
          }
        }
        {                                                      // distances 2 and 1 inside the lane
          u64 t0 = min(x[0], x[2]), t2 = max(x[0], x[2]), t1 = min(x[1], x[3]), t3 = max(x[1], x[3]);
          x[0] = min(t0, t1); x[1] = max(t0, t1); x[2] = min(t2, t3); x[3] = max(t2, t3);
        }
        PROF(4);   // bitonic sort
        const u32 d0 = min(lane * PW_ITEMS, cnt), d1 = min(d0 + PW_ITEMS, cnt);
        u64 val[PW_ITEMS];
        u32 src[PW_ITEMS], dup[PW_ITEMS], brk[PW_ITEMS], ctc[PW_ITEMS], al[PW_ITEMS];
        {
          const u64 thr = absR != 0 ? (u64)absR : INF;          // absR == 0: nothing ever breaks a chain
          // element before this lane's first one: the previous lane's last key, or the carried element
          const u64 pk = shfl_up64(x[PW_ITEMS - 1], 1);
          u64 pv = lane == 0 ? prev_val : (pk >> 1);
          u32 psrc = lane == 0 ? prev_src : (u32)(pk & 1ull);
          u32 hp = lane == 0 ? have_prev : 1u;
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) {
            const bool valid = d0 + k < d1;
            const u64 v = x[k] >> 1;
            const u32 sr = (u32)(x[k] & 1ull);
            const u64 gap = v - pv;
            const bool bk = !hp || gap > thr;
            val[k] = valid ? v : 0;
            src[k] = valid ? sr : 0u;
            dup[k] = (valid && sr == 1 && hp && psrc == 0 && gap == 0) ? 1u : 0u;
            brk[k] = (valid && bk) ? 1u : 0u;
            ctc[k] = (valid && hp && psrc != sr && !bk) ? 1u : 0u;   // neighbours from different operands within absR
            al[k] = valid ? 1u : 0u;
            pv = v; psrc = sr; hp = 1;
          }
        }
        PROF(5);   // merge
        // the rings have been read: store the refill (visible to the next tile after the wsync that ends this one)
        wsync();
#pragma unroll
        for (int q = 0; q < (int)PW_ITEMS; q++) {
          const u32 x = lane + (u32)q * PW_T;                 // sB = sA + PW_TILE
          const u32 slot = x < newA ? ((hA + ra + x) & (PW_TILE - 1)) : (PW_TILE + ((hB + rb + x - newA) & (PW_TILE - 1)));
          if (x < newA + newB) sA[slot] = pf[q];
        }
        PROF(6);   // refill store (waits for the HBM loads)
        // the tile's last element (all lanes)
        const u32 last_lane = (cnt - 1) / PW_ITEMS, last_k = (cnt - 1) % PW_ITEMS;
        u64 lv = 0; u32 ls = 0;
#pragma unroll
        for (int k = 0; k < (int)PW_ITEMS; k++) if ((u32)k == last_k) { lv = val[k]; ls = src[k]; }
        const u64 t_last_val = rdl64(lv, last_lane);
        const u32 t_last_src = rdl(ls, last_lane);
        const bool inputs_end = (ia + ca == na) && (ib + cb == nb);
        // ---- ownership: [t_lo, t_hi) of this tile belongs to the work item ------------------------------
        u32 t_lo = 0, t_hi = cnt;
        const bool may_end = g + cnt > d_end;
        if (skipping || may_end) {
          u32 fb = NONE32, fe = NONE32;
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) {
            if (d0 + k < d1 && (is_or || brk[k])) {
              if (fb == NONE32) fb = d0 + k;
              if (fe == NONE32 && g + d0 + k >= d_end) fe = d0 + k;
            }
          }
          fb = rfl(wave_min(fb)); fe = rfl(wave_min(fe));
          if (skipping) t_lo = min(fb, cnt);
          if (may_end) t_hi = min(fe, cnt);
          if (t_hi < t_lo) t_hi = t_lo;
        }
        const bool ends_here = t_hi < cnt;       // a chain break at or beyond the partition end: stop before it
        auto advance = [&]() {
          ia += ca; ib += cb;
          hA = (hA + ca) & (PW_TILE - 1); hB = (hB + cb) & (PW_TILE - 1);
          ra = nra; rb = nrb;
          prev_val = t_last_val; prev_src = t_last_src; have_prev = 1;
          g += cnt;
        };
        if (t_lo >= t_hi) {                      // nothing owned in this tile
          if (ends_here) {
            if (open && !committed) cursor = out_start;   // the carried chain closes without the other operand
            open = 0;
            break;
          }
          advance();
          wsync();
          continue;
        }
        if (skipping) { skipping = false; slot = g + t_lo; out = tmp_op + slot; }
        const bool has_dead = t_lo > 0 || ends_here;
        if (has_dead) {
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) {
            const bool a = d0 + k < d1 && d0 + k >= t_lo && d0 + k < t_hi;
            al[k] = a ? 1u : 0u;
            if (!a) { brk[k] = 0; ctc[k] = 0; }
          }
        }
        const bool final_tile = ends_here || inputs_end;
        u32 contact = 0, lastbrk1 = 0;
#pragma unroll
        for (int k = 0; k < (int)PW_ITEMS; k++) {
          contact |= ctc[k];
          if (brk[k]) lastbrk1 = d0 + k + 1;
        }
        PROF(7);   // ownership
        // ---- sparse fast path: no two neighbouring elements from different operands within absR ------
        if (!is_or && absR != 0 && !has_dead) {
          const u32 M = rfl(wave_max(contact ? 0xFFFFFFFFu : lastbrk1));
          const u32 t_first_brk = rfl(brk[0]);
          const bool carried_committed_continues = open && committed && !t_first_brk;
          if (M != 0xFFFFFFFFu && !carried_committed_continues) {
            if (M == 0) {                                  // the carried chain spans the whole tile
              const bool elig = !ordered || seenA;
              if (final_tile) cursor = out_start;          // closes without the other operand: roll back
              else if (elig) {
#pragma unroll
                for (int k = 0; k < (int)PW_ITEMS; k++) if (d0 + k < d1) out[cursor + d0 + k] = val[k];
                cursor += cnt;
              }
            } else {
              const u32 LB = M - 1;
              const u64 cursor0 = (open && !committed) ? out_start : cursor;   // the carried chain closes: roll back
              if (final_tile) cursor = cursor0;
              else {
                const bool elig = !ordered || t_last_src == 0;
                if (elig) {
#pragma unroll
                  for (int k = 0; k < (int)PW_ITEMS; k++) if (d0 + k < d1 && d0 + k >= LB) out[cursor0 + (d0 + k - LB)] = val[k];
                }
                cursor = cursor0 + (elig ? cnt - LB : 0u);
                out_start = cursor0;
                open = 1; committed = 0; seenA = t_last_src == 0 ? 1u : 0u;
              }
            }
            advance();
            wsync();
            PROF(8);   // fast path
            continue;
          }
        }
        // ---- general path ---------------------------------------------------------------------------------
        // A chain is emitted iff it holds both operands, i.e. iff two NEIGHBOURS of it come from different operands
        // (ordered mode: an element of the right operand directly after one of the left operand -- exactly "a right
        // element at or after the first left element").  Chains are numbered by a wave scan of the breaks; the few
        // elements that carry such a contact flag their chain in a small LDS array, every element reads its chain's
        // flag back.  Ordered mode additionally needs "a left element at or before me in my chain": a segmented
        // prefix-OR in registers (lane-local over PW_ITEMS elements, then log-step shuffles across the lanes).
        u32 xq[PW_ITEMS], aq[PW_ITEMS], cid[PW_ITEMS];
        u32 nb_l = 0;
#pragma unroll
        for (int k = 0; k < (int)PW_ITEMS; k++) {
          xq[k] = (ctc[k] && (!ordered || src[k] == 1)) ? 1u : 0u;      // ctc: predecessor in the same chain, other operand
          aq[k] = (al[k] && src[k] == 0) ? 1u : 0u;
          nb_l += brk[k];
        }
        const u32 incl_brk = wave_scan_incl(nb_l);
        const u32 last_cid = rdl(incl_brk, 63);                          // chain 0 = the chain carried in from the previous tile
        {
          u32 c = incl_brk - nb_l;
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) { c += brk[k]; cid[k] = c; }
        }
        for (u32 xx = lane; xx <= last_cid; xx += PW_T) s_cm[xx] = 0;
        wsync();
#pragma unroll
        for (int k = 0; k < (int)PW_ITEMS; k++) if (xq[k]) s_cm[cid[k]] = 1;   // (same value from every writer)
        wsync();
        u32 sa[PW_ITEMS];                                                // ordered mode: a left element at or before me in my chain
#pragma unroll
        for (int k = 0; k < (int)PW_ITEMS; k++) sa[k] = 1;
        if (ordered) {
          u32 lane_brk = 0, acc = 0;
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) {
            if (brk[k]) { acc = 0; lane_brk = 1; }
            acc |= aq[k];
            sa[k] = acc;
          }
          u32 w = (lane_brk << 1) | acc;                                // lane summary: flag + OR behind the lane's last break
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const u32 wp = __shfl_up(w, d, DCD_WAVE);
            if (lane >= (u32)d) { if (!(w & 2u)) w |= wp & 1u; w |= wp & 2u; }
          }
          u32 inc = __shfl_up(w, 1, DCD_WAVE);
          inc = lane == 0 ? 0u : (inc & 1u);
          bool before = true;                                           // no break at or before the element inside this lane
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) {
            if (brk[k]) before = false;
            if (before) sa[k] |= inc;
            if (cid[k] == 0 && seenA) sa[k] = 1;
          }
        }
        const bool chain0_commit = committed || s_cm[0] != 0;
        if (!is_or && open && !committed && (last_cid > 0 || final_tile) && !chain0_commit) cursor = out_start;   // roll back
        u32 emit[PW_ITEMS], nemit = 0;
#pragma unroll
        for (int k = 0; k < (int)PW_ITEMS; k++) {
          emit[k] = 0;
          if (al[k] && !dup[k]) {
            if (is_or) emit[k] = 1;
            else {
              const bool cm = s_cm[cid[k]] != 0 || (cid[k] == 0 && committed);
              const bool closed = cid[k] < last_cid || final_tile;
              emit[k] = (sa[k] && (closed ? cm : true)) ? 1u : 0u;
            }
          }
          nemit += emit[k];
        }
        const u32 incl_emit = wave_scan_incl(nemit);
        const u32 total_emit = rdl(incl_emit, 63);
        u32 lfe1 = 0, la = 0;                         // last chain: 1 + emit offset of its first element, "holds a left element"
        {
          u32 e = incl_emit - nemit;
#pragma unroll
          for (int k = 0; k < (int)PW_ITEMS; k++) {
            if (brk[k] && cid[k] == last_cid) lfe1 = e + 1;
            if (cid[k] == last_cid) la |= aq[k];
            if (emit[k]) out[cursor + e] = val[k];
            e += emit[k];
          }
        }
        lfe1 = rfl(wave_max(lfe1));
        {
          const bool com = s_cm[last_cid] != 0 || (last_cid == 0 && committed);
          const bool hasA = (ballot64(la != 0) != 0) || (last_cid == 0 && seenA);
          if (last_cid != 0 || !open) out_start = cursor + (lfe1 ? lfe1 - 1 : 0u);
          open = 1; seenA = hasA ? 1u : 0u; committed = com ? 1u : 0u;
          cursor += total_emit;
        }
        PROF(9);   // general path
        if (ends_here) { open = 0; break; }
        advance();
        wsync();
      }
      count = cursor;
    }
  }
  if (lane == 0 && pass == 0) { part_count[wi] = count; part_src[wi] = slot | ((u64)oi << PSRC_OP_SHIFT); }
  PROF(10);
  wsync();                                                     // (the next item reuses the rings / records)
  w = rfl64(w_next);
  }
  PROF_END();
}

}  // namespace dcd
#include "dcd_pair_stream.h"
namespace dcd {

// out[...] <- scratch: one group of EIGHT LANES per work item (64 bytes per step), eight items per wavefront at a time.
// (Round 3.  It used to be 64 items per wavefront, copied eight at a time one group after the other: with ~100 values per item
// -- phrases over head terms -- that was ~100 dependent load -> store round trips per wavefront, 190 us for a 64 MB copy.)
__global__ __launch_bounds__(256) void k_pair_gather(const PairOp* __restrict__ ops, u32 n_ops, const u64* __restrict__ part_base, u64 W,
                                                    const u64* __restrict__ part_count, const u64* __restrict__ part_src,
                                                    const u64* __restrict__ excl, const u64* __restrict__ tmp, u64* __restrict__ pool,
                                                    const u64* __restrict__ res_off, const u32* __restrict__ overflow,
                                                    const u64* __restrict__ tmp_off, const u8* __restrict__ op_mode) {
  if (*overflow) return;                                   // (the slice's results do not fit the pool: nothing is copied, the host retries)
  const u32 lane = lane_id();
  const u64 gw = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((u64)gridDim.x * blockDim.x) >> 6;
  const u64 W_act = min(W, part_base[n_ops]);              // work items the device plan really used
  const u32 grp = lane >> 3, sub = lane & 7u;
  for (u64 base = gw * 8; base < W_act; base += nw * 8) {
    const u64 w = base + grp;
    const u64 c = w < W_act ? part_count[w] : 0;
    if (!c) continue;
    const u64 ps = part_src[w];                              // operator of the item and its first scratch slot
    const u32 lo = (u32)(ps >> PSRC_OP_SHIFT);
    if (op_mode[lo] == MODE_GALLOP2) continue;               // (written in place by the second pass of k_pair_parts)
    const u64* sp = tmp + tmp_off[lo] + (ps & PSRC_SLOT_MASK);
    u64* dp = pool + res_off[ops[lo].out_cnt_ref] + (excl[w] - excl[part_base[lo]]);
    u64 k = sub;
    for (; k + 24 < c; k += 32) {                            // four steps in flight
      const u64 v0 = sp[k], v1 = sp[k + 8], v2 = sp[k + 16], v3 = sp[k + 24];
      dp[k] = v0; dp[k + 8] = v1; dp[k + 16] = v2; dp[k + 24] = v3;
    }
    for (; k < c; k += 8) dp[k] = sp[k];
  }
}

// The plan's five exclusive scans (scratch need, work items of every kind) in ONE single-workgroup launch: a pass's operators
// are a few thousand, five scan launches cost more in launch latency than in work (round 4: 0.26 ms of small kernels sit
// between the heavy kernels of consecutive passes).  out[k][i] = sum of in[k][0 .. i), i <= n (the arrays hold n + 1 entries).
struct Scan5 { const u64* in64; const u32* in32[4]; u64* out[5]; };
__global__ __launch_bounds__(1024) void k_pair_scan5(Scan5 a, u64 n1) {
  // the five arrays go through the workgroup TOGETHER: their loads are in flight at the same time and one pair of barriers serves
  // all five scans of a round (one array after the other, the single workgroup waited for memory ten times: 48 us)
  __shared__ u64 lds[5][17];
  constexpr u64 PER = 8;
  const u32 lane = lane_id(), w = threadIdx.x >> 6;
  u64 carry[5] = {0, 0, 0, 0, 0};
  for (u64 b0 = 0; b0 < n1; b0 += 1024 * PER) {
    const u64 i0 = b0 + (u64)threadIdx.x * PER;
    u64 v[5][PER], incl[5], sum[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      sum[k] = 0;
#pragma unroll
      for (u64 j = 0; j < PER; j++) {
        u64 x = 0;
        if (i0 + j < n1) { if (k == 0) x = a.in64[i0 + j]; else if (a.in32[k - 1]) x = (u64)a.in32[k - 1][i0 + j]; }
        v[k][j] = x; sum[k] += x;
      }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) { incl[k] = wave_scan_incl(sum[k]); if (lane == DCD_WAVE - 1) lds[k][w] = incl[k]; }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const u64 t = lane < 16u ? lds[k][lane] : 0ull;
        const u64 ti = wave_scan_incl(t);
        if (lane < 16u) lds[k][lane] = ti - t;
        if (lane == 15u) lds[k][16] = ti;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 5; k++) {
      u64 ex = carry[k] + lds[k][w] + incl[k] - sum[k];
      if (k == 0 || a.in32[k - 1]) {
#pragma unroll
        for (u64 j = 0; j < PER; j++) { if (i0 + j < n1) a.out[k][i0 + j] = ex; ex += v[k][j]; }
      }
      carry[k] += lds[k][16];
    }
    __syncthreads();
  }
}
__global__ void k_pair_totals(const PairOp* __restrict__ ops, u32 n_ops, const u64* __restrict__ part_base, const u64* __restrict__ excl,
                              u64* __restrict__ counts, u64* __restrict__ slice_cnt) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_ops) { const u64 c = excl[part_base[i + 1]] - excl[part_base[i]]; counts[ops[i].out_cnt_ref] = c; slice_cnt[i] = c; }
}
// places the results of a slice in the pool: res_off[ref] = elements used before the slice + exclusive scan of the slice's counts.
// `used` is a pair of words used alternately by consecutive slices (this launch reads used[seq & 1] and writes the other one: the
// launches of a stream run one after the other, so no thread of this launch can read what another has written).
__global__ void k_pair_assign(const PairOp* __restrict__ ops, u32 n_ops, const u64* __restrict__ slice_off, const u64* __restrict__ slice_total,
                              u64* __restrict__ used, u32 seq, u64 pool_cap, u64* __restrict__ res_off, u32* __restrict__ overflow) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  const u64 u0 = used[seq & 1u];
  if (i < n_ops) res_off[ops[i].out_cnt_ref] = u0 + slice_off[i];
  if (i == 0) {
    const u64 u1 = u0 + *slice_total;
    used[(seq + 1u) & 1u] = u1;
    if (u1 > pool_cap) *overflow = 1u;
  }
}

// ------------------------------------------------------------------------------------------------
// PrepareSearchResult + ranks
// ------------------------------------------------------------------------------------------------
struct ResParams {
  const u64* page_end;   // [n_pages]
  const u32* page_doc;   // [n_pages] or null (doc == page)
  const u8* page_hdr;    // [n_pages] 1 = page id "0", or null
  const u8* doc_match;   // [n_docs] 1 = the document passes the request's -filter: (Search.cs:386-394), or null (no filter)
  u64 n_pages;
  u32 max_pages, max_docs;
  float doc_mult;
  u32 uniform_chars;     // != 0: page p ends at uniform_base + (p + 1) * uniform_chars
  u64 uniform_base;
};
struct QCounts { u64 n_pages, n_pos, n_docs; int32_t status; u32 pad; };

constexpr int RS_T = 256;
#ifndef DCD_RS_I
#define DCD_RS_I 8
#endif
constexpr int RS_I = DCD_RS_I;          // coordinates per thread
constexpr int RS_TILE = RS_T * RS_I;    // coordinates per workgroup iteration (round 3: 8 per thread, 4 / 8 / 16 -> 707 / 692 / 690 ms per 2.5 k configs[4] queries)

__device__ __forceinline__ u64 page_of(const ResParams& P, u64 c) {  // first page with end > c
  if (P.uniform_chars) {
    if (c < P.uniform_base + P.uniform_chars) return 0;   // everything below the first page end belongs to page 0
    u64 d = c - P.uniform_base;
    u64 p = (P.uniform_chars & (P.uniform_chars - 1u)) == 0u ? d >> (31 - __builtin_clz(P.uniform_chars)) : d / P.uniform_chars;
    return p < P.n_pages ? p : P.n_pages;
  }
  u64 a = 0, b = P.n_pages;
  while (a < b) { u64 m = (a + b) >> 1; if (P.page_end[m] > c) b = m; else a = m + 1; }
  return a;
}

// One workgroup per query, 1024 coordinates per iteration (4 consecutive ones per lane).
// FILL = false: counts only.  FILL = true: writes the CSR arrays.
// The walk stops after the coordinate that makes foundPages.Count > MAX_FOUND_PAGES (Search.cs:409):
// validity is a prefix property, so one scan (page starts) + one min-reduction (first coordinate past
// the cap) + one scan (document starts among the valid ones) per iteration is enough.
template <bool FILL>
__global__ __launch_bounds__(RS_T) void k_result(const FinalSeq* __restrict__ fin, u32 nq, IdxVals idx_values, Pool pool, ResParams P,
                                                QCounts* __restrict__ qc, const u64* __restrict__ page_base, const u64* __restrict__ pos_base,
                                                const u64* __restrict__ doc_base, u32* __restrict__ o_page_index, u64* __restrict__ o_pos_off,
                                                int32_t* __restrict__ o_pos, u32* __restrict__ o_doc_index, u32* __restrict__ o_doc_first_page) {
  __shared__ u32 s_scan[RS_T / 64 + 1];
  __shared__ u64 s_last[RS_T];          // page of each lane's last coordinate
  __shared__ u32 s_min[RS_T / 64];
  __shared__ u64 c_prev_page;
  __shared__ u32 c_pages, c_docs, c_err, c_done;
  __shared__ u64 c_pos;
  const u32 q = blockIdx.x, tid = threadIdx.x;
  if (q >= nq) return;
  const Seg sg = fin[q].s;
  if (*pool.overflow) {                                    // abandoned sub-batch: no result is read
    if (!FILL && tid == 0) { QCounts r; r.status = DCD_E_NOMEM; r.n_pages = 0; r.n_pos = 0; r.n_docs = 0; r.pad = 0; qc[q] = r; }
    return;
  }
  const CV V = seg_cv(sg, idx_values, pool);
  const u64 n = seg_cnt(sg, pool);
  if (FILL && qc[q].status != DCD_OK) return;   // the count pass found a coordinate beyond the last page: no output
  if (tid == 0) { c_prev_page = ~0ull; c_pages = 0; c_docs = 0; c_err = 0; c_done = 0; c_pos = 0; }
  __syncthreads();
  const u64 pb = FILL ? page_base[q] : 0, sb = FILL ? pos_base[q] : 0, db = FILL ? doc_base[q] : 0;
  for (u64 t0 = 0; t0 < n && !c_done; t0 += RS_TILE) {
    const u64 i0 = t0 + (u64)tid * RS_I;
    u64 c[RS_I], p[RS_I];
    u32 np_mine = 0;
#pragma unroll
    for (int k = 0; k < RS_I; k++) {
      bool in = i0 + k < n;
      c[k] = in ? V[i0 + k] : 0;
      p[k] = in ? page_of(P, c[k]) : ~0ull;
    }
    // last valid page of this lane (for the next lane's first comparison)
    {
      u64 lp = ~0ull;
#pragma unroll
      for (int k = 0; k < RS_I; k++) if (i0 + k < n) lp = p[k];
      s_last[tid] = lp;
    }
    __syncthreads();
    u64 pprev = tid == 0 ? c_prev_page : s_last[tid - 1];   // every earlier lane is full unless it is past n
    bool newp[RS_I];
#pragma unroll
    for (int k = 0; k < RS_I; k++) {
      bool in = i0 + k < n;
      newp[k] = in && (pprev == ~0ull || p[k] != pprev);
      if (in) pprev = p[k];
      np_mine += newp[k] ? 1u : 0u;
    }
    u32 tot_np;
    u32 ord0 = c_pages + block_scan_excl<u32, RS_T / 64>(np_mine, s_scan, &tot_np);   // pages started before this lane's items
    // page ordinal per item and the first item past the cap
    u32 ord[RS_I];
    u32 first_bad = 0xFFFFFFFFu;
    {
      u32 o = ord0;
#pragma unroll
      for (int k = 0; k < RS_I; k++) {
        bool in = i0 + k < n;
        o += newp[k] ? 1u : 0u;
        ord[k] = o - 1u;                                     // ordinal of the page this item belongs to
        bool valid = in && (ord[k] < P.max_pages || (ord[k] == P.max_pages && newp[k]));
        if (in && !valid && first_bad == 0xFFFFFFFFu) first_bad = tid * RS_I + k;
      }
    }
    // block minimum of first_bad
    {
      u32 m = first_bad;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { u32 t = __shfl_xor(m, d, 64); m = t < m ? t : m; }
      if (lane_id() == 0) s_min[tid >> 6] = m;
    }
    __syncthreads();
    const u32 cut = min(min(s_min[0], s_min[1]), min(s_min[2], s_min[3]));   // items [0, cut) of this tile are valid
    const u32 tile_n = (u32)min((u64)RS_TILE, n - t0);
    const u32 nvalid = min(cut, tile_n);
    u32 nd_mine = 0, vnp_mine = 0;
    bool newd[RS_I];
    u32 docp[RS_I];
    u64 prevp = tid == 0 ? c_prev_page : s_last[tid - 1];
#pragma unroll
    for (int k = 0; k < RS_I; k++) {
      const u32 li = tid * RS_I + k;
      const bool valid = li < nvalid;
      newd[k] = false;
      docp[k] = 0;
      if (valid && p[k] >= P.n_pages) atomicOr(&c_err, 1u);   // coordinate beyond the last page (reached)
      if (valid && newp[k] && p[k] < P.n_pages) {
        docp[k] = P.page_doc ? P.page_doc[p[k]] : (u32)p[k];
        u32 docprev = 0;
        bool hasprev = prevp != ~0ull && prevp < P.n_pages;
        if (hasprev) docprev = P.page_doc ? P.page_doc[prevp] : (u32)prevp;
        newd[k] = !hasprev || docp[k] != docprev;
        if (newd[k] && P.doc_match && !P.doc_match[docp[k]]) newd[k] = false;   // a filtered-out document is never added to foundDocs
      }
      if (valid && newp[k]) vnp_mine++;
      if (newd[k]) nd_mine++;
      if (i0 + k < n) prevp = p[k];
    }
    u32 tot_nd;
    u32 dord0 = c_docs + block_scan_excl<u32, RS_T / 64>(nd_mine, s_scan, &tot_nd);
    u32 tot_vnp;
    (void)block_scan_excl<u32, RS_T / 64>(vnp_mine, s_scan, &tot_vnp);
    if (FILL && !c_err) {
      u32 dord = dord0;
#pragma unroll
      for (int k = 0; k < RS_I; k++) {
        const u32 li = tid * RS_I + k;
        if (li < nvalid && p[k] < P.n_pages) {
          const u64 i = i0 + k;
          u64 base = p[k] == 0 ? 0ull : (P.uniform_chars ? P.uniform_base + p[k] * (u64)P.uniform_chars : P.page_end[p[k] - 1]);
          o_pos[sb + i] = (int32_t)(c[k] - base);
          if (newp[k]) { o_page_index[pb + ord[k]] = (u32)p[k]; o_pos_off[pb + ord[k]] = sb + i; }
          if (newd[k]) {                                 // foundDocs.Count < MAX_FOUND_DOCS, Search.cs:387
            if (dord < P.max_docs) { o_doc_index[db + dord] = docp[k]; o_doc_first_page[db + q + dord] = ord[k]; }
            dord++;
          }
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      c_pages += tot_vnp;
      c_docs += tot_nd;
      c_pos += nvalid;
      if (nvalid < tile_n || c_pages > P.max_pages) c_done = 1;
      c_prev_page = s_last[(tile_n - 1) / RS_I];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (!FILL) {
      QCounts r;
      r.status = c_err ? DCD_E_NOTFOUND : DCD_OK;
      r.n_pages = c_err ? 0 : c_pages;
      r.n_pos = c_err ? 0 : c_pos;
      r.n_docs = c_err ? 0 : min(c_docs, P.max_docs);
      r.pad = 0;
      qc[q] = r;
    }
  }
}

// page ranks (Search.cs:99-111): one thread per page
__global__ void k_page_rank(const u64* __restrict__ pos_off, const int32_t* __restrict__ pos, u64 n_pages_total, float* __restrict__ rank) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_pages_total) return;
  u64 a = pos_off[j], b = pos_off[j + 1];
  float bonus = 0;
  for (u64 k = a + 1; k < b; k++) {
    int d = pos[k] - pos[k - 1];
    bonus += (float)(30 / (d > 5 ? d : 5));
  }
  rank[j] = 1 + bonus + (float)log((double)(b - a));
}

// doc ranks (Search.cs:552-557) + ascending stable order (Search.cs:599): one workgroup per query
__global__ __launch_bounds__(256) void k_doc_rank(u32 nq, const QCounts* __restrict__ qc, const u64* __restrict__ page_base,
                                                 const u64* __restrict__ doc_base, const u32* __restrict__ doc_first_page,
                                                 const u32* __restrict__ page_index, const float* __restrict__ page_rank, const u8* __restrict__ page_hdr,
                                                 float mult, const u32* __restrict__ doc_index_in, u32* __restrict__ o_doc_index,
                                                 float* __restrict__ o_doc_rank, u32* __restrict__ o_doc_npages, const u32* __restrict__ page_doc) {
  __shared__ float s_rank[512];
  __shared__ u32 s_np[512];
  const u32 q = blockIdx.x;
  if (q >= nq) return;
  const u32 nd = (u32)qc[q].n_docs;
  const u64 pb = page_base[q], db = doc_base[q];
  const u32* fp = doc_first_page + db + q;
  {   // max_found_docs <= 512 is enforced on the host
    for (u32 d = threadIdx.x; d < nd; d += blockDim.x) {
      // the pages of a document are consecutive in foundPages: its run ends where the document of the page changes
      // (documents removed by the -filter: sit between the found ones, so the next FOUND document does not mark the end)
      const u32 a = fp[d], np_q = (u32)qc[q].n_pages, doc = doc_index_in[db + d];
      u32 b = a;
      double sum = 0;
      for (; b < np_q; b++) {
        const u32 pg = page_index[pb + b];
        if ((page_doc ? page_doc[pg] : pg) != doc) break;
        sum += (double)page_rank[pb + b];
      }
      float r = (float)sum;
      r = 1 + (float)log((double)r);
      if (page_hdr && b > a && page_hdr[page_index[pb + a]]) r *= mult;
      s_rank[d] = r;
      s_np[d] = b - a;
    }
    __syncthreads();
    for (u32 d = threadIdx.x; d < nd; d += blockDim.x) {
      float r = s_rank[d];
      u32 pos = 0;
      for (u32 e = 0; e < nd; e++) pos += (s_rank[e] < r || (s_rank[e] == r && e < d)) ? 1u : 0u;
      o_doc_index[db + pos] = doc_index_in[db + d];
      o_doc_rank[db + pos] = r;
      o_doc_npages[db + pos] = s_np[d];
    }
    __syncthreads();
  }
}

struct QcPagesF { const QCounts* q; __device__ u64 operator()(u64 i) const { return q[i].n_pages; } };
struct QcPosF { const QCounts* q; __device__ u64 operator()(u64 i) const { return q[i].n_pos; } };
struct QcDocsF { const QCounts* q; __device__ u64 operator()(u64 i) const { return q[i].n_docs; } };

}  // namespace dcd

using namespace dcd;

// ================================================================================================
// host side
// ================================================================================================
struct dcd_results {
  std::vector<int32_t> status;
  std::vector<u64> page_off, pos_off, doc_off, seq_off;
  std::vector<u32> page_index, doc_index, doc_n_pages;
  std::vector<float> page_rank, doc_rank;
  std::vector<int32_t> pos, seq_R;
  std::vector<u16> seq_units;
  float ms_device = 0;
  bool have_seq = false;
  bool downloaded = true;        // false: dcd_query_opts.no_download -- only status / page_off / doc_off and the totals are here
  u64 n_pages_total = 0, n_pos_total = 0, n_docs_total = 0;
  u64 ops_named = 0, ops_evaluated = 0, units_named = 0, units_evaluated = 0;   // common sub-expressions, see SubBatch
  u64 sub_batches = 0;
};

static int32_t ensure_values(dcd_index* h, float* ms_device = nullptr) {
  if (ms_device) *ms_device = 0.f;
  if (h->values_ready) return DCD_OK;
  DeviceIndex& ix = h->ix;
  const int dev = ix.device;
  hipStream_t st = h->st;
  StreamScope ss_(st);
  const u64 U = ix.n_units, N = ix.n_postings, V = ix.n_terms;
  // DOCODO_QUERY_PACKED=1: packed planes (5 bytes per posting instead of 8) when every coordinate fits 40 bits -- for hosts that
  // are short of device memory.  NOT the default: measured on the 10 M-doc index (20 k three-term AND queries), packed 25.0 ms
  // against 20.4 ms plain -- the operators are bound by instruction issue and load latency, not by the bytes of this array
  // (two loads + a shift / or per access instead of one load), so the smaller footprint buys nothing in time.
  static const bool env_packed = getenv("DOCODO_QUERY_PACKED") && atoi(getenv("DOCODO_QUERY_PACKED")) != 0;
  h->values_packed = env_packed && ix.max_coord < (1ull << 40);
  if (h->values_packed) { DCD_TRY(h->val_lo.alloc(dev, N ? N : 1)); DCD_TRY(h->val_hi.alloc(dev, N ? N : 1)); }
  else DCD_TRY(h->values.alloc(dev, N ? N : 1));
  CV all;
  all.v = h->values_packed ? nullptr : h->values.p; all.lo = h->values_packed ? h->val_lo.p : nullptr; all.hi = h->values_packed ? h->val_hi.p : nullptr;
  if (N) {
    // cell directories for the longer lists: about one cell per posting (cell width 2^shift)
    constexpr u64 DIR_MIN = 1024;
    constexpr int DIR_FINER = 0;
    const u32 coord_bits = log2_ceil(ix.max_coord + 2);
    h->h_dir_off.assign(V + 1, 0);
    h->h_dir_shift.assign(V, 255);
    u64 tot = 0;
    for (u64 t = 0; t < V; t++) {
      h->h_dir_off[t] = tot;
      const u64 cnt = ix.h_term_item_start[t + 1] - ix.h_term_item_start[t];
      if (cnt >= DIR_MIN) {
        u32 lg = 0;
        while ((2ull << lg) <= cnt) lg++;                     // floor(log2 cnt)
        static const int finer = getenv("DOCODO_QUERY_DIR_FINER") ? atoi(getenv("DOCODO_QUERY_DIR_FINER")) : DIR_FINER;   // (A/B knob)
        const int sh0 = coord_bits > lg ? (int)(coord_bits - lg) : 0;
        const u32 sh = (u32)std::min(std::max(sh0 - finer, 0), 62);
        h->h_dir_shift[t] = (u8)sh;
        tot += (ix.max_coord >> sh) + 2;                       // cells 0 .. max_coord >> sh, plus the end entry
      }
    }
    h->h_dir_off[V] = tot;
    if (tot >= (1ull << 40)) DCD_FAIL(DCD_E_OVERFLOW, "cell directories too large");
    DCD_TRY(h->dir.alloc(dev, tot ? tot : 1));
    DevBuf<u64> d_off;
    DevBuf<u8> d_sh;
    DCD_TRY(d_off.alloc(dev, V + 1));
    DCD_TRY(d_sh.alloc(dev, V ? V : 1));
    StagedUploads up;
    DCD_TRY(up.copy(d_off.p, h->h_dir_off.data(), (V + 1) * 8, st));
    DCD_TRY(up.copy(d_sh.p, h->h_dir_shift.data(), V, st));
    u64* pw = pinned_words();
    if (!pw) DCD_FAIL(DCD_E_NOMEM, "no pinned host memory");
    static const bool env_fused = !(getenv("DOCODO_QUERY_ADOPT_FUSED") && atoi(getenv("DOCODO_QUERY_ADOPT_FUSED")) == 0);   // (A/B knob)
    EvTimer tm;
    DCD_TRY(tm.init(st, dev));
    DCD_TRY(tm.start());
    if (env_fused && !h->values_packed) {
      // two passes over the units (see k_adopt_reduce / k_adopt_apply)
      const u64 nt = div_up(U, AD_TILE);
      DevBuf<u64> t_cnt, t_sum, t_cnt_ex, t_sum_ex, g0, sscr;
      DCD_TRY(t_cnt.alloc(dev, nt)); DCD_TRY(t_sum.alloc(dev, nt)); DCD_TRY(t_cnt_ex.alloc(dev, nt + 1)); DCD_TRY(t_sum_ex.alloc(dev, nt + 1));
      DCD_TRY(g0.alloc(dev, V ? V : 1));
      DCD_TRY(sscr.alloc(dev, scan_scratch_elems(nt)));
      hipLaunchKernelGGL(k_adopt_reduce, dim3((u32)nt), dim3(AD_T), 0, st, ix.units.p, U, t_cnt.p, t_sum.p);
      scan_exclusive<u64, u64>(t_cnt.p, t_cnt_ex.p, nt, sscr.p, st);
      DCD_HIP(hipMemcpyAsync(&pw[8], sscr.p + scan_nblocks(nt), 8, hipMemcpyDeviceToHost, st));
      scan_exclusive<u64, u64>(t_sum.p, t_sum_ex.p, nt, sscr.p, st);
      hipLaunchKernelGGL(k_adopt_terms, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ix.units.p, U, ix.term_unit_start.p, V, t_sum_ex.p, g0.p);
      // (queued before the check of the count comes back: a damaged image overruns nothing -- k_adopt_apply addresses `values`
      // by the scanned tile counts, so the check below must come first)
      DCD_HIP(hipStreamSynchronize(st));
      if (pw[8] != N) DCD_FAIL(DCD_E_BADARG, "index is inconsistent: %llu postings in the unit stream, %llu in the term directory",
                               (unsigned long long)pw[8], (unsigned long long)N);
      hipLaunchKernelGGL(k_adopt_apply, dim3((u32)nt), dim3(AD_T), 0, st, ix.units.p, U, t_cnt_ex.p, t_sum_ex.p, ix.term_item_start.p, V, g0.p, d_off.p,
                         d_sh.p, h->values.p, h->dir.p);
    } else {
      DevBuf<u64> part, deltas, excl, sscr;
      u64 nb = div_up(U, CODEC_TILE);
      DCD_TRY(part.alloc(dev, nb + 1));
      DCD_TRY(deltas.alloc(dev, N));
      DCD_TRY(excl.alloc(dev, N));
      DCD_TRY(sscr.alloc(dev, scan_scratch_elems(N)));
      decode_count_launch(ix.units.p, U, part.p, st);
      {   // the posting counts of the term directory must match the unit stream (a damaged image would overrun below)
        DCD_HIP(hipMemcpyAsync(&pw[8], part.p + nb, 8, hipMemcpyDeviceToHost, st));
        DCD_HIP(hipStreamSynchronize(st));
        if (pw[8] != N) DCD_FAIL(DCD_E_BADARG, "index is inconsistent: %llu postings in the unit stream, %llu in the term directory",
                                 (unsigned long long)pw[8], (unsigned long long)N);
      }
      hipLaunchKernelGGL(k_dec_deltas, dim3((u32)nb), dim3(CODEC_THREADS), 0, st, ix.units.p, U, part.p, deltas.p);
      scan_exclusive<u64, u64>(deltas.p, excl.p, N, sscr.p, st);
      hipLaunchKernelGGL(k_term_relative, dim3((u32)div_up(N, 256)), dim3(256), 0, st, excl.p, deltas.p, N, ix.term_item_start.p, V,
                         (u64*)all.v, (u32*)all.lo, (u8*)all.hi);
      if (tot) hipLaunchKernelGGL(k_build_dir, dim3((u32)div_up(N, 256)), dim3(256), 0, st, all, N, ix.term_item_start.p, V, d_off.p, d_sh.p, h->dir.p);
    }
    DCD_TRY(tm.stop());
    DCD_HIP(hipStreamSynchronize(st));
    DCD_HIP(hipGetLastError());
    if (ms_device) *ms_device = tm.ms();
  }
  h->values_ready = true;
  return DCD_OK;
}

// ================================================================================================
// planning (host only, no HIP call: a chunk is planned on a helper thread while the previous one runs on the device)
// ================================================================================================
// Symbolic value on a query's evaluation stack.
struct SymVal { Seg s; int32_t R; u64 cap; u32 round; u64 key; };

// Round 4: cross-query COMMON SUB-EXPRESSIONS.  Two operators with the same operands, the same R on both sides and the same
// opcode compute the same sequence (IndexSequence.cs:218-322 is a pure function of those), so a sub-batch evaluates every distinct
// operator ONCE and every query that contains it refers to the one result.  Under a Zipf query load this is most of the work: 100 k
// phrases sampled from corpus positions name the pair of the two most frequent terms 1 800 times (p1 * p1 * 100 k), each
// evaluation a streaming merge of 3.7e8 postings.  Key of an operand: the term ordinal, or the number of the operator result.
// State of a SUB-BATCH: the scope of the work pool and of the common sub-expressions.  Its queries are planned and run in CHUNKS
// (a chunk's plan is made while the previous chunk runs); operator results stay in the pool until the sub-batch ends, so a later
// chunk refers to the results of an earlier one.
struct SubBatch {
  CseTable* cse = nullptr;          // the handle's table (dcd_index_int.h), emptied for this sub-batch
  bool use_cse = true;
  std::vector<u64> ref_cap;      // [n_refs] bound of the result length
  std::vector<u32> ref_chunk;    // [n_refs] chunk that evaluates it
  std::vector<u32> ref_round;    // [n_refs] round inside that chunk
  u32 n_refs = 0;
  // statistics: operators named by the programs / of those found in the table; ushort units of their index-term operands
  // as named (every query's operands: the SURVEY 8(d) bytes) and as evaluated (every distinct operator once)
  u64 ops_total = 0, ops_shared = 0, units_named = 0, units_evaluated = 0;
};

// A ROUND = the new operators of a chunk whose operands are all known once the previous round has run.  How many of a round's
// operators run together is decided ON THE DEVICE, pass by pass, from their REAL operand lengths (k_pair_plan / k_pair_admit):
// the host only knows capacity bounds (a result is at most as long as its operands together), which overstate a phrase match by
// orders of magnitude.
struct RoundInfo { u32 first, n; };
struct ChunkPlan {
  size_t q0 = 0, q1 = 0;
  u32 chunk = 0;
  PairOp* ops = nullptr;         // pinned staging owned by the executor: [ops_cap]
  FinalSeq* finals = nullptr;    // pinned staging: [q1 - q0]
  size_t ops_cap = 0, n_ops = 0;
  std::vector<int32_t> finalR;
  std::vector<RoundInfo> rounds;
  std::vector<u64> cap_prefix, w_prefix;   // [n_ops + 1] prefix sums of the operators' capacity bounds / work-item bounds
  u64 max_op_W = 0;              // largest work-item bound of a single operator
  u32 max_round_ops = 0;
  int32_t status = DCD_OK;
  std::string err;
};

static void plan_fail(ChunkPlan& P, int32_t code, const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  P.status = code;
  P.err = buf;
}

// queries [P.q0, P.q1) -> the operators that are new to the sub-batch (in execution order, round by round) + the final sequences
static void plan_chunk(const dcd_index* h, const std::vector<SymVal>* ext_leaves, SubBatch& sb, const int32_t* prog, const u64* prog_off,
                       u64 PART, ChunkPlan& P) {
  const DeviceIndex& ix = h->ix;
  struct Ord { u32 round; u32 idx; u64 key; };
  std::vector<PairOp> ops;
  std::vector<Ord> ord;
  std::vector<SymVal> stk;
  P.finalR.assign(P.q1 - P.q0, 0);
  P.rounds.clear();
  P.n_ops = 0; P.max_op_W = 0; P.max_round_ops = 0;
  const bool have_dir = !h->h_dir_off.empty();
  for (size_t q = P.q0; q < P.q1; q++) {
    stk.clear();
    for (u64 i = prog_off[q]; i < prog_off[q + 1];) {
      const int32_t opc = prog[i];
      if (opc == DCD_OP_TERM) {
        if (i + 3 > prog_off[q + 1]) return plan_fail(P, DCD_E_BADARG, "query %zu: truncated TERM", q);
        const int64_t t = prog[i + 1];
        SymVal v;
        if (ext_leaves) {
          if (t < 0 || (size_t)t >= ext_leaves->size()) return plan_fail(P, DCD_E_BADARG, "query %zu: leaf %lld out of range", q, (long long)t);
          v = (*ext_leaves)[(size_t)t];
        } else {
          if (t < 0 || (u64)t >= ix.n_terms) return plan_fail(P, DCD_E_BADARG, "query %zu: term %lld out of range", q, (long long)t);
          v.s.kind = 0; v.s.off = ix.h_term_item_start[t]; v.s.cnt_ref = NONE32;
          v.s.dir_off = have_dir ? h->h_dir_off[t] : 0;
          v.s.dir_shift = (!have_dir || h->h_dir_shift[t] == 255) ? NONE32 : (u32)h->h_dir_shift[t];
          v.s.pad = 0;
          v.s.cnt_imm = ix.h_term_item_start[t + 1] - ix.h_term_item_start[t];
          v.cap = v.s.cnt_imm;
        }
        v.round = 0;
        v.key = (u64)t;
        v.R = prog[i + 2];
        stk.push_back(v);
        i += 3;
      } else if (opc == DCD_OP_AND || opc == DCD_OP_OR) {
        if (stk.size() < 2) return plan_fail(P, DCD_E_BADARG, "query %zu: operator without two operands", q);
        const SymVal b = stk.back(); stk.pop_back();
        const SymVal a = stk.back(); stk.pop_back();
        sb.ops_total++;
        u64 op_units = 0;
        if (a.s.kind == 0) op_units += ext_leaves ? a.cap : ix.h_term_unit_start[a.key + 1] - ix.h_term_unit_start[a.key];
        if (b.s.kind == 0) op_units += ext_leaves ? b.cap : ix.h_term_unit_start[b.key + 1] - ix.h_term_unit_start[b.key];
        sb.units_named += op_units;
        CseTable::Ent* e = sb.use_cse ? &sb.cse->slot(a.key, b.key, a.R, b.R, (u32)opc) : nullptr;
        SymVal r;
        u32 ref;
        if (e && e->ref != NONE32) {                       // evaluated already (this chunk or an earlier one of the sub-batch)
          ref = e->ref;
          r.round = sb.ref_chunk[ref] == P.chunk ? sb.ref_round[ref] : 0u;
          sb.ops_shared++;
        } else {
          ref = sb.n_refs++;
          if (e) e->ref = ref;
          sb.units_evaluated += op_units;
          r.round = std::max(a.round, b.round) + 1;
          sb.ref_cap.push_back(a.cap + b.cap);
          sb.ref_chunk.push_back(P.chunk);
          sb.ref_round.push_back(r.round);
          PairOp p;
          p.a = a.s; p.b = b.s; p.Ra = a.R; p.Rb = b.R; p.op = (u32)opc;
          p.out_cnt_ref = ref; p.cap = a.cap + b.cap; p.pad2 = 0;
          // operators of a round run in the order of their LONGER index-term operand, so that the queries of a batch that probe the
          // same posting list (the head terms of a Zipf query load) do so at about the same time and share its cache lines
          const bool ta = a.s.kind == 0, tb = b.s.kind == 0;
          u64 key;
          if (ta && tb) key = a.s.cnt_imm >= b.s.cnt_imm ? a.s.off : b.s.off;
          else key = ta ? a.s.off : (tb ? b.s.off : ~0ull);
          ord.push_back(Ord{r.round, (u32)ops.size(), key});
          ops.push_back(p);
        }
        r.cap = sb.ref_cap[ref];
        r.s.kind = 1; r.s.off = 0; r.s.cnt_ref = ref; r.s.cnt_imm = 0;
        r.s.dir_off = 0; r.s.dir_shift = NONE32; r.s.pad = 0;
        const int absR = std::max(std::abs(a.R), std::abs(b.R));
        r.R = (a.R < 0 && b.R < 0) ? -absR : absR;                       // IndexSequence.cs:221-222, 288-289
        r.key = (1ull << 63) | ref;
        stk.push_back(r);
        i += 1;
      } else if (opc == DCD_OP_SETR) {
        if (i + 2 > prog_off[q + 1] || stk.empty()) return plan_fail(P, DCD_E_BADARG, "query %zu: bad SETR", q);
        stk.back().R = prog[i + 1];
        i += 2;
      } else {
        return plan_fail(P, DCD_E_BADARG, "query %zu: unknown opcode %d", q, opc);
      }
    }
    if (stk.size() != 1) return plan_fail(P, DCD_E_BADARG, "query %zu: program leaves %zu values on the stack", q, stk.size());
    P.finals[q - P.q0].s = stk[0].s;
    P.finalR[q - P.q0] = stk[0].R;
  }
  if (ops.size() > P.ops_cap) return plan_fail(P, DCD_E_BADARG, "query plan: %zu operators, %zu expected", ops.size(), P.ops_cap);
  static const bool sort_ops = !(getenv("DOCODO_QUERY_SORT_OPS") && atoi(getenv("DOCODO_QUERY_SORT_OPS")) == 0);
  if (sort_ops && ord.size() >= 4096) {
    // (round, key, idx) ascending = a STABLE sort by (round, key) of the operators in the order they were named: LSD radix over
    // 16-bit digits of key (posting offsets: below 2^48; "no index term" = all ones sorts last) and one pass over the round.
    // std::sort took a third of the planning time of a 16 k-query chunk (2.5 of 8 ms), and planning is what a batch on a small
    // index waits for.
    std::vector<Ord> tmp(ord.size());
    u64 kmax = 0;
    u32 rmax = 0;
    for (Ord& x : ord) { if (x.key > 0xFFFFFFFFFFFFull) x.key = 0xFFFFFFFFFFFFull; kmax = std::max(kmax, x.key); rmax = std::max(rmax, x.round); }
    std::vector<u32> cnt(65536 + 1);
    auto pass = [&](auto digit_of, u32 n_digits) {
      std::fill(cnt.begin(), cnt.begin() + n_digits + 1, 0u);
      for (const Ord& x : ord) cnt[digit_of(x) + 1]++;
      for (u32 d = 0; d < n_digits; d++) cnt[d + 1] += cnt[d];
      for (const Ord& x : ord) tmp[cnt[digit_of(x)]++] = x;
      ord.swap(tmp);
    };
    for (int sh = 0; sh < 48 && (kmax >> sh) != 0; sh += 16) pass([sh](const Ord& x) { return (u32)((x.key >> sh) & 0xFFFFu); }, 65536u);
    if (rmax >= 65536u) std::stable_sort(ord.begin(), ord.end(), [](const Ord& x, const Ord& y) { return x.round < y.round; });
    else if (rmax > 0) pass([](const Ord& x) { return x.round; }, rmax + 1);
  } else if (sort_ops) std::sort(ord.begin(), ord.end(), [](const Ord& x, const Ord& y) {
    return x.round != y.round ? x.round < y.round : (x.key != y.key ? x.key < y.key : x.idx < y.idx); });
  else std::stable_sort(ord.begin(), ord.end(), [](const Ord& x, const Ord& y) { return x.round < y.round; });
  // work items of an operator <= ceil(cap / PART) partitions of the union, or ceil(|short| / GAL_PER) when it gallops
  P.cap_prefix.assign(ord.size() + 1, 0);
  P.w_prefix.assign(ord.size() + 1, 0);
  for (size_t k = 0; k < ord.size(); k++) {
    const PairOp& p = ops[ord[k].idx];
    if (P.rounds.empty() || ord[k].round != ord[k - 1].round) P.rounds.push_back(RoundInfo{(u32)k, 0});
    P.rounds.back().n++;
    const u64 ca = p.a.kind == 0 ? p.a.cnt_imm : sb.ref_cap[p.a.cnt_ref];
    const u64 cb = p.b.kind == 0 ? p.b.cnt_imm : sb.ref_cap[p.b.cnt_ref];
    const u64 W = std::max<u64>(std::max<u64>(1, div_up(ca + cb, PART)), div_up(std::min(ca, cb), GAL_PER));
    if (W >= (1ull << 31)) return plan_fail(P, DCD_E_OVERFLOW, "operator too large: %llu work items", (unsigned long long)W);
    P.max_op_W = std::max(P.max_op_W, W);
    P.cap_prefix[k + 1] = P.cap_prefix[k] + p.cap;
    P.w_prefix[k + 1] = P.w_prefix[k] + W;
    P.ops[k] = p;
  }
  for (const RoundInfo& r : P.rounds) P.max_round_ops = std::max(P.max_round_ops, r.n);
  P.n_ops = ord.size();
}

// ================================================================================================
// execution of a planned chunk
// ================================================================================================
// Device state of a sub-batch: the pool's bump pointer, the result directory, the page metadata.
struct PageDev {
  DevBuf<u32> page_doc;
  DevBuf<u8> page_hdr;
  DevBuf<u8> doc_match;
};
struct SubBatchDev {
  DevBuf<u64> counts, res_off;     // [ops of the sub-batch]
  DevBuf<u64> used;                // [2] pool elements in use (alternating, see k_pair_assign) + [1] scratch
  DevBuf<u32> overflow;            // [1]
  u64* pool = nullptr;             // pool_cap elements
  u64* tmp = nullptr;              // scratch_cap elements
  u64 pool_cap = 0, scratch_cap = 0;
  u32 seq = 0;                     // passes executed so far
  u64 n_admit_syncs = 0;           // (statistics: passes whose size the device had to tell)
  PageDev pd;
  ResParams P;
};

static int32_t make_res_params(dcd_index* h, const dcd_query_opts& o, PageDev& pd, ResParams* P) {
  DeviceIndex& ix = h->ix;
  const PageTable& pt = ix.pages;
  const u64 np = pt.n_pages();
  P->n_pages = np;
  P->max_pages = (u32)o.max_found_pages;
  P->max_docs = (u32)o.max_found_docs;
  P->doc_mult = o.doc_rank_multiply;
  P->page_doc = nullptr;
  P->page_hdr = nullptr;
  P->doc_match = nullptr;
  P->uniform_chars = pt.uniform ? (u32)pt.uni_chars : 0u;
  P->uniform_base = pt.uniform ? pt.uni_base : 0ull;
  if (np && !ix.page_end.p) {
    std::vector<u64> e(np);
    for (u64 p = 0; p < np; p++) e[p] = pt.end_at(p);
    DCD_TRY(ix.page_end.alloc(ix.device, np));
    DCD_HIP(hipMemcpy(ix.page_end.p, e.data(), np * 8, hipMemcpyHostToDevice));
  }
  P->page_end = ix.page_end.p;
  if (!pt.doc_per_page && np) {
    DCD_TRY(pd.page_doc.alloc(ix.device, np));
    DCD_HIP(hipMemcpy(pd.page_doc.p, pt.page_doc.data(), np * 4, hipMemcpyHostToDevice));
    P->page_doc = pd.page_doc.p;
  }
  if (!pt.ids.empty()) {
    std::vector<u8> hdr(np, 0);
    bool any = false;
    for (u64 p = 0; p < np; p++) if (pt.ids[p] == u"0") { hdr[p] = 1; any = true; }
    if (any) {
      DCD_TRY(pd.page_hdr.alloc(ix.device, np));
      DCD_HIP(hipMemcpy(pd.page_hdr.p, hdr.data(), np, hipMemcpyHostToDevice));
      P->page_hdr = pd.page_hdr.p;
    }
  }
  if (o.doc_match) {
    if (o.n_doc_match != pt.n_docs()) DCD_FAIL(DCD_E_BADARG, "doc_match has %llu entries, the index has %llu documents",
                                               (unsigned long long)o.n_doc_match, (unsigned long long)pt.n_docs());
    DCD_TRY(pd.doc_match.alloc(ix.device, o.n_doc_match ? o.n_doc_match : 1));
    if (o.n_doc_match) DCD_HIP(hipMemcpy(pd.doc_match.p, o.doc_match, o.n_doc_match, hipMemcpyHostToDevice));
    P->doc_match = pd.doc_match.p;
  }
  return DCD_OK;
}

// wavefronts of k_pair_parts: a few per wavefront slot of the device (CUs x 4 SIMDs x DCD_PW_WAVES), so that the hardware still
// balances the tail (per device: a process may drive several)
static u64 pw_grid_of(int dev) {
  static const u64 env = [] { const char* e = getenv("DOCODO_QUERY_PW_GRID"); return (e && atoll(e) > 0) ? (u64)atoll(e) : 0ull; }();
  if (env) return env;
  static std::mutex mu;
  static std::vector<u64> per_dev;
  std::lock_guard<std::mutex> g(mu);
  if ((size_t)dev >= per_dev.size()) per_dev.resize(dev + 1, 0);
  if (!per_dev[dev]) {
    hipDeviceProp_t pr;
    const int cus = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256;
    per_dev[dev] = (u64)cus * 4u * DCD_PW_WAVES * 4u;
  }
  return per_dev[dev];
}

struct ResCarve {        // consecutive 256-byte aligned arrays out of one block
  u8* base;
  size_t cap, at = 0;
  template <typename T> T* take(size_t n) { at = (at + 255) & ~(size_t)255; T* p = reinterpret_cast<T*>(base + at); at += n * sizeof(T); return p; }
};
struct PinnedBlock {     // pinned staging returned to the cache when the scope ends
  void* p = nullptr;
  PinnedBlock() {}
  PinnedBlock(const PinnedBlock&) = delete;
  PinnedBlock& operator=(const PinnedBlock&) = delete;
  ~PinnedBlock() { if (p) pinned_stage_release(p); }
  int32_t get(size_t bytes) {
    if (p) pinned_stage_release(p);
    p = pinned_stage_acquire(bytes ? bytes : 1);
    return p ? DCD_OK : DCD_E_NOMEM;
  }
};

static int32_t exec_chunk(dcd_index* h, const dcd_query_opts& o, SubBatchDev& D, const ChunkPlan& C, u64 PART,
                          dcd_results* R, const u64* ext_values, u32 n_refs_after) {
  DeviceIndex& ix = h->ix;
  const int dev = ix.device;
  hipStream_t st = h->st;
  const size_t nq = C.q1 - C.q0;
  IdxVals idx_values;
  idx_values.v64 = ext_values ? ext_values : (h->values_packed ? nullptr : h->values.p);
  idx_values.lo = (!ext_values && h->values_packed) ? h->val_lo.p : nullptr;
  idx_values.hi = (!ext_values && h->values_packed) ? h->val_hi.p : nullptr;
  const Pool pool{D.pool, D.res_off.p, D.counts.p, D.overflow.p};

  // ---- device buffers of the chunk (from the library's device cache) --------------------------------
  DevBuf<PairOp> d_ops;
  DevBuf<u64> d_part_base, d_part_count, d_part_src, d_part_excl, d_part_scr, d_slice_cnt, d_slice_off, d_slice_scr, d_work_ctr;
  DevBuf<u64> d_need, d_need_off, d_base_old, d_base_s32, d_base_g2, d_nadm;
  DevBuf<u32> d_items, d_items_old, d_items_s32, d_items_g2;
  DevBuf<u8> d_mode;
  // work items a pass may hold (its three per-item arrays): as many as the largest round can need, at most ~4 M, at least the
  // largest single operator
  constexpr u64 W_PASS = 4ull << 20;
  u64 W_cap = 1;
  for (const RoundInfo& r : C.rounds) W_cap = std::max(W_cap, C.w_prefix[r.first + r.n] - C.w_prefix[r.first]);
  W_cap = std::max(std::min(W_cap, W_PASS), C.max_op_W);
  const size_t mro = (size_t)C.max_round_ops + 1;
  DCD_TRY(d_ops.alloc(dev, C.n_ops ? C.n_ops : 1));
  DCD_TRY(d_need.alloc(dev, mro)); DCD_TRY(d_need_off.alloc(dev, mro));
  DCD_TRY(d_items.alloc(dev, mro)); DCD_TRY(d_items_old.alloc(dev, mro)); DCD_TRY(d_items_s32.alloc(dev, mro)); DCD_TRY(d_items_g2.alloc(dev, mro));
  DCD_TRY(d_base_old.alloc(dev, mro)); DCD_TRY(d_base_s32.alloc(dev, mro)); DCD_TRY(d_base_g2.alloc(dev, mro));
  DCD_TRY(d_mode.alloc(dev, mro));
  DCD_TRY(d_part_base.alloc(dev, mro));
  DCD_TRY(d_slice_cnt.alloc(dev, mro));
  DCD_TRY(d_slice_off.alloc(dev, mro));
  DCD_TRY(d_slice_scr.alloc(dev, scan_scratch_elems((u64)mro)));
  DCD_TRY(d_part_count.alloc(dev, W_cap + 1));
  DCD_TRY(d_part_src.alloc(dev, W_cap + 1));
  DCD_TRY(d_part_excl.alloc(dev, W_cap + 1));
  DCD_TRY(d_part_scr.alloc(dev, scan_scratch_elems(std::max<u64>(W_cap + 1, mro))));
  DCD_TRY(d_work_ctr.alloc(dev, 4));
  DCD_TRY(d_nadm.alloc(dev, 1));
  DevBuf<FinalSeq> d_fin;
  DCD_TRY(d_fin.alloc(dev, nq ? nq : 1));
  DevBuf<QCounts> d_qc;
  DevBuf<u64> d_pbase, d_sbase, d_dbase, sscr;
  DCD_TRY(d_qc.alloc(dev, nq ? nq : 1));
  DCD_TRY(d_pbase.alloc(dev, nq + 1)); DCD_TRY(d_sbase.alloc(dev, nq + 1)); DCD_TRY(d_dbase.alloc(dev, nq + 1));
  DCD_TRY(sscr.alloc(dev, 3 * scan_scratch_elems(nq ? nq : 1)));
  // the plan was written into pinned staging by the planner: two asynchronous uploads
  if (C.n_ops) DCD_HIP(hipMemcpyAsync(d_ops.p, C.ops, C.n_ops * sizeof(PairOp), hipMemcpyHostToDevice, st));
  if (nq) DCD_HIP(hipMemcpyAsync(d_fin.p, C.finals, nq * sizeof(FinalSeq), hipMemcpyHostToDevice, st));
  // k_pair_stream (32-bit tile-relative keys) takes the streaming work items unless the resident coordinates are packed planes
  static const bool env_s32 = !(getenv("DOCODO_QUERY_STREAM32") && atoi(getenv("DOCODO_QUERY_STREAM32")) == 0);   // (A/B knob)
  const bool stream32 = env_s32 && idx_values.v64 != nullptr && !C.rounds.empty();
  struct SideStream {                                       // second stream + the two events that order it against the main one
    int dev; hipStream_t s = nullptr; hipEvent_t a = nullptr, b = nullptr;
    ~SideStream() { if (s) stream_release(dev, s); if (a) event_release(dev, a); if (b) event_release(dev, b); }
  } side{dev};
  hipStream_t st2 = nullptr;
  hipEvent_t ev_plan = nullptr, ev_s32 = nullptr;
  if (stream32) {
    DCD_TRY(stream_acquire(dev, &side.s));
    side.a = event_acquire(dev); side.b = event_acquire(dev);
    if (!side.a || !side.b) DCD_FAIL(DCD_E_HIP, "hipEventCreate failed");
    st2 = side.s; ev_plan = side.a; ev_s32 = side.b;
  }

  EvTimer tm;
  DCD_TRY(tm.init(st, dev));
  DCD_TRY(tm.start());
  const u64 pw_grid = pw_grid_of(dev);
  static const u64 sk_env = [] { const char* e = getenv("DOCODO_QUERY_SK_GRID"); return (e && atoll(e) > 0) ? (u64)atoll(e) : 0ull; }();
  const u64 sk_grid = sk_env ? sk_env : std::max<u64>(1, pw_grid * DCD_SK_WAVES / DCD_PW_WAVES);       // (the same multiple of ITS wavefront slots)
  static const u32 gal_ratio = getenv("DOCODO_QUERY_GAL_RATIO") ? (u32)atoi(getenv("DOCODO_QUERY_GAL_RATIO")) : GAL_RATIO;   // (A/B knob)
  static const u32 g2_ratio = getenv("DOCODO_QUERY_G2_RATIO") ? (u32)atoi(getenv("DOCODO_QUERY_G2_RATIO")) : (u32)DCD_GAL2_RATIO;   // (A/B knobs of the two-pass rule)
  static const u64 g2_div = getenv("DOCODO_QUERY_G2_NEED_DIV") ? std::max<u64>(1, strtoull(getenv("DOCODO_QUERY_G2_NEED_DIV"), nullptr, 10)) : 128;
  u64* pwa = pinned_words();
  if (!pwa) DCD_FAIL(DCD_E_NOMEM, "no pinned host memory");
  const PlanOut plan_out{d_need.p, d_items.p, d_items_old.p, d_items_s32.p, d_items_g2.p, d_mode.p};
  for (const RoundInfo& rd : C.rounds) {
    u32 first = rd.first;
    const u32 end = rd.first + rd.n;
    while (first < end) {
      const u32 n_rem = end - first;
      const PairOp* ops = d_ops.p + first;
      // device plan of everything that is left of the round: mode, work items and REAL scratch need per operator, prefix sums
      hipLaunchKernelGGL(k_pair_plan, dim3((u32)div_up((u64)n_rem + 1, 256)), dim3(256), 0, st, ops, n_rem, idx_values, pool, PART, plan_out, gal_ratio,
                         stream32 ? 1u : 0u, std::max<u64>(D.scratch_cap / g2_div, 1), g2_ratio);
      if ((u64)n_rem + 1 <= 64 * 1024) {
        Scan5 s5;
        s5.in64 = d_need.p; s5.out[0] = d_need_off.p;
        s5.in32[0] = d_items.p; s5.out[1] = d_part_base.p;
        s5.in32[1] = d_items_old.p; s5.out[2] = d_base_old.p;
        s5.in32[2] = d_items_g2.p; s5.out[3] = d_base_g2.p;
        s5.in32[3] = stream32 ? d_items_s32.p : nullptr; s5.out[4] = stream32 ? d_base_s32.p : nullptr;
        hipLaunchKernelGGL(k_pair_scan5, dim3(1), dim3(1024), 0, st, s5, (u64)n_rem + 1);
      } else {
        scan_exclusive<u64, u64>(d_need.p, d_need_off.p, (u64)n_rem + 1, d_part_scr.p, st);
        scan_exclusive<u32, u64>(d_items.p, d_part_base.p, (u64)n_rem + 1, d_part_scr.p, st);
        scan_exclusive<u32, u64>(d_items_old.p, d_base_old.p, (u64)n_rem + 1, d_part_scr.p, st);
        scan_exclusive<u32, u64>(d_items_g2.p, d_base_g2.p, (u64)n_rem + 1, d_part_scr.p, st);
        if (stream32) scan_exclusive<u32, u64>(d_items_s32.p, d_base_s32.p, (u64)n_rem + 1, d_part_scr.p, st);
      }
      DCD_HIP(hipMemsetAsync(d_work_ctr.p, 0, 4 * sizeof(u64), st));
      // how many of them run in this pass.  The host knows capacity BOUNDS: the longest prefix whose bounds fit is always safe, and
      // when it is long (ordinary conjunctions: the bound of a term-term operator is its real size) it is taken as it is, with no
      // round trip.  When it is short -- a few operators over head terms, whose bounds are the length of the long list while a
      // galloping evaluation needs a fraction of it or nothing -- the device says how many really fit (one round trip).
      u32 n;
      {
        u32 lo = first, hi = end;                            // largest m in (first, end] with cap / item bounds of [first, m) fitting
        while (lo < hi) {
          const u32 m = lo + (hi - lo + 1) / 2;
          if (C.cap_prefix[m] - C.cap_prefix[first] <= D.scratch_cap && C.w_prefix[m] - C.w_prefix[first] <= W_cap) lo = m; else hi = m - 1;
        }
        n = std::max<u32>(lo - first, 1);
      }
      // (round 4, second half: ALWAYS when the bounds do not admit everything that is left -- the second operator of a conjunction has a
      //  result as an operand, whose bound is the length of its operands: 16 k of them fell into four passes where one does;
      //  DOCODO_QUERY_ADMIT_BELOW = 256 is the rule of the first half: configs[3] 81.6 against 79.4 ms)
      static const u32 admit_below = getenv("DOCODO_QUERY_ADMIT_BELOW") ? (u32)atoll(getenv("DOCODO_QUERY_ADMIT_BELOW")) : 0xFFFFFFFFu;
      if (n < n_rem && n < admit_below) {
        hipLaunchKernelGGL(k_pair_admit, dim3(1), dim3(64), 0, st, n_rem, d_need_off.p, d_part_base.p, D.scratch_cap, W_cap, d_nadm.p);
        DCD_HIP(hipMemcpyAsync(&pwa[9], d_nadm.p, 8, hipMemcpyDeviceToHost, st));
        DCD_HIP(hipStreamSynchronize(st));
        n = std::max(n, (u32)std::min<u64>(std::max<u64>(pwa[9], 1), n_rem));
        D.n_admit_syncs++;
      }
      n = std::min<u32>(n, PSRC_MAX_OPS - 1u);               // (the operator's number travels in 28 bits of part_src)
      const u64 W = std::min<u64>(W_cap, C.w_prefix[first + n] - C.w_prefix[first]);
      if (stream32) {
        // the streaming items run BESIDE the galloping ones, on a second stream: memory-bound galloping wavefronts and the
        // merging ones share the CUs (round 3 measured the two kinds in consecutive launches: 35 % slower than in one)
        DCD_HIP(hipEventRecord(ev_plan, st));
        DCD_HIP(hipStreamWaitEvent(st2, ev_plan, 0));
        // (DOCODO_QUERY_STREAM_FORM=lanes: the first form of the kernel, per-lane flags -- kept for the A/B of round 4)
        static const bool lanes_form = getenv("DOCODO_QUERY_STREAM_FORM") && !strcmp(getenv("DOCODO_QUERY_STREAM_FORM"), "lanes");
        static const int sk_wpe = getenv("DOCODO_QUERY_STREAM_WPE") ? atoi(getenv("DOCODO_QUERY_STREAM_WPE")) : DCD_SK_WAVES;
        hipLaunchKernelGGL(lanes_form ? k_pair_stream_lanes : sk_wpe >= 8 ? k_pair_stream<8> : sk_wpe == 7 ? k_pair_stream<7> : k_pair_stream<6>,
                           dim3((u32)std::min<u64>(W, sk_grid)), dim3(64), 0, st2, ops, n,
                           d_base_s32.p, d_part_base.p, idx_values.v64, pool, D.tmp, d_part_count.p, d_part_src.p, PART,
                           reinterpret_cast<unsigned long long*>(d_work_ctr.p + 1), d_need_off.p);
        DCD_HIP(hipEventRecord(ev_s32, st2));
        static const bool serial = getenv("DOCODO_QUERY_STREAM_SERIAL") && atoi(getenv("DOCODO_QUERY_STREAM_SERIAL")) != 0;   // (diagnosis: one kernel after the other)
        if (serial) DCD_HIP(hipStreamWaitEvent(st, ev_s32, 0));
      }
      hipLaunchKernelGGL(k_pair_parts, dim3((u32)std::min<u64>(W, pw_grid)), dim3(PW_T), 0, st, ops, n, d_base_old.p, d_part_base.p, idx_values, pool,
                         D.tmp, d_part_count.p, d_part_src.p, PART, ext_values ? nullptr : h->dir.p, ix.max_coord, d_mode.p, W,
                         reinterpret_cast<unsigned long long*>(d_work_ctr.p), d_need_off.p, 0, D.pool, d_part_excl.p);
      if (stream32) DCD_HIP(hipStreamWaitEvent(st, ev_s32, 0));
      DCD_HIP(hipMemsetAsync(d_part_count.p + W, 0, 8, st));
      scan_exclusive<u64, u64>(d_part_count.p, d_part_excl.p, W + 1, d_part_scr.p, st);
      // result lengths -> their places in the pool (device-side bump allocation) -> copy out of the scratch / second pass
      // (one single-workgroup kernel for totals + scan + placement was tried: its 4 x 16 k dependent loads from ONE CU took 62 us
      //  where these three launches take 20)
      hipLaunchKernelGGL(k_pair_totals, dim3((u32)div_up((u64)n, 256)), dim3(256), 0, st, ops, n, d_part_base.p, d_part_excl.p, D.counts.p, d_slice_cnt.p);
      scan_exclusive<u64, u64>(d_slice_cnt.p, d_slice_off.p, (u64)n, d_slice_scr.p, st);
      hipLaunchKernelGGL(k_pair_assign, dim3((u32)div_up((u64)n, 256)), dim3(256), 0, st, ops, n, d_slice_off.p, d_slice_scr.p + scan_nblocks(n), D.used.p,
                         D.seq, D.pool_cap, D.res_off.p, D.overflow.p);
      D.seq++;
      const u32 gg = (u32)std::min<u64>(div_up(div_up(W, 8), 4), 16384);
      hipLaunchKernelGGL(k_pair_gather, dim3(gg), dim3(256), 0, st, ops, n, d_part_base.p, W, d_part_count.p, d_part_src.p, d_part_excl.p, D.tmp,
                         D.pool, D.res_off.p, D.overflow.p, d_need_off.p, d_mode.p);
      hipLaunchKernelGGL(k_pair_parts, dim3((u32)std::min<u64>(W, pw_grid)), dim3(PW_T), 0, st, ops, n, d_base_g2.p, d_part_base.p, idx_values, pool,
                         D.tmp, d_part_count.p, d_part_src.p, PART, ext_values ? nullptr : h->dir.p, ix.max_coord, d_mode.p, W,
                         reinterpret_cast<unsigned long long*>(d_work_ctr.p + 2), d_need_off.p, 1, D.pool, d_part_excl.p);
      first += n;
    }
  }
  u64* pw = pinned_words();
  if (!pw) DCD_FAIL(DCD_E_NOMEM, "no pinned host memory");
  pw[0] = pw[1] = pw[2] = 0; pw[3] = 0;
  const ResParams& P = D.P;
  if (nq) {
    hipLaunchKernelGGL((k_result<false>), dim3((u32)nq), dim3(RS_T), 0, st, d_fin.p, (u32)nq, idx_values, pool, P, d_qc.p,
                       nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    // three scans on separate scratch, totals copied to the host together: one round trip instead of three
    const u64 se = scan_scratch_elems(nq), sn = scan_nblocks(nq);
    scan_exclusive_f<QcPagesF, u64>(QcPagesF{d_qc.p}, d_pbase.p, nq, sscr.p, st);
    scan_exclusive_f<QcPosF, u64>(QcPosF{d_qc.p}, d_sbase.p, nq, sscr.p + se, st);
    scan_exclusive_f<QcDocsF, u64>(QcDocsF{d_qc.p}, d_dbase.p, nq, sscr.p + 2 * se, st);
    DCD_HIP(hipMemcpyAsync(d_pbase.p + nq, sscr.p + sn, 8, hipMemcpyDeviceToDevice, st));
    DCD_HIP(hipMemcpyAsync(d_sbase.p + nq, sscr.p + se + sn, 8, hipMemcpyDeviceToDevice, st));
    DCD_HIP(hipMemcpyAsync(d_dbase.p + nq, sscr.p + 2 * se + sn, 8, hipMemcpyDeviceToDevice, st));
    DCD_HIP(hipMemcpyAsync(&pw[0], sscr.p + sn, 8, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipMemcpyAsync(&pw[1], sscr.p + se + sn, 8, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipMemcpyAsync(&pw[2], sscr.p + 2 * se + sn, 8, hipMemcpyDeviceToHost, st));
  } else {
    DCD_HIP(hipMemsetAsync(d_pbase.p, 0, 8, st));
    DCD_HIP(hipMemsetAsync(d_sbase.p, 0, 8, st));
    DCD_HIP(hipMemsetAsync(d_dbase.p, 0, 8, st));
  }
  DCD_HIP(hipMemcpyAsync(&pw[3], D.overflow.p, 4, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipStreamSynchronize(st));
  if ((u32)pw[3] != 0) DCD_FAIL(DCD_E_NOMEM, "query work pool full (%llu elements)", (unsigned long long)D.pool_cap);
  const u64 tot_pages = pw[0], tot_pos = pw[1], tot_docs = pw[2];
  // The result arrays of a chunk come out of ONE block that belongs to the handle and only grows (x 1.5): their sizes differ from
  // chunk to chunk, so the device cache (which hands out a cached block only to requests within 25 % of its size) answered every
  // chunk of 16 k phrase queries -- gigabytes of positions -- with nine hipMalloc / hipFree pairs: 2.4 s of host time per 100 k.
  const size_t res_need = 256 * 10 + (tot_pages + 1) * (4 + 8 + 4) + (tot_pos + 1) * 4 + (tot_docs + 1) * (4 + 4 + 4 + 4) + (tot_docs + nq + 1) * 4;
  if (res_need > h->res_arena.n) {
    const size_t want = std::max(res_need, h->res_arena.n + h->res_arena.n / 2);
    h->res_arena.reset();
    if (h->res_arena.alloc(dev, want) != DCD_OK) DCD_TRY(h->res_arena.alloc(dev, res_need));
  }
  ResCarve cv{h->res_arena.p, h->res_arena.n};
  struct { u32* p; } o_page_index{cv.take<u32>(tot_pages + 1)}, o_doc_index_tmp{cv.take<u32>(tot_docs + 1)}, o_doc_first{cv.take<u32>(tot_docs + nq + 1)},
      o_doc_index{cv.take<u32>(tot_docs + 1)}, o_doc_np{cv.take<u32>(tot_docs + 1)};
  struct { u64* p; } o_pos_off{cv.take<u64>(tot_pages + 1)};
  struct { int32_t* p; } o_pos{cv.take<int32_t>(tot_pos + 1)};
  struct { float* p; } o_page_rank{cv.take<float>(tot_pages + 1)}, o_doc_rank{cv.take<float>(tot_docs + 1)};
  if (cv.at > cv.cap) DCD_FAIL(DCD_E_NOMEM, "result arena too small (%zu of %zu bytes)", cv.cap, cv.at);
  pw[4] = tot_pos;
  DCD_HIP(hipMemcpyAsync(o_pos_off.p + tot_pages, &pw[4], 8, hipMemcpyHostToDevice, st));   // (also for an empty batch: pos_off = [0])
  if (nq) {
    hipLaunchKernelGGL((k_result<true>), dim3((u32)nq), dim3(RS_T), 0, st, d_fin.p, (u32)nq, idx_values, pool, P, d_qc.p,
                       d_pbase.p, d_sbase.p, d_dbase.p, o_page_index.p, o_pos_off.p, o_pos.p, o_doc_index_tmp.p, o_doc_first.p);
    if (tot_pages) hipLaunchKernelGGL(k_page_rank, dim3((u32)div_up(tot_pages, 256)), dim3(256), 0, st, o_pos_off.p, o_pos.p, tot_pages, o_page_rank.p);
    hipLaunchKernelGGL(k_doc_rank, dim3((u32)nq), dim3(256), 0, st, (u32)nq, d_qc.p, d_pbase.p, d_dbase.p, o_doc_first.p, o_page_index.p,
                       o_page_rank.p, P.page_hdr, P.doc_mult, o_doc_index_tmp.p, o_doc_index.p, o_doc_rank.p, o_doc_np.p, P.page_doc);
  }
  DCD_TRY(tm.stop());
  // ---- the per-query counts come back through ONE pinned block, behind ONE synchronisation --------------
  PinnedBlock back;
  const size_t qc_bytes = nq * sizeof(QCounts), off_bytes = (nq + 1) * 8;
  DCD_TRY(back.get(qc_bytes + 2 * off_bytes));
  u8* bp = static_cast<u8*>(back.p);
  if (nq) DCD_HIP(hipMemcpyAsync(bp, d_qc.p, qc_bytes, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipMemcpyAsync(bp + qc_bytes, d_pbase.p, off_bytes, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipMemcpyAsync(bp + qc_bytes + off_bytes, d_dbase.p, off_bytes, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  R->ms_device = tm.ms();
  {
    const QCounts* qc = reinterpret_cast<const QCounts*>(bp);
    R->status.resize(nq);
    for (size_t q = 0; q < nq; q++) R->status[q] = qc[q].status;
    R->page_off.resize(nq + 1); R->doc_off.resize(nq + 1);
    memcpy(R->page_off.data(), bp + qc_bytes, off_bytes);
    memcpy(R->doc_off.data(), bp + qc_bytes + off_bytes, off_bytes);
  }
  R->n_pages_total = tot_pages; R->n_pos_total = tot_pos; R->n_docs_total = tot_docs;
  if (o.no_download) {
    // every result array was produced on the device; only the per-query counts travel (throughput measurements)
    R->downloaded = false;
    return DCD_OK;
  }
  // ---- download ------------------------------------------------------------------------------------
  R->page_index.resize(tot_pages); R->page_rank.resize(tot_pages); R->pos_off.resize(tot_pages + 1); R->pos.resize(tot_pos);
  R->doc_index.resize(tot_docs); R->doc_rank.resize(tot_docs); R->doc_n_pages.resize(tot_docs);
  if (tot_pages) {
    DCD_HIP(hipMemcpy(R->page_index.data(), o_page_index.p, tot_pages * 4, hipMemcpyDeviceToHost));
    DCD_HIP(hipMemcpy(R->page_rank.data(), o_page_rank.p, tot_pages * 4, hipMemcpyDeviceToHost));
  }
  DCD_HIP(hipMemcpy(R->pos_off.data(), o_pos_off.p, (tot_pages + 1) * 8, hipMemcpyDeviceToHost));
  if (tot_pos) DCD_HIP(hipMemcpy(R->pos.data(), o_pos.p, tot_pos * 4, hipMemcpyDeviceToHost));
  if (tot_docs) {
    DCD_HIP(hipMemcpy(R->doc_index.data(), o_doc_index.p, tot_docs * 4, hipMemcpyDeviceToHost));
    DCD_HIP(hipMemcpy(R->doc_rank.data(), o_doc_rank.p, tot_docs * 4, hipMemcpyDeviceToHost));
    DCD_HIP(hipMemcpy(R->doc_n_pages.data(), o_doc_np.p, tot_docs * 4, hipMemcpyDeviceToHost));
  }
  if (o.want_sequences) {
    // final IndexSequence units per query: re-encode the final coordinates from 0 (Builder.AddRange)
    R->have_seq = true;
    R->seq_off.assign(nq + 1, 0);
    R->seq_R = C.finalR;
    std::vector<u64> cnt(n_refs_after ? n_refs_after : 1), roff(n_refs_after ? n_refs_after : 1);
    if (n_refs_after) {
      DCD_HIP(hipMemcpy(cnt.data(), D.counts.p, (size_t)n_refs_after * 8, hipMemcpyDeviceToHost));
      DCD_HIP(hipMemcpy(roff.data(), D.res_off.p, (size_t)n_refs_after * 8, hipMemcpyDeviceToHost));
    }
    for (size_t q = 0; q < nq; q++) {
      const Seg& s = C.finals[q].s;
      const u64 n = s.kind == 0 ? s.cnt_imm : cnt[s.cnt_ref];
      const size_t base = R->seq_units.size();
      if (n) {
        DevBuf<u64> part, plain;
        DevBuf<u16> du;
        const u64* src;
        if (s.kind != 0) src = D.pool + roff[s.cnt_ref];
        else if (idx_values.v64) src = idx_values.v64 + s.off;
        else {                                   // a final sequence that is an index term itself (single-term query): unpack it
          DCD_TRY(plain.alloc(dev, n));
          hipLaunchKernelGGL(k_unpack, dim3((u32)div_up(n, 256)), dim3(256), 0, st, idx_values.lo + s.off, idx_values.hi + s.off, n, plain.p);
          src = plain.p;
        }
        const u64 nb = encode_nblocks(n);
        DCD_TRY(part.alloc(dev, nb + 1));
        DCD_TRY(du.alloc(dev, 5 * n));
        EncSrcValues es{src};
        encode_launch(es, n, part.p, du.p, st);
        DCD_HIP(hipMemcpyAsync(&pw[5], part.p + nb, 8, hipMemcpyDeviceToHost, st));
        DCD_HIP(hipStreamSynchronize(st));
        const u64 total = pw[5];
        R->seq_units.resize(base + total);
        DCD_HIP(hipMemcpy(R->seq_units.data() + base, du.p, total * 2, hipMemcpyDeviceToHost));
      }
      R->seq_off[q + 1] = R->seq_units.size();
    }
  }
  return DCD_OK;
}

#ifdef DCD_PROF
extern "C" int32_t dcd_debug_prof(unsigned long long* out24, int32_t reset) {
  if (out24) (void)hipMemcpyFromSymbol(out24, HIP_SYMBOL(dcd::g_prof), 24 * 8);
  if (reset) { unsigned long long z[24] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(dcd::g_prof), z, 24 * 8); }
  return 0;
}
#endif
extern "C" {

void dcd_query_default_opts(dcd_query_opts* o) {
  if (!o) return;
  o->max_found_pages = 30000;   // Index.cs:101
  o->max_found_docs = 500;      // Index.cs:102
  o->doc_rank_multiply = 10.f;  // Index.cs:115
  o->want_sequences = 0;
  o->doc_match = nullptr;
  o->n_doc_match = 0;
  o->no_download = 0;
  o->reserved = 0;
}

// operators a query names and the postings of its leaves (= the capacity bound of its last operator)
static void query_prepass(const DeviceIndex& ix, const std::vector<SymVal>* ext_leaves, const int32_t* prog, u64 a, u64 b, u64* n_ops, u64* leaf_sum) {
  u64 ops = 0, sum = 0;
  for (u64 i = a; i < b;) {
    const int32_t opc = prog[i];
    if (opc == DCD_OP_TERM) {
      if (i + 3 > b) break;
      const int64_t t = prog[i + 1];
      if (ext_leaves) { if (t >= 0 && (size_t)t < ext_leaves->size()) sum += (*ext_leaves)[(size_t)t].cap; }
      else if (t >= 0 && (u64)t < ix.n_terms) sum += ix.h_term_item_start[t + 1] - ix.h_term_item_start[t];
      i += 3;
    } else if (opc == DCD_OP_SETR) i += 2;
    else { ops++; i += 1; }
  }
  *n_ops = ops;
  *leaf_sum = sum;
}

static void append_results(dcd_results& dst, const dcd_results& src) {
  const u64 p0 = dst.n_pages_total, s0 = dst.n_pos_total, d0 = dst.n_docs_total, u0 = dst.seq_units.size();
  dst.n_pages_total += src.n_pages_total; dst.n_pos_total += src.n_pos_total; dst.n_docs_total += src.n_docs_total;
  dst.downloaded = src.downloaded;
  const bool first = dst.page_off.empty();
  dst.status.insert(dst.status.end(), src.status.begin(), src.status.end());
  auto app_off = [&](std::vector<u64>& d, const std::vector<u64>& sv, u64 base) {
    const size_t k0 = first ? 0 : 1;                   // later chunks drop their leading 0
    for (size_t k = k0; k < sv.size(); k++) d.push_back(sv[k] + base);
  };
  app_off(dst.page_off, src.page_off, p0);
  app_off(dst.doc_off, src.doc_off, d0);
  // pos_off has one entry per page plus a final total: keep a single running terminator
  if (src.downloaded) {
    if (!dst.pos_off.empty()) dst.pos_off.pop_back();
    for (size_t k = 0; k < src.pos_off.size(); k++) dst.pos_off.push_back(src.pos_off[k] + s0);
  }
  dst.page_index.insert(dst.page_index.end(), src.page_index.begin(), src.page_index.end());
  dst.page_rank.insert(dst.page_rank.end(), src.page_rank.begin(), src.page_rank.end());
  dst.pos.insert(dst.pos.end(), src.pos.begin(), src.pos.end());
  dst.doc_index.insert(dst.doc_index.end(), src.doc_index.begin(), src.doc_index.end());
  dst.doc_rank.insert(dst.doc_rank.end(), src.doc_rank.begin(), src.doc_rank.end());
  dst.doc_n_pages.insert(dst.doc_n_pages.end(), src.doc_n_pages.begin(), src.doc_n_pages.end());
  if (src.have_seq) {
    dst.have_seq = true;
    app_off(dst.seq_off, src.seq_off, u0);
    dst.seq_units.insert(dst.seq_units.end(), src.seq_units.begin(), src.seq_units.end());
    dst.seq_R.insert(dst.seq_R.end(), src.seq_R.begin(), src.seq_R.end());
  }
  dst.ms_device += src.ms_device;
  dst.ops_named += src.ops_named; dst.ops_evaluated += src.ops_evaluated;
  dst.units_named += src.units_named; dst.units_evaluated += src.units_evaluated;
  dst.sub_batches += src.sub_batches;
}

// The handle's work arena = [pool | scratch].  Round 4: its DEFAULT size is an eighth of the device memory (round 3: half of
// it, because every operator reserved cap(a) + cap(b) pool elements and as much scratch): results are placed by the device where
// the previous one ended, and the scratch is reused slice by slice.  It grows only when one operator's capacity does not fit the
// scratch half; dcd_index_trim gives it back.
static int32_t arena_ensure(dcd_index* h, u64 pool_elems, u64 scratch_elems, u64 bound_elems) {
  const u64 want = pool_elems + scratch_elems;
  if (h->work_pool.p && h->work_pool_elems >= want) return DCD_OK;
  // (a growth is a hipMalloc of gigabytes: in steps of at least twice the old size, up to the bound)
  const u64 size = std::max(want, std::min(2 * h->work_pool_elems, bound_elems));
  h->work_pool.reset();
  h->work_pool_elems = 0;
  if (h->work_pool.alloc(h->ix.device, size) == DCD_OK) { h->work_pool_elems = size; return DCD_OK; }
  dev_pool_trim(h->ix.device);                             // (the cache may hold what is missing)
  DCD_TRY(h->work_pool.alloc(h->ix.device, want));
  h->work_pool_elems = want;
  return DCD_OK;
}

// one sub-batch: queries [q0, q1) with one pool and one table of common sub-expressions, run in pipelined chunks
static int32_t run_sub_batch(dcd_index* h, const int32_t* prog, const u64* prog_off, size_t q0, size_t q1, const dcd_query_opts& o,
                             dcd_results* R, const u64* ext_values, const std::vector<SymVal>* ext_leaves, u64 pool_bound, u64 scratch_bound,
                             bool exact_pool) {
  DeviceIndex& ix = h->ix;
  const int dev = ix.device;
  hipStream_t st = h->st;
  StreamScope ss_(st);
  if (o.max_found_docs > 512 || o.max_found_docs < 0 || o.max_found_pages < 0) DCD_FAIL(DCD_E_BADARG, "max_found_docs must be in [0, 512]");
  const auto t_sb0 = std::chrono::steady_clock::now();          // (DOCODO_TRACE: host phases of the sub-batch)
  auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  double ms_pre = 0, ms_setup = 0, ms_plan0 = 0, ms_exec = 0, ms_append = 0, ms_join = 0;
  if (!ext_values) DCD_TRY(ensure_values(h));
  u64 PART = PART_DEFAULT;
  if (const char* e = getenv("DOCODO_QUERY_PART")) PART = std::min<u64>(std::max<u64>(strtoull(e, nullptr, 10), 1), 1ull << 31);   // tests: tiny partitions (k_pair_stream counts a partition in 32 bits)
  size_t chunk_q = 16384;                                   // queries per chunk (plan of chunk k + 1 is made while chunk k runs)
  if (const char* e = getenv("DOCODO_QUERY_CHUNK")) chunk_q = std::max<size_t>(strtoull(e, nullptr, 10), 1);
  const size_t nq = q1 - q0;
  // pre-pass: operators per query (sizes of the staging blocks and of the result directory), largest operator capacity
  std::vector<u32> q_ops(nq);
  u64 ops_total = 0, max_cap = 0, cap_sum = 0;
  for (size_t q = q0; q < q1; q++) {
    u64 no = 0, ls = 0;
    query_prepass(ix, ext_leaves, prog, prog_off[q], prog_off[q + 1], &no, &ls);
    q_ops[q - q0] = (u32)std::min<u64>(no, 0xFFFFFFFFull);
    ops_total += no;
    if (no) max_cap = std::max(max_cap, ls);
    cap_sum += no * ls;                                     // (bound of the results of the query's operators together)
  }
  // chunk boundaries: the FIRST chunk is short (its plan is the only one the device waits for) and the following ones grow by
  // half: planning a query takes about 0.6 of the time its evaluation takes on a 10 M-document index, so the plan of a chunk
  // 1.5 x as long as the one that is running is still ready in time (with full chunks right behind a first one of 2048 queries the
  // device idled for 8 of the 100 ms of 100 k conjunctions).  A batch without operators (single-term queries) has nothing to plan:
  // full chunks from the start.
  std::vector<size_t> cb(1, q0);
  {
    size_t at = q0;
    size_t len = (ops_total != 0 && nq > chunk_q / 4) ? std::max<size_t>(1, chunk_q / 8) : chunk_q;
    do {                                                    // (an empty batch is one empty chunk)
      at = std::min(q1, at + len);
      cb.push_back(at);
      len = std::min(chunk_q, len + std::max<size_t>(1, len / 2));
    } while (at < q1);
    // (a short tail joins the chunk before it)
    if (cb.size() > 2 && cb.back() - cb[cb.size() - 2] < chunk_q / 16) cb.erase(cb.end() - 2);
  }
  const size_t n_chunks = cb.size() - 1;
  std::vector<u64> chunk_ops(n_chunks, 0);
  for (size_t c = 0; c < n_chunks; c++)
    for (size_t q = cb[c]; q < cb[c + 1]; q++) chunk_ops[c] += q_ops[q - q0];
  ms_pre = ms_since(t_sb0);
  if (ops_total >= 0xFFFFFFF0ull) DCD_FAIL(DCD_E_OVERFLOW, "query batch too large: %llu operators", (unsigned long long)ops_total);
  // the arena is as large as this sub-batch can need, up to the bounds: a small index never reserves gigabytes
  const u64 scratch_elems = std::max<u64>(std::max(std::min(scratch_bound, cap_sum), max_cap), 1);
  const u64 pool_elems = std::max<u64>(std::min(pool_bound, cap_sum), 1);
  DCD_TRY(arena_ensure(h, pool_elems, scratch_elems, pool_bound + std::max(scratch_bound, scratch_elems)));
  SubBatch sb;
  static const bool env_cse = !(getenv("DOCODO_QUERY_CSE") && atoi(getenv("DOCODO_QUERY_CSE")) == 0);   // (A/B knob)
  sb.use_cse = env_cse;
  sb.cse = &h->cse;
  if (sb.use_cse) sb.cse->init(ops_total);
  sb.ref_cap.reserve(ops_total); sb.ref_chunk.reserve(ops_total); sb.ref_round.reserve(ops_total);
  SubBatchDev D;
  DCD_TRY(D.counts.alloc(dev, ops_total ? ops_total : 1));
  DCD_TRY(D.res_off.alloc(dev, ops_total ? ops_total : 1));
  DCD_TRY(D.used.alloc(dev, 2));
  DCD_TRY(D.overflow.alloc(dev, 1));
  DCD_HIP(hipMemsetAsync(D.used.p, 0, 16, st));
  DCD_HIP(hipMemsetAsync(D.overflow.p, 0, 4, st));
  // [pool | scratch]: everything the arena holds beyond the scratch belongs to the pool
  D.scratch_cap = scratch_elems;
  D.pool_cap = exact_pool ? pool_elems : h->work_pool_elems - scratch_elems;
  D.pool = h->work_pool.p;
  D.tmp = h->work_pool.p + D.pool_cap;
  DCD_TRY(make_res_params(h, o, D.pd, &D.P));

  ChunkPlan plans[2];
  PinnedBlock pin_ops[2], pin_fin[2];
  auto prepare = [&](size_t c) -> int32_t {                // staging of chunk c (the executor owns every HIP object)
    ChunkPlan& P = plans[c & 1];
    P.chunk = (u32)c;
    P.q0 = cb[c];
    P.q1 = cb[c + 1];
    P.status = DCD_OK;
    P.err.clear();
    P.ops_cap = chunk_ops[c];
    DCD_TRY(pin_ops[c & 1].get(std::max<size_t>(1, P.ops_cap) * sizeof(PairOp)));
    DCD_TRY(pin_fin[c & 1].get(std::max<size_t>(1, P.q1 - P.q0) * sizeof(FinalSeq)));
    P.ops = static_cast<PairOp*>(pin_ops[c & 1].p);
    P.finals = static_cast<FinalSeq*>(pin_fin[c & 1].p);
    return DCD_OK;
  };
  ms_setup = ms_since(t_sb0) - ms_pre;
  DCD_TRY(prepare(0));
  plan_chunk(h, ext_leaves, sb, prog, prog_off, PART, plans[0]);
  ms_plan0 = ms_since(t_sb0) - ms_pre - ms_setup;
  int32_t status = DCD_OK;
  for (size_t c = 0; c < n_chunks && status == DCD_OK; c++) {
    ChunkPlan& P = plans[c & 1];
    if (P.status != DCD_OK) { set_error("%s", P.err.c_str()); status = P.status; break; }
    const u32 refs_after = sb.n_refs;                      // (results this chunk's final sequences can name)
    std::thread planner;
    if (c + 1 < n_chunks) {
      status = prepare(c + 1);
      if (status != DCD_OK) break;
      planner = std::thread(plan_chunk, h, ext_leaves, std::ref(sb), prog, prog_off, PART, std::ref(plans[(c + 1) & 1]));
    }
    const auto t_c = std::chrono::steady_clock::now();
    if (n_chunks == 1) status = exec_chunk(h, o, D, P, PART, R, ext_values, refs_after);
    else {
      dcd_results part;
      status = exec_chunk(h, o, D, P, PART, &part, ext_values, refs_after);
      const double e = ms_since(t_c);
      if (status == DCD_OK) append_results(*R, part);
      ms_append += ms_since(t_c) - e;
    }
    ms_exec += ms_since(t_c);
    const auto t_j = std::chrono::steady_clock::now();
    if (planner.joinable()) planner.join();
    ms_join += ms_since(t_j);
  }
  if (status != DCD_OK) { (void)hipStreamSynchronize(st); return status; }
  R->ops_named += sb.ops_total;
  R->ops_evaluated += sb.n_refs;
  R->units_named += sb.units_named;
  R->units_evaluated += sb.units_evaluated;
  R->sub_batches += 1;
  static const bool trace = getenv("DOCODO_TRACE") && atoi(getenv("DOCODO_TRACE")) != 0;
  if (trace) {
    u64 used[2] = {0, 0};
    (void)hipMemcpy(used, D.used.p, 16, hipMemcpyDeviceToHost);
    fprintf(stderr, "[docodo] query sub-batch: %zu queries, %zu chunks, %llu operators named, %u evaluated, %u passes (%llu sized by the device), pool %llu of %llu elements, scratch %llu\n",
            nq, n_chunks, (unsigned long long)sb.ops_total, sb.n_refs, D.seq, (unsigned long long)D.n_admit_syncs,
            (unsigned long long)std::max(used[0], used[1]), (unsigned long long)D.pool_cap, (unsigned long long)D.scratch_cap);
    fprintf(stderr, "[docodo]   host ms: prepass %.2f, setup %.2f, first plan %.2f, chunks %.2f (appending results %.2f), waiting for plans %.2f, total %.2f\n",
            ms_pre, ms_setup, ms_plan0, ms_exec, ms_append, ms_join, ms_since(t_sb0));
  }
  return DCD_OK;
}

// queries of one call: one sub-batch if the pool holds its results, otherwise halves (quarters, ...) of it
static int32_t run_queries(dcd_index* h, const int32_t* prog, const u64* prog_off, size_t n_queries, const dcd_query_opts& o,
                           std::unique_ptr<dcd_results>& out, const u64* ext_values, const std::vector<SymVal>* ext_leaves) {
  size_t free_b = 0, total_b = 0;
  DCD_HIP(hipMemGetInfo(&free_b, &total_b));
  static const u64 env_arena = getenv("DOCODO_QUERY_ARENA_BYTES") ? strtoull(getenv("DOCODO_QUERY_ARENA_BYTES"), nullptr, 10) : 0;
  const u64 arena_b = env_arena ? env_arena : (u64)total_b / 8;
  // elements: a quarter of the arena for the pool (operator results: doubled when it runs full), three quarters for the scratch --
  // the scratch decides into how many passes a round of operators falls (configs[3]: 16 k conjunctions of a chunk need 52 GB of
  // regions; arena of 36 / 77 / 154 GB split in halves: 84.5 / 80.6 / 78.7 ms per 100 k queries, 2.36 / 2.28 / 2.24 s per 100 k phrases)
  u64 pool_bound = std::max<u64>(arena_b / 32, 1ull << 16), scratch_bound = std::max<u64>(arena_b / 32 * 3, 1ull << 16);
  bool env_pool = false;
  if (const char* e = getenv("DOCODO_QUERY_POOL_ELEMS")) {   // tests / hosts short of memory: pool and scratch of this many elements
    pool_bound = scratch_bound = std::max<u64>(strtoull(e, nullptr, 10), 1);
    env_pool = true;
  }
  std::unique_ptr<dcd_results> R(new dcd_results());
  size_t q0 = 0, span = std::max<size_t>(n_queries, 1);
  bool first = true;
  while (q0 < n_queries || first) {
    const size_t q1 = std::min(n_queries, q0 + span);
    // (an attempt works on an object of its own that is adopted only on success: a failed one may have filled arrays already)
    std::unique_ptr<dcd_results> part(new dcd_results());
    const int32_t st = run_sub_batch(h, prog, prog_off, q0, q1, o, part.get(), ext_values, ext_leaves, pool_bound, scratch_bound, env_pool);
    if (st == DCD_OK) {
      if (first && q1 == n_queries) { R = std::move(part); break; }       // the common case: one sub-batch, no copies
      append_results(*R, *part);
      q0 = q1;
      first = false;
      continue;
    }
    if (st != DCD_E_NOMEM) return st;
    static const bool trace = getenv("DOCODO_TRACE") && atoi(getenv("DOCODO_TRACE")) != 0;
    if (trace) fprintf(stderr, "[docodo] query sub-batch of %zu queries failed (%s): retrying with %zu\n", q1 - q0, last_error().c_str(), (q1 - q0 + 1) / 2);
    // The pool ran full.  A larger pool first (up to a quarter of the device memory): the results of a sub-batch are shared by all
    // its queries, so halving the sub-batch evaluates the operators that both halves name twice -- with phrases over a Zipf
    // vocabulary that is most of the work (the results of ALL adjacent-word pairs together are only twice the corpus' postings).
    // Then half as many queries per sub-batch.
    (void)hipMemGetInfo(&free_b, &total_b);
    const u64 avail = ((u64)free_b + (u64)dev_pool_cached_bytes(h->ix.device) + 8 * h->work_pool_elems) / 8;
    const u64 pool_max = std::min<u64>((u64)total_b / 4 / 8, avail / 4 * 3 > 2 * scratch_bound ? avail / 4 * 3 - 2 * scratch_bound : 0);
    if (!env_pool && 2 * pool_bound <= pool_max) {
      pool_bound = std::min(2 * pool_bound, pool_max);        // (x 2, not x 4: 100 k phrases need 34 GB of results -- 36 GB do, 72 GB were held)
      h->work_pool.reset();                                    // (the next attempt allocates the larger arena)
      h->work_pool_elems = 0;
      continue;
    }
    if (q1 - q0 > 1) { span = (q1 - q0 + 1) / 2; first = false; continue; }
    return st;
  }
  out = std::move(R);
  return DCD_OK;
}

// On-disk postings (dcd_index_open_lazy; Index.cs:198-216, 346-348): a batch reads the lists of the terms its programs name,
// builds a small RESIDENT index over them (same device layout as a loaded index; it borrows the page table of the handle) and
// evaluates the programs there with the term ids translated.  The subset of the previous batches is kept while the union fits
// the handle's resident budget, so that a working set of hot terms is read from the file once.
static int32_t lazy_query_batch(dcd_index* h, const int32_t* prog, const uint64_t* prog_off, size_t n_queries, const dcd_query_opts* opts,
                                dcd_results** out) {
  LazyPostings& lz = *h->lazy;
  const u64 V = h->ix.n_terms;
  std::vector<u32> need;
  const u64 end = n_queries ? prog_off[n_queries] : 0;
  for (size_t q = 0; q < n_queries; q++) {
    for (u64 i = prog_off[q]; i < prog_off[q + 1];) {
      const int32_t opc = prog[i];
      if (opc == DCD_OP_TERM) {
        if (i + 3 > prog_off[q + 1]) DCD_FAIL(DCD_E_BADARG, "query %zu: truncated TERM", q);
        if (prog[i + 1] < 0 || (u64)prog[i + 1] >= V) DCD_FAIL(DCD_E_BADARG, "query %zu: term %d out of range", q, prog[i + 1]);
        need.push_back((u32)prog[i + 1]);
        i += 3;
      } else if (opc == DCD_OP_SETR) i += 2;
      else i += 1;
    }
  }
  std::sort(need.begin(), need.end());
  need.erase(std::unique(need.begin(), need.end()), need.end());
  if (!std::includes(lz.terms.begin(), lz.terms.end(), need.begin(), need.end()) || !lz.sub) {
    // new subset: the union with what is resident if that fits the budget, else the batch's own terms
    std::vector<u32> uni;
    std::set_union(lz.terms.begin(), lz.terms.end(), need.begin(), need.end(), std::back_inserter(uni));
    auto bytes_of = [&](const std::vector<u32>& ts) { u64 b = 0; for (u32 t : ts) b += 2ull * lz.n_units[t]; return b; };
    // resident cost of a term: its units (2 B) + the decoded coordinates the operators stream (5 B per posting, <= per unit)
    const std::vector<u32>& pick = (7 * bytes_of(uni) / 2 <= lz.resident_budget) ? uni : need;
    std::unique_ptr<dcd_index> sub(new dcd_index());
    DeviceIndex& sx = sub->ix;
    sx.max_coord = h->ix.max_coord;
    sx.n_terms = pick.size();
    sx.h_term_str_off.assign(pick.size() + 1, 0);          // (the subset is addressed by ordinal only: no keys)
    sx.h_term_unit_start.assign(1, 0);
    std::vector<u16> units;
    {
      u64 tot = 0;
      for (u32 t : pick) tot += lz.n_units[t];
      units.resize(tot);
    }
    u64 at = 0;
    for (u32 t : pick) {
      // a list that is resident already is read again only if the subset is rebuilt (the file cache of the OS holds it)
      if (lz.n_units[t]) DCD_TRY(lazy_read_units(lz, t, units.data() + at));
      lz.fetches++; lz.fetched_bytes += 2ull * lz.n_units[t];
      at += lz.n_units[t];
      sx.h_term_unit_start.push_back(at);
    }
    sx.n_units = at;
    DCD_TRY(index_from_host_units(sx, units, h->ix.device));
    DCD_TRY(stream_acquire(h->ix.device, &sub->st));
    if (lz.sub) { dcd_index* old = lz.sub.release(); dcd_index_free(old); }
    lz.sub = std::move(sub);
    lz.terms = pick;
  }
  // translate the programs
  std::vector<int32_t> p2(prog, prog + end);
  for (size_t q = 0; q < n_queries; q++) {
    for (u64 i = prog_off[q]; i < prog_off[q + 1];) {
      const int32_t opc = p2[i];
      if (opc == DCD_OP_TERM) {
        p2[i + 1] = (int32_t)(std::lower_bound(lz.terms.begin(), lz.terms.end(), (u32)p2[i + 1]) - lz.terms.begin());
        i += 3;
      } else if (opc == DCD_OP_SETR) i += 2;
      else i += 1;
    }
  }
  // the subset borrows the page table (host side and device copy) for the call
  dcd_index* sub = lz.sub.get();
  std::swap(sub->ix.pages, h->ix.pages);
  std::swap(sub->ix.page_end, h->ix.page_end);
  const int32_t st = dcd_query_batch(sub, p2.data(), prog_off, n_queries, opts, out);
  std::swap(sub->ix.pages, h->ix.pages);
  std::swap(sub->ix.page_end, h->ix.page_end);
  return st;
}

int32_t dcd_query_batch(dcd_index* h, const int32_t* prog, const uint64_t* prog_off, size_t n_queries, const dcd_query_opts* opts,
                        dcd_results** out) {
  if (!h || !prog_off || !out || (!prog && n_queries && prog_off[n_queries] != 0)) DCD_FAIL(DCD_E_BADARG, "dcd_query_batch: NULL argument");
  dcd_query_opts o;
  if (opts) o = *opts; else dcd_query_default_opts(&o);
  std::lock_guard<std::mutex> guard(h->mu);
  if (h->lazy) {
    DCD_TRY(ensure_device(h->ix.device));
    return lazy_query_batch(h, prog, prog_off, n_queries, opts, out);
  }
  DCD_TRY(ensure_device(h->ix.device));
  DCD_TRY(index_fetch_host(h->ix));     // (host mirrors of a merged index are fetched on first use)
  std::unique_ptr<dcd_results> R;
  DCD_TRY(run_queries(h, prog, prog_off, n_queries, o, R, nullptr, nullptr));
  *out = R.release();
  return DCD_OK;
}

// tests: the merge network of k_pair_stream on 256 keys (host memory, in place); the input must be a bitonic sequence
int32_t dcd_debug_sk_network(uint32_t* keys256) {
  if (!keys256) DCD_FAIL(DCD_E_BADARG, "dcd_debug_sk_network: NULL argument");
  DCD_TRY(ensure_device(0));
  hipStream_t st = nullptr;
  DCD_TRY(stream_acquire(0, &st));
  struct Guard { hipStream_t s; ~Guard() { stream_release(0, s); } } g{st};
  StreamScope ss_(st);
  DevBuf<u32> d;
  DCD_TRY(d.alloc(0, 256));
  DCD_HIP(hipMemcpyAsync(d.p, keys256, 1024, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_sk_network_probe, dim3(1), dim3(64), 0, st, d.p);
  DCD_HIP(hipMemcpyAsync(keys256, d.p, 1024, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  return DCD_OK;
}

// Adoption made explicit: the resident coordinates and the cell directories of the query side (otherwise made by the first batch).
// ms_device (may be NULL): device time of the kernels that made them, 0 when they existed already.
int32_t dcd_index_prepare(dcd_index* h, float* ms_device) {
  if (!h) DCD_FAIL(DCD_E_BADARG, "dcd_index_prepare: NULL argument");
  if (ms_device) *ms_device = 0.f;
  std::lock_guard<std::mutex> guard(h->mu);
  if (h->lazy) return DCD_OK;                                // (an on-disk index adopts the lists a batch names)
  DCD_TRY(ensure_device(h->ix.device));
  DCD_TRY(index_fetch_host(h->ix));
  StreamScope ss_(h->st);
  return ensure_values(h, ms_device);
}

// gives the handle's work arena (and, for an on-disk index, the resident subset) back to the device
int32_t dcd_index_trim(dcd_index* h) {
  if (!h) DCD_FAIL(DCD_E_BADARG, "dcd_index_trim: NULL argument");
  std::lock_guard<std::mutex> guard(h->mu);
  DCD_TRY(ensure_device(h->ix.device));
  if (h->st) DCD_HIP(hipStreamSynchronize(h->st));
  h->work_pool.reset();
  h->work_pool_elems = 0;
  h->res_arena.reset();
  { std::vector<CseTable::Ent>().swap(h->cse.t); h->cse.mask = 0; h->cse.gen = 0; }     // (host memory: the table of common sub-expressions)
  if (h->lazy && h->lazy->sub) { dcd_index* old = h->lazy->sub.release(); dcd_index_free(old); h->lazy->terms.clear(); }
  dev_pool_trim(h->ix.device);
  return DCD_OK;
}

int32_t dcd_results_get(dcd_results* r, dcd_results_view* v) {
  if (!r || !v) DCD_FAIL(DCD_E_BADARG, "dcd_results_get: NULL argument");
  memset(v, 0, sizeof *v);
  v->n_queries = r->status.size();
  v->status = r->status.data();
  v->page_off = r->page_off.data();
  v->page_index = r->page_index.data();
  v->page_rank = r->page_rank.data();
  v->pos_off = r->pos_off.data();
  v->pos = r->pos.data();
  v->doc_off = r->doc_off.data();
  v->doc_index = r->doc_index.data();
  v->doc_rank = r->doc_rank.data();
  v->doc_n_pages = r->doc_n_pages.data();
  if (r->have_seq) { v->seq_off = r->seq_off.data(); v->seq_units = r->seq_units.data(); v->seq_R = r->seq_R.data(); }
  v->ms_device = r->ms_device;
  v->n_pages_total = r->n_pages_total; v->n_pos_total = r->n_pos_total; v->n_docs_total = r->n_docs_total;
  v->ops_named = r->ops_named; v->ops_evaluated = r->ops_evaluated;
  v->units_named = r->units_named; v->units_evaluated = r->units_evaluated;
  v->sub_batches = r->sub_batches;
  if (!r->downloaded) {          // (no_download: the CSR arrays stayed on the device and are gone; the offsets and totals are valid)
    v->page_index = nullptr; v->page_rank = nullptr; v->pos_off = nullptr; v->pos = nullptr;
    v->doc_index = nullptr; v->doc_rank = nullptr; v->doc_n_pages = nullptr;
  }
  return DCD_OK;
}

void dcd_results_free(dcd_results* r) { delete r; }

}  // extern "C"

// ---- operator * / + on host buffers (IndexSequence shim) ----------------------------------------
static int32_t seq_binop(int opc, const u16* a, size_t na, int32_t Ra, const u16* b, size_t nb, int32_t Rb, u16* out, size_t cap,
                         size_t* n_out, int32_t* R_out) {
  if ((!a && na) || (!b && nb) || !n_out) DCD_FAIL(DCD_E_BADARG, "dcd_seq_and/or: NULL argument");
  // a throw-away index handle on device 0 carrying only a stream; operands are decoded into a
  // private value buffer that plays the role of the index values
  dcd_index tmp;
  tmp.ix.device = 0;
  DCD_TRY(ensure_device(0));
  DCD_TRY(stream_acquire(0, &tmp.st));
  struct Guard { hipStream_t s; ~Guard() { stream_release(0, s); } } g{tmp.st};
  hipStream_t st = tmp.st;
  StreamScope ss_(st);
  const u16* src[2] = {a, b};
  const size_t nu[2] = {na, nb};
  u64 nv[2] = {0, 0};
  DevBuf<u16> du[2];
  DevBuf<u64> part[2];
  for (int k = 0; k < 2; k++) {
    if (!nu[k]) continue;
    u64 nbk = div_up(nu[k], CODEC_TILE);
    DCD_TRY(du[k].alloc(0, nu[k]));
    DCD_TRY(part[k].alloc(0, nbk + 1));
    DCD_HIP(hipMemcpyAsync(du[k].p, src[k], nu[k] * 2, hipMemcpyHostToDevice, st));
    decode_count_launch(du[k].p, nu[k], part[k].p, st);
    DCD_HIP(hipMemcpyAsync(&nv[k], part[k].p + nbk, 8, hipMemcpyDeviceToHost, st));
  }
  DCD_HIP(hipStreamSynchronize(st));
  DevBuf<u64> vals, deltas, sscr;
  DCD_TRY(vals.alloc(0, nv[0] + nv[1] + 1));
  DCD_TRY(deltas.alloc(0, std::max(nv[0], nv[1]) + 1));
  DCD_TRY(sscr.alloc(0, scan_scratch_elems(std::max(nv[0], nv[1]) + 1)));
  u64 off = 0;
  std::vector<SymVal> leaves(2);
  for (int k = 0; k < 2; k++) {
    if (nv[k]) decode_values_launch(du[k].p, nu[k], part[k].p, nv[k], deltas.p, vals.p + off, sscr.p, st);
    DCD_HIP(hipStreamSynchronize(st));
    leaves[k].s.kind = 0; leaves[k].s.off = off; leaves[k].s.cnt_ref = NONE32; leaves[k].s.cnt_imm = nv[k];
    leaves[k].s.dir_off = 0; leaves[k].s.dir_shift = NONE32; leaves[k].s.pad = 0;
    leaves[k].cap = nv[k]; leaves[k].round = 0; leaves[k].R = 0; leaves[k].key = (u64)k;
    off += nv[k];
  }
  int32_t prog[7] = {DCD_OP_TERM, 0, Ra, DCD_OP_TERM, 1, Rb, opc};
  u64 poff[2] = {0, 7};
  dcd_query_opts o;
  dcd_query_default_opts(&o);
  o.want_sequences = 1;
  o.max_found_pages = 0;  // no page table here: results are not wanted
  std::unique_ptr<dcd_results> Rp;
  tmp.ix.pages = PageTable();
  DCD_TRY(run_queries(&tmp, prog, poff, 1, o, Rp, vals.p, &leaves));
  const dcd_results& R = *Rp;
  *n_out = R.seq_units.size();
  if (R_out) *R_out = R.seq_R[0];
  if (R.seq_units.size() > cap || (!out && !R.seq_units.empty())) return DCD_E_OVERFLOW;
  if (!R.seq_units.empty()) memcpy(out, R.seq_units.data(), R.seq_units.size() * 2);
  return DCD_OK;
}

extern "C" {
int32_t dcd_seq_and(const uint16_t* a, size_t na, int32_t Ra, const uint16_t* b, size_t nb, int32_t Rb, uint16_t* out, size_t cap,
                    size_t* n_out, int32_t* R_out) {
  return seq_binop(DCD_OP_AND, a, na, Ra, b, nb, Rb, out, cap, n_out, R_out);
}
int32_t dcd_seq_or(const uint16_t* a, size_t na, int32_t Ra, const uint16_t* b, size_t nb, int32_t Rb, uint16_t* out, size_t cap,
                   size_t* n_out, int32_t* R_out) {
  return seq_binop(DCD_OP_OR, a, na, Ra, b, nb, Rb, out, cap, n_out, R_out);
}
}

// dcd_sort.h -- stable LSD radix sort of (key u32, value u32) pairs for gfx950.
//
// Used for (a) the posting tuples (key = term, value = segment-local coordinate; the input is in
// coordinate order, so a STABLE sort by term alone yields each term's coordinates ascending --
// this is the "vocabulary hash -> segmented radix sort" stage replacing the red-black-tree
// descent of Index.Builder.Add, Build.cs:324-341) and (b) the ordinal sort of the term strings.
//
// One pass = histogram kernel (per-block digit counts, [digit][block] layout) -> device-wide
// exclusive scan -> scatter kernel.  Each block owns a contiguous range of items so the count
// matrix stays tiny (<= 2048 blocks).  The scatter ranks a 2048-item tile with 64-lane ballots
// (match-any on the digit), re-orders it in LDS and writes each digit run with consecutive lanes
// -> coalesced HBM stores; its barriers order LDS only, so stores and the next tile's loads overlap
// the ranking.  Per pass and item: 8 B read (hist) + 8 B read + 8 B write (scatter).
#pragma once
#include "dcd_common.h"
#include "dcd_prims.h"
#include <vector>

#include "dcd_scan.h"

namespace dcd {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
#ifndef DCD_RS_ROWS
#define DCD_RS_ROWS 12
#endif
constexpr int RS_ROWS = DCD_RS_ROWS;               // 64-item rows per wave and tile (8 waves x 12 rows: 6144-item tiles measured best)
#ifndef DCD_RSS_THREADS
#define DCD_RSS_THREADS 512
#endif
constexpr int RSS_THREADS = DCD_RSS_THREADS;        // workgroup of the scatter kernel (the histogram kernel keeps RS_THREADS)
constexpr int RSS_WAVES = RSS_THREADS / 64;
constexpr int RS_TILE = RSS_THREADS * RS_ROWS;      // items per scatter tile
#ifndef DCD_RS_MAX_BLOCKS
#define DCD_RS_MAX_BLOCKS 2048
#endif
constexpr int RS_MAX_BLOCKS = DCD_RS_MAX_BLOCKS;

struct RsPlan {
  u64 n;
  u32 nblk;
  u64 ipb;  // items per block (multiple of RS_TILE)
};
inline RsPlan rs_plan(u64 n) {
  RsPlan p;
  p.n = n;
  u64 tiles = div_up(n ? n : 1, RS_TILE);
  p.nblk = (u32)(tiles < (u64)RS_MAX_BLOCKS ? tiles : (u64)RS_MAX_BLOCKS);
  p.ipb = div_up(tiles, p.nblk) * RS_TILE;
  p.nblk = (u32)div_up(n ? n : 1, p.ipb);
  return p;
}

// key transform: optional gather through `remap` (first pass over hash-slot ids)
__device__ __forceinline__ u32 rs_key(u32 k, const u32* __restrict__ remap) { return remap ? remap[k] : k; }

#ifndef DCD_RS_UN
#define DCD_RS_UN 4
#endif
#ifndef DCD_RS_HCOPIES
#define DCD_RS_HCOPIES 8
#endif
// remap[key] through a small direct-mapped LDS cache: the keys of the first pass are vocabulary slots with a Zipf
// distribution, the table is sparse (one useful entry per cache line) and every miss of the CU's L1 is an L2 request.
// Entry = (key >> bits) << 20 | value in ONE 32-bit word (no torn reads); usable while values < 2^20 - 1 (so that no
// valid entry equals the empty word 0xFFFFFFFF) and keys < 2^(bits + 12) (the host checks).
#ifndef DCD_RC_BITS
#define DCD_RC_BITS 12
#endif
constexpr int RC_BITS = DCD_RC_BITS;        // histogram kernel: 4096 entries
constexpr int RCS_BITS = 11;       // scatter kernel: 2048 entries (its LDS is nearly full with the tile)
template <int CB>
__device__ __forceinline__ u32 rs_key_cached(u32 k, const u32* __restrict__ remap, u32* cache) {
  const u32 e = cache[k & ((1u << CB) - 1)];
  if ((e >> 20) == (k >> CB) && e != 0xFFFFFFFFu) return e & 0xFFFFFu;   // (an empty entry carries the tag 0xFFF)
  const u32 v = remap[k];
  cache[k & ((1u << CB) - 1)] = ((k >> CB) << 20) | v;
  return v;
}
inline bool rs_cache_ok(u64 max_key_excl, u64 max_value_excl) {
  return max_key_excl <= (1ull << (RCS_BITS + 12)) && max_value_excl < (1ull << 20) - 1;
}

template <int BITS, bool CACHE>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint2* __restrict__ in, const u32* __restrict__ remap, u64 n, u64 ipb,
                                                       u32 nblk, int shift, u32* __restrict__ hist) {
  constexpr int NB = 1 << BITS;
  __shared__ u32 s_cache[CACHE ? (1 << RC_BITS) : 1];
  if (CACHE)
    for (int i = threadIdx.x; i < (1 << RC_BITS); i += RS_THREADS) s_cache[i] = 0xFFFFFFFFu;
  auto key_of = [&](u32 k) -> u32 { return CACHE ? rs_key_cached<RC_BITS>(k, remap, s_cache) : rs_key(k, remap); };
  // Same-address LDS atomics of one wave instruction are executed one after the other (a Zipf head puts ~9 of 64
  // lanes on one digit): lane l counts into copy l % HC of the histogram, so that peers mostly hit different
  // addresses, and the copies are NB + 1 words apart, so that they hit different banks.  The copies are shared by
  // the four waves of the block.  (Finding the peer lanes with BITS ballots first costs more VALU time than the
  // remaining conflicts -- measured.)
  constexpr int HC = DCD_RS_HCOPIES, HS = NB + 1;
  __shared__ u32 h[HC * HS];
  for (int i = threadIdx.x; i < HC * HS; i += RS_THREADS) h[i] = 0;
  __syncthreads();
  u32* hl = h + (threadIdx.x % HC) * HS;
  u64 beg = (u64)blockIdx.x * ipb, end = beg + ipb < n ? beg + ipb : n;
  // The order of the items inside a block does not matter for the counts: every lane reads TWO tuples per load
  // (16 B) and keeps four loads in flight, so that enough bytes are outstanding to cover the HBM latency.
  constexpr int UN = DCD_RS_UN;
  // (beg is a multiple of RS_TILE, but `in` itself may be only 8-byte aligned -- the cold region of a hot / cold sort starts
  // behind an arbitrary number of hot tuples: a block then peels its first tuple, so that the pairs are 16-byte aligned;
  // a last tuple without a partner is counted by itself, so that no load reaches past `end`)
  if (end > beg) {
    u64 b0 = beg;
    if ((reinterpret_cast<uintptr_t>(in + beg) & 15) != 0) {
      if (threadIdx.x == 0) atomicAdd(&hl[(key_of(in[beg].x) >> shift) & (NB - 1)], 1u);
      b0 = beg + 1;
    }
    const u64 span = end - b0;
    const uint4* in2 = reinterpret_cast<const uint4*>(in + b0);
    const u64 pairs = span >> 1;                               // full pairs
    if ((span & 1) && threadIdx.x == 64) atomicAdd(&hl[(key_of(in[end - 1].x) >> shift) & (NB - 1)], 1u);
    for (u64 o = 0; o < pairs; o += (u64)RS_THREADS * UN) {
      uint4 kv[UN];
      u64 p[UN];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        p[u] = o + (u64)u * RS_THREADS + threadIdx.x;
        kv[u] = p[u] < pairs ? in2[p[u]] : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int u = 0; u < UN; u++) {
        if (p[u] < pairs) {
          atomicAdd(&hl[(key_of(kv[u].x) >> shift) & (NB - 1)], 1u);
          atomicAdd(&hl[(key_of(kv[u].z) >> shift) & (NB - 1)], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < NB; d += RS_THREADS) {
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < HC; k++) s += h[k * HS + d];
    hist[(u64)d * nblk + blockIdx.x] = s;
  }
}

// ---- first pass over a TILE-STRIDED input ------------------------------------------------------------
// The tokeniser can write tile t's tuples at in[t * stride + 0 .. cnt[t]) without knowing where the tile lands in a dense
// array (no counting pass over the text).  The first radix pass reads that layout directly: a block owns a fixed number
// of TILES (tpb) instead of a fixed number of tuples, its tuples in order are the tiles' tuples in order, 64 per "row"
// (the last row of a tile is partly empty).  Everything behind the ranking is the same as for dense input.
struct RsStrided {
  const u32* cnt;        // [n_tiles]
  u32 stride;            // tuples reserved per tile (>= any cnt), even
  u64 n_tiles;
  // hot / cold split (hot_s != 0): the remapped key is a dense id d whose first hot_n values are the hot terms; the sort
  // key becomes d << hot_s for a hot term and (hot_n << hot_s) + (d - hot_n) for a cold one (d - hot_n < 2^hot_s), so that
  // the digit (key >> hot_s) gives every hot term a bucket of its own and all cold terms the bucket hot_n
  u32 hot_n = 0, hot_s = 0;
  // host side only: the caller has filled the pass's histogram (hist[digit * nblk + block], digit = key >> shift) itself -- the
  // tokeniser counts the digits of the first pass while it writes the tuples -- so the histogram kernel is not launched
  bool hist_ready = false;
  // ---- the pass BEHIND a hot / cold first pass without a histogram kernel of its own (round 3) -------------------------------
  // first pass (cnt != nullptr, hot_s != 0): next_hist != nullptr = the scatter kernel also counts, per block, the digit
  // (key & ((1 << next_bits) - 1)) of its COLD tuples and leaves them as next_hist[digit * nblk + block] -- exactly the histogram
  // of a following pass whose block b sorts the cold tuples block b of this pass wrote (contiguous, in order).
  u32* next_hist = nullptr;
  u32 next_bits = 0;
  // first pass with a hot set (round 4): the tuples of the hot digits leave as their 4-byte COORDINATES only -- the bucket implies
  // the term -- at out32[position] (out32 = the output array seen as u32: the hot region [0, N_hot) then takes the first half of
  // the bytes its tuples would; the cold tuples keep their places behind it).  The pack kernels read 4 instead of 8 bytes for
  // two thirds of the postings (EncSrcPairs::n_hot32).
  bool hot32 = false;
  // following pass (cnt == nullptr, range != nullptr): dense input, but block b owns the tuples [range[b], range[b + 1]) of `in`
  // (the last block ends at range_end) instead of a fixed number: range = the first pass's scanned offsets of the cold digit.
  const u32* range = nullptr;
  u32 range_nblk = 0;
  u64 range_end = 0;
};
__host__ __device__ __forceinline__ u32 rs_hot_key(u32 d, u32 hot_n, u32 hot_s) { return d < hot_n ? d << hot_s : (hot_n << hot_s) + (d - hot_n); }
// the key has to fit 32 bits: digits 0 .. hot_n above hot_s bits of cold id (the host gives up the split otherwise)
inline bool rs_hot_key_fits(u32 hot_n, u32 hot_s) { return hot_s < 32 && (((u64)hot_n << hot_s) + ((1ull << hot_s) - 1)) <= 0xFFFFFFFFull; }
constexpr int RS_STRIDE_MAX = 4096;      // largest tile population handled (the host checks its stride against it; counts travel as u16)
constexpr int RS_TPB_MAX = 2304;         // tiles per block (LDS slice of the tile table in the scatter kernel)
struct RsTilePlan { u32 nblk, tpb; };
inline RsTilePlan rs_tile_plan(u64 n_tiles) {
  RsTilePlan p;
  u64 nb = n_tiles < (u64)RS_MAX_BLOCKS ? (n_tiles ? n_tiles : 1) : (u64)RS_MAX_BLOCKS;
  p.tpb = (u32)div_up(n_tiles ? n_tiles : 1, nb);
  p.nblk = (u32)div_up(n_tiles ? n_tiles : 1, p.tpb);
  return p;
}

#ifndef DCD_RS_TU
#define DCD_RS_TU 2
#endif
#ifndef DCD_RS_HPR
#define DCD_RS_HPR 5
#endif
constexpr int RS_HPR = DCD_RS_HPR;
#ifndef DCD_RS_NT
#define DCD_RS_NT 1      // pass A reads the tokeniser's tuples with non-temporal loads: they are touched once and would evict the
                         // 4 MB dense-id table from L2 (measured: histogram 0.77 -> 0.74 ms, scatter 1.36 -> 1.32..1.36)
#endif
#ifndef DCD_RSH_WPE
#define DCD_RSH_WPE 1
#endif
template <int BITS, bool CACHE>
__global__ __launch_bounds__(RS_THREADS, DCD_RSH_WPE) void k_rs_hist_strided(const uint2* __restrict__ in, RsStrided sd, const u32* __restrict__ remap,
                                                               u32 tpb, u32 nblk, int shift, u32* __restrict__ hist) {
  constexpr int NB = 1 << BITS;
  __shared__ u32 s_cache[CACHE ? (1 << RC_BITS) : 1];
  if (CACHE)
    for (int i = threadIdx.x; i < (1 << RC_BITS); i += RS_THREADS) s_cache[i] = 0xFFFFFFFFu;
  auto key_of = [&](u32 k) -> u32 {
    const u32 d = CACHE ? rs_key_cached<RC_BITS>(k, remap, s_cache) : rs_key(k, remap);
    return sd.hot_s ? rs_hot_key(d, sd.hot_n, sd.hot_s) : d;
  };
  constexpr int HC = DCD_RS_HCOPIES, HS = NB + 1;
  __shared__ u32 h[HC * HS];
  for (int i = threadIdx.x; i < HC * HS; i += RS_THREADS) h[i] = 0;
  __syncthreads();
  u32* hl = h + (threadIdx.x % HC) * HS;
  const u32 lane = lane_id(), w = threadIdx.x >> 6;
  const u64 ta = (u64)blockIdx.x * tpb, tb = ta + tpb < sd.n_tiles ? ta + tpb : sd.n_tiles;
  // a wave takes DCD_RS_TU tiles at a time; a lane reads tuple pairs (16 B; the stride is even, so tiles start 16-byte aligned),
  // RS_HPR rounds of 64 pairs per tile in flight (640 tuples: a group of four wave tiles of the tokeniser holds ~520 on
  // the benchmark corpus); what a fuller tile holds beyond that is read by the loop behind
  constexpr int TU = DCD_RS_TU, PR = RS_HPR;
  for (u64 t0 = ta + (u64)w * TU; t0 < tb; t0 += (u64)RS_WAVES * TU) {
    u32 c[TU];
    uint4 kv[TU][PR];
#pragma unroll
    for (int u = 0; u < TU; u++) c[u] = t0 + u < tb ? sd.cnt[t0 + u] : 0u;
#pragma unroll
    for (int u = 0; u < TU; u++) {
      const uint4* tp = reinterpret_cast<const uint4*>(in + (t0 + u) * sd.stride);
#pragma unroll
      for (int k = 0; k < PR; k++) {
        const u32 pi = lane + 64u * k;
#if DCD_RS_NT
        if (2 * pi < c[u]) {
          typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
          const u32x4_t q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(tp + pi));
          kv[u][k] = make_uint4(q.x, q.y, q.z, q.w);
        } else kv[u][k] = make_uint4(0u, 0u, 0u, 0u);
#else
        kv[u][k] = 2 * pi < c[u] ? tp[pi] : make_uint4(0u, 0u, 0u, 0u);     // (the second tuple of the last pair may be unused)
#endif
      }
    }
#pragma unroll
    for (int u = 0; u < TU; u++) {
#pragma unroll
      for (int k = 0; k < PR; k++) {
        const u32 pi = lane + 64u * k;
        if (2 * pi < c[u]) atomicAdd(&hl[(key_of(kv[u][k].x) >> shift) & (NB - 1)], 1u);
        if (2 * pi + 1 < c[u]) atomicAdd(&hl[(key_of(kv[u][k].z) >> shift) & (NB - 1)], 1u);
      }
      if (c[u] > 128u * PR) {          // (wave-uniform)
        const uint4* tp = reinterpret_cast<const uint4*>(in + (t0 + u) * sd.stride);
        for (u32 pi = lane + 64u * PR; 2 * pi < c[u]; pi += 64u) {
          const uint4 q = tp[pi];
          atomicAdd(&hl[(key_of(q.x) >> shift) & (NB - 1)], 1u);
          if (2 * pi + 1 < c[u]) atomicAdd(&hl[(key_of(q.z) >> shift) & (NB - 1)], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < NB; d += RS_THREADS) {
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < HC; k++) s += h[k * HS + d];
    hist[(u64)d * nblk + blockIdx.x] = s;
  }
}

#ifndef DCD_RS_WPE
#define DCD_RS_WPE 4
#endif
template <int BITS, bool CACHE, bool STRIDED>
__global__ __launch_bounds__(RSS_THREADS) __attribute__((amdgpu_waves_per_eu(DCD_RS_WPE, DCD_RS_WPE))) void k_rs_scatter(const uint2* __restrict__ in, const u32* __restrict__ remap, u64 n, u64 ipb,
                                                          u32 nblk, int shift, const u32* __restrict__ hist_scanned,
                                                          uint2* __restrict__ out, RsStrided sd, u32 tpb) {
  constexpr int NB = 1 << BITS;
  constexpr int Q = (NB + RSS_THREADS - 1) / RSS_THREADS;   // digits per owner thread: d = threadIdx.x * Q + q (adjacent)
  __shared__ u32 s_cnt[RSS_WAVES][NB];   // ranking: per-wave digit counts; afterwards: position of (wave, digit) in the sorted tile
  __shared__ u32 s_off[NB];             // output position of the digit's run minus its start in the sorted tile (mod 2^32)
  __shared__ uint2 s_tile[RS_TILE];
  __shared__ u32 s_wtot[RSS_WAVES];
  __shared__ u32 s_cache[CACHE ? (1 << RCS_BITS) : 1];
  const u32 lane = lane_id(), w = threadIdx.x >> 6;
  if (CACHE) {
    for (int i = threadIdx.x; i < (1 << RCS_BITS); i += RSS_THREADS) s_cache[i] = 0xFFFFFFFFu;
    lds_barrier();
  }

  // running output position of the owned digits: registers of the owner thread
  u32 base[Q];
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int d = threadIdx.x * Q + q;
    base[q] = d < NB ? hist_scanned[(u64)d * nblk + blockIdx.x] : 0u;
  }
  // dense input: the block owns tuples [beg, end).  Strided input: it owns tiles [ta, ta + nt); its tuples in order are
  // the tiles' tuples, 64 per row, every tile starting a new row (its last row is partly empty: ~20 % more rows than a
  // dense layout, but a row is one contiguous load; rows packed across tile borders cost more than they save --
  // measured).  s_rowpre[t] = number of rows before tile t (block-local), rows_b = all of them.
  u64 beg = (u64)blockIdx.x * ipb, end = beg + ipb < n ? beg + ipb : n;
  if (!STRIDED && sd.range) {                      // variable block ranges (see RsStrided::range)
    beg = sd.range[blockIdx.x];
    end = blockIdx.x + 1 < nblk ? (u64)sd.range[blockIdx.x + 1] : sd.range_end;
  }
  // counts of the NEXT pass's digit among this block's cold tuples (first pass of a hot / cold sort only)
  __shared__ u32 s_nx[STRIDED ? 256 : 1];
  const bool count_next = STRIDED && sd.next_hist != nullptr;
  if (count_next) { for (int i = threadIdx.x; i < 256; i += RSS_THREADS) s_nx[i] = 0; }
  __shared__ u16 s_tcnt[STRIDED ? RS_TPB_MAX : 1];
  __shared__ u16 s_rowpre[STRIDED ? RS_TPB_MAX + 1 : 1];
  const u64 ta = (u64)blockIdx.x * tpb;
  u32 nt = 0, rows_b = 0;
  if (STRIDED) {
    nt = (u32)(ta + tpb < sd.n_tiles ? tpb : sd.n_tiles - ta);
    constexpr int K = (RS_TPB_MAX + RSS_THREADS - 1) / RSS_THREADS;
    u32 loc[K], sum = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const u32 i = threadIdx.x * K + k;
      const u32 c = i < nt ? sd.cnt[ta + i] : 0u;
      if (i < nt) s_tcnt[i] = (u16)c;
      loc[k] = sum;
      sum += (c + 63u) >> 6;
    }
    const u32 incl = wave_scan_incl(sum);
    if (lane == 63) s_wtot[w] = incl;
    lds_barrier();
    u32 ex = incl - sum;
#pragma unroll
    for (int k = 0; k < RSS_WAVES; k++) ex += k < (int)w ? s_wtot[k] : 0u;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const u32 i = threadIdx.x * K + k;
      if (i < nt) s_rowpre[i] = (u16)(ex + loc[k]);
    }
    if (threadIdx.x == RSS_THREADS - 1) s_rowpre[nt] = (u16)(ex + sum);
    lds_barrier();
    rows_b = s_rowpre[nt];
  }
  // rows q .. q + RS_ROWS of the block (strided input) -> item[], vrow[] (number of tuples in the row; wave-uniform)
  u32 vrow[RS_ROWS];
  uint2 item[RS_ROWS];
  auto load_rows = [&](u32 q) {
    u32 t = nt, c = 0, ct = 0;                 // (rows behind the block's last one: no tile, nothing to advance to)
    if (q < rows_b) {
      u32 lo = 0, hi = nt;                     // last tile with s_rowpre <= q: the non-empty one among equal prefixes
      while (lo + 1 < hi) { const u32 m = (lo + hi) >> 1; if (s_rowpre[m] <= q) lo = m; else hi = m; }
      t = lo; c = q - s_rowpre[lo]; ct = s_tcnt[lo];
    }
    t = __builtin_amdgcn_readfirstlane(t); c = __builtin_amdgcn_readfirstlane(c); ct = __builtin_amdgcn_readfirstlane(ct);
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
      const u32 j0 = c * 64u;
      const u32 left = ct > j0 ? ct - j0 : 0u;
      vrow[r] = left < 64u ? left : 64u;
#if DCD_RS_NT
      if (lane < vrow[r]) {
        typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t q = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(in + (ta + t) * sd.stride + j0 + lane));
        item[r] = make_uint2(q.x, q.y);
      } else item[r] = make_uint2(0xFFFFFFFFu, 0u);
#else
      item[r] = lane < vrow[r] ? in[(ta + t) * sd.stride + j0 + lane] : make_uint2(0xFFFFFFFFu, 0u);
#endif
      // next row: the next 64 tuples of this tile, else the first ones of the next non-empty tile
      if (left > 64u) {
        c++;
      } else {
        c = 0; ct = 0;
        u32 t2 = t + 1;
        while (t2 < nt && s_tcnt[t2] == 0) t2++;
        if (t2 < nt) ct = s_tcnt[t2];
        t = __builtin_amdgcn_readfirstlane(t2); ct = __builtin_amdgcn_readfirstlane(ct);
      }
    }
  };

  // Four LDS-only barriers per tile; the scattered stores of a tile and the loads of the next one stay in flight
  // across them (the next tile is fetched into the item registers as soon as the current one sits in LDS).
  if (STRIDED) {
    load_rows(w * RS_ROWS);
  } else {
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
      u64 i = beg + (u64)(w * RS_ROWS + r) * 64 + lane;
      item[r] = i < end ? in[i] : make_uint2(0xFFFFFFFFu, 0u);
    }
  }
  const u64 n_steps = STRIDED ? ((u64)rows_b + RS_ROWS * RSS_WAVES - 1) / (RS_ROWS * RSS_WAVES)
                              : (end > beg ? (end - beg + RS_TILE - 1) / RS_TILE : 0);
  for (u64 step = 0; step < n_steps; step++) {
    const u64 t0 = beg + step * RS_TILE;                          // (dense input)
    const u32 q0 = (u32)step * RS_ROWS * RSS_WAVES;               // (strided input)
    // ---- A: rank inside the wave (rows in order, lanes in order: stable); s_cnt[w] is private to the wave here
    for (int i = lane; i < NB; i += 64) s_cnt[w][i] = 0;
    u32 dig[RS_ROWS], rk[RS_ROWS];
    bool valid[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
      valid[r] = STRIDED ? lane < vrow[r] : t0 + (u64)(w * RS_ROWS + r) * 64 + lane < end;
      if (valid[r]) {
        item[r].x = CACHE ? rs_key_cached<RCS_BITS>(item[r].x, remap, s_cache) : rs_key(item[r].x, remap);   // (first pass only)
        if (STRIDED && sd.hot_s) item[r].x = rs_hot_key(item[r].x, sd.hot_n, sd.hot_s);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
      u32 d = (item[r].x >> shift) & (NB - 1);
      // match-any on the digit: per bit, the lanes that agree with this lane are m (bit set) or ~m, i.e. m ^ (bit - 1);
      // kept in 32-bit halves so that each step is one three-input bit operation per half
      const u64 vm = ballot64(valid[r]);
      u32 plo = (u32)vm, phi = (u32)(vm >> 32);
#pragma unroll
      for (int b = 0; b < BITS; b++) {
        const u32 bit = (d >> b) & 1u;
        const u64 m = ballot64(bit != 0u);
        const u32 t = bit - 1u;
        plo &= (u32)m ^ t;
        phi &= (u32)(m >> 32) ^ t;
      }
      const u64 peers = ((u64)phi << 32) | plo;
      u32 prev = valid[r] ? s_cnt[w][d] : 0u;   // count of this digit in earlier rows of this wave
      u32 below = mbcnt(peers);
      if (valid[r] && below == 0) s_cnt[w][d] = prev + popc64(peers);  // leader = lowest peer lane
      dig[r] = valid[r] ? d : 0xFFFFFFFFu;
      rk[r] = prev + below;
    }
    lds_barrier();
    // ---- B: per digit, exclusive offsets across the waves; exclusive scan of the tile totals over the digits
    u32 wex[Q][RSS_WAVES], tot[Q], mine = 0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int d = threadIdx.x * Q + q;
      u32 run = 0;
#pragma unroll
      for (int k = 0; k < RSS_WAVES; k++) {
        u32 c = d < NB ? s_cnt[k][d] : 0u;
        wex[q][k] = run;
        run += c;
      }
      tot[q] = run;
      mine += run;
    }
    const u32 incl = wave_scan_incl(mine);
    if (lane == 63) s_wtot[w] = incl;
    lds_barrier();
    u32 ex = incl - mine;
#pragma unroll
    for (int k = 0; k < RSS_WAVES; k++) ex += k < (int)w ? s_wtot[k] : 0u;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int d = threadIdx.x * Q + q;
      if (d < NB) {
#pragma unroll
        for (int k = 0; k < RSS_WAVES; k++) s_cnt[k][d] = ex + wex[q][k];
        s_off[d] = base[q] - ex;
      }
      base[q] += tot[q];
      ex += tot[q];
    }
    lds_barrier();
    // ---- C: tile -> LDS in sorted order; fetch the next tile
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++)
      if (dig[r] != 0xFFFFFFFFu) s_tile[s_cnt[w][dig[r]] + rk[r]] = item[r];
    if (STRIDED) {
      load_rows(q0 + RS_ROWS * RSS_WAVES + w * RS_ROWS);
    } else {
#pragma unroll
      for (int r = 0; r < RS_ROWS; r++) {
        u64 i = t0 + RS_TILE + (u64)(w * RS_ROWS + r) * 64 + lane;
        item[r] = i < end ? in[i] : make_uint2(0xFFFFFFFFu, 0u);
      }
    }
    lds_barrier();
    // ---- D: digit runs -> HBM with consecutive lanes
    u32 tile_n;
    if (STRIDED) {
      tile_n = 0;
#pragma unroll
      for (int k = 0; k < RSS_WAVES; k++) tile_n += s_wtot[k];    // tuples of this step = sum of the waves' digit totals
    } else {
      tile_n = (u32)((end - t0) < (u64)RS_TILE ? (end - t0) : (u64)RS_TILE);
    }
    for (u32 j = threadIdx.x; j < tile_n; j += RSS_THREADS) {
      uint2 it = s_tile[j];
      u32 d = (it.x >> shift) & (NB - 1);
      if (count_next && d == sd.hot_n) atomicAdd(&s_nx[it.x & ((1u << sd.next_bits) - 1u)], 1u);
      if (STRIDED && sd.hot32 && d < sd.hot_n) { reinterpret_cast<u32*>(out)[(u64)(u32)(s_off[d] + j)] = it.y; continue; }
#if DCD_RS_NT >= 2
      if (STRIDED) {
        typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
        u32x2_t q; q.x = it.x; q.y = it.y;
        __builtin_nontemporal_store(q, reinterpret_cast<u32x2_t*>(out + (u64)(u32)(s_off[d] + j)));
      } else
#endif
      out[(u64)(u32)(s_off[d] + j)] = it;
    }
    // no barrier here: phase A of the next tile touches only the wave's own s_cnt row, and everything that phase D
    // reads is rewritten behind the next tile's first barrier at the earliest
  }
  if (count_next) {
    lds_barrier();
    for (u32 dd = threadIdx.x; dd < (1u << sd.next_bits); dd += RSS_THREADS) sd.next_hist[(u64)dd * nblk + blockIdx.x] = s_nx[dd];
  }
}

inline u64 rs_hist_elems(u64 n, int bits) { return (u64)rs_plan(n).nblk << bits; }

// optional per-kernel event probes (build stats): pairs of events recorded around the two kernels
struct RsProbe {
  std::vector<hipEvent_t>* ev_hist = nullptr;     // appended: begin, end per pass (if the kernel's probes are on: event_probe_on)
  std::vector<hipEvent_t>* ev_scatter = nullptr;
  int device = 0;
  int k_hist = 0, k_scatter = 0;                  // the kernels' DCD_K_* ids
  // optional: called between the scan of the histogram and the scatter kernel of a pass (the scanned counts are final
  // there: a caller can queue a read-back of bucket boundaries that completes while the scatter kernel still runs)
  void (*after_scan)(void* arg, hipStream_t st) = nullptr;
  // optional: called by rs_sort in front of every pass but the first (the scatter kernel of the pass before is then the last
  // thing queued on the stream)
  void (*before_pass)(void* arg, hipStream_t st) = nullptr;
  void* after_scan_arg = nullptr;        // (the argument of both)
  void rec(hipStream_t st, bool scatter) const {
    std::vector<hipEvent_t>* ev = scatter ? ev_scatter : ev_hist;
    if (!ev || !event_probe_on(scatter ? k_scatter : k_hist)) return;
    hipEvent_t e = event_acquire(device);
    if (e) { (void)hipEventRecord(e, st); ev->push_back(e); }
  }
};

template <int BITS>
inline void rs_pass(const uint2* in, uint2* out, const u32* remap, u64 n, int shift, u32* hist, u32* scan_scratch,
                    hipStream_t st, const RsProbe& pr, bool remap_cache, const RsStrided* sd = nullptr) {
  RsPlan p = rs_plan(n);
  if (sd && !sd->cnt && sd->range) {
    // dense input in variable block ranges, histogram supplied by the pass before (RsStrided::range / next_hist)
    // (no histogram kernel: no histogram probe either)
    const u32 nblk = sd->range_nblk;
    u64 ne = (u64)nblk << BITS;
    scan_exclusive<u32, u32>(hist, hist, ne, scan_scratch, st);
    pr.rec(st, true);
    hipLaunchKernelGGL((k_rs_scatter<BITS, false, false>), dim3(nblk), dim3(RSS_THREADS), 0, st, in, remap, n, (u64)0, nblk, shift, hist, out, *sd, 0u);
    pr.rec(st, true);
    return;
  }
  if (sd) {
    // tile-strided input: blocks own tiles, not tuples
    const RsTilePlan tp = rs_tile_plan(sd->n_tiles);
    if (!sd->hist_ready) {
      pr.rec(st, false);
      if (remap && remap_cache)
        hipLaunchKernelGGL((k_rs_hist_strided<BITS, true>), dim3(tp.nblk), dim3(RS_THREADS), 0, st, in, *sd, remap, tp.tpb, tp.nblk, shift, hist);
      else
        hipLaunchKernelGGL((k_rs_hist_strided<BITS, false>), dim3(tp.nblk), dim3(RS_THREADS), 0, st, in, *sd, remap, tp.tpb, tp.nblk, shift, hist);
      pr.rec(st, false);
    }
    u64 ne = (u64)tp.nblk << BITS;
    scan_exclusive<u32, u32>(hist, hist, ne, scan_scratch, st);
    if (pr.after_scan) pr.after_scan(pr.after_scan_arg, st);
    pr.rec(st, true);
    if (remap && remap_cache)
      hipLaunchKernelGGL((k_rs_scatter<BITS, true, true>), dim3(tp.nblk), dim3(RSS_THREADS), 0, st, in, remap, n, (u64)0, tp.nblk, shift, hist, out, *sd,
                         tp.tpb);
    else
      hipLaunchKernelGGL((k_rs_scatter<BITS, false, true>), dim3(tp.nblk), dim3(RSS_THREADS), 0, st, in, remap, n, (u64)0, tp.nblk, shift, hist, out, *sd,
                         tp.tpb);
    pr.rec(st, true);
    return;
  }
  pr.rec(st, false);
  if (remap && remap_cache)
    hipLaunchKernelGGL((k_rs_hist<BITS, true>), dim3(p.nblk), dim3(RS_THREADS), 0, st, in, remap, n, p.ipb, p.nblk, shift, hist);
  else
    hipLaunchKernelGGL((k_rs_hist<BITS, false>), dim3(p.nblk), dim3(RS_THREADS), 0, st, in, remap, n, p.ipb, p.nblk, shift, hist);
  pr.rec(st, false);
  u64 ne = (u64)p.nblk << BITS;
  scan_exclusive<u32, u32>(hist, hist, ne, scan_scratch, st);
  pr.rec(st, true);
  if (remap && remap_cache)
    hipLaunchKernelGGL((k_rs_scatter<BITS, true, false>), dim3(p.nblk), dim3(RSS_THREADS), 0, st, in, remap, n, p.ipb, p.nblk, shift, hist, out,
                       RsStrided{nullptr, 0, 0}, 0u);
  else
    hipLaunchKernelGGL((k_rs_scatter<BITS, false, false>), dim3(p.nblk), dim3(RSS_THREADS), 0, st, in, remap, n, p.ipb, p.nblk, shift, hist, out,
                       RsStrided{nullptr, 0, 0}, 0u);
  pr.rec(st, true);
}

inline void rs_pass_dyn(int bits, const uint2* in, uint2* out, const u32* remap, u64 n, int shift, u32* hist,
                        u32* scan_scratch, hipStream_t st, const RsProbe& pr, bool remap_cache, const RsStrided* sd = nullptr) {
  switch (bits) {
    case 6: rs_pass<6>(in, out, remap, n, shift, hist, scan_scratch, st, pr, remap_cache, sd); break;
    case 7: rs_pass<7>(in, out, remap, n, shift, hist, scan_scratch, st, pr, remap_cache, sd); break;
    case 8: rs_pass<8>(in, out, remap, n, shift, hist, scan_scratch, st, pr, remap_cache, sd); break;
    case 9: rs_pass<9>(in, out, remap, n, shift, hist, scan_scratch, st, pr, remap_cache, sd); break;
    default: rs_pass<10>(in, out, remap, n, shift, hist, scan_scratch, st, pr, remap_cache, sd); break;
  }
}

// Plans the digit widths for `total_bits` significant key bits: passes of 6..10 bits, <= 9 preferred.
struct RsSchedule {
  int npass;
  int width[8];
};
inline RsSchedule rs_schedule(int total_bits, int max_width = 9) {
  RsSchedule s;
  if (total_bits < 1) total_bits = 1;
  s.npass = (total_bits + max_width - 1) / max_width;
  int rem = total_bits;
  for (int i = 0; i < s.npass; i++) {
    // narrower digits first: the first pass scatters text-order input (short digit runs, fragmented stores), the
    // later ones see skewed high digits and long runs (17 bits: 8 + 9 measured faster than 9 + 8)
    int wdt = i + 1 < s.npass ? rem / (s.npass - i) : rem;
    if (wdt < 6) wdt = 6;
    if (wdt > 10) wdt = 10;
    s.width[i] = wdt;
    rem -= wdt;
    if (rem < 0) rem = 0;
  }
  return s;
}

// Sorts n pairs by key bits [0, total_bits) (stable).  buf_a holds the input; buf_b is scratch of
// the same size.  Returns the buffer that holds the result.  After the first pass the keys are the
// remapped ones (remap applied once).  hist: >= RS_MAX_BLOCKS << 10 u32; scan_scratch: >=
// scan_scratch_elems(RS_MAX_BLOCKS << 10) u32.  remap_cache: the caller guarantees rs_cache_ok(key range, value range).
// strided: buf_a is tile-strided (RsStrided) instead of dense (the first pass reads it in place); it must also hold n
// dense tuples, because later passes ping-pong between buf_b and buf_a.
inline uint2* rs_sort(uint2* buf_a, uint2* buf_b, const u32* remap, u64 n, int total_bits, u32* hist, u32* scan_scratch,
                      hipStream_t st, int* passes_out = nullptr, const RsProbe& pr = RsProbe(), int max_width = 9,
                      bool remap_cache = false, const RsStrided* strided = nullptr) {
  RsSchedule sc = rs_schedule(total_bits, max_width);
  int shift = 0;
  uint2 *in = buf_a, *out = buf_b;
  for (int i = 0; i < sc.npass; i++) {
    if (i && pr.before_pass) pr.before_pass(pr.after_scan_arg, st);
    rs_pass_dyn(sc.width[i], in, out, i == 0 ? remap : nullptr, n, shift, hist, scan_scratch, st, pr, remap_cache,
                i == 0 ? strided : nullptr);
    shift += sc.width[i];
    uint2* t = in; in = out; out = t;
  }
  if (passes_out) *passes_out = sc.npass;
  return in;
}

}  // namespace dcd

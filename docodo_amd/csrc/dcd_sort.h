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
// -> coalesced HBM stores.  Per pass and item: 8 B read (hist) + 8 B read + 8 B write (scatter).
#pragma once
#include "dcd_common.h"
#include "dcd_prims.h"
#include <vector>

#include "dcd_scan.h"

namespace dcd {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
#ifndef DCD_RS_ROWS
#define DCD_RS_ROWS 8
#endif
constexpr int RS_ROWS = DCD_RS_ROWS;               // 64-item rows per wave and tile
constexpr int RS_TILE = RS_THREADS * RS_ROWS;       // 2048 items
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

template <int BITS>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist(const uint2* __restrict__ in, const u32* __restrict__ remap, u64 n, u64 ipb,
                                                       u32 nblk, int shift, u32* __restrict__ hist) {
  constexpr int NB = 1 << BITS;
  __shared__ u32 h[RS_WAVES][NB];
  for (int i = threadIdx.x; i < RS_WAVES * NB; i += RS_THREADS) (&h[0][0])[i] = 0;
  __syncthreads();
  const u32 w = threadIdx.x >> 6;
  u64 beg = (u64)blockIdx.x * ipb, end = beg + ipb < n ? beg + ipb : n;
  // Zipf keys: a plain per-item LDS atomic serialises on the hot digit (13.5 % of all items carry the
  // top term).  Lanes holding the same digit are found with 64-lane ballots (match-any) and only the
  // lowest of them adds the group's size.  Four independent loads are issued before the first use so a
  // wave keeps several HBM requests in flight.
  constexpr int UN = 4;
  const u64 span = end > beg ? end - beg : 0;
  for (u64 o = 0; o < span; o += (u64)RS_THREADS * UN) {
    u32 key[UN];
    bool valid[UN];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      u64 i = beg + o + (u64)u * RS_THREADS + threadIdx.x;
      valid[u] = i < end;
      key[u] = valid[u] ? in[i].x : 0u;
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
      u32 d = valid[u] ? (rs_key(key[u], remap) >> shift) & (NB - 1) : 0u;
      u64 peers = ballot64(valid[u]);
#pragma unroll
      for (int b = 0; b < BITS; b++) {
        u64 m = ballot64((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
      }
      if (valid[u] && mbcnt(peers) == 0) atomicAdd(&h[w][d], popc64(peers));
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < NB; d += RS_THREADS) {
    u32 s = 0;
#pragma unroll
    for (int k = 0; k < RS_WAVES; k++) s += h[k][d];
    hist[(u64)d * nblk + blockIdx.x] = s;
  }
}

template <int BITS>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint2* __restrict__ in, const u32* __restrict__ remap, u64 n, u64 ipb,
                                                          u32 nblk, int shift, const u32* __restrict__ hist_scanned,
                                                          uint2* __restrict__ out) {
  constexpr int NB = 1 << BITS;
  __shared__ u32 s_cnt[RS_WAVES][NB];   // per-wave digit counts of the tile -> per-wave exclusive offsets
  __shared__ u32 s_base[NB];            // running global output position per digit for this block
  __shared__ u32 s_dstart[NB];          // start of each digit inside the LDS-sorted tile
  __shared__ uint2 s_tile[RS_TILE];
  __shared__ u32 s_scan[RS_WAVES + 1];
  const u32 lane = lane_id(), w = threadIdx.x >> 6;

  for (int d = threadIdx.x; d < NB; d += RS_THREADS) s_base[d] = hist_scanned[(u64)d * nblk + blockIdx.x];
  u64 beg = (u64)blockIdx.x * ipb, end = beg + ipb < n ? beg + ipb : n;

  for (u64 t0 = beg; t0 < end; t0 += RS_TILE) {
    for (int i = threadIdx.x; i < RS_WAVES * NB; i += RS_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    uint2 item[RS_ROWS];
    u32 dig[RS_ROWS], rk[RS_ROWS];
    // rank inside the wave: rows in order, lanes in order (stable)
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
      u64 i = t0 + (u64)(w * RS_ROWS + r) * 64 + lane;
      bool valid = i < end;
      uint2 it = valid ? in[i] : make_uint2(0xFFFFFFFFu, 0u);
      if (valid) it.x = rs_key(it.x, remap);
      u32 d = (it.x >> shift) & (NB - 1);
      u64 peers = ballot64(valid);
#pragma unroll
      for (int b = 0; b < BITS; b++) {
        u64 m = ballot64((d >> b) & 1u);
        peers &= ((d >> b) & 1u) ? m : ~m;
      }
      u32 prev = valid ? s_cnt[w][d] : 0u;   // count of this digit in earlier rows of this wave
      u32 below = mbcnt(peers);
      if (valid && below == 0) s_cnt[w][d] = prev + popc64(peers);  // leader = lowest peer lane
      item[r] = it;
      dig[r] = valid ? d : 0xFFFFFFFFu;
      rk[r] = prev + below;
    }
    __syncthreads();
    // per digit: exclusive offsets across waves, tile totals
    u32 tot_d[(NB + RS_THREADS - 1) / RS_THREADS];
#pragma unroll
    for (int q = 0; q < (NB + RS_THREADS - 1) / RS_THREADS; q++) {
      int d = threadIdx.x + q * RS_THREADS;
      u32 run = 0;
      if (d < NB) {
#pragma unroll
        for (int k = 0; k < RS_WAVES; k++) { u32 c = s_cnt[k][d]; s_cnt[k][d] = run; run += c; }
      }
      tot_d[q] = run;
    }
    // exclusive scan over digits of the tile totals -> s_dstart
    {
      u32 mine = 0;
#pragma unroll
      for (int q = 0; q < (NB + RS_THREADS - 1) / RS_THREADS; q++) mine += tot_d[q];
      // thread t owns digits {t, t+256, ...}: scan needs digit order, so do it in q-major passes
      u32 carry = 0;
#pragma unroll
      for (int q = 0; q < (NB + RS_THREADS - 1) / RS_THREADS; q++) {
        u32 tot;
        u32 ex = block_scan_excl<u32, RS_WAVES>(tot_d[q], s_scan, &tot);
        int d = threadIdx.x + q * RS_THREADS;
        if (d < NB) s_dstart[d] = carry + ex;
        carry += tot;
      }
      (void)mine;
    }
    __syncthreads();
    // local sorted position -> LDS
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
      if (dig[r] != 0xFFFFFFFFu) {
        u32 pos = s_dstart[dig[r]] + s_cnt[w][dig[r]] + rk[r];
        s_tile[pos] = item[r];
      }
    }
    __syncthreads();
    u32 tile_n = (u32)((end - t0) < (u64)RS_TILE ? (end - t0) : (u64)RS_TILE);
    for (u32 j = threadIdx.x; j < tile_n; j += RS_THREADS) {
      uint2 it = s_tile[j];
      u32 d = (it.x >> shift) & (NB - 1);
      out[(u64)s_base[d] + (j - s_dstart[d])] = it;
    }
    __syncthreads();
    // advance the running bases by this tile's totals
#pragma unroll
    for (int q = 0; q < (NB + RS_THREADS - 1) / RS_THREADS; q++) {
      int d = threadIdx.x + q * RS_THREADS;
      if (d < NB) s_base[d] += tot_d[q];
    }
    __syncthreads();
  }
}

inline u64 rs_hist_elems(u64 n, int bits) { return (u64)rs_plan(n).nblk << bits; }

// optional per-kernel event probes (build stats): pairs of events recorded around the two kernels
struct RsProbe {
  std::vector<hipEvent_t>* ev = nullptr;  // appended: hist_begin, hist_end, scatter_begin, scatter_end per pass
  void rec(hipStream_t st) const {
    if (!ev) return;
    hipEvent_t e;
    if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, st); ev->push_back(e); }
  }
};

template <int BITS>
inline void rs_pass(const uint2* in, uint2* out, const u32* remap, u64 n, int shift, u32* hist, u32* scan_scratch,
                    hipStream_t st, const RsProbe& pr) {
  RsPlan p = rs_plan(n);
  pr.rec(st);
  hipLaunchKernelGGL((k_rs_hist<BITS>), dim3(p.nblk), dim3(RS_THREADS), 0, st, in, remap, n, p.ipb, p.nblk, shift, hist);
  pr.rec(st);
  u64 ne = (u64)p.nblk << BITS;
  scan_exclusive<u32, u32>(hist, hist, ne, scan_scratch, st);
  pr.rec(st);
  hipLaunchKernelGGL((k_rs_scatter<BITS>), dim3(p.nblk), dim3(RS_THREADS), 0, st, in, remap, n, p.ipb, p.nblk, shift, hist, out);
  pr.rec(st);
}

inline void rs_pass_dyn(int bits, const uint2* in, uint2* out, const u32* remap, u64 n, int shift, u32* hist,
                        u32* scan_scratch, hipStream_t st, const RsProbe& pr) {
  switch (bits) {
    case 6: rs_pass<6>(in, out, remap, n, shift, hist, scan_scratch, st, pr); break;
    case 7: rs_pass<7>(in, out, remap, n, shift, hist, scan_scratch, st, pr); break;
    case 8: rs_pass<8>(in, out, remap, n, shift, hist, scan_scratch, st, pr); break;
    case 9: rs_pass<9>(in, out, remap, n, shift, hist, scan_scratch, st, pr); break;
    default: rs_pass<10>(in, out, remap, n, shift, hist, scan_scratch, st, pr); break;
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
    int wdt = (rem + (s.npass - i) - 1) / (s.npass - i);
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
// scan_scratch_elems(RS_MAX_BLOCKS << 10) u32.
inline uint2* rs_sort(uint2* buf_a, uint2* buf_b, const u32* remap, u64 n, int total_bits, u32* hist, u32* scan_scratch,
                      hipStream_t st, int* passes_out = nullptr, const RsProbe& pr = RsProbe(), int max_width = 9) {
  RsSchedule sc = rs_schedule(total_bits, max_width);
  int shift = 0;
  uint2 *in = buf_a, *out = buf_b;
  for (int i = 0; i < sc.npass; i++) {
    rs_pass_dyn(sc.width[i], in, out, i == 0 ? remap : nullptr, n, shift, hist, scan_scratch, st, pr);
    shift += sc.width[i];
    uint2* t = in; in = out; out = t;
  }
  if (passes_out) *passes_out = sc.npass;
  return in;
}

}  // namespace dcd

// dcd_prims.h -- device primitives for gfx950: 64-lane wave scans / ballots, block scans through
// LDS, and the 15-bit var-ushort unit arithmetic shared by codec, pack and query kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

#define DCD_WAVE 64

namespace dcd {

__device__ __forceinline__ u32 lane_id() { return __lane_id(); }

// lanes strictly below this lane
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

__device__ __forceinline__ u32 popc64(u64 m) { return (u32)__popcll(m); }

// number of set bits of m at lanes below the caller (v_mbcnt_lo/hi)
__device__ __forceinline__ u32 mbcnt(u64 m) {
  return __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

__device__ __forceinline__ u64 ballot64(bool p) { return __ballot(p); }

// inclusive wave scan (sum)
__device__ __forceinline__ u32 wave_scan_incl(u32 v) {
#pragma unroll
  for (int d = 1; d < DCD_WAVE; d <<= 1) {
    u32 t = __shfl_up(v, d, DCD_WAVE);
    if ((int)lane_id() >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ u64 wave_scan_incl(u64 v) {
#pragma unroll
  for (int d = 1; d < DCD_WAVE; d <<= 1) {
    u64 t = __shfl_up(v, d, DCD_WAVE);
    if ((int)lane_id() >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, DCD_WAVE);
  return v;
}
__device__ __forceinline__ u64 wave_sum(u64 v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, DCD_WAVE);
  return v;
}
__device__ __forceinline__ u32 wave_max(u32 v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { u32 t = __shfl_xor(v, d, DCD_WAVE); v = t > v ? t : v; }
  return v;
}

// Workgroup barrier that orders LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() also drains
// vmcnt, i.e. it would wait for every scattered store (and prefetched load) of the tile before the next phase starts.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Block-wide exclusive scan of one value per thread.  NW = waves per block.  `lds` needs NW+1
// entries of T.  Returns the exclusive prefix; *total receives the block sum (all threads).
template <typename T, int NW>
__device__ __forceinline__ T block_scan_excl(T v, T* lds, T* total) {
  const u32 lane = lane_id(), w = threadIdx.x >> 6;
  T incl = wave_scan_incl(v);
  if (lane == DCD_WAVE - 1) lds[w] = incl;
  __syncthreads();
  if (w == 0) {
    T s = lane < (u32)NW ? lds[lane] : (T)0;
    T si = wave_scan_incl(s);
    if (lane < (u32)NW) lds[lane] = si - s;  // exclusive per-wave base
    if (lane == NW - 1) lds[NW] = si;
  }
  __syncthreads();
  T base = lds[w];
  *total = lds[NW];
  __syncthreads();
  return base + incl - v;
}

// ---- 15-bit var-ushort code (IndexSequence.cs:13-15, 63-84) --------------------------------
// number of ushort units the delta d occupies: 1 + [d >= 2^15] + [d >= 2^30] + [d >= 2^45] + [d >= 2^60]
__device__ __forceinline__ u32 seq_nunits(u64 d) {
  return 1u + (d >= (1ull << 15)) + (d >= (1ull << 30)) + (d >= (1ull << 45)) + (d >= (1ull << 60));
}
__device__ __forceinline__ u32 seq_nunits32(u32 d) { return 1u + (d >= (1u << 15)) + (d >= (1u << 30)); }
// unit #g (0-based, g < n) of delta d occupying n units: low group first, bit 15 = "more follow"
__device__ __forceinline__ u16 seq_unit(u64 d, u32 g, u32 n) {
  u32 grp = (u32)(d >> (15u * g)) & 0x7FFFu;
  return (u16)(grp | (g + 1 < n ? 0x8000u : 0u));
}

// ---- term ids handed out by a numbering vocabulary table (vocab_lookup_insert_ids, dcd_tokenize.h) --------------------------
// id = local * VOC_SHARDS + shard: dense per shard.  One address takes ~88 atomics/us on this chip and the vocabulary grows
// fastest at the start of the emit kernel, when every wave inserts: 8 counters made the inserting lanes queue for 0.4 ms
// (measured); 64 counters, each in a 4 KB block of its own (VOC_CTR_STRIDE words apart), do not.
#ifndef DCD_VOC_SHARDS
#define DCD_VOC_SHARDS 64
#endif
constexpr u32 VOC_SHARDS = DCD_VOC_SHARDS;
constexpr u32 VOC_CTR_STRIDE = 1024;
static_assert((VOC_SHARDS & (VOC_SHARDS - 1)) == 0, "VOC_SHARDS must be a power of two");
struct IdCounts { u32 c[VOC_SHARDS]; };
__host__ __device__ __forceinline__ u32 dense_of_id(const u32* __restrict__ c, u32 id) {
  u32 d = 0;
#pragma unroll 8
  for (u32 s = 0; s < VOC_SHARDS; s++) { const u32 m = (id + (VOC_SHARDS - 1u) - s) / VOC_SHARDS; const u32 n = c[s]; d += m < n ? m : n; }
  return d;
}
__host__ __device__ __forceinline__ u32 dense_of_id(const IdCounts& n, u32 id) { return dense_of_id(n.c, id); }
// ... with a hot set: the ids below n_hot belong to the hot terms (all of them in use), the table numbers the others from n_hot
__host__ __device__ __forceinline__ u32 dense_of_key(const u32* __restrict__ c, u32 n_hot, u32 id) {
  return id < n_hot ? id : n_hot + dense_of_id(c, id - n_hot);
}

}  // namespace dcd

// dcd_scan.h -- device-wide exclusive prefix sum (reduce -> single-block scan of partials ->
// apply).  HBM traffic: 2 reads + 1 write of the array; the partial array is tiny.
#pragma once
#include "dcd_common.h"
#include "dcd_prims.h"

namespace dcd {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <typename TIn, typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const TIn* __restrict__ in, u64 n, T* __restrict__ partial) {
  __shared__ T lds[8];
  u64 base = (u64)blockIdx.x * SCAN_TILE;
  T s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    u64 i = base + (u64)j * SCAN_THREADS + threadIdx.x;
    if (i < n) s += (T)in[i];
  }
  s = wave_sum(s);
  if (lane_id() == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// one block; partial[0..nb) -> exclusive in place, partial[nb] = grand total.  Thread t owns 8 consecutive elements
// of every 8192-element round (one block scan and three barriers per round instead of per 1024 elements: the pack
// stage scans ~67 k tile totals through here).
template <typename T>
__global__ __launch_bounds__(1024) void k_scan_partials(T* __restrict__ partial, u64 nb) {
  constexpr int PER = 8;
  __shared__ T lds[17];
  T carry = 0;
  for (u64 b0 = 0; b0 < nb; b0 += 1024 * PER) {
    const u64 i0 = b0 + (u64)threadIdx.x * PER;
    T v[PER], s = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) { v[j] = i0 + j < nb ? partial[i0 + j] : (T)0; s += v[j]; }
    T tot;
    T ex = carry + block_scan_excl<T, 16>(s, lds, &tot);
#pragma unroll
    for (int j = 0; j < PER; j++) { if (i0 + j < nb) partial[i0 + j] = ex; ex += v[j]; }
    carry += tot;
  }
  if (threadIdx.x == 0) partial[nb] = carry;
}

template <typename TIn, typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const TIn* __restrict__ in, u64 n, const T* __restrict__ partial,
                                                            T* __restrict__ out) {
  __shared__ T lds[5];
  // blocked arrangement: thread t owns SCAN_ITEMS consecutive elements
  u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  T v[SCAN_ITEMS];
  T s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    u64 i = base + j;
    v[j] = i < n ? (T)in[i] : (T)0;
    s += v[j];
  }
  T tot;
  T ex = block_scan_excl<T, 4>(s, lds, &tot) + partial[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    u64 i = base + j;
    if (i < n) out[i] = ex;
    ex += v[j];
  }
}

// ---- functor-input variants: F(i) -> T computed on the fly (no materialised flag array) --------
template <typename F, typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce_f(F f, u64 n, T* __restrict__ partial) {
  __shared__ T lds[8];
  u64 base = (u64)blockIdx.x * SCAN_TILE;
  T s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    u64 i = base + (u64)j * SCAN_THREADS + threadIdx.x;
    if (i < n) s += f(i);
  }
  s = wave_sum(s);
  if (lane_id() == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
template <typename F, typename T>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply_f(F f, u64 n, const T* __restrict__ partial, T* __restrict__ out) {
  __shared__ T lds[5];
  u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  T v[SCAN_ITEMS];
  T s = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    u64 i = base + j;
    v[j] = i < n ? f(i) : (T)0;
    s += v[j];
  }
  T tot;
  T ex = block_scan_excl<T, 4>(s, lds, &tot) + partial[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    u64 i = base + j;
    if (i < n) out[i] = ex;
    ex += v[j];
  }
}

// scratch: at least scan_scratch_elems(n) elements of T.  The grand total is left in scratch[nb].
inline u64 scan_nblocks(u64 n) { return div_up(n ? n : 1, SCAN_TILE); }
inline u64 scan_scratch_elems(u64 n) { return scan_nblocks(n) + 1; }

// small inputs: ONE workgroup does the whole scan (one launch instead of three; in == out allowed).  The many
// short radix passes of the vocabulary sort are launch-latency bound.
constexpr u64 SCAN_SMALL = 64 * 1024;
template <typename F, typename T>
__global__ __launch_bounds__(1024) void k_scan_small_f(F f, u64 n, T* __restrict__ out, T* __restrict__ total_out) {
  __shared__ T lds[17];
  T carry = 0;
  constexpr u64 PER = 16;
  for (u64 b0 = 0; b0 < n; b0 += 1024 * PER) {
    u64 i0 = b0 + (u64)threadIdx.x * PER;
    T v[PER];
    T s = 0;
#pragma unroll
    for (u64 j = 0; j < PER; j++) { v[j] = i0 + j < n ? (T)f(i0 + j) : (T)0; s += v[j]; }
    T tot;
    T ex = carry + block_scan_excl<T, 16>(s, lds, &tot);
#pragma unroll
    for (u64 j = 0; j < PER; j++) { if (i0 + j < n) out[i0 + j] = ex; ex += v[j]; }
    carry += tot;
  }
  if (threadIdx.x == 0) *total_out = carry;
}
template <typename TIn>
struct ScanLoadF { const TIn* in; __device__ __forceinline__ TIn operator()(u64 i) const { return in[i]; } };

template <typename TIn, typename T>
inline void scan_exclusive(const TIn* in, T* out, u64 n, T* scratch, hipStream_t st) {
  u64 nb = scan_nblocks(n);
  if (n <= SCAN_SMALL) {
    hipLaunchKernelGGL((k_scan_small_f<ScanLoadF<TIn>, T>), dim3(1), dim3(1024), 0, st, ScanLoadF<TIn>{in}, n, out, scratch + nb);
    return;
  }
  hipLaunchKernelGGL((k_scan_reduce<TIn, T>), dim3((u32)nb), dim3(SCAN_THREADS), 0, st, in, n, scratch);
  hipLaunchKernelGGL((k_scan_partials<T>), dim3(1), dim3(1024), 0, st, scratch, nb);
  hipLaunchKernelGGL((k_scan_apply<TIn, T>), dim3((u32)nb), dim3(SCAN_THREADS), 0, st, in, n, scratch, out);
}

template <typename F, typename T>
inline void scan_exclusive_f(const F& f, T* out, u64 n, T* scratch, hipStream_t st) {
  u64 nb = scan_nblocks(n);
  if (n <= SCAN_SMALL) {
    hipLaunchKernelGGL((k_scan_small_f<F, T>), dim3(1), dim3(1024), 0, st, f, n, out, scratch + nb);
    return;
  }
  hipLaunchKernelGGL((k_scan_reduce_f<F, T>), dim3((u32)nb), dim3(SCAN_THREADS), 0, st, f, n, scratch);
  hipLaunchKernelGGL((k_scan_partials<T>), dim3(1), dim3(1024), 0, st, scratch, nb);
  hipLaunchKernelGGL((k_scan_apply_f<F, T>), dim3((u32)nb), dim3(SCAN_THREADS), 0, st, f, n, scratch, out);
}

}  // namespace dcd

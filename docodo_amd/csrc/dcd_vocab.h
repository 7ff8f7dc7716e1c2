// dcd_vocab.h -- vocabulary finalisation shared by the build and the shard merge: dense ids for the
// used hash-table slots, ordinal (String.CompareOrdinal, Index.cs:142-151) sort of the key strings by
// LSD radix over big-endian character chunks, rank per slot, key pool in rank order.
#pragma once
#include "dcd_common.h"
#include "dcd_scan.h"
#include "dcd_sort.h"
#include "dcd_tokenize.h"

namespace dcd {

struct SlotUsed {
  const u64* slots;
  __device__ __forceinline__ u32 operator()(u64 i) const { return slots[i] != 0 ? 1u : 0u; }
};

static __global__ void k_vocab_compact(const u64* __restrict__ slots, u64 cap, const u32* __restrict__ excl, u64* __restrict__ term_word,
                                u32* __restrict__ term_slot) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  u64 w = slots[i];
  if (w) {
    u32 d = excl[i];
    term_word[d] = w;
    term_slot[d] = (u32)i;
  }
}
// longest key: one atomic per workgroup over the V compacted terms (never one per term: a single
// address takes ~88 atomics/us on this chip)
static __global__ __launch_bounds__(256) void k_max_len(const u64* __restrict__ term_word, u64 n, u32* __restrict__ maxlen) {
  __shared__ u32 s[4];
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 l = i < n ? voc_len(term_word[i]) : 0u;
  l = wave_max(l);
  if (lane_id() == 0) s[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(maxlen, max(max(s[0], s[1]), max(s[2], s[3])));
}

// sort key chunk: characters [c*K, c*K+K) of the term, big-endian packed (K = 4 for bytes, 2 for
// UTF-16 units) so that integer order == String.CompareOrdinal order; missing characters are 0.
template <typename CharT>
__global__ void k_term_keys(uint2* __restrict__ perm, u64 n, const u64* __restrict__ term_word, TextView<CharT> tv, u32 chunk,
                            bool init) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 t = init ? (u32)i : perm[i].y;
  u64 w = term_word[t];
  u32 len = voc_len(w);
  u64 pos = voc_pos(w);
  constexpr u32 K = sizeof(CharT) == 1 ? 4 : 2;
  u32 key = 0;
#pragma unroll
  for (u32 j = 0; j < K; j++) {
    u32 ci = chunk * K + j;
    u32 ch = ci < len ? tv.canon_char(pos + ci) : 0u;
    key = (key << (32 / K)) | ch;
  }
  perm[i] = make_uint2(key, t);
}

static __global__ void k_assign_rank(const uint2* __restrict__ perm, u64 n, const u32* __restrict__ term_slot, const u64* __restrict__ term_word,
                              u32* __restrict__ rank_of_slot, u64* __restrict__ rk_word) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  u32 d = perm[r].y;
  rank_of_slot[term_slot[d]] = (u32)r;
  rk_word[r] = term_word[d];
}

struct TermLenF {
  const u64* rk_word;
  __device__ __forceinline__ u32 operator()(u64 r) const { return voc_len(rk_word[r]); }
};

template <typename CharT>
__global__ void k_term_strings(const u64* __restrict__ rk_word, u64 n, const u32* __restrict__ str_off, TextView<CharT> tv,
                               u16* __restrict__ out) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  u64 w = rk_word[r];
  u32 len = voc_len(w);
  u64 pos = voc_pos(w);
  u32 o = str_off[r];
  for (u32 j = 0; j < len; j++) out[o + j] = (u16)tv.canon_char(pos + j);
}


// Result of vocab_rank: keys in ordinal order.
struct VocabRanks {
  u64 V = 0;
  DevBuf<u32> rank_of_slot;   // [capacity] valid at used slots
  DevBuf<u64> rk_word;        // [V] slot word {tag,len,pos} of the term with rank r
};

// slots[0..cap) is a filled table, slot_excl its exclusive "used" scan (dense ids), V the number of
// used slots.  d_hist / scan_scr32 as rs_sort needs them; d_maxlen: one zeroed u32 on the device.
template <typename CharT>
inline int32_t vocab_rank(int dev, hipStream_t st, const u64* slots, u64 cap, const u32* slot_excl, u64 V, const TextView<CharT>& tv,
                          u32* d_hist, u32* scan_scr32, u32* d_maxlen, VocabRanks* out) {
  out->V = V;
  DevBuf<u64> term_word;
  DevBuf<u32> term_slot;
  DevBuf<uint2> permA, permB;
  DCD_TRY(term_word.alloc(dev, V ? V : 1));
  DCD_TRY(out->rk_word.alloc(dev, V ? V : 1));
  DCD_TRY(term_slot.alloc(dev, V ? V : 1));
  DCD_TRY(out->rank_of_slot.alloc(dev, cap));
  DCD_TRY(permA.alloc(dev, V ? V : 1));
  DCD_TRY(permB.alloc(dev, V ? V : 1));
  if (!V) return DCD_OK;
  u32 maxlen = 0;
  hipLaunchKernelGGL(k_vocab_compact, dim3((u32)div_up(cap, 256)), dim3(256), 0, st, slots, cap, slot_excl, term_word.p, term_slot.p);
  hipLaunchKernelGGL(k_max_len, dim3((u32)div_up(V, 256)), dim3(256), 0, st, term_word.p, V, d_maxlen);
  DCD_HIP(hipMemcpyAsync(&maxlen, d_maxlen, 4, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipStreamSynchronize(st));
  constexpr u32 K = sizeof(CharT) == 1 ? 4 : 2;
  const u32 nchunks = (maxlen + K - 1) / K;
  uint2* cur = permA.p;
  uint2* other = permB.p;
  for (int c = (int)nchunks - 1; c >= 0; c--) {
    hipLaunchKernelGGL((k_term_keys<CharT>), dim3((u32)div_up(V, 256)), dim3(256), 0, st, cur, V, term_word.p, tv, (u32)c,
                       c == (int)nchunks - 1);
    uint2* res = rs_sort(cur, other, nullptr, V, 32, d_hist, scan_scr32, st);
    if (res != cur) { other = cur; cur = res; }
  }
  hipLaunchKernelGGL(k_assign_rank, dim3((u32)div_up(V, 256)), dim3(256), 0, st, cur, V, term_slot.p, term_word.p, out->rank_of_slot.p,
                     out->rk_word.p);
  DCD_HIP(hipStreamSynchronize(st));   // the temporaries above are released when this function returns
  return DCD_OK;
}

}  // namespace dcd

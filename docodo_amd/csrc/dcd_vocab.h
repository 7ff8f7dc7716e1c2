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
  u32 stride = 1;      // u64 units per slot (VocabTable::stride)
  __device__ __forceinline__ u32 operator()(u64 i) const { return slots[i * stride] != 0 ? 1u : 0u; }
};

// (n_out: size of the output arrays -- a caller that runs this before it knows V passes a bound; ids beyond it belong to a
// table that is about to be discarded for being too full)
static __global__ void k_vocab_compact(const u64* __restrict__ slots, u64 cap, const u32* __restrict__ excl, u64* __restrict__ term_word,
                                u32* __restrict__ term_slot, u64 n_out, u32 stride) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  u64 w = slots[i * stride];
  if (w) {
    u32 d = excl[i];
    if (d < n_out) {
      term_word[d] = w;
      term_slot[d] = (u32)i;
    }
  }
}
// dense arrays of the vocabulary from a numbering table (the counterpart of k_vocab_compact)
// id_ctr: the VOC_SHARDS counters, VOC_CTR_STRIDE words apart; counts[VOC_SHARDS + 1] receives them densely + V = their sum
// + *n_hot (the hot terms hold the ids below *n_hot, the counters number the others from there: dense_of_key)
static __global__ __launch_bounds__(256) void k_vocab_compact_ids(const u64* __restrict__ slots, u64 cap, u32 stride, const u32* __restrict__ id_ctr,
                                    u64* __restrict__ term_word, u32* __restrict__ term_slot, u64 n_out, u32* __restrict__ counts,
                                    const u32* __restrict__ n_hot_dev) {
  __shared__ u32 n[VOC_SHARDS];
  if (threadIdx.x < VOC_SHARDS) n[threadIdx.x] = id_ctr[threadIdx.x * VOC_CTR_STRIDE];
  __syncthreads();
  const u32 n_hot = *n_hot_dev;
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < VOC_SHARDS) counts[threadIdx.x] = n[threadIdx.x];
  if (i == 0) {
    u32 sum = n_hot;
    for (u32 s = 0; s < VOC_SHARDS; s++) sum += n[s];
    counts[VOC_SHARDS] = sum;
  }
  if (i >= cap) return;
  const u64 w = slots[i * stride];
  if (w) {
    const u32 d = dense_of_key(n, n_hot, reinterpret_cast<const u32*>(slots + i * stride)[2] - 1u);
    if (d < n_out) { term_word[d] = w; term_slot[d] = (u32)i; }
  }
}

// longest key: one atomic per workgroup over the V compacted terms (never one per term: a single
// address takes ~88 atomics/us on this chip)
// (n_dev != nullptr: the number of terms is still on the device -- the grid covers an upper bound)
static __global__ __launch_bounds__(256) void k_max_len(const u64* __restrict__ term_word, u64 n, u32* __restrict__ maxlen, const u32* __restrict__ n_dev) {
  __shared__ u32 s[4];
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 l = i < n ? voc_len(term_word[i]) : 0u;
  l = wave_max(l);
  if (lane_id() == 0) s[threadIdx.x >> 6] = l;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(maxlen, max(max(s[0], s[1]), max(s[2], s[3])));
}

// ---- alphabet compaction -------------------------------------------------------------------------------------
// The ordinal string sort is a chain of short radix passes, i.e. launch-latency bound; the number of passes is
// proportional to (key length) x (bits per character).  The characters that actually occur are ranked (order
// preserving, 0 = "past the end"), so a character costs ceil(log2(#distinct + 1)) bits instead of 8 / 16:
// 6 bits for lower-case ASCII + digits -> five characters per 32-bit chunk.
// blockmax != nullptr: the workgroup also leaves the length of its longest key in blockmax[blockIdx.x] (k_char_rank reduces
// them: no atomics, and no k_max_len launch).  Workgroups past the n_dev terms do nothing.
template <typename CharT>
__global__ __launch_bounds__(256) void k_char_used(const u64* __restrict__ term_word, u64 n, TextView<CharT> tv, u32* __restrict__ bitmap,
                                                   const u32* __restrict__ n_dev, u32* __restrict__ blockmax = nullptr) {
  __shared__ u32 s_bits[2048];      // per-workgroup bitmap in LDS; only its non-zero words reach the global one
  __shared__ u32 s_max;
  if (n_dev) n = *n_dev < n ? *n_dev : n;
  if ((u64)blockIdx.x * blockDim.x >= n) return;
  for (u32 x = threadIdx.x; x < 2048; x += 256) s_bits[x] = 0;
  if (threadIdx.x == 0) s_max = 0;
  __syncthreads();
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockmax) {
    const u32 l = wave_max(i < n ? voc_len(term_word[i]) : 0u);
    if (lane_id() == 0) atomicMax(&s_max, l);
  }
  if (i < n) {
    u64 w = term_word[i];
    u32 len = voc_len(w);
    u64 pos = voc_pos(w);
    for (u32 j = 0; j < len; j++) {
      u32 c = (u32)tv.canon_char(pos + j);
      u32 m = 1u << (c & 31);
      if (!(s_bits[c >> 5] & m)) atomicOr(&s_bits[c >> 5], m);
    }
  }
  __syncthreads();
  for (u32 x = threadIdx.x; x < 2048; x += 256) { u32 v = s_bits[x]; if (v) atomicOr(&bitmap[x], v); }
  if (blockmax && threadIdx.x == 0) blockmax[blockIdx.x] = s_max;
}
// one workgroup: rank[c] = 1 + number of used characters below c (0 for unused ones); *n_sym = number of used characters
// blockmax != nullptr: also *maxlen = the largest of the ceil(*n_dev / 256) per-workgroup maxima k_char_used left
static __global__ __launch_bounds__(1024) void k_char_rank(const u32* __restrict__ bitmap, u16* __restrict__ rank, u32* __restrict__ n_sym,
                                                           const u32* __restrict__ blockmax = nullptr, const u32* __restrict__ n_dev = nullptr,
                                                           u64 n_bound = 0, u32* __restrict__ maxlen = nullptr) {
  __shared__ u32 lds[17];
  __shared__ u32 s_max;
  const u32 t = threadIdx.x;
  if (blockmax) {
    if (t == 0) s_max = 0;
    __syncthreads();
    const u64 n = *n_dev < n_bound ? *n_dev : n_bound;
    const u32 nb = (u32)((n + 255) / 256);
    u32 m = 0;
    for (u32 b = t; b < nb; b += 1024) m = max(m, blockmax[b]);
    m = wave_max(m);
    if (lane_id() == 0 && m) atomicMax(&s_max, m);
    __syncthreads();
    if (t == 0) *maxlen = s_max;
  }
  const u32 w0 = bitmap[2 * t], w1 = bitmap[2 * t + 1];
  u32 tot;
  u32 ex = block_scan_excl<u32, 16>((u32)__popc(w0) + (u32)__popc(w1), lds, &tot);
  u32 r = ex;
  for (u32 b = 0; b < 64; b++) {
    const u32 wv = b < 32 ? w0 : w1;
    const bool used = (wv >> (b & 31)) & 1u;
    rank[t * 64 + b] = used ? (u16)(++r) : (u16)0;
  }
  if (t == 0) *n_sym = tot;
}

// sort key chunk: characters [c*cpc, c*cpc+cpc) of the term as B-bit ranks, big-endian packed so that integer
// order == String.CompareOrdinal order; missing characters are 0.
template <typename CharT>
__global__ void k_term_keys(uint2* __restrict__ perm, u64 n, const u64* __restrict__ term_word, TextView<CharT> tv, u32 chunk,
                            bool init, const u16* __restrict__ char_rank, u32 cpc, u32 bits) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 t = init ? (u32)i : perm[i].y;
  u64 w = term_word[t];
  u32 len = voc_len(w);
  u64 pos = voc_pos(w);
  u32 key = 0;
  for (u32 j = 0; j < cpc; j++) {
    u32 ci = chunk * cpc + j;
    u32 ch = ci < len ? (u32)char_rank[(u32)tv.canon_char(pos + ci)] : 0u;
    key = (key << bits) | ch;
  }
  perm[i] = make_uint2(key, t);
}

static __global__ void k_assign_rank(const uint2* __restrict__ perm, u64 n, const u32* __restrict__ term_slot, const u64* __restrict__ term_word,
                              u32* __restrict__ rank_of_slot, u64* __restrict__ rk_word, u32* __restrict__ dense_of_rank) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  u32 d = perm[r].y;
  rank_of_slot[term_slot[d]] = (u32)r;
  rk_word[r] = term_word[d];
  dense_of_rank[r] = d;
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
  DevBuf<u32> dense_of_rank;  // [V] dense id (position among the used slots) of the term with rank r
};

// Vocabulary finalisation in two halves, so that a caller can fetch the two host-side parameters (longest key, alphabet
// size) with a read-back it makes anyway and then queue the whole sort without blocking:
//   vocab_prepare: dense arrays of the used slots, longest key, ranked alphabet       (4 small kernels)
//   vocab_sort:    LSD radix passes over the ranked characters, ranks, dense_of_rank   (no host round trip)
struct VocabPrep {
  DevBuf<u64> term_word;      // [V] slot word of the term with dense id d
  DevBuf<u32> term_slot;      // [V] its slot
  DevBuf<u32> cbits;          // [2048] used-character bitmap, [2048] = number of used characters
  DevBuf<u16> crank;          // [65536] character -> rank (0 = unused)
  const u32* d_maxlen = nullptr;
  const u32* d_nsym = nullptr;
};
// slots[0..cap) is a filled table, slot_excl a bijection used slot -> [0, V) (dense ids), V the number of used slots.
// d_maxlen: one zeroed u32 on the device.
template <typename CharT>
// V_dev != nullptr: V is only an upper bound, the true count is read from the device (the caller has not fetched it yet).
inline int32_t vocab_prepare(int dev, hipStream_t st, const u64* slots, u64 cap, const u32* slot_excl, u64 V, const TextView<CharT>& tv,
                             u32* d_maxlen, VocabPrep* P, const u32* V_dev = nullptr, u32 slot_stride = 1) {
  DCD_TRY(P->term_word.alloc(dev, V ? V : 1));
  DCD_TRY(P->term_slot.alloc(dev, V ? V : 1));
  DCD_TRY(P->cbits.alloc(dev, 2049));
  DCD_TRY(P->crank.alloc(dev, 65536));
  P->d_maxlen = d_maxlen; P->d_nsym = P->cbits.p + 2048;
  DCD_HIP(hipMemsetAsync(P->cbits.p, 0, 2049 * 4, st));
  if (!V) return DCD_OK;
  hipLaunchKernelGGL(k_vocab_compact, dim3((u32)div_up(cap, 256)), dim3(256), 0, st, slots, cap, slot_excl, P->term_word.p, P->term_slot.p, V, slot_stride);
  hipLaunchKernelGGL(k_max_len, dim3((u32)div_up(V, 256)), dim3(256), 0, st, P->term_word.p, V, d_maxlen, V_dev);
  hipLaunchKernelGGL((k_char_used<CharT>), dim3((u32)div_up(V, 256)), dim3(256), 0, st, P->term_word.p, V, tv, P->cbits.p, V_dev);
  hipLaunchKernelGGL(k_char_rank, dim3(1), dim3(1024), 0, st, P->cbits.p, P->crank.p, P->cbits.p + 2048);
  return DCD_OK;
}
// The same for a table that numbered its terms (k_vocab_compact_ids), with the launches cut to three: `zeroed` = 2052
// zeroed words (the character bitmap), blockmax = div_up(V_bound, 256) words of scratch, counts[VOC_SHARDS + 1] receives
// the id counts and V, *d_nsym / *d_maxlen the two parameters of the ordinal sort.
template <typename CharT>
inline int32_t vocab_prepare_ids(int dev, hipStream_t st, const u64* slots, u64 cap, u64 V_bound, const TextView<CharT>& tv, VocabPrep* P,
                                 u32 slot_stride, const u32* id_ctr, u32* counts, u32* zeroed, u32* blockmax, u32* d_nsym, u32* d_maxlen,
                                 const u32* n_hot_dev) {
  DCD_TRY(P->term_word.alloc(dev, V_bound ? V_bound : 1));
  DCD_TRY(P->term_slot.alloc(dev, V_bound ? V_bound : 1));
  DCD_TRY(P->crank.alloc(dev, 65536));
  P->d_maxlen = d_maxlen; P->d_nsym = d_nsym;
  const u32* V_dev = counts + VOC_SHARDS;
  hipLaunchKernelGGL(k_vocab_compact_ids, dim3((u32)div_up(cap, 256)), dim3(256), 0, st, slots, cap, slot_stride, id_ctr, P->term_word.p,
                     P->term_slot.p, V_bound, counts, n_hot_dev);
  hipLaunchKernelGGL((k_char_used<CharT>), dim3((u32)div_up(V_bound ? V_bound : 1, 256)), dim3(256), 0, st, P->term_word.p, V_bound, tv, zeroed, V_dev,
                     blockmax);
  hipLaunchKernelGGL(k_char_rank, dim3(1), dim3(1024), 0, st, (const u32*)zeroed, P->crank.p, d_nsym, (const u32*)blockmax, V_dev, V_bound, d_maxlen);
  return DCD_OK;
}
// d_hist / scan_scr32 as rs_sort needs them.  Nothing in here blocks the host.
template <typename CharT>
inline int32_t vocab_sort(int dev, hipStream_t st, u64 cap, u64 V, const TextView<CharT>& tv, const VocabPrep& P, u32 maxlen, u32 n_sym,
                          u32* d_hist, u32* scan_scr32, VocabRanks* out) {
  out->V = V;
  DevBuf<uint2> permA, permB;
  DCD_TRY(out->rk_word.alloc(dev, V ? V : 1));
  DCD_TRY(out->rank_of_slot.alloc(dev, cap));
  DCD_TRY(out->dense_of_rank.alloc(dev, V ? V : 1));
  DCD_TRY(permA.alloc(dev, V ? V : 1));
  DCD_TRY(permB.alloc(dev, V ? V : 1));
  if (!V) return DCD_OK;
  const u32 bits = std::max<u32>(1, log2_ceil((u64)n_sym + 1));     // ranks 0 .. n_sym
  // measured on MI355X (V = 1e5): 8-bit passes over 32-bit chunks 0.35 ms, 9 over 27: 0.35, 10 over 30: 0.56; before (8 or 16 bits per character): 0.40
  // characters per chunk: the choice that needs the fewest radix passes for this vocabulary (every pass is three small
  // launches and the stage is launch-bound): chunks of cpc characters, each sorted with ceil(cpc * bits / VOC_WIDTH) passes
  constexpr int VOC_WIDTH = 8, VOC_CHUNK = 32;
  u32 cpc = 1, best = ~0u;
  for (u32 c = 1; c * bits <= (u32)VOC_CHUNK; c++) {
    const u32 passes = ((maxlen + c - 1) / c) * ((c * bits + VOC_WIDTH - 1) / VOC_WIDTH);
    if (passes <= best) { best = passes; cpc = c; }       // ties: fewer, larger chunks (fewer key kernels)
  }
  const u32 nchunks = (maxlen + cpc - 1) / cpc;
  uint2* cur = permA.p;
  uint2* other = permB.p;
  for (int c = (int)nchunks - 1; c >= 0; c--) {
    hipLaunchKernelGGL((k_term_keys<CharT>), dim3((u32)div_up(V, 256)), dim3(256), 0, st, cur, V, P.term_word.p, tv, (u32)c,
                       c == (int)nchunks - 1, P.crank.p, cpc, bits);
    uint2* res = rs_sort(cur, other, nullptr, V, (int)(cpc * bits), d_hist, scan_scr32, st, nullptr, RsProbe(), VOC_WIDTH);
    if (res != cur) { other = cur; cur = res; }
  }
  hipLaunchKernelGGL(k_assign_rank, dim3((u32)div_up(V, 256)), dim3(256), 0, st, cur, V, P.term_slot.p, P.term_word.p, out->rank_of_slot.p,
                     out->rk_word.p, out->dense_of_rank.p);
  // (no synchronisation: the temporaries above go back to the stream-ordered device cache, which hands them to another
  // stream only once this one has passed this point)
  return DCD_OK;
}
// both halves with the host round trip in between
template <typename CharT>
inline int32_t vocab_rank(int dev, hipStream_t st, const u64* slots, u64 cap, const u32* slot_excl, u64 V, const TextView<CharT>& tv,
                          u32* d_hist, u32* scan_scr32, u32* d_maxlen, VocabRanks* out, u32* maxlen_out = nullptr, u32 slot_stride = 1) {
  VocabPrep P;
  DCD_TRY(vocab_prepare<CharT>(dev, st, slots, cap, slot_excl, V, tv, d_maxlen, &P, nullptr, slot_stride));
  u32 maxlen = 0, n_sym = 0;
  if (V) {
    DCD_HIP(hipMemcpyAsync(&maxlen, P.d_maxlen, 4, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipMemcpyAsync(&n_sym, P.d_nsym, 4, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipStreamSynchronize(st));
  }
  if (maxlen_out) *maxlen_out = maxlen;
  return vocab_sort<CharT>(dev, st, cap, V, tv, P, maxlen, n_sym, d_hist, scan_scr32, out);
}

}  // namespace dcd

// dcd_codec.h -- device kernels of the 15-bit var-ushort delta code (IndexSequence.cs:63-84
// encode, :114-136 decode), shared by the codec entry points, the build's pack stage and the
// query path.  All work is HBM-streaming: reduce -> scan of per-block partials -> apply.
#pragma once
#include "dcd_common.h"
#include "dcd_prims.h"
#include "dcd_scan.h"

namespace dcd {

constexpr int CODEC_THREADS = 256;
#ifndef DCD_CODEC_ITEMS
#define DCD_CODEC_ITEMS 8
#endif
constexpr int CODEC_ITEMS = DCD_CODEC_ITEMS;
constexpr int CODEC_TILE = CODEC_THREADS * CODEC_ITEMS;
#ifndef DCD_PACK_TPB
#define DCD_PACK_TPB 1             // measured (round 2): one tile per workgroup compiled for 8 waves / SIMD 0.32 ms; four tiles
#endif                             // with the next tile prefetched (163 VGPRs, 3 waves / SIMD) 0.42 ms
constexpr int PACK_TPB = DCD_PACK_TPB;               // tiles per workgroup of k_pack_write

// ---- delta sources for the encoder -----------------------------------------------------------
// Src::delta(i) -> u64 delta of item i; Src::on_item(i, unit_offset, delta) -> side outputs.
struct EncSrcValues {  // plain ascending values, first delta from 0 (IndexSequence.Builder.Add)
  const u64* v;
  __device__ __forceinline__ u64 delta(u64 i) const { return v[i] - (i ? v[i - 1] : 0ull); }
  __device__ __forceinline__ void on_item(u64, u64, u64) const {}
};

// Sorted (term, coord) pairs of the build: the first posting of a term is coded from 0 (absolute
// coordinate = base + local), the others as differences (Build.cs:324-333 -> IndexSequence.cs:63).
struct EncSrcPairs {
  const uint2* p;      // x = term rank, y = segment-local coordinate
  u64 base;            // global coordinate of local 0
  u64* term_unit_start;  // [n_terms+1]
  u64* term_item_start;  // [n_terms+1]
  // hot / cold split keys (RsStrided::hot_n / hot_s, dcd_sort.h): x is the transformed key, the directory is indexed by the dense id
  u32 hot_n = 0, hot_s = 0;
  // ... or ids of a numbering vocabulary table (id_dense != nullptr): the directory is indexed by the dense id = number of used
  // ids below x, tabulated (term heads only gather from it)
  const u32* id_dense = nullptr;
  // inverse of rs_hot_key (dcd_sort.h)
  static __host__ __device__ __forceinline__ u32 hot_key_id(u32 x, u32 hot_n, u32 hot_s) {
    const u32 hi = x >> hot_s;
    return hi < hot_n ? hi : hot_n + (x - (hot_n << hot_s));
  }
  // round 4: the first n_hot32 postings (the hot terms' buckets of pass A, RsStrided::hot32) are stored as 4-byte coordinates at
  // ((const u32*)p)[i]; their key is the bucket: hot term d owns [hstart[d * hstride], hstart[(d + 1) * hstride]) -- the scanned
  // histogram of pass A, entry (digit d, block 0)
  u64 n_hot32 = 0;
  const u32* hstart = nullptr;
  u32 hstride = 0;
  // per tile of the pack kernels that starts at or before n_hot32 (k_pack_hot_tiles): bucket + 1 if the tile is full, lies inside
  // ONE hot bucket together with the posting before it (no term head in it, every key the bucket's), else 0 (posting by posting)
  const u32* hot_tile = nullptr;
  __device__ __forceinline__ u32 id_of(u32 x) const {
    const u32 id = hot_s ? hot_key_id(x, hot_n, hot_s) : x;
    return id_dense ? id_dense[id] : id;
  }
  __device__ __forceinline__ u32 hot_bucket(u64 i) const {       // the last hot term whose bucket starts at or before posting i
    u32 lo = 0, hi = hot_n;
    while (lo + 1 < hi) { const u32 m = (lo + hi) >> 1; if ((u64)hstart[(u64)m * hstride] <= i) lo = m; else hi = m; }
    return lo;
  }
  __device__ __forceinline__ uint2 at(u64 i) const {             // (key, coordinate) of posting i, wherever it is stored
    if (i < n_hot32) return make_uint2(hot_bucket(i) << hot_s, reinterpret_cast<const u32*>(p)[i]);
    return p[i];
  }
  __device__ __forceinline__ bool head(u64 i) const { return i == 0 || at(i).x != at(i - 1).x; }
  __device__ __forceinline__ u64 delta(u64 i) const {
    uint2 c = at(i);
    if (i == 0) return base + c.y;
    uint2 q = at(i - 1);
    return q.x != c.x ? base + c.y : (u64)c.y - (u64)q.y;
  }
  __device__ __forceinline__ void on_item(u64 i, u64 off, u64) const {
    if (head(i)) { const u32 t = id_of(at(i).x); term_unit_start[t] = off; term_item_start[t] = i; }
  }
};

template <typename Src>
__global__ __launch_bounds__(CODEC_THREADS) void k_enc_reduce(Src src, u64 n, u64* __restrict__ partial) {
  __shared__ u32 lds[4];
  u64 base = (u64)blockIdx.x * CODEC_TILE;
  u32 s = 0;
#pragma unroll
  for (int j = 0; j < CODEC_ITEMS; j++) {
    u64 i = base + (u64)j * CODEC_THREADS + threadIdx.x;
    if (i < n) s += seq_nunits(src.delta(i));
  }
  s = wave_sum(s);
  if (lane_id() == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (u64)lds[0] + lds[1] + lds[2] + lds[3];
}

// Writes the units of one 2048-item tile.  Items are taken in blocked order (lane t owns 8
// consecutive items, so one block scan gives every unit offset); the units are first laid out in
// LDS and then copied to HBM by consecutive lanes, so a wave-instruction stores 128 contiguous bytes
// instead of 64 scattered 2-byte words.
template <typename Src>
__global__ __launch_bounds__(CODEC_THREADS) void k_enc_write(Src src, u64 n, const u64* __restrict__ partial,
                                                            u16* __restrict__ out) {
  __shared__ u32 lds[5];
  __shared__ u16 s_units[CODEC_TILE * 5];
  u64 base = (u64)blockIdx.x * CODEC_TILE + (u64)threadIdx.x * CODEC_ITEMS;
  u64 d[CODEC_ITEMS];
  u32 nu[CODEC_ITEMS];
  u32 s = 0;
#pragma unroll
  for (int j = 0; j < CODEC_ITEMS; j++) {
    u64 i = base + j;
    d[j] = i < n ? src.delta(i) : 0;
    nu[j] = i < n ? seq_nunits(d[j]) : 0;
    s += nu[j];
  }
  u32 tot;
  u32 loff = block_scan_excl<u32, 4>(s, lds, &tot);
  const u64 gbase = partial[blockIdx.x];
#pragma unroll
  for (int j = 0; j < CODEC_ITEMS; j++) {
    u64 i = base + j;
    if (i < n) {
      src.on_item(i, gbase + loff, d[j]);
      for (u32 g = 0; g < nu[j]; g++) s_units[loff + g] = seq_unit(d[j], g, nu[j]);
      loff += nu[j];
    }
  }
  __syncthreads();
  // coalesced copy; 32-bit stores where the destination is 4-byte aligned
  u16* dst = out + gbase;
  u32 head = (u32)(((uintptr_t)dst >> 1) & 1u) & (tot ? 1u : 0u);   // one leading u16 to reach alignment
  if (threadIdx.x == 0 && head) dst[0] = s_units[0];
  u32 npair = (tot - head) >> 1;
  u32* dst32 = reinterpret_cast<u32*>(dst + head);
  for (u32 k = threadIdx.x; k < npair; k += CODEC_THREADS)
    dst32[k] = (u32)s_units[head + 2 * k] | ((u32)s_units[head + 2 * k + 1] << 16);
  if (threadIdx.x == 0 && ((tot - head) & 1u)) dst[tot - 1] = s_units[tot - 1];
}

// ---- pack of the build: sorted (term, coord) pairs -> units ------------------------------------------
// Same arithmetic as k_enc_write<EncSrcPairs>, but the tile's pairs are first loaded with coalesced
// 8-byte-per-lane reads into LDS (plus the item before the tile) and the units share that LDS
// afterwards; every HBM access of the kernel is then a full-width wave access.
#ifndef DCD_PACK_WPE
#define DCD_PACK_WPE 8
#endif
__global__ __launch_bounds__(CODEC_THREADS, DCD_PACK_WPE) void k_pack_write(EncSrcPairs src, u64 n, const u64* __restrict__ partial, u16* __restrict__ out);
__global__ __launch_bounds__(CODEC_THREADS) void k_pack_count(EncSrcPairs src, u64 n, u64* __restrict__ partial);   // == k_enc_reduce<EncSrcPairs>
__global__ void k_pack_hot_tiles(EncSrcPairs src, u64 n_tiles_hot, u32* __restrict__ hot_tile);
inline u64 pack_hot_tiles(u64 n_hot32) { return n_hot32 ? n_hot32 / CODEC_TILE + 1 : 0; }      // tiles with t0 <= n_hot32

// Encodes n items.  partial: >= div_up(n, CODEC_TILE) + 1 u64 (grand total of units left in
// partial[nblocks]).  out must hold the total (<= 5 n).
template <typename Src>
inline void encode_launch(const Src& src, u64 n, u64* partial, u16* out, hipStream_t st) {
  u64 nb = div_up(n ? n : 1, CODEC_TILE);
  hipLaunchKernelGGL((k_enc_reduce<Src>), dim3((u32)nb), dim3(CODEC_THREADS), 0, st, src, n, partial);
  hipLaunchKernelGGL((k_scan_partials<u64>), dim3(1), dim3(1024), 0, st, partial, nb);
  hipLaunchKernelGGL((k_enc_write<Src>), dim3((u32)nb), dim3(CODEC_THREADS), 0, st, src, n, partial, out);
}
inline u64 encode_nblocks(u64 n) { return div_up(n ? n : 1, CODEC_TILE); }

// ---- decoder ----------------------------------------------------------------------------------
// A value ends at every unit whose bit 15 is clear.  Step 1 counts terminators per block, step 2
// lets each terminator thread gather its group (walking back over continuation units) into a
// delta, step 3 prefix-sums the deltas (u64, wrapping like the C# ulong).
__global__ void k_dec_count(const u16* __restrict__ u, u64 n, u64* __restrict__ partial);
__global__ void k_dec_deltas(const u16* __restrict__ u, u64 n, const u64* __restrict__ partial, u64* __restrict__ deltas);
__global__ void k_add_inplace(u64* __restrict__ excl, const u64* __restrict__ deltas, u64 n);

// units[0..n) -> values.  scratch_a: >= div_up(n,CODEC_TILE)+1 u64; deltas: >= n_values u64
// (n is always enough); scan_scratch: >= scan_scratch_elems(n) u64.  *n_values_dev = partial[nb].
void decode_count_launch(const u16* units, u64 n, u64* partial, hipStream_t st);   // partial[nb] = #values
void decode_values_launch(const u16* units, u64 n, const u64* partial_scanned, u64 n_values, u64* deltas,
                          u64* values, u64* scan_scratch, hipStream_t st);

}  // namespace dcd

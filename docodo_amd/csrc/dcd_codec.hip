// dcd_codec.hip -- IndexSequence codec entry points (dcd_seq_encode / decode / count / shift).
// Reference: IndexSequence.cs:63-90 (Builder.Add/AddRange), :114-136 (SeqEnumerator.MoveNext),
// :191-202 (Shift).
#include "dcd_codec.h"

namespace dcd {

__global__ __launch_bounds__(CODEC_THREADS) void k_dec_count(const u16* __restrict__ u, u64 n, u64* __restrict__ partial) {
  __shared__ u32 lds[4];
  u64 base = (u64)blockIdx.x * CODEC_TILE;
  u32 s = 0;
#pragma unroll
  for (int j = 0; j < CODEC_ITEMS; j++) {
    u64 i = base + (u64)j * CODEC_THREADS + threadIdx.x;
    if (i < n) s += (u[i] & 0x8000u) ? 0u : 1u;
  }
  s = wave_sum(s);
  if (lane_id() == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (u64)lds[0] + lds[1] + lds[2] + lds[3];
}

__global__ __launch_bounds__(CODEC_THREADS) void k_dec_deltas(const u16* __restrict__ u, u64 n, const u64* __restrict__ partial,
                                                             u64* __restrict__ deltas) {
  __shared__ u32 lds[5];
  u64 base = (u64)blockIdx.x * CODEC_TILE + (u64)threadIdx.x * CODEC_ITEMS;
  u16 v[CODEC_ITEMS];
  u32 cnt = 0;
#pragma unroll
  for (int j = 0; j < CODEC_ITEMS; j++) {
    u64 i = base + j;
    v[j] = i < n ? u[i] : (u16)0x8000;
    cnt += (v[j] & 0x8000u) ? 0u : 1u;
  }
  u32 tot;
  u64 vi = partial[blockIdx.x] + block_scan_excl<u32, 4>(cnt, lds, &tot);
#pragma unroll
  for (int j = 0; j < CODEC_ITEMS; j++) {
    u64 i = base + j;
    if (i < n && !(v[j] & 0x8000u)) {
      // group = maximal run of continuation units ending at i; low group first.  C# shifts a
      // ulong by (15 k) & 63 (IndexSequence.cs:126), reproduced for non-canonical input.
      u64 s = i;
      while (s > 0 && (u[s - 1] & 0x8000u)) s--;
      u64 d = 0;
      u32 sh = 0;
      for (u64 k = s; k <= i; k++) { d += ((u64)(u[k] & 0x7FFFu)) << (sh & 63u); sh += 15; }
      deltas[vi++] = d;
    }
  }
}

__global__ void k_add_inplace(u64* __restrict__ excl, const u64* __restrict__ deltas, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) excl[i] += deltas[i];
}

// Unit count of a tile of the build's sorted pairs (k_enc_reduce<EncSrcPairs> for this source): every lane reads TWO
// pairs per 16-byte load and keeps all four loads of its tile in flight; the pair before a lane's first one comes from
// the neighbouring lane.  Inside a term the delta of two segment-local coordinates fits 32 bits (at most 3 units);
// only the first posting of a term carries the 64-bit base.
__global__ __launch_bounds__(CODEC_THREADS) void k_pack_count(EncSrcPairs src, u64 n, u64* __restrict__ partial) {
  __shared__ u32 lds[4];
  const u64 t0 = (u64)blockIdx.x * CODEC_TILE;
  const u32 lane = lane_id();
  if (src.n_hot32 && t0 <= src.n_hot32) {
    // round 4: a tile of the hot region -- 4-byte coordinates, the term is the bucket.  A full tile inside ONE bucket (the element
    // before it included: by weight all of them -- 249 hot terms hold 65 % of the postings) has no term head: every posting is a
    // 32-bit delta to the coordinate before it; four coordinates per 16-byte load.  The few tiles that hold a bucket border, and the
    // one that holds the border to the cold region, go posting by posting through the general accessor.
    u32 s = 0;
    if (src.hot_tile[blockIdx.x]) {
      const u32* c32 = reinterpret_cast<const u32*>(src.p);
      const uint4* c4 = reinterpret_cast<const uint4*>(c32 + t0);          // (t0 is a multiple of the tile: 16-byte aligned)
      uint4 v[CODEC_ITEMS / 4];
      u32 before[CODEC_ITEMS / 4];
#pragma unroll
      for (int j = 0; j < CODEC_ITEMS / 4; j++) {
        v[j] = c4[j * CODEC_THREADS + threadIdx.x];
        before[j] = lane == 0 ? c32[t0 + 4u * (u32)(j * CODEC_THREADS + threadIdx.x) - 1u] : 0u;
      }
#pragma unroll
      for (int j = 0; j < CODEC_ITEMS / 4; j++) {
        u32 q = __shfl_up(v[j].w, 1, DCD_WAVE);
        if (lane == 0) q = before[j];
        s += seq_nunits32(v[j].x - q) + seq_nunits32(v[j].y - v[j].x) + seq_nunits32(v[j].z - v[j].y) + seq_nunits32(v[j].w - v[j].z);
      }
    } else {
      for (int j = 0; j < CODEC_ITEMS; j++) {
        const u64 i = t0 + (u64)j * CODEC_THREADS + threadIdx.x;
        if (i < n) s += seq_nunits(src.delta(i));
      }
    }
    s = wave_sum(s);
    if (lane == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (u64)lds[0] + lds[1] + lds[2] + lds[3];
    return;
  }
  const uint4* p2 = reinterpret_cast<const uint4*>(src.p);      // t0 is even and src.p 16-byte aligned (checked by the host)
  constexpr int NL = CODEC_ITEMS / 2;
  uint4 kv[NL];
  uint2 before[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const u64 i = t0 + 2 * ((u64)j * CODEC_THREADS + threadIdx.x);     // first item of the lane's pair
    kv[j] = i < n ? p2[i >> 1] : make_uint4(0u, 0u, 0u, 0u);           // (the second item may lie past n: masked below)
    before[j] = (lane == 0 && i > 0 && i < n) ? src.p[i - 1] : make_uint2(0xFFFFFFFFu, 0u);
  }
  u32 s = 0;
#pragma unroll
  for (int j = 0; j < NL; j++) {
    const u64 i = t0 + 2 * ((u64)j * CODEC_THREADS + threadIdx.x);
    uint2 q;
    q.x = __shfl_up(kv[j].z, 1, DCD_WAVE);
    q.y = __shfl_up(kv[j].w, 1, DCD_WAVE);
    if (lane == 0) q = before[j];
    if (i < n) s += (i == 0 || q.x != kv[j].x) ? seq_nunits(src.base + kv[j].y) : seq_nunits32(kv[j].y - q.y);
    if (i + 1 < n) s += kv[j].x != kv[j].z ? seq_nunits(src.base + kv[j].w) : seq_nunits32(kv[j].w - kv[j].y);
  }
  s = wave_sum(s);
  if (lane == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (u64)lds[0] + lds[1] + lds[2] + lds[3];
}

// which tiles of the hot region lie inside one bucket (EncSrcPairs::hot_tile): one thread per tile, two searches over the <= 256
// bucket starts (in the pack kernels themselves every workgroup would repeat them: sixteen dependent loads in front of 8 KB of work)
__global__ void k_pack_hot_tiles(EncSrcPairs src, u64 n_tiles_hot, u32* __restrict__ hot_tile) {
  const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles_hot) return;
  const u64 t0 = t * CODEC_TILE;
  u32 v = 0;
  if (t0 != 0 && t0 + CODEC_TILE <= src.n_hot32) {
    const u32 b = src.hot_bucket(t0 - 1);
    if (b == src.hot_bucket(t0 + CODEC_TILE - 1)) v = b + 1u;
  }
  hot_tile[t] = v;
}

// A workgroup writes PACK_TPB consecutive tiles.  The barriers order LDS only and the pairs of the next tile are fetched
// into registers while the current one is encoded, so that loads and stores stay in flight across tiles.
__global__ __launch_bounds__(CODEC_THREADS, DCD_PACK_WPE) void k_pack_write(EncSrcPairs src, u64 n, const u64* __restrict__ partial, u16* __restrict__ out) {
  __shared__ u32 lds[CODEC_THREADS / 64];
  // pairs are read back in blocked order (lane t reads elements 8t .. 8t+8): one element of padding per
  // 8 keeps those 64-bit LDS reads conflict-free (stride 72 B instead of 64 B)
  __shared__ union { uint2 pairs[(CODEC_TILE + 1) + (CODEC_TILE + 1) / 8 + 1]; u16 units[CODEC_TILE * 5]; } sh;
  auto pad = [](u32 k) -> u32 { return k + (k >> 3); };
  constexpr int NLD = (CODEC_TILE + 1 + CODEC_THREADS - 1) / CODEC_THREADS;       // loads per thread and tile (incl. the pair before it)
  const u64 n_tiles = (n + CODEC_TILE - 1) / CODEC_TILE;
  const u64 tile_a = (u64)blockIdx.x * PACK_TPB, tile_b = tile_a + PACK_TPB < n_tiles ? tile_a + PACK_TPB : n_tiles;
  const u32 lane = lane_id(), w = threadIdx.x >> 6;
  uint2 pf[NLD];
  auto fetch = [&](u64 tile) {
    const u64 t0 = tile * CODEC_TILE;
    const u32 cnt = (u32)min((u64)CODEC_TILE, n - t0);
    if (src.n_hot32 && t0 <= src.n_hot32) {                              // a tile of the hot region (see k_pack_count)
      const u32 ti = src.hot_tile[tile];
      if (ti) {
        const u32 hk = (ti - 1u) << src.hot_s;
        const u32* c32 = reinterpret_cast<const u32*>(src.p);
#pragma unroll
        for (int i = 0; i < NLD; i++) {
          const u32 k = threadIdx.x + i * CODEC_THREADS;
          pf[i] = k < cnt + 1 ? make_uint2(hk, c32[t0 + k - 1]) : make_uint2(0xFFFFFFFFu, 0u);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NLD; i++) {
          const u32 k = threadIdx.x + i * CODEC_THREADS;
          const u64 g = t0 + k;
          pf[i] = (k < cnt + 1 && g) ? src.at(g - 1) : make_uint2(0xFFFFFFFFu, 0u);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const u32 k = threadIdx.x + i * CODEC_THREADS;                   // pairs[k] = p[t0 + k - 1]
      const u64 g = t0 + k;
      pf[i] = (k < cnt + 1 && g) ? src.p[g - 1] : make_uint2(0xFFFFFFFFu, 0u);
    }
  };
  // round 4: a full tile inside one hot bucket (hot_tile) whose postings take ONE unit each -- the scanned counts say so: as many
  // units as postings; the head of a Zipf index: gaps of tens to hundreds of characters -- is a plain difference of neighbouring
  // 4-byte coordinates: no staging of pairs, no scan, no term heads (PACK_TPB == 1: the tile is the workgroup's only one)
  auto one_unit_tile = [&](u64 tile) -> bool {
    if (PACK_TPB != 1 || !src.n_hot32 || tile * CODEC_TILE > src.n_hot32 || !src.hot_tile[tile] || tile + 1 >= n_tiles) return false;   // (partial[] has n_tiles entries)
    return partial[tile + 1] - partial[tile] == (u64)CODEC_TILE;
  };
  if (tile_a < tile_b && one_unit_tile(tile_a)) {
    const u64 t0 = tile_a * CODEC_TILE;
    const u64 gbase = partial[tile_a];
    const u32* c32 = reinterpret_cast<const u32*>(src.p) + t0;
    const u32 l0 = threadIdx.x * CODEC_ITEMS;
    u32 c[CODEC_ITEMS];
#pragma unroll
    for (int j = 0; j < CODEC_ITEMS; j += 4) {
      const uint4 v = *reinterpret_cast<const uint4*>(c32 + l0 + j);
      c[j] = v.x; c[j + 1] = v.y; c[j + 2] = v.z; c[j + 3] = v.w;
    }
    u32 q = __shfl_up(c[CODEC_ITEMS - 1], 1, DCD_WAVE);
    if (lane == 0) q = c32[(int64_t)l0 - 1];                          // (t0 >= one tile: the posting before the tile exists)
#pragma unroll
    for (int j = 0; j < CODEC_ITEMS; j++) { sh.units[l0 + j] = (u16)(c[j] - q); q = c[j]; }
    lds_barrier();
    u16* dst = out + gbase;
    const u32 tot = CODEC_TILE;
    u32 head = (u32)(((uintptr_t)dst >> 1) & 1u);
    if (threadIdx.x == 0 && head) dst[0] = sh.units[0];
    u32 npair = (tot - head) >> 1;
    u32* dst32 = reinterpret_cast<u32*>(dst + head);
    for (u32 k = threadIdx.x; k < npair; k += CODEC_THREADS)
      dst32[k] = (u32)sh.units[head + 2 * k] | ((u32)sh.units[head + 2 * k + 1] << 16);
    if (threadIdx.x == 0 && ((tot - head) & 1u)) dst[tot - 1] = sh.units[tot - 1];
    return;
  }
  if (tile_a < tile_b) fetch(tile_a);
  for (u64 tile = tile_a; tile < tile_b; tile++) {
    const u64 t0 = tile * CODEC_TILE;
    const u32 cnt = (u32)min((u64)CODEC_TILE, n - t0);
    const u64 gbase = partial[tile];
#pragma unroll
    for (int i = 0; i < NLD; i++) {
      const u32 k = threadIdx.x + i * CODEC_THREADS;
      if (k < cnt + 1) sh.pairs[pad(k)] = pf[i];
    }
    lds_barrier();
    // Inside a term the delta of two segment-local coordinates fits 32 bits (<= 3 units); the first posting of a term
    // carries the 64-bit base and may need up to 5.  d_lo keeps the low 32 bits, heads re-derive the full value.
    u32 d_lo[CODEC_ITEMS], nu[CODEC_ITEMS], term[CODEC_ITEMS], cy[CODEC_ITEMS];
    bool hd[CODEC_ITEMS];
    u32 s = 0;
    const u32 l0 = threadIdx.x * CODEC_ITEMS;
#pragma unroll
    for (int j = 0; j < CODEC_ITEMS; j++) {
      u32 li = l0 + j;
      d_lo[j] = 0; nu[j] = 0; term[j] = 0; cy[j] = 0; hd[j] = false;
      if (li < cnt) {
        uint2 q = sh.pairs[pad(li)], c = sh.pairs[pad(li + 1)];
        hd[j] = (t0 + li == 0) || q.x != c.x;
        term[j] = c.x;
        cy[j] = c.y;
        if (hd[j]) {
          const u64 d = src.base + c.y;
          d_lo[j] = (u32)d;
          nu[j] = seq_nunits(d);
        } else {
          d_lo[j] = c.y - q.y;
          nu[j] = seq_nunits32(d_lo[j]);
        }
        s += nu[j];
      }
    }
    if (tile + 1 < tile_b) fetch(tile + 1);                            // in flight while this tile is encoded and stored
    // block scan of the unit counts (LDS-only barriers)
    const u32 incl = wave_scan_incl(s);
    if (lane == 63) lds[w] = incl;
    lds_barrier();                                                     // (also: every pair read of the tile is done)
    u32 loff = incl - s, tot = 0;
#pragma unroll
    for (int k = 0; k < CODEC_THREADS / 64; k++) { const u32 v = lds[k]; tot += v; loff += k < (int)w ? v : 0u; }
#pragma unroll
    for (int j = 0; j < CODEC_ITEMS; j++) {
      if (l0 + j < cnt) {
        if (hd[j]) { const u32 t = src.id_of(term[j]); src.term_unit_start[t] = gbase + loff; src.term_item_start[t] = t0 + l0 + j; }
        if (hd[j] && nu[j] > 2) {                                   // 64-bit head (rare): the general code
          const u64 d = src.base + cy[j];
          for (u32 g = 0; g < nu[j]; g++) sh.units[loff + g] = seq_unit(d, g, nu[j]);
        } else {                                                    // <= 3 units of a 32-bit value, no loop
          const u32 d = d_lo[j];
          sh.units[loff] = (u16)((d & 0x7FFFu) | (nu[j] > 1 ? 0x8000u : 0u));
          if (nu[j] > 1) sh.units[loff + 1] = (u16)(((d >> 15) & 0x7FFFu) | (nu[j] > 2 ? 0x8000u : 0u));
          if (nu[j] > 2) sh.units[loff + 2] = (u16)(d >> 30);
        }
        loff += nu[j];
      }
    }
    lds_barrier();
    u16* dst = out + gbase;
    u32 head = (u32)(((uintptr_t)dst >> 1) & 1u) & (tot ? 1u : 0u);
    if (threadIdx.x == 0 && head) dst[0] = sh.units[0];
    u32 npair = (tot - head) >> 1;
    u32* dst32 = reinterpret_cast<u32*>(dst + head);
    for (u32 k = threadIdx.x; k < npair; k += CODEC_THREADS)
      dst32[k] = (u32)sh.units[head + 2 * k] | ((u32)sh.units[head + 2 * k + 1] << 16);
    if (threadIdx.x == 0 && ((tot - head) & 1u)) dst[tot - 1] = sh.units[tot - 1];
    lds_barrier();                                                     // the units are out of LDS: the next tile's pairs may come in
  }
}

void decode_count_launch(const u16* units, u64 n, u64* partial, hipStream_t st) {
  u64 nb = div_up(n ? n : 1, CODEC_TILE);
  hipLaunchKernelGGL(k_dec_count, dim3((u32)nb), dim3(CODEC_THREADS), 0, st, units, n, partial);
  hipLaunchKernelGGL((k_scan_partials<u64>), dim3(1), dim3(1024), 0, st, partial, nb);
}

void decode_values_launch(const u16* units, u64 n, const u64* partial_scanned, u64 n_values, u64* deltas, u64* values,
                          u64* scan_scratch, hipStream_t st) {
  u64 nb = div_up(n ? n : 1, CODEC_TILE);
  hipLaunchKernelGGL(k_dec_deltas, dim3((u32)nb), dim3(CODEC_THREADS), 0, st, units, n, partial_scanned, deltas);
  if (n_values) {
    scan_exclusive<u64, u64>(deltas, values, n_values, scan_scratch, st);
    hipLaunchKernelGGL(k_add_inplace, dim3((u32)div_up(n_values, 256)), dim3(256), 0, st, values, deltas, n_values);
  }
}

// per-thread stream + device for the stateless seq_* entry points
struct SeqCtx {
  int device = 0;
  hipStream_t st = nullptr;
};
static int32_t seq_ctx(SeqCtx** out) {
  static thread_local SeqCtx ctx;
  DCD_TRY(ensure_device(ctx.device));
  if (!ctx.st) DCD_HIP(hipStreamCreateWithFlags(&ctx.st, hipStreamNonBlocking));
  *out = &ctx;
  return DCD_OK;
}

}  // namespace dcd

using namespace dcd;

extern "C" {

int32_t dcd_seq_encode(const uint64_t* values, size_t n, uint16_t* out, size_t cap, size_t* n_units) {
  if ((!values && n) || !n_units) DCD_FAIL(DCD_E_BADARG, "dcd_seq_encode: NULL argument");
  if (n == 0) { *n_units = 0; return DCD_OK; }
  SeqCtx* c;
  DCD_TRY(seq_ctx(&c));
  StreamScope ss_(c->st);
  DevBuf<u64> dv, part;
  DevBuf<u16> du;
  u64 nb = encode_nblocks(n);
  DCD_TRY(dv.alloc(c->device, n));
  DCD_TRY(part.alloc(c->device, nb + 1));
  DCD_TRY(du.alloc(c->device, 5 * n));
  DCD_HIP(hipMemcpyAsync(dv.p, values, n * 8, hipMemcpyHostToDevice, c->st));
  EncSrcValues src{dv.p};
  encode_launch(src, n, part.p, du.p, c->st);
  u64 total = 0;
  DCD_HIP(hipMemcpyAsync(&total, part.p + nb, 8, hipMemcpyDeviceToHost, c->st));
  DCD_HIP(hipStreamSynchronize(c->st));
  *n_units = (size_t)total;
  if (total > cap || !out) {
    if (out || cap) DCD_FAIL(DCD_E_OVERFLOW, "dcd_seq_encode: need %llu units, cap %zu", (unsigned long long)total, cap);
    return DCD_E_OVERFLOW;
  }
  DCD_HIP(hipMemcpy(out, du.p, total * 2, hipMemcpyDeviceToHost));
  return DCD_OK;
}

static int32_t seq_decode_impl(const uint16_t* units, size_t n_units, uint64_t* out, size_t cap, size_t* n, bool want) {
  if ((!units && n_units) || !n) DCD_FAIL(DCD_E_BADARG, "dcd_seq_decode: NULL argument");
  if (n_units == 0) { *n = 0; return DCD_OK; }
  SeqCtx* c;
  DCD_TRY(seq_ctx(&c));
  StreamScope ss_(c->st);
  DevBuf<u16> du;
  DevBuf<u64> part;
  u64 nb = div_up(n_units, CODEC_TILE);
  DCD_TRY(du.alloc(c->device, n_units));
  DCD_TRY(part.alloc(c->device, nb + 1));
  DCD_HIP(hipMemcpyAsync(du.p, units, n_units * 2, hipMemcpyHostToDevice, c->st));
  decode_count_launch(du.p, n_units, part.p, c->st);
  u64 nv = 0;
  DCD_HIP(hipMemcpyAsync(&nv, part.p + nb, 8, hipMemcpyDeviceToHost, c->st));
  DCD_HIP(hipStreamSynchronize(c->st));
  *n = (size_t)nv;
  if (!want) return DCD_OK;
  if (nv > cap || (!out && nv)) {
    if (out || cap) DCD_FAIL(DCD_E_OVERFLOW, "dcd_seq_decode: need %llu values, cap %zu", (unsigned long long)nv, cap);
    return DCD_E_OVERFLOW;
  }
  if (nv == 0) return DCD_OK;
  DevBuf<u64> deltas, vals, sscr;
  DCD_TRY(deltas.alloc(c->device, nv));
  DCD_TRY(vals.alloc(c->device, nv));
  DCD_TRY(sscr.alloc(c->device, scan_scratch_elems(nv)));
  decode_values_launch(du.p, n_units, part.p, nv, deltas.p, vals.p, sscr.p, c->st);
  DCD_HIP(hipMemcpyAsync(out, vals.p, nv * 8, hipMemcpyDeviceToHost, c->st));
  DCD_HIP(hipStreamSynchronize(c->st));
  return DCD_OK;
}

int32_t dcd_seq_decode(const uint16_t* units, size_t n_units, uint64_t* out, size_t cap, size_t* n) {
  return seq_decode_impl(units, n_units, out, cap, n, true);
}

int32_t dcd_seq_count(const uint16_t* units, size_t n_units, size_t* n_values) {
  return seq_decode_impl(units, n_units, nullptr, 0, n_values, false);
}

// IndexSequence.Shift (IndexSequence.cs:191-202) rewrites only the first value's <= 5 units and
// keeps the tail verbatim: O(1) arithmetic + a memcpy, so it is done in place on the host buffer
// (there is nothing data-parallel to launch).
int32_t dcd_seq_shift(const uint16_t* units, size_t n_units, uint64_t shift, uint16_t* out, size_t cap, size_t* n_out) {
  if ((!units && n_units) || !n_out) DCD_FAIL(DCD_E_BADARG, "dcd_seq_shift: NULL argument");
  if (n_units == 0 || shift == 0) {
    *n_out = n_units;
    if (n_units > cap) return DCD_E_OVERFLOW;
    if (out && n_units) memmove(out, units, n_units * 2);
    return DCD_OK;
  }
  // first value (MoveNext once)
  u64 first = 0;
  size_t q = 0;
  u32 sh = 0;
  bool more = true;
  while (more && q < n_units) {
    u16 v = units[q++];
    more = (v & 0x8000u) != 0;
    first += ((u64)(v & 0x7FFFu)) << (sh & 63u);
    sh += 15;
  }
  if (more) DCD_FAIL(DCD_E_BADARG, "dcd_seq_shift: sequence holds no complete value");
  u64 d = shift + first;
  u16 head[5];
  size_t nh = 0;
  do {
    if (d > 0x7FFF) { head[nh++] = (u16)(0x8000u | (d & 0x7FFFu)); d >>= 15; }
    else { head[nh++] = (u16)d; break; }
  } while (d > 0);
  size_t total = nh + (n_units - q);
  *n_out = total;
  if (total > cap || !out) return DCD_E_OVERFLOW;
  memmove(out + nh, units + q, (n_units - q) * 2);
  memcpy(out, head, nh * 2);
  return DCD_OK;
}

}  // extern "C"

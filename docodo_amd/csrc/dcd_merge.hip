// dcd_merge.hip -- doc-range shard merge ("reduce by key over xGMI").
//
// Reference behaviour being replaced: MergeFiles / MergeIndexes with shift_coords (Index.cs:577-752):
// equal terms' IndexSequences are concatenated in shard order; every appended sequence has its first
// value re-coded relative to the previous shard's last value (Shift + AddRange, Index.cs:665-674).
// Shards here carry GLOBAL coordinates already (coord_base = text length of the previous shards,
// which also avoids the reference's "shift by last word start" quirk), so the
// merge only re-codes 1..5 head units per (term, shard) segment and copies the rest verbatim.
//
// Export (per shard):  owner(term) = hash(key) % n_owners; terms are grouped by owner (stable, so
// ordinal order survives inside an owner) and three buffers are filled: metadata, keys, units.
// Merge (per owner): keys of all incoming segments go through the same exact hash table as the
// build's vocabulary, the union vocabulary is sorted ordinally, segments are stably sorted by term
// (source order = coordinate order is kept), head units are re-coded, output offsets come from a scan
// and one copy kernel writes the merged posting lists.  The library never communicates: the caller
// (bench.py / docodo_amd.dist) moves the buffers with RCCL all-to-all.
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>

#include "dcd_codec.h"
#include "dcd_index_int.h"
#include "dcd_scan.h"
#include "dcd_sort.h"
#include "dcd_tokenize.h"
#include "dcd_vocab.h"

struct dcd_build;
namespace dcd {
DeviceIndex* build_peek_index(dcd_build* b);
hipStream_t build_stream(dcd_build* b);
int build_device(dcd_build* b);

__device__ __forceinline__ u32 key_hash16(const u16* s, u32 n) {
  u32 h = 0x811C9DC5u;
  for (u32 i = 0; i < n; i++) h = (h ^ s[i]) * 0x01000193u;
  return fmix32(h ^ n);
}

// 64-bit fingerprint of a key (FNV-1a over the UTF-16 units + a finaliser): names a HEAVY term in the owner overrides of a
// balanced export.  Two keys with one fingerprint simply share an owner -- any function of the key is a valid assignment.
__device__ __forceinline__ u64 key_fp64(const u16* s, u32 n) {
  u64 h = 0xCBF29CE484222325ull;
  for (u32 i = 0; i < n; i++) h = (h ^ s[i]) * 0x100000001B3ull;
  h ^= n;
  h ^= h >> 33; h *= 0xFF51AFD7ED558CCDull; h ^= h >> 33; h *= 0xC4CEB9FE1A85EC53ull; h ^= h >> 33;
  return h;
}

// ---- export ------------------------------------------------------------------------------------
// owner(term) = hash(key) % n_owners, unless the key's fingerprint is in the (sorted, small) override table
// light_thr != nullptr: the light terms are spread by WEIGHT -- owner = first o with hash(key) <= light_thr[o] (ascending, last =
// 2^32 - 1): an owner that holds a heavy term gets a smaller slice of the hash space
__global__ void k_owner_pairs(const u16* __restrict__ chars, const u32* __restrict__ str_off, u64 n_terms, u32 n_owners,
                              const u64* __restrict__ over_fp, const u32* __restrict__ over_owner, u32 n_over,
                              const u32* __restrict__ light_thr, uint2* __restrict__ pairs) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const u16* k = chars + str_off[t];
  const u32 len = str_off[t + 1] - str_off[t];
  const u32 hk = key_hash16(k, len);
  u32 o = hk % n_owners;
  if (light_thr) {
    o = 0;
    while (o + 1 < n_owners && hk > light_thr[o]) o++;
  }
  if (n_over) {
    const u64 fp = key_fp64(k, len);
    u32 a = 0, b = n_over;                     // first entry >= fp
    while (a < b) { u32 m = (a + b) >> 1; if (over_fp[m] < fp) a = m + 1; else b = m; }
    if (a < n_over && over_fp[a] == fp) o = over_owner[a];
  }
  pairs[t] = make_uint2(o, (u32)t);
}

// the terms with at least min_units posting units: fingerprint + units each (order of arrival; the host sorts)
__global__ void k_heavy_terms(const u16* __restrict__ chars, const u32* __restrict__ str_off, const u64* __restrict__ unit_start, u64 n_terms,
                              u64 min_units, u32 cap, u32* __restrict__ count, u64* __restrict__ out) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const u64 u = unit_start[t + 1] - unit_start[t];
  if (u < min_units) return;
  const u32 slot = atomicAdd(count, 1u);
  if (slot < cap) { out[2 * slot] = key_fp64(chars + str_off[t], str_off[t + 1] - str_off[t]); out[2 * slot + 1] = u; }
}

struct PermUnitsF {   // units of the term at export position j
  const uint2* perm; const u64* unit_start;
  __device__ u64 operator()(u64 j) const { u32 t = perm[j].y; return unit_start[t + 1] - unit_start[t]; }
};
struct PermKeyLenF {
  const uint2* perm; const u32* str_off;
  __device__ u64 operator()(u64 j) const { u32 t = perm[j].y; return (u64)(str_off[t + 1] - str_off[t]); }
};

// per-owner totals: [owner][0] terms, [1] key units, [2] posting units
__global__ __launch_bounds__(256) void k_owner_totals(const uint2* __restrict__ perm, u64 n_terms, const u64* __restrict__ unit_start,
                                                     const u32* __restrict__ str_off, u32 n_owners, unsigned long long* __restrict__ totals) {
  __shared__ unsigned long long s[64 * 3];
  for (u32 i = threadIdx.x; i < n_owners * 3; i += blockDim.x) s[i] = 0;
  __syncthreads();
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_terms) {
    uint2 p = perm[j];
    atomicAdd(&s[p.x * 3 + 0], 1ull);
    atomicAdd(&s[p.x * 3 + 1], (unsigned long long)(str_off[p.y + 1] - str_off[p.y]));
    atomicAdd(&s[p.x * 3 + 2], (unsigned long long)(unit_start[p.y + 1] - unit_start[p.y]));
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < n_owners * 3; i += blockDim.x)
    if (s[i]) atomicAdd(&totals[i], s[i]);
}

__global__ void k_export_meta(const uint2* __restrict__ perm, u64 n_terms, const u64* __restrict__ unit_start, const u64* __restrict__ item_start,
                              const u32* __restrict__ str_off, const u16* __restrict__ chars, const u64* __restrict__ first,
                              const u64* __restrict__ last, const u64* __restrict__ key_dst, dcd_term_meta* __restrict__ meta,
                              u16* __restrict__ keys) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_terms) return;
  u32 t = perm[j].y;
  dcd_term_meta m;
  m.first = first[t];
  m.last = last[t];
  m.n_values = item_start[t + 1] - item_start[t];
  m.n_units = (u32)(unit_start[t + 1] - unit_start[t]);
  m.key_len = str_off[t + 1] - str_off[t];
  meta[j] = m;
  const u16* s = chars + str_off[t];
  u16* d = keys + key_dst[j];
  for (u32 i = 0; i < m.key_len; i++) d[i] = s[i];
}

// segmented copy: dst_off[j] .. dst_off[j+1] <- src[src_start(j) ..].  A workgroup owns a tile of 8192 output units,
// each of its four waves a span of 2048 (4 x 8 consecutive units = 4 x 16 bytes per lane, lane-interleaved so that
// every store instruction of the wave is contiguous).  The tile's first and last segment are found once per workgroup.
// A span that lies inside ONE segment (by weight almost all of a Zipf index) takes the segment's bookkeeping once per
// wave and streams; other spans fall back to one search per 8-unit chunk.  8 units inside one segment move as one
// 16-byte store fed by (possibly 2-byte-aligned) dword loads.
constexpr int CP_UPT = 8;
constexpr int CP_THREADS = 256;
constexpr int CP_CPL = 4;                               // chunks per lane
constexpr int CP_SPAN = 64 * CP_UPT * CP_CPL;           // units per wave
constexpr int CP_TILE = CP_SPAN * (CP_THREADS / 64);
typedef u32 __attribute__((aligned(2))) u32_a2;

// last j in [0, n) with off[j] <= u  (off ascending, off[0] <= u)
__device__ __forceinline__ u64 seg_of(const u64* __restrict__ off, u64 lo, u64 hi, u64 u) {
  while (lo + 1 < hi) { u64 m = (lo + hi) >> 1; if (off[m] <= u) lo = m; else hi = m; }
  return lo;
}
__device__ __forceinline__ void tile_seg_range(const u64* __restrict__ off, u64 n_seg, u64 tile_base, u64 total, u64* s_q) {
  if (threadIdx.x == 0) s_q[0] = seg_of(off, 0, n_seg, tile_base);
  if (threadIdx.x == 64) { u64 e = tile_base + CP_TILE; s_q[1] = seg_of(off, 0, n_seg, (e < total ? e : total) - 1); }
  __syncthreads();
}
__device__ __forceinline__ void copy8(const u16* __restrict__ src, u16* __restrict__ dst) {
  const u32_a2* sp = reinterpret_cast<const u32_a2*>(src);
  uint4 v = make_uint4(sp[0], sp[1], sp[2], sp[3]);
  *reinterpret_cast<uint4*>(dst) = v;
}

__global__ __launch_bounds__(CP_THREADS) void k_export_units(const uint2* __restrict__ perm, u64 n_terms, const u64* __restrict__ dst_off,
                                                            u64 total_units, const u64* __restrict__ unit_start,
                                                            const u16* __restrict__ units, u16* __restrict__ out) {
  __shared__ u64 s_q[2];
  const u64 tile_base = (u64)blockIdx.x * CP_TILE;
  tile_seg_range(dst_off, n_terms, tile_base, total_units, s_q);
  const u32 lane = threadIdx.x & 63;
  const u64 span = tile_base + (u64)(threadIdx.x >> 6) * CP_SPAN;
  if (span >= total_units) return;
  const bool aligned = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const u64 j0 = seg_of(dst_off, s_q[0], s_q[1] + 1, span);                 // wave-uniform
  const u64 next0 = j0 + 1 < n_terms ? dst_off[j0 + 1] : total_units;
  if (aligned && span + CP_SPAN <= next0) {                                 // the whole span inside one segment
    const u16* src = units + unit_start[perm[j0].y] + (span - dst_off[j0]);
#pragma unroll
    for (int c = 0; c < CP_CPL; c++) copy8(src + (c * 64 + lane) * CP_UPT, out + span + (c * 64 + lane) * CP_UPT);
    return;
  }
  for (int c = 0; c < CP_CPL; c++) {
    const u64 u0 = span + (u64)(c * 64 + lane) * CP_UPT;
    if (u0 >= total_units) break;
    u64 j = seg_of(dst_off, j0, s_q[1] + 1, u0);
    u64 next = j + 1 < n_terms ? dst_off[j + 1] : total_units;
    u64 src = unit_start[perm[j].y] + (u0 - dst_off[j]);
    if (aligned && u0 + CP_UPT <= next) { copy8(units + src, out + u0); continue; }
    for (int k = 0; k < CP_UPT; k++) {
      u64 u = u0 + k;
      if (u >= total_units) break;
      while (u >= next) { j++; next = j + 1 < n_terms ? dst_off[j + 1] : total_units; src = unit_start[perm[j].y]; }
      out[u] = units[src++];
    }
  }
}

// ---- merge ---------------------------------------------------------------------------------------
struct MetaKeyLenF { const dcd_term_meta* m; __device__ u64 operator()(u64 j) const { return m[j].key_len; } };
struct MetaUnitsF { const dcd_term_meta* m; __device__ u64 operator()(u64 j) const { return m[j].n_units; } };

__global__ void k_insert_keys(const dcd_term_meta* __restrict__ meta, const u64* __restrict__ key_off, u64 n, TextView<u16> tv, VocabTable vt,
                              uint2* __restrict__ pairs) {
  u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const u32 len = meta[j].key_len, nbytes = len * 2u, nw = (nbytes + 3u) >> 2;
  const u32* pool32 = reinterpret_cast<const u32*>(tv.pool);
  const u64 boff = key_off[j] * 2u, pbytes = tv.n_pool * 2u;
  auto word = [pool32, boff, pbytes, nbytes](u32 q) -> u32 { return str_word(pool32, boff, q, pbytes) & tail_mask(nbytes, q); };
  HashState hs;
  hs.init(len);
  for (u32 q = 0; q < nw; q++) hs.add(word(q));
  u32 slot = vocab_lookup_insert<u16>(vt, tv, hs.fin(), len, key_off[j], word);
  pairs[j] = make_uint2(slot, (u32)j);
}

// per sorted segment q: new head delta and output size
struct SegPlan {
  const uint2* sorted;          // x = term rank, y = segment index
  const dcd_term_meta* meta;
  __device__ __forceinline__ bool head(u64 q) const { return q == 0 || sorted[q].x != sorted[q - 1].x; }
  __device__ __forceinline__ u64 new_delta(u64 q) const {
    const dcd_term_meta& m = meta[sorted[q].y];
    return head(q) ? m.first : m.first - meta[sorted[q - 1].y].last;
  }
  __device__ __forceinline__ u64 out_units(u64 q) const {
    const dcd_term_meta& m = meta[sorted[q].y];
    return (u64)m.n_units - seq_nunits(m.first) + seq_nunits(new_delta(q));
  }
};
struct SegUnitsF { SegPlan p; __device__ u64 operator()(u64 q) const { return p.out_units(q); } };
struct SegValuesF { SegPlan p; __device__ u64 operator()(u64 q) const { return p.meta[p.sorted[q].y].n_values; } };

__global__ void k_merge_heads(SegPlan p, u64 n_seg, const u64* __restrict__ dst_off, const u64* __restrict__ val_off,
                              u64* __restrict__ term_unit_start, u64* __restrict__ term_item_start, u64* __restrict__ term_first,
                              u64* __restrict__ term_last) {
  u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_seg) return;
  const u32 t = p.sorted[q].x;
  if (p.head(q)) { term_unit_start[t] = dst_off[q]; term_item_start[t] = val_off[q]; term_first[t] = p.meta[p.sorted[q].y].first; }
  if (q + 1 == n_seg || p.sorted[q + 1].x != t) term_last[t] = p.meta[p.sorted[q].y].last;   // last segment of the term
}

__global__ __launch_bounds__(CP_THREADS) void k_merge_copy(SegPlan p, u64 n_seg, const u64* __restrict__ dst_off, u64 total_units,
                                                          const u64* __restrict__ src_off, const u16* __restrict__ units, u16* __restrict__ out) {
  __shared__ u64 s_q[2];
  const u64 tile_base = (u64)blockIdx.x * CP_TILE;
  tile_seg_range(dst_off, n_seg, tile_base, total_units, s_q);
  const u32 lane = threadIdx.x & 63;
  const u64 span = tile_base + (u64)(threadIdx.x >> 6) * CP_SPAN;
  if (span >= total_units) return;
  const bool aligned = (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  const u64 q0 = seg_of(dst_off, s_q[0], s_q[1] + 1, span);                 // wave-uniform
  {
    // the whole span inside one segment and behind its (re-coded) head: the segment's bookkeeping once per wave
    const u64 next0 = q0 + 1 < n_seg ? dst_off[q0 + 1] : total_units;
    if (aligned && span + CP_SPAN <= next0) {
      const u32 sg = p.sorted[q0].y;
      const u32 newlen = seq_nunits(p.new_delta(q0)), oldlen = seq_nunits(p.meta[sg].first);
      const u64 r = span - dst_off[q0];
      if (r >= newlen) {
        const u16* src = units + src_off[sg] + oldlen + (r - newlen);
#pragma unroll
        for (int c = 0; c < CP_CPL; c++) copy8(src + (c * 64 + lane) * CP_UPT, out + span + (c * 64 + lane) * CP_UPT);
        return;
      }
    }
  }
  for (int c = 0; c < CP_CPL; c++) {
    const u64 u0 = span + (u64)(c * 64 + lane) * CP_UPT;
    if (u0 >= total_units) break;
    u64 q = seg_of(dst_off, q0, s_q[1] + 1, u0);
    {
      // 8 units of one segment, all behind its (re-coded) head: plain copy
      const u64 next = q + 1 < n_seg ? dst_off[q + 1] : total_units;
      if (aligned && u0 + CP_UPT <= next) {
        const dcd_term_meta& m = p.meta[p.sorted[q].y];
        const u32 newlen = seq_nunits(p.new_delta(q)), oldlen = seq_nunits(m.first);
        const u64 r = u0 - dst_off[q];
        if (r >= newlen) { copy8(units + src_off[p.sorted[q].y] + oldlen + (r - newlen), out + u0); continue; }
      }
    }
    for (int k = 0; k < CP_UPT; k++) {
      u64 u = u0 + k;
      if (u >= total_units) break;
      while (q + 1 < n_seg && u >= dst_off[q + 1]) q++;
      const dcd_term_meta& m = p.meta[p.sorted[q].y];
      u64 nd = p.new_delta(q);
      u32 newlen = seq_nunits(nd), oldlen = seq_nunits(m.first);
      u64 r = u - dst_off[q];
      out[u] = r < newlen ? seq_unit(nd, (u32)r, newlen) : units[src_off[p.sorted[q].y] + oldlen + (r - newlen)];
    }
  }
}

}  // namespace dcd

using namespace dcd;

struct ExportPlan {
  u32 n_owners = 0;
  DevBuf<uint2> perm;
  DevBuf<u64> key_dst, unit_dst;
  u64 n_terms = 0, total_keys = 0, total_units = 0;
};
static std::map<const void*, std::unique_ptr<ExportPlan>>& export_plans() {
  static std::map<const void*, std::unique_ptr<ExportPlan>> m;
  return m;
}
static std::mutex g_export_mu;

// export of a device-resident index (a finished build, or an index handle: the owned partition after a merge)
static int32_t export_plan_impl(const void* h, DeviceIndex* ix, hipStream_t st, uint32_t n_owners, uint64_t* term_count, uint64_t* key_count,
                                uint64_t* unit_count, const uint64_t* over_fp = nullptr, const uint32_t* over_owner = nullptr, size_t n_over = 0,
                                const uint32_t* light_thr = nullptr) {
  if (n_owners == 0 || n_owners > 64) DCD_FAIL(DCD_E_BADARG, "n_owners must be in [1, 64]");
  if (n_over > 4096 || (n_over && (!over_fp || !over_owner))) DCD_FAIL(DCD_E_BADARG, "owner overrides: at most 4096 entries, both arrays given");
  // the override table, sorted by fingerprint (the kernel searches it); an owner out of range is the caller's bug
  std::vector<std::pair<u64, u32>> over(n_over);
  for (size_t i = 0; i < n_over; i++) {
    if (over_owner[i] >= n_owners) DCD_FAIL(DCD_E_BADARG, "owner override %zu names owner %u of %u", i, over_owner[i], n_owners);
    over[i] = {over_fp[i], over_owner[i]};
  }
  std::sort(over.begin(), over.end());
  if (ix->n_terms && (!ix->term_first.p || !ix->term_last.p)) DCD_FAIL(DCD_E_STATE, "this index carries no first/last coordinates per term (loaded from files)");
  const int dev = ix->device;
  DCD_TRY(ensure_device(dev));
  StreamScope ss_(st);
  const u64 V = ix->n_terms;
  std::unique_ptr<ExportPlan> ep(new ExportPlan());
  ep->n_owners = n_owners;
  ep->n_terms = V;
  DevBuf<uint2> pa, pb;
  DevBuf<u32> hist, sscr;
  DevBuf<u64> sscr64;
  DevBuf<unsigned long long> totals;
  DCD_TRY(pa.alloc(dev, V ? V : 1)); DCD_TRY(pb.alloc(dev, V ? V : 1));
  DCD_TRY(hist.alloc(dev, (u64)RS_MAX_BLOCKS << 10));
  DCD_TRY(sscr.alloc(dev, scan_scratch_elems((u64)RS_MAX_BLOCKS << 10)));
  DCD_TRY(sscr64.alloc(dev, scan_scratch_elems(V ? V : 1)));
  DCD_TRY(totals.alloc(dev, 64 * 3));
  DCD_TRY(ep->key_dst.alloc(dev, V + 1)); DCD_TRY(ep->unit_dst.alloc(dev, V + 1));
  DCD_HIP(hipMemsetAsync(totals.p, 0, 64 * 3 * 8, st));
  std::vector<unsigned long long> tot(64 * 3, 0);
  DevBuf<u64> d_ofp;
  DevBuf<u32> d_oown, d_thr;
  if (light_thr) {
    for (u32 o = 0; o + 1 < n_owners; o++) if (light_thr[o] > light_thr[o + 1]) DCD_FAIL(DCD_E_BADARG, "light_thr must ascend");
    if (light_thr[n_owners - 1] != 0xFFFFFFFFu) DCD_FAIL(DCD_E_BADARG, "light_thr must end with 2^32 - 1");
    if (V) {
      DCD_TRY(d_thr.alloc(dev, n_owners));
      DCD_HIP(hipMemcpyAsync(d_thr.p, light_thr, n_owners * 4, hipMemcpyHostToDevice, st));
      DCD_HIP(hipStreamSynchronize(st));
    }
  }
  if (V && n_over) {
    std::vector<u64> f(n_over); std::vector<u32> o(n_over);
    for (size_t i = 0; i < n_over; i++) { f[i] = over[i].first; o[i] = over[i].second; }
    DCD_TRY(d_ofp.alloc(dev, n_over)); DCD_TRY(d_oown.alloc(dev, n_over));
    DCD_HIP(hipMemcpyAsync(d_ofp.p, f.data(), n_over * 8, hipMemcpyHostToDevice, st));
    DCD_HIP(hipMemcpyAsync(d_oown.p, o.data(), n_over * 4, hipMemcpyHostToDevice, st));
    DCD_HIP(hipStreamSynchronize(st));                  // (the staging vectors die with this scope)
  }
  if (V) {
    hipLaunchKernelGGL(k_owner_pairs, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ix->term_chars.p, ix->term_str_off.p, V, n_owners,
                       (const u64*)d_ofp.p, (const u32*)d_oown.p, (u32)n_over, (const u32*)d_thr.p, pa.p);
    uint2* sorted = rs_sort(pa.p, pb.p, nullptr, V, 6, hist.p, sscr.p, st);
    if (sorted == pa.p) ep->perm = std::move(pa); else ep->perm = std::move(pb);
    hipLaunchKernelGGL(k_owner_totals, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ep->perm.p, V, ix->term_unit_start.p, ix->term_str_off.p,
                       n_owners, totals.p);
    // (two scratch areas: both totals come back with ONE host round trip)
    DevBuf<u64> sscr64b;
    DCD_TRY(sscr64b.alloc(dev, scan_scratch_elems(V)));
    scan_exclusive_f<PermKeyLenF, u64>(PermKeyLenF{ep->perm.p, ix->term_str_off.p}, ep->key_dst.p, V, sscr64.p, st);
    scan_exclusive_f<PermUnitsF, u64>(PermUnitsF{ep->perm.p, ix->term_unit_start.p}, ep->unit_dst.p, V, sscr64b.p, st);
    DCD_HIP(hipMemcpyAsync(&ep->total_keys, sscr64.p + scan_nblocks(V), 8, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipMemcpyAsync(&ep->total_units, sscr64b.p + scan_nblocks(V), 8, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipMemcpyAsync(tot.data(), totals.p, 64 * 3 * 8, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipStreamSynchronize(st));
    DCD_HIP(hipGetLastError());
  }
  for (u32 o = 0; o < n_owners; o++) { term_count[o] = tot[o * 3]; key_count[o] = tot[o * 3 + 1]; unit_count[o] = tot[o * 3 + 2]; }
  std::lock_guard<std::mutex> g(g_export_mu);
  export_plans()[h] = std::move(ep);
  return DCD_OK;
}

static int32_t export_fill_impl(const void* h, DeviceIndex* ix, hipStream_t st, void* meta_dev, uint16_t* keys_dev, uint16_t* units_dev) {
  std::unique_ptr<ExportPlan> ep;
  {
    std::lock_guard<std::mutex> g(g_export_mu);
    auto it = export_plans().find(h);
    if (it == export_plans().end()) DCD_FAIL(DCD_E_STATE, "dcd_build_export_plan must run first");
    ep = std::move(it->second);
    export_plans().erase(it);
  }
  const u64 V = ep->n_terms;
  if (V && (!meta_dev || (!keys_dev && ep->total_keys) || (!units_dev && ep->total_units))) DCD_FAIL(DCD_E_BADARG, "export fill: NULL buffer");
  DCD_TRY(ensure_device(ix->device));
  StreamScope ss_(st);
  if (V) {
    hipLaunchKernelGGL(k_export_meta, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ep->perm.p, V, ix->term_unit_start.p, ix->term_item_start.p,
                       ix->term_str_off.p, ix->term_chars.p, ix->term_first.p, ix->term_last.p, ep->key_dst.p, (dcd_term_meta*)meta_dev, keys_dev);
    if (ep->total_units)
      // (a DCD_BUILD_SHARD_ONLY build keeps its units in sort-key order: term r's run starts at unit_src_start[r])
      hipLaunchKernelGGL(k_export_units, dim3((u32)div_up(ep->total_units, CP_TILE)), dim3(CP_THREADS), 0, st, ep->perm.p, V, ep->unit_dst.p,
                         ep->total_units, ix->units_by_key.p ? ix->unit_src_start.p : ix->term_unit_start.p,
                         ix->units_by_key.p ? ix->units_by_key.p : ix->units.p, units_dev);
  }
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  return DCD_OK;
}

static int32_t export_heavy_impl(DeviceIndex* ix, hipStream_t st, uint64_t min_units, uint64_t* fp, uint64_t* n_units, size_t cap, size_t* n_out) {
  if (cap > 65536) cap = 65536;
  const int dev = ix->device;
  DCD_TRY(ensure_device(dev));
  StreamScope ss_(st);
  const u64 V = ix->n_terms;
  *n_out = 0;
  if (!V) return DCD_OK;
  DevBuf<u32> cnt;
  DevBuf<u64> out;
  DCD_TRY(cnt.alloc(dev, 1)); DCD_TRY(out.alloc(dev, 2 * (cap ? cap : 1)));
  DCD_HIP(hipMemsetAsync(cnt.p, 0, 4, st));
  hipLaunchKernelGGL(k_heavy_terms, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ix->term_chars.p, ix->term_str_off.p, ix->term_unit_start.p, V,
                     min_units, (u32)cap, cnt.p, out.p);
  u32 n = 0;
  DCD_HIP(hipMemcpyAsync(&n, cnt.p, 4, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipStreamSynchronize(st));
  std::vector<u64> host(2 * (size_t)std::min<u64>(n, cap));
  if (!host.empty()) DCD_HIP(hipMemcpy(host.data(), out.p, host.size() * 8, hipMemcpyDeviceToHost));
  std::vector<std::pair<u64, u64>> v(host.size() / 2);
  for (size_t i = 0; i < v.size(); i++) v[i] = {host[2 * i], host[2 * i + 1]};
  std::sort(v.begin(), v.end());                         // (deterministic order: by fingerprint)
  for (size_t i = 0; i < v.size(); i++) { fp[i] = v[i].first; n_units[i] = v[i].second; }
  *n_out = n;
  if (n > cap) DCD_FAIL(DCD_E_OVERFLOW, "dcd_build_export_heavy: %u terms reach %llu units, cap is %zu", n, (unsigned long long)min_units, cap);
  return DCD_OK;
}

extern "C" {

int32_t dcd_build_export_plan(dcd_build* h, uint32_t n_owners, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count) {
  if (!h || !term_count || !key_count || !unit_count) DCD_FAIL(DCD_E_BADARG, "dcd_build_export_plan: NULL argument");
  DeviceIndex* ix = build_peek_index(h);
  if (!ix) DCD_FAIL(DCD_E_STATE, "build is not finished (or its index was adopted)");
  return export_plan_impl(h, ix, build_stream(h), n_owners, term_count, key_count, unit_count);
}
int32_t dcd_build_export_plan_balanced(dcd_build* h, uint32_t n_owners, const uint64_t* over_fp, const uint32_t* over_owner, size_t n_over,
                                       const uint32_t* light_thr, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count) {
  if (!h || !term_count || !key_count || !unit_count) DCD_FAIL(DCD_E_BADARG, "dcd_build_export_plan_balanced: NULL argument");
  DeviceIndex* ix = build_peek_index(h);
  if (!ix) DCD_FAIL(DCD_E_STATE, "build is not finished (or its index was adopted)");
  return export_plan_impl(h, ix, build_stream(h), n_owners, term_count, key_count, unit_count, over_fp, over_owner, n_over, light_thr);
}
int32_t dcd_build_export_heavy(dcd_build* h, uint64_t min_units, uint64_t* fp, uint64_t* n_units, size_t cap, size_t* n_out) {
  if (!h || !n_out || (cap && (!fp || !n_units))) DCD_FAIL(DCD_E_BADARG, "dcd_build_export_heavy: NULL argument");
  DeviceIndex* ix = build_peek_index(h);
  if (!ix) DCD_FAIL(DCD_E_STATE, "build is not finished (or its index was adopted)");
  return export_heavy_impl(ix, build_stream(h), min_units, fp, n_units, cap, n_out);
}
int32_t dcd_index_export_heavy(dcd_index* h, uint64_t min_units, uint64_t* fp, uint64_t* n_units, size_t cap, size_t* n_out) {
  if (!h || !n_out || (cap && (!fp || !n_units))) DCD_FAIL(DCD_E_BADARG, "dcd_index_export_heavy: NULL argument");
  if (h && h->lazy) DCD_FAIL(DCD_E_STATE, "dcd_index_export_heavy: the postings of an on-disk index (dcd_index_open_lazy) are not in memory");
  return export_heavy_impl(&h->ix, h->st, min_units, fp, n_units, cap, n_out);
}
int32_t dcd_index_export_plan_balanced(dcd_index* h, uint32_t n_owners, const uint64_t* over_fp, const uint32_t* over_owner, size_t n_over,
                                       const uint32_t* light_thr, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count) {
  if (!h || !term_count || !key_count || !unit_count) DCD_FAIL(DCD_E_BADARG, "dcd_index_export_plan_balanced: NULL argument");
  if (h && h->lazy) DCD_FAIL(DCD_E_STATE, "dcd_index_export_plan_balanced: the postings of an on-disk index (dcd_index_open_lazy) are not in memory");
  return export_plan_impl(h, &h->ix, h->st, n_owners, term_count, key_count, unit_count, over_fp, over_owner, n_over, light_thr);
}
int32_t dcd_build_export_fill(dcd_build* h, void* meta_dev, uint16_t* keys_dev, uint16_t* units_dev) {
  if (!h) DCD_FAIL(DCD_E_BADARG, "dcd_build_export_fill: NULL handle");
  DeviceIndex* ix = build_peek_index(h);
  if (!ix) DCD_FAIL(DCD_E_STATE, "build is not finished (or its index was adopted)");
  return export_fill_impl(h, ix, build_stream(h), meta_dev, keys_dev, units_dev);
}
int32_t dcd_index_export_plan(dcd_index* h, uint32_t n_owners, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count) {
  if (!h || !term_count || !key_count || !unit_count) DCD_FAIL(DCD_E_BADARG, "dcd_index_export_plan: NULL argument");
  if (h && h->lazy) DCD_FAIL(DCD_E_STATE, "dcd_index_export_plan: the postings of an on-disk index (dcd_index_open_lazy) are not in memory");
  return export_plan_impl(h, &h->ix, h->st, n_owners, term_count, key_count, unit_count);
}
int32_t dcd_index_export_fill(dcd_index* h, void* meta_dev, uint16_t* keys_dev, uint16_t* units_dev) {
  if (!h) DCD_FAIL(DCD_E_BADARG, "dcd_index_export_fill: NULL handle");
  return export_fill_impl(h, &h->ix, h->st, meta_dev, keys_dev, units_dev);
}

// A merge in two halves: everything that needs only the segment table and the keys (vocabulary, ordinal ranks, output
// layout, term strings, host mirrors), then the one kernel that touches the posting units -- so that a caller can run
// the first half while the units are still in flight on the fabric.
struct dcd_merge {
  std::unique_ptr<dcd_index> h;
  DevBuf<uint2> pa, pb;
  DevBuf<u64> dst_off, src_off;
  SegPlan sp{nullptr, nullptr};
  u64 M = 0, U = 0, Uout = 0;
};

int32_t dcd_merge_begin(int32_t device, uint32_t n_sources, const uint64_t* term_count, const uint64_t* key_count,
                        const uint64_t* unit_count, const void* meta_dev, const uint16_t* keys_dev, uint64_t max_coord,
                        dcd_merge** out) {
  if (!term_count || !key_count || !unit_count || !out) DCD_FAIL(DCD_E_BADARG, "dcd_merge_begin: NULL argument");
  DCD_TRY(ensure_device(device));
  u64 M = 0, K = 0, U = 0;
  for (u32 s = 0; s < n_sources; s++) { M += term_count[s]; K += key_count[s]; U += unit_count[s]; }
  if (M && (!meta_dev || (!keys_dev && K))) DCD_FAIL(DCD_E_BADARG, "dcd_merge_begin: NULL buffer");
  if (((uintptr_t)keys_dev & 3) != 0) DCD_FAIL(DCD_E_BADARG, "keys buffer must be 4-byte aligned");
  if (M >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "too many segments");
  std::unique_ptr<dcd_merge> mg(new dcd_merge());
  mg->h.reset(new dcd_index());
  std::unique_ptr<dcd_index>& h = mg->h;
  DevBuf<uint2>&pa = mg->pa, &pb = mg->pb;
  DevBuf<u64>&dst_off = mg->dst_off, &src_off = mg->src_off;
  DCD_TRY(stream_acquire(device, &h->st));
  hipStream_t st = h->st;
  StreamScope ss_(st);
  DeviceIndex& ix = h->ix;
  ix.device = device;
  ix.max_coord = max_coord;
  const dcd_term_meta* meta = (const dcd_term_meta*)meta_dev;

  DevBuf<u64> key_off, sscr64;
  DCD_TRY(key_off.alloc(device, M + 1)); DCD_TRY(src_off.alloc(device, M + 1));
  DCD_TRY(sscr64.alloc(device, scan_scratch_elems(M ? M : 1)));
  u64 V = 0;
  u32 maxlen = 0;
  VocabRanks vr;
  DevBuf<u64> slots;
  DevBuf<u32> flags, slot_excl, hist, sscr32;
  DCD_TRY(flags.alloc(device, 4));
  DCD_TRY(hist.alloc(device, (u64)RS_MAX_BLOCKS << 10));
  DCD_TRY(pa.alloc(device, M ? M : 1)); DCD_TRY(pb.alloc(device, M ? M : 1));
  TextView<u16> tv;
  tv.text = nullptr; tv.n_chars = 0; tv.pool = keys_dev; tv.n_pool = K; tv.tab = CharTables16{nullptr, nullptr};
  uint2* sorted = pa.p;
  u64 cap = 0;
  if (M) {
    scan_exclusive_f<MetaKeyLenF, u64>(MetaKeyLenF{meta}, key_off.p, M, sscr64.p, st);
    scan_exclusive_f<MetaUnitsF, u64>(MetaUnitsF{meta}, src_off.p, M, sscr64.p, st);
    u32 cap_log2 = log2_ceil(M * 2 + 1);
    if (cap_log2 < 12) cap_log2 = 12;
    cap = 1ull << cap_log2;
    DCD_TRY(slots.alloc(device, cap));
    DCD_TRY(slot_excl.alloc(device, cap));
    DCD_TRY(sscr32.alloc(device, std::max<u64>(scan_scratch_elems(cap), scan_scratch_elems((u64)RS_MAX_BLOCKS << 10))));
    DCD_HIP(hipMemsetAsync(slots.p, 0, cap * 8, st));
    DCD_HIP(hipMemsetAsync(flags.p, 0, 16, st));
    VocabTable vt;
    vt.slots = slots.p; vt.mask = (u32)(cap - 1); vt.overflow = flags.p; vt.pool = nullptr;
    hipLaunchKernelGGL(k_insert_keys, dim3((u32)div_up(M, 256)), dim3(256), 0, st, meta, key_off.p, M, tv, vt, pa.p);
    scan_exclusive_f<SlotUsed, u32>(SlotUsed{slots.p}, slot_excl.p, cap, sscr32.p, st);
    u32 v32 = 0, fl[4] = {0, 0, 0, 0};
    DCD_HIP(hipMemcpyAsync(&v32, sscr32.p + scan_nblocks(cap), 4, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipMemcpyAsync(fl, flags.p, 16, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipStreamSynchronize(st));
    if (fl[0]) DCD_FAIL(DCD_E_OVERFLOW, "merge vocabulary table overflow");
    V = v32;
    DCD_TRY(vocab_rank<u16>(device, st, slots.p, cap, slot_excl.p, V, tv, hist.p, sscr32.p, flags.p + 1, &vr, &maxlen));
    sorted = rs_sort(pa.p, pb.p, vr.rank_of_slot.p, M, (int)log2_ceil(V ? V : 1), hist.p, sscr32.p, st);
  }
  // output offsets
  DevBuf<u64> val_off;
  DCD_TRY(dst_off.alloc(device, M + 1)); DCD_TRY(val_off.alloc(device, M + 1));
  DCD_TRY(ix.term_unit_start.alloc(device, V + 1));
  DCD_TRY(ix.term_item_start.alloc(device, V + 1));
  DCD_TRY(ix.term_first.alloc(device, V ? V : 1));
  DCD_TRY(ix.term_last.alloc(device, V ? V : 1));
  u64 Uout = 0, Nout = 0;
  SegPlan sp{sorted, meta};
  // everything below is queued without a host round trip in between: the two totals come back together at the end, the key
  // pool is sized for V keys of the longest length, the host mirrors are fetched when the index is first asked for them
  DevBuf<u64> sscr64b;
  DevBuf<u32> s32;
  DCD_TRY(sscr64b.alloc(device, scan_scratch_elems(M ? M : 1)));
  DCD_TRY(s32.alloc(device, scan_scratch_elems(V ? V : 1)));
  DCD_TRY(ix.term_str_off.alloc(device, V + 1));
  DCD_TRY(ix.term_chars.alloc(device, V * (u64)maxlen + 1));
  if (M) {
    scan_exclusive_f<SegUnitsF, u64>(SegUnitsF{sp}, dst_off.p, M, sscr64.p, st);
    scan_exclusive_f<SegValuesF, u64>(SegValuesF{sp}, val_off.p, M, sscr64b.p, st);
    hipLaunchKernelGGL(k_merge_heads, dim3((u32)div_up(M, 256)), dim3(256), 0, st, sp, M, dst_off.p, val_off.p, ix.term_unit_start.p,
                       ix.term_item_start.p, ix.term_first.p, ix.term_last.p);
    DCD_HIP(hipMemcpyAsync(ix.term_unit_start.p + V, sscr64.p + scan_nblocks(M), 8, hipMemcpyDeviceToDevice, st));
    DCD_HIP(hipMemcpyAsync(ix.term_item_start.p + V, sscr64b.p + scan_nblocks(M), 8, hipMemcpyDeviceToDevice, st));
    DCD_HIP(hipMemcpyAsync(&Uout, sscr64.p + scan_nblocks(M), 8, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipMemcpyAsync(&Nout, sscr64b.p + scan_nblocks(M), 8, hipMemcpyDeviceToHost, st));
  } else {
    DCD_HIP(hipMemsetAsync(ix.term_unit_start.p + V, 0, 8, st));
    DCD_HIP(hipMemsetAsync(ix.term_item_start.p + V, 0, 8, st));
  }
  // keys in rank order
  if (V) {
    scan_exclusive_f<TermLenF, u32>(TermLenF{vr.rk_word.p}, ix.term_str_off.p, V, s32.p, st);
    DCD_HIP(hipMemcpyAsync(ix.term_str_off.p + V, s32.p + scan_nblocks(V), 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL((k_term_strings<u16>), dim3((u32)div_up(V, 256)), dim3(256), 0, st, vr.rk_word.p, V, ix.term_str_off.p, tv, ix.term_chars.p);
  } else {
    DCD_HIP(hipMemsetAsync(ix.term_str_off.p, 0, 4, st));
  }
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  DCD_TRY(ix.units.alloc(device, Uout ? Uout : 1));
  ix.n_terms = V; ix.n_units = Uout; ix.n_postings = Nout;
  ix.host_ready = false;
  mg->sp = sp; mg->M = M; mg->U = U; mg->Uout = Uout;
  *out = mg.release();
  return DCD_OK;
}

int32_t dcd_merge_finish(dcd_merge* m, const uint16_t* units_dev, dcd_index** out) {
  if (!m || !out) DCD_FAIL(DCD_E_BADARG, "dcd_merge_finish: NULL argument");
  std::unique_ptr<dcd_merge> mg(m);   // consumed whatever happens
  if (!units_dev && mg->U) DCD_FAIL(DCD_E_BADARG, "dcd_merge_finish: NULL units buffer");
  dcd_index* h = mg->h.get();
  DCD_HIP(hipSetDevice(h->ix.device));
  StreamScope ss_(h->st);
  if (mg->Uout) {
    hipLaunchKernelGGL(k_merge_copy, dim3((u32)div_up(mg->Uout, CP_TILE)), dim3(CP_THREADS), 0, h->st, mg->sp, mg->M, mg->dst_off.p, mg->Uout,
                       mg->src_off.p, units_dev, h->ix.units.p);
    DCD_HIP(hipStreamSynchronize(h->st));
    DCD_HIP(hipGetLastError());
  }
  *out = mg->h.release();
  return DCD_OK;
}

void dcd_merge_free(dcd_merge* m) {
  if (!m) return;
  hipStream_t st = m->h ? m->h->st : nullptr;
  const int dev = m->h ? m->h->ix.device : 0;
  {
    StreamScope ss_(st);   // the buffers go back to the pool ordered behind whatever is still queued on the stream
    if (m->h) m->h->st = nullptr;
    delete m;
  }
  if (st) stream_release(dev, st);
}

int32_t dcd_merge_shards(int32_t device, uint32_t n_sources, const uint64_t* term_count, const uint64_t* key_count,
                         const uint64_t* unit_count, const void* meta_dev, const uint16_t* keys_dev, const uint16_t* units_dev,
                         uint64_t max_coord, dcd_index** out) {
  if (!out) DCD_FAIL(DCD_E_BADARG, "dcd_merge_shards: NULL argument");
  dcd_merge* m = nullptr;
  DCD_TRY(dcd_merge_begin(device, n_sources, term_count, key_count, unit_count, meta_dev, keys_dev, max_coord, &m));
  return dcd_merge_finish(m, units_dev, out);
}

int32_t dcd_index_set_uniform_pages(dcd_index* h, uint64_t n_pages, uint32_t page_chars, uint64_t first_doc, const uint16_t* source,
                                    size_t source_len) {
  if (!h || (!source && source_len)) DCD_FAIL(DCD_E_BADARG, "dcd_index_set_uniform_pages: NULL argument");
  if (page_chars == 0) DCD_FAIL(DCD_E_BADARG, "page_chars must be positive");
  PageTable& pt = h->ix.pages;
  pt = PageTable();
  pt.uniform = true; pt.uni_n = n_pages; pt.uni_chars = page_chars; pt.uni_base = 0;
  pt.doc_per_page = true; pt.first_doc = first_doc; pt.uniform_id = u"1";
  pt.source = source_len ? ustr((const char16_t*)source, source_len) : ustr(u"syn");
  h->ix.page_end.reset();   // rebuilt on the next query
  return DCD_OK;
}

}  // extern "C"

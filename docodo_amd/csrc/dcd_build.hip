// dcd_build.hip -- the build path on the device: tokenise -> exact vocabulary -> stable radix
// sort by term -> 15-bit var-ushort pack -> ".index" byte image, everything resident in HBM.
//
// Replaces IndexTask's page loop (Build.cs:477-541), Index.Builder.Add/Save (Build.cs:324-404)
// and, for one coordinate space, the merge functions (Index.cs:537-752).
#include <algorithm>
#include <chrono>
#include <memory>

#include <functional>
#include <thread>

#include "dcd_build_int.h"
#include "dcd_codec.h"
#include "dcd_scan.h"
#include "dcd_sort.h"
#include "dcd_tokenize.h"
#include "dcd_vocab.h"

namespace dcd {

constexpr u32 MAX_CODE_LEN = 63;   // explicit index codes (Builder.Add); text tokens are <= 32
#ifndef DCD_CAPDIV
#define DCD_CAPDIV 256             // first guess of the vocabulary table: one slot per DCD_CAPDIV postings
#endif
constexpr u32 UNRES_CAP = 4096;    // tiles on the fix-up list of an id-numbering build (beyond that the fix-up pass walks all tiles)
#ifndef DCD_TK_WPE
#define DCD_TK_WPE 8               // waves per SIMD the emit kernel is compiled for (register budget)
#endif
#ifndef DCD_HOT_SAMPLE_TILES
#define DCD_HOT_SAMPLE_TILES 1024
#endif

// ================================================================================================
// kernels
// ================================================================================================
// One wave per 944-character tile; the four waves of a workgroup are independent (no s_barrier).
template <typename CharT>
__global__ __launch_bounds__(TK_THREADS) void k_tok_count(const CharT* __restrict__ text, TokParams P, CharTables16 tab,
                                                         u32* __restrict__ tile_count, u64 n_tiles) {
  __shared__ TokLds<CharT, false> SW[TK_WAVES];
  const u64 tile = (u64)blockIdx.x * TK_WAVES + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  u32 keep;
  u64 bnd;
  tokenize_chunk<CharT, false>(text, P, tab, tile, SW[threadIdx.x >> 6], &keep, &bnd);
  u32 c = wave_sum((u32)__builtin_popcount(keep));
  if (lane_id() == 0) tile_count[tile] = c;
}

// FAST: the configuration of a plain text build with early ids, known at compile time -- strided output, a numbering table,
// no coordinate runs, no per-slot sample counts: the kernel has ~35 scalar arguments and an 80-SGPR budget (8 waves / SIMD),
// and every argument that is dead here is one the compiler does not have to spill to VGPR lanes around the token loop.
template <typename CharT, bool SAMPLE = false, bool FAST = false>
__global__ __launch_bounds__(TK_THREADS, DCD_TK_WPE) void k_tok_emit(const CharT* __restrict__ text, TokParams P, CharTables16 tab,
                                                        TextView<CharT> tv, VocabTable vt, const u64* __restrict__ tile_off,
                                                        uint2* __restrict__ tuples, u64 n_tiles, u32 stride, u32* __restrict__ tile_cnt,
                                                        u64 tile0, u32* __restrict__ sample_cnt, u32 sample_mask, u32* __restrict__ unres,
                                                        u32 sample_step) {
  // A workgroup owns TK_WAVES consecutive wave tiles -- one GROUP.  Every wave tokenises its own tile (no barrier in there);
  // then the group's tokens, in text order, are dealt out to the lanes of the whole workgroup: token k goes to thread
  // k mod 256.  (Before, every wave looked up its own tile's tokens, 64 at a time: a tile of this corpus holds 129 of them
  // on average -- three rounds, the third nearly empty.  A group holds 516: nine wave rounds instead of twelve.)
  // stride != 0: no counting pass ran; the group's tuples go to tuples[group * stride ...] and their number to
  // tile_cnt[group] (the first sort pass compacts, RsStrided in dcd_sort.h).  stride == 0: dense output at
  // tile_off[first tile of the group].  The launch covers the tiles [tile0, n_tiles), tile0 a multiple of TK_WAVES (see the
  // seed launch in build_phase1).
  __shared__ TokLds<CharT, true> SW[TK_WAVES];
  __shared__ u32 s_wtot[TK_WAVES];
  // FAST (round 3): the digit histogram of the FIRST RADIX PASS as a by-product.  With a hot set the first pass scatters by
  // min(id, n_hot) -- one bucket per hot term, one for all cold ones -- and its histogram kernel read all 1.1 GB of tuples
  // again only to count them.  The group's counts are collected here in LDS (1 KB: what the page-start mask used to take) and
  // leave as ONE ROW of 256 u16 counts per group (a coalesced 512-byte store; `sample_cnt` is the row array in FAST builds --
  // the kernel's scalar-register budget has no room for another argument); k_hist_rows adds the rows of every radix block up
  // behind the launch.  (First attempt: one global atomic per non-empty bin straight into the pass's histogram -- 35 M
  // device-scope atomics per build, executed at the memory side: the emit kernel went from 1.10 to 2.54 ms.)  Keys that are
  // still unresolved are counted by the epilogue.
  __shared__ u32 s_h[FAST ? 256 : 1];
  const bool do_hist = FAST && !SAMPLE && sample_cnt != nullptr;
  if (do_hist) s_h[threadIdx.x] = 0;
  const u32 w = threadIdx.x >> 6, lane = lane_id();
  // SAMPLE: a SAMPLING launch (builds with early ids, in front of the real one): workgroup b takes group b *
  // sample_step, looks its tokens up / inserts them WITHOUT numbering them and counts them in their records (word 3); it
  // writes no tuples.  k_hot_select then makes the most frequent terms the hot set.
  const u64 group = SAMPLE ? (u64)blockIdx.x * sample_step : tile0 / TK_WAVES + blockIdx.x;
  const u64 tile = group * TK_WAVES + w;
  TokLds<CharT, true>& S = SW[w];
  u32 total = 0;
  if (tile < n_tiles) {
    Chunk<CharT> cur;
    cur.fetch(text, tile_lane_g0(tile), P.n_chars, tile_inside(tile, P.n_chars));
    u32 keep;
    u64 bnd;
    tokenize_chunk<CharT, true>(text, P, tab, tile, S, &keep, &bnd, &cur);
    const u32 mine = (u32)__builtin_popcount(keep);
    const u32 incl = wave_scan_incl(mine);
    total = __shfl(incl, 63, 64);
    u32 ex = incl - mine;
    for (u32 m = keep; m;) {
      u32 j = (u32)__builtin_ctz(m);
      m &= m - 1;
      u32 len = (u32)__builtin_ctzll(bnd >> (j + 1)) + 1u;
      S.tok[ex++] = (lane * TK_CPT + j) | (len << 16);   // staged position
    }
  }
  if (lane == 0) s_wtot[w] = total;
  __syncthreads();
  // prefix of the waves' token counts: token k of the group belongs to the wave ws with pre[ws] <= k < pre[ws + 1]
  u32 pre[TK_WAVES + 1];
  pre[0] = 0;
#pragma unroll
  for (int k = 0; k < TK_WAVES; k++) pre[k + 1] = pre[k] + s_wtot[k];
  const u32 gtotal = pre[TK_WAVES];
  // every (sample_mask + 1)-th group counts its tokens per vocabulary slot: the frequency sample behind the hot / cold
  // split of the tuple sort (build_phase2)
  const bool sampled = !FAST && sample_cnt != nullptr && ((u32)group & sample_mask) == 0u;
  const u32 n_hot_h = do_hist ? *vt.id_base : 0u;
  const u64 obase = SAMPLE ? 0 : (FAST || stride) ? group * stride : tile_off[group * TK_WAVES];
  if (!SAMPLE && (FAST || stride) && threadIdx.x == 0) tile_cnt[group] = gtotal;
  constexpr u32 CS = sizeof(CharT);
  constexpr u32 PW = VocPool<CharT>::W;
  auto coord_of = [&](u64 gpos) -> u32 {
    u32 coord = (u32)gpos;
    if (!FAST && P.n_runs) {
      u32 a = 0, b = P.n_runs;   // last run with run_off <= gpos
      while (a + 1 < b) { u32 m = (a + b) >> 1; if (P.run_off[m] <= (u32)gpos) a = m; else b = m; }
      coord += P.run_delta[a];
    }
    return coord;
  };
  bool any_unres = false;
#if defined(DCD_DIAG_TOK) && DCD_DIAG_TOK == 7    // (timing only: the front half -- classification, boundaries, token list)
  if (gtotal != 0xFFFFFFFFu) return;
#endif
  for (u32 k = threadIdx.x; k < gtotal; k += TK_THREADS) {
    u32 ws = 0;
#pragma unroll
    for (int q = 1; q < TK_WAVES; q++) ws += k >= pre[q] ? 1u : 0u;
    const u32 t = SW[ws].tok[k - pre[ws]];
    const u32* txt = SW[ws].txt;
    // (a segment holds < 2^32 characters: 32-bit positions; the origin of tile 0 wraps, the positions of its tokens do not)
    const u32 org32 = (u32)(group * TK_WAVES) * (u32)TK_TILE - (u32)(TK_PRE * TK_CPT) + __umul24(ws, (u32)TK_TILE);
    const u32 i = t & 0xFFFFu, len = t >> 16;
    const u32 nbytes = len * CS, nw = (nbytes + 3u) >> 2;
    auto word = [txt, i, nbytes](u32 q) -> u32 { return lds_word(txt, i * CS, q) & tail_mask(nbytes, q); };
    HashState hs;
    hs.init(len);
    u32 wv[PW];
#pragma unroll
    for (u32 q = 0; q < PW; q++) {
      wv[q] = q < nw ? word(q) : 0u;
      if (q < nw) hs.add(wv[q]);
    }
    for (u32 q = PW; q < nw; q++) hs.add(word(q));
    const u64 gpos = (u64)(org32 + i);
#if defined(DCD_DIAG_TOK)   // timing-only builds (scripts/diag_tok.py): the outputs are wrong on purpose
    u32 slot;
    {
      const TokHash hh = hs.fin();
      if (DCD_DIAG_TOK == 1) {          // slot word only: a tag + length match counts as a hit (no string compare)
        slot = hh.idx & vt.mask;
        const u64 sw = vt.slots[(u64)slot * vt.stride];
        if ((sw >> 42) != (voc_word(hh.tag, len, gpos) >> 42)) slot ^= 1u;
      } else if (DCD_DIAG_TOK == 2) {   // no table access at all
        slot = hh.idx & vt.mask;
      } else if (DCD_DIAG_TOK == 3) {   // every fourth token looks itself up (an ideal 75 % LDS cache, its own cost excluded)
        slot = (hh.tag & 3u) ? (hh.idx & vt.mask) : vocab_lookup_insert_pooled<CharT>(vt, tv, hh, len, gpos, wv, word);
      } else if (DCD_DIAG_TOK >= 6) {   // production lookup (6: the launch is repeated on the populated table)
        slot = vocab_lookup_insert_pooled<CharT>(vt, tv, hh, len, gpos, wv, word);
      } else {                          // 4: pool entry and slot word in ONE gather (pool only), first probe only
        slot = hh.idx & vt.mask;
        const uint4 pq = *reinterpret_cast<const uint4*>(vt.pool + (u64)slot * vt.pstride);
        if (((pq.x ^ wv[0]) | (pq.y ^ wv[1]) | (pq.z ^ wv[2]) | (pq.w ^ wv[3])) != 0u) slot ^= 1u;
      }
    }
#else
    if (SAMPLE) {
      // one atomic per distinct term and wave round, not per token: the top term of a Zipf corpus is every seventh token, and
      // same-address atomics execute one after the other (measured: the launch 0.10 -> 0.05 ms)
      const u32 sl = vocab_lookup_insert_pooled<CharT>(vt, tv, hs.fin(), len, gpos, wv, word);
      u64 act = ballot64(sl != 0xFFFFFFFFu);
      while (act) {
        const u32 lead = (u32)__builtin_ctzll(act);
        const u32 s0 = __shfl(sl, lead, 64);
        const u64 m = ballot64(sl == s0) & act;
        if (lane == lead) atomicAdd(reinterpret_cast<u32*>(vt.slots + (u64)s0 * vt.stride) + 3, (u32)__builtin_popcountll(m));
        act &= ~m;
      }
      continue;
    }
    // (a table that numbers its terms returns the term's id, i.e. the sort key itself -- or VOC_UNRESOLVED | slot)
    u32 slot = (FAST || vt.id_ctr) ? vocab_lookup_insert_ids<CharT>(vt, tv, hs.fin(), len, gpos, wv, word, (blockIdx.x & (VOC_SHARDS - 1u)) * VOC_CTR_STRIDE)
               : vt.pool ? vocab_lookup_insert_pooled<CharT>(vt, tv, hs.fin(), len, gpos, wv, word)
                         : vocab_lookup_insert<CharT>(vt, tv, hs.fin(), len, gpos, word);
    any_unres |= (FAST || vt.id_ctr != nullptr) && slot != 0xFFFFFFFFu && (slot & VOC_UNRESOLVED) != 0u;
    // (one LDS atomic per token; gathering the cold bucket's third of the tokens with a ballot first measured slower: the kernel
    //  is bound by instruction issue, not by the same-address atomics)
    if (do_hist && (slot & VOC_UNRESOLVED) == 0u) atomicAdd(&s_h[slot < n_hot_h ? slot : n_hot_h], 1u);   // (overflow = 0xFFFFFFFF has the bit too)
#endif
#if DCD_TOK_NT
    {
      typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
      u32x2_t tv2; tv2.x = slot; tv2.y = coord_of(gpos);
      __builtin_nontemporal_store(tv2, reinterpret_cast<u32x2_t*>(tuples + obase + k));
    }
#else
    tuples[obase + k] = make_uint2(slot, coord_of(gpos));
#endif
    if (sampled && slot != 0xFFFFFFFFu) atomicAdd(&sample_cnt[slot], 1u);
  }
  if (do_hist) {
    __syncthreads();
    reinterpret_cast<u16*>(sample_cnt)[group * 256u + threadIdx.x] = (u16)s_h[threadIdx.x];   // (a group holds <= 4 * 318 tokens)
  }
  // groups that hold unresolved keys go on the fix-up list: unres[0] = their number, unres[1 ..] = the first UNRES_CAP of them
  // (a wave reports its group; the list may name a group more than once -- the fix-up is idempotent)
  if ((FAST || unres != nullptr) && ballot64(any_unres) != 0ull && lane == 0) {
    const u32 k = atomicAdd(unres, 1u);
    if (k < UNRES_CAP) unres[1 + k] = (u32)group;
  }
}

// explicit postings (Builder.Add(code, coord), Build.cs:324): codes live in the pool, verbatim
template <typename CharT>
__global__ void k_explicit(const u32* __restrict__ xoff, const u32* __restrict__ xlen, const u32* __restrict__ xcoord, u64 n,
                           TextView<CharT> tv, VocabTable vt, uint2* __restrict__ tuples) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  constexpr u32 CS = sizeof(CharT);
  const u32 len = xlen[i], nbytes = len * CS, nw = (nbytes + 3u) >> 2;
  const u32* pool32 = reinterpret_cast<const u32*>(tv.pool);
  const u64 boff = (u64)xoff[i] * CS, pbytes = tv.n_pool * CS;
  auto word = [pool32, boff, pbytes, nbytes](u32 q) -> u32 { return str_word(pool32, boff, q, pbytes) & tail_mask(nbytes, q); };
  HashState hs;
  hs.init(len);
  for (u32 q = 0; q < nw; q++) hs.add(word(q));
  u32 slot = vocab_lookup_insert<CharT>(vt, tv, hs.fin(), len, tv.n_chars + xoff[i], word);
  tuples[i] = make_uint2(slot, xcoord[i]);
}

__global__ void k_set_u64(u64* p, u64 v) { *p = v; }
// rows[g][d] (u16: tokens of digit d in group g, written by k_tok_emit) -> hist[d * nblk + b] += sum over the groups of radix
// block b (the epilogue has already added the keys it resolved: plain read-modify-write, every element has one owner)
__global__ __launch_bounds__(256) void k_hist_rows(const u16* __restrict__ rows, u64 n_groups, u32 tpb, u32 nblk, u32* __restrict__ hist) {
  const u32 b = blockIdx.x, d = threadIdx.x;
  const u64 g0 = (u64)b * tpb, g1 = g0 + tpb < n_groups ? g0 + tpb : n_groups;
  u32 s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  u64 g = g0;
  for (; g + 4 <= g1; g += 4) {
    s0 += rows[(g + 0) * 256 + d]; s1 += rows[(g + 1) * 256 + d]; s2 += rows[(g + 2) * 256 + d]; s3 += rows[(g + 3) * 256 + d];
  }
  for (; g < g1; g++) s0 += rows[g * 256 + d];
  hist[(u64)d * nblk + b] += s0 + s1 + s2 + s3;
}
// explicit page offsets -> one bit per character of the text: bit i = a page starts at i (TokParams::page_bits)
__global__ void k_page_bits(const u64* __restrict__ page_off, u64 n, u32* __restrict__ bits) {
  u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const u64 o = page_off[p];
  atomicOr(&bits[o >> 5], 1u << (o & 31));
}
// coordinate of the last tuple of a tile-strided buffer (the last non-empty tile is almost always the last one)
__global__ void k_last_strided_coord(const uint2* __restrict__ tuples, const u32* __restrict__ cnt, u64 n_tiles, u32 stride, u32* out) {
  for (u64 t = n_tiles; t-- > 0;) {
    const u32 c = cnt[t];
    if (c) { *out = tuples[t * stride + c - 1].y; return; }
  }
  *out = 0;
}
__global__ void k_term_first_last(const uint2* __restrict__ sorted, const u64* __restrict__ item_start, u64 n_terms, u64 base,
                                  u64* __restrict__ first, u64* __restrict__ last) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  first[t] = base + sorted[item_start[t]].y;
  last[t] = base + sorted[item_start[t + 1] - 1].y;
}
// ---- early ids: the table numbers its terms (vocab_lookup_insert_ids) ---------------------------------------------------------
// id = local * VOC_SHARDS + shard with local < cnt[shard].  The DENSE id of a term is the number of used ids below its id, so
// that the order of the ids (the sort order of the tuples) is the order of the dense ids; with the VOC_SHARDS counts it is
// that many min / add operations -- tabulated once per build (beside the tuple sort), looked up at term heads only.
__global__ __launch_bounds__(256) void k_id_dense_table(IdCounts n, u32 n_hot, u32 id_range, u32* __restrict__ dense) {
  const u32 id = blockIdx.x * 256u + threadIdx.x;
  if (id < id_range) dense[id] = dense_of_key(n.c, n_hot, id);
}
// Behind the emit kernel of a build with early ids, ONE launch:
//  * unresolved keys (VOC_UNRESOLVED | slot) -> ids, now that every inserter has published its id; one wave per listed tile,
//    or -- when the list overflowed -- every tile;
//  * the number of postings = sum of the tile counts (*n_tok, zeroed; one atomic per workgroup);
//  * the coordinate of the last posting (maxCoord, Build.cs:326).
__global__ __launch_bounds__(256) void k_emit_epilogue(uint2* __restrict__ tuples, const u32* __restrict__ tile_cnt, u64 n_tiles, u32 stride,
                                                       const u32* __restrict__ unres, const u64* __restrict__ slots, u32 slot_stride,
                                                       u64* __restrict__ n_tok, u32* __restrict__ last_coord,
                                                       u32* __restrict__ hist_a, u32 hist_tpb, u32 hist_nblk, const u32* __restrict__ n_hot_p) {
  __shared__ u64 s_sum[4];
  const u32 n = unres[0];
  if (n) {
    const u64 wave = (u64)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (u64)gridDim.x * 4;
    const bool all = n > UNRES_CAP;
    const u64 count = all ? n_tiles : n;
    for (u64 k = wave; k < count; k += n_waves) {
      const u64 t = all ? k : unres[1 + k];
      const u32 c = tile_cnt[t];
      for (u32 j = lane_id(); j < c; j += 64) {
        const u32 x = tuples[t * stride + j].x;
        if (x != 0xFFFFFFFFu && (x & VOC_UNRESOLVED)) {
          const u32 slot = x & ~VOC_UNRESOLVED;
          const u32 id = reinterpret_cast<const u32*>(slots + (u64)slot * slot_stride)[2] - 1u;
          // (the list may name a group twice, see k_tok_emit: the visit that REPLACES the key -- one compare-and-swap wins --
          //  is the one that counts it in the first radix pass's histogram)
          const u32 was = atomicCAS(&tuples[t * stride + j].x, x, id);
          if (hist_a && was == x) { const u32 nh = *n_hot_p; atomicAdd(&hist_a[(u64)(id < nh ? id : nh) * hist_nblk + (u32)(t / hist_tpb)], 1u); }
        }
      }
    }
  }
  u64 s = 0;
  for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < n_tiles; t += (u64)gridDim.x * 256) s += tile_cnt[t];
  s = wave_sum(s);
  if (lane_id() == 0) s_sum[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const u64 tot = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    if (tot) atomicAdd((unsigned long long*)n_tok, (unsigned long long)tot);
    if (blockIdx.x == 0) {       // (the last non-empty tile is almost always the last one)
      u32 y = 0;
      for (u64 t = n_tiles; t-- > 0;) {
        const u32 c = tile_cnt[t];
        if (c) { y = tuples[t * stride + c - 1].y; break; }
      }
      *last_coord = y;
    }
  }
}
// ---- hot / cold split of the vocabulary ---------------------------------------------------------------------------
// The H slots with the largest sample counts are "hot" (Zipf: a few hundred terms hold most postings).  Dense ids: hot
// terms 0 .. n_hot-1 (in slot order), the others n_hot + (rank among the used cold slots).  The first radix pass gives
// every hot term a bucket of its own -- their tuples are final after ONE pass -- and only the cold tuples are sorted further.
constexpr int HOT_BINS = 4096;
__global__ __launch_bounds__(256) void k_hot_hist(const u32* __restrict__ cnt, u64 cap, u32* __restrict__ hist) {
  __shared__ u32 s_h[HOT_BINS];
  for (u32 i = threadIdx.x; i < HOT_BINS; i += 256) s_h[i] = 0;
  __syncthreads();
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < cap; i += (u64)gridDim.x * 256) {
    const u32 c = cnt[i];
    if (c) atomicAdd(&s_h[c < HOT_BINS ? c : HOT_BINS - 1], 1u);
  }
  __syncthreads();
  for (u32 i = threadIdx.x; i < HOT_BINS; i += 256) { const u32 v = s_h[i]; if (v) atomicAdd(&hist[i], v); }
}
// one workgroup: theta = the smallest sample count such that at most H slots reach it (0xFFFFFFFF: nothing is hot)
__global__ __launch_bounds__(1024) void k_hot_pick(const u32* __restrict__ hist, u32 H, u32* __restrict__ theta) {
  __shared__ u32 lds[17];
  __shared__ u32 s_best;
  constexpr int PER = HOT_BINS / 1024;
  const u32 t = threadIdx.x;
  u32 v[PER], s = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) { v[k] = t * PER + k ? hist[t * PER + k] : 0u; s += v[k]; }   // (bin 0 = never sampled)
  if (t == 0) s_best = 0xFFFFFFFFu;
  u32 tot;
  u32 before = block_scan_excl<u32, 16>(s, lds, &tot);          // slots in lower bins
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const u32 b = t * PER + k;
    if (b >= 1 && tot - before <= H) atomicMin(&s_best, b);      // tot - before = slots with count >= b
    before += v[k];
  }
  __syncthreads();
  if (t == 0) *theta = s_best;
}
// Builds with early ids, behind the SAMPLING launch of the emit kernel (which counted the tokens of a few groups in their
// records and listed the slots it claimed): ONE workgroup picks theta as k_hot_pick does and numbers the sampled terms -- those
// that reach theta 0 .. n_hot - 1, the others from n_hot on through the id counters, like the real launch will (they stay in the
// table: emptied again, the next few thousand terms of the distribution would meet the real launch as an insert storm).
// The hot terms are then exactly the ids below n_hot: the first radix pass gives each a bucket of its own (RsStrided::hot_n).
__global__ __launch_bounds__(1024) void k_hot_select(u64* __restrict__ slots, u32 stride, const u32* __restrict__ list, u32 H, u32* __restrict__ n_hot_out,
                                                     u32* __restrict__ id_ctr) {
  __shared__ u32 s_h[HOT_BINS];
  __shared__ u32 lds[17];
  __shared__ u32 s_best, s_nhot, s_n, s_c;
  constexpr int PER = HOT_BINS / 1024;
  constexpr int K = 4;                 // list entries a thread keeps in registers (4096 sampled terms; beyond that: re-read)
  const u32 t = threadIdx.x, n = list[0];
  for (u32 i = t; i < HOT_BINS; i += 1024) s_h[i] = 0;
  if (t == 0) { s_best = 0xFFFFFFFFu; s_nhot = 0; s_n = 0; s_c = 0; }
  __syncthreads();
  u32 sl[K], cn[K];
#pragma unroll
  for (int k = 0; k < K; k++) { const u32 i = t + 1024u * k; sl[k] = i < n ? list[1 + i] : 0xFFFFFFFFu; }
#pragma unroll
  for (int k = 0; k < K; k++) cn[k] = sl[k] != 0xFFFFFFFFu ? reinterpret_cast<const u32*>(slots + (u64)sl[k] * stride)[3] : 0u;
#pragma unroll
  for (int k = 0; k < K; k++) if (sl[k] != 0xFFFFFFFFu) atomicAdd(&s_h[cn[k] < HOT_BINS ? cn[k] : HOT_BINS - 1], 1u);
  for (u32 i = t + 1024u * K; i < n; i += 1024) {
    const u32 c = reinterpret_cast<const u32*>(slots + (u64)list[1 + i] * stride)[3];
    atomicAdd(&s_h[c < HOT_BINS ? c : HOT_BINS - 1], 1u);
  }
  __syncthreads();
  u32 v[PER], s = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) { v[k] = t * PER + k ? s_h[t * PER + k] : 0u; s += v[k]; }
  u32 tot;
  u32 before = block_scan_excl<u32, 16>(s, lds, &tot);          // slots in lower bins
  u32 reach[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const u32 b = t * PER + k;
    reach[k] = tot - before;                                      // slots with count >= b
    if (b >= 1 && reach[k] <= H) atomicMin(&s_best, b);
    before += v[k];
  }
  __syncthreads();
  const u32 theta = s_best;
#pragma unroll
  for (int k = 0; k < PER; k++) if (t * PER + k == theta) s_nhot = reach[k];
  __syncthreads();
  // hot terms 0 .. n_hot - 1; the other sampled terms n_hot + j, j = 0 .. n_cold - 1 -- which is what the id counters hand out
  // (id = n_hot + local * VOC_SHARDS + shard) once counter s stands at the number of j with j mod VOC_SHARDS == s
  const u32 n_hot = s_nhot;
  auto number = [&](u32 slot, u32 c) {
    u32* r32 = reinterpret_cast<u32*>(slots + (u64)slot * stride);
    const bool hot = c >= theta;
    const u32 j = atomicAdd(hot ? &s_n : &s_c, 1u);
    *reinterpret_cast<uint2*>(r32 + 2) = make_uint2((hot ? j : n_hot + j) + 1u, 0u);      // id + 1, count word back to 0
  };
#pragma unroll
  for (int k = 0; k < K; k++) if (sl[k] != 0xFFFFFFFFu) number(sl[k], cn[k]);
  for (u32 i = t + 1024u * K; i < n; i += 1024) {
    const u32 slot = list[1 + i];
    number(slot, reinterpret_cast<const u32*>(slots + (u64)slot * stride)[3]);
  }
  __syncthreads();
  if (t < VOC_SHARDS) id_ctr[t * VOC_CTR_STRIDE] = (s_c + (VOC_SHARDS - 1u) - t) / VOC_SHARDS;
  if (t == 0) *n_hot_out = n_hot;
}
struct HotUsedF {   // low word: the slot is used; high word: it holds a hot term
  const u64* slots;
  const u32* cnt;
  const u32* theta;
  u32 stride;      // u64 units per slot
  __device__ __forceinline__ u64 operator()(u64 i) const {
    if (!slots[i * stride]) return 0ull;
    return 1ull | ((u64)(cnt[i] >= *theta ? 1u : 0u) << 32);
  }
};
__global__ void k_dense_table(const u64* __restrict__ slots, const u32* __restrict__ cnt, const u32* __restrict__ theta,
                              const u64* __restrict__ excl, const u64* __restrict__ total, u64 cap, u32* __restrict__ dense, u32 stride) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  u32 d = 0;
  if (slots[i * stride]) {
    const u64 e = excl[i];
    const u32 n_hot = (u32)(*total >> 32), hot_before = (u32)(e >> 32), used_before = (u32)e;
    d = cnt[i] >= *theta ? hot_before : n_hot + (used_before - hot_before);
  }
  dense[i] = d;
}

// id-keyed builds (tuples sorted by dense id d, terms delivered in ordinal rank order r; d = dense_of_rank[r])
__global__ void k_term_first_last_perm(const uint2* __restrict__ sorted, const u64* __restrict__ item_start_id, const u32* __restrict__ dense_of_rank,
                                       u64 n_terms, u64 base, u64* __restrict__ first, u64* __restrict__ last) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_terms) return;
  const u32 d = dense_of_rank[r];
  first[r] = base + sorted[item_start_id[d]].y;
  last[r] = base + sorted[item_start_id[d + 1] - 1].y;
}
__global__ void k_copy_u64(u64* dst, const u64* src) { *dst = *src; }
__global__ void k_copy_u32(u32* dst, const u32* src) { *dst = *src; }
struct PermLenF {   // length of the term with rank r in an array of starts indexed by dense id
  const u32* dense_of_rank;
  const u64* start_id;
  __device__ __forceinline__ u64 operator()(u64 r) const { const u32 d = dense_of_rank[r]; return start_id[d + 1] - start_id[d]; }
};
__global__ void k_perm_src_start(const u32* __restrict__ dense_of_rank, const u64* __restrict__ start_id, u64 n, u64* __restrict__ out) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = start_id[dense_of_rank[r]];
}
// END coordinate of page p of a bulk batch = coordinate of the batch's first character + the offset behind the page
__global__ void k_page_end_from_off(u64* __restrict__ out, const u64* __restrict__ off, u64 n, u64 c0) {
  const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) out[p] = c0 + off[p + 1];
}
__global__ void k_uniform_page_end(u64* out, u64 n, u64 base, u64 chars) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = base + (i + 1) * chars;
}
__global__ void k_swap_xy(uint2* p, u64 n) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { uint2 v = p[i]; p[i] = make_uint2(v.y, v.x); }
}
__global__ void k_max_y(const uint2* p, u64 n, u32* out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 v = i < n ? p[i].y : 0u;
  v = wave_max(v);
  if (lane_id() == 0) atomicMax(out, v);
}

// ".index" record sizes: 7-bit-length-prefixed UTF-8 key + i32 unit count + units (Appendix A.3)
__device__ __forceinline__ u32 utf8_len_units(const u16* s, u32 n) {
  u32 b = 0;
  for (u32 i = 0; i < n; i++) {
    u32 c = s[i];
    if (c < 0x80) b += 1;
    else if (c < 0x800) b += 2;
    else if (c >= 0xD800 && c <= 0xDBFF && i + 1 < n && s[i + 1] >= 0xDC00 && s[i + 1] <= 0xDFFF) { b += 4; i++; }
    else b += 3;
  }
  return b;
}
struct RecSizeF {
  const u16* chars;
  const u32* str_off;
  const u64* unit_start;
  __device__ __forceinline__ u64 operator()(u64 r) const {
    u32 kb = utf8_len_units(chars + str_off[r], str_off[r + 1] - str_off[r]);
    return (u64)(kb < 128 ? 1 : 2) + kb + 4 + 2 * (unit_start[r + 1] - unit_start[r]);
  }
};

// ---- id-keyed builds: the whole term directory in ordinal order in THREE launches ------------------------------------------
// Per rank r (d = dense_of_rank[r]) three lengths -- units, items, bytes of the ".index" record -- are scanned together
// (reduce -> scan of the partials -> apply); the apply kernel also leaves the source position of the term's units and its
// first / last coordinate.  (Before: three scans of three launches each + five small kernels, ~0.1 ms of launch latency.)
struct PermDirF {
  const u32* dense_of_rank;
  const u64* us_id;        // [V] unit start per dense id (the entry [V] would be *n_units)
  const u64* is_id;        // [V] item start per dense id (the entry [V] would be n_items)
  const u64* n_units;      // device: total number of units
  u64 n_items, V;
  const u16* chars;        // the terms' strings, ordinal order
  const u32* str_off;
  __device__ __forceinline__ void operator()(u64 r, u64& units, u64& items, u64& rec, u32* d_out = nullptr) const {
    const u32 d = dense_of_rank[r];
    const bool last = (u64)d + 1 == V;
    units = (last ? *n_units : us_id[d + 1]) - us_id[d];
    items = (last ? n_items : is_id[d + 1]) - is_id[d];
    const u32 kb = utf8_len_units(chars + str_off[r], str_off[r + 1] - str_off[r]);
    rec = (u64)(kb < 128 ? 1 : 2) + kb + 4 + 2 * units;           // (as RecSizeF)
    if (d_out) *d_out = d;
  }
};
// partial: three arrays of `pstride` entries (units, items, bytes), one entry per workgroup
__global__ __launch_bounds__(SCAN_THREADS) void k_dir_reduce(PermDirF f, u64 n, u64* __restrict__ partial, u64 pstride) {
  __shared__ u64 lds[3][4];
  const u64 base = (u64)blockIdx.x * SCAN_TILE;
  u64 a = 0, b = 0, c = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    const u64 i = base + (u64)j * SCAN_THREADS + threadIdx.x;
    if (i < n) { u64 x, y, z; f(i, x, y, z); a += x; b += y; c += z; }
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if (lane_id() == 0) { lds[0][threadIdx.x >> 6] = a; lds[1][threadIdx.x >> 6] = b; lds[2][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x < 3) partial[threadIdx.x * pstride + blockIdx.x] = lds[threadIdx.x][0] + lds[threadIdx.x][1] + lds[threadIdx.x][2] + lds[threadIdx.x][3];
}
// one workgroup: the three partial arrays -> exclusive in place; totals[0..3) = the grand totals (units, items, bytes)
__global__ __launch_bounds__(1024) void k_dir_partials(u64* __restrict__ partial, u64 pstride, u64 nb, u64* __restrict__ totals) {
  constexpr int PER = 8;
  __shared__ u64 lds[17];
  for (int c = 0; c < 3; c++) {
    u64* p = partial + (u64)c * pstride;
    u64 carry = 0;
    for (u64 b0 = 0; b0 < nb; b0 += 1024 * PER) {
      const u64 i0 = b0 + (u64)threadIdx.x * PER;
      u64 v[PER], s = 0;
#pragma unroll
      for (int j = 0; j < PER; j++) { v[j] = i0 + j < nb ? p[i0 + j] : 0ull; s += v[j]; }
      u64 tot;
      u64 ex = carry + block_scan_excl<u64, 16>(s, lds, &tot);
#pragma unroll
      for (int j = 0; j < PER; j++) { if (i0 + j < nb) p[i0 + j] = ex; ex += v[j]; }
      carry += tot;
    }
    if (threadIdx.x == 0) totals[c] = carry;
  }
}
// unit_start / item_start / rec_off [n + 1] (the last entries = the totals), src_start / first / last [n]
__global__ __launch_bounds__(SCAN_THREADS) void k_dir_apply(PermDirF f, u64 n, const u64* __restrict__ partial, u64 pstride, const u64* __restrict__ totals,
                                                           u64* __restrict__ unit_start, u64* __restrict__ item_start, u64* __restrict__ rec_off,
                                                           u64* __restrict__ src_start, const uint2* __restrict__ sorted, u64 base_coord,
                                                           u64* __restrict__ first, u64* __restrict__ last, u64 n_hot32) {
  // (round 4: the first n_hot32 postings are 4-byte coordinates, EncSrcPairs::n_hot32)
  auto coord_at = [sorted, n_hot32](u64 idx) -> u32 { return idx < n_hot32 ? reinterpret_cast<const u32*>(sorted)[idx] : sorted[idx].y; };
  __shared__ u64 lds[5];
  const u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
  u64 va[SCAN_ITEMS], vb[SCAN_ITEMS], vc[SCAN_ITEMS];
  u64 sa = 0, sb = 0, sc = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    const u64 i = base + j;
    va[j] = vb[j] = vc[j] = 0;
    if (i < n) {
      u32 d;
      f(i, va[j], vb[j], vc[j], &d);
      const u64 is = f.is_id[d];
      src_start[i] = f.us_id[d];
      first[i] = base_coord + coord_at(is);
      last[i] = base_coord + coord_at(is + vb[j] - 1);
    }
    sa += va[j]; sb += vb[j]; sc += vc[j];
  }
  u64 tot;
  u64 ea = block_scan_excl<u64, 4>(sa, lds, &tot) + partial[blockIdx.x];
  u64 eb = block_scan_excl<u64, 4>(sb, lds, &tot) + partial[pstride + blockIdx.x];
  u64 ec = block_scan_excl<u64, 4>(sc, lds, &tot) + partial[2 * pstride + blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; j++) {
    const u64 i = base + j;
    if (i < n) { unit_start[i] = ea; item_start[i] = eb; rec_off[i] = ec; }
    ea += va[j]; eb += vb[j]; ec += vc[j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { unit_start[n] = totals[0]; item_start[n] = totals[1]; rec_off[n] = totals[2]; }
}

__global__ void k_img_headers(const u16* __restrict__ chars, const u32* __restrict__ str_off, const u64* __restrict__ unit_start,
                              const u64* __restrict__ rec_off, u64 n, u64 img_base, u8* __restrict__ img, u64* __restrict__ pay_off) {
  u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const u16* s = chars + str_off[r];
  u32 len = str_off[r + 1] - str_off[r];
  u32 kb = utf8_len_units(s, len);
  u64 o = img_base + rec_off[r];
  if (kb < 128) img[o++] = (u8)kb;
  else { img[o++] = (u8)(kb | 0x80); img[o++] = (u8)(kb >> 7); }
  for (u32 i = 0; i < len; i++) {
    u32 c = s[i];
    if (c >= 0xD800 && c <= 0xDBFF && i + 1 < len && s[i + 1] >= 0xDC00 && s[i + 1] <= 0xDFFF) {
      c = 0x10000 + ((c - 0xD800) << 10) + (s[i + 1] - 0xDC00);
      i++;
    } else if (c >= 0xD800 && c <= 0xDFFF) c = 0xFFFD;
    if (c < 0x80) img[o++] = (u8)c;
    else if (c < 0x800) { img[o++] = (u8)(0xC0 | (c >> 6)); img[o++] = (u8)(0x80 | (c & 0x3F)); }
    else if (c < 0x10000) { img[o++] = (u8)(0xE0 | (c >> 12)); img[o++] = (u8)(0x80 | ((c >> 6) & 0x3F)); img[o++] = (u8)(0x80 | (c & 0x3F)); }
    else { img[o++] = (u8)(0xF0 | (c >> 18)); img[o++] = (u8)(0x80 | ((c >> 12) & 0x3F)); img[o++] = (u8)(0x80 | ((c >> 6) & 0x3F)); img[o++] = (u8)(0x80 | (c & 0x3F)); }
  }
  u32 nu = (u32)(unit_start[r + 1] - unit_start[r]);
  img[o++] = (u8)nu; img[o++] = (u8)(nu >> 8); img[o++] = (u8)(nu >> 16); img[o++] = (u8)(nu >> 24);
  pay_off[r] = o;
}

// units -> image payloads.  A workgroup owns a tile of 8192 units, each of its four waves a span of 2048 (4 x 8 units = 4 x
// 16 bytes per lane, lane-interleaved) and the tile's first / last term are found once; a span inside ONE term (by weight
// almost all of a Zipf index) looks its term up once per wave and streams; 8 units of one term move as one 16-byte store
// at the payload's (arbitrary) byte alignment.
constexpr int IMG_UPT = 8;  // units per chunk
constexpr int IMG_THREADS = 256;
constexpr int IMG_CPL = 4;  // chunks per lane
constexpr int IMG_SPAN = 64 * IMG_UPT * IMG_CPL;
constexpr int IMG_TILE = IMG_SPAN * (IMG_THREADS / 64);
typedef u32 __attribute__((aligned(1))) u32_a1;
__device__ __forceinline__ u64 img_term_of(const u64* __restrict__ unit_start, u64 lo, u64 hi, u64 u) {   // last t in [lo, hi) with unit_start[t] <= u
  while (lo + 1 < hi) { u64 m = (lo + hi) >> 1; if (unit_start[m] <= u) lo = m; else hi = m; }
  return lo;
}
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
// (round 4: ONE 16-byte access at the run's byte / 2-byte alignment -- the unaligned access mode of the memory pipeline takes it --
//  instead of four dword accesses 16 bytes apart: a quarter of the memory instructions)
__device__ __forceinline__ void img_copy8(const u16* __restrict__ units, u64 u0, u8* __restrict__ img, u64 dst) {
  const uint4 v = *reinterpret_cast<const uint4*>(units + u0);     // units is 16-byte aligned at u0 (a multiple of 8)
  u32x4_t w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_memcpy(img + dst, &w, 16);
}
// PERM: the source holds the terms in another order (src_start[t] = first unit of term t in `units`); the units also go,
// in term order, to units_out (the index's posting array): one read, two writes.  The source run of a term starts at an
// arbitrary 2-byte boundary (dword loads at that alignment), units_out[u0] is 16-byte aligned.
template <bool W_IMG, bool W_OUT>
__device__ __forceinline__ void img_copy8_perm(const u16* __restrict__ units, u64 s0, u8* __restrict__ img, u64 dst, u16* __restrict__ units_out,
                                               u64 u0) {
  u32x4_t w;
  __builtin_memcpy(&w, units + s0, 16);
  if (W_IMG) __builtin_memcpy(img + dst, &w, 16);
  if (W_OUT) { uint4 v; v.x = w.x; v.y = w.y; v.z = w.z; v.w = w.w; *reinterpret_cast<uint4*>(units_out + u0) = v; }
}
// W_IMG / W_OUT (PERM only): which of the two destinations is written.  Round 3: the BUILD writes the image only (the units stay
// where the pack kernel left them, in sort-key order); the term-ordered posting array is an auxiliary structure of a QUERYABLE
// index and is made when a handle adopts the build (index_materialize_units: the same kernel with W_IMG = false).
template <bool PERM, bool W_IMG = true, bool W_OUT = true>
__global__ __launch_bounds__(IMG_THREADS) void k_img_units(const u16* __restrict__ units, u64 n_units, const u64* __restrict__ unit_start,
                                                          u64 n_terms, const u64* __restrict__ pay_off, u8* __restrict__ img,
                                                          const u64* __restrict__ src_start, u16* __restrict__ units_out) {
  __shared__ u64 s_t[2];
  const u64 tile_base = (u64)blockIdx.x * IMG_TILE;
  if (threadIdx.x == 0) s_t[0] = img_term_of(unit_start, 0, n_terms, tile_base);
  if (threadIdx.x == 64) { u64 e = tile_base + IMG_TILE; s_t[1] = img_term_of(unit_start, 0, n_terms, (e < n_units ? e : n_units) - 1); }
  __syncthreads();
  const u32 lane = threadIdx.x & 63;
  const u64 span = tile_base + (u64)(threadIdx.x >> 6) * IMG_SPAN;
  if (span >= n_units) return;
  const u64 t0 = img_term_of(unit_start, s_t[0], s_t[1] + 1, span);          // wave-uniform
  if (span + IMG_SPAN <= unit_start[t0 + 1]) {                               // the whole span inside one term
    const u64 dst0 = W_IMG ? pay_off[t0] + 2 * (span - unit_start[t0]) : 0;
    if (PERM) {
      const u64 s0 = src_start[t0] + (span - unit_start[t0]);
#pragma unroll
      for (int c = 0; c < IMG_CPL; c++) {
        const u64 o = (u64)(c * 64 + lane) * IMG_UPT;
        img_copy8_perm<W_IMG, W_OUT>(units, s0 + o, img, dst0 + o * 2, units_out, span + o);
      }
      return;
    }
#pragma unroll
    for (int c = 0; c < IMG_CPL; c++) img_copy8(units, span + (u64)(c * 64 + lane) * IMG_UPT, img, dst0 + (u64)(c * 64 + lane) * IMG_UPT * 2);
    return;
  }
  // (round 4: the four chunks of a lane find their terms TOGETHER -- the searches, then the directory loads, advance in lock step,
  //  their dependent loads overlap -- instead of chunk after chunk)
  u64 tc[IMG_CPL], lo[IMG_CPL], hi[IMG_CPL];
#pragma unroll
  for (int c = 0; c < IMG_CPL; c++) { lo[c] = t0; hi[c] = s_t[1] + 1; }
  for (bool more = true; more;) {
    more = false;
    u64 mv[IMG_CPL], sv[IMG_CPL];
#pragma unroll
    for (int c = 0; c < IMG_CPL; c++) { mv[c] = (lo[c] + hi[c]) >> 1; sv[c] = unit_start[mv[c]]; }
#pragma unroll
    for (int c = 0; c < IMG_CPL; c++) {
      const u64 u0 = min(span + (u64)(c * 64 + lane) * IMG_UPT, n_units - 1);
      if (lo[c] + 1 < hi[c]) { if (sv[c] <= u0) lo[c] = mv[c]; else hi[c] = mv[c]; }
      more |= lo[c] + 1 < hi[c];
    }
  }
  u64 nx[IMG_CPL], us[IMG_CPL], po[IMG_CPL], ss[IMG_CPL];
#pragma unroll
  for (int c = 0; c < IMG_CPL; c++) {
    tc[c] = lo[c];
    nx[c] = unit_start[tc[c] + 1];
    us[c] = unit_start[tc[c]];
    po[c] = W_IMG ? pay_off[tc[c]] : 0;
    ss[c] = PERM ? src_start[tc[c]] : 0;
  }
#pragma unroll
  for (int c = 0; c < IMG_CPL; c++) {
    const u64 u0 = span + (u64)(c * 64 + lane) * IMG_UPT;
    if (u0 >= n_units) break;
    u64 t = tc[c];
    u64 next = nx[c];
    u64 dst = W_IMG ? po[c] + 2 * (u0 - us[c]) : 0;
    u64 sp = PERM ? ss[c] + (u0 - us[c]) : u0;                       // source position of unit u0
    if (u0 + IMG_UPT <= next) {
      if (PERM) img_copy8_perm<W_IMG, W_OUT>(units, sp, img, dst, units_out, u0); else img_copy8(units, u0, img, dst);
      continue;
    }
    for (int k = 0; k < IMG_UPT; k++) {
      u64 u = u0 + k;
      if (u >= n_units) break;
      while (u >= next) { t++; next = unit_start[t + 1]; if (W_IMG) dst = pay_off[t]; if (PERM) sp = src_start[t]; }
      u16 v = units[PERM ? sp : u];
      if (PERM) { if (W_OUT) units_out[u] = v; sp++; }
      if (W_IMG) { img[dst] = (u8)v; img[dst + 1] = (u8)(v >> 8); }
      dst += 2;
    }
  }
}

__global__ void k_put_u64_bytes(u8* img, u64 v) {
  for (int i = 0; i < 8; i++) img[i] = (u8)(v >> (8 * i));
}

}  // namespace dcd

using namespace dcd;

// ================================================================================================
// handle
// ================================================================================================
struct ExplicitPosting { u32 off, len; u64 coord; bool is_word; };
struct BuildMid;

struct dcd_build {
  dcd_build_opts opts;
  int device = 0;
  hipStream_t st = nullptr;
  bool finished = false;
  // text
  DevBuf<u8> text_own;        // raw bytes of the staged text (char size 1 or 2)
  const void* text_dev = nullptr;
  bool text_borrowed = false;
  u64 n_chars = 0, text_cap = 0;
  std::vector<u64> page_off;  // [n_pages + 1] text-buffer offsets of the BODY pages staged on the device
  // ... or, for ONE bulk batch on device text (one document per page, id "1": dcd_build_add_pages_dev as bench.py and a C# host that
  // indexes a prepared buffer call it), the caller's offset array as it came, in pinned memory: uploaded from there without
  // another copy; the pages' END coordinates are derived from it on the device and, lazily, on the host (PageTable::end)
  std::shared_ptr<u64> bulk_off;
  u64 bulk_n = 0, bulk_c0 = 0;
  DevBuf<u64> d_bulk_off;     // ... and its device copy, uploaded piece by piece while the host still copies the next piece
  PageTable pages;
  u64 coord_cursor = 0;       // global coordinate of the next character (IndexTask's `coord`, Build.cs:459)
  std::vector<u32> run_off, run_delta;  // coordinate runs: text index -> local coordinate offset (header pages)
  // explicit postings
  std::vector<u16> xpool;     // code characters (UTF-16 units; narrowed for Latin-1 builds)
  std::vector<ExplicitPosting> xps;
  bool last_is_explicit = false;
  u64 last_explicit_coord = 0;
  ustr pending_source;        // AddDoc bookkeeping
  // results
  DeviceIndex ix;
  DevBuf<u8> image;
  u64 image_bytes = 0;
  std::vector<u8> list_image;
  dcd_build_stats stats;
  bool adopted = false;       // DeviceIndex moved into a dcd_index
  std::shared_ptr<BuildMid> mid;   // between dcd_build_words / phase 1 and phase 2
  int csize() const { return opts.encoding == DCD_ENC_UTF16LE ? 2 : 1; }
};

namespace dcd {
// UTF-16 class/lower tables on the device (built lazily per device from dcd_unicode.cpp)
int32_t get_char_tables16(int device, CharTables16* out);
void host_char_tables16(const u8** cls, const u16** lower);
DeviceIndex* build_take_index(dcd_build* b) {
  if (!b->finished || b->adopted || b->ix.shard_only || b->ix.units_by_key.p) return nullptr;   // (shard-only: nothing to hand over; else the caller materialises the units first)
  b->adopted = true;
  return &b->ix;
}
int build_device(dcd_build* b) { return b->device; }
DeviceIndex* build_peek_index(dcd_build* b) { return (b->finished && !b->adopted) ? &b->ix : nullptr; }
hipStream_t build_stream(dcd_build* b) { return b->st; }
int32_t index_make_image(const DeviceIndex& ix, hipStream_t st, DevBuf<u8>& image, u64* image_bytes) {
  const int dev = ix.device;
  DCD_TRY(ensure_device(dev));
  StreamScope ss_(st);
  const u64 V = ix.n_terms, U = ix.n_units;
  DevBuf<u64> rec_off, pay_off, scr64b;
  DCD_TRY(rec_off.alloc(dev, V + 1));
  DCD_TRY(pay_off.alloc(dev, V + 1));
  DCD_TRY(scr64b.alloc(dev, scan_scratch_elems(V ? V : 1)));
  u64 rec_total = 0;
  if (V) {
    scan_exclusive_f<RecSizeF, u64>(RecSizeF{ix.term_chars.p, ix.term_str_off.p, ix.term_unit_start.p}, rec_off.p, V, scr64b.p, st);
    DCD_HIP(hipMemcpyAsync(&rec_total, scr64b.p + scan_nblocks(V), 8, hipMemcpyDeviceToHost, st));
  }
  DCD_HIP(hipStreamSynchronize(st));
  *image_bytes = 8 + rec_total;
  DCD_TRY(image.alloc(dev, *image_bytes));
  hipLaunchKernelGGL(k_put_u64_bytes, dim3(1), dim3(1), 0, st, image.p, ix.max_coord);
  if (V) {
    hipLaunchKernelGGL(k_img_headers, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ix.term_chars.p, ix.term_str_off.p, ix.term_unit_start.p,
                       rec_off.p, V, (u64)8, image.p, pay_off.p);
    if (U) hipLaunchKernelGGL(k_img_units<false>, dim3((u32)div_up(U, IMG_TILE)), dim3(IMG_THREADS), 0, st, ix.units.p, U, ix.term_unit_start.p, V,
                              pay_off.p, image.p, (const u64*)nullptr, (u16*)nullptr);
  }
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  return DCD_OK;
}
int32_t index_fetch_host(DeviceIndex& ix) {
  if (ix.host_ready) return DCD_OK;
  DCD_TRY(ensure_device(ix.device));
  const u64 V = ix.n_terms;
  u32 total_chars = 0;
  ix.h_term_str_off.resize(V + 1);
  ix.h_term_unit_start.resize(V + 1);
  ix.h_term_item_start.resize(V + 1);
  DCD_HIP(hipMemcpy(ix.h_term_str_off.data(), ix.term_str_off.p, (V + 1) * 4, hipMemcpyDeviceToHost));
  total_chars = ix.h_term_str_off[V];
  ix.h_term_chars.resize(total_chars);
  if (total_chars) DCD_HIP(hipMemcpy(ix.h_term_chars.data(), ix.term_chars.p, (size_t)total_chars * 2, hipMemcpyDeviceToHost));
  // The key pool was sized for V * (longest key) units so that no host round trip sat inside the build / the merge; now that
  // the real total is known, an index that lives on (adopted by a handle, queried) keeps only what it uses -- one 63-unit code
  // would otherwise hold 4-8 x the real size for the lifetime of the index (640 MB instead of ~100 MB at 10 M terms).
  if (ix.term_chars.n > (size_t)total_chars + (size_t)total_chars / 4 + 4096) {
    DevBuf<u16> exact;
    if (exact.alloc(ix.device, total_chars ? total_chars : 1) == DCD_OK) {
      if (total_chars) DCD_HIP(hipMemcpy(exact.p, ix.term_chars.p, (size_t)total_chars * 2, hipMemcpyDeviceToDevice));
      ix.term_chars = std::move(exact);
    }
  }
  DCD_HIP(hipMemcpy(ix.h_term_unit_start.data(), ix.term_unit_start.p, (V + 1) * 8, hipMemcpyDeviceToHost));
  DCD_HIP(hipMemcpy(ix.h_term_item_start.data(), ix.term_item_start.p, (V + 1) * 8, hipMemcpyDeviceToHost));
  ix.host_ready = true;
  return DCD_OK;
}
}  // namespace dcd

namespace dcd {
// The term-ordered posting array of an id-keyed build (DeviceIndex::units), made from the sort-key-ordered units the pack kernel
// left -- when an index handle adopts the build (the query path and dcd_index_get address postings by ordinal term).
int32_t index_materialize_units(DeviceIndex& ix, hipStream_t st) {
  if (!ix.units_by_key.p || ix.units.p) return DCD_OK;
  if (ix.shard_only) DCD_FAIL(DCD_E_STATE, "a DCD_BUILD_SHARD_ONLY build can only be exported");
  DCD_TRY(ensure_device(ix.device));
  StreamScope ss_(st);
  const u64 U = ix.n_units, V = ix.n_terms;
  DCD_TRY(ix.units.alloc(ix.device, U ? U : 1));
  if (U && V)
    hipLaunchKernelGGL((k_img_units<true, false, true>), dim3((u32)div_up(U, IMG_TILE)), dim3(IMG_THREADS), 0, st, (const u16*)ix.units_by_key.p, U,
                       (const u64*)ix.term_unit_start.p, V, (const u64*)nullptr, (u8*)nullptr, (const u64*)ix.unit_src_start.p, ix.units.p);
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  ix.units_by_key.reset();
  ix.unit_src_start.reset();
  return DCD_OK;
}
}  // namespace dcd

static ustr mk16(const u16* p, size_t n) { return ustr((const char16_t*)p, n); }

static int32_t add_explicit(dcd_build* h, const u16* code, size_t len, u64 coord, bool is_word) {
  if (h->mid) DCD_FAIL(DCD_E_STATE, "postings cannot be added after dcd_build_words");
  if (len == 0 || len > MAX_CODE_LEN) DCD_FAIL(DCD_E_UNSUPPORTED, "index codes must be 1..%u UTF-16 units long", MAX_CODE_LEN);
  if (coord < h->opts.coord_base || coord - h->opts.coord_base >= (1ull << 32))
    DCD_FAIL(DCD_E_OVERFLOW, "explicit coordinate outside this segment's 2^32 window");
  if (h->opts.encoding == DCD_ENC_LATIN1)
    for (size_t i = 0; i < len; i++)
      if (code[i] > 0xFF) DCD_FAIL(DCD_E_UNSUPPORTED, "code unit > 0xFF in a LATIN1 build; create the build with DCD_ENC_UTF16LE");
  ExplicitPosting x;
  x.off = (u32)h->xpool.size();
  x.len = (u32)len;
  x.coord = coord;
  x.is_word = is_word;
  h->xpool.insert(h->xpool.end(), code, code + len);
  h->xps.push_back(x);
  h->last_is_explicit = true;
  h->last_explicit_coord = coord;
  return DCD_OK;
}

extern "C" {

void dcd_build_default_opts(dcd_build_opts* o) {
  if (!o) return;
  memset(o, 0, sizeof *o);
  o->device = 0;
  o->min_word_length = 3;
  o->max_word_length = 32;
  o->encoding = DCD_ENC_LATIN1;
  o->coord_base = 0;
  o->vocab_capacity_log2 = 0;
}

int32_t dcd_build_create(const dcd_build_opts* opts, dcd_build** out) {
  if (!out) DCD_FAIL(DCD_E_BADARG, "dcd_build_create: out is NULL");
  dcd_build_opts o;
  if (opts) o = *opts; else dcd_build_default_opts(&o);
  if (o.encoding != DCD_ENC_LATIN1 && o.encoding != DCD_ENC_UTF16LE)
    DCD_FAIL(DCD_E_UNSUPPORTED, "unsupported text encoding %d (supported: LATIN1, UTF16LE)", o.encoding);
  if (o.min_word_length < 1 || o.max_word_length > 32 || o.min_word_length > o.max_word_length)
    DCD_FAIL(DCD_E_BADARG, "word length limits must satisfy 1 <= min <= max <= 32");
  // (`flags` used to be a reserved word: a caller that never initialised the struct with dcd_build_default_opts must not end
  // up with a shard-only build whose image / save / from_build fail much later)
  if (o.flags & ~DCD_BUILD_SHARD_ONLY) DCD_FAIL(DCD_E_BADARG, "dcd_build_create: unknown flags 0x%x (initialise dcd_build_opts with dcd_build_default_opts)", o.flags);
  DCD_TRY(ensure_device(o.device));
  dcd_build* h = new dcd_build();
  h->opts = o;
  h->device = o.device;
  memset(&h->stats, 0, sizeof h->stats);
  if (stream_acquire(o.device, &h->st) != DCD_OK) { delete h; return DCD_E_HIP; }
  h->page_off.push_back(0);
  h->coord_cursor = o.coord_base;
  *out = h;
  return DCD_OK;
}

void dcd_build_free(dcd_build* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  hipStream_t st = h->st;
  StreamScope ss_(st);
  if (st) (void)hipStreamSynchronize(st);
  const int dev = h->device;
  delete h;                       // (buffers go back to the pool before the stream is parked)
  stream_release(dev, st);
}

int32_t dcd_build_add_doc(dcd_build* h, const uint16_t* source, size_t source_len, const uint16_t* name, size_t name_len) {
  if (!h || (!source && source_len) || (!name && name_len)) DCD_FAIL(DCD_E_BADARG, "dcd_build_add_doc: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  if (source_len == 0) DCD_FAIL(DCD_E_BADARG, "sourceid must not be empty (Build.cs:350)");
  if (h->pages.doc_per_page) DCD_FAIL(DCD_E_STATE, "cannot mix explicit documents with docs_per_page pages");
  DocMeta d;
  d.name = mk16(source, source_len) + u":" + mk16(name, name_len);
  d.first_page = h->pages.end.size();
  h->pages.docs.push_back(d);
  return DCD_OK;
}

static int32_t add_explicit(dcd_build* h, const u16* code, size_t len, u64 coord, bool is_word);

// Header page (id "0"), Build.cs:485-524: one "name=value" per line; every \w-run of the value
// yields Add("&" + name, (uint)(coord + dc - 1)) and AddWord(word.ToLower(), coord + dc); the
// coordinate advances by line.Length + 1 per line.  A line with a long-enough name but no '='
// throws in the reference (fields[1]) and aborts the page WITHOUT EndPage; reproduced.
static int32_t host_header_page(dcd_build* h, const u16* text, size_t n, bool* ended) {
  const u8* cls; const u16* lower;
  host_char_tables16(&cls, &lower);
  u64 coord = h->coord_cursor;
  size_t p = 0;
  *ended = true;
  while (p < n) {
    size_t e = p;
    while (e < n && text[e] != u'\n' && text[e] != u'\r') e++;
    ustr line(e - p, u' ');
    for (size_t i = p; i < e; i++) line[i - p] = (char16_t)lower[text[i]];
    p = e;
    if (p < n) { if (text[p] == u'\r' && p + 1 < n && text[p + 1] == u'\n') p += 2; else p += 1; }
    size_t q = line.find(u'=');
    size_t f0 = q == ustr::npos ? line.size() : q;
    if ((int)f0 >= h->opts.min_word_length) {
      if (q == ustr::npos) { *ended = false; break; }
      size_t q2 = line.find(u'=', q + 1);
      ustr v = line.substr(q + 1, q2 == ustr::npos ? ustr::npos : q2 - q - 1);
      ustr marker = u"&" + line.substr(0, f0);
      u64 dc = f0 + 1;
      size_t i = 0;
      while (i < v.size()) {
        size_t j = i;
        bool w = (cls[(u16)v[i]] & 4) != 0;
        while (j < v.size() && ((cls[(u16)v[j]] & 4) != 0) == w) j++;
        if (w) {
          DCD_TRY(add_explicit(h, (const u16*)marker.data(), marker.size(), (u64)(u32)(coord + dc - 1), false));   // (uint) cast, Build.cs:509
          DCD_TRY(add_explicit(h, (const u16*)v.data() + i, j - i, coord + dc, true));                      // AddWord
        }
        dc += j - i;
        i = j;
      }
    }
    coord += line.size() + 1;
  }
  h->coord_cursor = coord;
  return DCD_OK;
}

// Book-keeping + staging of one batch of pages.  Body pages are copied to the device text buffer
// (contiguous ranges in one hipMemcpy each); header pages are tokenised here.
static int32_t add_pages_impl(dcd_build* h, const void* text, u64 n_chars, const uint64_t* page_off, size_t n_pages,
                              const uint16_t* page_ids, const uint64_t* ids_off, int32_t docs_per_page, uint64_t first_doc,
                              const uint16_t* source, size_t source_len, bool device_text) {
  if (h->mid) DCD_FAIL(DCD_E_STATE, "pages cannot be added after dcd_build_words");
  StreamScope ss_(h->st);
  if (docs_per_page) {
    if (!h->pages.docs.empty() || (h->pages.n_pages() && !h->pages.doc_per_page))
      DCD_FAIL(DCD_E_STATE, "cannot mix docs_per_page with explicit documents");
    if (h->pages.end.empty()) {
      h->pages.doc_per_page = true;
      h->pages.first_doc = first_doc;
      h->pages.source = source_len ? mk16(source, source_len) : ustr(u"syn");
    } else if (first_doc != h->pages.first_doc + h->pages.end.size()) {
      DCD_FAIL(DCD_E_BADARG, "docs_per_page batches must continue the document numbering");
    }
  } else if (h->pages.docs.empty()) {
    DCD_FAIL(DCD_E_STATE, "dcd_build_add_doc must precede pages (Build.cs:345-347)");
  }
  if (page_off[0] != 0 || page_off[n_pages] != n_chars) DCD_FAIL(DCD_E_BADARG, "page_off must start at 0 and end at n_chars");
  if (h->bulk_off) DCD_FAIL(DCD_E_STATE, "dcd_build_add_pages_dev must be the only text batch of its build");
  const bool uniform = ids_off == nullptr;
  if (uniform) {
    if (!h->pages.ids.empty()) DCD_FAIL(DCD_E_STATE, "cannot mix uniform and explicit page ids");
    h->pages.uniform_id = u"1";
  } else if (h->pages.ids.empty() && !h->pages.end.empty()) {
    DCD_FAIL(DCD_E_STATE, "cannot mix uniform and explicit page ids");
  }
  const int cs = h->csize();
  // pending contiguous body range of the caller's buffer: [r0, r1)
  u64 r0 = 0, r1 = 0;
  auto flush = [&]() -> int32_t {
    if (r1 == r0 || device_text) { r0 = r1; return DCD_OK; }
    u64 len = r1 - r0;
    u64 need = (h->n_chars + len) * cs + 64;
    if (need > h->text_cap) {
      u64 ncap = std::max<u64>(need, h->text_cap * 2);
      DevBuf<u8> nb;
      DCD_TRY(nb.alloc(h->device, ncap));
      if (h->n_chars) DCD_HIP(hipMemcpyAsync(nb.p, h->text_own.p, h->n_chars * cs, hipMemcpyDeviceToDevice, h->st));
      DCD_HIP(hipStreamSynchronize(h->st));
      h->text_own = std::move(nb);
      h->text_cap = ncap;
    }
    DCD_HIP(hipMemcpyAsync(h->text_own.p + h->n_chars * cs, (const u8*)text + r0 * cs, len * cs, hipMemcpyHostToDevice, h->st));
    h->n_chars += len;
    h->text_dev = h->text_own.p;
    r0 = r1;
    return DCD_OK;
  };
  u64 staged = h->n_chars;   // text index the next body character will get
  h->page_off.reserve(h->page_off.size() + n_pages);
  h->pages.end.reserve(h->pages.end.size() + n_pages);
  if (!uniform) h->pages.ids.reserve(h->pages.ids.size() + n_pages);
  if (!docs_per_page) h->pages.page_doc.reserve(h->pages.page_doc.size() + n_pages);
  if (uniform && docs_per_page && device_text && h->page_off.size() == 1 && h->pages.end.empty() && staged == 0 && !h->bulk_off) {
    // Round 4 -- the bulk case as the FIRST batch of its build: ONE pass over the caller's offsets (copy into pinned memory + the
    // ascending check); no per-page vectors at all.  (Round 3 filled two 8 MB vectors per million pages, uploaded both and copied
    // the page table into the index: 3.4 ms of host time around 3.9 ms of device time.)
    void* pin = pinned_stage_acquire((n_pages + 1) * sizeof(u64));
    if (!pin) DCD_FAIL(DCD_E_NOMEM, "hipHostMalloc failed");
    std::shared_ptr<u64> sp(static_cast<u64*>(pin), [](u64* q) { pinned_stage_release(q); });
    u64* dst = sp.get();
    DCD_TRY(ensure_device(h->device));
    DCD_TRY(h->d_bulk_off.alloc(h->device, n_pages + 1));
    u64 bad = 0, prev = page_off[0];
    dst[0] = prev;
    // (copy + check in pieces of 128 K offsets; every piece goes to the device as soon as it is in pinned memory, so that the
    // upload of a piece runs while the host copies the next one)
    constexpr size_t PIECE = 128 * 1024;
    for (size_t a = 0; a <= n_pages; a += PIECE) {
      const size_t b = std::min(n_pages + 1, a + PIECE);
      for (size_t p = std::max<size_t>(a, 1); p < b; p++) { const u64 v = page_off[p]; bad |= (u64)(v <= prev); dst[p] = v; prev = v; }
      DCD_HIP(hipMemcpyAsync(h->d_bulk_off.p + a, dst + a, (b - a) * sizeof(u64), hipMemcpyHostToDevice, h->st));
    }
    if (bad) { (void)hipStreamSynchronize(h->st); h->d_bulk_off.reset(); DCD_FAIL(DCD_E_BADARG, "page_off must be strictly ascending (docs_per_page requires non-empty pages)"); }
    const u64 c0 = h->coord_cursor;
    h->bulk_off = sp;
    h->bulk_n = n_pages;
    h->bulk_c0 = c0;
    const size_t np = n_pages;
    h->pages.end.set_lazy(np, [sp, c0, np](std::vector<u64>& v) { v.resize(np); const u64* o = sp.get(); for (size_t p = 0; p < np; p++) v[p] = c0 + o[p + 1]; });
    h->coord_cursor = c0 + n_chars;
    h->last_is_explicit = false;
    return DCD_OK;
  }
  if (uniform && docs_per_page && device_text && !h->bulk_off) {
    // the bulk case (one document per page, id "1", text already on the device): page bookkeeping only -- two appends per page
    // in one tight loop (a million pages took 14 ms through the general loop below; the text itself is not touched)
    const size_t e0 = h->pages.end.size();
    h->page_off.resize(h->page_off.size() + n_pages);
    h->pages.end.resize(e0 + n_pages);
    u64* po = h->page_off.data() + (h->page_off.size() - n_pages);
    u64* pe = h->pages.end.data() + e0;
    const u64 c0 = h->coord_cursor, s0 = staged;
    u64 bad = 0;
    for (size_t p = 0; p < n_pages; p++) {
      bad |= (u64)(page_off[p + 1] <= page_off[p]);
      po[p] = s0 + page_off[p + 1];
      pe[p] = c0 + page_off[p + 1];
    }
    if (bad) {
      h->page_off.resize(h->page_off.size() - n_pages);
      h->pages.end.resize(e0);
      DCD_FAIL(DCD_E_BADARG, "page_off must be strictly ascending (docs_per_page requires non-empty pages)");
    }
    h->coord_cursor = c0 + n_chars;
    h->last_is_explicit = false;
    return DCD_OK;
  }
  for (size_t p = 0; p < n_pages; p++) {
    if (page_off[p + 1] < page_off[p]) DCD_FAIL(DCD_E_BADARG, "page_off must be ascending");
    const u64 len = page_off[p + 1] - page_off[p];
    if (len == 0) {                                   // empty pages are skipped entirely (Build.cs:483)
      if (docs_per_page) DCD_FAIL(DCD_E_BADARG, "docs_per_page requires non-empty pages");
      if (r1 == page_off[p]) r1 = page_off[p + 1];
      continue;
    }
    ustr id;
    if (!uniform) id = mk16(page_ids + ids_off[p], (size_t)(ids_off[p + 1] - ids_off[p]));
    if (!uniform && id == u"0") {
      if (device_text) DCD_FAIL(DCD_E_UNSUPPORTED, "header pages need host text (dcd_build_add_pages)");
      if (r1 != page_off[p]) { DCD_TRY(flush()); }
      DCD_TRY(flush());
      r0 = r1 = page_off[p + 1];
      std::vector<u16> wide(len);
      if (cs == 1) for (u64 i = 0; i < len; i++) wide[i] = ((const u8*)text)[page_off[p] + i];
      else memcpy(wide.data(), (const u16*)text + page_off[p], len * 2);
      bool ended = true;
      DCD_TRY(host_header_page(h, wide.data(), len, &ended));
      if (ended) {
        h->pages.ids.push_back(id);
        h->pages.end.push_back(h->coord_cursor);
        h->pages.page_doc.push_back((u32)(h->pages.docs.size() - 1));
      }
      // the coordinate jumped without text: open a new run for the body characters that follow
      u64 delta = (h->coord_cursor - h->opts.coord_base) - staged;
      if (delta >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "coordinate window of the segment exceeded");
      if (h->run_off.empty()) { h->run_off.push_back(0); h->run_delta.push_back(0); }
      if (h->run_off.back() == (u32)staged) h->run_delta.back() = (u32)delta;
      else { h->run_off.push_back((u32)staged); h->run_delta.push_back((u32)delta); }
      continue;
    }
    if (r1 != page_off[p]) { DCD_TRY(flush()); r0 = page_off[p]; }
    r1 = page_off[p + 1];
    staged += len;
    h->coord_cursor += len;                            // coord += c.Length (Build.cs:531)
    if (!uniform) h->pages.ids.push_back(id);
    h->page_off.push_back(staged);
    h->pages.end.push_back(h->coord_cursor);           // EndPage(page.id, coord) (Build.cs:532)
    if (!docs_per_page) h->pages.page_doc.push_back((u32)(h->pages.docs.size() - 1));
    h->last_is_explicit = false;
  }
  DCD_TRY(flush());
  if (!device_text) DCD_HIP(hipStreamSynchronize(h->st));   // the caller may reuse `text` as soon as we return
  if (staged - (h->n_chars) != 0 && !device_text) DCD_FAIL(DCD_E_BADARG, "internal staging mismatch");
  return DCD_OK;
}

int32_t dcd_build_add_pages(dcd_build* h, const void* text, uint64_t n_chars, const uint64_t* page_off, size_t n_pages,
                            const uint16_t* page_ids, const uint64_t* ids_off, int32_t docs_per_page, uint64_t first_doc,
                            const uint16_t* source, size_t source_len) {
  if (!h || (!text && n_chars) || !page_off) DCD_FAIL(DCD_E_BADARG, "dcd_build_add_pages: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  if (h->text_borrowed) DCD_FAIL(DCD_E_STATE, "cannot append host text after dcd_build_add_pages_dev");
  if (ids_off && !page_ids) DCD_FAIL(DCD_E_BADARG, "ids_off without page_ids");
  if (h->n_chars + n_chars >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "one build segment holds < 2^32 characters; shard larger corpora");
  DCD_TRY(ensure_device(h->device));
  return add_pages_impl(h, text, n_chars, page_off, n_pages, page_ids, ids_off, docs_per_page, first_doc, source, source_len, false);
}

int32_t dcd_build_add_pages_dev(dcd_build* h, const void* text_dev, uint64_t n_chars, const uint64_t* page_off_host, size_t n_pages,
                                int32_t docs_per_page, uint64_t first_doc, const uint16_t* source, size_t source_len) {
  if (!h || !text_dev || !page_off_host) DCD_FAIL(DCD_E_BADARG, "dcd_build_add_pages_dev: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  if (h->n_chars) DCD_FAIL(DCD_E_STATE, "dcd_build_add_pages_dev must be the only text batch of its build");
  if (((uintptr_t)text_dev & 15) != 0) DCD_FAIL(DCD_E_BADARG, "device text must be 16-byte aligned");
  if (n_chars >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "one build segment holds < 2^32 characters; shard larger corpora");
  DCD_TRY(add_pages_impl(h, nullptr, n_chars, page_off_host, n_pages, nullptr, nullptr, docs_per_page, first_doc, source, source_len, true));
  h->text_dev = text_dev;
  h->text_borrowed = true;
  h->n_chars = n_chars;
  return DCD_OK;
}

int32_t dcd_build_add_uniform_pages_dev(dcd_build* h, const void* text_dev, uint64_t n_pages, uint32_t page_chars, uint64_t first_doc,
                                        const uint16_t* source, size_t source_len) {
  if (!h || !text_dev || (!source && source_len)) DCD_FAIL(DCD_E_BADARG, "dcd_build_add_uniform_pages_dev: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  if (h->n_chars || !h->pages.docs.empty() || h->pages.n_pages()) DCD_FAIL(DCD_E_STATE, "uniform pages must be the only text batch of their build");
  if (((uintptr_t)text_dev & 15) != 0) DCD_FAIL(DCD_E_BADARG, "device text must be 16-byte aligned");
  if (page_chars == 0 || n_pages == 0) DCD_FAIL(DCD_E_BADARG, "n_pages and page_chars must be positive");
  u64 n_chars = n_pages * (u64)page_chars;
  if (n_chars >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "one build segment holds < 2^32 characters; shard larger corpora");
  PageTable& pt = h->pages;
  pt.uniform = true; pt.uni_n = n_pages; pt.uni_chars = page_chars; pt.uni_base = h->opts.coord_base;
  pt.doc_per_page = true; pt.first_doc = first_doc; pt.uniform_id = u"1";
  pt.source = source_len ? mk16(source, source_len) : ustr(u"syn");
  h->text_dev = text_dev;
  h->text_borrowed = true;
  h->n_chars = n_chars;
  h->coord_cursor = h->opts.coord_base + n_chars;
  h->last_is_explicit = false;
  return DCD_OK;
}

int32_t dcd_build_add(dcd_build* h, const uint16_t* code, size_t len, uint64_t coord) {
  if (!h || (!code && len)) DCD_FAIL(DCD_E_BADARG, "dcd_build_add: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  return add_explicit(h, code, len, coord, false);
}

int32_t dcd_build_add_word(dcd_build* h, const uint16_t* word, size_t len, uint64_t coord) {
  if (!h || (!word && len)) DCD_FAIL(DCD_E_BADARG, "dcd_build_add_word: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  return add_explicit(h, word, len, coord, true);
}

int32_t dcd_build_end_page(dcd_build* h, const uint16_t* id, size_t len) {
  if (!h || (!id && len)) DCD_FAIL(DCD_E_BADARG, "dcd_build_end_page: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  if (h->pages.docs.empty()) DCD_FAIL(DCD_E_STATE, "dcd_build_add_doc must precede pages (Build.cs:345-347)");
  if (h->n_chars) DCD_FAIL(DCD_E_UNSUPPORTED, "EndPage(id) after device-tokenised text is not supported; pages end where their text ends");
  // EndPage(id) -> EndPage(id, maxCoord): the page ends at the last added coordinate (Build.cs:359-367)
  if (!h->pages.uniform_id.empty() && h->pages.ids.empty() && !h->pages.end.empty()) DCD_FAIL(DCD_E_STATE, "cannot mix uniform and explicit page ids");
  h->pages.ids.push_back(mk16(id, len));
  h->pages.end.push_back(h->last_explicit_coord);
  h->pages.page_doc.push_back((u32)(h->pages.docs.size() - 1));
  return DCD_OK;
}

}  // extern "C"

// ================================================================================================
// finish: the device pipeline
// ================================================================================================
// State that lives from the tokenise/vocabulary phase to the sort/pack phase.  Normally both phases run
// back to back inside dcd_build_finish; dcd_build_words stops after the first one so that the host can
// map every distinct word to its index codes (GetWordCodes, Build.cs:175-247) before the second.
struct BuildMid {
  bool codes_mode = false;          // phase 1 ran for dcd_build_words: explicit CODES were kept on the host
  u64 N = 0, n_text_tok = 0, n_xdev = 0, V = 0;
  u32 cap_log2 = 0, last_text_coord = 0;
  DevBuf<uint2> tupA, tupB;
  // tile-strided tuples (no counting pass): tupA holds tile t at [t * stride, t * stride + tile_cnt[t])
  bool strided = false;
  u32 stride = 0;
  u64 n_tiles = 0;
  // plain text builds: the tuples are sorted by DENSE id (position among the used table slots) and the vocabulary's
  // ordinal sort runs on a second stream beside the tuple sort; term order is restored when the units are copied out
  bool id_keys = false;
  u32 n_hot = 0;                    // hot / cold split: dense ids 0 .. n_hot-1 are the hot terms (0 = no split)
  bool early_ids = false;           // the tuples carry ids the table handed out while it was filled (id_cnt: per shard)
  IdCounts id_cnt{};
  u32 id_hot = 0;                   // ... the ids below id_hot belong to the hot terms (k_hot_select), the counters number the others from there
  DevBuf<u32> tile_cnt;
  DevBuf<u64> tile_off;
  DevBuf<u8> d_pool_raw;            // explicit strings on the device (CharT units)
  u64 n_pool = 0;
  DevBuf<u64> slots;
  u32 slot_stride = 1;              // u64 units per slot: 1, or VocPool::W when the string pool is interleaved (VocabTable)
  DevBuf<u32> d_flags, slot_excl, scan_scr32, d_hist;
  DevBuf<u32> d_histA;              // first radix pass's histogram, from the emit kernel's per-group rows (early ids + hot set); see k_tok_emit
  DevBuf<u16> d_hist_rows;          // [groups][256]
  DevBuf<u32> d_hist1;              // first cold pass's histogram, counted by pass A's scatter kernel (RsStrided::next_hist)
  bool histA_ready = false;
  VocabRanks vr;
  VocabPrep vprep;                  // id-keyed builds: first half of the vocabulary finalisation, run at the end of phase 1
  bool vprep_done = false;
  u32 voc_maxlen = 0, voc_nsym = 0;
  DevBuf<u64> d_page_end;
  DevBuf<u16> word_chars;           // the distinct words in ordinal order
  DevBuf<u32> word_str_off;
  u32 total_chars = 0;
  std::vector<u16> h_word_chars;    // host copies (codes mode)
  std::vector<u32> h_word_str_off;
  // codes handed over by dcd_build_set_codes
  bool have_codes = false;
  std::vector<u32> code_off;        // [V + 1]
  std::vector<ustr> code_str;       // one string per entry
  float ms_tok = 0, ms_voc = 0, ms_total = 0;
  float ms_kernel[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  u32 n_launch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

namespace {
// DOCODO_TRACE=1: host-side wall-clock marks of the build phases on stderr (where the non-device time goes)
struct HostTrace {
  bool on;
  std::vector<std::pair<const char*, double>> m;
  HostTrace() : on(getenv("DOCODO_TRACE") != nullptr) {}
  static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void mark(const char* l) { if (on) m.emplace_back(l, now()); }
  ~HostTrace() {
    if (!on || m.empty()) return;
    std::string o = "[docodo trace]";
    for (size_t i = 1; i < m.size(); i++) { char b[96]; snprintf(b, sizeof b, " %s %.3f", m[i].first, m[i].second - m[i - 1].second); o += b; }
    fprintf(stderr, "%s | total %.3f ms\n", o.c_str(), m.back().second - m.front().second);
  }
};
struct Probes {
  struct P { int id; hipEvent_t a, b; };
  std::vector<P> v;
  hipStream_t st;
  int dev;
  explicit Probes(hipStream_t s, int device) : st(s), dev(device) {}
  bool open_ = false;
  void begin(int id) {
    open_ = event_probe_on(id);
    if (!open_) return;
    P p{id, event_acquire(dev), event_acquire(dev)}; if (p.a) (void)hipEventRecord(p.a, st); v.push_back(p);
  }
  void end() { if (open_ && v.back().b) (void)hipEventRecord(v.back().b, st); open_ = false; }
  void collect(float* ms, u32* n) {   // after the stream was synchronised
    for (auto& p : v) {
      float t = 0;
      if (p.a && p.b && hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) { ms[p.id] += t; n[p.id]++; }
      event_release(dev, p.a); event_release(dev, p.b);
    }
    v.clear();
  }
  ~Probes() { for (auto& p : v) { event_release(dev, p.a); event_release(dev, p.b); } }
};
}  // namespace

// Vocabulary: compact, sort the word strings ordinally, materialise them (ordinal order).  Runs on stream `vs` (the build's
// own stream, or a second one beside the tuple sort) and returns with `vs` drained -- or, for id-keyed builds, with the work
// queued (see below).
template <typename CharT>
static int32_t build_vocab(dcd_build* h, BuildMid& M, hipStream_t vs, const TextView<CharT>& tv, u32* d_hist, u32* scan_scr32) {
  const int dev = h->device;
  const u64 cap = 1ull << M.cap_log2, V = M.V;
  // ONE host round trip inside (the longest key and the alphabet size decide the radix passes); the key pool is sized for
  // V keys of the longest length, so that the total needs no round trip of its own
  // (id-keyed builds ran the first half and fetched both parameters at the end of phase 1: nothing in here blocks then,
  // and the function returns with the work queued, not done)
  u32 maxlen = M.voc_maxlen;
  const bool queued_only = M.vprep_done;
  if (queued_only) DCD_TRY(vocab_sort<CharT>(dev, vs, cap, V, tv, M.vprep, M.voc_maxlen, M.voc_nsym, d_hist, scan_scr32, &M.vr));
  else DCD_TRY(vocab_rank<CharT>(dev, vs, M.slots.p, cap, M.slot_excl.p, V, tv, d_hist, scan_scr32, M.d_flags.p + 1, &M.vr, &maxlen, M.slot_stride));
  DCD_TRY(M.word_str_off.alloc(dev, V + 1));
  DCD_TRY(M.word_chars.alloc(dev, V * (u64)maxlen + 1));
  u32 total_chars = 0;
  if (V) {
    u32* tot = scan_scr32 + scan_nblocks(V);
    scan_exclusive_f<TermLenF, u32>(TermLenF{M.vr.rk_word.p}, M.word_str_off.p, V, scan_scr32, vs);
    hipLaunchKernelGGL(k_copy_u32, dim3(1), dim3(1), 0, vs, M.word_str_off.p + V, (const u32*)tot);
    hipLaunchKernelGGL((k_term_strings<CharT>), dim3((u32)div_up(V, 256)), dim3(256), 0, vs, M.vr.rk_word.p, V, M.word_str_off.p, tv,
                       M.word_chars.p);
    if (!queued_only) DCD_HIP(hipMemcpyAsync(&total_chars, tot, 4, hipMemcpyDeviceToHost, vs));
  } else {
    DCD_HIP(hipMemsetAsync(M.word_str_off.p, 0, 4, vs));
  }
  if (!queued_only) {
    DCD_HIP(hipStreamSynchronize(vs));
    M.total_chars = total_chars;
  }
  return DCD_OK;
}

// ---- phase 1: tokenise -> exact vocabulary -> distinct words in ordinal order ----------------------------------
template <typename CharT>
static int32_t build_phase1(dcd_build* h, bool codes_mode) {
  const int dev = h->device;
  hipStream_t st = h->st;
  StreamScope ss_(st);
  DCD_TRY(ensure_device(dev));
  h->mid.reset(new BuildMid());
  BuildMid& M = *h->mid;
  M.codes_mode = codes_mode;
  HostTrace tr;
  tr.mark("begin");
  EvTimer t_all, t_tok, t_voc;
  DCD_TRY(t_all.init(st, dev)); DCD_TRY(t_tok.init(st, dev)); DCD_TRY(t_voc.init(st, dev));
  Probes probes(st, dev);

  const u64 n_chars = h->n_chars;
  const bool uniform = h->pages.uniform;
  const u64 n_pages_dev = uniform ? h->pages.uni_n : (h->bulk_off ? h->bulk_n : h->page_off.size() - 1);
  const u64 n_tiles = div_up(n_chars, TK_TILE);
  const CharT* text = (const CharT*)h->text_dev;

  CharTables16 tab{nullptr, nullptr};
  if (sizeof(CharT) == 2) DCD_TRY(get_char_tables16(dev, &tab));

  tr.mark("timers");
  // ---- uploads (not part of the timed device pipeline: the text is already resident) ----------
  DevBuf<u64> d_page_off;
  if (h->bulk_off && h->d_bulk_off.p) d_page_off = std::move(h->d_bulk_off);      // (uploaded by dcd_build_add_pages_dev already)
  // (host -> device through pinned staging, see pinned_stage_acquire: a pageable upload of these megabytes made the runtime lock
  //  the vectors' pages, and freeing the vectors with the handle then stalled the next build by 20 ms)
  struct Staged { void* p[2] = {nullptr, nullptr}; ~Staged() { pinned_stage_release(p[0]); pinned_stage_release(p[1]); } } staged;
  auto upload = [&](int slot, void* dst, const void* src, size_t bytes) -> int32_t {
    staged.p[slot] = pinned_stage_acquire(bytes);
    if (!staged.p[slot]) DCD_FAIL(DCD_E_NOMEM, "hipHostMalloc failed");
    memcpy(staged.p[slot], src, bytes);
    DCD_HIP(hipMemcpyAsync(dst, staged.p[slot], bytes, hipMemcpyHostToDevice, st));
    return DCD_OK;
  };
  if (!uniform) {
    if (!d_page_off.p) {
      DCD_TRY(d_page_off.alloc(dev, n_pages_dev + 1));
      if (h->bulk_off) DCD_HIP(hipMemcpyAsync(d_page_off.p, h->bulk_off.get(), (n_pages_dev + 1) * 8, hipMemcpyHostToDevice, st));   // (pinned already)
      else DCD_TRY(upload(0, d_page_off.p, h->page_off.data(), (n_pages_dev + 1) * 8));
    }
  }
  // explicit postings that enter the device vocabulary: all of them, or (codes mode) only the WORDS
  std::vector<u32> xo, xl, xc;
  for (const ExplicitPosting& x : h->xps)
    if (!codes_mode || x.is_word) { xo.push_back(x.off); xl.push_back(x.len); xc.push_back((u32)(x.coord - h->opts.coord_base)); }
  const u64 n_x = xo.size();
  M.n_xdev = n_x;
  DevBuf<u32> d_xoff, d_xlen, d_xcoord;
  if (n_x) {
    std::vector<CharT> pool((h->xpool.size() + 8) / 4 * 4, (CharT)0);   // zero padded: read as aligned 32-bit words
    for (size_t i = 0; i < h->xpool.size(); i++) pool[i] = (CharT)h->xpool[i];
    M.n_pool = pool.size();
    DCD_TRY(M.d_pool_raw.alloc(dev, pool.size() * sizeof(CharT)));
    DCD_TRY(d_xoff.alloc(dev, n_x)); DCD_TRY(d_xlen.alloc(dev, n_x)); DCD_TRY(d_xcoord.alloc(dev, n_x));
    DCD_HIP(hipMemcpyAsync(M.d_pool_raw.p, pool.data(), pool.size() * sizeof(CharT), hipMemcpyHostToDevice, st));
    DCD_HIP(hipMemcpyAsync(d_xoff.p, xo.data(), n_x * 4, hipMemcpyHostToDevice, st));
    DCD_HIP(hipMemcpyAsync(d_xlen.p, xl.data(), n_x * 4, hipMemcpyHostToDevice, st));
    DCD_HIP(hipMemcpyAsync(d_xcoord.p, xc.data(), n_x * 4, hipMemcpyHostToDevice, st));
    DCD_HIP(hipStreamSynchronize(st));
  }
  tr.mark("page-off");
  if (uniform) {
    DCD_TRY(M.d_page_end.alloc(dev, h->pages.uni_n));
    tr.mark("page-end-alloc");
    hipLaunchKernelGGL(k_uniform_page_end, dim3((u32)div_up(h->pages.uni_n, 256)), dim3(256), 0, st, M.d_page_end.p, h->pages.uni_n,
                       h->pages.uni_base, h->pages.uni_chars);
  } else if (h->bulk_off) {
    DCD_TRY(M.d_page_end.alloc(dev, h->bulk_n));
    hipLaunchKernelGGL(k_page_end_from_off, dim3((u32)div_up(h->bulk_n, 256)), dim3(256), 0, st, M.d_page_end.p, d_page_off.p, h->bulk_n, h->bulk_c0);
  } else if (!h->pages.end.empty()) {
    DCD_TRY(M.d_page_end.alloc(dev, h->pages.end.size()));
    DCD_TRY(upload(1, M.d_page_end.p, h->pages.end.data(), h->pages.end.size() * 8));
  }
  tr.mark("page-end");
  DCD_HIP(hipStreamSynchronize(st));

  tr.mark("setup");
  DCD_TRY(t_all.start());
  // ---- 1. tokenise: count pass -------------------------------------------------------------------
  t_tok.start_at_start_of(t_all);
  TokParams P;
  P.n_chars = n_chars;
  P.n_pages = n_pages_dev;
  // explicit page offsets: the bitmap of page starts the tokeniser reads (inside the timed region: it is device work the
  // uniform-pages entry point does not have -- a memset of n_chars / 8 bytes and one atomic per page)
  DevBuf<u32> d_page_bits;
  P.page_bits = nullptr;
  if (!uniform) {
    const u64 words = (n_chars + TK_SPAN + 128) / 32 + 4;
    DCD_TRY(d_page_bits.alloc(dev, words));
    DCD_HIP(hipMemsetAsync(d_page_bits.p, 0, words * 4, st));
    hipLaunchKernelGGL(k_page_bits, dim3((u32)div_up(n_pages_dev + 1, 256)), dim3(256), 0, st, d_page_off.p, n_pages_dev + 1, d_page_bits.p);
    P.page_bits = d_page_bits.p;
  }
  P.min_len = (u32)h->opts.min_word_length;
  P.max_len = (u32)h->opts.max_word_length;
  P.uniform_page_chars = uniform ? (u32)h->pages.uni_chars : 0u;
  DevBuf<u32> d_run_off, d_run_delta;
  P.run_off = nullptr; P.run_delta = nullptr; P.n_runs = 0;
  if (!h->run_off.empty()) {
    DCD_TRY(d_run_off.alloc(dev, h->run_off.size()));
    DCD_TRY(d_run_delta.alloc(dev, h->run_delta.size()));
    DCD_HIP(hipMemcpyAsync(d_run_off.p, h->run_off.data(), h->run_off.size() * 4, hipMemcpyHostToDevice, st));
    DCD_HIP(hipMemcpyAsync(d_run_delta.p, h->run_delta.data(), h->run_delta.size() * 4, hipMemcpyHostToDevice, st));
    DCD_HIP(hipStreamSynchronize(st));
    P.run_off = d_run_off.p; P.run_delta = d_run_delta.p; P.n_runs = (u32)h->run_off.size();
  }
  // Plain text builds (no explicit postings, no word codes) skip the counting pass: the emit kernel writes every tile's
  // tuples at a fixed stride and the first sort pass reads that layout in place (RsStrided, dcd_sort.h).  A tile of
  // TK_TILE characters holds at most TK_TILE / min_word_length + 1 kept tokens.  DOCODO_TOK_TWO_PASS=1 forces the
  // counting pass.
  static const bool env_two_pass = getenv("DOCODO_TOK_TWO_PASS") && atoi(getenv("DOCODO_TOK_TWO_PASS")) != 0;
  // (kept tokens may follow each other WITHOUT a separator -- a class change, "abc123" -- so they start min_len apart at least)
  const u32 stride_need = ((u32)(TK_TILE / (P.min_len ? P.min_len : 1u)) + 2u + 1u) & ~1u;     // even: tiles start 16-byte aligned
  // The unit of the strided tuple buffer (a "tile" of the first sort pass) is the GROUP of TK_WAVES wave tiles one workgroup
  // of the emit kernel handles: n_st of them, TK_WAVES * stride_need tuple slots each.
  const u64 n_st = div_up(n_tiles, TK_WAVES);
  const bool strided = !env_two_pass && !codes_mode && n_x == 0 && n_tiles != 0 && (u64)TK_WAVES * stride_need <= (u64)RS_STRIDE_MAX &&
                       rs_tile_plan(n_st).tpb <= (u32)RS_TPB_MAX;
  M.strided = strided; M.stride = strided ? (u32)TK_WAVES * stride_need : 0u; M.n_tiles = strided ? n_st : n_tiles;
  DevBuf<u64> scan_scr64;
  u64 n_text_tok = 0;
  if (n_tiles) {
    DCD_TRY(M.tile_cnt.alloc(dev, n_tiles));
    DCD_TRY(M.tile_off.alloc(dev, n_tiles));                         // (builds with early ids use neither: the pool keeps them)
    DCD_TRY(scan_scr64.alloc(dev, scan_scratch_elems(n_tiles)));
  }
  if (n_tiles && !strided) {
    probes.begin(DCD_K_TOK_COUNT);
    hipLaunchKernelGGL((k_tok_count<CharT>), dim3((u32)div_up(n_tiles, TK_WAVES)), dim3(TK_THREADS), 0, st, text, P, tab, M.tile_cnt.p,
                       n_tiles);
    probes.end();
    scan_exclusive<u32, u64>(M.tile_cnt.p, M.tile_off.p, n_tiles, scan_scr64.p, st);
    DCD_HIP(hipMemcpyAsync(&n_text_tok, scan_scr64.p + scan_nblocks(n_tiles), 8, hipMemcpyDeviceToHost, st));
    DCD_HIP(hipStreamSynchronize(st));
  }
  tr.mark("count");
  u64 N = n_text_tok + n_x;
  if (N >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "more than 2^32 postings in one build segment");

  // ---- 2. vocabulary table + emit ---------------------------------------------------------------
  if (strided) {
    DCD_TRY(M.tupA.alloc(dev, n_st * (u64)M.stride));
  } else {
    DCD_TRY(M.tupA.alloc(dev, N ? N : 1));
    DCD_TRY(M.tupB.alloc(dev, N ? N : 1));
  }
  u32 cap_log2 = h->opts.vocab_capacity_log2;
  if (cap_log2 == 0) {
    // first guess: one slot per 256 postings -- small enough to stay close to the caches (every probe of a
    // tail term is a random access, and the table is scanned a few times); grown x4 and the emit re-run if the load exceeds 1/2
    // (measured at C2 size: 2^20 slots against 2^22: tokenise the same, everything behind it ~0.07 ms faster).  Without a counting pass
    // the postings are estimated as one per 7 characters.
    const u64 n_est = strided ? n_chars / 7 : N;
    cap_log2 = log2_ceil((n_est ? n_est : 1) / DCD_CAPDIV + 1);
    if (cap_log2 < 12) cap_log2 = 12;
    if (cap_log2 > 24) cap_log2 = 24;
  }
  DCD_TRY(M.d_flags.alloc(dev, 4));   // [0] overflow, [1] maxlen, [2] max coord (explicit)
  TextView<CharT> tv;
  tv.text = text; tv.n_chars = n_chars; tv.pool = (const CharT*)M.d_pool_raw.p; tv.n_pool = M.n_pool; tv.tab = tab;
  u64 V = 0;
  // hot / cold split (plain text builds): sample counts per slot from every (mask + 1)-th tile, ~1000-2000 tiles in all
  // (~200 k tokens: the 255th term of a Zipf-1.1 vocabulary still shows up ~60 times; every count is an atomic, and those
  // of the top terms hit ONE address each -- 4096 sampled tiles slowed the emit kernel by 6 %)
  static const int env_hot = getenv("DOCODO_BUILD_HOT") ? atoi(getenv("DOCODO_BUILD_HOT")) : 255;   // hot terms at most; 0 = off
  static const bool env_rank_keys0 = getenv("DOCODO_BUILD_RANK_KEYS") && atoi(getenv("DOCODO_BUILD_RANK_KEYS")) != 0;
  // early ids (default for plain text builds): the table numbers its terms while it is filled and the tokeniser writes the
  // sort key itself -- no slot -> id table, no remap gathers in the first radix pass, no hot / cold selection.
  // DOCODO_BUILD_EARLY_IDS=0: ids from the table scan after the emit kernel (+ hot / cold split), as before.
  static const bool env_early = !(getenv("DOCODO_BUILD_EARLY_IDS") && atoi(getenv("DOCODO_BUILD_EARLY_IDS")) == 0);
  bool early_ids = strided && !env_rank_keys0 && env_early;
  const bool fast_end = strided && !env_rank_keys0;
  // Early ids: ONE buffer holds the small device-side state of the round, so that one fill zeroes it and one copy fetches
  // its results.  VOC_SHARDS id counters VOC_CTR_STRIDE words apart, then the results R[0 .. EI_WORDS): the counts again,
  // densely, + their sum V (k_vocab_compact_ids), the two parameters of the ordinal sort, maxCoord, the number of postings,
  // the number of hot terms, the overflow flag; then the character bitmap, the fix-up list (count + UNRES_CAP tiles) and the
  // count of the sampling launch's slot list (all of that zeroed), the list itself and the per-workgroup key-length maxima.
  DevBuf<u32> id_ctr;
  constexpr u32 ID_CTR_WORDS = VOC_SHARDS * VOC_CTR_STRIDE;
  constexpr u32 EI_V = VOC_SHARDS, EI_NSYM = EI_V + 1, EI_MAXLEN = EI_V + 2, EI_LASTC = EI_V + 3, EI_NTOK = EI_V + 4 /* u64 */, EI_OVF = EI_V + 6,
                EI_NHOT = EI_V + 7, EI_WORDS = EI_V + 8, EI_CBITS = EI_WORDS, EI_UNRES = EI_CBITS + 2052, EI_LIST = EI_UNRES + 1 + UNRES_CAP,
                EI_ZEROED = EI_LIST + 1;       // (then: the sampled slots, the per-workgroup key-length maxima)
  static_assert(EI_WORDS * 4 <= (64 - 24) * 8, "the results must fit the pinned words behind pin[24]");
  static_assert(EI_NTOK % 2 == 0 && EI_CBITS % 4 == 0, "alignment of the 64-bit count / the bitmap");
  // early ids: the sampling launch in front of the real one (k_hot_select): DOCODO_TOK_SAMPLE_GROUPS groups of TK_WAVES tiles,
  // evenly spaced over the text (0, or a text of fewer than four times as many groups: no sampling, no hot set)
  static const u32 env_sample_groups = getenv("DOCODO_TOK_SAMPLE_GROUPS") ? (u32)atoi(getenv("DOCODO_TOK_SAMPLE_GROUPS")) : 64u;
  const u32 n_sample = (env_hot > 0 && n_st >= 4ull * env_sample_groups) ? env_sample_groups : 0u;
  const u32 sample_step = n_sample ? (u32)(n_st / n_sample) : 0u;
  const u64 list_cap = (u64)n_sample * TK_WAVES * stride_need + 1;          // (every sampled token a new term, at worst)
  u32 sample_mask = 0;
  while (((u64)sample_mask + 1) * DCD_HOT_SAMPLE_TILES < n_st) sample_mask = sample_mask * 2 + 1;     // (groups of TK_WAVES tiles)
  DevBuf<u32> sample_cnt, hot_hist;           // [cap] / [HOT_BINS + 1] (theta behind the bins)
  DevBuf<u64> hot_excl, hot_scr;
  u32 n_hot = 0;
  u64* pin = pinned_words();
  if (!pin) DCD_FAIL(DCD_E_NOMEM, "hipHostMalloc failed");
  u64* rb_ntok = pin + 16;
  u64* rb_total = pin + 17;
  u32* rb_flags = (u32*)(pin + 18);     // 4 words
  u32* rb_nsym = (u32*)(pin + 20);
  u32* rb_ids = (u32*)(pin + 24);       // early ids: EI_WORDS words (the results of the round)
  *rb_ntok = 0; *rb_total = 0; *rb_nsym = 0;
  for (;;) {
    n_hot = 0;                          // (a round that ends without a hot set must not inherit the one of the round before)
    if (cap_log2 > 30) DCD_FAIL(DCD_E_OVERFLOW, "vocabulary does not fit a 2^30-slot table");
    const u64 cap = 1ull << cap_log2;
    static const bool no_pool = getenv("DOCODO_VOCAB_POOL") && atoi(getenv("DOCODO_VOCAB_POOL")) == 0;
    // The string pool is an accelerator, not a requirement: very large tables (> 2^26 slots: 2-4 GB) or a failed allocation
    // fall back to a plain slot array and to verifying every hit against the text.  With the pool the table is ONE array of
    // records {slot word, padding, string} (32 bytes for Latin-1, 64 for UTF-16): the slot word and the string of a term
    // share a cache line, so a token of a rare term -- the ones that miss the XCD's L2: a tenth of the tokens of a Zipf
    // corpus -- costs one line from the fabric instead of two.
    constexpr u32 PWW = VocPool<CharT>::W;
    M.slot_stride = 1;
    if (!no_pool && cap_log2 <= 26 && M.slots.alloc(dev, cap * PWW) == DCD_OK) M.slot_stride = PWW;
    else DCD_TRY(M.slots.alloc(dev, cap));
    DCD_HIP(hipMemsetAsync(M.slots.p, 0, cap * 8 * M.slot_stride, st));
    if (early_ids && M.slot_stride == 1) early_ids = false;       // (no record table: very large vocabulary or no memory)
    if (!early_ids) DCD_HIP(hipMemsetAsync(M.d_flags.p, 0, 16, st));      // (early ids keep their flags with the id counters)
    VocabTable vt;
    vt.slots = M.slots.p; vt.mask = (u32)(cap - 1); vt.overflow = M.d_flags.p;
    vt.pool = nullptr; vt.stride = M.slot_stride; vt.pstride = 0;
    if (M.slot_stride > 1) { vt.pool = reinterpret_cast<u32*>(M.slots.p) + PWW; vt.pstride = 2 * PWW; }
    const bool hot_try = strided && !env_rank_keys0 && env_hot > 0 && !early_ids;
    u32* unres = nullptr;
    if (early_ids) {
      DCD_TRY(id_ctr.alloc(dev, ID_CTR_WORDS + EI_ZEROED + list_cap + div_up(cap / 2 + 1, 256) + 1));
      DCD_HIP(hipMemsetAsync(id_ctr.p, 0, (size_t)(ID_CTR_WORDS + EI_ZEROED) * 4, st));
      vt.id_ctr = id_ctr.p;
      vt.overflow = id_ctr.p + ID_CTR_WORDS + EI_OVF;
      unres = id_ctr.p + ID_CTR_WORDS + EI_UNRES;
    }
    if (hot_try) {
      DCD_TRY(sample_cnt.alloc(dev, cap));
      DCD_TRY(hot_hist.alloc(dev, HOT_BINS + 1));
      DCD_HIP(hipMemsetAsync(sample_cnt.p, 0, cap * 4, st));
      DCD_HIP(hipMemsetAsync(hot_hist.p, 0, (HOT_BINS + 1) * 4, st));
    }
    if (n_tiles) {
      // A small SEED launch first: on an empty table every resident wave (7000+) finds the slots of the most frequent
      // terms empty and tries to insert them -- tens of thousands of compare-and-swaps on a handful of addresses that
      // execute one after the other (measured: the launch on an empty table takes 0.15 ms longer than on a populated
      // one).  256 tiles (~33 k tokens) insert the head of the distribution with 1/28 of the contention; the main launch
      // then finds those slots populated (a kernel boundary makes them visible to every XCD).
      static const u64 seed_tiles = getenv("DOCODO_TOK_SEED_TILES") ? strtoull(getenv("DOCODO_TOK_SEED_TILES"), nullptr, 10) : 256;
      const u64 n_seed = n_tiles > 2 * seed_tiles ? (seed_tiles & ~(u64)(TK_WAVES - 1)) : 0;   // small builds: one launch (whole groups)
      if (early_ids) {
        // Early ids: the SAMPLING launch takes the seed launch's place -- it fills the table with the head of the
        // distribution just as well -- and k_hot_select turns its counts into the hot set; the real launch then covers every
        // tile (the sampled groups are tokenised twice: 64 of 270 k).
        u32* R = id_ctr.p + ID_CTR_WORDS;
        if (n_sample) {
          VocabTable vs_ = vt;
          vs_.id_ctr = nullptr; vs_.new_slots = R + EI_LIST;
          hipLaunchKernelGGL((k_tok_emit<CharT, true>), dim3(n_sample), dim3(TK_THREADS), 0, st, text, P, tab, tv, vs_, M.tile_off.p, M.tupA.p, n_tiles,
                             M.stride, M.tile_cnt.p, (u64)0, (u32*)nullptr, 0u, (u32*)nullptr, sample_step);
          hipLaunchKernelGGL(k_hot_select, dim3(1), dim3(1024), 0, st, M.slots.p, M.slot_stride, (const u32*)(R + EI_LIST), (u32)std::min(env_hot, 255),
                             R + EI_NHOT, id_ctr.p);
        }
        vt.id_base = R + EI_NHOT;
      }
      if (early_ids && n_sample) {
        // the emit kernel also fills the histogram of the first radix pass (digit = min(id, n_hot)): zeroed per round
        static const bool env_emit_hist = !(getenv("DOCODO_EMIT_HIST") && atoi(getenv("DOCODO_EMIT_HIST")) == 0);
        const RsTilePlan tpa = rs_tile_plan(n_st);
        M.histA_ready = false;
        if (P.n_runs == 0 && env_emit_hist && env_hot > 0) {
          DCD_TRY(M.d_histA.alloc(dev, (u64)tpa.nblk << 8));
          DCD_TRY(M.d_hist_rows.alloc(dev, n_st * 256));
          DCD_HIP(hipMemsetAsync(M.d_histA.p, 0, ((u64)tpa.nblk << 8) * 4, st));
          M.histA_ready = true;
        }
        probes.begin(DCD_K_TOK_EMIT);
        if (P.n_runs == 0)      // (early ids imply strided output and no per-slot sample counts)
          hipLaunchKernelGGL((k_tok_emit<CharT, false, true>), dim3((u32)div_up(n_tiles, TK_WAVES)), dim3(TK_THREADS), 0, st, text, P, tab, tv, vt,
                             M.tile_off.p, M.tupA.p, n_tiles, M.stride, M.tile_cnt.p, (u64)0, M.histA_ready ? (u32*)M.d_hist_rows.p : (u32*)nullptr,
                             0u, unres, 0u);
        else
          hipLaunchKernelGGL((k_tok_emit<CharT>), dim3((u32)div_up(n_tiles, TK_WAVES)), dim3(TK_THREADS), 0, st, text, P, tab, tv, vt,
                             M.tile_off.p, M.tupA.p, n_tiles, M.stride, M.tile_cnt.p, (u64)0, sample_cnt.p, sample_mask, unres, 0u);
        probes.end();
      } else {
        if (n_seed) {   // (not a DCD_K_TOK_EMIT probe: per-launch averages are quoted for the main launch)
          hipLaunchKernelGGL((k_tok_emit<CharT>), dim3((u32)div_up(n_seed, TK_WAVES)), dim3(TK_THREADS), 0, st, text, P, tab, tv, vt,
                             M.tile_off.p, M.tupA.p, n_seed, M.stride, M.tile_cnt.p, (u64)0, sample_cnt.p, sample_mask, unres, 0u);
        }
        probes.begin(DCD_K_TOK_EMIT);
        if (n_tiles > n_seed)
          hipLaunchKernelGGL((k_tok_emit<CharT>), dim3((u32)div_up(n_tiles - n_seed, TK_WAVES)), dim3(TK_THREADS), 0, st, text, P, tab, tv, vt,
                             M.tile_off.p, M.tupA.p, n_tiles, M.stride, M.tile_cnt.p, n_seed, sample_cnt.p, sample_mask, unres, 0u);
        probes.end();
      }
#if defined(DCD_DIAG_TOK)
      if (DCD_DIAG_TOK == 6) {          // second launch: every term is in the table, no CAS, no insert storm
        DCD_HIP(hipStreamSynchronize(st));
        float msa[8] = {0}; u32 nla[8] = {0};
        probes.collect(msa, nla);
        fprintf(stderr, "DIAG_TOK mode 6: first launch %.3f ms\n", msa[DCD_K_TOK_EMIT]);
        probes.begin(DCD_K_TOK_EMIT);
        hipLaunchKernelGGL((k_tok_emit<CharT>), dim3((u32)div_up(n_tiles, TK_WAVES)), dim3(TK_THREADS), 0, st, text, P, tab, tv, vt,
                           M.tile_off.p, M.tupA.p, n_tiles, M.stride, M.tile_cnt.p, (u64)0, sample_cnt.p, sample_mask, unres, 0u);
        probes.end();
      }
      {
        DCD_HIP(hipStreamSynchronize(st));
        float ms8[8] = {0}; u32 nl8[8] = {0};
        probes.collect(ms8, nl8);
        fprintf(stderr, "DIAG_TOK mode %d: k_tok_emit %.3f ms\n", (int)DCD_DIAG_TOK, ms8[DCD_K_TOK_EMIT]);
        DCD_FAIL(DCD_E_STATE, "diagnostic build (DCD_DIAG_TOK): timing only");
      }
#endif
      u32* R = early_ids ? id_ctr.p + ID_CTR_WORDS : nullptr;
      if (early_ids) {
        // keys whose id was not visible when the token was looked up -> ids; the number of postings; maxCoord
        const RsTilePlan tpe = rs_tile_plan(n_st);
        hipLaunchKernelGGL(k_emit_epilogue, dim3(256), dim3(256), 0, st, M.tupA.p, (const u32*)M.tile_cnt.p, n_st, M.stride, (const u32*)unres,
                           (const u64*)M.slots.p, M.slot_stride, (u64*)(R + EI_NTOK), R + EI_LASTC,
                           M.histA_ready ? M.d_histA.p : (u32*)nullptr, tpe.tpb, tpe.nblk, (const u32*)(R + EI_NHOT));
        if (M.histA_ready)
          hipLaunchKernelGGL(k_hist_rows, dim3(tpe.nblk), dim3(256), 0, st, (const u16*)M.d_hist_rows.p, n_st, tpe.tpb, tpe.nblk, M.d_histA.p);
      } else if (strided) {
        // the tiles' dense offsets and the number of postings, now that the emit kernel has counted
        scan_exclusive<u32, u64>(M.tile_cnt.p, M.tile_off.p, n_st, scan_scr64.p, st);
        DCD_HIP(hipMemcpyAsync(rb_ntok, scan_scr64.p + scan_nblocks(n_st), 8, hipMemcpyDeviceToHost, st));
      }
    }
    if (n_x)
      hipLaunchKernelGGL((k_explicit<CharT>), dim3((u32)div_up(n_x, 256)), dim3(256), 0, st, d_xoff.p, d_xlen.p, d_xcoord.p,
                         n_x, tv, vt, M.tupA.p + n_text_tok);
    // compaction: dense ids for the used slots
    DCD_TRY(M.slot_excl.alloc(dev, cap));
    DCD_TRY(M.scan_scr32.alloc(dev, std::max<u64>(scan_scratch_elems(cap), scan_scratch_elems((u64)RS_MAX_BLOCKS << 10))));
    u32 v32 = 0;
    if (strided && !early_ids) {
      // maxCoord = coordinate of the last posting added (Build.cs:326): fetched with the counters below, ONE host round trip
      hipLaunchKernelGGL(k_last_strided_coord, dim3(1), dim3(1), 0, st, M.tupA.p, M.tile_cnt.p, n_st, M.stride, M.d_flags.p + 2);
    }
    const bool last_round = fast_end;      // id-keyed builds: phase 1 ends with this round trip (the stage timers stop here)
    auto stop_timers = [&]() -> int32_t {
      // (one event: the vocabulary timer is empty here -- id-keyed builds take its time from the side stream in phase 2)
      DCD_TRY(t_tok.stop()); t_voc.start_at_stop_of(t_tok); t_voc.stop_at_stop_of(t_tok); t_all.stop_at_stop_of(t_tok);
      return DCD_OK;
    };
    if (early_ids) {
      // the counters are the vocabulary: V = their sum; dense arrays for the ordinal sort straight from the ids; ONE copy
      // brings everything the host needs
      u32* R = id_ctr.p + ID_CTR_WORDS;
      DCD_TRY(vocab_prepare_ids<CharT>(dev, st, M.slots.p, cap, cap / 2 + 1, tv, &M.vprep, M.slot_stride, (const u32*)id_ctr.p, R, R + EI_CBITS,
                                       R + EI_ZEROED + list_cap, R + EI_NSYM, R + EI_MAXLEN, R + EI_NHOT));
      M.vprep_done = true;
      DCD_TRY(stop_timers());
      DCD_HIP(hipMemcpyAsync(rb_ids, R, EI_WORDS * 4, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipStreamSynchronize(st));
      for (u32 s = 0; s < VOC_SHARDS; s++) M.id_cnt.c[s] = rb_ids[s];
      v32 = rb_ids[EI_V];
      n_hot = rb_ids[EI_NHOT];
      *rb_nsym = rb_ids[EI_NSYM];
      rb_flags[0] = rb_ids[EI_OVF]; rb_flags[1] = rb_ids[EI_MAXLEN]; rb_flags[2] = rb_ids[EI_LASTC];
      *rb_ntok = *(const u64*)(rb_ids + EI_NTOK);
    } else if (hot_try) {
      // hot terms first: theta from the histogram of the sample counts, ONE 64-bit scan counts used and hot slots, the
      // dense-id table (hot terms 0 .. n_hot-1, then the others in slot order) takes the place of the plain "used" scan
      u32* theta = hot_hist.p + HOT_BINS;
      DCD_TRY(hot_excl.alloc(dev, cap));
      DCD_TRY(hot_scr.alloc(dev, scan_scratch_elems(cap)));
      hipLaunchKernelGGL(k_hot_hist, dim3(512), dim3(256), 0, st, (const u32*)sample_cnt.p, cap, hot_hist.p);
      hipLaunchKernelGGL(k_hot_pick, dim3(1), dim3(1024), 0, st, (const u32*)hot_hist.p, (u32)env_hot, theta);
      scan_exclusive_f<HotUsedF, u64>(HotUsedF{M.slots.p, sample_cnt.p, theta, M.slot_stride}, hot_excl.p, cap, hot_scr.p, st);
      const u64* total = hot_scr.p + scan_nblocks(cap);
      hipLaunchKernelGGL(k_dense_table, dim3((u32)div_up(cap, 256)), dim3(256), 0, st, (const u64*)M.slots.p, (const u32*)sample_cnt.p,
                         (const u32*)theta, (const u64*)hot_excl.p, total, cap, M.slot_excl.p, M.slot_stride);
      if (last_round) {
        // first half of the vocabulary finalisation here, so that its two parameters come back with this round trip and
        // phase 2 can queue the ordinal sort without ever blocking (V itself is still on the device: bound = cap / 2 + 1)
        DCD_TRY(vocab_prepare<CharT>(dev, st, M.slots.p, cap, M.slot_excl.p, cap / 2 + 1, tv, M.d_flags.p + 1, &M.vprep, (const u32*)total,
                                     M.slot_stride));
        DCD_HIP(hipMemcpyAsync(rb_nsym, M.vprep.d_nsym, 4, hipMemcpyDeviceToHost, st));
        M.vprep_done = true;
        DCD_TRY(stop_timers());
      }
      DCD_HIP(hipMemcpyAsync(rb_flags, M.d_flags.p, 16, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipMemcpyAsync(rb_total, total, 8, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipStreamSynchronize(st));
      const u64 tot64 = *rb_total;
      v32 = (u32)tot64;
      n_hot = (u32)(tot64 >> 32);
    } else {
      scan_exclusive_f<SlotUsed, u32>(SlotUsed{M.slots.p, M.slot_stride}, M.slot_excl.p, cap, M.scan_scr32.p, st);
      if (last_round) {
        DCD_TRY(vocab_prepare<CharT>(dev, st, M.slots.p, cap, M.slot_excl.p, cap / 2 + 1, tv, M.d_flags.p + 1, &M.vprep,
                                     (const u32*)(M.scan_scr32.p + scan_nblocks(cap)), M.slot_stride));
        DCD_HIP(hipMemcpyAsync(rb_nsym, M.vprep.d_nsym, 4, hipMemcpyDeviceToHost, st));
        M.vprep_done = true;
        DCD_TRY(stop_timers());
      }
      DCD_HIP(hipMemcpyAsync(rb_flags, M.d_flags.p, 16, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipMemcpyAsync(rb_total, M.scan_scr32.p + scan_nblocks(cap), 4, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipStreamSynchronize(st));
      v32 = (u32)*rb_total;
    }
    // (everything above was read back into pinned words: a copy into pageable memory would hold the host until it has
    // arrived, i.e. the kernels behind the emit kernel would only be queued once that one has finished)
    const u32* flags = rb_flags;
    M.voc_maxlen = flags[1];
    M.voc_nsym = *rb_nsym;
    M.last_text_coord = flags[2];
    if (strided) n_text_tok = *rb_ntok;
    V = v32;
    if (flags[0] || V * 2 > cap) { cap_log2 += 2; continue; }  // overflow or load > 0.5: rebuild larger
    break;
  }
  if (strided) {
    N = n_text_tok;
    if (N >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "more than 2^32 postings in one build segment");
    DCD_TRY(M.tupB.alloc(dev, N ? N : 1));
  }
  M.N = N; M.n_text_tok = n_text_tok;
  tr.mark("emit");
  M.cap_log2 = cap_log2; M.V = V;
  if (!fast_end) DCD_TRY(t_tok.stop());
  // maxCoord = coordinate of the last posting added (Build.cs:326); strided builds fetched it above
  if (n_text_tok && !strided) DCD_HIP(hipMemcpyAsync(&M.last_text_coord, &M.tupA.p[n_text_tok - 1].y, 4, hipMemcpyDeviceToHost, st));

  // ---- 3. vocabulary: compact, sort the word strings ordinally, materialise them -------------------
  // Plain text builds defer it: phase 2 sorts the tuples by dense id and runs the vocabulary beside that sort.
  // DOCODO_BUILD_RANK_KEYS=1 keeps the ordinal rank as the sort key (vocabulary first, as explicit postings / word codes need).
  static const bool env_rank_keys = getenv("DOCODO_BUILD_RANK_KEYS") && atoi(getenv("DOCODO_BUILD_RANK_KEYS")) != 0;
  M.id_keys = strided && !env_rank_keys && V != 0;
  M.early_ids = M.id_keys && early_ids;
  // the split pays when the cold vocabulary is much larger than the hot set (and fits the 2^20 remap-cache values)
  // (without the split the table is still a bijection slot -> [0, V): any such ids sort the tuples by term)
  M.n_hot = (M.id_keys && n_hot && V > 4ull * n_hot) ? n_hot : 0;
  M.id_hot = M.early_ids ? n_hot : 0;
  DCD_TRY(M.d_hist.alloc(dev, (u64)RS_MAX_BLOCKS << 10));
  if (!M.id_keys) {
    if (M.vprep_done) { M.vprep_done = false; DCD_HIP(hipMemsetAsync(M.d_flags.p + 1, 0, 4, st)); }   // (V == 0: redone below)
    if (fast_end) DCD_TRY(t_tok.start());     // (V == 0: the timers were stopped in the loop; nothing left to measure)
    if (fast_end) DCD_TRY(t_tok.stop());
    DCD_TRY(t_voc.start());
    probes.begin(DCD_K_VOCAB_SORT);
    DCD_TRY(build_vocab<CharT>(h, M, st, tv, M.d_hist.p, M.scan_scr32.p));
    probes.end();
    DCD_TRY(t_voc.stop());
    DCD_TRY(t_all.stop());
    DCD_HIP(hipStreamSynchronize(st));
  }
  const u32 total_chars = M.total_chars;
  DCD_HIP(hipGetLastError());
  tr.mark("vocab");
  M.ms_tok = t_tok.ms(); M.ms_voc = t_voc.ms(); M.ms_total = t_all.ms();
  probes.collect(M.ms_kernel, M.n_launch);
  tr.mark("collect");
  if (codes_mode) {
    M.h_word_chars.resize(total_chars);
    M.h_word_str_off.resize(V + 1);
    if (total_chars) DCD_HIP(hipMemcpy(M.h_word_chars.data(), M.word_chars.p, (size_t)total_chars * 2, hipMemcpyDeviceToHost));
    DCD_HIP(hipMemcpy(M.h_word_str_off.data(), M.word_str_off.p, (V + 1) * 4, hipMemcpyDeviceToHost));
  }
  return DCD_OK;
}

// one tuple per (posting, code of its word): counts, then the tuples in posting order (Builder.AddWord, Build.cs:302-316)
__global__ void k_codes_count(const uint2* __restrict__ tup, u64 n, const u32* __restrict__ rank_of_slot, const u32* __restrict__ code_off,
                              u32* __restrict__ cnt) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 w = rank_of_slot[tup[i].x];
  cnt[i] = code_off[w + 1] - code_off[w];
}
__global__ void k_codes_expand(const uint2* __restrict__ tup, u64 n, const u32* __restrict__ rank_of_slot, const u32* __restrict__ code_off,
                               const u32* __restrict__ code_key, const u64* __restrict__ out_off, uint2* __restrict__ out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint2 t = tup[i];
  const u32 w = rank_of_slot[t.x];
  u64 o = out_off[i];
  for (u32 e = code_off[w]; e < code_off[w + 1]; e++) out[o++] = make_uint2(code_key[e], t.y);
}

// ---- phase 2: (word -> codes) -> stable sort by key -> var-ushort pack -> ".index" image ---------------------------
template <typename CharT>
static int32_t build_phase2(dcd_build* h) {
  const int dev = h->device;
  hipStream_t st = h->st;
  StreamScope ss_(st);
  DCD_TRY(ensure_device(dev));
  BuildMid& M = *h->mid;
  HostTrace tr;
  tr.mark("begin2");
  EvTimer t_all, t_sort, t_pack, t_img;
  DCD_TRY(t_all.init(st, dev)); DCD_TRY(t_sort.init(st, dev)); DCD_TRY(t_pack.init(st, dev)); DCD_TRY(t_img.init(st, dev));
  Probes probes(st, dev);
  DeviceIndex& ix = h->ix;
  ix.device = dev;

  u64 N = M.N, V = M.V;
  uint2* a = M.tupA.p;
  uint2* b = M.tupB.p;
  // tuple.x = hash slot -> sort key: the ordinal rank of the word, or (id-keyed plain text builds) its dense id = the number of
  // used slots before it, which the scan behind the emit kernel left in slot_excl -- no vocabulary sort needed to start
  const bool idk = M.id_keys;
  const bool early = idk && M.early_ids;         // the tuples already carry their sort keys (ids of the numbering table)
  const u32* remap = early ? nullptr : idk ? M.slot_excl.p : M.vr.rank_of_slot.p;
  bool coord_sort = M.n_xdev != 0;            // mixed input is not in coordinate order
  DevBuf<uint2> expA, expB;
  u64 n_text_postings = M.n_text_tok;          // postings that came from the text (they precede the explicit ones)
  u64 max_coord_codes = 0;
  bool have_max_coord_codes = false;

  if (M.codes_mode) {
    // ---- host: the key table = all distinct code strings (String.CompareOrdinal order, Index.cs:142-151) ----
    if (!M.have_codes) {   // no dcd_build_set_codes: identity codes
      M.code_off.resize(V + 1);
      M.code_str.resize(V);
      for (u64 w = 0; w < V; w++) {
        M.code_off[w] = (u32)w;
        M.code_str[w] = ustr((const char16_t*)M.h_word_chars.data() + M.h_word_str_off[w], M.h_word_str_off[w + 1] - M.h_word_str_off[w]);
      }
      M.code_off[V] = (u32)V;
    }
    std::vector<ustr> keys(M.code_str);
    std::vector<const ExplicitPosting*> xcodes;
    for (const ExplicitPosting& x : h->xps)
      if (!x.is_word) { xcodes.push_back(&x); keys.emplace_back((const char16_t*)h->xpool.data() + x.off, x.len); }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    auto key_id = [&](const ustr& s) -> u32 { return (u32)(std::lower_bound(keys.begin(), keys.end(), s) - keys.begin()); };
    std::vector<u32> code_key(M.code_str.size());
    for (size_t e = 0; e < M.code_str.size(); e++) code_key[e] = key_id(M.code_str[e]);
    const u64 K = keys.size();
    // ---- device: expand the word postings ----
    DCD_TRY(t_all.start());
    t_sort.start_at_start_of(t_all);
    DevBuf<u32> d_code_off, d_code_key, d_cnt;
    DevBuf<u64> d_off, d_scr;
    DCD_TRY(d_code_off.alloc(dev, V + 1)); DCD_TRY(d_code_key.alloc(dev, code_key.size() + 1));
    DCD_HIP(hipMemcpyAsync(d_code_off.p, M.code_off.data(), (V + 1) * 4, hipMemcpyHostToDevice, st));
    if (!code_key.empty()) DCD_HIP(hipMemcpyAsync(d_code_key.p, code_key.data(), code_key.size() * 4, hipMemcpyHostToDevice, st));
    u64 NE = 0, n_text_exp = 0;
    if (N) {
      DCD_TRY(d_cnt.alloc(dev, N + 1)); DCD_TRY(d_off.alloc(dev, N + 1)); DCD_TRY(d_scr.alloc(dev, scan_scratch_elems(N + 1)));
      hipLaunchKernelGGL(k_codes_count, dim3((u32)div_up(N, 256)), dim3(256), 0, st, a, N, remap, d_code_off.p, d_cnt.p);
      DCD_HIP(hipMemsetAsync(d_cnt.p + N, 0, 4, st));
      scan_exclusive<u32, u64>(d_cnt.p, d_off.p, N + 1, d_scr.p, st);
      DCD_HIP(hipMemcpyAsync(&NE, d_off.p + N, 8, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipMemcpyAsync(&n_text_exp, d_off.p + M.n_text_tok, 8, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipStreamSynchronize(st));
    }
    const u64 NX = xcodes.size();
    const u64 NT = NE + NX;
    if (NT >= (1ull << 32)) DCD_FAIL(DCD_E_OVERFLOW, "more than 2^32 postings in one build segment");
    DCD_TRY(expA.alloc(dev, NT ? NT : 1));
    DCD_TRY(expB.alloc(dev, NT ? NT : 1));
    if (N) hipLaunchKernelGGL(k_codes_expand, dim3((u32)div_up(N, 256)), dim3(256), 0, st, a, N, remap, d_code_off.p, d_code_key.p, d_off.p, expA.p);
    if (NX) {
      std::vector<uint2> xt(NX);
      for (u64 i = 0; i < NX; i++) {
        xt[i].x = key_id(ustr((const char16_t*)h->xpool.data() + xcodes[i]->off, xcodes[i]->len));
        xt[i].y = (u32)(xcodes[i]->coord - h->opts.coord_base);
      }
      DCD_HIP(hipMemcpyAsync(expA.p + NE, xt.data(), NX * sizeof(uint2), hipMemcpyHostToDevice, st));
      DCD_HIP(hipStreamSynchronize(st));
    }
    if (n_text_exp) {
      u32 c = 0;
      DCD_HIP(hipMemcpyAsync(&c, &expA.p[n_text_exp - 1].y, 4, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipStreamSynchronize(st));
      max_coord_codes = h->opts.coord_base + c;
      have_max_coord_codes = true;
    }
    // the key strings replace the word strings
    std::vector<u16> kc;
    std::vector<u32> ko(K + 1, 0);
    for (u64 k = 0; k < K; k++) { ko[k] = (u32)kc.size(); kc.insert(kc.end(), (const u16*)keys[k].data(), (const u16*)keys[k].data() + keys[k].size()); }
    ko[K] = (u32)kc.size();
    DCD_TRY(M.word_chars.alloc(dev, kc.size() ? kc.size() : 1));
    DCD_TRY(M.word_str_off.alloc(dev, K + 1));
    if (!kc.empty()) DCD_HIP(hipMemcpyAsync(M.word_chars.p, kc.data(), kc.size() * 2, hipMemcpyHostToDevice, st));
    DCD_HIP(hipMemcpyAsync(M.word_str_off.p, ko.data(), (K + 1) * 4, hipMemcpyHostToDevice, st));
    DCD_HIP(hipStreamSynchronize(st));
    a = expA.p; b = expB.p; remap = nullptr;
    coord_sort = M.n_xdev != 0 || NX != 0;
    n_text_postings = n_text_exp;
    N = NT; V = K;
  } else {
    DCD_TRY(t_all.start());
    t_sort.start_at_start_of(t_all);
  }

  // ---- id-keyed builds: the vocabulary on a second (high-priority) stream, beside the tuple sort ------------------
  // Phase 1 ran the first half and fetched the two host-side parameters, so the chain (~40 small kernels, 0.15-0.2 ms of host
  // time to queue) is queued here without blocking; `ev_side` marks its end and the main stream waits for it in front of the
  // unit copy (step 6).  WHERE it is queued matters, because the kernels of the tuple sort fill every CU:
  //  * hot / cold builds queue it between the scan and the scatter kernel of pass A and order the main stream behind it
  //    right there: it runs beside the (long, 0.47 ms) histogram kernel, not beside the scatter kernel, which fills the LDS of
  //    every CU for a millisecond (the chain then takes 1.1-1.5 ms instead of 0.3-0.5 and slows the scatter kernel by 0.1 ms);
  //  * the others (early ids: histogram kernel of pass A 0.22 ms, shorter than the chain) queue it BEHIND the launch of the
  //    scatter kernel of pass A and order nothing: the first kernels of the chain run beside the histogram kernel, the rest
  //    squeeze in beside the scatter kernel, which they slow by ~0.05 ms -- measured against ordering the main stream behind
  //    the chain in front of that kernel (+0.14 ms of idle main stream) and against queueing the chain in two portions, one
  //    beside each histogram kernel (the histogram kernel of pass B occupies every wave slot until it ends: +0.12 ms).
  DevBuf<u32> side_hist, side_scr, id_dense;
  u64 id_range = 1;                    // early ids: they are dense per shard; the largest one + 1
  if (early) {
    id_range = std::max<u64>(1, M.id_hot);
    for (u32 s = 0; s < VOC_SHARDS; s++)
      if (M.id_cnt.c[s]) id_range = std::max<u64>(id_range, (u64)M.id_hot + (u64)(M.id_cnt.c[s] - 1) * VOC_SHARDS + s + 1);
  }
  hipStream_t vs = nullptr;
  hipEvent_t ev_side = nullptr, ev_dense = nullptr;     // (ev_dense: id_dense is ready -- the pack kernels wait for it)
  EvTimer t_vs;
  struct SideGuard {                   // stream and events go back to their caches (drained) however this function is left
    int dev; hipStream_t* s; hipEvent_t* e[2];
    ~SideGuard() {
      if (*s) { stream_release_high(dev, *s); *s = nullptr; }
      for (hipEvent_t* x : e) if (*x) { event_release(dev, *x); *x = nullptr; }
    }
  } side_guard{dev, &vs, {&ev_side, &ev_dense}};
  bool side_waited = false;
  auto queue_vocab_side = [&]() -> int32_t {
    if (ev_side) return DCD_OK;        // (queued already)
    DCD_TRY(stream_acquire_high(dev, &vs));
    ev_side = event_acquire(dev);
    ev_dense = event_acquire(dev);
    if (!ev_side || !ev_dense) DCD_FAIL(DCD_E_HIP, "hipEventCreate failed");
    StreamScope ss_v(vs);              // allocations below belong to the side stream
    TextView<CharT> tv;
    tv.text = (const CharT*)h->text_dev; tv.n_chars = h->n_chars; tv.pool = (const CharT*)M.d_pool_raw.p; tv.n_pool = M.n_pool;
    tv.tab = CharTables16{nullptr, nullptr};
    if (sizeof(CharT) == 2) DCD_TRY(get_char_tables16(dev, &tv.tab));
    DCD_TRY(side_hist.alloc(dev, (u64)RS_MAX_BLOCKS << 10));     // (the tuple sort is using M.d_hist / M.scan_scr32)
    DCD_TRY(side_scr.alloc(dev, scan_scratch_elems((u64)RS_MAX_BLOCKS << 10) + scan_scratch_elems(V) + 8));
    DCD_TRY(t_vs.init(vs, dev));
    DCD_TRY(t_vs.start());
    if (early) {                       // id -> dense id, for the term heads of the pack kernels
      DCD_TRY(id_dense.alloc(dev, id_range));
      hipLaunchKernelGGL(k_id_dense_table, dim3((u32)div_up(id_range, 256)), dim3(256), 0, vs, M.id_cnt, M.id_hot, (u32)id_range, id_dense.p);
      DCD_HIP(hipEventRecord(ev_dense, vs));
    }
    DCD_TRY(build_vocab<CharT>(h, M, vs, tv, side_hist.p, side_scr.p));
    DCD_TRY(t_vs.stop());
    DCD_HIP(hipEventRecord(ev_side, vs));
    return DCD_OK;
  };
  auto wait_vocab_side = [&]() -> int32_t {   // queues the chain (if that has not happened yet) and orders the main stream
    if (side_waited) return DCD_OK;           // behind it; no host blocking
    side_waited = true;
    DCD_TRY(queue_vocab_side());
    DCD_HIP(hipStreamWaitEvent(st, ev_side, 0));
    return DCD_OK;
  };
  if (idk && !M.vprep_done) DCD_FAIL(DCD_E_STATE, "id-keyed build without the prepared vocabulary");

  // ---- 4. sort the tuples by key (stable; explicit postings first need coordinate order) ----
  uint2* sorted = a;
  int passes = 0;
  u32 hot_n = 0, hot_s = 0;
  bool hot32 = false;                  // the hot region of `sorted` holds 4-byte coordinates
  const u32* hot_starts = nullptr;     // scanned histogram of pass A: hot term d starts at hot_starts[d * hot_stride]
  u32 hot_stride = 0;
  u64 n_hot_postings = 0;
  std::vector<hipEvent_t> hist_events, scatter_events;
  if (N) {
    if (coord_sort) {
      // mixed input is not in coordinate order: LSD over the coordinate bits first
      DCD_HIP(hipMemsetAsync(M.d_flags.p + 2, 0, 4, st));
      hipLaunchKernelGGL(k_max_y, dim3((u32)div_up(N, 256)), dim3(256), 0, st, a, N, M.d_flags.p + 2);
      u32 maxc = 0;
      DCD_HIP(hipMemcpyAsync(&maxc, M.d_flags.p + 2, 4, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipStreamSynchronize(st));
      hipLaunchKernelGGL(k_swap_xy, dim3((u32)div_up(N, 256)), dim3(256), 0, st, a, N);
      int p1 = 0;
      uint2* r = rs_sort(a, b, nullptr, N, (int)log2_ceil((u64)maxc + 1), M.d_hist.p, M.scan_scr32.p, st, &p1);
      passes += p1;
      if (r != a) { b = a; a = r; }
      hipLaunchKernelGGL(k_swap_xy, dim3((u32)div_up(N, 256)), dim3(256), 0, st, a, N);
    }
    int p2 = 0;
    RsProbe rp;
    rp.ev_hist = &hist_events; rp.ev_scatter = &scatter_events; rp.device = dev; rp.k_hist = DCD_K_RS_HIST; rp.k_scatter = DCD_K_RS_SCATTER;
    RsStrided sd;
    sd.cnt = M.tile_cnt.p; sd.stride = M.stride; sd.n_tiles = M.n_tiles;
    const bool strided = M.strided && a == M.tupA.p && (remap != nullptr || early);     // (codes mode sorts its own expanded tuples)
    const bool cache_ok = remap != nullptr && rs_cache_ok(1ull << M.cap_log2, V);
    hot_n = (idk && strided && M.n_hot && V > M.n_hot) ? M.n_hot : 0;
    // rs_hot_key packs {digit of pass A, cold id} into 32 bits: (hot_n << hot_s) + (d - hot_n) needs
    // log2_ceil(hot_n + 1) + hot_s of them.  A cold id range beyond 2^24 with 255 hot terms would wrap -- distinct hot terms would
    // share a key and their lists would be merged -- so such a build sorts by the plain ids (no split).
    if (hot_n && !rs_hot_key_fits(hot_n, std::max<u32>(1, log2_ceil((early ? id_range : V) - hot_n)))) hot_n = 0;
    if (hot_n) {
      // Hot / cold split.  Pass A scatters by the digit (key >> hot_s): every hot term gets a bucket of its own, all cold
      // tuples share the bucket hot_n.  A hot term is then complete -- contiguous, in coordinate order -- after ONE pass,
      // and only the cold region (a third of the tuples of a Zipf corpus) is sorted further, by the low hot_s bits.
      hot_s = std::max<u32>(1, log2_ceil((early ? id_range : V) - hot_n));
      int bits_a = (int)log2_ceil((u64)hot_n + 1);
      if (bits_a < 6) bits_a = 6;
      sd.hot_n = hot_n; sd.hot_s = hot_s;
      // the size of the hot region = scanned count of (digit hot_n, block 0): read back between the scan and the scatter
      // kernel of pass A, so that the host has it long before that kernel ends and queues the cold passes behind it
      struct NhRead { const u32* src; u32* dst; hipEvent_t ev; int32_t status; std::function<int32_t()>* side; int32_t side_status; } nh;
      const RsTilePlan tp = rs_tile_plan(M.n_tiles);
      u64* pin = pinned_words();
      if (!pin) DCD_FAIL(DCD_E_NOMEM, "hipHostMalloc failed");
      // (early ids: the emit kernel has counted the digits of this pass already -- no histogram kernel)
      const bool have_hist = early && M.histA_ready && hot_n == M.id_hot && bits_a <= 8;
      u32* const histA = have_hist ? M.d_histA.p : M.d_hist.p;
      sd.hist_ready = have_hist;
      // ... and its scatter kernel counts the first COLD pass's digit per block (see RsStrided::next_hist): that pass then needs
      // no histogram kernel either.  (The cold passes sort by the low hot_s bits, narrow digit first.)
      static const bool env_next = !(getenv("DOCODO_SORT_NEXT_HIST") && atoi(getenv("DOCODO_SORT_NEXT_HIST")) == 0);
      const int w0 = rs_schedule((int)hot_s, 9).width[0];
      const bool next_ok = env_next && w0 <= 8 && ((u64)tp.nblk << w0) <= ((u64)RS_MAX_BLOCKS << 10);
      if (next_ok) {
        DCD_TRY(M.d_hist1.alloc(dev, (u64)tp.nblk << w0));      // (every block of pass A writes its row: no memset)
        sd.next_hist = M.d_hist1.p; sd.next_bits = (u32)w0;
      }
      // round 4: the hot buckets leave pass A as 4-byte coordinates (RsStrided::hot32).  Only with the emit kernel's own histogram:
      // its scanned copy -- the buckets' borders, which the pack kernels need -- then lives in a buffer no later pass reuses.
      static const bool env_hot32 = !(getenv("DOCODO_BUILD_HOT32") && atoi(getenv("DOCODO_BUILD_HOT32")) == 0);   // (A/B knob)
      hot32 = env_hot32 && have_hist;
      sd.hot32 = hot32;
      hot_starts = histA; hot_stride = tp.nblk;
      nh.src = histA + (u64)hot_n * tp.nblk; nh.dst = (u32*)pin; nh.ev = event_acquire(dev); nh.status = DCD_OK;
      if (!nh.ev) DCD_FAIL(DCD_E_HIP, "hipEventCreate failed");
      // ... and the vocabulary chain is queued right there, behind the histogram kernel it is to run beside, and the main
      // stream ordered behind it (see queue_vocab_side)
      std::function<int32_t()> side_fn = wait_vocab_side;
      nh.side = early ? nullptr : &side_fn; nh.side_status = DCD_OK;     // (early ids: the chain goes behind the scatter kernel's launch, below)
      RsProbe rpa = rp;
      rpa.after_scan_arg = &nh;
      rpa.after_scan = [](void* arg, hipStream_t s) {
        NhRead* r = (NhRead*)arg;
        if (hipMemcpyAsync(r->dst, r->src, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipEventRecord(r->ev, s) != hipSuccess) r->status = DCD_E_HIP;
        if (r->side) r->side_status = (*r->side)();
      };
      rs_pass_dyn(bits_a, a, b, remap, N, (int)hot_s, histA, M.scan_scr32.p, st, rpa, cache_ok, &sd);
      passes++;
      if (early) nh.side_status = queue_vocab_side();
      int32_t vst = nh.side_status;
      if (vst == DCD_OK && nh.status == DCD_OK && hipEventSynchronize(nh.ev) != hipSuccess) nh.status = DCD_E_HIP;
      event_release(dev, nh.ev);
      DCD_TRY(vst);
      if (nh.status != DCD_OK) { (void)hipGetLastError(); DCD_FAIL(DCD_E_HIP, "reading the hot / cold boundary failed"); }
      const u64 Nh = *nh.dst, Nc = N - Nh;
      n_hot_postings = Nh;
      if (Nc && next_ok) {
        // first cold pass without a histogram kernel: pass A's scatter kernel counted its digit per block (sd.next_hist), and
        // block k of this pass sorts exactly the cold tuples block k of pass A wrote -- the range [off_k, off_k+1) of pass A's
        // scanned offsets of the cold digit (histA is scanned in place)
        const RsSchedule sc = rs_schedule((int)hot_s, 9);
        RsStrided s1;
        s1.cnt = nullptr; s1.stride = 0; s1.n_tiles = 0;
        s1.range = histA + (u64)hot_n * tp.nblk; s1.range_nblk = tp.nblk; s1.range_end = N;
        rs_pass_dyn(sc.width[0], b, a + Nh, nullptr, N, 0, M.d_hist1.p, M.scan_scr32.p, st, rp, false, &s1);
        uint2 *in2 = a + Nh, *out2 = b + Nh;
        int shift2 = sc.width[0];
        for (int i = 1; i < sc.npass; i++) {
          rs_pass_dyn(sc.width[i], in2, out2, nullptr, Nc, shift2, M.d_hist.p, M.scan_scr32.p, st, rp, false, nullptr);
          shift2 += sc.width[i];
          uint2* t2 = in2; in2 = out2; out2 = t2;
        }
        passes += sc.npass;
        if (in2 != b + Nh) DCD_HIP(hipMemcpyAsync(b + Nh, in2, Nc * sizeof(uint2), hipMemcpyDeviceToDevice, st));   // (odd number of cold passes)
      } else if (Nc) {
        int pc = 0;
        uint2* r = rs_sort(b + Nh, a + Nh, nullptr, Nc, (int)hot_s, M.d_hist.p, M.scan_scr32.p, st, &pc, rp, 9);
        passes += pc;
        if (r != b + Nh) DCD_HIP(hipMemcpyAsync(b + Nh, r, Nc * sizeof(uint2), hipMemcpyDeviceToDevice, st));   // (odd number of cold passes)
      }
      sorted = b;
    } else {
      const u64 key_range = early ? id_range : V ? V : 1;
      // id-keyed: the vocabulary chain is queued behind the launch of the first scatter kernel (see queue_vocab_side);
      // the hook fires in front of every pass but the first, the chain is queued once
      struct SideHook { std::function<int32_t()>* fn; int32_t status; } sh;
      std::function<int32_t()> side_fn = queue_vocab_side;
      sh.fn = &side_fn; sh.status = DCD_OK;
      RsProbe rps = rp;
      if (idk) {
        rps.after_scan_arg = &sh;
        rps.before_pass = [](void* arg, hipStream_t) { SideHook* s = (SideHook*)arg; if (s->status == DCD_OK) s->status = (*s->fn)(); };
      }
      sorted = rs_sort(a, b, remap, N, (int)log2_ceil(key_range), M.d_hist.p, M.scan_scr32.p, st, &p2, rps, 9, cache_ok, strided ? &sd : nullptr);
      passes += p2;
      DCD_TRY(sh.status);
    }
  }
  DCD_TRY(t_sort.stop());
  if (idk) {
    DCD_TRY(queue_vocab_side());       // (one-pass sorts: the hook never fired)
    if (early) DCD_HIP(hipStreamWaitEvent(st, ev_dense, 0));
  }
  tr.mark("sort-enqueue");

  // ---- 5. pack: deltas -> var-ushort units (IndexSequence.Builder.Add) -----------------------
  // id-keyed: the units come out in dense-id order into a scratch array sized for the worst case (a delta inside a term is
  // < 2^32: <= 3 units; the first posting of a term <= 5), so that no host round trip sits between the sort and the pack;
  // step 6 copies them into ordinal term order.
  t_pack.start_at_stop_of(t_sort);
  DCD_TRY(ix.term_unit_start.alloc(dev, V + 1));
  DCD_TRY(ix.term_item_start.alloc(dev, V + 1));
  DevBuf<u64> us_id, is_id;            // id-keyed: unit / item starts per dense id
  DevBuf<u16> units_id;
  if (idk) {
    DCD_TRY(us_id.alloc(dev, V + 1));
    DCD_TRY(is_id.alloc(dev, V + 1));
    DCD_TRY(units_id.alloc(dev, 3 * N + 2 * V + 8));
  }
  DevBuf<u64> enc_part, enc_scr;
  u64 enb = encode_nblocks(N);
  DCD_TRY(enc_part.alloc(dev, enb + 1));
  DCD_TRY(enc_scr.alloc(dev, scan_scratch_elems(enb) + 1));
  u64 U = 0;
  EncSrcPairs src;
  src.p = sorted; src.base = h->opts.coord_base;
  src.term_unit_start = idk ? us_id.p : ix.term_unit_start.p;
  src.term_item_start = idk ? is_id.p : ix.term_item_start.p;
  src.hot_n = hot_n; src.hot_s = hot_s;
  src.id_dense = early ? id_dense.p : nullptr;
  DevBuf<u32> hot_tile;
  if (hot32 && hot_n && n_hot_postings) {
    src.n_hot32 = n_hot_postings; src.hstart = hot_starts; src.hstride = hot_stride;
    const u64 nth = pack_hot_tiles(n_hot_postings);
    DCD_TRY(hot_tile.alloc(dev, nth));
    src.hot_tile = hot_tile.p;
    hipLaunchKernelGGL(k_pack_hot_tiles, dim3((u32)div_up(nth, 256)), dim3(256), 0, st, src, nth, hot_tile.p);
  }
  u64* d_U = enc_scr.p + scan_nblocks(enb);        // the scan leaves the number of units here
  if (N) {
    probes.begin(DCD_K_ENC_REDUCE);
    if ((reinterpret_cast<uintptr_t>(sorted) & 15) == 0)
      hipLaunchKernelGGL(k_pack_count, dim3((u32)enb), dim3(CODEC_THREADS), 0, st, src, N, enc_part.p);
    else
      hipLaunchKernelGGL((k_enc_reduce<EncSrcPairs>), dim3((u32)enb), dim3(CODEC_THREADS), 0, st, src, N, enc_part.p);
    probes.end();
    // ~67 k tile totals at C2 size: the multi-block scan (in place) instead of the single-block one
    scan_exclusive<u64, u64>(enc_part.p, enc_part.p, enb, enc_scr.p, st);
    if (!idk) {
      DCD_HIP(hipMemcpyAsync(&U, d_U, 8, hipMemcpyDeviceToHost, st));
      DCD_HIP(hipStreamSynchronize(st));
    }
  } else {
    DCD_HIP(hipMemsetAsync(d_U, 0, 8, st));
  }
  tr.mark("enc-reduce");
  if (!idk) DCD_TRY(ix.units.alloc(dev, U ? U : 1));
  if (N) {
    probes.begin(DCD_K_ENC_WRITE);
    hipLaunchKernelGGL(k_pack_write, dim3((u32)div_up(enb, PACK_TPB)), dim3(CODEC_THREADS), 0, st, src, N, enc_part.p, idk ? units_id.p : ix.units.p);
    probes.end();
  }
  DCD_TRY(ix.term_first.alloc(dev, V ? V : 1));
  DCD_TRY(ix.term_last.alloc(dev, V ? V : 1));
  if (!idk) {
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, st, ix.term_unit_start.p + V, U);
    hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, st, ix.term_item_start.p + V, N);
    if (V) hipLaunchKernelGGL(k_term_first_last, dim3((u32)div_up(V, 256)), dim3(256), 0, st, sorted, ix.term_item_start.p, V,
                              h->opts.coord_base, ix.term_first.p, ix.term_last.p);
  } else {
    // (id-keyed: the entries [V] of the id-ordered starts are never read -- PermDirF takes the totals instead)
  }
  DCD_TRY(t_pack.stop());

  // ---- 5b. id-keyed without the hot / cold split: the vocabulary now, while the main stream works through sort + pack
  if (idk) DCD_TRY(wait_vocab_side());
  DevBuf<u64> src_start;               // [V] first unit (in units_id) of the term with rank r

  // ---- 6. keys + ".index" image -----------------------------------------------------------------
  t_img.start_at_stop_of(t_pack);      // (includes waiting for the vocabulary chain, if the main stream has to)
  ix.term_str_off = std::move(M.word_str_off);
  ix.term_chars = std::move(M.word_chars);
  DevBuf<u64> rec_off, pay_off, scr64b;
  DCD_TRY(rec_off.alloc(dev, V + 1));
  DCD_TRY(pay_off.alloc(dev, V + 1));
  DCD_TRY(scr64b.alloc(dev, scan_scratch_elems(V ? V : 1)));
  u64 rec_total_pin = 0;
  if (idk) {
    // term order: unit / item starts and record offsets by ordinal rank = ONE scan of three lengths through dense_of_rank
    // (k_dir_*).  What the host is waiting for (the number of units and the size of the image) is read back into pinned
    // memory behind an event as soon as the partials are scanned; the apply kernel runs while the host takes that round trip.
    const u32* dor = M.vr.dense_of_rank.p;
    u64* pin = pinned_words();
    hipEvent_t ev = event_acquire(dev);
    if (!pin || !ev) { if (ev) event_release(dev, ev); DCD_FAIL(DCD_E_NOMEM, "pinned read-back buffer / event"); }
    const u64 nbd = scan_nblocks(V), pstride = nbd + 1;
    DevBuf<u64> dir_scr;
    int32_t a1 = dir_scr.alloc(dev, 3 * pstride + 4);
    int32_t a2 = src_start.alloc(dev, V ? V : 1);
    hipError_t e1 = hipSuccess, e3 = hipSuccess;
    if (a1 == DCD_OK && a2 == DCD_OK) {
      u64* totals = dir_scr.p + 3 * pstride;
      const PermDirF f{dor, us_id.p, is_id.p, (const u64*)d_U, N, V, ix.term_chars.p, ix.term_str_off.p};
      hipLaunchKernelGGL(k_dir_reduce, dim3((u32)nbd), dim3(SCAN_THREADS), 0, st, f, V, dir_scr.p, pstride);
      hipLaunchKernelGGL(k_dir_partials, dim3(1), dim3(1024), 0, st, dir_scr.p, pstride, nbd, totals);
      e1 = hipMemcpyAsync(pin + 8, totals, 24, hipMemcpyDeviceToHost, st);        // units, items, bytes
      e3 = hipEventRecord(ev, st);
      hipLaunchKernelGGL(k_dir_apply, dim3((u32)nbd), dim3(SCAN_THREADS), 0, st, f, V, (const u64*)dir_scr.p, pstride, (const u64*)totals,
                         ix.term_unit_start.p, ix.term_item_start.p, rec_off.p, src_start.p, (const uint2*)sorted, h->opts.coord_base,
                         ix.term_first.p, ix.term_last.p, src.n_hot32);
    }
    hipError_t e4 = (a1 == DCD_OK && a2 == DCD_OK && e1 == hipSuccess && e3 == hipSuccess) ? hipEventSynchronize(ev) : hipErrorUnknown;
    event_release(dev, ev);
    DCD_TRY(a1);
    DCD_TRY(a2);
    if (e4 != hipSuccess) { (void)hipGetLastError(); DCD_FAIL(DCD_E_HIP, "reading back the image size failed"); }
    U = pin[8];
    rec_total_pin = pin[10];
  }
  u64 rec_total = rec_total_pin;
  if (V && !idk) {
    scan_exclusive_f<RecSizeF, u64>(RecSizeF{ix.term_chars.p, ix.term_str_off.p, ix.term_unit_start.p}, rec_off.p, V, scr64b.p, st);
    DCD_HIP(hipMemcpyAsync(&rec_total, scr64b.p + scan_nblocks(V), 8, hipMemcpyDeviceToHost, st));
  }
  if (!idk) DCD_HIP(hipStreamSynchronize(st));
  tr.mark("pack");
  // DCD_BUILD_SHARD_ONLY (id-keyed builds): the shard is exported, not serialised or queried -- no image, no copy of the units
  // into term order; the export reads them where the pack kernel left them (units_by_key + unit_src_start)
  const bool shard_only = idk && (h->opts.flags & DCD_BUILD_SHARD_ONLY) != 0;
  // (id-keyed builds: the term-ordered posting array is not made here any more -- index_materialize_units, when a handle adopts
  //  the build; the build's own product is the image)
  ix.shard_only = shard_only;
  u64 max_coord = 0;
  {
    // coordinate of the last Add call: the later of (last explicit posting, last text token)
    const bool any_x = !h->xps.empty();
    const u64 text_last = have_max_coord_codes ? max_coord_codes : h->opts.coord_base + M.last_text_coord;
    if (h->last_is_explicit && any_x) max_coord = h->last_explicit_coord;
    else if (n_text_postings) max_coord = text_last;
    else if (any_x) max_coord = h->last_explicit_coord;
  }
  h->image_bytes = shard_only ? 0 : 8 + rec_total;
  if (!shard_only) {
  DCD_TRY(h->image.alloc(dev, h->image_bytes));
  hipLaunchKernelGGL(k_put_u64_bytes, dim3(1), dim3(1), 0, st, h->image.p, max_coord);
  }
  if (V && !shard_only) {
    hipLaunchKernelGGL(k_img_headers, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ix.term_chars.p, ix.term_str_off.p,
                       ix.term_unit_start.p, rec_off.p, V, (u64)8, h->image.p, pay_off.p);
    if (U) {
      probes.begin(DCD_K_IMG_UNITS);
      if (idk)
        hipLaunchKernelGGL((k_img_units<true, true, false>), dim3((u32)div_up(U, IMG_TILE)), dim3(IMG_THREADS), 0, st, (const u16*)units_id.p, U,
                           (const u64*)ix.term_unit_start.p, V, (const u64*)pay_off.p, h->image.p, (const u64*)src_start.p, (u16*)nullptr);
      else
        hipLaunchKernelGGL(k_img_units<false>, dim3((u32)div_up(U, IMG_TILE)), dim3(IMG_THREADS), 0, st, (const u16*)ix.units.p, U,
                           (const u64*)ix.term_unit_start.p, V, (const u64*)pay_off.p, h->image.p, (const u64*)nullptr, (u16*)nullptr);
      probes.end();
    }
  }
  DCD_TRY(t_img.stop());
  t_all.stop_at_stop_of(t_img);
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  if (idk) {                      // the units stay in sort-key order: term r's run starts at units_by_key[unit_src_start[r]]
    ix.units_by_key = std::move(units_id);
    ix.unit_src_start = std::move(src_start);
  }
  tr.mark("image");

  // ---- host mirrors + stats ----------------------------------------------------------------------
  ix.n_terms = V; ix.n_units = U; ix.n_postings = N; ix.max_coord = max_coord;
  ix.page_end = std::move(M.d_page_end);
  ix.host_ready = false;   // key / offset mirrors are fetched when an Index handle adopts the result
  ix.pages = h->pages;

  dcd_build_stats& s = h->stats;
  s.n_chars = h->n_chars; s.n_pages = h->pages.n_pages(); s.n_docs = h->pages.n_docs();
  s.n_postings = N; s.n_terms = V; s.n_units = U; s.max_coord = max_coord;
  s.index_bytes = h->image_bytes; s.list_bytes = 0;  // the list image is generated on demand
  if (idk) {   // (side-stream time, overlapped with the sort; the final synchronisation of the main stream covered its end)
    const float ms_voc_side = t_vs.ms();
    M.ms_voc = ms_voc_side; M.ms_kernel[DCD_K_VOCAB_SORT] += ms_voc_side; M.n_launch[DCD_K_VOCAB_SORT]++;
  }
  s.ms_tokenize = M.ms_tok; s.ms_vocab = M.ms_voc; s.ms_sort = t_sort.ms(); s.ms_pack = t_pack.ms(); s.ms_image = t_img.ms();
  s.ms_total = M.ms_total + t_all.ms();
  s.sort_passes = (u32)passes; s.vocab_capacity_log2 = M.cap_log2;
  s.id_keys = early ? 2u : idk ? 1u : 0u; s.n_hot_terms = hot_n; s.n_hot_postings = n_hot_postings;
  for (int k = 0; k < 8; k++) { s.ms_kernel[k] = M.ms_kernel[k]; s.n_launch[k] = M.n_launch[k]; }
  probes.collect(s.ms_kernel, s.n_launch);
  for (int which = 0; which < 2; which++) {
    std::vector<hipEvent_t>& ev = which ? scatter_events : hist_events;
    const int k = which ? DCD_K_RS_SCATTER : DCD_K_RS_HIST;
    for (size_t i = 0; i + 1 < ev.size(); i += 2) {
      float t = 0;
      if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) { s.ms_kernel[k] += t; s.n_launch[k]++; }
    }
    for (auto e : ev) event_release(dev, e);
  }
  tr.mark("mirrors+stats");
  h->mid.reset();
  tr.mark("free");
  h->finished = true;
  return DCD_OK;
}

extern "C" {

int32_t dcd_build_finish(dcd_build* h) {
  if (!h) DCD_FAIL(DCD_E_BADARG, "dcd_build_finish: NULL handle");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  DCD_TRY(ensure_device(h->device));
  EventScope ev_;            // (one setting for the whole build, whatever dcd_set_build_events does meanwhile)
  const bool wide = h->opts.encoding == DCD_ENC_UTF16LE;
  if (!h->mid) DCD_TRY(wide ? build_phase1<u16>(h, false) : build_phase1<u8>(h, false));
  return wide ? build_phase2<u16>(h) : build_phase2<u8>(h);
}

int32_t dcd_build_words(dcd_build* h, size_t* n_words, size_t* n_chars) {
  if (!h || !n_words || !n_chars) DCD_FAIL(DCD_E_BADARG, "dcd_build_words: NULL argument");
  if (h->finished) DCD_FAIL(DCD_E_STATE, "build already finished");
  if (!h->mid) {
    DCD_TRY(ensure_device(h->device));
    DCD_TRY(h->opts.encoding == DCD_ENC_UTF16LE ? build_phase1<u16>(h, true) : build_phase1<u8>(h, true));
  }
  *n_words = h->mid->V;
  *n_chars = h->mid->total_chars;
  return DCD_OK;
}

int32_t dcd_build_words_get(dcd_build* h, uint16_t* chars, size_t chars_cap, uint32_t* off, size_t off_cap) {
  if (!h || !off || (!chars && chars_cap)) DCD_FAIL(DCD_E_BADARG, "dcd_build_words_get: NULL argument");
  if (!h->mid || !h->mid->codes_mode) DCD_FAIL(DCD_E_STATE, "dcd_build_words_get: call dcd_build_words first");
  const BuildMid& M = *h->mid;
  if (chars_cap < M.h_word_chars.size() || off_cap < M.h_word_str_off.size()) return DCD_E_OVERFLOW;
  if (!M.h_word_chars.empty()) memcpy(chars, M.h_word_chars.data(), M.h_word_chars.size() * 2);
  memcpy(off, M.h_word_str_off.data(), M.h_word_str_off.size() * 4);
  return DCD_OK;
}

int32_t dcd_build_set_codes(dcd_build* h, const uint32_t* code_off, size_t n_words, const uint32_t* str_off, const uint16_t* chars,
                            size_t n_entries) {
  if (!h || !code_off || !str_off || (!chars && n_entries && str_off[n_entries])) DCD_FAIL(DCD_E_BADARG, "dcd_build_set_codes: NULL argument");
  if (!h->mid || !h->mid->codes_mode) DCD_FAIL(DCD_E_STATE, "dcd_build_set_codes: call dcd_build_words first");
  BuildMid& M = *h->mid;
  if (n_words != M.V) DCD_FAIL(DCD_E_BADARG, "dcd_build_set_codes: %zu words given, the build has %llu", n_words, (unsigned long long)M.V);
  if (code_off[0] != 0 || code_off[n_words] != n_entries) DCD_FAIL(DCD_E_BADARG, "dcd_build_set_codes: code_off must run from 0 to n_entries");
  for (size_t w = 0; w < n_words; w++) if (code_off[w] > code_off[w + 1]) DCD_FAIL(DCD_E_BADARG, "dcd_build_set_codes: code_off not ascending");
  std::vector<ustr> cs(n_entries);
  for (size_t e = 0; e < n_entries; e++) {
    if (str_off[e] > str_off[e + 1]) DCD_FAIL(DCD_E_BADARG, "dcd_build_set_codes: str_off not ascending");
    const size_t len = str_off[e + 1] - str_off[e];
    if (len == 0 || len > MAX_CODE_LEN) DCD_FAIL(DCD_E_UNSUPPORTED, "index codes must be 1..%u UTF-16 units long", MAX_CODE_LEN);
    cs[e] = ustr((const char16_t*)chars + str_off[e], len);
  }
  M.code_off.assign(code_off, code_off + n_words + 1);
  M.code_str = std::move(cs);
  M.have_codes = true;
  return DCD_OK;
}

int32_t dcd_build_get_stats(dcd_build* h, dcd_build_stats* out) {
  if (!h || !out) DCD_FAIL(DCD_E_BADARG, "dcd_build_get_stats: NULL argument");
  if (!h->finished) DCD_FAIL(DCD_E_STATE, "dcd_build_finish has not run");
  *out = h->stats;
  return DCD_OK;
}

int32_t dcd_build_index_image(dcd_build* h, uint8_t* out, size_t cap, size_t* n) {
  if (!h || !n) DCD_FAIL(DCD_E_BADARG, "dcd_build_index_image: NULL argument");
  if (!h->finished) DCD_FAIL(DCD_E_STATE, "dcd_build_finish has not run");
  if (h->ix.shard_only) DCD_FAIL(DCD_E_STATE, "a DCD_BUILD_SHARD_ONLY build has no .index image");
  *n = (size_t)h->image_bytes;
  if (!out || cap < h->image_bytes) return DCD_E_OVERFLOW;
  DCD_TRY(ensure_device(h->device));
  DCD_HIP(hipMemcpy(out, h->image.p, h->image_bytes, hipMemcpyDeviceToHost));
  return DCD_OK;
}

int32_t dcd_build_list_image(dcd_build* h, uint8_t* out, size_t cap, size_t* n) {
  if (!h || !n) DCD_FAIL(DCD_E_BADARG, "dcd_build_list_image: NULL argument");
  if (!h->finished) DCD_FAIL(DCD_E_STATE, "dcd_build_finish has not run");
  if (h->list_image.empty()) { write_list_image(h->pages, h->list_image); h->stats.list_bytes = h->list_image.size(); }
  *n = h->list_image.size();
  if (!out || cap < h->list_image.size()) return DCD_E_OVERFLOW;
  memcpy(out, h->list_image.data(), h->list_image.size());
  return DCD_OK;
}

int32_t dcd_build_save(dcd_build* h, const char* dir_utf8) {
  if (!h || !dir_utf8) DCD_FAIL(DCD_E_BADARG, "dcd_build_save: NULL argument");
  if (!h->finished) DCD_FAIL(DCD_E_STATE, "dcd_build_finish has not run");
  if (h->ix.shard_only) DCD_FAIL(DCD_E_STATE, "a DCD_BUILD_SHARD_ONLY build cannot be saved");
  std::vector<u8> img(h->image_bytes);
  size_t n = 0;
  DCD_TRY(dcd_build_index_image(h, img.data(), img.size(), &n));
  std::string d(dir_utf8);
  if (!d.empty() && d.back() != '/') d.push_back('/');
  FILE* f = fopen((d + ".index").c_str(), "wb");
  if (!f) DCD_FAIL(DCD_E_IO, "cannot create %s.index", d.c_str());
  bool ok = fwrite(img.data(), 1, img.size(), f) == img.size();
  ok = fclose(f) == 0 && ok;
  if (!ok) DCD_FAIL(DCD_E_IO, "short write to %s.index", d.c_str());
  if (h->list_image.empty()) { write_list_image(h->pages, h->list_image); h->stats.list_bytes = h->list_image.size(); }
  f = fopen((d + ".index.list").c_str(), "wb");
  if (!f) DCD_FAIL(DCD_E_IO, "cannot create %s.index.list", d.c_str());
  ok = fwrite(h->list_image.data(), 1, h->list_image.size(), f) == h->list_image.size();
  ok = fclose(f) == 0 && ok;
  if (!ok) DCD_FAIL(DCD_E_IO, "short write to %s.index.list", d.c_str());
  return DCD_OK;
}

}  // extern "C"

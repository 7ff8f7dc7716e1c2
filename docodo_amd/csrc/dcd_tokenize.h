// dcd_tokenize.h -- tokenise + vocabulary kernels of the build path.
//
// Replaces the hot loop of IndexTask (Build.cs:477-532): c = page.text.ToLower();
// foreach Match in Regex.Matches(c, @"\p{L}+|\p{N}+") with MIN_WORD_LENGTH <= len <= MAX_WORD_LENGTH
// -> index.AddWord(m.Value, coord + m.Index); and the dictionary lookup of Index.Builder.Add
// (Build.cs:324-333).  Coordinates are UTF-16 code-unit indices.
//
// Data flow per 4096-character tile (one 256-thread workgroup):
//   HBM text --16 B/lane--> lower-case + classify --> LDS (chars, boundary bitmask)
//   token starts = class != 0 at a class change or page start; length = distance to the next
//   boundary bit (one 64-bit window read + ctz);  kept tokens are compacted through a block scan;
//   each kept token is hashed from LDS, looked up / inserted in the global open-addressing table
//   (one 64-bit CAS word per term = {tag, length, position of the canonical occurrence}; equality
//   is verified against the canonical occurrence's characters, so the vocabulary is EXACT, not
//   probabilistic) and emitted as (slot, local coordinate) in text order.
#pragma once
#include "dcd_common.h"
#include "dcd_prims.h"

namespace dcd {

constexpr int TK_THREADS = 256;
constexpr int TK_CPT = 16;                           // characters per thread
constexpr int TK_TILE = TK_THREADS * TK_CPT;         // 4096
constexpr int TK_PRE = 16;                           // staged before the tile (previous character)
constexpr int TK_POST = 48;                          // staged after the tile (token end look-ahead >= 33)
constexpr int TK_SPAN = TK_PRE + TK_TILE + TK_POST;  // 4160
constexpr int TK_CHUNKS = TK_SPAN / TK_CPT;          // 260
constexpr int TK_MAXTOK = TK_TILE / 2 + 1;           // tokens of length 1 separated by one character

// vocabulary slot word: [63:48] tag  [47:42] length (1..32 -> stored as-is, 6 bits incl. 32)  [41:0] position
constexpr u64 VOC_POS_MASK = (1ull << 42) - 1;
__host__ __device__ inline u64 voc_word(u32 tag, u32 len, u64 pos) { return ((u64)(tag & 0xFFFFu) << 48) | ((u64)(len & 63u) << 42) | pos; }
__host__ __device__ inline u32 voc_len(u64 w) { return (u32)(w >> 42) & 63u; }
__host__ __device__ inline u64 voc_pos(u64 w) { return w & VOC_POS_MASK; }

// ---- character classes -----------------------------------------------------------------------
// class: 0 = other, 1 = \p{L}, 2 = \p{N}.  Latin-1 (byte == code unit) is computed arithmetically;
// UTF-16 goes through 64 Ki-entry tables (L2 resident).
struct CharTables16 {
  const u8* cls;     // [65536]
  const u16* lower;  // [65536]
};

__device__ __forceinline__ u32 lower_latin1(u32 c) {
  bool up = (c - 0x41u) < 26u || (c >= 0xC0u && c <= 0xDEu && c != 0xD7u);
  return c + (up ? 32u : 0u);
}
__device__ __forceinline__ u32 class_latin1_lower(u32 lc) {  // lc already lower-cased
  bool L = (lc - 0x61u) < 26u || (lc >= 0xDFu && lc != 0xF7u) || lc == 0xAAu || lc == 0xB5u || lc == 0xBAu;
  bool N = (lc - 0x30u) < 10u || lc == 0xB2u || lc == 0xB3u || lc == 0xB9u || (lc >= 0xBCu && lc <= 0xBEu);
  return L ? 1u : (N ? 2u : 0u);
}

template <typename CharT>
struct CharOps;
template <>
struct CharOps<u8> {
  __device__ static __forceinline__ u32 lower(u32 c, const CharTables16&) { return lower_latin1(c); }
  __device__ static __forceinline__ u32 cls(u32 lc, const CharTables16&) { return class_latin1_lower(lc); }
};
template <>
struct CharOps<u16> {
  __device__ static __forceinline__ u32 lower(u32 c, const CharTables16& t) { return t.lower[c]; }
  __device__ static __forceinline__ u32 cls(u32 lc, const CharTables16& t) { return t.cls[lc] & 3u; }
};

// ---- hashing ------------------------------------------------------------------------------------
__device__ __forceinline__ u32 fmix32(u32 h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
struct TokHash { u32 idx, tag; };
struct HashState {
  u32 h1, h2;
  __device__ __forceinline__ void init(u32 len) { h1 = 0x811C9DC5u ^ len; h2 = 0x9E3779B9u + len * 0x85EBCA6Bu; }
  __device__ __forceinline__ void add(u32 c) {
    h1 = (h1 ^ c) * 0x01000193u;
    h2 = (h2 + c) * 0xCC9E2D51u;
    h2 = (h2 << 13) | (h2 >> 19);
  }
  __device__ __forceinline__ TokHash fin() const {
    TokHash r;
    r.idx = fmix32(h1 ^ ((h2 << 16) | (h2 >> 16)));
    r.tag = fmix32(h2 + 0x27D4EB2Fu) >> 16;
    return r;
  }
};

// ---- text sources --------------------------------------------------------------------------------
// Position space of canonical occurrences: [0, n_chars) = the text buffer (original case),
// [n_chars, n_chars + n_pool) = the explicit-code pool (verbatim, Builder.Add codes).
template <typename CharT>
struct TextView {
  const CharT* text;
  u64 n_chars;
  const CharT* pool;
  CharTables16 tab;
  __device__ __forceinline__ u32 canon_char(u64 pos) const {
    if (pos < n_chars) return CharOps<CharT>::lower((u32)text[pos], tab);
    return (u32)pool[pos - n_chars];
  }
};

struct VocabTable {
  u64* slots;
  u32 mask;        // capacity - 1
  u32* overflow;   // set to 1 when a probe sequence gets too long
};

// Look up / insert the string `get(j)`, j < len, whose canonical position would be `pos`.
// Returns the slot index (0xFFFFFFFF on table overflow).
template <typename CharT, typename GetChar>
__device__ __forceinline__ u32 vocab_lookup_insert(const VocabTable& vt, const TextView<CharT>& tv, TokHash h, u32 len, u64 pos,
                                                   GetChar get) {
  const u64 mine = voc_word(h.tag, len, pos);
  u32 idx = h.idx & vt.mask;
  for (u32 probes = 0; probes < 4096u; probes++) {
    // agent-scope relaxed load: bypasses the per-CU L1 so a slot filled by another CU is seen as
    // soon as this XCD's L2 line is fresh; a stale "empty" is resolved by the CAS below.
    u64 w = __hip_atomic_load(&vt.slots[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w == 0) {
      u64 old = atomicCAS((unsigned long long*)&vt.slots[idx], 0ull, (unsigned long long)mine);
      if (old == 0) return idx;
      w = old;
    }
    if ((w >> 42) == (mine >> 42)) {  // tag and length agree -> verify the characters
      u64 cp = voc_pos(w);
      bool eq = true;
      for (u32 j = 0; j < len; j++) {
        if (tv.canon_char(cp + j) != get(j)) { eq = false; break; }
      }
      if (eq) return idx;
    }
    idx = (idx + 1) & vt.mask;
  }
  *vt.overflow = 1u;
  return 0xFFFFFFFFu;
}

// ---- tile tokeniser --------------------------------------------------------------------------------
struct TokParams {
  u64 n_chars;
  const u64* page_off;   // [n_pages + 1] device, ascending, page_off[0] = 0, page_off[n_pages] = n_chars
  u64 n_pages;
  u32 min_len, max_len;
  u32 uniform_page_chars;  // != 0: pages start at every multiple of this (page_off unused)
};

template <typename CharT>
struct TokLds {
  CharT txt[TK_SPAN];          // lower-cased characters, local index = global - tile_start + TK_PRE
  u32 bnd[TK_SPAN / 32 + 2];   // bit i: a token boundary lies before local position i
  u32 nz[TK_SPAN / 32 + 2];    // bit i: class(local i) != 0
  u32 pg[TK_SPAN / 32 + 2];    // bit i: a page starts at local position i
  u32 tok[TK_MAXTOK];          // compacted kept tokens: local start | len << 16
  u32 scan[TK_THREADS / 64 + 1];
};

// Stages one tile, finds the kept tokens and leaves them compacted in L.tok; returns their count.
template <typename CharT>
__device__ __forceinline__ u32 tokenize_tile(const CharT* __restrict__ text, const TokParams& P, const CharTables16& tab, u64 tile,
                                             TokLds<CharT>& L) {
  const u32 tid = threadIdx.x;
  const int64_t lo = (int64_t)(tile * TK_TILE) - TK_PRE;          // global index of local 0
  for (u32 i = tid; i < TK_SPAN / 32 + 2; i += TK_THREADS) L.pg[i] = 0;
  __syncthreads();
  // page starts inside [lo, lo + TK_SPAN)
  if (P.uniform_page_chars) {
    const u64 S = P.uniform_page_chars;
    u64 first = lo < 0 ? 0 : (u64)lo;
    u64 o = (first + S - 1) / S * S + (u64)tid * S;
    for (; (int64_t)o < lo + TK_SPAN && o <= P.n_chars; o += (u64)TK_THREADS * S) {
      u32 li = (u32)((int64_t)o - lo);
      atomicOr(&L.pg[li >> 5], 1u << (li & 31));
    }
  } else {
    u64 first = lo < 0 ? 0 : (u64)lo;
    u64 a = 0, b = P.n_pages + 1;  // lower_bound over page_off[0 .. n_pages]
    while (a < b) { u64 m = (a + b) >> 1; if (P.page_off[m] < first) a = m + 1; else b = m; }
    for (u64 p = a + tid; p <= P.n_pages; p += TK_THREADS) {
      u64 o = P.page_off[p];
      if ((int64_t)o >= lo + TK_SPAN) break;
      u32 li = (u32)((int64_t)o - lo);
      atomicOr(&L.pg[li >> 5], 1u << (li & 31));
    }
  }
  __syncthreads();
  // load, lower-case, classify; chunk c covers local [16c, 16c+16)
  for (u32 c = tid; c < TK_CHUNKS; c += TK_THREADS) {
    int64_t g0 = lo + (int64_t)c * TK_CPT;
    u32 ch[TK_CPT];
    if (g0 >= 0 && (u64)g0 + TK_CPT <= P.n_chars) {
      if (sizeof(CharT) == 1) {
        uint4 v = *reinterpret_cast<const uint4*>(text + g0);
        u32 wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < TK_CPT; j++) ch[j] = (wv[j >> 2] >> (8 * (j & 3))) & 0xFFu;
      } else {
        const uint4* q = reinterpret_cast<const uint4*>(text + g0);
        uint4 v0 = q[0], v1 = q[1];
        u32 wv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < TK_CPT; j++) ch[j] = (wv[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
      }
    } else {
#pragma unroll
      for (int j = 0; j < TK_CPT; j++) {
        int64_t g = g0 + j;
        ch[j] = (g >= 0 && (u64)g < P.n_chars) ? (u32)text[g] : 0u;
      }
    }
    // class of the character just before this chunk
    u32 prev_cls = 0;
    if (g0 - 1 >= 0 && (u64)(g0 - 1) < P.n_chars)
      prev_cls = CharOps<CharT>::cls(CharOps<CharT>::lower((u32)text[g0 - 1], tab), tab);
    u32 nzm = 0, bm = 0;
    u32 pgm = (L.pg[(c * TK_CPT) >> 5] >> ((c * TK_CPT) & 31)) & 0xFFFFu;
#pragma unroll
    for (int j = 0; j < TK_CPT; j++) {
      u32 lc = CharOps<CharT>::lower(ch[j], tab);
      u32 k = CharOps<CharT>::cls(lc, tab);
      L.txt[c * TK_CPT + j] = (CharT)lc;
      nzm |= (k != 0 ? 1u : 0u) << j;
      bm |= (k != prev_cls ? 1u : 0u) << j;
      prev_cls = k;
    }
    bm |= pgm;
    reinterpret_cast<u16*>(L.bnd)[c] = (u16)bm;
    reinterpret_cast<u16*>(L.nz)[c] = (u16)nzm;
  }
  if (tid < 4) {  // bits past the staged span read as "boundary" so windows terminate
    reinterpret_cast<u16*>(L.bnd)[TK_CHUNKS + tid] = 0xFFFFu;
    reinterpret_cast<u16*>(L.nz)[TK_CHUNKS + tid] = 0;
  }
  __syncthreads();
  // tokens starting in this thread's 16 tile positions (chunk tid + 1)
  const u32 c = tid + TK_PRE / TK_CPT;
  u32 starts = reinterpret_cast<u16*>(L.bnd)[c] & reinterpret_cast<u16*>(L.nz)[c];
  // pass 1: which starts are kept (length filter, Build.cs:528); pass 2 (after the scan) writes them
  u32 keepm = 0;
  for (u32 m = starts; m;) {
    u32 j = (u32)__builtin_ctz(m);
    m &= m - 1;
    u32 b = c * TK_CPT + j + 1;      // first candidate end
    u64 win = (((u64)L.bnd[(b >> 5) + 1] << 32) | L.bnd[b >> 5]) >> (b & 31);
    if ((b & 31) != 0) win |= (u64)L.bnd[(b >> 5) + 2] << (64 - (b & 31));
    u32 len = win ? (u32)__builtin_ctzll(win) + 1u : 65u;
    if (len >= P.min_len && len <= P.max_len) keepm |= 1u << j;
  }
  u32 total;
  u32 ex = block_scan_excl<u32, TK_THREADS / 64>((u32)__builtin_popcount(keepm), L.scan, &total);
  for (u32 m = keepm; m;) {
    u32 j = (u32)__builtin_ctz(m);
    m &= m - 1;
    u32 i = c * TK_CPT + j;
    u32 b = i + 1;
    u64 win = (((u64)L.bnd[(b >> 5) + 1] << 32) | L.bnd[b >> 5]) >> (b & 31);
    if ((b & 31) != 0) win |= (u64)L.bnd[(b >> 5) + 2] << (64 - (b & 31));
    u32 len = (u32)__builtin_ctzll(win) + 1u;
    L.tok[ex++] = i | (len << 16);
  }
  __syncthreads();
  return total;
}

}  // namespace dcd

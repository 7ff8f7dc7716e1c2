// dcd_tokenize.h -- tokenise + vocabulary device code of the build path (gfx950).
//
// Replaces the hot loop of IndexTask (Build.cs:477-532): c = page.text.ToLower();
// foreach Match in Regex.Matches(c, @"\p{L}+|\p{N}+") with MIN_WORD_LENGTH <= len <= MAX_WORD_LENGTH
// -> index.AddWord(m.Value, coord + m.Index); and the dictionary lookup of Index.Builder.Add
// (Build.cs:324-333).  Coordinates are UTF-16 code-unit indices.
//
// Data flow per 4096-character tile (one 256-thread workgroup, one 16-character chunk per lane):
//   HBM --16 B/lane--> registers: SWAR lower-casing + classification of 4 bytes per ALU op
//   (Latin-1 path; UTF-16 goes through 64 Ki-entry tables) -> two 16-bit masks per chunk
//   (letter / number) -> LDS (520 B) -> ONE barrier -> every lane rebuilds a 64-bit boundary window
//   (its own chunk + 48 characters of look-ahead) from the neighbours' masks: token starts =
//   class != 0 at a class change or page start, token length = ctz of the shifted window.
//   The count kernel stops here (popcount -> tile total).  The emit kernel also stages the
//   lower-cased text in LDS, compacts the kept tokens through a block scan, and for each token
//   hashes its 32-bit words from LDS, looks it up / inserts it in the global open-addressing table
//   (one 64-bit CAS word per term = {tag, length, position of the canonical occurrence}; equality is
//   verified word-wise against the canonical occurrence, so the vocabulary is EXACT, not
//   probabilistic) and writes (slot, local coordinate) in text order.
#pragma once
#include "dcd_common.h"
#include "dcd_prims.h"

namespace dcd {

// Geometry: the unit of work is ONE WAVE (64 lanes), every lane loads exactly one 16-character chunk.
// Lane 0 holds the chunk before the tile (only the class of its last character is used), lanes 1..59
// own the tile's chunks, lanes 60..63 hold the look-ahead (a token may run 32 characters past its
// start).  Wave tiles advance by 59 * 16 = 944 characters.  Nothing in the tokeniser needs a workgroup
// barrier: neighbour masks travel by 64-lane shuffles, the staged text and the token list live in a
// wave-private slice of LDS (ordered by a wavefront-scope fence), so the four waves of a workgroup run
// completely independently and a stalled wave never holds the other three at an s_barrier.
#ifndef DCD_TOK_NT
#define DCD_TOK_NT 1               // non-temporal text loads and tuple stores in the emit kernel: both are touched once, and as
#endif                             // plain accesses they evict the vocabulary table from L2 (measured: fetches 2.68 -> 2.31 GB, 1.41 -> 1.37 ms)
constexpr int TK_THREADS = 256;
constexpr int TK_WAVES = TK_THREADS / 64;
constexpr int TK_CPT = 16;                       // characters per lane
constexpr int TK_PRE = 1;                        // chunks before the tile
constexpr int TK_HALO = 4;                       // look-ahead chunks after the tile (>= 33 characters)
constexpr int TK_OWN = 64 - TK_PRE - TK_HALO;    // 59 chunks
constexpr int TK_TILE = TK_OWN * TK_CPT;         // 944 characters per wave tile
constexpr int TK_SPAN = 64 * TK_CPT;             // 1024 characters staged per wave tile
constexpr int TK_MAXTOK = TK_TILE;               // min_word_length 1 and alternating classes ("a1a1..."): every character is a token

// orders a wave's LDS writes before its later LDS reads of other lanes' data (no s_barrier involved)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// vocabulary slot word: [63:48] tag  [47:42] length (1..63)  [41:0] position of the canonical occurrence
constexpr u64 VOC_POS_MASK = (1ull << 42) - 1;
__host__ __device__ inline u64 voc_word(u32 tag, u32 len, u64 pos) { return ((u64)(tag & 0xFFFFu) << 48) | ((u64)(len & 63u) << 42) | pos; }
__host__ __device__ inline u32 voc_len(u64 w) { return (u32)(w >> 42) & 63u; }
__host__ __device__ inline u64 voc_pos(u64 w) { return w & VOC_POS_MASK; }

struct CharTables16 {
  const u8* cls;     // [65536] bit0 = \p{L}, bit1 = \p{N}
  const u16* lower;  // [65536] String.ToLower per code unit
};

// ---- SWAR Latin-1 (byte == UTF-16 code unit) --------------------------------------------------
// flags live in bit 7 of each byte; additions never carry across bytes (operands <= 0x7F + 0x7F)
__device__ __forceinline__ u32 swar_eq7(u32 x7, u32 k) {   // bit7 set where the 7-bit byte equals k
  u32 y = x7 ^ k;
  return ~((y + 0x7F7F7F7Fu) | y) & 0x80808080u;
}
template <bool LATIN = true>
__device__ __forceinline__ u32 swar_lower4(u32 w) {
  const u32 hi = w & 0x80808080u, x = w & 0x7F7F7F7Fu;
  u32 up = ((x + 0x3F3F3F3Fu) & ~(x + 0x25252525u)) & 0x80808080u & ~hi;        // 'A'..'Z'
  if (LATIN && hi) {
    u32 lat = ((x + 0x40404040u) & ~(x + 0x21212121u)) & hi;                    // 0xC0..0xDE
    u32 y = x ^ 0x57575757u;                                                    // except 0xD7
    up |= lat & (((y + 0x7F7F7F7Fu) | y) & 0x80808080u);
  }
  return w + (up >> 2);  // + 0x20 where upper case
}
// class flags of already lower-cased bytes: *Lf letters, *Df numbers (bit 7 per byte)
template <bool LATIN = true>
__device__ __forceinline__ void swar_class4(u32 lw, u32* Lf, u32* Df) {
  const u32 hi = lw & 0x80808080u, x = lw & 0x7F7F7F7Fu;
  u32 L = ((x + 0x1F1F1F1Fu) & ~(x + 0x05050505u)) & 0x80808080u & ~hi;         // 'a'..'z'
  u32 D = ((x + 0x50505050u) & ~(x + 0x46464646u)) & 0x80808080u & ~hi;         // '0'..'9'
  if (LATIN && hi) {
    u32 y = x ^ 0x77777777u;                                                    // >= 0xDF except 0xF7
    L |= (x + 0x21212121u) & hi & (((y + 0x7F7F7F7Fu) | y) & 0x80808080u);
    L |= (swar_eq7(x, 0x2A2A2A2Au) | swar_eq7(x, 0x35353535u) | swar_eq7(x, 0x3A3A3A3Au)) & hi;   // ª µ º
    D |= (swar_eq7(x, 0x32323232u) | swar_eq7(x, 0x33333333u) | swar_eq7(x, 0x39393939u) |       // ² ³ ¹
          swar_eq7(x, 0x3C3C3C3Cu) | swar_eq7(x, 0x3D3D3D3Du) | swar_eq7(x, 0x3E3E3E3Eu)) & hi;   // ¼ ½ ¾
  }
  *Lf = L;
  *Df = D;
}
__device__ __forceinline__ u32 swar_nibble(u32 f) { return ((((f >> 7) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu; }
// the flags (bit 7 of each byte, nothing else set) of TWO words -> one byte, flag of byte b of `lo` at bit b, of `hi` at bit 4 + b.
// One multiplication gathers all eight: the flags sit at bits 8b (lo) and 8b + 4 (hi); times 2^24 + 2^17 + 2^10 + 2^3 every one
// lands once in bits 24..31 and no two partial products meet below them (no carries).  (A 32-bit integer multiply issues at a
// quarter of the rate of the other operations: eight of them per 16-character chunk were a tenth of the tokeniser's front half.)
__device__ __forceinline__ u32 swar_byte8(u32 lo, u32 hi) { return (((lo >> 7) | (hi >> 3)) * 0x01020408u) >> 24; }

__device__ __forceinline__ u32 lower_latin1(u32 c) {
  bool up = (c - 0x41u) < 26u || (c >= 0xC0u && c <= 0xDEu && c != 0xD7u);
  return c + (up ? 32u : 0u);
}

// ---- chunk loading + classification ------------------------------------------------------------
// Loads 16 characters starting at global index g0 (may lie partly outside [0, n_chars): zeros),
// returns the lower-cased characters as packed 32-bit words and the letter / number masks.
template <typename CharT>
struct Chunk;

template <>
struct Chunk<u8> {
  static constexpr int NW = 4;
  u32 w[NW];
  u32 L, D;
  // `inside`: the whole staged span of the tile lies inside the text (workgroup-uniform) -> no guards
  __device__ __forceinline__ void load(const u8* __restrict__ text, int64_t g0, u64 n_chars, const CharTables16& tab, bool inside) {
    fetch(text, g0, n_chars, inside);
    classify(tab);
  }
  // fetch: the raw characters (a kernel that walks several tiles issues the next tile's fetch before it works on the
  // current one); classify: lower-casing + class masks
  __device__ __forceinline__ void fetch(const u8* __restrict__ text, int64_t g0, u64 n_chars, bool inside) {
    if (inside || (g0 >= 0 && (u64)g0 + TK_CPT <= n_chars)) {
#if DCD_TOK_NT
      typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(text + g0));   // read once: keep it from evicting the table
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
#else
      uint4 v = *reinterpret_cast<const uint4*>(text + g0);
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
#endif
    } else {
#pragma unroll
      for (int k = 0; k < NW; k++) {
        u32 x = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          int64_t g = g0 + k * 4 + b;
          if (g >= 0 && (u64)g < n_chars) x |= (u32)text[g] << (8 * b);
        }
        w[k] = x;
      }
    }
  }
  __device__ __forceinline__ void classify(const CharTables16&) {
    L = D = 0;
    // pure-ASCII waves (the common case) skip the Latin-1 ranges with ONE wave-uniform branch
    if (__any(((w[0] | w[1] | w[2] | w[3]) & 0x80808080u) != 0u)) {
      u32 lf[NW], df[NW];
#pragma unroll
      for (int k = 0; k < NW; k++) {
        w[k] = swar_lower4<true>(w[k]);
        swar_class4<true>(w[k], &lf[k], &df[k]);
      }
      L = swar_byte8(lf[0], lf[1]) | (swar_byte8(lf[2], lf[3]) << 8);
      D = swar_byte8(df[0], df[1]) | (swar_byte8(df[2], df[3]) << 8);
    } else {
      u32 lf[NW], df[NW];
#pragma unroll
      for (int k = 0; k < NW; k++) {
        w[k] = swar_lower4<false>(w[k]);
        swar_class4<false>(w[k], &lf[k], &df[k]);
      }
      L = swar_byte8(lf[0], lf[1]) | (swar_byte8(lf[2], lf[3]) << 8);
      D = swar_byte8(df[0], df[1]) | (swar_byte8(df[2], df[3]) << 8);
    }
  }
};

template <>
struct Chunk<u16> {
  static constexpr int NW = 8;
  u32 w[NW];
  u32 L, D;
  __device__ __forceinline__ void load(const u16* __restrict__ text, int64_t g0, u64 n_chars, const CharTables16& tab, bool inside) {
    fetch(text, g0, n_chars, inside);
    classify(tab);
  }
  __device__ __forceinline__ void fetch(const u16* __restrict__ text, int64_t g0, u64 n_chars, bool inside) {
    if (inside || (g0 >= 0 && (u64)g0 + TK_CPT <= n_chars)) {
      const uint4* q = reinterpret_cast<const uint4*>(text + g0);
      uint4 a = q[0], b = q[1];
      w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < NW; k++) {
        u32 x = 0;
#pragma unroll
        for (int b = 0; b < 2; b++) {
          int64_t g = g0 + k * 2 + b;
          if (g >= 0 && (u64)g < n_chars) x |= (u32)text[g] << (16 * b);
        }
        w[k] = x;
      }
    }
  }
  __device__ __forceinline__ void classify(const CharTables16& tab) {
    L = D = 0;
    // pure-ASCII waves (English text in a C# string) are classified arithmetically, two characters per 32-bit operation,
    // with ONE wave-uniform branch; anything else goes through the Unicode tables (two gathers per character)
    if (__any(((w[0] | w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) & 0xFF80FF80u) != 0u)) {
#pragma unroll
      for (int k = 0; k < NW; k++) {
        u32 c0 = tab.lower[w[k] & 0xFFFFu], c1 = tab.lower[w[k] >> 16];
        u32 k0 = tab.cls[c0], k1 = tab.cls[c1];
        w[k] = c0 | (c1 << 16);
        L |= ((k0 & 1u) | ((k1 & 1u) << 1)) << (2 * k);
        D |= (((k0 >> 1) & 1u) | (((k1 >> 1) & 1u) << 1)) << (2 * k);
      }
    } else {
      // x >= a per 16-bit field (x, a < 0x8000): bit 15 of ((x | 0x8000) - a); no borrow crosses the fields
      auto ge2 = [](u32 x, u32 a) -> u32 { return ((x | 0x80008000u) - a * 0x00010001u) & 0x80008000u; };
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const u32 x = w[k];
        const u32 up = ge2(x, 'A') & ~ge2(x, 'Z' + 1);          // upper-case letters: + 0x20
        const u32 l = x + (up >> 10);
        const u32 lf = ge2(l, 'a') & ~ge2(l, 'z' + 1);
        const u32 df = ge2(x, '0') & ~ge2(x, '9' + 1);
        w[k] = l;
        L |= (((lf >> 15) & 1u) | ((lf >> 30) & 2u)) << (2 * k);
        D |= (((df >> 15) & 1u) | ((df >> 30) & 2u)) << (2 * k);
      }
    }
  }
};

// ---- hashing (over the 32-bit words of the lower-cased string, zero padded) -------------------------
__device__ __forceinline__ u32 fmix32(u32 h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
struct TokHash { u32 idx, tag; };
#ifdef DCD_HASH2   // the two-lane hash of round 1 (A/B builds)
struct HashState {
  u32 h1, h2;
  __device__ __forceinline__ void init(u32 len) { h1 = 0x811C9DC5u ^ (len * 0x9E3779B1u); h2 = 0x85EBCA6Bu + len; }
  __device__ __forceinline__ void add(u32 w) {
    h1 = (h1 ^ w) * 0xCC9E2D51u;
    h1 = (h1 << 15) | (h1 >> 17);
    h2 = (h2 + w) * 0x1B873593u;
    h2 ^= h2 >> 13;
  }
  __device__ __forceinline__ TokHash fin() const {
    TokHash r;
    r.idx = fmix32(h1 ^ ((h2 << 16) | (h2 >> 16)));
    r.tag = fmix32(h2 + 0x27D4EB2Fu) >> 16;
    return r;
  }
};
#else
// One 32-bit lane: multiply-xorshift per word, murmur3 finaliser for the slot index; the 16-bit tag is a second
// multiplicative scramble of the same 32 bits.  The emit kernel is bound by VALU issue and the hash is a quarter of a
// token's instructions; the table is EXACT whatever the hash (a tag match is always confirmed against the string), so the
// hash only decides probe lengths and how often the 16-byte compare runs in vain (index and tag share the 32 bits:
// ~2^-12 of the probes at 2^20 slots instead of 2^-16).
// (Measured in round 2 and not kept: a combine without multiplies -- xor, rotate, times 5 plus a constant -- and the top half
// of the index as the tag: no difference in the emit kernel, 1.22-1.25 ms either way.)
struct HashState {
  u32 h;
  __device__ __forceinline__ void init(u32 len) { h = 0x811C9DC5u ^ (len * 0x9E3779B1u); }
  __device__ __forceinline__ void add(u32 w) {
    h = (h ^ w) * 0xCC9E2D51u;
    h ^= h >> 15;
  }
  __device__ __forceinline__ TokHash fin() const {
    TokHash r;
    r.idx = fmix32(h);
    r.tag = (r.idx * 0x9E3779B1u) >> 16;
    return r;
  }
};
#endif

// word k of the byte string starting at byte offset `off` of a 4-byte aligned array
__device__ __forceinline__ u32 str_word(const u32* __restrict__ base, u64 off, u32 k, u64 n_bytes) {
  u64 a = (off >> 2) + k;
  u32 sh = (u32)off & 3u;
  u32 lo = base[a];
  if (sh == 0) return lo;
  u32 hi = ((a + 1) << 2) < n_bytes ? base[a + 1] : 0u;
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
}
// same for an LDS array (32-bit offsets, the array is padded so the second word always exists)
__device__ __forceinline__ u32 lds_word(const u32* base, u32 off, u32 k) {
  u32 a = (off >> 2) + k;
  return __builtin_amdgcn_alignbyte(base[a + 1], base[a], off & 3u);
}
__device__ __forceinline__ u32 tail_mask(u32 nbytes, u32 k) {  // mask of word k of an nbytes-long string
  u32 rem = nbytes - 4u * k;
  return rem >= 4u ? 0xFFFFFFFFu : ((1u << (8u * rem)) - 1u);
}

// ---- canonical occurrences ------------------------------------------------------------------------
// Position space: [0, n_chars) = the text buffer (original case), [n_chars, ...) = the explicit-code
// pool (verbatim Builder.Add codes).  Both arrays are 4-byte aligned.
template <typename CharT>
struct TextView {
  const CharT* text;
  u64 n_chars;
  const CharT* pool;
  u64 n_pool;
  CharTables16 tab;
  __device__ __forceinline__ u32 canon_char(u64 pos) const {   // lower-cased code unit at pos
    if (pos < n_chars) {
      u32 c = (u32)text[pos];
      return sizeof(CharT) == 1 ? lower_latin1(c) : (u32)tab.lower[c];
    }
    return (u32)pool[pos - n_chars];
  }
  // lower-cased word k (unmasked) of the string whose first character is at `pos`
  __device__ __forceinline__ u32 canon_word(u64 pos, u32 k) const {
    if (pos < n_chars) {
      u32 w = str_word(reinterpret_cast<const u32*>(text), pos * sizeof(CharT), k, n_chars * sizeof(CharT));
      if (sizeof(CharT) == 1) return swar_lower4<true>(w);
      return (u32)tab.lower[w & 0xFFFFu] | ((u32)tab.lower[w >> 16] << 16);
    }
    return str_word(reinterpret_cast<const u32*>(pool), (pos - n_chars) * sizeof(CharT), k, n_pool * sizeof(CharT));
  }
};

struct VocabTable {
  u64* slots;      // slot i's word is slots[i * stride]
  u32 mask;        // capacity - 1
  u32* overflow;   // set to 1 when a probe sequence gets too long
  u32* pool;       // optional: VocPool<CharT>::W words per slot, zeroed; the first words of the slot's string (see below):
                   // slot i's entry is pool[i * pstride ...]
  u32 stride = 1;  // in u64 units.  Pooled tables are arrays of records {slot word, padding, string}: stride = W, and
  u32 pstride = 0; // pool = (u32*)slots + W with pstride = 2 W -- a token that misses the caches fetches ONE line, not two
  u32* id_ctr = nullptr;   // optional (record tables only): VOC_SHARDS counters -- the table NUMBERS its terms (see vocab_lookup_insert_ids)
  const u32* id_base = nullptr;   // ... new terms are numbered from *id_base (the ids below it belong to the hot terms, k_hot_select)
  u32* new_slots = nullptr;       // sampling launch: [0] = number of slots it claimed, [1 ..] = the slots
};
constexpr u32 VOC_UNRESOLVED = 0x80000000u;   // tuple key = VOC_UNRESOLVED | slot: the term's id was not visible yet (fixed up after the launch)
// words of a string kept next to its slot: 16 characters' worth
template <typename CharT> struct VocPool { static constexpr u32 W = 4 * sizeof(CharT); };

// Look up / insert the string whose (masked) words are get(k), k < nw, nbytes long, `len` characters,
// canonical position `pos`.  Returns the slot index (0xFFFFFFFF on table overflow).
template <typename CharT, typename GetWord>
__device__ __forceinline__ u32 vocab_lookup_insert(const VocabTable& vt, const TextView<CharT>& tv, TokHash h, u32 len, u64 pos,
                                                   GetWord get) {
  const u64 mine = voc_word(h.tag, len, pos);
  const u32 nbytes = len * (u32)sizeof(CharT), nw = (nbytes + 3u) >> 2;
  u32 idx = h.idx & vt.mask;
  for (u32 probes = 0; probes < 4096u; probes++) {
    // Plain (L1-cacheable) load: hot slots (Zipf head) are served by each CU's own L1 instead of
    // hammering one L2 channel.  A slot word only ever changes 0 -> value, so a stale read can only
    // be a stale "empty", which the CAS below repairs (it returns the current value); a non-empty
    // read is final.  Correct for any workgroup -> XCD placement.
    u64* sp = vt.slots + (u64)idx * vt.stride;
    u64 w = *sp;
    if (w == 0) {
      u64 old = atomicCAS((unsigned long long*)sp, 0ull, (unsigned long long)mine);
      if (old == 0) return idx;
      w = old;
    }
    if ((w >> 42) == (mine >> 42)) {  // tag and length agree -> verify the characters word by word
      const u64 cp = voc_pos(w);
      bool eq = true;
      for (u32 k = 0; k < nw; k++) {
        if (((tv.canon_word(cp, k) ^ get(k)) & tail_mask(nbytes, k)) != 0u) { eq = false; break; }
      }
      if (eq) return idx;
    }
    idx = (idx + 1) & vt.mask;
  }
  *vt.overflow = 1u;
  return 0xFFFFFFFFu;
}

// The same with a STRING POOL beside the table.  Verifying a hit against the canonical occurrence costs, per word, an
// unaligned two-load fetch from the text, its lower-casing and a re-read of the token's own word from LDS -- half of the
// instructions of the emit kernel, which is bound by VALU issue.  Here the first W words of every slot's string (lower-cased,
// zero padded) live in pool[slot * W ...]: a hit is confirmed by one aligned 16-byte load per four words and a few XORs.
// Soundness does not depend on cache coherence: a pool word is only ever 0 or the word of the slot's string (whoever writes
// it -- the inserting lane, or a lane that has just verified the string against the text -- writes the same values), token
// words are never 0, so "all words equal" implies the strings are equal; any other outcome (entry not written yet, stale
// zeros in this CU's or XCD's cache, a different string, a string longer than the pool entry) falls back to the exact
// comparison with the text, after which the lane writes the entry itself, healing its own cache hierarchy.
// wv[0..W): the token's masked words, 0 beyond its length.
template <typename CharT, typename GetWord>
__device__ __forceinline__ u32 vocab_lookup_insert_pooled(const VocabTable& vt, const TextView<CharT>& tv, TokHash h, u32 len, u64 pos,
                                                          const u32 (&wv)[VocPool<CharT>::W], GetWord get) {
  constexpr u32 W = VocPool<CharT>::W;
  const u64 mine = voc_word(h.tag, len, pos);
  const u32 nbytes = len * (u32)sizeof(CharT), nw = (nbytes + 3u) >> 2;
  u32 idx = h.idx & vt.mask;
  for (u32 probes = 0; probes < 4096u; probes++) {
    u64* sp = vt.slots + (u64)idx * vt.stride;
    u64 w = *sp;
    uint4* pe = reinterpret_cast<uint4*>(vt.pool + (u64)idx * vt.pstride);
    if (w == 0) {
      u64 old = atomicCAS((unsigned long long*)sp, 0ull, (unsigned long long)mine);
      if (old == 0) {
#pragma unroll
        for (u32 q = 0; q < W / 4; q++) pe[q] = make_uint4(wv[4 * q], wv[4 * q + 1], wv[4 * q + 2], wv[4 * q + 3]);
        if (vt.new_slots) vt.new_slots[1u + atomicAdd(vt.new_slots, 1u)] = idx;      // (sampling launch: the caller sized the list for every token)
        return idx;
      }
      w = old;
    }
    if ((w >> 42) == (mine >> 42)) {  // tag and length agree
      u32 diff = 0;
#pragma unroll
      for (u32 q = 0; q < W / 4; q++) {
        const uint4 p = pe[q];
        diff |= (p.x ^ wv[4 * q]) | (p.y ^ wv[4 * q + 1]) | (p.z ^ wv[4 * q + 2]) | (p.w ^ wv[4 * q + 3]);
      }
      if (diff == 0 && nw <= W) return idx;
      const u64 cp = voc_pos(w);
      bool eq = true;
      for (u32 k = 0; k < nw; k++) {
        if (((tv.canon_word(cp, k) ^ get(k)) & tail_mask(nbytes, k)) != 0u) { eq = false; break; }
      }
      if (eq) {
        if (diff != 0) {
#pragma unroll
          for (u32 q = 0; q < W / 4; q++) pe[q] = make_uint4(wv[4 * q], wv[4 * q + 1], wv[4 * q + 2], wv[4 * q + 3]);
        }
        return idx;
      }
    }
    idx = (idx + 1) & vt.mask;
  }
  *vt.overflow = 1u;
  return 0xFFFFFFFFu;
}

// The same for a record table that NUMBERS its terms while it is filled (VocabTable::id_ctr): the lane that wins the slot
// takes the next number of its shard and publishes id + 1 in the record's padding word, next to the slot word.  A later token
// of the term gets the id from the 16 bytes {slot word, id + 1, -} it loads anyway: the slot word agrees on tag and length,
// the string granule confirms the characters (granule words are only ever 0 or final, token words are never 0), and a non-zero
// id word is final, whatever the caches did to the record.  An id that is not visible (this CU / XCD holds the line from
// before it was written -- plain loads are served by L1 and by the XCD's own L2, which nobody refreshes) is fetched with an
// atomic, executed at the memory side where the inserter put it, and the lane then stores it into the padding word itself,
// which heals its own cache hierarchy for the tokens that follow; if the inserter is still between its compare-and-swap and
// its publish, the lane returns VOC_UNRESOLVED | slot, which a fix-up pass behind the launch replaces.  The tuples then carry
// the sort key straight from the tokeniser: no slot -> id table, no remap gathers in the first radix pass.
// shard_off = shard * VOC_CTR_STRIDE.  Returns the id, VOC_UNRESOLVED | slot, or 0xFFFFFFFF on table overflow.
template <typename CharT, typename GetWord>
__device__ __forceinline__ u32 vocab_lookup_insert_ids(const VocabTable& vt, const TextView<CharT>& tv, TokHash h, u32 len, u64 pos,
                                                       const u32 (&wv)[VocPool<CharT>::W], GetWord get, u32 shard_off) {
  constexpr u32 W = VocPool<CharT>::W;
  const u64 mine = voc_word(h.tag, len, pos);
  const u32 nbytes = len * (u32)sizeof(CharT), nw = (nbytes + 3u) >> 2;
  u32 idx = h.idx & vt.mask;
  for (u32 probes = 0; probes < 4096u; probes++) {
    u64* sp = vt.slots + (u64)idx * vt.stride;
    const uint4 head = *reinterpret_cast<const uint4*>(sp);                      // slot word, id + 1, -
    u64 w = (u64)head.x | ((u64)head.y << 32);
    u32 idw = head.z;
    u32* pad = reinterpret_cast<u32*>(sp) + 2;
    uint4* pe = reinterpret_cast<uint4*>(vt.pool + (u64)idx * vt.pstride);
    // the string granule is fetched WITH the head, not behind the test of the slot word: one memory round trip per probe
    // instead of two dependent ones (measured: 0.3 ms of the emit kernel)
    uint4 gr[W / 4];
#pragma unroll
    for (u32 q = 0; q < W / 4; q++) gr[q] = pe[q];
    if (w == 0) {
      u64 old = atomicCAS((unsigned long long*)sp, 0ull, (unsigned long long)mine);
      if (old == 0) {
        const u32 id = (vt.id_base ? *vt.id_base : 0u) + atomicAdd(vt.id_ctr + shard_off, 1u) * VOC_SHARDS + shard_off / VOC_CTR_STRIDE;
        atomicExch(pad, id + 1u);          // published at the memory side, like the slot word
#pragma unroll
        for (u32 q = 0; q < W / 4; q++) pe[q] = make_uint4(wv[4 * q], wv[4 * q + 1], wv[4 * q + 2], wv[4 * q + 3]);
        return id;
      }
      w = old;
      idw = 0;
    }
    if ((w >> 42) == (mine >> 42)) {  // tag and length agree
      u32 diff = 0;
#pragma unroll
      for (u32 q = 0; q < W / 4; q++) {
        const uint4 p = gr[q];
        diff |= (p.x ^ wv[4 * q]) | (p.y ^ wv[4 * q + 1]) | (p.z ^ wv[4 * q + 2]) | (p.w ^ wv[4 * q + 3]);
      }
      bool eq = diff == 0 && nw <= W;
      if (eq && idw != 0) return idw - 1u;
      if (!eq) {
        const u64 cp = voc_pos(w);
        eq = true;
        for (u32 k = 0; k < nw; k++) {
          if (((tv.canon_word(cp, k) ^ get(k)) & tail_mask(nbytes, k)) != 0u) { eq = false; break; }
        }
      }
      if (eq) {
        if (diff != 0) {
#pragma unroll
          for (u32 q = 0; q < W / 4; q++) pe[q] = make_uint4(wv[4 * q], wv[4 * q + 1], wv[4 * q + 2], wv[4 * q + 3]);
        }
        if (idw) return idw - 1u;
        const u32 idp = atomicOr(pad, 0u);
        if (!idp) return VOC_UNRESOLVED | idx;      // the inserter is between its compare-and-swap and its atomicExch
        *pad = idp;
        return idp - 1u;
      }
    }
    idx = (idx + 1) & vt.mask;
  }
  *vt.overflow = 1u;
  return 0xFFFFFFFFu;
}

// ---- tile tokeniser ----------------------------------------------------------------------------------
struct TokParams {
  // coordinate runs: characters [run_off[r], run_off[r+1]) have local coordinate = index + run_delta[r]
  // (header pages advance the coordinate without contributing text, Build.cs:485-524)
  const u32* run_off;
  const u32* run_delta;
  u32 n_runs;              // <= 1: coordinate == index (+ run_delta[0] if n_runs == 1)
  u64 n_chars;
  const u32* page_bits;    // explicit pages (uniform_page_chars == 0): bit i = a page starts at character i; zero padded
                           // for TK_SPAN + 128 bits beyond n_chars (k_page_bits, built once per build from the page offsets)
  u64 n_pages;
  u32 min_len, max_len;
  u32 uniform_page_chars;  // != 0: pages start at every multiple of this
};

template <typename CharT, bool STAGE_TEXT>
struct TokLds {                                                  // one per WAVE
  u32 txt[STAGE_TEXT ? TK_SPAN * sizeof(CharT) / 4 + 4 : 1];     // lower-cased staged text (index 0 = staged position 0)
  u32 tok[STAGE_TEXT ? TK_MAXTOK + 3 : 1];                       // kept tokens: staged position | len << 16
};

// staged position p of wave tile t  <->  text index  t * TK_TILE - TK_PRE * TK_CPT + p
__device__ __forceinline__ int64_t tile_origin(u64 tile) { return (int64_t)(tile * TK_TILE) - TK_PRE * TK_CPT; }

// After this call: `keep` = 16-bit mask of kept token starts in this lane's chunk (0 for the lanes that
// only hold context), `bnd` = 64-bit boundary window whose bit 0 is the first character of the chunk.
// Wave-synchronous: no workgroup barrier.
// pre: the tile's characters already fetched by the caller (Chunk::fetch at tile_lane_g0), or nullptr.
__device__ __forceinline__ int64_t tile_lane_g0(u64 tile) { return tile_origin(tile) + (int64_t)lane_id() * TK_CPT; }
__device__ __forceinline__ bool tile_inside(u64 tile, u64 n_chars) {
  const int64_t org = tile_origin(tile);
  return org >= 0 && (u64)org + TK_SPAN <= n_chars;
}
template <typename CharT, bool STAGE_TEXT>
__device__ __forceinline__ void tokenize_chunk(const CharT* __restrict__ text, const TokParams& P, const CharTables16& tab, u64 tile,
                                               TokLds<CharT, STAGE_TEXT>& S, u32* keep_out, u64* bnd_out,
                                               Chunk<CharT>* pre = nullptr) {
  const u32 lane = lane_id();
  const int64_t org = tile_origin(tile);
  const bool inside = org >= 0 && (u64)org + TK_SPAN <= P.n_chars;   // wave-uniform
  const bool general_pages = P.uniform_page_chars == 0;
  // Explicit page offsets: the page starts of the lane's 64-character window come straight from a BITMAP of the text (one bit
  // per character, built once per build): three dword loads at a fixed place, issued with the text fetch.  (Rounds 1-2 searched
  // the offset array per wave tile -- log64 rounds of dependent loads -- and OR-ed the starts into an LDS mask with atomics: a
  // build with explicit offsets took 5.5 ms against 3.9 ms with arithmetic uniform pages.)
  const int64_t g0 = org + (int64_t)lane * TK_CPT;
  u64 pgw = 0;
  if (general_pages && g0 >= 0) {
    const u64 gb = (u64)g0;                                   // a multiple of 16
    const u32* w = P.page_bits + (gb >> 5);
    const u32 w0 = w[0], w1 = w[1], w2 = w[2];
    pgw = (((u64)w1 << 32) | w0) >> (gb & 31);
    if (gb & 31) pgw |= (u64)w2 << 48;
  }
  u32 Lc, Dc;
  {
    Chunk<CharT> c;
    if (pre) c = *pre; else c.fetch(text, g0, P.n_chars, inside);
    c.classify(tab);
    Lc = c.L; Dc = c.D;
    if (STAGE_TEXT) {
#pragma unroll
      for (int k = 0; k < Chunk<CharT>::NW; k++) S.txt[lane * Chunk<CharT>::NW + k] = c.w[k];
    }
  }
  if (STAGE_TEXT) wave_sync();
  // neighbour masks by shuffle: chunks lane+1 .. lane+3 (look-ahead) and the last bit of chunk lane-1
  const u32 L1 = __shfl_down(Lc, 1, 64), L2 = __shfl_down(Lc, 2, 64), L3 = __shfl_down(Lc, 3, 64);
  const u32 D1 = __shfl_down(Dc, 1, 64), D2 = __shfl_down(Dc, 2, 64), D3 = __shfl_down(Dc, 3, 64);
  const u32 Lp = __shfl_up(Lc, 1, 64), Dp = __shfl_up(Dc, 1, 64);
  u32 keep = 0;
  u64 bnd = 0;
  if (lane >= TK_PRE && lane < TK_PRE + TK_OWN) {
    const u64 L64 = (u64)Lc | ((u64)L1 << 16) | ((u64)L2 << 32) | ((u64)L3 << 48);
    const u64 D64 = (u64)Dc | ((u64)D1 << 16) | ((u64)D2 << 32) | ((u64)D3 << 48);
    const u64 pL = (Lp >> 15) & 1u, pD = (Dp >> 15) & 1u;
    bnd = (L64 ^ ((L64 << 1) | pL)) | (D64 ^ ((D64 << 1) | pD));
    if (general_pages) {
      bnd |= pgw;
    } else {
      // segment < 2^32 characters: 32-bit arithmetic; power-of-two page sizes need no division
      const u32 Sz = P.uniform_page_chars;
      const u32 g32 = (u32)g0;   // g0 >= 0 for the owning lanes
      u32 r = (Sz & (Sz - 1u)) == 0u ? (g32 & (Sz - 1u)) : (g32 % Sz);
      for (u32 o = r ? Sz - r : 0u; o < 64u; o += Sz) bnd |= 1ull << o;
    }
    // A token that starts at bit j ends at the next boundary behind j; it is kept if its length lies in [min_len, max_len]:
    // no boundary among the bits j+1 .. j+min_len-1 (SHORT = OR of bnd >> 1 .. min_len-1) and one among j+1 .. j+max_len
    // (ENDS = OR of bnd >> 1 .. max_len) -- for all sixteen starts of the chunk at once, the ORs by doubling (wave-uniform
    // trip counts) instead of one ctz / shift / compare round per start of the lane with the most starts.
    // (j <= 15 and max_len <= 32: every bit looked at lies inside the 64-bit window)
    auto or_shifts = [](u64 b, u32 R) -> u64 {     // OR of b >> 1 .. b >> R (0 for R == 0)
      if (R == 0) return 0ull;
      u64 x = b >> 1;
      for (u32 c = 1; c < R;) { const u32 sft = c < R - c ? c : R - c; x |= x >> sft; c += sft; }
      return x;
    };
    const u32 starts = (u32)(bnd & (L64 | D64)) & 0xFFFFu;
    keep = starts & ~(u32)or_shifts(bnd, P.min_len - 1u) & (u32)or_shifts(bnd, P.max_len);
  }
  *keep_out = keep;
  *bnd_out = bnd;
}

}  // namespace dcd

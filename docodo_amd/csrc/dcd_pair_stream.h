// dcd_pair_stream.h -- the streaming form of IndexSequence '*' / '+' (IndexSequence.cs:218-322) on 32-bit TILE-RELATIVE keys.
//
// Round 4.  k_pair_parts merges on 64-bit keys (coordinate << 1 | operand): every compare-exchange of its bitonic network is two
// LDS-crossbar shuffles, a 64-bit compare and two selects, its chain scans are log-step shuffles, and it shares one register
// budget with the galloping path (106 SGPRs + 135 spilled).  The kernels of this file evaluate the SAME work items -- partitions
// of the operand union cut at fixed diagonals, a wavefront per item, chains owned by the partition they start in, exactly the
// rules of k_pair_parts -- with:
//   * keys (coordinate - base) << 1 | operand in 32 bits, exact by construction: an element too far beyond the tile's head gets
//     the key +inf, i.e. counts as not staged yet -- the tile then consumes fewer than 256 elements (at least the head itself) and
//     the next tile starts from a new head.  No 64-bit fallback path exists or is needed;
//   * the bitonic merge as ONE v_med3_u32 per compare-exchange (med3(x, y, 0) = min, med3(x, y, ~0) = max: the third operand is
//     the lane's side of the stage), partners fetched by DPP (row_ror:8, quad_perm, row_shl/shr:4) and by the gfx950 lane swaps
//     v_permlane32_swap / v_permlane16_swap -- no LDS crossbar, no address registers; 61 instructions per 256-key tile;
//   * the cross-lane part of the chain logic -- "does the chain hold a contact", "a left element at or before me" -- as a
//     segmented OR over 64 lanes, which is the carry chain of ONE 64-bit scalar addition (generate = the lanes' flags, propagate =
//     lanes without a break); positions come from s_ff1 / s_flbit, counts from s_bcnt1.
// Two forms of the tile logic:
//   k_pair_stream_lanes  (first half of round 4; DOCODO_QUERY_STREAM_FORM=lanes)  per-lane 0 / 1 flags for the lane's four elements,
//                        ballots only for the cross-lane part; 64-bit coordinates in the LDS rings (four padded planes).
//   k_pair_stream<WPE>   (the default)  every per-element flag as a 64-bit LANE MASK per element slot, the whole chain logic on the
//                        scalar unit; rings of 32-bit keys relative to an epoch of the work item (see below).  40 % fewer vector
//                        instructions per tile (profiles/r04_pmc_c4.txt), the same scalar count.
// Results are bit-identical to k_pair_parts (tests/test_gpu_query.py, stress_ops.py, stress_queries.py run all paths;
// tests/helpers/pair_stream_model.py holds a Python model of either form, checked against the oracle on the CPU).
#pragma once

namespace dcd {

#ifndef DCD_SK_WAVES
#define DCD_SK_WAVES 8
#endif
constexpr u32 SK_TILE = 256;
constexpr u32 SK_PLANE = 64 + 4;                 // elements per ring plane (padded: the refill's 8-byte stores hit distinct banks)
constexpr u32 SK_RING = 4 * SK_PLANE;
constexpr u32 SK_INF = 0xFFFFFFFFu;
constexpr u64 SK_LIM = 0x7FFFFFFFull;            // relative coordinates below this are representable (key <= 0xFFFFFFFD)

__device__ __forceinline__ u32 sk_phys(u32 j) { return (j & 3u) * SK_PLANE + (j >> 2); }   // ring slot j in [0, 256)
__device__ __forceinline__ u32 sk_med3(u32 a, u32 b, u32 c) {
  u32 r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// bit `lane` of a wave-uniform mask as 0 / 1 (one v_cndmask on the SGPR pair)
__device__ __forceinline__ u32 sk_lane_bit(u64 m) {
  u32 r;
  asm("v_cndmask_b32 %0, 0, 1, %1" : "=v"(r) : "s"(m));
  return r;
}
// One stage of the merge network = the compare-exchange of all FOUR elements of the lane with the lane whose number differs in one
// bit: x[k] = med3(x[k], partner's x[k], side), side = 0 in the lanes that keep the minimum, ~0 in the others.  Every stage is
// ONE asm block in which the four elements are interleaved: a DPP move must not read a register less than two wait states after
// a vector instruction wrote it (likewise the lane swaps), and the compiler cannot see that hazard between inline-asm statements
// -- a block of 4 moves + 4 v_med3 keeps every such pair at least three instructions apart, with no s_nop.
// (first block: the keys come straight from compiler-generated selects, hence the leading s_nop 1)
#define SK_OPS4(x, t) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
#define SK_MED3_4 "v_med3_u32 %0, %0, %4, %8\n v_med3_u32 %1, %1, %5, %8\n v_med3_u32 %2, %2, %6, %8\n v_med3_u32 %3, %3, %7, %8\n"
__device__ __forceinline__ void sk_stage32(u32 (&x)[4], u32 side) {
  u32 t[4];
  asm volatile("s_nop 1\n"
               "v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3\n"
               // v_permlane32_swap a, b: a = {a[0..31], b[0..31]}, b = {a[32..63], b[32..63]} -- with a == b == x: the low half of x
               // in both halves of a, the high half in both halves of b: min belongs to lanes 0..31, max to lanes 32..63
               "v_permlane32_swap_b32 %0, %4\n v_permlane32_swap_b32 %1, %5\n v_permlane32_swap_b32 %2, %6\n v_permlane32_swap_b32 %3, %7\n"
               SK_MED3_4 : SK_OPS4(x, t) : "v"(side));
}
__device__ __forceinline__ void sk_stage16(u32 (&x)[4], u32 side) {
  u32 t[4];
  asm volatile("v_mov_b32 %4, %0\n v_mov_b32 %5, %1\n v_mov_b32 %6, %2\n v_mov_b32 %7, %3\n"
               // v_permlane16_swap a, b: odd rows of a <-> even rows of b: a = rows {0, 0, 2, 2} of x, b = rows {1, 1, 3, 3}
               "v_permlane16_swap_b32 %0, %4\n v_permlane16_swap_b32 %1, %5\n v_permlane16_swap_b32 %2, %6\n v_permlane16_swap_b32 %3, %7\n"
               SK_MED3_4 : SK_OPS4(x, t) : "v"(side));
}
#define SK_DPP4(ctrl) "v_mov_b32_dpp %4, %0 " ctrl "\n v_mov_b32_dpp %5, %1 " ctrl "\n v_mov_b32_dpp %6, %2 " ctrl "\n v_mov_b32_dpp %7, %3 " ctrl "\n"
__device__ __forceinline__ void sk_stage8(u32 (&x)[4], u32 side) {      // row_ror:8 = lane ^ 8
  u32 t[4];
  asm volatile(SK_DPP4("row_ror:8 row_mask:0xf bank_mask:0xf") SK_MED3_4 : SK_OPS4(x, t) : "v"(side));
}
__device__ __forceinline__ void sk_stage4(u32 (&x)[4], u32 side) {      // banks 0, 2 read lane + 4 (row_shl:4), banks 1, 3 lane - 4 (row_shr:4)
  u32 t[4];
  asm volatile(SK_DPP4("row_shl:4 row_mask:0xf bank_mask:0x5") SK_DPP4("row_shr:4 row_mask:0xf bank_mask:0xa") SK_MED3_4
               : SK_OPS4(x, t) : "v"(side));
}
__device__ __forceinline__ void sk_stage2(u32 (&x)[4], u32 side) {      // quad_perm [2,3,0,1] = lane ^ 2
  u32 t[4];
  asm volatile(SK_DPP4("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") SK_MED3_4 : SK_OPS4(x, t) : "v"(side));
}
__device__ __forceinline__ void sk_stage1(u32 (&x)[4], u32 side) {      // quad_perm [1,0,3,2] = lane ^ 1
  u32 t[4];
  asm volatile(SK_DPP4("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") SK_MED3_4 : SK_OPS4(x, t) : "v"(side));
}
// bitonic sequence of 256 keys (element t = 4 * lane + k) -> ascending: distances 128 .. 4 across lanes, 2 and 1 inside the lane
__device__ __forceinline__ void sk_merge_network(u32 (&x)[4], u32 lane) {
  sk_stage32(x, (u32)((int)(lane << 26) >> 31));
  sk_stage16(x, (u32)((int)(lane << 27) >> 31));
  sk_stage8(x, (u32)((int)(lane << 28) >> 31));
  sk_stage4(x, (u32)((int)(lane << 29) >> 31));
  sk_stage2(x, (u32)((int)(lane << 30) >> 31));
  sk_stage1(x, (u32)((int)(lane << 31) >> 31));
  const u32 t0 = min(x[0], x[2]), t2 = max(x[0], x[2]), t1 = min(x[1], x[3]), t3 = max(x[1], x[3]);
  x[0] = min(t0, t1); x[1] = max(t0, t1); x[2] = min(t2, t3); x[3] = max(t2, t3);
}
// debug / tests: the network on 256 keys in memory
__global__ void k_sk_network_probe(u32* __restrict__ keys) {
  const u32 lane = threadIdx.x;
  u32 x[4];
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = keys[lane * 4u + (u32)k];
  sk_merge_network(x, lane);
#pragma unroll
  for (int k = 0; k < 4; k++) keys[lane * 4u + (u32)k] = x[k];
}

// filled[i] = g[i] | (p[i] & filled[i - 1]), filled[-1] = cin: the carry chain of (g | p) + g + cin
__device__ __forceinline__ u64 sk_fill_fwd(u64 g, u64 p, u32 cin) {
  const u64 A = g | p, B = g;
  const u64 S = A + B + (u64)cin;
  const u64 c = S ^ A ^ B;                                          // c[i] = carry into bit i
  const u64 cout = ((A & B) | ((A | B) & ~S)) >> 63;                // carry out of bit 63
  return (c >> 1) | (cout << 63);
}
// filled[i] = g[i] | (p[i] & filled[i + 1])
__device__ __forceinline__ u64 sk_fill_bwd(u64 g, u64 p) { return __brevll(sk_fill_fwd(__brevll(g), __brevll(p), 0u)); }

__device__ __forceinline__ u64 sk_split(const u64* __restrict__ A, u64 na, const u64* __restrict__ B, u64 nb, u64 d) {
  u64 lo = d > nb ? d - nb : 0, hi = d < na ? d : na;
  const u32 lane = lane_id();
  while (lo < hi) {
    const u64 step = (hi - lo + 63) >> 6;
    const u64 p = lo + (u64)lane * step + (step - 1);
    bool pr = false;
    if (p < hi) pr = A[p] <= B[d - 1 - p];
    const u32 c = popc64(ballot64(pr));          // the predicate is monotone: the true lanes form a prefix
    const u64 nlo = lo + (u64)c * step;
    const u64 nhi = nlo + step - 1;
    lo = nlo;
    hi = nhi < hi ? nhi : hi;
  }
  return lo;
}

// kind_base: exclusive scan of the STREAMING work items per operator (this kernel's items); part_base: of all items (where an
// item's count and scratch slot are recorded).  idx_v64: the index's resident coordinates (never the packed planes).
__global__ __launch_bounds__(64, DCD_SK_WAVES) void k_pair_stream_lanes(const PairOp* __restrict__ ops, u32 n_ops, const u64* __restrict__ kind_base,
                                                                  const u64* __restrict__ part_base, const u64* __restrict__ idx_v64, Pool pool,
                                                                  u64* __restrict__ tmp, u64* __restrict__ part_count, u64* __restrict__ part_src,
                                                                  u64 PART, unsigned long long* __restrict__ work_ctr, const u64* __restrict__ tmp_off) {
  __shared__ __attribute__((aligned(16))) u64 s_ring[2 * SK_RING];
  u64* sA = s_ring;
  u64* sB = s_ring + SK_RING;
  const u32 lane = threadIdx.x;
  const u64 n_items = kind_base[n_ops];
  u64 w = blockIdx.x;
  while (w < n_items) {
    unsigned long long w_next = 0;
    if (lane == 0) w_next = atomicAdd(work_ctr, 1ull) + gridDim.x;
    u32 oi;
    {   // operator of this work item: last op with kind_base[op] <= w
      u32 lo = 0, hi = n_ops;
      while (lo + 1 < hi) { u32 m = (lo + hi) >> 1; if (kind_base[m] <= w) lo = m; else hi = m; }
      oi = lo;
    }
    const u64 p = w - kind_base[oi];
    const u64 wi = part_base[oi] + p;
    const PairOp op = ops[oi];
    const u64* __restrict__ A = op.a.kind ? pool.base + pool.res_off[op.a.cnt_ref] : idx_v64 + op.a.off;
    const u64* __restrict__ B = op.b.kind ? pool.base + pool.res_off[op.b.cnt_ref] : idx_v64 + op.b.off;
    const u64 na = seg_cnt(op.a, pool), nb = seg_cnt(op.b, pool);
    u64* tmp_op = tmp + tmp_off[oi];
    const u32 absR = (u32)max(abs(op.Ra), abs(op.Rb));
    const bool ordered = op.Ra < 0 && op.Rb < 0 && absR != 0;   // R = -absR < 0  (IndexSequence.cs:221-222)
    const bool is_or = op.op == DCD_OP_OR;
    const u32 thr = absR != 0 ? absR : 0xFFFFFFFFu;             // absR == 0: nothing ever breaks a chain
    const u64 d_begin = p * PART, d_end = d_begin + PART;
    u64 ia = rfl64(sk_split(A, na, B, nb, d_begin)), ib = d_begin - ia;
    u64 prev_val = 0;
    u32 prev_src = 0, have_prev = 0;
    if (d_begin > 0) {
      have_prev = 1;
      const bool ua = ia > 0, ub = ib > 0;
      const u64 pa = ua ? A[ia - 1] : 0, pb = ub ? B[ib - 1] : 0;
      if (ua && ub) { if (pb >= pa) { prev_val = pb; prev_src = 1; } else { prev_val = pa; prev_src = 0; } }
      else if (ua) { prev_val = pa; prev_src = 0; }
      else { prev_val = pb; prev_src = 1; }
      prev_val = rfl64(prev_val); prev_src = rfl(prev_src);
    }
    u64 g = d_begin;                 // union index of the next unconsumed element
    u64 cursor = 0, out_start = 0;
    u32 open = 0, committed = 0, seenA = 0;
    u32 hA = 0, hB = 0;
    bool skipping = !is_or;          // '*': nothing is owned before the first chain break of the partition
    u64 slot = d_begin;
    u64* out = tmp_op + slot;
    u32 ra = (u32)min((u64)SK_TILE, na - ia), rb = (u32)min((u64)SK_TILE, nb - ib);
    for (u32 i = lane; i < ra; i += 64) sA[sk_phys(i)] = A[ia + i];
    for (u32 i = lane; i < rb; i += 64) sB[sk_phys(i)] = B[ib + i];
    wsync();
    while (ra + rb != 0) {
      if (g >= d_end && (skipping || is_or)) break;
      // ---- keys relative to the smallest staged coordinate --------------------------------------------------------
      u64 base;
      {
        const u64 ha = ra ? sA[sk_phys(hA)] : ~0ull, hb = rb ? sB[sk_phys(hB)] : ~0ull;     // (one address: every lane gets the value)
        base = rfl64(ha < hb ? ha : hb);
      }
      u32 x[4];
      u32 ca = 0, cb = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 t = lane * 4u + (u32)k, u = SK_TILE - 1u - t;
        const u64 da = sA[sk_phys((hA + t) & (SK_TILE - 1u))] - base, db = sB[sk_phys((hB + u) & (SK_TILE - 1u))] - base;
        const u32 ka = (t < ra && da < SK_LIM) ? ((u32)da << 1) : SK_INF;
        const u32 kb = (u < rb && db < SK_LIM) ? (((u32)db << 1) | 1u) : SK_INF;
        const bool ta = ka <= kb;                              // (equal only when both are +inf)
        x[k] = ta ? ka : kb;
        ca += popc64(ballot64(ta && ka != SK_INF));
        cb += popc64(ballot64(!ta && kb != SK_INF));
      }
      ca = rfl(ca); cb = rfl(cb);
      const u32 cnt = ca + cb;                                 // >= 1: the base element itself
      // ---- request the refill (one stream of four loads per lane covers both operands) ---------------------------
      const u32 keepA = ra - ca, keepB = rb - cb;
      const u32 nra = (u32)min((u64)SK_TILE, na - ia - ca), nrb = (u32)min((u64)SK_TILE, nb - ib - cb);
      const u32 newA = nra - keepA, newB = nrb - keepB;        // newA + newB <= 256: the tile frees ca + cb slots
      u64 pf[4];
      {
        const u64 offA = ia + ra, offB = ib + rb - newA;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u32 xx = lane + (u32)q * 64u;
          const bool fa = xx < newA, act = xx < newA + newB;
          const u64* src = fa ? A + offA : B + offB;
          pf[q] = act ? src[xx] : 0ull;
        }
      }
      // ---- bitonic merge: distances 128 .. 4 across lanes, 2 and 1 inside the lane (element t = 4 * lane + k) -------
      sk_merge_network(x, lane);
      // ---- neighbours: gap to the predecessor, chain breaks, duplicates, cross-operand contacts -------------------
      // (the element before the tile is the carried one: its distance to the base saturates far above any absR)
      u32 pgap = 0xFFFFFFFFu;
      if (have_prev) { const u64 dd = base - prev_val; pgap = dd < 0xFFFFFFFFull ? (u32)dd : 0xFFFFFFFFu; }
      u32 brk[4], dup[4], ctc[4], al[4], src[4];
      {
        u32 pkey = (u32)__builtin_amdgcn_update_dpp(0, (int)x[3], 0x138, 0xF, 0xF, false);     // wave_shr:1: x[3] of the lane below
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const bool valid = x[k] != SK_INF;                   // (the cnt valid keys are the first cnt elements)
          const u32 sr = x[k] & 1u;
          u32 gap = (x[k] >> 1) - (pkey >> 1), psrc = pkey & 1u, hp = 1u;
          if (k == 0) {
            const bool l0 = lane == 0;
            gap = l0 ? pgap : gap; psrc = l0 ? prev_src : psrc; hp = l0 ? have_prev : 1u;
          }
          const bool bk = !hp || gap > thr;
          src[k] = valid ? sr : 0u;
          al[k] = valid ? 1u : 0u;
          brk[k] = (valid && bk) ? 1u : 0u;
          dup[k] = (valid && sr == 1u && hp && psrc == 0u && gap == 0u) ? 1u : 0u;
          ctc[k] = (valid && hp && psrc != sr && !bk) ? 1u : 0u;   // neighbours from different operands within absR
          pkey = x[k];
        }
      }
      // the rings have been read: store the refill (visible to the next tile after the wsync that ends this one)
      wsync();
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32 xx = lane + (u32)q * 64u;
        if (xx < newA) sA[sk_phys((hA + ra + xx) & (SK_TILE - 1u))] = pf[q];
        else if (xx < newA + newB) sB[sk_phys((hB + rb + xx - newA) & (SK_TILE - 1u))] = pf[q];
      }
      // the tile's last element
      const u32 last_lane = (cnt - 1u) >> 2, last_k = (cnt - 1u) & 3u;
      u32 lkey = x[0];
#pragma unroll
      for (int k = 1; k < 4; k++) lkey = (u32)k == last_k ? x[k] : lkey;
      lkey = rdl(lkey, last_lane);
      const u64 t_last_val = base + (u64)(lkey >> 1);
      const u32 t_last_src = lkey & 1u;
      const bool inputs_end = (ia + ca == na) && (ib + cb == nb);
      // ---- ownership: [t_lo, t_hi) of this tile belongs to the work item ------------------------------------------
      u32 t_lo = 0, t_hi = cnt;
      const bool may_end = g + cnt > d_end;
      if (skipping || may_end) {
        const u32 lim = may_end ? (d_end > g ? (u32)(d_end - g) : 0u) : 0xFFFFFFFFu;
        u32 fbk = 4, fek = 4;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
          const bool c = al[k] && (is_or || brk[k]);
          if (c) fbk = (u32)k;
          if (c && lane * 4u + (u32)k >= lim) fek = (u32)k;
        }
        const u64 Bf = ballot64(fbk < 4u), Ef = ballot64(fek < 4u);
        u32 fb = NONE32, fe = NONE32;
        if (Bf) { const u32 l = (u32)__builtin_ctzll(Bf); fb = l * 4u + rdl(fbk, l); }
        if (Ef) { const u32 l = (u32)__builtin_ctzll(Ef); fe = l * 4u + rdl(fek, l); }
        if (skipping) t_lo = min(fb, cnt);
        if (may_end) t_hi = min(fe, cnt);
        if (t_hi < t_lo) t_hi = t_lo;
      }
      const bool ends_here = t_hi < cnt;       // a chain break at or beyond the partition end: stop before it
      auto advance = [&]() {
        ia += ca; ib += cb;
        hA = (hA + ca) & (SK_TILE - 1u); hB = (hB + cb) & (SK_TILE - 1u);
        ra = nra; rb = nrb;
        prev_val = t_last_val; prev_src = t_last_src; have_prev = 1;
        g += cnt;
      };
      if (t_lo >= t_hi) {                      // nothing owned in this tile
        if (ends_here) {
          if (open && !committed) cursor = out_start;   // the carried chain closes without the other operand
          open = 0;
          break;
        }
        advance();
        wsync();
        continue;
      }
      if (skipping) { skipping = false; slot = g + t_lo; out = tmp_op + slot; }
      if (t_lo > 0 || ends_here) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32 t = lane * 4u + (u32)k;
          const bool a = t >= t_lo && t < t_hi;
          if (!a) { al[k] = 0; brk[k] = 0; ctc[k] = 0; }
        }
      }
      const bool final_tile = ends_here || inputs_end;
      // ---- chains ----------------------------------------------------------------------------------------------------
      // A chain is emitted iff it holds both operands, i.e. iff two NEIGHBOURS of it come from different operands (ordered
      // mode: an element of the right operand directly after one of the left operand -- "a right element at or after the
      // first left element"), and ordered mode emits only from the first left element of the chain on.
      // In the lane: running ORs since the lane's last break (forward) and up to its next break (backward).
      u32 fx[4], fa[4], bx[4], head[4], tailp[4];
      u32 any_brk = 0, tail_x, tail_a, head_x, last_bk = 0;
      {
        u32 ax = 0, aa = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u32 xq = (ctc[k] && (!ordered || src[k] == 1u)) ? 1u : 0u;     // contact with the predecessor (never on a break)
          const u32 aq = (al[k] && src[k] == 0u) ? 1u : 0u;
          if (brk[k]) { ax = 0; aa = 0; any_brk = 1; last_bk = (u32)k; }
          ax |= xq; aa |= aq;
          fx[k] = ax; fa[k] = aa;
          head[k] = any_brk ^ 1u;                                              // no break at or before k in this lane
          bx[k] = xq;
        }
        tail_x = ax; tail_a = aa;
        u32 b = 0, later_brk = 0;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
          b |= bx[k];
          bx[k] = b;                                                           // OR from k to the end of its run inside the lane
          tailp[k] = later_brk ^ 1u;                                           // no break behind k in this lane
          if (brk[k]) { b = 0; later_brk = 1; }
        }
        head_x = b;                                                            // OR over the elements before the lane's first break
      }
      // Across the lanes: a chain runs from the tail part of a lane with a break through the lanes without one into the head
      // part of the next lane with a break -- segmented ORs over 64 lanes = carry chains of one scalar addition each.
      const u64 Bm = ballot64(any_brk != 0u);
      const u64 Pm = ~Bm;
      const u64 Fx = sk_fill_fwd(ballot64(tail_x != 0u), Pm, committed);       // chain leaving lane l holds a contact (so far)
      const u64 Gx = sk_fill_bwd(ballot64(head_x != 0u), Pm);                  // chain entering lane l holds one further on
      const u32 in_x = sk_lane_bit((Fx << 1) | (u64)committed), out_x = sk_lane_bit(Gx >> 1);
      u32 in_a = 1u;
      u64 Fa = 0;
      if (ordered) {
        Fa = sk_fill_fwd(ballot64(tail_a != 0u), Pm, seenA);
        in_a = sk_lane_bit((Fa << 1) | (u64)seenA);
      }
      const bool has_brk = Bm != 0ull;                                         // a chain starts in this tile
      u32 LB = 0;                                                              // position of the tile's last break
      if (has_brk) { const u32 l = 63u - (u32)__builtin_clzll(Bm); LB = l * 4u + rdl(last_bk, l); }
      const bool chain0_commit = committed || ((Gx & 1ull) != 0ull);           // (Gx bit 0: a contact before the tile's first break)
      if (!is_or && open && !committed && (has_brk || final_tile) && !chain0_commit) cursor = out_start;   // roll back
      u32 emit[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const u32 t = lane * 4u + (u32)k;
        u32 e = 0;
        if (al[k] && !dup[k]) {
          if (is_or) e = 1;
          else {
            const u32 cm = fx[k] | bx[k] | (head[k] & in_x) | (tailp[k] & out_x);
            const u32 sa = ordered ? (fa[k] | (head[k] & in_a)) : 1u;
            const bool closed = final_tile || (has_brk && t < LB);
            e = (sa && (closed ? cm : 1u)) ? 1u : 0u;
          }
        }
        emit[k] = e;
      }
      // compaction: position = emitted elements before (lanes below: v_mbcnt over the four ballots; in the lane: running sum)
      u64 Em[4];
      u32 pos = 0, total_emit = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) { Em[k] = ballot64(emit[k] != 0u); pos += mbcnt(Em[k]); total_emit += popc64(Em[k]); }
      {
        u32 e = pos;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (emit[k]) out[cursor + e] = base + (u64)(x[k] >> 1);
          e += emit[k];
        }
      }
      {
        // the chain that is open at the end of the tile: where its output starts, whether it holds a contact / a left element
        u32 before_lb = 0;                                                     // emitted elements in front of the last break
        if (has_brk) {
          const u32 ll = LB >> 2, lk = LB & 3u;
          const u64 below = (1ull << ll) - 1ull, incl = below | (1ull << ll);
#pragma unroll
          for (int k = 0; k < 4; k++) before_lb += popc64(Em[k] & ((u32)k < lk ? incl : below));
        }
        const bool com = (Fx >> 63) != 0ull;
        const bool hasA = (Fa >> 63) != 0ull;                                  // (only ordered mode looks at it)
        if (has_brk || !open) out_start = cursor + before_lb;
        open = 1; seenA = hasA ? 1u : 0u; committed = com ? 1u : 0u;
        cursor += total_emit;
      }
      if (ends_here) { open = 0; break; }
      advance();
      wsync();
    }
    if (lane == 0) { part_count[wi] = cursor; part_src[wi] = slot | ((u64)oi << PSRC_OP_SHIFT); }
    wsync();                                                     // (the next item reuses the rings)
    w = rfl64(w_next);
  }
}


// ------------------------------------------------------------------------------------------------------------------------
// Mask form (round 4, second half).  k_pair_stream_lanes above spends two thirds of its ~700 vector instructions per tile on
// per-lane 0 / 1 flags (selects, compares, 64-bit key arithmetic) and on scalar registers spilled to vector lanes; configs[3]
// and configs[4] are bound by exactly that: vector-instruction issue (profiles/r04_pmc_c4.txt).  This form keeps
//   * every per-element flag as a 64-bit LANE MASK per element slot k (element t = 4 * lane + k: four masks cover the tile).  A
//     v_cmp writes such a mask straight into a scalar register pair; the chain logic -- running ORs inside the lane, heads and
//     tails, the two carry-chain fills across the lanes, ownership windows, the emit decision -- is then scalar bit arithmetic,
//     ONE instruction for all 64 lanes (the scalar unit issues beside the vector unit and was idle); a mask gets back into the
//     lanes as an exec mask or a carry-in, not through a select;
//   * the rings as 32-bit keys (coordinate - E) << 1 | operand relative to an EPOCH coordinate E of the work item, saturated
//     at 2^31 - 1 and mirrored (slot j also at j + 256: a lane's four consecutive elements are at one address + constant
//     offsets, no wrap-around arithmetic).  A tile's key is the ring value minus twice the head's distance from E -- one
//     subtraction; keys at or above 2^31 (an element 2^30 coordinates or more beyond the tile's head, a saturated one, the
//     padding behind a list's end) count as not staged yet, exactly as in the lane form.  When the head itself comes within
//     2^30 of the window's end, E moves to the head and both rings are staged again.
// Work items, ownership and roll-back rules and results are those of the lane form (tests/helpers/pair_stream_model.py holds both
// models; tests/test_pair_stream_model.py checks them against the oracle without a GPU).
// ------------------------------------------------------------------------------------------------------------------------
constexpr u32 SKM_SLOTS = 256;
constexpr u32 SKM_RING = 2 * SKM_SLOTS;          // mirrored
constexpr u64 SKM_SAT = 0x7FFFFFFFull;           // ring keys: coordinate - E below this is representable
constexpr u32 SKM_INF = 0xFFFFFFFEu;             // saturated / padding key of operand 0 (operand 1: | 1)

// ring key of a coordinate >= E; Elim = E + SKM_SAT
__device__ __forceinline__ u32 skm_enc(u64 coord, u64 E, u64 Elim, u32 src) {
  return coord < Elim ? ((((u32)coord - (u32)E) << 1) | src) : (SKM_INF | src);
}
// lanes l with 4 * l + k < v
__device__ __forceinline__ u64 skm_lt(u32 v, u32 k) {
  const u32 n = (min(v, 256u) + 3u - k) >> 2;
  return n >= 64u ? ~0ull : ((1ull << n) - 1ull);
}
// v + (bit `lane` of the wave-uniform mask m): the mask is the carry-in of one add
__device__ __forceinline__ u32 skm_add_bit(u32 v, u64 m) {
  u32 r;
  u64 co;
  asm("v_addc_co_u32 %0, %1, %2, 0, %3" : "=v"(r), "=s"(co) : "v"(v), "s"(m));
  return r;
}
__device__ __forceinline__ bool skm_bit(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
template <int V> struct SkmMode { static constexpr int value = V; };      // 0: '*', 1: '*' ordered (R < 0), 2: '+'

// hides where a scalar comes from: the halves of a 64-bit scalar compared on their own stay 32-bit SCALAR compares (put back
// together the compiler issues a 64-bit vector compare: the scalar unit has none)
__device__ __forceinline__ u32 skm_opaque(u32 v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ bool skm_gt(u64 v, u32 k) { return skm_opaque((u32)(v >> 32)) != 0u || (u32)v > k; }

// PART < 2^32 (the host checks).  Scalar state of the tile loop, kept small on purpose -- the lane masks need the registers:
// pA / pB point at the operands' next unconsumed element, remA / remB count what is left of them; `left` = union elements up to
// the partition's end (0 once it is passed); `slot` follows the union index until the first owned element is found.
// WPE: wavefronts per SIMD the kernel is compiled for = its scalar-register budget (8: 80 registers, 7: 96, 6 and fewer: 102)
template <int WPE>
__global__ __launch_bounds__(64, WPE) void k_pair_stream(const PairOp* __restrict__ ops, u32 n_ops, const u64* __restrict__ kind_base,
                                                         const u64* __restrict__ part_base, const u64* __restrict__ idx_v64, Pool pool,
                                                         u64* __restrict__ tmp, u64* __restrict__ part_count, u64* __restrict__ part_src,
                                                         u64 PART, unsigned long long* __restrict__ work_ctr, const u64* __restrict__ tmp_off) {
  __shared__ __attribute__((aligned(16))) u32 s_ring[2 * SKM_RING];
  u32* sA = s_ring;
  u32* sB = s_ring + SKM_RING;
  const u32 lane = threadIdx.x;
  const u64 n_items = kind_base[n_ops];
  u64 w = blockIdx.x;
  while (w < n_items) {
    unsigned long long w_next = 0;
    if (lane == 0) w_next = atomicAdd(work_ctr, 1ull) + gridDim.x;
    u32 oi;
    {   // operator of this work item: last op with kind_base[op] <= w
      u32 lo = 0, hi = n_ops;
      while (lo + 1 < hi) { u32 m = (lo + hi) >> 1; if (kind_base[m] <= w) lo = m; else hi = m; }
      oi = lo;
    }
    const u64 p = w - kind_base[oi];
    const u64 wi = part_base[oi] + p;
    const PairOp op = ops[oi];
    const u64* __restrict__ A = op.a.kind ? pool.base + pool.res_off[op.a.cnt_ref] : idx_v64 + op.a.off;
    const u64* __restrict__ B = op.b.kind ? pool.base + pool.res_off[op.b.cnt_ref] : idx_v64 + op.b.off;
    const u64 na = seg_cnt(op.a, pool), nb = seg_cnt(op.b, pool);
    u64* tmp_op = tmp + tmp_off[oi];
    const u32 absR = (u32)max(abs(op.Ra), abs(op.Rb));
    const bool ordered = op.Ra < 0 && op.Rb < 0 && absR != 0;   // R = -absR < 0  (IndexSequence.cs:221-222)
    const bool is_or = op.op == DCD_OP_OR;
    const u32 thr = absR != 0 ? absR : 0xFFFFFFFFu;             // absR == 0: nothing ever breaks a chain
    const u64 d_begin = p * PART;
    u64 prev_val = 0;
    u32 prev_src = 0, have_prev = 0;
    const u64* __restrict__ pA;
    const u64* __restrict__ pB;
    u64 remA, remB;
    {
      const u64 ia = rfl64(sk_split(A, na, B, nb, d_begin)), ib = d_begin - ia;
      if (d_begin > 0) {
        have_prev = 1;
        const bool ua = ia > 0, ub = ib > 0;
        const u64 pa = ua ? A[ia - 1] : 0, pb = ub ? B[ib - 1] : 0;
        if (ua && ub) { if (pb >= pa) { prev_val = pb; prev_src = 1; } else { prev_val = pa; prev_src = 0; } }
        else if (ua) { prev_val = pa; prev_src = 0; }
        else { prev_val = pb; prev_src = 1; }
        prev_val = rfl64(prev_val); prev_src = rfl(prev_src);
      }
      pA = A + ia; pB = B + ib;             // (no detour through integers: the pointers must stay GLOBAL ones -- flat loads otherwise)
      remA = rfl64(na - ia); remB = rfl64(nb - ib);
    }
    u64 cursor = 0, slot = d_begin;
    // The tile loop, compiled once per operator kind: the masks an ordered '*' needs on top of a plain one (left elements seen) and
    // everything a '+' does not need (chains) would otherwise cost scalar registers in all three.
    auto run = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    u32 left = (u32)PART;            // union elements up to the partition's end
    u64 out_start = 0;
    u32 open = 0, committed = 0, seenA = 0;
    u32 hA = 0, hB = 0;              // ring slot of the operands' next unconsumed element
    bool skipping = MODE != 2;       // '*': nothing is owned before the first chain break of the partition
    u64 E = 0;
    // stages pA[0 ..], pB[0 ..] (256 slots each, padding behind the lists' ends) relative to a new epoch at the smaller head
    auto stage = [&]() {
      const u64 ha = remA ? pA[0] : ~0ull, hb = remB ? pB[0] : ~0ull;
      E = rfl64(ha < hb ? ha : hb);
      const u64 Elim = E + SKM_SAT;
      hA = 0; hB = 0;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32 t = lane + (u32)q * 64u;
        const u32 va = (u64)t < remA ? skm_enc(pA[t], E, Elim, 0u) : SKM_INF;
        const u32 vb = (u64)t < remB ? skm_enc(pB[t], E, Elim, 1u) : (SKM_INF | 1u);
        sA[t] = va; sA[t + SKM_SLOTS] = va;
        sB[t] = vb; sB[t + SKM_SLOTS] = vb;
      }
      wsync();
    };
    if ((remA | remB) != 0) stage();
    while ((remA | remB) != 0) {
      if (left == 0u && (skipping || MODE == 2)) break;
      // ---- the tile's head and keys ------------------------------------------------------------------------------------
      u32 hk = rfl(min(sA[hA], sB[hB]));                         // (one address each: every lane gets the value)
      if (hk >= 0x80000000u) {                                   // the head left the epoch's window (or is saturated)
        stage();
        hk = rfl(min(sA[0], sB[0]));
      }
      const u32 twob = hk & ~1u;
      const u64 base = E + (u64)(hk >> 1);
      u32 x[4];
      {
        const u32* pa = sA + hA + 4u * lane;                     // elements t = 4 * lane + k of A's staged run
        const u32* pb = sB + hB + 252u - 4u * lane;              // elements 255 - t of B's
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = min(pa[k] - twob, pb[3 - k] - twob);
      }
      // ---- bitonic merge: distances 128 .. 4 across lanes, 2 and 1 inside the lane (element t = 4 * lane + k) -------------
      sk_merge_network(x, lane);
      // ---- lane masks of the sorted tile: per element slot k, breaks / valid non-duplicates / contacts (/ left elements) ---
      u32 c[4];
#pragma unroll
      for (int k = 0; k < 4; k++) c[k] = x[k] >> 1;
      u64 BRK[4], ALND[4], XQ[4], AQ[4];
      u32 cnt = 0, ca = 0;
      {
        u32 pgap = 0xFFFFFFFFu;      // (the element before the tile is the carried one: its distance saturates far above any absR)
        if (have_prev) { const u64 dd = base - prev_val; pgap = skm_opaque((u32)(dd >> 32)) != 0u ? 0xFFFFFFFFu : (u32)dd; }
        u32 cp = (u32)__builtin_amdgcn_update_dpp(0, (int)c[3], 0x138, 0xF, 0xF, false);           // wave_shr:1: c[3] of the lane below
        asm volatile("" : "+v"(cp));     // (keeps the move a v_mov_b32_dpp: folded into the subtraction -- v_subrev_u32_dpp ... wave_shr:1 --
                                         //  the lanes got wrong neighbours on gfx950: every cross-lane gap was garbage)
        u32 gap[4];
        gap[0] = lane == 0u ? pgap : c[0] - cp;
#pragma unroll
        for (int k = 1; k < 4; k++) gap[k] = c[k] - c[k - 1];
        const u64 S3 = ballot64((x[3] & 1u) != 0u);
        u64 ps = (S3 << 1) | (u64)prev_src;                      // operand of the predecessor
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u64 V = ballot64((int)x[k] >= 0);                // (the valid keys are the first cnt elements)
          const u64 S = k == 3 ? S3 : ballot64((x[k] & 1u) != 0u);
          const u64 G = ballot64(gap[k] > thr);
          const u64 Zk = ballot64(gap[k] == 0u);
          const u64 HP = (k == 0 && !have_prev) ? ~1ull : ~0ull;
          cnt += popc64(V);                                      // >= 1: the head itself
          ca += popc64(V & ~S);
          BRK[k] = V & (G | ~HP);
          ALND[k] = V & ~(S & ~ps & Zk & HP);                    // valid and not the right operand's copy of the element before
          const u64 CT = V & HP & (ps ^ S) & ~G;                 // neighbours from different operands within absR (never a break)
          XQ[k] = MODE == 1 ? (CT & S) : CT;                     // (ordered mode: a right element directly after a left one)
          AQ[k] = V & ~S;
          ps = S;
          __builtin_amdgcn_sched_barrier(0);                     // (slot by slot: all sixteen compare results at once do not fit the scalar registers)
        }
      }
      const u32 cb = cnt - ca;
      // ---- request the refill: the ca / cb freed slots take the elements behind the 256 staged ones -------------------------
      u64 pfA[4], pfB[4];
      {
        const u32 loadA = skm_gt(remA, SKM_SLOTS + ca) ? ca : skm_gt(remA, SKM_SLOTS) ? (u32)remA - SKM_SLOTS : 0u;   // min(ca, remA - 256)
        const u32 loadB = skm_gt(remB, SKM_SLOTS + cb) ? cb : skm_gt(remB, SKM_SLOTS) ? (u32)remB - SKM_SLOTS : 0u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u32 xx = lane + (u32)q * 64u;
          pfA[q] = ~0ull; pfB[q] = ~0ull;                        // (behind a list's end: beyond every epoch window = padding)
          if ((u32)q * 64u < loadA) { if (xx < loadA) pfA[q] = pA[SKM_SLOTS + xx]; }
          if ((u32)q * 64u < loadB) { if (xx < loadB) pfB[q] = pB[SKM_SLOTS + xx]; }
        }
      }
      // the tile's last element
      u32 lkey;
      {
        const u32 last = cnt - 1u, last_lane = last >> 2, last_k = last & 3u;
        const u32 l0 = rdl(x[0], last_lane), l1 = rdl(x[1], last_lane), l2 = rdl(x[2], last_lane), l3 = rdl(x[3], last_lane);
        lkey = last_k == 0u ? l0 : last_k == 1u ? l1 : last_k == 2u ? l2 : l3;
      }
      const bool inputs_end = remA == (u64)ca && remB == (u64)cb;
      // ---- ownership: [t_lo, t_hi) of this tile belongs to the work item ------------------------------------------------------
      u32 t_lo = 0, t_hi = cnt;
      const bool may_end = cnt > left;
      if (skipping || may_end) {
        const u32 lim = may_end ? left : 0xFFFFFFFFu;
        if (MODE == 2) {                                         // '+': every element starts a chain of its own
          t_hi = min(lim, cnt);
        } else {
          u32 fb = NONE32, fe = NONE32;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const u64 C = BRK[k];
            if (C) fb = min(fb, 4u * (u32)__builtin_ctzll(C) + (u32)k);
            const u64 Ce = C & ~skm_lt(lim, (u32)k);
            if (Ce) fe = min(fe, 4u * (u32)__builtin_ctzll(Ce) + (u32)k);
          }
          if (skipping) t_lo = min(fb, cnt);
          if (may_end) t_hi = min(fe, cnt);
          if (t_hi < t_lo) t_hi = t_lo;
        }
      }
      const bool ends_here = t_hi < cnt;       // a chain break at or beyond the partition end: stop before it
      auto advance = [&]() {
        const u64 Elim = E + SKM_SAT;
        wsync();                               // (the rings were read at the top of the tile)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const u32 xx = lane + (u32)q * 64u;
          if ((u32)q * 64u < ca) {
            if (xx < ca) {
              const u32 v = skm_enc(pfA[q], E, Elim, 0u);
              const u32 j = (hA + xx) & (SKM_SLOTS - 1u);
              sA[j] = v; sA[j + SKM_SLOTS] = v;
            }
          }
          if ((u32)q * 64u < cb) {
            if (xx < cb) {
              const u32 v = skm_enc(pfB[q], E, Elim, 1u);
              const u32 j = (hB + xx) & (SKM_SLOTS - 1u);
              sB[j] = v; sB[j + SKM_SLOTS] = v;
            }
          }
        }
        pA += ca; pB += cb; remA -= ca; remB -= cb;
        hA = (hA + ca) & (SKM_SLOTS - 1u); hB = (hB + cb) & (SKM_SLOTS - 1u);
        prev_val = base + (u64)(lkey >> 1); prev_src = lkey & 1u; have_prev = 1;
        left = left > cnt ? left - cnt : 0u;
        if (skipping) slot += cnt;             // (slot = union index of the tile's first element while nothing is owned yet)
        wsync();
      };
      if (t_lo >= t_hi) {                      // nothing owned in this tile
        if (ends_here) {
          if (open && !committed) cursor = out_start;   // the carried chain closes without the other operand
          open = 0;
          break;
        }
        advance();
        continue;
      }
      if (skipping) { skipping = false; slot += t_lo; }
      if (t_lo > 0 || ends_here) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u64 W = skm_lt(t_hi, (u32)k) & ~skm_lt(t_lo, (u32)k);
          ALND[k] &= W; BRK[k] &= W; XQ[k] &= W; AQ[k] &= W;
        }
      }
      const bool final_tile = ends_here || inputs_end;
      u64 EM[4];
      u32 total_emit = 0;
      if (MODE == 2) {
#pragma unroll
        for (int k = 0; k < 4; k++) { EM[k] = ALND[k]; total_emit += popc64(EM[k]); }
      } else {
        // ---- chains ---------------------------------------------------------------------------------------------------------
        // A chain is emitted iff it holds both operands, i.e. iff two NEIGHBOURS of it come from different operands (ordered
        // mode: an element of the right operand directly after one of the left operand), and ordered mode emits only from the
        // first left element of the chain on.  Inside the lane: running ORs since the lane's last break (forward) and up to its
        // next break (backward), as masks over the lanes; across the lanes: the carry chains of sk_fill_fwd / sk_fill_bwd.
        u64 nobrk = ~0ull, headx = 0, fx = 0, fa = 0;
        u32 LB = 0;                                              // position of the tile's last break
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u64 NBk = ~BRK[k];
          nobrk &= NBk;                                          // no break at or before k in the lane
          headx |= nobrk & XQ[k];                                // contacts before the lane's first break
          fx = XQ[k] | (fx & NBk);                               // running OR since the lane's last break
          if (MODE == 1) fa = AQ[k] | (fa & NBk);
          if (BRK[k]) LB = max(LB, 4u * (63u - (u32)__builtin_clzll(BRK[k])) + (u32)k);
        }
        const bool has_brk = nobrk != ~0ull;                     // a chain starts in this tile
        const u64 Fx = sk_fill_fwd(fx, nobrk, committed);        // chain leaving lane l holds a contact (so far)
        const u64 Gx = sk_fill_bwd(headx, nobrk);                // chain entering lane l holds one further on
        const u64 INX = (Fx << 1) | (u64)committed, OUTX = Gx >> 1;
        u64 Fa = 0, INA = ~0ull;
        if (MODE == 1) {
          Fa = sk_fill_fwd(fa, nobrk, seenA);
          INA = (Fa << 1) | (u64)seenA;
        }
        const bool chain0_commit = committed || ((Gx & 1ull) != 0ull);         // (Gx bit 0: a contact before the tile's first break)
        if (open && !committed && (has_brk || final_tile) && !chain0_commit) cursor = out_start;   // roll back
        // backward: contacts behind k in k's run (or beyond the lane, if no break follows)
        u64 R[4];
        R[3] = OUTX;
#pragma unroll
        for (int k = 2; k >= 0; k--) R[k] = ~BRK[k + 1] & (XQ[k + 1] | R[k + 1]);
        u64 F = INX, FAr = INA;
        u32 before_lb = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const u64 NBk = ~BRK[k];
          F = XQ[k] | (F & NBk);                                 // contacts at or before k in k's run (or in front of the lane)
          const u64 LTLB = has_brk ? skm_lt(LB, (u32)k) : 0ull;  // elements in front of the last break
          const u64 CLOSED = final_tile ? ~0ull : LTLB;
          u64 e = ALND[k] & (F | R[k] | ~CLOSED);
          if (MODE == 1) { FAr = AQ[k] | (FAr & NBk); e &= FAr; }   // a left element at or before k in the chain
          EM[k] = e;
          total_emit += popc64(e);
          before_lb += popc64(e & LTLB);
        }
        // the chain that is open at the end of the tile: where its output starts, whether it holds a contact / a left element
        if (has_brk || !open) out_start = cursor + before_lb;
        open = 1; seenA = (u32)(Fa >> 63); committed = (u32)(Fx >> 63);
      }
      // compaction: position = emitted elements before (lanes below: v_mbcnt over the four masks; in the lane: the masks' own bits)
      {
        u32 pos = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) pos = __builtin_amdgcn_mbcnt_hi((u32)(EM[k] >> 32), __builtin_amdgcn_mbcnt_lo((u32)EM[k], pos));
        u64* __restrict__ oc = tmp_op + slot + cursor;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (skm_bit(EM[k])) oc[pos] = base + (u64)c[k];
          if (k < 3) pos = skm_add_bit(pos, EM[k]);
        }
      }
      cursor += total_emit;
      if (ends_here) { open = 0; break; }
      advance();
    }
    };
    if (is_or) run(SkmMode<2>());
    else if (ordered) run(SkmMode<1>());
    else run(SkmMode<0>());
    if (lane == 0) { part_count[wi] = cursor; part_src[wi] = slot | ((u64)oi << PSRC_OP_SHIFT); }
    wsync();                                                     // (the next item reuses the rings)
    w = rfl64(w_next);
  }
}

}  // namespace dcd

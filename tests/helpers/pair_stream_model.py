"""Python model of k_pair_stream (docodo_amd/csrc/dcd_pair_stream.h), wavefront by wavefront and lane by lane: 32-bit tile-relative
keys with saturation, half-cleaner + bitonic network on the blocked layout (element t = 4 * lane + k), the lane summaries, the two
carry-chain fills over the 64 lanes, ownership of the partition's chains, optimistic write + roll-back.  The kernel is a
transcription of this model; tests/test_pair_stream_model.py checks the model against the oracle's operators on the CPU, so the
LOGIC of the kernel is pinned without a GPU (the GPU tests then pin the transcription)."""
import numpy as np

M64 = (1 << 64) - 1
INF = 0xFFFFFFFF
LIM = 0x7FFFFFFF
TILE = 256


def fill_fwd(g, p, cin):
    A, B = (g | p) & M64, g & M64
    S = (A + B + cin) & M64
    c = S ^ A ^ B
    cout = (((A & B) | ((A | B) & (~S & M64))) >> 63) & 1
    return ((c >> 1) | (cout << 63)) & M64


def brev(x):
    return int('{:064b}'.format(x)[::-1], 2)


def fill_bwd(g, p):
    return brev(fill_fwd(brev(g), brev(p), 0))


def ballot(flags):
    m = 0
    for l, f in enumerate(flags):
        if f:
            m |= 1 << l
    return m


def popc(m):
    return bin(m).count("1")


def split(A, B, d):
    """number of A elements among the first d of the union (ties: A first)"""
    na, nb = len(A), len(B)
    lo, hi = max(0, d - nb), min(d, na)
    while lo < hi:
        m = (lo + hi) // 2
        if A[m] <= B[d - 1 - m]:
            lo = m + 1
        else:
            hi = m
    return lo


def network(x):
    """x: list of 64 lists of 4 keys (blocked layout); bitonic merge of a bitonic sequence, ascending"""
    for m in (32, 16, 8, 4, 2, 1):
        y = [[x[l ^ m][k] for k in range(4)] for l in range(64)]
        for l in range(64):
            for k in range(4):
                x[l][k] = min(x[l][k], y[l][k]) if (l & m) == 0 else max(x[l][k], y[l][k])
    for l in range(64):
        a = x[l]
        t0, t2, t1, t3 = min(a[0], a[2]), max(a[0], a[2]), min(a[1], a[3]), max(a[1], a[3])
        x[l] = [min(t0, t1), max(t0, t1), min(t2, t3), max(t2, t3)]
    return x


def work_item(A, B, Ra, Rb, is_or, PART, p):
    """-> list of emitted coordinates of work item p (partition p of the union)"""
    na, nb = len(A), len(B)
    absR = max(abs(Ra), abs(Rb))
    ordered = Ra < 0 and Rb < 0 and absR != 0
    thr = absR if absR != 0 else 0xFFFFFFFF
    d_begin, d_end = p * PART, p * PART + PART
    ia = split(A, B, d_begin)
    ib = d_begin - ia
    have_prev, prev_val, prev_src = 0, 0, 0
    if d_begin > 0:
        have_prev = 1
        ua, ub = ia > 0, ib > 0
        pa, pb = (A[ia - 1] if ua else 0), (B[ib - 1] if ub else 0)
        if ua and ub:
            prev_val, prev_src = (pb, 1) if pb >= pa else (pa, 0)
        elif ua:
            prev_val, prev_src = pa, 0
        else:
            prev_val, prev_src = pb, 1
    g = d_begin
    out = []            # scratch of the item from `slot` on; cursor = len(out) after truncation
    cursor, out_start = 0, 0
    opn, committed, seenA = 0, 0, 0
    skipping = not is_or
    ra, rb = min(TILE, na - ia), min(TILE, nb - ib)
    while ra + rb != 0:
        if g >= d_end and (skipping or is_or):
            break
        ha = A[ia] if ra else M64
        hb = B[ib] if rb else M64
        base = min(ha, hb)
        x = [[0] * 4 for _ in range(64)]
        ca = cb = 0
        for l in range(64):
            for k in range(4):
                t = 4 * l + k
                u = TILE - 1 - t
                ka = INF
                if t < ra and A[ia + t] - base < LIM:
                    ka = ((A[ia + t] - base) << 1) & 0xFFFFFFFF
                kb = INF
                if u < rb and B[ib + u] - base < LIM:
                    kb = (((B[ib + u] - base) << 1) | 1) & 0xFFFFFFFF
                ta = ka <= kb
                x[l][k] = ka if ta else kb
                ca += 1 if (ta and ka != INF) else 0
                cb += 1 if ((not ta) and kb != INF) else 0
        cnt = ca + cb
        assert cnt >= 1
        nra, nrb = min(TILE, na - ia - ca), min(TILE, nb - ib - cb)
        x = network(x)
        flat = [x[l][k] for l in range(64) for k in range(4)]
        assert all(flat[i] <= flat[i + 1] for i in range(255)), "network"
        pgap = 0xFFFFFFFF
        if have_prev:
            dd = base - prev_val
            pgap = dd if dd < 0xFFFFFFFF else 0xFFFFFFFF
        brk = [[0] * 4 for _ in range(64)]
        dup = [[0] * 4 for _ in range(64)]
        ctc = [[0] * 4 for _ in range(64)]
        al = [[0] * 4 for _ in range(64)]
        src = [[0] * 4 for _ in range(64)]
        for l in range(64):
            pkey = x[l - 1][3] if l > 0 else 0
            for k in range(4):
                valid = x[l][k] != INF
                sr = x[l][k] & 1
                gap = ((x[l][k] >> 1) - (pkey >> 1)) & 0xFFFFFFFF
                psrc, hp = pkey & 1, 1
                if k == 0 and l == 0:
                    gap, psrc, hp = pgap, prev_src, have_prev
                bk = (not hp) or gap > thr
                src[l][k] = sr if valid else 0
                al[l][k] = 1 if valid else 0
                brk[l][k] = 1 if (valid and bk) else 0
                dup[l][k] = 1 if (valid and sr == 1 and hp and psrc == 0 and gap == 0) else 0
                ctc[l][k] = 1 if (valid and hp and psrc != sr and not bk) else 0
                pkey = x[l][k]
        last_lane, last_k = (cnt - 1) >> 2, (cnt - 1) & 3
        lkey = x[last_lane][last_k]
        t_last_val, t_last_src = base + (lkey >> 1), lkey & 1
        inputs_end = (ia + ca == na) and (ib + cb == nb)
        t_lo, t_hi = 0, cnt
        may_end = g + cnt > d_end
        if skipping or may_end:
            lim = (d_end - g if d_end > g else 0) if may_end else 0xFFFFFFFF
            fbk, fek = [4] * 64, [4] * 64
            for l in range(64):
                for k in (3, 2, 1, 0):
                    c = al[l][k] and (is_or or brk[l][k])
                    if c:
                        fbk[l] = k
                    if c and 4 * l + k >= lim:
                        fek[l] = k
            Bf, Ef = ballot([f < 4 for f in fbk]), ballot([f < 4 for f in fek])
            fb = fe = 0xFFFFFFFF
            if Bf:
                l = (Bf & -Bf).bit_length() - 1
                fb = 4 * l + fbk[l]
            if Ef:
                l = (Ef & -Ef).bit_length() - 1
                fe = 4 * l + fek[l]
            if skipping:
                t_lo = min(fb, cnt)
            if may_end:
                t_hi = min(fe, cnt)
            if t_hi < t_lo:
                t_hi = t_lo
        ends_here = t_hi < cnt

        def advance():
            nonlocal ia, ib, ra, rb, prev_val, prev_src, have_prev, g
            ia += ca
            ib += cb
            ra, rb = nra, nrb
            prev_val, prev_src, have_prev = t_last_val, t_last_src, 1
            g += cnt
        if t_lo >= t_hi:
            if ends_here:
                if opn and not committed:
                    cursor = out_start
                opn = 0
                break
            advance()
            continue
        if skipping:
            skipping = False
        if t_lo > 0 or ends_here:
            for l in range(64):
                for k in range(4):
                    t = 4 * l + k
                    if not (t_lo <= t < t_hi):
                        al[l][k] = brk[l][k] = ctc[l][k] = 0
        final_tile = ends_here or inputs_end
        fx = [[0] * 4 for _ in range(64)]
        fa = [[0] * 4 for _ in range(64)]
        bx = [[0] * 4 for _ in range(64)]
        head = [[0] * 4 for _ in range(64)]
        tailp = [[0] * 4 for _ in range(64)]
        any_brk, tail_x, tail_a, head_x, last_bk = [0] * 64, [0] * 64, [0] * 64, [0] * 64, [0] * 64
        for l in range(64):
            ax = aa = 0
            for k in range(4):
                xq = 1 if (ctc[l][k] and ((not ordered) or src[l][k] == 1)) else 0
                aq = 1 if (al[l][k] and src[l][k] == 0) else 0
                if brk[l][k]:
                    ax = aa = 0
                    any_brk[l] = 1
                    last_bk[l] = k
                ax |= xq
                aa |= aq
                fx[l][k], fa[l][k] = ax, aa
                head[l][k] = any_brk[l] ^ 1
                bx[l][k] = xq
            tail_x[l], tail_a[l] = ax, aa
            b = later = 0
            for k in (3, 2, 1, 0):
                b |= bx[l][k]
                bx[l][k] = b
                tailp[l][k] = later ^ 1
                if brk[l][k]:
                    b = 0
                    later = 1
            head_x[l] = b
        Bm = ballot(any_brk)
        Pm = ~Bm & M64
        Fx = fill_fwd(ballot(tail_x), Pm, committed)
        Gx = fill_bwd(ballot(head_x), Pm)
        INx, OUTx = ((Fx << 1) | committed) & M64, Gx >> 1
        Fa = 0
        INa = M64
        if ordered:
            Fa = fill_fwd(ballot(tail_a), Pm, seenA)
            INa = ((Fa << 1) | seenA) & M64
        has_brk = Bm != 0
        LB = 0
        if has_brk:
            l = Bm.bit_length() - 1
            LB = 4 * l + last_bk[l]
        chain0_commit = committed or (Gx & 1)
        if (not is_or) and opn and (not committed) and (has_brk or final_tile) and not chain0_commit:
            cursor = out_start
        del out[cursor:]
        emitted = []
        before_lb = 0
        for l in range(64):
            in_x, out_x, in_a = (INx >> l) & 1, (OUTx >> l) & 1, (INa >> l) & 1
            for k in range(4):
                t = 4 * l + k
                e = 0
                if al[l][k] and not dup[l][k]:
                    if is_or:
                        e = 1
                    else:
                        cm = fx[l][k] | bx[l][k] | (head[l][k] & in_x) | (tailp[l][k] & out_x)
                        sa = (fa[l][k] | (head[l][k] & in_a)) if ordered else 1
                        closed = final_tile or (has_brk and t < LB)
                        e = 1 if (sa and (cm if closed else 1)) else 0
                if e:
                    emitted.append(base + (x[l][k] >> 1))
                    if has_brk and t < LB:
                        before_lb += 1
        out.extend(emitted)
        com = (Fx >> 63) & 1
        hasA = (Fa >> 63) & 1
        if has_brk or not opn:
            out_start = cursor + before_lb
        opn, seenA, committed = 1, hasA, com
        cursor += len(emitted)
        if ends_here:
            opn = 0
            break
        advance()
    del out[cursor:]
    return out


def evaluate(A, B, Ra, Rb, is_or, PART=8192):
    A, B = [int(v) for v in A], [int(v) for v in B]
    if (not is_or) and (len(A) == 0 or len(B) == 0):
        return []
    n = (len(A) + len(B) + PART - 1) // PART
    res = []
    for p in range(n):
        res += work_item(A, B, Ra, Rb, is_or, PART, p)
    return res


# ------------------------------------------------------------------------------------------------------------------------
# Mask form (round 4, second half): the same work items, but every per-element flag is a 64-bit LANE MASK per element slot k
# (element t = 4 * lane + k: four masks cover the tile), so the kernel evaluates the chain logic with scalar bit operations on
# SGPR pairs -- one instruction for 64 lanes -- instead of per-lane selects, and the rings hold 32-bit keys relative to an
# EPOCH coordinate E of the work item ((coordinate - E) << 1 | operand, saturated), so a tile's keys are one subtraction away.
# ------------------------------------------------------------------------------------------------------------------------
SAT = 0x7FFFFFFF            # ring keys: coordinate - E below this is representable
KINF = 0xFFFFFFFE           # saturated / padding key of operand 0 (operand 1: | 1)
REBASE = 0x80000000         # head key at or above this: new epoch


def enc(coord, E, src):
    d = coord - E
    return ((d << 1) | src) if d < SAT else (KINF | src)


def ff1(m):
    return (m & -m).bit_length() - 1


def lt_mask(v, k):
    """lanes l with 4 * l + k < v"""
    n = (min(v, 256) + 3 - k) >> 2
    return M64 if n >= 64 else (1 << n) - 1


def work_item_masks(A, B, Ra, Rb, is_or, PART, p, stats=None):
    na, nb = len(A), len(B)
    absR = max(abs(Ra), abs(Rb))
    ordered = Ra < 0 and Rb < 0 and absR != 0
    thr = absR if absR != 0 else 0xFFFFFFFF
    d_begin, d_end = p * PART, p * PART + PART
    ia = split(A, B, d_begin)
    ib = d_begin - ia
    have_prev, prev_val, prev_src = 0, 0, 0
    if d_begin > 0:
        have_prev = 1
        ua, ub = ia > 0, ib > 0
        pa, pb = (A[ia - 1] if ua else 0), (B[ib - 1] if ub else 0)
        if ua and ub:
            prev_val, prev_src = (pb, 1) if pb >= pa else (pa, 0)
        elif ua:
            prev_val, prev_src = pa, 0
        else:
            prev_val, prev_src = pb, 1
    g = d_begin
    out = []
    cursor, out_start = 0, 0
    opn, committed, seenA = 0, 0, 0
    skipping = not is_or
    if ia >= na and ib >= nb:
        return out
    E = min(A[ia] if ia < na else M64, B[ib] if ib < nb else M64)

    def stage():
        rA = [enc(A[ia + t], E, 0) if ia + t < na else KINF for t in range(TILE)]
        rB = [enc(B[ib + t], E, 1) if ib + t < nb else (KINF | 1) for t in range(TILE)]
        return rA, rB
    ringA, ringB = stage()
    while (na - ia) + (nb - ib) != 0:
        if g >= d_end and (skipping or is_or):
            break
        hk = min(ringA[0], ringB[0])
        if hk >= REBASE:                               # the head left the epoch's window (or is saturated): new epoch at the head
            E = min(A[ia] if ia < na else M64, B[ib] if ib < nb else M64)
            ringA, ringB = stage()
            hk = min(ringA[0], ringB[0])
            if stats is not None:
                stats["rebase"] = stats.get("rebase", 0) + 1
            assert hk < 2
        twob = hk & ~1
        base = E + (hk >> 1)
        x = [[min(ringA[4 * l + k] - twob, ringB[TILE - 1 - 4 * l - k] - twob) for k in range(4)] for l in range(64)]
        x = network(x)
        VALID = [ballot([x[l][k] < 0x80000000 for l in range(64)]) for k in range(4)]
        SRC = [ballot([x[l][k] & 1 for l in range(64)]) for k in range(4)]
        cnt = sum(popc(v) for v in VALID)
        ca = sum(popc(VALID[k] & ~SRC[k] & M64) for k in range(4))
        cb = cnt - ca
        assert cnt >= 1
        pgap = 0xFFFFFFFF
        if have_prev:
            dd = base - prev_val
            pgap = dd if dd < 0xFFFFFFFF else 0xFFFFFFFF
        GT, Z = [0] * 4, [0] * 4
        for k in range(4):
            for l in range(64):
                cp = (x[l][k - 1] >> 1) if k else ((x[l - 1][3] >> 1) if l else 0)
                gap = ((x[l][k] >> 1) - cp) & 0xFFFFFFFF
                if k == 0 and l == 0:
                    gap = pgap
                GT[k] |= (1 << l) if gap > thr else 0
                Z[k] |= (1 << l) if gap == 0 else 0
        # (exactly the kernel's arrangement: per slot k the breaks, the valid non-duplicates, the contacts and the left elements)
        BRK, ALND, XQ, AQ = [0] * 4, [0] * 4, [0] * 4, [0] * 4
        ps = ((SRC[3] << 1) | prev_src) & M64
        for k in range(4):
            HP = (M64 ^ 1) if (k == 0 and not have_prev) else M64
            V, S, G, Zk = VALID[k], SRC[k], GT[k], Z[k]
            BRK[k] = V & (G | (~HP & M64))
            ALND[k] = V & ~(S & ~ps & Zk & HP) & M64
            CT = V & HP & (ps ^ S) & ~G & M64
            XQ[k] = (CT & S) if ordered else CT
            AQ[k] = V & ~S & M64
            ps = S
        last_lane, last_k = (cnt - 1) >> 2, (cnt - 1) & 3
        lkey = x[last_lane][last_k]
        t_last_val, t_last_src = base + (lkey >> 1), lkey & 1
        inputs_end = (ia + ca == na) and (ib + cb == nb)
        t_lo, t_hi = 0, cnt
        may_end = g + cnt > d_end
        if skipping or may_end:
            lim = (d_end - g if d_end > g else 0) if may_end else 0xFFFFFFFF
            if is_or:
                t_hi = min(lim, cnt)
            else:
                fb = fe = 0xFFFFFFFF
                for k in range(4):
                    C = BRK[k]
                    if C:
                        fb = min(fb, 4 * ff1(C) + k)
                    Ce = C & ~lt_mask(lim, k) & M64
                    if Ce:
                        fe = min(fe, 4 * ff1(Ce) + k)
                if skipping:
                    t_lo = min(fb, cnt)
                if may_end:
                    t_hi = min(fe, cnt)
                if t_hi < t_lo:
                    t_hi = t_lo
        ends_here = t_hi < cnt

        def advance():
            nonlocal ia, ib, prev_val, prev_src, have_prev, g, ringA, ringB
            # the ca / cb consumed slots are refilled behind the 256 - ca / 256 - cb elements that stay
            ringA = ringA[ca:] + [enc(A[ia + TILE + i], E, 0) if ia + TILE + i < na else KINF for i in range(ca)]
            ringB = ringB[cb:] + [enc(B[ib + TILE + i], E, 1) if ib + TILE + i < nb else (KINF | 1) for i in range(cb)]
            ia += ca
            ib += cb
            prev_val, prev_src, have_prev = t_last_val, t_last_src, 1
            g += cnt
        if t_lo >= t_hi:
            if ends_here:
                if opn and not committed:
                    cursor = out_start
                opn = 0
                break
            advance()
            continue
        if skipping:
            skipping = False
        if t_lo > 0 or ends_here:
            for k in range(4):
                W = lt_mask(t_hi, k) & ~lt_mask(t_lo, k) & M64
                ALND[k] &= W
                BRK[k] &= W
                XQ[k] &= W
                AQ[k] &= W
        final_tile = ends_here or inputs_end
        EM = [0] * 4
        has_brk, LB = False, 0
        if is_or:
            EM = list(ALND)
        else:
            nobrk, headx, fx, fa = M64, 0, 0, 0
            for k in range(4):
                NBk = ~BRK[k] & M64
                nobrk &= NBk
                headx |= nobrk & XQ[k]
                fx = XQ[k] | (fx & NBk)
                fa = AQ[k] | (fa & NBk)
                if BRK[k]:
                    LB = max(LB, 4 * (BRK[k].bit_length() - 1) + k)
            has_brk = nobrk != M64
            Fx = fill_fwd(fx, nobrk, committed)
            Gx = fill_bwd(headx, nobrk)
            INX, OUTX = ((Fx << 1) | committed) & M64, Gx >> 1
            Fa, INA = 0, M64
            if ordered:
                Fa = fill_fwd(fa, nobrk, seenA)
                INA = ((Fa << 1) | seenA) & M64
            chain0_commit = committed or (Gx & 1)
            if opn and (not committed) and (has_brk or final_tile) and not chain0_commit:
                cursor = out_start
            R = [0, 0, 0, OUTX]
            for k in (2, 1, 0):
                R[k] = ~BRK[k + 1] & (XQ[k + 1] | R[k + 1]) & M64
            F, FAr = INX, INA
            for k in range(4):
                NBk = ~BRK[k] & M64
                F = XQ[k] | (F & NBk)
                LTLB = lt_mask(LB, k) if has_brk else 0
                CLOSED = M64 if final_tile else LTLB
                e = ALND[k] & (F | R[k] | (~CLOSED & M64))
                if ordered:
                    FAr = AQ[k] | (FAr & NBk)
                    e &= FAr
                EM[k] = e
        del out[cursor:]
        emitted = [None] * sum(popc(m) for m in EM)
        for l in range(64):
            below = (1 << l) - 1
            pos = sum(popc(EM[k] & below) for k in range(4))
            for k in range(4):
                if (EM[k] >> l) & 1:
                    emitted[pos] = base + (x[l][k] >> 1)
                    pos += 1
        before_lb = sum(popc(EM[k] & lt_mask(LB, k)) for k in range(4)) if has_brk else 0
        out.extend(emitted)
        if not is_or:
            if has_brk or not opn:
                out_start = cursor + before_lb
            opn, seenA, committed = 1, (Fa >> 63) & 1, (Fx >> 63) & 1
        cursor += len(emitted)
        if ends_here:
            opn = 0
            break
        advance()
    del out[cursor:]
    return out


def evaluate_masks(A, B, Ra, Rb, is_or, PART=8192, stats=None):
    A, B = [int(v) for v in A], [int(v) for v in B]
    if (not is_or) and (len(A) == 0 or len(B) == 0):
        return []
    n = (len(A) + len(B) + PART - 1) // PART
    res = []
    for p in range(n):
        res += work_item_masks(A, B, Ra, Rb, is_or, PART, p, stats)
    return res

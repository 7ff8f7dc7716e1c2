"""Coordinates >= 2^32 through build, merge and query, against the CPU oracle (BASELINE.json configs[2..4] are defined on a
10 M-doc index whose coordinates reach 1.024e10).

The device keeps segment-local 32-bit coordinates + a 64-bit base, the pack kernel has a 32-bit delta fast path, the AND
operator works on keys `(coord << 1) | operand` and on cell directories `coord >> shift`, PrepareSearchResult casts
`(int)(coord - pageBase)`, the shard merge re-codes the first delta of every appended segment: every one of those is only
correct "by construction" until a coordinate that does not fit 32 bits goes through it.  Here they do:

 * doc-range shards whose coord_base straddles 2^32 -> exported lists == IndexSequence.Shift of the oracle's (Index.cs:587-593,
   665-674), merge + gather -> ".index" image == the oracle's lists shifted, byte for byte;
 * that image + a hand-written page list whose page ends lie beyond 2^32 -> single-term, 3-term AND (Dist = 255), exact
   phrases, OR and a galloping pair through dcd_query_batch == oracle.Index.load(img, lst).search(...) (positions, pages, ranks);
 * the 10 M-doc index (3 segments merged on the device): oracle windows at the start, around 2^32, around a segment border
   and at the end of the corpus; conservation; 3-term AND / phrase batches == the oracle evaluating the same ".index" bytes.
"""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DOC = 1024
TWO32 = 1 << 32


def _net_string(s):
    """BinaryWriter.Write(string): 7-bit-encoded UTF-8 byte length + bytes."""
    b = s.encode("utf-8")
    n, out = len(b), bytearray()
    while n >= 0x80:
        out.append((n & 0x7F) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out) + b


def _index_image(max_coord, entries):
    """Appendix A.3: u64 MaxCoord, then per key: string, i32 n_units, units (Build.cs:375-386)."""
    out = [struct.pack("<Q", max_coord)]
    for k, units in entries:
        out.append(_net_string(k))
        out.append(struct.pack("<i", units.size))
        out.append(np.ascontiguousarray(units, "<u2").tobytes())
    return b"".join(out)


def _list_image(docs):
    """Appendix A.4 (Build.cs:99-118): docs = [(name, [(page id, end coordinate), ...])]."""
    out = []
    for name, pages in docs:
        for i, (pid, end) in enumerate(pages):
            out.append(struct.pack("<Q", end))
            if i == 0:
                out.append(_net_string(name))
                out.append(struct.pack("<Q", end))
            out.append(_net_string(":" + pid))
    return b"".join(out)


def _oracle_index(o, text, n_docs, first_doc=0):
    oix = o.Index(stemmers=[])
    ob = o.Builder(oix)
    ob.index_docs_latin1(text, np.arange(n_docs + 1, dtype=np.uint64) * DOC, first_doc=first_doc)
    img, lst = ob.save()
    oix.load(img, lst)
    return oix, ob


def _shifted_image(o, oix, base):
    """What the reference's merge leaves of a task whose coordinates start at `base`: every list Shift()ed
    (IndexSequence.cs:191-202, Index.cs:587-593), MaxCoord moved along."""
    return _index_image(oix.max_coord + base, [(k, o.seq_shift(oix.get(k), base)) for k in oix.keys()])


def _build(text, n_docs, first_doc, base0, **kw):
    import docodo_amd
    b = docodo_amd.Builder(coord_base=base0 + first_doc * DOC, **kw)
    b.AddPages(text, np.arange(n_docs + 1, dtype=np.uint64) * DOC, docs_per_page=True, first_doc=first_doc)
    b.Finish()
    return b


@pytest.mark.parametrize("base0,bounds", [
    (TWO32 - 3 * DOC, [0, 2000, 4500]),                       # 2^32 three pages into shard 0; shard 1 entirely above
    (TWO32 - 2000 * DOC, [0, 2000, 3100, 4500]),              # shard border exactly ON 2^32
    (TWO32 - 2001 * DOC - 17, [0, 2000, 4500]),               # 2^32 inside a page, shard bases not multiples of anything
    (10 * TWO32 + 12345, [0, 1500, 3000]),                    # far above (the 10 M-doc corpus ends at 2.4 * 2^32)
])
def test_shards_straddling_2_32(oracle_mod, base0, bounds):
    import docodo_amd
    from docodo_amd import dist, synth
    o = oracle_mod
    n = bounds[-1]
    text = synth.corpus_host(n)
    G = len(bounds) - 1
    shards = []
    for g in range(G):
        lo, hi = bounds[g], bounds[g + 1]
        b = _build(text[lo * DOC:hi * DOC], hi - lo, lo, base0)
        # every list of the shard == Shift(oracle's list of the same documents, base) -- IndexSequence.Shift, Index.cs:671-672
        oix, ob = _oracle_index(o, text[lo * DOC:hi * DOC], hi - lo, first_doc=lo)
        base = base0 + lo * DOC
        assert b.stats.max_coord == oix.max_coord + base
        assert b.index_image() == _shifted_image(o, oix, base), (g, base)
        ix = docodo_amd.Index.from_images(b.index_image(), b"")       # (Build() would adopt the builder's index: it is exported below)
        for k in oix.keys()[::97]:
            assert np.array_equal(ix[k].values(), o.seq_decode(oix.get(k)) + np.uint64(base)), k
        shards.append(b)
    # merge per owner + gather == the whole corpus' lists shifted by base0 (AddRange of the shifted inputs, Index.cs:665-674)
    whole_o, _ = _oracle_index(o, text, n)
    want = _shifted_image(o, whole_o, base0)
    for n_owners in (1, G):
        exports = [dist.export_shard(b, n_owners) for b in shards]
        owned = [dist.local_merge(exports, ow, two_phase=bool(ow & 1)) for ow in range(n_owners)]
        whole = dist.local_gather(owned) if n_owners > 1 else owned[0]
        assert whole.MaxCoord == whole_o.max_coord + base0
        assert whole.index_image() == want, n_owners
    # shard-only builds (what SegmentedBuilder and the distributed bench use) give the same merged image
    so = [_build(text[bounds[g] * DOC:bounds[g + 1] * DOC], bounds[g + 1] - bounds[g], bounds[g], base0, shard_only=True) for g in range(G)]
    assert dist.local_merge([dist.export_shard(b, 1) for b in so], 0).index_image() == want
    # the reference's literal merge of one term: decode(Shift(list_g, base_g)) appended in shard order, re-encoded
    k = synth.term(1)
    parts = []
    for g in range(G):
        og, _ = _oracle_index(o, text[bounds[g] * DOC:bounds[g + 1] * DOC], bounds[g + 1] - bounds[g], first_doc=bounds[g])
        parts.append(o.seq_decode(o.seq_shift(og.get(k), base0 + bounds[g] * DOC)))
    assert np.array_equal(whole[k].units, o.seq_encode(np.concatenate(parts)))


@pytest.fixture(scope="module")
def high_index(oracle_mod):
    """20 000 docs whose coordinates run from 2^32 - 10 000 pages to 2^32 + 10 000 pages, the GPU-built ".index" (checked
    against the oracle's shifted lists) under a hand-written page list: one padding page [0, base0), then the real pages."""
    import docodo_amd
    from docodo_amd import synth
    o = oracle_mod
    n, base0 = 20000, TWO32 - 10000 * DOC
    text = synth.corpus_host(n)
    b = _build(text, n, 0, base0)
    img = b.index_image()
    o0, _ = _oracle_index(o, text, n)
    assert img == _shifted_image(o, o0, base0)
    lst = _list_image([("pad:gap", [("1", base0)])] + [("syn:%d" % d, [("1", base0 + (d + 1) * DOC)]) for d in range(n)])
    # the builder's own page list: page d ends at base0 + (d + 1) * 1024 (its first page therefore starts at 0)
    assert b.list_image() == _list_image([("syn:%d" % d, [("1", base0 + (d + 1) * DOC)]) for d in range(n)])
    ix = docodo_amd.Index.from_images(img, lst)
    oix = o.Index(stemmers=[])
    assert oix.load(img, lst)
    assert ix.n_pages == n + 1 == oix.n_pages()
    return ix, oix, text, base0


def _prog(ix, words, Rs, ops):
    import docodo_amd
    p = []
    for k, (w, R) in enumerate(zip(words, Rs)):
        t = ix.find(w)
        if t is None:
            return None
        p += [docodo_amd.DCD_OP_TERM, t, R]
        if k:
            p.append(ops[k - 1])
    return p


def test_queries_beyond_2_32(high_index, oracle_mod):
    import docodo_amd
    from docodo_amd import synth
    from test_gpu_query import _check_result
    ix, oix, text, base0 = high_index
    AND, OR = docodo_amd.DCD_OP_AND, docodo_amd.DCD_OP_OR
    prog, off, reqs = [], [0], []

    def add(words, Rs, ops, req, opt=None):
        p = _prog(ix, words, Rs, ops)
        if p is None:
            return
        prog.extend(p)
        off.append(len(prog))
        reqs.append((req, opt))

    near = lambda w: 255 + len(w)
    exact = lambda w: -(len(w) + 4)
    # single terms: frequent (page cap, one gap chain across 2^32), medium, rare
    for r in (1, 2, 3, 17, 50, 101, 400, 3000, 20000):
        w = synth.term(r)
        add([w], [near(w)], [], w)
    # 3-term AND, Dist = 255: head terms (the union is one long chain through 2^32) and Zipf-sampled ones
    rows = [[1, 2, 3], [1, 5, 40], [2, 100, 7]] + [list(r) for r in synth.query_ranks(40, 3, lo_rank=20)]
    for row in rows:
        ws = [synth.term(int(k)) for k in row]
        add(ws, [near(w) for w in ws], [AND, AND], " ".join(ws))
    # OR and mixed programs: a | b, a (b | c)
    for row in synth.query_ranks(12, 3, lo_rank=5, seed=99):
        ws = [synth.term(int(k)) for k in row]
        add(ws[:2], [near(w) for w in ws[:2]], [OR], "%s | %s" % (ws[0], ws[1]))
        p = _prog(ix, ws, [near(w) for w in ws], [OR, AND])
        if p is not None:                  # a, b, c, OR, AND  ==  a * (b + c)
            t = [x for x in p]
            a, b_, c = t[0:3], t[3:6], t[7:10]
            prog.extend(a + b_ + c + [OR, AND])
            off.append(len(prog))
            reqs.append(("%s (%s | %s)" % tuple(ws), None))
    # exact phrases taken from the text itself: around 2^32 (docs 9 998 .. 10 002), at both ends and elsewhere; 2 and 3 words
    opt0 = oracle_mod.SearchOptions(dist=0)
    for qi, d in enumerate([9998, 9999, 10000, 10001, 10002, 0, 19999, 1234, 15000, 9999, 10000]):
        words = bytes(text[d * DOC:(d + 1) * DOC]).decode("ascii").split()
        k = 2 + (qi & 1)
        s = 0 if qi < 5 else (len(words) - k if qi >= 9 else 7)       # first words of the page right after 2^32, last words before it
        ws = words[s:s + k]
        add(ws, [exact(w) for w in ws], [AND] * (k - 1), '"' + " ".join(ws) + '"', opt0)
    # galloping pairs: a long list with a cell directory (>= 1024 postings, mean gap > 4 absR) x a short one
    for long_r, short_r in ((100, 3000), (150, 9000), (60, 20000), (101, 40000)):
        lw, sw = synth.term(long_r), synth.term(short_r)
        tl, ts = ix.find(lw), ix.find(sw)
        if tl is None or ts is None:
            continue
        assert ix.term_count(tl) >= 1024 and 2 * ix.term_count(ts) <= ix.term_count(tl)
        add([sw, lw], [near(sw), near(lw)], [AND], "%s %s" % (sw, lw))
        add([lw, sw], [exact(lw), exact(sw)], [AND], '"%s %s"' % (lw, sw), opt0)
    res = ix.query_batch(prog, off, want_sequences=True)
    assert res.n_queries == len(reqs) >= 90
    hi_pages = 0
    for q, (req, opt) in enumerate(reqs):
        ores = oix.search(req, opt)
        _check_result(res, q, ores)
        hi_pages += sum(1 for p in ores.foundPages if p.page_index > 10000)
        # the final sequence itself: decoded values beyond 2^32 are what the positions came from
        vals = docodo_amd.IndexSequence(res.seq_units[int(res.seq_off[q]):int(res.seq_off[q + 1])]).values()
        assert vals.size == sum(len(p.pos) for p in ores.foundPages) or len(ores.foundPages) > 30000
    assert hi_pages > 1000          # (results do lie beyond 2^32)
    # a coordinate beyond the last page: ErrorSearchResult (Build.cs:96, Search.cs:609-613) also up here
    short = _list_image([("pad:gap", [("1", base0)])] + [("syn:%d" % d, [("1", base0 + (d + 1) * DOC)]) for d in range(10001)])
    ix2 = docodo_amd.Index.from_images(ix.index_image(), short)
    oix2 = oracle_mod.Index(stemmers=[])
    assert oix2.load(ix.index_image(), short)
    w = synth.term(1)
    r2 = ix2.query_batch(_prog(ix2, [w], [near(w)], []), [0, 3])
    from docodo_amd import _lib
    assert r2.status[0] == _lib.DCD_E_NOTFOUND and not oix2.search(w).Success


N10M = 10_000_000
WIN = 3000
NPHR = 40        # (a phrase of two head terms is ~1e8 postings: ~1 s each in the single-threaded oracle)


@pytest.fixture(scope="module")
def index_10m():
    """configs[2] on one GPU: 10 M docs (10.24 GB) as three doc-range segments with global coordinates, merged on the device."""
    import torch
    from docodo_amd import dist, synth
    d = torch.empty(N10M * DOC, dtype=torch.uint8, device="cuda")
    synth.corpus_dev(d.data_ptr(), N10M)
    sb = dist.SegmentedBuilder(DOC, max_segment_chars=3_584_000_000)
    sb.add_pages_dev(d.data_ptr(), N10M)
    n_seg, n_post = len(sb.exports), sb.n_postings
    seg_docs = sb.max_pages
    ix = sb.finish()
    del d
    torch.cuda.empty_cache()
    return ix, n_seg, n_post, seg_docs


def test_10m_doc_index_against_oracle_windows(index_10m, oracle_mod):
    """The merged 10 M-doc index against the oracle on windows of WIN documents: the start (prefix), the documents around
    coordinate 2^32, around the first segment border (first-delta rebase), and the end (suffix, coordinates ~ 1.024e10)."""
    import docodo_amd
    from docodo_amd import synth
    o = oracle_mod
    ix, n_seg, n_post, seg_docs = index_10m
    assert n_seg == 3 and ix.Count > 90000 and ix.n_pages == N10M
    counts = np.array([ix.term_count(t) for t in range(ix.Count)], dtype=np.uint64)
    assert int(counts.sum()) == n_post                     # conservation: every posting is in exactly one list
    keys = [ix.key(t) for t in range(0, ix.Count, 499)]
    assert keys == sorted(keys)
    starts = [0, TWO32 // DOC - WIN // 2, seg_docs - WIN // 2, 2 * seg_docs - WIN // 2, N10M - WIN]
    wins = []
    for s in starts:
        oix, _ = _oracle_index(o, synth.corpus_host(WIN, first_doc=s), WIN, first_doc=s)
        wins.append((s * DOC, oix))
    sample = wins[0][1].keys()[::60] + [synth.term(r) for r in (1, 2, 3, 10, 100, 1000)]
    checked = 0
    for k in sample:
        t = ix.find(k)
        assert t is not None, k
        seq = ix.get(t)
        got = seq.values()
        assert got.size == counts[t] and np.all(got[1:] > got[:-1])
        for base, oix in wins:
            ou = oix.get(k)
            want = (o.seq_decode(ou) + np.uint64(base)) if ou is not None else np.empty(0, np.uint64)
            lo, hi = np.searchsorted(got, [base, base + WIN * DOC])
            assert np.array_equal(got[lo:hi], want), (k, base)
        if got.size > 100000:                              # codec round trip at scale, values up to 1.024e10
            assert np.array_equal(docodo_amd.IndexSequence.from_values(got).units, seq.units)
        checked += 1
    assert checked > 100
    assert ix.MaxCoord == max(int(ix.get(ix.find(k)).values()[-1]) for k in wins[-1][1].keys())


def test_10m_doc_queries_against_oracle(index_10m, oracle_mod):
    """configs[3] / configs[4] shapes on the 10 M-doc index: 3-term AND (Dist = 255) over Zipf ranks 101.., exact phrases
    sampled from corpus positions -- the oracle evaluates the SAME ".index" / ".index.list" bytes (the lists themselves are
    checked against oracle builds in the window test): pages / positions / documents of every query, complete results
    (positions, ranks within 1e-5) for a sample."""
    import docodo_amd
    from docodo_amd import synth
    from test_gpu_query import _check_result
    o = oracle_mod
    ix, _, _, _ = index_10m
    img, lst = ix.index_image(), ix.list_image()
    oix = o.Index(stemmers=[])
    assert oix.load(img, lst)
    del img, lst
    AND = docodo_amd.DCD_OP_AND
    prog, off, reqs = [], [0], []
    for row in synth.query_ranks(300, 3):
        ws = [synth.term(int(k)) for k in row]
        p = _prog(ix, ws, [255 + len(w) for w in ws], [AND, AND])
        if p is None:
            continue
        prog += p
        off.append(len(prog))
        reqs.append((" ".join(ws), None))
    rng = np.random.default_rng(0x5EA2C4)
    opt0 = o.SearchOptions(dist=0)
    for q in range(NPHR):
        doc = int(rng.integers(0, N10M)) if q >= 8 else [N10M - 1, TWO32 // DOC, TWO32 // DOC - 1, 0, 7_000_000, 3_500_000, 9_999_998, 4_194_305][q]
        words = bytes(synth.corpus_host(1, first_doc=doc)).decode("ascii").split()
        k = 2 + (q & 1)
        s = int(rng.integers(0, len(words) - k + 1))
        ws = words[s:s + k]
        p = _prog(ix, ws, [-(len(w) + 4) for w in ws], [AND] * (k - 1))
        assert p is not None            # (the phrase occurs in the corpus)
        prog += p
        off.append(len(prog))
        reqs.append(('"' + " ".join(ws) + '"', opt0))
    res = ix.query_batch(prog, off)
    o_pages, o_pos, o_docs, _ = oix.query_batch(prog, off)
    assert np.array_equal(np.diff(res.page_off), o_pages)
    assert np.array_equal(np.diff(res.doc_off), o_docs)
    got_pos = np.array([int(res.pos_off[int(res.page_off[q + 1])]) - int(res.pos_off[int(res.page_off[q])]) for q in range(res.n_queries)], np.uint64)
    assert np.array_equal(got_pos, o_pos)
    assert int(o_pages[len(reqs) - NPHR:].min()) >= 1       # every sampled phrase is found at least where it was taken from
    for q in list(range(0, len(reqs) - NPHR, 23)) + list(range(len(reqs) - NPHR, len(reqs), 7)):
        _check_result(res, q, oix.search(reqs[q][0], reqs[q][1]))

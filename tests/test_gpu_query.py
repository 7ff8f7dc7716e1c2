"""GPU parity of the query path against the CPU oracle: operators * / + (bit-exact units and R),
PrepareSearchResult (pages, page-relative positions, caps), page / doc ranks (1e-5)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_seq(rng, n, max_gap, start=0):
    if n == 0:
        return np.array([], np.uint64)
    return (np.cumsum(rng.integers(1, max_gap, n, dtype=np.uint64)) + np.uint64(start)).astype(np.uint64)


def _ops_cases():
    rng = np.random.default_rng(321)
    cases = []
    for na, nb, ga, gb in [(0, 0, 10, 10), (0, 5, 10, 10), (7, 0, 10, 10), (1, 1, 5, 5), (50, 60, 40, 30), (1000, 1200, 300, 250),
                           (5000, 300, 50, 4000), (3000, 3000, 3, 3), (20000, 17000, 700, 900), (4096, 4096, 2, 2),
                           (100000, 2500, 60, 2400),
                           # sparse + skewed: the galloping path (16 |short| <= |long|, mean gap of the long list > 4 absR)
                           (300, 30000, 300000, 3000), (20000, 700, 2000, 60000), (5, 4000, 9000000, 12000), (1, 2000, 10, 5000),
                           (1500, 60000, 40, 1500),
                           # gaps beyond 2^31: k_pair_stream's ring keys leave their epoch (the rings are staged again at the head)
                           (300, 500, 1 << 33, 1 << 33), (700, 600, 40, 1 << 34), (2000, 1500, 1 << 30, 1 << 31)]:
        a, b = _rand_seq(rng, na, ga), _rand_seq(rng, nb, gb)
        # force shared values (double-tagged elements)
        if na and nb:
            k = min(na, nb) // 3
            b = np.unique(np.concatenate([b, a[rng.integers(0, na, k)]])).astype(np.uint64)
        cases.append((a, b))
    return cases


def test_neighbours_across_lane_borders(oracle_mod):
    """k_pair_stream keeps four consecutive union elements per lane; the gap to the element before a lane's first one comes from
    the lane below (a DPP move).  Round 4: folded into the subtraction by the compiler (v_subrev_u32_dpp ... wave_shr:1) that
    move delivered garbage on gfx950 -- every chain broke at a lane border and a coordinate held by both operands was emitted
    twice when its two copies sat in different lanes.  Tiny operands that put a pair exactly on every border position."""
    import docodo_amd
    o = oracle_mod
    cases = []
    for n in range(1, 10):                                   # [1 .. n - 1, 5 n] x [5 n]: the shared coordinate at union index n - 1, n
        a = list(range(1, n)) + [5 * n]
        cases.append((a, [5 * n]))
        cases.append((a, [5 * n, 5 * n + 4]))
        cases.append(([5 * n], a))
    cases.append((list(range(10, 100, 10)), list(range(10, 100, 10))))
    cases.append(([5, 9], [5, 9, 13]))
    for Ra, Rb in [(258, 259), (-7, -9), (0, 0), (3, 2)]:
        for a, b in cases:
            a, b = np.array(a, np.uint64), np.array(b, np.uint64)
            ua, ub = o.seq_encode(a), o.seq_encode(b)
            sa, sb = docodo_amd.IndexSequence(ua, Ra), docodo_amd.IndexSequence(ub, Rb)
            want, _ = o.seq_and(ua, Ra, ub, Rb)
            assert np.array_equal((sa * sb).units, want), ("and", list(a), list(b), Ra, Rb)
            want, _ = o.seq_or(ua, Ra, ub, Rb)
            assert np.array_equal((sa + sb).units, want), ("or", list(a), list(b), Ra, Rb)


def test_stream_kernel_merge_network_sorts_bitonic_tiles():
    """k_pair_stream's merge network (v_med3 compare-exchanges fed by DPP moves and the gfx950 lane swaps) on its own: the half-
    cleaner output of two ascending key sequences -- min(A[t], B[255 - t]), a bitonic sequence incl. +inf padding -- must come
    back ascending and as the same multiset."""
    from docodo_amd import _lib
    rng = np.random.default_rng(11)
    for case in range(40):
        na, nb = int(rng.integers(0, 257)), int(rng.integers(0, 257))
        a = np.sort(rng.choice(1 << 20, na, replace=False).astype(np.uint32) << 1)
        b = np.sort((rng.choice(1 << 20, nb, replace=False).astype(np.uint32) << 1) | 1)
        A = np.full(256, 0xFFFFFFFF, np.uint32); A[:na] = a
        B = np.full(256, 0xFFFFFFFF, np.uint32); B[:nb] = b
        x = np.minimum(A, B[::-1]).copy()
        want = np.sort(x)
        _lib.check(_lib.lib().dcd_debug_sk_network(x.ctypes.data))
        assert np.array_equal(x, want), (case, na, nb, int(np.argmax(x != want)))


@pytest.mark.parametrize("Ra,Rb", [(258, 259), (0, 0), (-7, -9), (-7, 258), (258, -9), (40, 3), (-1, -1), (0, -5), (100000, 1)])
def test_operators_parity(oracle_mod, Ra, Rb):
    import docodo_amd
    o = oracle_mod
    for a, b in _ops_cases():
        ua, ub = o.seq_encode(a), o.seq_encode(b)
        sa, sb = docodo_amd.IndexSequence(ua, Ra), docodo_amd.IndexSequence(ub, Rb)
        want, wR = o.seq_and(ua, Ra, ub, Rb)
        got = sa * sb
        assert got.R == wR
        assert np.array_equal(got.units, want), ("and", a.size, b.size, Ra, Rb)
        want, wR = o.seq_or(ua, Ra, ub, Rb)
        got = sa + sb
        assert got.R == wR
        assert np.array_equal(got.units, want), ("or", a.size, b.size)


@pytest.mark.parametrize("part", [64, 300, 1024])
@pytest.mark.parametrize("Ra,Rb", [(258, 259), (-7, -9), (0, 0), (40, -3)])
def test_operators_partition_sizes(oracle_mod, monkeypatch, part, Ra, Rb):
    """Operators are evaluated by one wavefront per partition of the operand union (8192 elements by default);
    tiny partitions put a partition border inside almost every gap chain: skipped heads, chains followed past
    the partition end, duplicates and ties on a border, empty partitions."""
    import docodo_amd
    o = oracle_mod
    monkeypatch.setenv("DOCODO_QUERY_PART", str(part))
    for a, b in _ops_cases():
        ua, ub = o.seq_encode(a), o.seq_encode(b)
        sa, sb = docodo_amd.IndexSequence(ua, Ra), docodo_amd.IndexSequence(ub, Rb)
        want, wR = o.seq_and(ua, Ra, ub, Rb)
        got = sa * sb
        assert got.R == wR
        assert np.array_equal(got.units, want), ("and", a.size, b.size, Ra, Rb, part)
        want, wR = o.seq_or(ua, Ra, ub, Rb)
        got = sa + sb
        assert np.array_equal(got.units, want), ("or", a.size, b.size, part)


@pytest.mark.parametrize("part", [None, 256])
def test_long_chain_across_tiles(oracle_mod, monkeypatch, part):
    """One gap chain spanning many 256-element tiles (and many partitions) that gains its second operand only at the very
    end (optimistic write kept) or never (rolled back); ordered mode with a late first-A."""
    import docodo_amd
    o = oracle_mod
    if part:
        monkeypatch.setenv("DOCODO_QUERY_PART", str(part))
    a = np.arange(10, 10 + 5 * 6000, 5, dtype=np.uint64)          # dense: every gap 5
    for b in (np.array([a[-1] + 3], np.uint64), np.array([a[-1] + 300], np.uint64), np.array([a[0] - 3], np.uint64),
              np.array([a[3000] + 1], np.uint64)):
        for Ra, Rb in ((10, 10), (-10, -10)):
            for x, y in ((a, b), (b, a)):
                ux, uy = o.seq_encode(x), o.seq_encode(y)
                want, wR = o.seq_and(ux, Ra, uy, Rb)
                got = docodo_amd.IndexSequence(ux, Ra) * docodo_amd.IndexSequence(uy, Rb)
                assert got.R == wR and np.array_equal(got.units, want), (b, Ra, x.size)


@pytest.fixture(scope="module")
def synth_index(oracle_mod):
    import docodo_amd
    from docodo_amd import synth
    n = 40000
    text = synth.corpus_host(n)
    off = np.arange(n + 1, dtype=np.uint64) * synth.DOC_BYTES
    b = docodo_amd.Builder()
    b.AddPages(text, off, docs_per_page=True)
    b.Finish()
    img, lst = b.index_image(), b.list_image()
    ix = b.Build()
    oix = oracle_mod.Index(stemmers=[])
    assert oix.load(img, lst)
    return ix, oix, img, lst


def _check_result(res, q, ores, tol=1e-5):
    assert res.status[q] == 0 and ores.Success
    pages = res.pages(q)
    assert len(pages) == len(ores.foundPages)
    for (pi, pos, rank), op in zip(pages, ores.foundPages):
        assert pi == op.page_index
        assert np.array_equal(pos, op.pos)
        assert abs(rank - op.rank) <= tol * max(1.0, abs(op.rank))
    docs = res.docs(q)
    assert len(docs) == len(ores.foundDocs)
    for (di, rank, npg), od in zip(docs, ores.foundDocs):
        assert abs(rank - od.rank) <= tol * max(1.0, abs(od.rank))
        assert npg == od.n_pages
    # same doc set; order may differ only among equal ranks
    assert sorted(d[0] for d in docs) == sorted(d.doc_index for d in ores.foundDocs)


def test_single_term_queries(synth_index):
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    ranks = list(synth.query_ranks(40, 1, lo_rank=101)[:, 0]) + [1, 2, 3, 50]   # incl. terms past the 30000-page cap
    words = [synth.term(int(k)) for k in ranks]
    prog, off, kept = [], [0], []
    for w in words:
        t = ix.find(w)
        if t is None:
            assert oix.get(w) is None
            continue
        prog += [docodo_amd.DCD_OP_TERM, t, 255 + len(w)]
        off.append(len(prog))
        kept.append(w)
    res = ix.query_batch(prog, off)
    assert res.n_queries == len(kept)
    for q, w in enumerate(kept):
        _check_result(res, q, oix.search(w))


def test_three_term_and_queries(synth_index, oracle_mod):
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    ranks = synth.query_ranks(60, 3, lo_rank=20)
    prog, off, reqs = [], [0], []
    for row in ranks:
        ws = [synth.term(int(k)) for k in row]
        ts = [ix.find(w) for w in ws]
        if any(t is None for t in ts):
            continue
        for k, (w, t) in enumerate(zip(ws, ts)):
            prog += [docodo_amd.DCD_OP_TERM, t, 255 + len(w)]       # opt == null -> Dist = 255 (Search.cs:472)
            if k:
                prog.append(docodo_amd.DCD_OP_AND)
        off.append(len(prog))
        reqs.append(" ".join(ws))
    res = ix.query_batch(prog, off, want_sequences=True)
    for q, req in enumerate(reqs):
        _check_result(res, q, oix.search(req))
    # exact phrases: "w1 w2" -> R = -(len + 4) on both (Search.cs:308)
    prog, off, reqs = [], [0], []
    for row in ranks[:30]:
        ws = [synth.term(int(k)) for k in row[:2]]
        ts = [ix.find(w) for w in ws]
        if any(t is None for t in ts):
            continue
        for k, (w, t) in enumerate(zip(ws, ts)):
            prog += [docodo_amd.DCD_OP_TERM, t, -(len(w) + 4)]
            if k:
                prog.append(docodo_amd.DCD_OP_AND)
        off.append(len(prog))
        reqs.append('"' + " ".join(ws) + '"')
    res = ix.query_batch(prog, off)
    for q, req in enumerate(reqs):
        _check_result(res, q, oix.search(req, oracle_mod.SearchOptions(dist=0)))


def test_three_term_queries_small_partitions(synth_index, monkeypatch):
    """Same batch evaluated with 8192-element and 200-element partitions: identical results."""
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    ranks = synth.query_ranks(200, 3, lo_rank=1)
    prog, off = [], [0]
    for row in ranks:
        ts = [ix.find(synth.term(int(k))) for k in row]
        if any(t is None for t in ts):
            continue
        for k, t in enumerate(ts):
            prog += [docodo_amd.DCD_OP_TERM, t, (255 + len(synth.term(int(row[k])))) if (len(off) & 1) else -(len(synth.term(int(row[k]))) + 4)]
            if k:
                prog.append(docodo_amd.DCD_OP_AND if (k == 1 or len(off) % 3) else docodo_amd.DCD_OP_OR)
        off.append(len(prog))
    monkeypatch.delenv("DOCODO_QUERY_PART", raising=False)
    a = ix.query_batch(prog, off, want_sequences=True)
    monkeypatch.setenv("DOCODO_QUERY_PART", "200")
    b = ix.query_batch(prog, off, want_sequences=True)
    for name in _RESULT_ARRAYS:
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


_RESULT_ARRAYS = ("status", "page_off", "page_index", "page_rank", "pos_off", "pos", "doc_off", "doc_index", "doc_rank",
                  "doc_n_pages", "seq_off", "seq_units", "seq_R")


def _and_programs(ix, ranks):
    import docodo_amd
    from docodo_amd import synth
    prog, off = [], [0]
    for row in ranks:
        ts = [ix.find(synth.term(int(k))) for k in row]
        if any(t is None for t in ts):
            continue
        for k, t in enumerate(ts):
            prog += [docodo_amd.DCD_OP_TERM, t, 255 + len(synth.term(int(row[k])))]
            if k:
                prog.append(docodo_amd.DCD_OP_AND)
        off.append(len(prog))
    return prog, off


def test_a_full_work_pool_is_retried_with_smaller_sub_batches(synth_index, monkeypatch):
    """Operator results are placed in the work pool by the device, one behind the other (round 4).  When the pool runs full the
    sub-batch is abandoned (every kernel behind that point leaves at once, nothing of the attempt is kept) and evaluated again
    as two halves, each with a pool and a table of common sub-expressions of its own.  Pools from far too large down to too
    small for a single query: every evaluation that succeeds equals the undisturbed one; one that cannot succeed says NOMEM."""
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    prog, off = _and_programs(ix, synth.query_ranks(100, 3, lo_rank=5))
    monkeypatch.delenv("DOCODO_QUERY_POOL_ELEMS", raising=False)
    a = ix.query_batch(prog, off, want_sequences=True)
    split, failed = 0, 0
    for lg in range(22, 7, -1):
        monkeypatch.setenv("DOCODO_QUERY_POOL_ELEMS", str(1 << lg))
        try:
            b = ix.query_batch(prog, off, want_sequences=True)
        except docodo_amd.DocodoError as e:
            assert e.code == docodo_amd.DCD_E_NOMEM
            failed += 1
            continue
        assert failed == 0, "a smaller pool worked after a larger one had failed"
        assert len(b.status) == len(off) - 1
        for name in _RESULT_ARRAYS:
            assert np.array_equal(getattr(a, name), getattr(b, name)), (lg, name)
        split += b.sub_batches > 1
    assert a.sub_batches == 1 and split >= 1


def test_chunked_batches_and_shared_operators(synth_index, monkeypatch):
    """A batch runs in chunks whose plans are made on a helper thread while the previous chunk is on the device, and an operator
    that an earlier query of the sub-batch named (same operands, R values, opcode) is evaluated once -- also across chunks.
    Every query here occurs three times, spread over the batch; chunks of 7 queries; results as in one chunk, each copy equal."""
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    ranks = synth.query_ranks(40, 3, lo_rank=10)
    ranks = np.concatenate([ranks, ranks[::-1], ranks])
    prog, off = _and_programs(ix, ranks)
    monkeypatch.delenv("DOCODO_QUERY_CHUNK", raising=False)
    a = ix.query_batch(prog, off, want_sequences=True)
    monkeypatch.setenv("DOCODO_QUERY_CHUNK", "7")
    b = ix.query_batch(prog, off, want_sequences=True)
    for name in _RESULT_ARRAYS:
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    n = (len(off) - 1) // 3
    assert 3 * n == len(off) - 1
    for q in range(n):
        for c in (2 * n - 1 - q, 2 * n + q):
            assert np.array_equal(a.seq_units[a.seq_off[q]:a.seq_off[q + 1]], a.seq_units[a.seq_off[c]:a.seq_off[c + 1]])
            assert a.page_off[q + 1] - a.page_off[q] == a.page_off[c + 1] - a.page_off[c]
    # 2 operators per query named, at most one third of them evaluated
    assert a.ops_named == 2 * (len(off) - 1) and a.ops_evaluated <= a.ops_named // 3
    assert b.ops_evaluated == a.ops_evaluated


def test_concurrent_batches_on_one_index(synth_index):
    """The reference issues Search from many pool threads (server.cs:16-30) under one lock (Search.cs:448);
    dcd_query_batch on one handle is serialised internally: concurrent callers get the single-threaded answers."""
    import threading
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    ranks = synth.query_ranks(64, 3, lo_rank=1)
    progs = []
    for row in ranks:
        ts = [ix.find(synth.term(int(k))) for k in row]
        if any(t is None for t in ts):
            continue
        prog = []
        for k, t in enumerate(ts):
            prog += [docodo_amd.DCD_OP_TERM, t, 255 + len(synth.term(int(row[k])))]
            if k:
                prog.append(docodo_amd.DCD_OP_AND)
        progs.append(prog)
    want = [ix.query_batch(p, [0, len(p)], want_sequences=True) for p in progs]
    got = [None] * len(progs)

    def work(lo, hi):
        for i in range(lo, hi):
            got[i] = ix.query_batch(progs[i], [0, len(progs[i])], want_sequences=True)

    n = len(progs)
    th = [threading.Thread(target=work, args=(j * n // 4, (j + 1) * n // 4)) for j in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for a, b in zip(want, got):
        for name in ("status", "page_index", "page_rank", "pos", "doc_index", "doc_rank", "seq_units"):
            assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_or_and_mixed_plan(synth_index):
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    ws = [synth.term(k) for k in (120, 340, 777, 45)]
    ts = [ix.find(w) for w in ws]
    assert all(t is not None for t in ts)
    T, A, O = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND, docodo_amd.DCD_OP_OR
    R = [255 + len(w) for w in ws]
    # a (b | c) d  ->  a.d()*(b.d()+c.d())*d.d()
    prog = [T, ts[0], R[0], T, ts[1], R[1], T, ts[2], R[2], O, A, T, ts[3], R[3], A]
    res = ix.query_batch(prog, [0, len(prog)])
    _check_result(res, 0, oix.search("%s (%s | %s) %s" % tuple(ws)))


def test_index_roundtrip_through_images(synth_index, oracle_mod):
    """Index.Load (Index.cs:312-380): an index re-opened from the .index/.index.list bytes answers the same."""
    import docodo_amd
    from docodo_amd import synth
    ix, oix, img, lst = synth_index
    ix2 = docodo_amd.Index.from_images(img, lst)
    assert ix2.Count == ix.Count == oix.count and ix2.MaxCoord == ix.MaxCoord == oix.max_coord
    assert ix2.n_pages == oix.n_pages() and ix2.n_docs == oix.n_docs()
    for k in (1, 150, 5000):
        w = synth.term(k)
        t = ix2.find(w)
        if t is None:
            continue
        assert ix2.key(t) == w
        assert np.array_equal(ix2.get(t).units, oix.get(w))
        assert ix2.term_count(t) == oracle_mod.seq_decode(oix.get(w)).size
        res = ix2.query_batch([docodo_amd.DCD_OP_TERM, t, 255 + len(w)], [0, 3])
        _check_result(res, 0, oix.search(w))
    assert ix2.page_info(3) == (oix.page_end(3), oix.page_doc(3), oix.page_id(3))
    assert ix2.doc_name(5) == oix.doc_name(5)


def test_sub_batching_matches_single_batch(synth_index, monkeypatch):
    """dcd_query_batch splits a batch whose worst-case work pool exceeds the HBM budget; the concatenated
    result must equal the one-shot evaluation (budget forced down to 200 k elements here)."""
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    ranks = synth.query_ranks(300, 3, lo_rank=20)
    prog, off = [], [0]
    for row in ranks:
        ts = [ix.find(synth.term(int(k))) for k in row]
        if any(t is None for t in ts):
            continue
        for k, t in enumerate(ts):
            prog += [docodo_amd.DCD_OP_TERM, t, 255 + len(synth.term(int(row[k])))]
            if k:
                prog.append(docodo_amd.DCD_OP_AND)
        off.append(len(prog))
    monkeypatch.delenv("DOCODO_QUERY_POOL_ELEMS", raising=False)
    a = ix.query_batch(prog, off, want_sequences=True)
    monkeypatch.setenv("DOCODO_QUERY_POOL_ELEMS", "200000")
    b = ix.query_batch(prog, off, want_sequences=True)
    for name in _RESULT_ARRAYS:
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_no_download_returns_the_same_counts(synth_index):
    """dcd_query_opts.no_download: the batch is evaluated and every result array produced on the device as always, only the
    per-query counts and the totals come back (bench.py's query legs; statistics-only callers)."""
    import docodo_amd
    from docodo_amd import synth
    ix, oix, _, _ = synth_index
    prog, off = [], [0]
    for row in synth.query_ranks(50, 3, lo_rank=5):
        ts = [ix.find(synth.term(int(k))) for k in row]
        if any(t is None for t in ts):
            continue
        for k, (r, t) in enumerate(zip(row, ts)):
            prog += [docodo_amd.DCD_OP_TERM, t, 255 + len(synth.term(int(r)))]
            if k:
                prog.append(docodo_amd.DCD_OP_AND)
        off.append(len(prog))
    full = ix.query_batch(prog, off)
    tot = ix.query_batch(prog, off, totals=True)
    assert tot["n_pages"] == full.page_index.size and tot["n_pos"] == full.pos.size and tot["n_docs"] == full.doc_index.size
    assert tot["ms"] > 0
    empty = ix.query_batch([], [0], totals=True)
    assert (empty["n_pages"], empty["n_pos"], empty["n_docs"]) == (0, 0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("script,args", [("stress_ops.py", ["80", "11"]), ("stress_queries.py", ["30000", "300", "5"])])
def test_work_counter_loop_against_oracle(script, args):
    """k_pair_parts takes its work items from a device counter with a fixed grid of wavefronts (round 3).  With a grid of
    THREE wavefronts every wavefront works through many items of many operators -- galloping and streaming ones in turn --
    and the randomised differential tests (operators through dcd_seq_and / dcd_seq_or, programs through dcd_query_batch)
    must still match the oracle bit for bit."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DOCODO_QUERY_PW_GRID="3")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", script)] + args, capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ", 0 mismatches" in r.stdout.strip().splitlines()[-1], r.stdout[-2000:]

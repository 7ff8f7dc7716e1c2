"""CPU-side pieces of bench.py that need no GPU: the oracle query baselines (one thread, and query-sharded over ALL cores on one
shared index with a private page cursor per thread -- SURVEY 8d; the oracle's page cursor, like the reference's
IndexPageList.GetPage (Build.cs:74-97), is the only state a search changes, which is why the reference serialises Search under one
lock, Search.cs:448)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _small_index(n_docs=1500):
    import oracle
    from docodo_amd import synth
    text = synth.corpus_host(n_docs)
    ix = oracle.Index(stemmers=[])
    b = oracle.Builder(ix)
    b.index_docs_latin1(text, np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES, first_doc=0)
    return b.save()


def test_query_baselines_one_thread_and_query_sharded():
    import bench
    import oracle
    img, lst = _small_index()
    o = oracle.Index(stemmers=[])
    o.load(img, lst)
    prog, off = [], [0]
    for t in range(0, 240, 2):                       # two-term AND programs over the first terms of the dictionary
        prog += [0, t, 258, 0, t + 1, 259, 1]
        off.append(len(prog))
    progs = [(np.array(prog, np.int32), np.array(off, np.uint64))]
    one = bench.oracle_query_sample(o, progs, 0.05)
    assert one["cores"] == 1 and one["kind"] == "port" and 0 < one["queries"] <= 120 and one["queries_per_s"] > 0
    allc = bench.oracle_query_all_cores(o, progs, one["queries"], 0.05)
    if allc is not None:                             # (None on a one-core host)
        assert allc["cores"] >= 2 and allc["queries_per_s"] > 0
    # the threads answer exactly like the one-thread evaluation: same per-query page / position counts and checksums, for any
    # number of threads (private page cursors on one shared index), also when the programs jump back and forth in the corpus
    pa, oa = progs[0]
    ref = o.query_batch(pa, oa)
    for nt in (1, 2, 3, 7):
        got = o.query_batch_mt(pa, oa, nt)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b), nt
    o2 = oracle.Index(stemmers=[])
    o2.load(img, lst)
    got = o2.query_batch(pa, oa)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)

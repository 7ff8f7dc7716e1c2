"""Pins the CPU oracle against the reference's own known-answer tests.

Each test mirrors one xUnit fact of /root/reference/XUnitDocodoTest (file:line in the docstring);
inputs come from tests/golden (TestText1 / TestHeaders1 extracted by make_fixtures.py).
"""
import numpy as np
import pytest

from oracle import porter2

# Index.cs:175-181 -- default stemmer table (only "digit" and "en" can fire on the test corpus)
DEFAULT_STEMMERS = [("0-9", None), ("а-яё", None), ("a-z", porter2.stem), ("a-zẞäüö", None),
                    ("a-zéâàêèëçîïôûùüÿ", None)]


def build_test_index(o, text, headers, n_pages, stemmers, vocab=None):
    """TestDataSource (IndexTest.cs:87-124) through one IndexTask (Build.cs:445-564)."""
    ix = o.Index(stemmers=stemmers)
    if vocab:
        ix.add_voc(vocab["words"], vocab["range"])
    b = o.Builder(ix)
    for name in ("Sample", "Dump"):
        b.add_doc("Test", name)
        b.index_page("0", headers + "Name=%s\n" % name)
        b.index_page("1", text)
        for q in range(n_pages - 1):
            b.index_page(str(q + 2), text)
    ib, lb = b.save()
    assert ix.load(ib, lb)
    return ix, b, ib, lb


@pytest.fixture(scope="module")
def idx_nostem(oracle_mod, testtext1, testheaders1):
    return build_test_index(oracle_mod, testtext1, testheaders1, 50, [])


@pytest.fixture(scope="module")
def idx_stem(oracle_mod, testtext1, testheaders1):
    return build_test_index(oracle_mod, testtext1, testheaders1, 100, DEFAULT_STEMMERS)


def test_D1_coord(idx_nostem, known):
    """IndexTest.CoordTest, IndexTest.cs:126-157 (N reduced from 1000 to 50 pages per doc: the
    assertion is per page and N-independent)."""
    ix = idx_nostem[0]
    for w, pos in known["D1_coord"]["words"].items():
        res = ix.search(w)
        assert res.Success
        assert len(res.foundPages) == 2 * 50
        for p in res.foundPages:
            assert list(p.pos) == pos
    assert [len(v) for v in known["D1_coord"]["words"].values()] == [29, 2, 3, 11]


def test_D2_D3_request_syntax(idx_stem, known, oracle_mod):
    """IndexTest.RequestSyntaxTest, IndexTest.cs:180-196."""
    ix = idx_stem[0]
    res = ix.search(known["D2"]["req"])
    assert res.Success and len(res.foundDocs) == 2
    assert res.foundDocs[0].n_pages == 100 and res.foundDocs[-1].n_pages == 100
    res = ix.search(known["D3"]["req"])
    assert res.Success and len(res.foundDocs) == 1
    assert res.foundDocs[0].n_pages == 100
    assert res.foundDocs[0].name == "Test:Dump"
    assert len(res.foundPages[0].pos) == len(res.foundPages[1].pos)
    for q in range(200):
        assert len(res.foundPages[q].pos) == 42
    assert list(res.foundPages[0].pos) == list(res.foundPages[1].pos)


@pytest.mark.parametrize("key", ["D4", "D5", "D6", "D7", "D8", "D9a", "D9b", "D10", "D11"])
def test_D4_D11_dist40(idx_stem, known, oracle_mod, key):
    """IndexTest.cs:198-220 (dist = 40)."""
    ix = idx_stem[0]
    ka = known[key]
    res = ix.search(ka["req"], oracle_mod.SearchOptions(dist=ka["dist"]))
    assert res.Success, res.Error
    if "n_found_pages" in ka:
        assert len(res.foundPages) == ka["n_found_pages"]
    else:
        assert len(res.foundPages[0].pos) == ka["pos_page0"]


def test_D12_vocab(oracle_mod, testtext1, testheaders1, known):
    """IndexTest.VocabTest, IndexTest.cs:228-267."""
    ka = known["D12"]
    voc = {"words": ka["vocab"], "range": ka["range"]}
    vix, _, _, _ = build_test_index(oracle_mod, testtext1, testheaders1, ka["n_pages_per_doc"], DEFAULT_STEMMERS, voc)
    ix, _, _, _ = build_test_index(oracle_mod, testtext1, testheaders1, ka["n_pages_per_doc"], DEFAULT_STEMMERS)
    assert oracle_mod.seq_decode(vix.get("#1")).size == ka["count_code_1"]
    for req in ("Tupman", "and", "Tupman and"):
        a, b = vix.search(req), ix.search(req)
        assert a.Success and b.Success
        assert len(a.foundPages) == len(b.foundPages)
        for p, q in zip(a.foundPages, b.foundPages):
            assert p.id == q.id and list(p.pos) == list(q.pos)


def test_D13_builder(oracle_mod, known):
    """IndexTest.BuilderTest, IndexTest.cs:272-313: AddWord(match, index) for \\b\\w+\\b tokens."""
    ka = known["D13_builder"]
    ix = oracle_mod.Index(stemmers=DEFAULT_STEMMERS)
    ix.add_voc(known["D12"]["vocab"], known["D12"]["range"])
    b = oracle_mod.Builder(ix)
    b.add_doc("A", "")
    for off, tok in ka["tokens"]:
        b.add_word(tok, off)
    b.end_page("1")
    ib, lb = b.save()
    assert ix.load(ib, lb)
    for w, pos in ka["words"].items():
        res = ix.search(w)
        assert res.Success and len(res.foundPages) == 1
        assert list(res.foundPages[0].pos) == pos


def test_D14_codec_properties(oracle_mod):
    """IndexSequenceTest.cs:17-79, 118-134: round trip, Shift by 0/100/0xFFFFF/0xFFFFFFFF."""
    o = oracle_mod
    rng = np.random.default_rng(1)
    vals = np.cumsum(rng.integers(0, 0x10000, 100, dtype=np.uint64))
    u = o.seq_encode(vals)
    assert np.array_equal(o.seq_decode(u), vals)
    one = o.seq_encode([62])
    assert list(o.seq_shift(one, 0)) == [62] and list(o.seq_decode(one)) == [62]
    vals0 = np.concatenate([[0], vals]).astype(np.uint64)
    u0 = o.seq_encode(vals0)
    s100 = o.seq_shift(u0, 100)
    assert s100.size == u0.size
    for s in (100, 0xFFFFF, 0xFFFFFFFF):
        d = o.seq_decode(o.seq_shift(u0, s))
        assert d.size == vals0.size
        assert np.array_equal(d - vals0, np.full(vals0.size, s, np.uint64))


def test_codec_known_vectors(oracle_mod):
    """Format definition, IndexSequence.cs:13-15, 63-84."""
    o = oracle_mod
    assert list(o.seq_encode([0])) == [0]
    assert list(o.seq_encode([0x7FFF])) == [0x7FFF]
    assert list(o.seq_encode([0x8000])) == [0x8000, 1]
    assert list(o.seq_encode([5, 5])) == [5, 0]
    assert list(o.seq_encode([(1 << 30)])) == [0x8000, 0x8000, 1]
    assert list(o.seq_encode([0xFFFFFFFFFFFFFFFF])) == [0xFFFF, 0xFFFF, 0xFFFF, 0xFFFF, 0xF]
    assert list(o.seq_decode(np.array([0x8000, 1, 3], np.uint16))) == [0x8000, 0x8003]
    # truncated continuation: the value is not yielded (IndexSequence.cs:121-128)
    assert list(o.seq_decode(np.array([7, 0x8001], np.uint16))) == [7]

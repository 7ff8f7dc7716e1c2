"""The reference's own known-answer tests (XUnitDocodoTest/IndexTest.cs) replayed
through the GPU path: header pages + body pages through dcd_build_add_pages, request strings through
the host front end docodo_amd.Search, evaluation by dcd_query_batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _corpus(text, headers, n_pages):
    docs = []
    for name in ("Sample", "Dump"):
        pages = [("0", headers + "Name=%s\n" % name), ("1", text)] + [(str(q + 2), text) for q in range(n_pages - 1)]
        docs.append((name, pages))
    return docs


def _gpu_index(docs):
    import docodo_amd
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE)
    for name, pages in docs:
        b.AddDoc("Test", name)
        texts = [t for _, t in pages]
        off = np.zeros(len(texts) + 1, np.uint64)
        off[1:] = np.cumsum([len(t.encode("utf-16-le")) // 2 for t in texts])
        b.AddPages("".join(texts), off, page_ids=[i for i, _ in pages])
    b.Finish()
    return b


def _oracle_index(o, docs, stemmers):
    ix = o.Index(stemmers=stemmers)
    ob = o.Builder(ix)
    for name, pages in docs:
        ob.add_doc("Test", name)
        for pid, t in pages:
            ob.index_page(pid, t)
    img, lst = ob.save()
    ix.load(img, lst)
    return ix, img, lst


@pytest.fixture(scope="module")
def gpu_nostem(oracle_mod, testtext1, testheaders1):
    docs = _corpus(testtext1, testheaders1, 60)
    b = _gpu_index(docs)
    oix, oimg, olst = _oracle_index(oracle_mod, docs, [])
    return b, b.index_image(), b.list_image(), oix, oimg, olst


def test_header_pages_images_equal_oracle(gpu_nostem):
    """Header page path (Build.cs:485-524): &field markers, value words, coordinate accounting."""
    b, img, lst, oix, oimg, olst = gpu_nostem
    assert lst == olst
    assert img == oimg


def _same(res, ores):
    assert res.Success == ores.Success, (res.Error, ores.Error)
    assert len(res.foundPages) == len(ores.foundPages)
    for (pi, pid, pos, rank), op in zip(res.foundPages, ores.foundPages):
        assert pi == op.page_index and pid == op.id and np.array_equal(pos, op.pos)
        assert abs(rank - op.rank) <= 1e-5 * max(1.0, abs(op.rank))
    assert len(res.foundDocs) == len(ores.foundDocs)
    for (d, name, rank, npg), od in zip(res.foundDocs, ores.foundDocs):
        assert name == od.name and npg == od.n_pages and abs(rank - od.rank) <= 1e-5 * max(1.0, abs(od.rank))


def test_D1_coord_and_D2_D3_on_gpu(gpu_nostem, known, oracle_mod):
    import docodo_amd
    b, _, _, oix, _, _ = gpu_nostem
    ix = docodo_amd.Index.from_images(b.index_image(), b.list_image())
    S = docodo_amd.Search(ix)
    for w, pos in known["D1_coord"]["words"].items():                     # IndexTest.cs:140-157
        r = S(w)
        assert r.Success and len(r.foundPages) == 2 * 60
        for _, pid, p, _ in r.foundPages:
            assert list(p) == pos
        _same(r, oix.search(w))
    r = S(known["D2"]["req"])                                             # IndexTest.cs:180-185
    assert len(r.foundDocs) == 2 and all(d[3] == 60 for d in r.foundDocs)
    _same(r, oix.search(known["D2"]["req"]))
    r = S(known["D3"]["req"])                                             # IndexTest.cs:187-196
    assert len(r.foundDocs) == 1 and r.foundDocs[0][1] == "Test:Dump" and r.foundDocs[0][3] == 60
    assert all(len(p[2]) == 42 for p in r.foundPages) and len(r.foundPages) == 120
    _same(r, oix.search(known["D3"]["req"]))


@pytest.mark.parametrize("key", ["D6", "D7", "D9a", "D9b", "D10", "D11"])
def test_exact_phrases_and_wildcard_on_gpu(gpu_nostem, known, oracle_mod, key):
    """IndexTest.cs:205-220 (dist = 40): the answers that do not depend on the third-party stemmer."""
    import docodo_amd
    b, _, _, oix, _, _ = gpu_nostem
    S = docodo_amd.Search(b_index(b))
    ka = known[key]
    r = S(ka["req"], docodo_amd.SearchOptions(dist=ka["dist"]))
    assert r.Success, r.Error
    if "n_found_pages" in ka:
        assert len(r.foundPages) == ka["n_found_pages"]
    else:
        assert len(r.foundPages[0][2]) == ka["pos_page0"]
    _same(r, oix.search(ka["req"], oracle_mod.SearchOptions(dist=ka["dist"])))


_IX_CACHE = {}


def b_index(b):
    import docodo_amd
    if id(b) not in _IX_CACHE:
        _IX_CACHE[id(b)] = docodo_amd.Index.from_images(b.index_image(), b.list_image())
    return _IX_CACHE[id(b)]


@pytest.mark.parametrize("key", ["D4", "D5", "D8"])
def test_stemmer_dependent_answers_via_explicit_codes(oracle_mod, known, testtext1, testheaders1, key):
    """IndexTest.cs:198-210.  These depend on the host's stemmer ($ladi = lady + ladies): the host computes
    the codes (GetWordCodes, Build.cs:175-247 -- here the oracle's restatement) and feeds (code, coord) through
    the explicit API (Builder.Add), the GPU does vocabulary / sort / pack and evaluates the query."""
    import docodo_amd
    from oracle import porter2
    stem = [("0-9", None), ("а-яё", None), ("a-z", porter2.stem), ("a-zẞäüö", None), ("a-zéâàêèëçîïôûùüÿ", None)]
    docs = _corpus(testtext1, testheaders1, 3)
    oix, oimg, olst = _oracle_index(oracle_mod, docs, stem)
    # replay the oracle index through Builder.Add: every (key, coordinate) pair, coordinate-ascending
    posts = []
    for k in oix.keys():
        for c in oracle_mod.seq_decode(oix.get(k)).tolist():
            posts.append((c, k))
    posts.sort()
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE)
    b.AddDoc("Test", "all")
    for c, k in posts:
        b.Add(k, c)
    b.EndPage("1")
    b.Finish()
    # posting lists are the oracle's; page table comes from the oracle images
    ix = docodo_amd.Index.from_images(b.index_image(), olst)
    S = docodo_amd.Search(ix, get_word_codes=oix.get_word_codes)
    ka = known[key]
    r = S(ka["req"], docodo_amd.SearchOptions(dist=ka["dist"]))
    assert r.Success and len(r.foundPages[0][2]) == ka["pos_page0"]
    _same(r, oix.search(ka["req"], oracle_mod.SearchOptions(dist=ka["dist"])))


_STEM_RANGES = ["0-9", "а-яё", "a-z", "a-zẞäüö", "a-zéâàêèëçîïôûùüÿ"]   # Index.cs:175-181 (ranges of the default stemmers)


def _stemmers():
    from oracle import porter2   # the application's stemmer: here the test infrastructure's Snowball English
    return [(r, porter2.stem if r == "a-z" else None) for r in _STEM_RANGES]


@pytest.fixture(scope="module")
def gpu_stemmed(oracle_mod, testtext1, testheaders1):
    """Default configuration of the reference (stemmers on): the device tokenises and builds the vocabulary, the
    host maps every DISTINCT word to its codes (WordCoder = GetWordCodes), the device emits one posting per
    (token, code).  Header-page value words take the same route, "&field" markers stay explicit codes."""
    import docodo_amd
    docs = _corpus(testtext1, testheaders1, 5)
    coder = docodo_amd.WordCoder(stemmers=_stemmers())
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE, get_word_codes=coder)
    for name, pages in docs:
        b.AddDoc("Test", name)
        texts = [t for _, t in pages]
        off = np.zeros(len(texts) + 1, np.uint64)
        off[1:] = np.cumsum([len(t.encode("utf-16-le")) // 2 for t in texts])
        b.AddPages("".join(texts), off, page_ids=[i for i, _ in pages])
    b.Finish()
    oix, oimg, olst = _oracle_index(oracle_mod, docs, _stemmers())
    return b, coder, oix, oimg, olst


def test_word_codes_images_equal_oracle(gpu_stemmed, oracle_mod):
    """GetWordCodes through dcd_build_words / dcd_build_set_codes: ".index" / ".index.list" bit-exact against the
    oracle built with the same stemmers ($stem codes shared by several words, e.g. lady + ladies -> $ladi)."""
    b, coder, oix, oimg, olst = gpu_stemmed
    assert b.list_image() == olst
    assert b.index_image() == oimg
    assert oix.get("$ladi") is not None and b.stats.n_terms == oix.count


@pytest.mark.parametrize("key", ["D4", "D5", "D8"])
def test_stemmer_dependent_answers_device_build(gpu_stemmed, oracle_mod, known, key):
    """IndexTest.cs:198-210 on an index the DEVICE built with word codes."""
    import docodo_amd
    b, coder, oix, _, _ = gpu_stemmed
    S = docodo_amd.Search(b_index(b), get_word_codes=coder)
    ka = known[key]
    r = S(ka["req"], docodo_amd.SearchOptions(dist=ka["dist"]))
    assert r.Success and len(r.foundPages[0][2]) == ka["pos_page0"]
    _same(r, oix.search(ka["req"], oracle_mod.SearchOptions(dist=ka["dist"])))


def test_word_codes_stop_words_and_identity(oracle_mod):
    """Stop words drop their postings (GetWordCodes returns an empty list, Build.cs:185); dcd_build_words without
    dcd_build_set_codes keeps identity codes; explicit codes (Builder.Add) never pass through GetWordCodes."""
    import docodo_amd
    text = "the quick brown fox jumps over the lazy dog and the lazy cat"
    oix = oracle_mod.Index(stemmers=[])
    ob = oracle_mod.Builder(oix)
    ob.add_doc("s", "d")
    ob.index_page("1", text)
    oimg, olst = ob.save()
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE)
    b.AddDoc("s", "d")
    b.AddPages(text, [0, len(text)], page_ids=["1"])
    words = b.Words()                       # codes mode, no SetCodes -> identity
    assert words == sorted(set(w for w in text.split() if len(w) >= 3))
    b.Finish()
    assert b.index_image() == oimg and b.list_image() == olst
    # stop words + a shared extra code + an explicit code equal to a word
    coder = docodo_amd.WordCoder(stop_words={"the", "and"}, stemmers=[("a-z", lambda w: "laz" if w == "lazy" else w)])
    b2 = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE, get_word_codes=coder)
    b2.AddDoc("s", "d")
    b2.AddPages(text, [0, len(text)], page_ids=["1"])
    b2.Finish()
    ix = b2.Build()
    assert ix.find("the") is None and ix.find("and") is None
    assert ix.term_count(ix.find("lazy")) == 2 and ix.term_count(ix.find("$laz")) == 2
    assert ix.find("$quick") is None and ix.term_count(ix.find("quick")) == 1


@pytest.mark.parametrize("req", ["and -filter:ump", "and -filter:ample,ump", "and -filter:nomatch", "(tupman | old) -filter:est:d.mp",
                                 "and {Name=Dump} -filter:ump", "and {Name=Dump} -filter:ample", "and -filter:dump"])
def test_doc_name_filter(gpu_nostem, oracle_mod, req):
    """-filter: of the request (Search.cs:455-466, 386-396): regexes over the document names, evaluated by the host into
    one byte per document; a filtered-out document never reaches foundDocs, its pages stay in foundPages; the field
    query of CombineSearchResults is evaluated without the filter."""
    import docodo_amd
    b, _, _, oix, _, _ = gpu_nostem
    S = docodo_amd.Search(b_index(b))
    r, o = S(req), oix.search(req)
    assert r.Success == o.Success
    _same(r, o)
    if "nomatch" in req or req.endswith("dump"):
        assert len(r.foundDocs) == 0 and len(r.foundPages) > 0      # names are matched case-sensitively: "Test:Dump"


def test_doc_filter_with_many_docs(oracle_mod):
    """Filtered-out documents between found ones: page runs of the found documents, ranks, the 500-document cap counted
    over FOUND documents only (query_batch level, against the oracle)."""
    import docodo_amd
    from docodo_amd import synth
    n = 3000
    text = synth.corpus_host(n)
    off = np.arange(n + 1, dtype=np.uint64) * synth.DOC_BYTES
    b = docodo_amd.Builder()
    b.AddPages(text, off, docs_per_page=True)
    b.Finish()
    ix = b.Build()
    oix = oracle_mod.Index(stemmers=[])
    oix.load(b.index_image(), b.list_image())
    w = synth.term(1)
    t = ix.find(w)
    for mod in (2, 3, 7):
        mask = np.array([1 if d % mod == 0 else 0 for d in range(n)], np.uint8)
        r = ix.query_batch([docodo_amd.DCD_OP_TERM, t, 255 + len(w)], [0, 3], doc_match=mask)
        units, R = oix.search_word(w)
        o = oix.prepare_result_masked(units, mask)
        docs = r.docs(0)
        assert len(docs) == len(o) and len(docs) == min(500, len(docs))
        for (d, rank, npg), (od, orank, onpg) in zip(docs, o):
            assert d == od and npg == onpg and abs(rank - orank) <= 1e-5 * max(1.0, abs(orank))


def test_index_surface_this_key_and_search(gpu_nostem, known, oracle_mod):
    """Index.this[key] (Index.cs:255-263; the SortedList indexer throws for a missing key) and Index.Search(req)."""
    import docodo_amd
    b, _, _, oix, _, _ = gpu_nostem
    ix = b_index(b)
    assert ix.CanSearch
    s = ix["pickwick"]
    assert np.array_equal(s.units, oix.get("pickwick"))
    with pytest.raises(KeyError):
        ix["zzzzunknown"]
    _same(ix.Search(known["D2"]["req"]), oix.search(known["D2"]["req"]))


def test_close_words_and_suggestions(gpu_nostem):
    """GetCloseWords (Levenshtein scan over the dictionary, Search.cs:168-174) and GetSuggessions (Search.cs:176-188)
    against a plain-Python restatement over the same keys."""
    import docodo_amd
    b, _, _, _, _, _ = gpu_nostem
    ix = b_index(b)
    S = docodo_amd.Search(ix)
    keys = [ix.key(t) for t in range(ix.Count)]

    def lev(s, t):
        if not s:
            return len(t)
        if not t:
            return len(s)
        row = list(range(len(t) + 1))
        for i in range(1, len(s) + 1):
            diag, row[0] = row[0], i
            for j in range(1, len(t) + 1):
                v = min(row[j] + 1, row[j - 1] + 1, diag + (0 if s[i - 1] == t[j - 1] else 1))
                diag, row[j] = row[j], v
        return row[len(t)]

    oix = gpu_nostem[3]           # the oracle over the same index bytes
    for w in ("pikwick", "tupmann", "ld", "", "winkle", "&nam", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz"):
        want = [k for _, _, k in sorted((lev(k, w), i, k) for i, k in enumerate(keys))[:10]]
        assert S.GetCloseWords(w) == want == oix.close_words(w, 10), w
    for req in ("mr pi", "the tu", "x", "snod", "and wi", "&n", "12"):
        last = [p for p in __import__("regex").split(r"\b", req) if p][-1].lower()
        cand = [(-(len(ix.get(t).units)), t) for t, k in enumerate(keys) if len(last) >= 2 and k[0] >= "A" and k.startswith(last) and len(k) > len(last)]
        want = [keys[t][len(last):] for _, t in sorted(cand)[:10]] if len(req) >= 2 else []
        assert S.GetSuggessions(req) == want, req
        if len(req) >= 2 and len(last) >= 2:
            assert want == oix.suggest_tails(last, 10), req
    # GetLikeWords (Search.cs:158-167) as a dictionary kernel == the oracle's scan
    for pat in ("_an_", "t_p", "pick_", "_ick_ic_", "__", "a_", "_", "zz_", "wel_er", "&n_"):
        got = [ix.key(t) for t in ix.like_words(pat, 100)]
        assert got == oix.like_words(pat), pat


def test_search_many_and_micro_batcher(gpu_nostem, known, oracle_mod):
    """Many requests in ONE dcd_query_batch (Search.search_many) and through the serving micro-batcher: the same answers
    as one call per request (incl. field queries, a filter, syntax errors, unknown words)."""
    import threading
    import docodo_amd
    b, _, _, oix, _, _ = gpu_nostem
    S = docodo_amd.Search(b_index(b))
    reqs = [known["D2"]["req"], known["D3"]["req"], "and", "zzzzunknown", "(", "and -filter:ump", "tupman | old", '"mr pickwick"',
            "and {Name=Dump}", "", "pickwick and tupman"] * 3
    one = [S(r) for r in reqs]
    many = S.search_many(reqs)

    def same(a, b_):
        assert a.Success == b_.Success and len(a.foundPages) == len(b_.foundPages) and len(a.foundDocs) == len(b_.foundDocs)
        for x, y in zip(a.foundPages, b_.foundPages):
            assert x[0] == y[0] and x[1] == y[1] and np.array_equal(x[2], y[2]) and x[3] == y[3]
        assert [d[:2] for d in a.foundDocs] == [d[:2] for d in b_.foundDocs]

    for a, m in zip(one, many):
        same(a, m)
    mb = docodo_amd.MicroBatcher(S, max_batch=64, max_wait_ms=20.0)
    futs = [None] * len(reqs)

    def work(lo, hi):
        for i in range(lo, hi):
            futs[i] = mb.submit(reqs[i])

    n = len(reqs)
    th = [threading.Thread(target=work, args=(j * n // 3, (j + 1) * n // 3)) for j in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    got = [f.result(timeout=60) for f in futs]
    mb.close()
    for a, g in zip(one, got):
        same(a, g)
    assert mb.batches < len(reqs)            # requests were grouped


def test_latin1_header_pages_and_edge_lines(oracle_mod):
    """Header pages in a byte (Latin-1) build, odd header lines: short names (skipped), no trailing newline,
    CR LF line ends, '=' inside the value, digits, a line without '=' after a valid one (aborts the page
    without EndPage, Build.cs:537-540), plus body pages after each header."""
    import docodo_amd
    hdrs = ["Name=Alpha Beta\nSize=190\nab=skipped value\nTitle=Charles Diskense=ignored tail\n",
            "Source=Test\r\nKeywords=één twee_drie 42abc",      # no trailing newline, CR LF, Latin-1 letters, '_' and digits
            "Name=ok line\nbrokenlinewithoutequals\nName=never reached\n"]
    body = "The quick brown fox; jumps over the lazy dog. ÀÉÎ naïve 12345 a1b2"
    docs = []
    for d, h in enumerate(hdrs):
        docs.append(("doc%d" % d, [("0", h), ("1", body), ("2", body * 3)]))
    oix = oracle_mod.Index(stemmers=[])
    ob = oracle_mod.Builder(oix)
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_LATIN1)
    for name, pages in docs:
        ob.add_doc("src", name)
        b.AddDoc("src", name)
        for pid, t in pages:
            ob.index_page(pid, t)
        texts = [t for _, t in pages]
        off = np.zeros(len(texts) + 1, np.uint64)
        off[1:] = np.cumsum([len(t) for t in texts])
        b.AddPages("".join(texts), off, page_ids=[i for i, _ in pages])
    b.Finish()
    oi, ol = ob.save()
    assert b.list_image() == ol
    assert b.index_image() == oi


def test_query_edge_cases(gpu_nostem, oracle_mod):
    """Empty batch, unknown words, a query whose program is a single missing term, error status for a
    coordinate beyond the last page (ErrorSearchResult in the reference, Search.cs:609-613)."""
    import docodo_amd
    b, img, lst, oix, _, _ = gpu_nostem
    ix = b_index(b)
    r = ix.query_batch([], [0])
    assert r.n_queries == 0
    S = docodo_amd.Search(ix)
    for req in ("zzzzunknown", "zzzzunknown and", "and zzzzunknown", "zzzz | and", "(", "and )", "", "a b"):
        _same(S(req), oix.search(req)) if S(req).Success else None
        assert S(req).Success == oix.search(req).Success, req
    # page table that stops before the postings: every query hits "coordinate beyond the last page"
    short_list = lst[:lst.index(b"Test:Dump") - 8 - 1] if b"Test:Dump" in lst else lst
    ix2 = docodo_amd.Index.from_images(img, short_list)
    oix2 = oracle_mod.Index(stemmers=[])
    oix2.load(img, short_list)
    t = ix2.find("and")
    r = ix2.query_batch([docodo_amd.DCD_OP_TERM, t, 258], [0, 3])
    assert (r.status[0] != 0) == (not oix2.search("and").Success)

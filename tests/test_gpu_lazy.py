"""On-disk postings -- Index(path, InMemory: false), Index.cs:133-136, 198-216, 346-348: the dictionary and the page table are
loaded, the posting lists stay in ".index" and are read when a key or a query batch asks for them.  Same results as the
in-memory handle and as the oracle on the known-answer corpus and on a synthetic index; the file is really read lazily."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _known_corpus_dir(tmp_path, testtext1, testheaders1, n_pages=3):
    import docodo_amd
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE)
    for name in ("Sample", "Dump"):
        b.AddDoc("Test", name)
        pages = [testheaders1 + "Name=%s\n" % name] + [testtext1] * n_pages
        off = np.zeros(len(pages) + 1, np.uint64)
        off[1:] = np.cumsum([len(p.encode("utf-16-le")) // 2 for p in pages])
        b.AddPages("".join(pages), off, page_ids=["0"] + [str(i + 1) for i in range(n_pages)])
    b.Finish()
    b.Save(str(tmp_path))
    return b


def test_known_answers_on_disk(tmp_path, testtext1, testheaders1, oracle_mod):
    """D1 / D2 / D6 / D9 of SURVEY Appendix D (IndexTest.cs:126-217) through Index.Search on an on-disk index == the
    in-memory index == the oracle."""
    import docodo_amd
    from test_gpu_query import _check_result
    b = _known_corpus_dir(tmp_path, testtext1, testheaders1)
    mem = docodo_amd.Index.Load(str(tmp_path))
    disk = docodo_amd.Index.LoadOnDisk(str(tmp_path))
    assert disk.Count == mem.Count and disk.MaxCoord == mem.MaxCoord and disk.n_pages == mem.n_pages and disk.n_docs == mem.n_docs
    assert disk.lazy_stats() == (0, 0, 0)                       # nothing read yet: only keys and stubs (Index.cs:346-348)
    # this[key] = LoadSequence: the list's units from its file position
    for w in ("and", "tupman", "everybody", "old", "&name"):
        t = disk.find(w)
        assert t == mem.find(w) and t is not None
        assert np.array_equal(disk.get(t).units, mem.get(t).units)
        assert disk.term_count(t) == mem.term_count(t)
    lists, nbytes, resident = disk.lazy_stats()
    assert lists == 10 and nbytes > 0 and resident == 0
    oix = oracle_mod.Index(stemmers=[])
    oix.load(open(os.path.join(str(tmp_path), ".index"), "rb").read(), open(os.path.join(str(tmp_path), ".index.list"), "rb").read())
    opt40 = docodo_amd.SearchOptions(dist=40)
    for req, opt in (("and", None), ("and (tupman|old)", None), ("and (tupman|old) {Name=Dump}", None), ('"old lady"', opt40),
                     ('"old ladies were"', opt40), ("lady old", opt40), ("?an?", opt40), ("nosuchword and", None)):
        rd, rm = disk.Search(req, opt), mem.Search(req, opt)
        oo = oix.search(req, oracle_mod.SearchOptions(dist=opt.dist) if opt else None)
        assert rd.Success == rm.Success == oo.Success
        # (docodo_amd results: foundPages = (page index, page id, positions, rank), foundDocs = (doc index, name, rank, pages))
        assert [(p[0], list(p[2])) for p in rd.foundPages] == [(p[0], list(p[2])) for p in rm.foundPages] == \
               [(p.page_index, list(p.pos)) for p in oo.foundPages], req
        assert [d[1] for d in rd.foundDocs] == [d[1] for d in rm.foundDocs]
        for a, c in zip(rd.foundPages, oo.foundPages):
            assert abs(a[3] - c.rank) <= 1e-5 * max(1.0, abs(c.rank))
    assert disk.lazy_stats()[2] > 0                              # a resident subset exists now
    # what needs all postings in memory fails loudly
    for fn in (disk.index_image, lambda: disk.Save(str(tmp_path / "x"))):
        with pytest.raises(docodo_amd.DocodoError) as e:
            fn()
        assert e.value.code == docodo_amd._lib.DCD_E_STATE


def test_opening_an_on_disk_index_reads_the_file_about_once(tmp_path):
    """dcd_index_open_lazy scans keys and stubs with ONE sliding window (round 3 read a fresh 64 KiB window per record: V x 64 KiB).
    A dictionary of 60 k short lists: the bytes read by the process while opening stay within twice the file size."""
    import struct
    import docodo_amd

    def rec(k, units):
        b = k.encode("utf-8")
        return bytes([len(b)]) + b + struct.pack("<i", len(units)) + np.array(units, np.uint16).tobytes()

    V = 60000
    keys = sorted("k%07d" % i for i in range(V))
    img = struct.pack("<Q", 10 * V) + b"".join(rec(k, [i % 30000 + 1, 7]) for i, k in enumerate(keys))
    d = tmp_path / "ix"
    d.mkdir()
    (d / ".index").write_bytes(img)
    (d / ".index.list").write_bytes(b"")

    def rchar():
        for line in open("/proc/self/io"):
            if line.startswith("rchar:"):
                return int(line.split()[1])
        return 0
    r0 = rchar()
    ix = docodo_amd.Index.LoadOnDisk(str(d))
    read = rchar() - r0
    assert ix.Count == V and ix.find("k0000007") is not None
    assert read < 2 * len(img) + (1 << 20), (read, len(img))
    got = ix.get(ix.find(keys[12345]))
    assert got.units.tolist() == [12345 % 30000 + 1, 7]
    ix.close()


def test_synthetic_index_on_disk_batches(tmp_path, oracle_mod):
    """20 000 synthetic docs: batches of 3-term AND / phrase / OR programs on the on-disk handle == the in-memory handle, the
    resident subset grows with the union of the batches while it fits the budget and is replaced when it does not."""
    import docodo_amd
    from docodo_amd import synth
    n = 20000
    text = synth.corpus_host(n)
    b = docodo_amd.Builder()
    b.AddPages(text, np.arange(n + 1, dtype=np.uint64) * 1024, docs_per_page=True)
    b.Finish()
    b.Save(str(tmp_path))
    mem = b.Build()
    T, A, O = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND, docodo_amd.DCD_OP_OR

    def batch(seed, nq):
        prog, off = [], [0]
        for qi, row in enumerate(synth.query_ranks(nq, 3, lo_rank=10, seed=seed)):
            ts = [mem.find(synth.term(int(k))) for k in row]
            if any(t is None for t in ts):
                continue
            exact = qi % 3 == 1
            for k, (r, t) in enumerate(zip(row, ts)):
                w = synth.term(int(r))
                prog += [T, t, -(len(w) + 4) if exact else 255 + len(w)]
                if k:
                    prog.append(O if (qi % 3 == 2 and k == 2) else A)
            off.append(len(prog))
        return prog, off

    def same(a, c):
        names = ["status", "page_off", "page_index", "pos_off", "pos", "doc_off", "doc_index", "doc_n_pages"]
        assert (a.seq_off is None) == (c.seq_off is None)
        if a.seq_off is not None:
            names += ["seq_off", "seq_units", "seq_R"]
        for name in names:
            assert np.array_equal(getattr(a, name), getattr(c, name)), name
        assert np.allclose(a.page_rank, c.page_rank, rtol=1e-6) and np.allclose(a.doc_rank, c.doc_rank, rtol=1e-6)

    disk = docodo_amd.Index.LoadOnDisk(str(tmp_path), resident_bytes=64 << 20)
    p1, o1 = batch(1, 60)
    same(disk.query_batch(p1, o1, want_sequences=True), mem.query_batch(p1, o1, want_sequences=True))
    l1, b1, r1 = disk.lazy_stats()
    assert 0 < r1 <= 180 and l1 == r1
    same(disk.query_batch(p1, o1, want_sequences=True), mem.query_batch(p1, o1, want_sequences=True))
    assert disk.lazy_stats() == (l1, b1, r1)                     # the same terms again: nothing is read
    p2, o2 = batch(2, 60)
    same(disk.query_batch(p2, o2, want_sequences=True), mem.query_batch(p2, o2, want_sequences=True))
    l2, b2, r2 = disk.lazy_stats()
    assert r2 > r1 and l2 == l1 + r2                             # the union became resident (one rebuild reads every list of it)
    # a budget that holds almost nothing: every batch brings its own subset
    tiny = docodo_amd.Index.LoadOnDisk(str(tmp_path), resident_bytes=1 << 12)
    same(tiny.query_batch(p1, o1, want_sequences=True), mem.query_batch(p1, o1, want_sequences=True))
    ra = tiny.lazy_stats()[2]
    same(tiny.query_batch(p2, o2, want_sequences=True), mem.query_batch(p2, o2, want_sequences=True))
    assert tiny.lazy_stats()[2] <= max(ra, r2 - 1)
    # single-term query whose final sequence is an index term itself + an empty batch
    w = synth.term(3)
    q = [T, mem.find(w), 255 + len(w)]
    same(disk.query_batch(q, [0, 3], want_sequences=True), mem.query_batch(q, [0, 3], want_sequences=True))
    same(disk.query_batch([], [0]), mem.query_batch([], [0]))
    # dictionary scans work on the on-disk handle like on any other
    assert disk.close_words("aaaab", 5) == mem.close_words("aaaab", 5)
    assert disk.suggest("aa", 8) == mem.suggest("aa", 8)


@pytest.mark.parametrize("spill", ["disk", "host"])
def test_out_of_core_build_equals_one_shot(tmp_path, spill, oracle_mod):
    """dist.OutOfCoreBuilder -- the temp-index flush + MergeAll equivalent with a host / disk tier (Build.cs:335-340,
    Index.cs:537-573): 5 segments x 6 hash partitions spilled, merged partition by partition on the device, the sorted runs
    merged on the host: ".index" / ".index.list" byte-identical to the one-shot build (and therefore to the oracle's), and the
    on-disk handle answers queries like the in-memory one."""
    import torch
    import docodo_amd
    from docodo_amd import dist, synth
    n = 9000
    d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
    synth.corpus_dev(d.data_ptr(), n)
    full = docodo_amd.Builder()
    full.AddUniformPagesDev(d.data_ptr(), n, 1024)
    full.Finish()
    ob = dist.OutOfCoreBuilder(1024, str(tmp_path / "work"), partitions=6, max_segment_chars=2000 * 1024, spill=spill)
    ob.add_pages_dev(d.data_ptr(), 5000)
    ob.add_pages_dev(d.data_ptr() + 5000 * 1024, 4000)
    assert len(ob.segments) == 5 and ob.spilled_bytes > 0
    if spill == "disk":
        assert sorted(f for f in os.listdir(str(tmp_path / "work")) if f.startswith("seg0."))  == ["seg0.keys.npy", "seg0.meta.npy", "seg0.units.npy"]
    ix = ob.finish(str(tmp_path / "out"))
    img = open(str(tmp_path / "out" / ".index"), "rb").read()
    lst = open(str(tmp_path / "out" / ".index.list"), "rb").read()
    assert img == full.index_image() and lst == full.list_image()
    text = synth.corpus_host(n)
    oix = oracle_mod.Index(stemmers=[])
    obld = oracle_mod.Builder(oix)
    obld.index_docs_latin1(text, np.arange(n + 1, dtype=np.uint64) * 1024)
    assert obld.save() == (img, lst)
    mem = full.Build()
    T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
    prog, off = [], [0]
    for row in synth.query_ranks(40, 2, lo_rank=5, seed=9):
        ts = [ix.find(synth.term(int(k))) for k in row]
        if any(t is None for t in ts):
            continue
        for k, (r, t) in enumerate(zip(row, ts)):
            prog += [T, t, 255 + len(synth.term(int(r)))]
            if k:
                prog.append(A)
        off.append(len(prog))
    a, c = ix.query_batch(prog, off), mem.query_batch(prog, off)
    for name in ("status", "page_off", "page_index", "pos_off", "pos", "doc_off", "doc_index"):
        assert np.array_equal(getattr(a, name), getattr(c, name)), name

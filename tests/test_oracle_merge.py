"""The oracle's restatement of the reference's out-of-core machinery -- temp-index flush (Build.cs:335-340), MergeFiles
(Index.cs:577-689), MergeAll (Index.cs:537-573) -- pinned by the property the reference relies on: flushing + merging leaves
the same ".index" as never flushing; merging doc-range tasks with their coordinate shifts leaves the one-task index
(two inputs: the reference's default MaxDegreeOfParallelism).  The >= 3-input quirk of MergeFiles is pinned as a quirk."""
import struct

import numpy as np


def _corpus(n):
    from docodo_amd import synth
    return synth.corpus_host(n), np.arange(n + 1, dtype=np.uint64) * 1024


def _one_task(o, text, off, first_doc=0):
    ix = o.Index(stemmers=[])
    b = o.Builder(ix)
    b.index_docs_latin1(text, off, first_doc=first_doc)
    return b


def test_flush_and_merge_all_equal_no_flush(oracle_mod):
    o = oracle_mod
    text, off = _corpus(300)
    full = _one_task(o, text, off)
    want, _ = full.save()
    N = full.total_count
    for max_items in (1000, 5000, 17000, 30000):      # 41 k postings: 41, 8, 2 and 1 flushes -> up to three MergeAll rounds
        ix = o.Index(stemmers=[])
        b = o.Builder(ix)
        b.set_max_items(max_items)
        b.index_docs_latin1(text, off)
        assert b.n_tmp == N // (max_items + 1)        # a flush whenever TotalCount EXCEEDS MaxItems (Build.cs:335)
        assert b.max_coord == full.max_coord          # (maxCoord survives Clear())
        files = b.tmp_images() + [b.save()[0]]        # Build.cs:541: the rest is saved as the last N.tmpind
        # every temp file starts its lists from scratch: first value absolute (Clear() drops the IndexSequence.Builders)
        t = o.Index(stemmers=[])
        t.load(files[1], b"")
        k = t.keys()[0]
        assert o.seq_decode(t.get(k))[0] >= o.seq_decode(full_list(o, want, k))[0]
        # pairwise MergeFiles of the temp files (creation order): the flush-free index, always
        assert o.merge_pairwise(files) == want, max_items
        got = b.save_merged()                         # the reference's MergeAll: groups of FIVE
        if len(files) <= 2:
            assert got == want, max_items
        else:
            assert got[:8] == want[:8]                # (>= 3 inputs: MergeFiles' quirk may strike; see the quirk test)


def full_list(o, img, k):
    t = o.Index(stemmers=[])
    t.load(img, b"")
    return t.get(k)


def test_two_tasks_with_shift_equal_one_task(oracle_mod):
    """MergeIndexes -> MergeFiles(shift_coords = true) over two index tasks (Index.cs:700, 587-593, 665-674).  The
    reference shifts task 1 by task 0's MaxCoord (coordinate of its LAST WORD, Appendix E.3); with the text-length base
    instead, the merged lists are the one-task lists."""
    o = oracle_mod
    text, off = _corpus(400)
    want, _ = _one_task(o, text, off).save()
    a = _one_task(o, text[:150 * 1024], off[:151]).save()[0]
    b = _one_task(o, text[150 * 1024:], off[150:] - off[150], first_doc=150).save()[0]
    got = o.merge_files([a, b], True, [0, 150 * 1024])
    # header: MaxCoord = max over the inputs (Index.cs:596) -- NOT shifted (Appendix E.5)
    assert struct.unpack("<Q", got[:8])[0] == max(struct.unpack("<Q", a[:8])[0], struct.unpack("<Q", b[:8])[0])
    assert got[8:] == want[8:]
    # the reference's own shift (MaxCoord of task 0 = start of its last word) moves task 1 down by the tail of task 0's text
    quirk = o.merge_files([a, b], True)
    assert quirk != got
    ixq, ixw = o.Index(stemmers=[]), o.Index(stemmers=[])
    ixq.load(quirk, b"")
    ixw.load(got, b"")
    k = ixw.keys()[0]
    mc0 = struct.unpack("<Q", a[:8])[0]
    vq, vw = o.seq_decode(ixq.get(k)), o.seq_decode(ixw.get(k))
    n0 = int(np.searchsorted(vw, 150 * 1024))
    assert np.array_equal(vq[:n0], vw[:n0]) and np.array_equal(vq[n0:] + np.uint64(150 * 1024 - mc0), vw[n0:])


def test_pairwise_merge_of_many_tasks_equals_one_task(oracle_mod):
    o = oracle_mod
    text, off = _corpus(500)
    want, _ = _one_task(o, text, off).save()
    bounds = [0, 60, 61, 200, 333, 400, 500]
    imgs, shifts = [], []
    for lo, hi in zip(bounds, bounds[1:]):
        imgs.append(_one_task(o, text[lo * 1024:hi * 1024], off[lo:hi + 1] - off[lo], first_doc=lo).save()[0])
        shifts.append(lo * 1024)
    got = o.merge_pairwise(imgs, shifts)
    # header = max of the (unshifted) input headers; the lists are the one-task lists
    assert got[8:] == want[8:]


def test_merge_files_quirk_with_three_inputs(oracle_mod):
    """Index.cs:655-660 marks list[q] whenever it equals list[q-1]: heads  a < b == b  append the second "b" under "a"."""
    o = oracle_mod

    def img(mx, entries):
        out = [struct.pack("<Q", mx)]
        for k, vals in entries:
            u = o.seq_encode(np.array(vals, np.uint64))
            out.append(bytes([len(k)]) + k.encode() + struct.pack("<i", u.size) + u.tobytes())
        return b"".join(out)

    f0 = img(10, [("aaa", [1, 2])])
    f1 = img(20, [("bbb", [11])])
    f2 = img(30, [("bbb", [21])])
    two = o.merge_files([f1, f2], False)
    ix = o.Index(stemmers=[])
    ix.load(two, b"")
    assert ix.keys() == ["bbb"] and o.seq_decode(ix.get("bbb")).tolist() == [11, 21]
    three = o.merge_files([f0, f1, f2], False)
    ix.load(three, b"")
    assert o.seq_decode(ix.get("aaa")).tolist() == [1, 2, 21]          # the quirk: f2's "bbb" landed under "aaa"
    assert o.seq_decode(ix.get("bbb")).tolist() == [11]
    ix.load(o.merge_pairwise([f0, f1, f2]), b"")
    assert o.seq_decode(ix.get("aaa")).tolist() == [1, 2] and o.seq_decode(ix.get("bbb")).tolist() == [11, 21]

"""Doc-range shard merge on one device: G fake shards (each built with its global coord_base), exported,
routed like the all-to-all would, merged per owner -- must reproduce the one-shot build's posting lists
bit for bit (and therefore the oracle's, which the one-shot build is checked against)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _build(text, n_docs, first_doc=0):
    import docodo_amd
    from docodo_amd import synth
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    b = docodo_amd.Builder(coord_base=first_doc * synth.DOC_BYTES)
    b.AddPages(text, off, docs_per_page=True, first_doc=first_doc)
    b.Finish()
    return b


@pytest.mark.parametrize("G,n_docs", [(1, 500), (2, 3000), (3, 6001), (8, 4000)])
def test_fake_shards_equal_one_shot_build(oracle_mod, G, n_docs):
    import docodo_amd
    from docodo_amd import dist, synth
    text = synth.corpus_host(n_docs)
    full = _build(text, n_docs)
    full_img, full_lst = full.index_image(), full.list_image()
    ref = full.Build()
    bounds = [n_docs * g // G for g in range(G + 1)]
    shards = [_build(text[bounds[g] * 1024:bounds[g + 1] * 1024], bounds[g + 1] - bounds[g], first_doc=bounds[g]) for g in range(G)]
    exports = [dist.export_shard(b, G) for b in shards]
    owned = [dist.local_merge(exports, o) for o in range(G)]
    assert sum(ix.Count for ix in owned) == ref.Count
    assert all(ix.MaxCoord == ref.MaxCoord for ix in owned)
    seen = 0
    for ix in owned:
        prev = None
        for t in range(ix.Count):
            k = ix.key(t)
            assert prev is None or prev < k          # ordinal order inside an owner (ASCII keys)
            prev = k
            rt = ref.find(k)
            assert rt is not None
            if t % 37 == 0 or ix.term_count(t) > 1000:   # all big terms, a sample of the rest
                assert np.array_equal(ix.get(t).units, ref.get(rt).units), k
            assert ix.term_count(t) == ref.term_count(rt)
            seen += 1
    assert seen == ref.Count
    # and the one-shot build is the oracle's
    oix = oracle_mod.Index(stemmers=[])
    ob = oracle_mod.Builder(oix)
    ob.index_docs_latin1(text, np.arange(n_docs + 1, dtype=np.uint64) * 1024)
    oi, ol = ob.save()
    assert full_img == oi and full_lst == ol


@pytest.mark.parametrize("G,n_docs", [(1, 700), (3, 5000), (4, 40000)])
def test_shard_only_builds_export_the_same_buffers(G, n_docs):
    """DCD_BUILD_SHARD_ONLY (Builder(shard_only=True)): Finish stops in front of the ".index" image and the term-ordered posting
    array, the export reads the units where the pack kernel left them.  Same export buffers as a full build, byte for byte;
    what needs the image or the index fails loudly."""
    import docodo_amd
    from docodo_amd import dist, synth
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    outs = []
    for shard_only in (False, True):
        b = docodo_amd.Builder(coord_base=5 * synth.DOC_BYTES, shard_only=shard_only)
        b.AddPages(text, off, docs_per_page=True, first_doc=5)
        st = b.Finish()
        if shard_only:
            assert st.index_bytes == 0
            with pytest.raises(docodo_amd.DocodoError):
                b.index_image()
            with pytest.raises(docodo_amd.DocodoError):
                b.Build()
        e = dist.export_shard(b, G)
        outs.append((e.term_count.tolist(), e.key_count.tolist(), e.unit_count.tolist(), e.meta.cpu().numpy().tobytes(),
                     e.keys.cpu().numpy().tobytes(), e.units.cpu().numpy().tobytes(), e.max_coord, st.n_postings, st.n_terms, st.n_units))
        b.close()
    assert outs[0] == outs[1]


@pytest.mark.parametrize("G,n_docs", [(2, 2000), (5, 6003)])
def test_gathered_partitions_give_the_one_shot_index_file(G, n_docs):
    """Doc-range shards -> per-owner merge (hash partitions) -> gather of the partitions on one GPU: ".index" and
    ".index.list" are byte-identical to what a single builder writes for the whole corpus."""
    import docodo_amd
    from docodo_amd import dist, synth
    text = synth.corpus_host(n_docs)
    full = _build(text, n_docs)
    bounds = [n_docs * g // G for g in range(G + 1)]
    shards = [_build(text[bounds[g] * 1024:bounds[g + 1] * 1024], bounds[g + 1] - bounds[g], first_doc=bounds[g]) for g in range(G)]
    exports = [dist.export_shard(b, G) for b in shards]
    owned = [dist.local_merge(exports, o) for o in range(G)]
    whole = dist.local_gather(owned)
    assert whole.Count == full.stats.n_terms and whole.MaxCoord == full.stats.max_coord
    assert whole.index_image() == full.index_image()
    whole.set_list_image(b"".join(b.list_image() for b in shards))      # shards are split at document borders
    assert whole.list_image() == full.list_image()
    # the gathered index is searchable like the one-shot one
    ref = full.Build()
    w = synth.term(7)
    a = whole.query_batch([docodo_amd.DCD_OP_TERM, whole.find(w), 255 + len(w)], [0, 3])
    b = ref.query_batch([docodo_amd.DCD_OP_TERM, ref.find(w), 255 + len(w)], [0, 3])
    for name in ("page_index", "pos", "page_rank", "doc_index", "doc_rank"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_segmented_builder_equals_one_shot():
    """SegmentedBuilder (the temp-index flush + MergeAll equivalent, Build.cs:335-340 / Index.cs:537-689): a device
    corpus cut into 4 segments -> the one-shot ".index"."""
    import torch
    import docodo_amd
    from docodo_amd import dist, synth
    n_docs = 7000
    d = torch.empty(n_docs * 1024, dtype=torch.uint8, device="cuda")
    synth.corpus_dev(d.data_ptr(), n_docs)
    full = docodo_amd.Builder()
    full.AddUniformPagesDev(d.data_ptr(), n_docs, 1024)
    full.Finish()
    sb = dist.SegmentedBuilder(1024, max_segment_chars=2000 * 1024)
    sb.add_pages_dev(d.data_ptr(), 3000)
    sb.add_pages_dev(d.data_ptr() + 3000 * 1024, 4000)
    assert len(sb.exports) == 4
    ix = sb.finish()
    assert ix.index_image() == full.index_image()
    assert ix.list_image() == full.list_image()


def test_merged_index_answers_queries(oracle_mod):
    """A merged (hash-partitioned) sub-index + the global page table answers its terms like the oracle."""
    import docodo_amd
    from docodo_amd import dist, synth
    n_docs, G = 5000, 4
    text = synth.corpus_host(n_docs)
    bounds = [n_docs * g // G for g in range(G + 1)]
    shards = [_build(text[bounds[g] * 1024:bounds[g + 1] * 1024], bounds[g + 1] - bounds[g], first_doc=bounds[g]) for g in range(G)]
    exports = [dist.export_shard(b, G) for b in shards]
    oix = oracle_mod.Index(stemmers=[])
    ob = oracle_mod.Builder(oix)
    ob.index_docs_latin1(text, np.arange(n_docs + 1, dtype=np.uint64) * 1024)
    oix.load(*ob.save())
    ix = dist.local_merge(exports, 1)
    dist.set_uniform_pages(ix, n_docs, 1024)
    checked = 0
    for t in range(0, ix.Count, max(1, ix.Count // 25)):
        w = ix.key(t)
        res = ix.query_batch([docodo_amd.DCD_OP_TERM, t, 255 + len(w)], [0, 3])
        o = oix.search(w)
        pages = res.pages(0)
        assert len(pages) == len(o.foundPages)
        for (pi, pos, rank), op in zip(pages, o.foundPages):
            assert pi == op.page_index and np.array_equal(pos, op.pos) and abs(rank - op.rank) <= 1e-5 * max(1, abs(op.rank))
        checked += 1
    assert checked >= 20


def test_two_phase_merge_equals_one_call():
    """dcd_merge_begin (segment table + keys only) + dcd_merge_finish (units) == dcd_merge_shards; an unfinished merge
    is dropped with dcd_merge_free."""
    from docodo_amd import dist, synth
    n_docs, G = 5000, 3
    text = synth.corpus_host(n_docs)
    bounds = [n_docs * g // G for g in range(G + 1)]
    shards = [_build(text[bounds[g] * 1024:bounds[g + 1] * 1024], bounds[g + 1] - bounds[g], first_doc=bounds[g]) for g in range(G)]
    exports = [dist.export_shard(b, G) for b in shards]
    for o in range(G):
        a = dist.local_merge(exports, o)
        b = dist.local_merge(exports, o, two_phase=True)
        assert a.Count == b.Count and a.MaxCoord == b.MaxCoord
        assert a.index_image() == b.index_image()
    e = exports[0]
    import numpy as np
    one = np.array([1], np.uint64)
    pend = dist.PendingMerge(e.term_count[:1], e.key_count[:1], e.unit_count[:1], e.meta, e.keys, e.max_coord)
    del pend                       # never finished: freed by dcd_merge_free
    empty = dist.PendingMerge(one * 0, one * 0, one * 0, e.meta[:0], e.keys[:0], 0).finish(e.units[:0])
    assert empty.Count == 0


def test_distributed_merge_over_rccl_world1():
    """The real RCCL path (nccl backend, async all-to-alls overlapped with dcd_merge_begin) with world_size 1."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DCD_PORT="29671")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "rccl_world1_worker.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "rccl world1 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_three_ranks_on_one_gpu_give_the_one_shot_index():
    """Three processes (ragged doc ranges, global coordinates) build, exchange, merge and gather on ONE GPU over gloo: the
    gathered `.index` / `.index.list` are byte-identical to a single builder's over the whole corpus."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", "29683", os.path.join(root, "tests", "helpers", "gpu_dist_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and all("rank %d ok" % k in r.stdout for k in range(3)), r.stdout[-2000:] + r.stderr[-4000:]


def test_balanced_owner_table_and_export(oracle_mod):
    """Owners balanced by posting units (dcd_build_export_heavy + dist.assign_heavy + dcd_build_export_plan_balanced): the heavy
    terms of a Zipf vocabulary move to the least loaded owners, every other term stays with hash(key) % n_owners, the merged +
    gathered ".index" is still the one-shot one, and the largest owner share drops."""
    import docodo_amd
    from docodo_amd import dist, synth
    n_docs, G = 12000, 8
    text = synth.corpus_host(n_docs)
    full = _build(text, n_docs)
    bounds = [n_docs * g // G for g in range(G + 1)]
    shards = [_build(text[bounds[g] * 1024:bounds[g + 1] * 1024], bounds[g + 1] - bounds[g], first_doc=bounds[g]) for g in range(G)]
    lists, totals = [], []
    for b in shards:
        U = int(b.stats.n_units)
        fp, un = dist.heavy_terms(b, -(-U // (8 * G)), 8 * G)
        assert 1 <= fp.size <= 8 * G and np.all(un >= U // (8 * G)) and np.all(fp[1:] > fp[:-1])
        lists.append((fp, un))
        totals.append(U)
    with pytest.raises(docodo_amd.DocodoError):                   # more heavy terms than the caller made room for
        dist.heavy_terms(shards[0], 1, 4)
    ofp, oow, thr = dist.assign_heavy(lists, totals, G)
    ofp2, oow2, thr2 = dist.assign_heavy(lists[::-1], totals[::-1], G)  # (a pure function of the gathered data, whatever its order)
    assert np.array_equal(ofp, ofp2) and np.array_equal(oow, oow2) and np.array_equal(thr, thr2)
    assert thr[-1] == 0xFFFFFFFF and np.all(thr[1:] >= thr[:-1])
    plain = [dist.export_shard(b, G) for b in shards]
    bal = [dist.export_shard(b, G, overrides=(ofp, oow, thr)) for b in shards]
    # heavy terms re-assigned, light terms by plain hash: valid too, just less even
    mid = [dist.export_shard(b, G, overrides=(ofp, oow, None)) for b in shards]
    assert dist.local_gather([dist.local_merge(mid, o) for o in range(G)]).index_image() == full.index_image()
    share = lambda ex: np.sum([e.unit_count for e in ex], axis=0).astype(np.float64)
    sp, sb = share(plain), share(bal)
    assert sp.sum() == sb.sum()
    assert sb.max() / sb.mean() < 1.15 < sp.max() / sp.mean()     # hash: the owner of term 1 holds ~1.9 x the mean (term 1 alone is 1.08 x)
    owned = [dist.local_merge(bal, o) for o in range(G)]
    whole = dist.local_gather(owned)
    assert whole.index_image() == full.index_image()
    # a term is owned by exactly one owner, the one the table (or the hash) names on EVERY shard
    top = synth.term(1)
    holders = [o for o, ix in enumerate(owned) if ix.find(top) is not None]
    assert len(holders) == 1

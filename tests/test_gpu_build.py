"""GPU parity of the build path: the .index / .index.list byte images produced by the HIP pipeline
must equal the CPU oracle's (bit-exact), on the synthetic Appendix-F corpora and on adversarial
Latin-1 / UTF-16 texts (page boundaries inside words, tile-boundary tokens, over-long words, case)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_images(o, pages_text, ids=None, docs=None, latin1=True):
    """pages_text: list of str (one per page).  docs: list of (source, name, n_pages) or None = doc per page."""
    ix = o.Index(stemmers=[])
    b = o.Builder(ix)
    p = 0
    if docs is None:
        for i, t in enumerate(pages_text):
            b.add_doc("syn", str(i))
            b.index_page(ids[i] if ids else "1", t)
    else:
        for src, name, n in docs:
            b.add_doc(src, name)
            for _ in range(n):
                b.index_page(ids[p] if ids else "1", pages_text[p])
                p += 1
    return b.save(), b


def _gpu_images(pages_text, ids=None, docs=None, encoding=None, **kw):
    import docodo_amd
    enc = encoding or docodo_amd.DCD_ENC_LATIN1
    b = docodo_amd.Builder(encoding=enc, **kw)
    text = "".join(pages_text)
    off = np.zeros(len(pages_text) + 1, np.uint64)
    off[1:] = np.cumsum([len(t.encode("utf-16-le")) // 2 for t in pages_text])
    if docs is None:
        b.AddPages(text, off, page_ids=ids, docs_per_page=True)
    else:
        p = 0
        for src, name, n in docs:
            b.AddDoc(src, name)
            sub = pages_text[p:p + n]
            so = np.zeros(n + 1, np.uint64)
            so[1:] = np.cumsum([len(t.encode("utf-16-le")) // 2 for t in sub])
            b.AddPages("".join(sub), so, page_ids=ids[p:p + n] if ids else None)
            p += n
    st = b.Finish()
    return (b.index_image(), b.list_image()), b, st


@pytest.mark.parametrize("n_docs", [1, 7, 2000, 20000])
def test_synthetic_corpus_images(oracle_mod, n_docs):
    import docodo_amd
    from docodo_amd import synth
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    b = docodo_amd.Builder()
    b.AddPages(text, off, docs_per_page=True)
    st = b.Finish()
    oix = oracle_mod.Index(stemmers=[])
    ob = oracle_mod.Builder(oix)
    ob.index_docs_latin1(text, off)
    oi, ol = ob.save()
    assert st.n_postings == ob.total_count and st.n_terms == ob.n_terms and st.max_coord == ob.max_coord
    assert b.list_image() == ol
    assert b.index_image() == oi


def test_synthetic_device_generator_matches_host():
    import torch
    from docodo_amd import synth
    n = 3000
    d = torch.empty(n * synth.DOC_BYTES, dtype=torch.uint8, device="cuda")
    synth.corpus_dev(d.data_ptr(), n, first_doc=17)
    assert np.array_equal(d.cpu().numpy(), synth.corpus_host(n, first_doc=17))


def test_device_resident_text_and_coord_base(oracle_mod):
    """dcd_build_add_pages_dev + coord_base (doc-range shard with global coordinates)."""
    import torch
    import docodo_amd
    from docodo_amd import synth
    n, first = 1500, 4000
    text = synth.corpus_host(n, first_doc=first)
    off = np.arange(n + 1, dtype=np.uint64) * synth.DOC_BYTES
    base = first * synth.DOC_BYTES
    d = torch.from_numpy(text).cuda()
    b = docodo_amd.Builder(coord_base=base)
    b.AddPagesDev(d.data_ptr(), text.size, off, docs_per_page=True, first_doc=first)
    b.Finish()
    # oracle: index the first `first` docs as blanks is wasteful; instead shift-check per term
    oix = oracle_mod.Index(stemmers=[])
    ob = oracle_mod.Builder(oix)
    ob.index_docs_latin1(text, off, first_doc=first)
    oi, ol = ob.save()
    oix.load(oi, ol)
    ix = b.Build()
    assert ix.Count == oix.count
    keys = oix.keys()
    for k in keys[:50] + keys[-50:]:
        want = oracle_mod.seq_shift(oix.get(k), base)
        assert np.array_equal(ix[k].units, want), k
    assert ix.MaxCoord == oix.max_coord + base


def _adversarial_pages():
    rng = np.random.default_rng(99)
    words = ["Pickwick", "ÀÉÎõü", "x", "ab", "abc", "A1b2C3", "12345", "mañana", "Straße", "ÞORN", "µ", "ª", "¼½", "²",
             "a" * 32, "b" * 33, "c" * 40, "Z" * 31 + "9", "don't", "e-mail", "naïve", "ÿ", "×", "÷", "OLD", "old", "Old"]
    seps = [" ", ", ", ".", "\n", "; ", "--", "\t", "×", " ÷ "]
    pages = []
    for p in range(300):
        n = int(rng.integers(0, 120))
        s = "".join(words[int(rng.integers(len(words)))] + seps[int(rng.integers(len(seps)))] for _ in range(n))
        # cut pages at arbitrary points so that words straddle page boundaries
        cut = int(rng.integers(0, len(s) + 1)) if s else 0
        pages.append(s[:cut])
        pages.append(s[cut:])
    # long pages spanning several 4096-character tiles with words across tile borders
    pages.append(("tileborder " * 2000))
    pages.append("q" * 4090 + " word straddle" + "w" * 20 + " " + "v" * 5000 + " end")
    return [p for p in pages]


def test_adversarial_latin1(oracle_mod):
    pages = _adversarial_pages()
    docs = [("src", "doc%d" % i, 7) for i in range(len(pages) // 7)] + [("src", "tail", len(pages) % 7)]
    docs = [d for d in docs if d[2]]
    ids = [str(1 + (i % 7)) for i in range(len(pages))]
    (oi, ol), ob = _oracle_images(oracle_mod, pages, ids=ids, docs=docs)
    (gi, gl), gb, st = _gpu_images(pages, ids=ids, docs=docs)
    assert st.n_postings == ob.total_count
    assert gl == ol
    assert gi == oi


def test_utf16_testtext1_body_pages(oracle_mod, testtext1):
    """TestText1 (IndexTest.cs:19-37, contains U+FFFD) through the UTF-16 tokeniser: 50 identical body
    pages in 2 docs; per-word postings equal the oracle's (CoordTest's assertion, IndexTest.cs:140-157)."""
    import docodo_amd
    greek = "Ελληνικά ΚΕΦΑΛΑΊΑ кириллица ЗАГЛАВНЫЕ 数字１２３ ǅ ǆ İi ﬁ"
    pages = [testtext1, greek + " " + testtext1[:100]] * 25
    docs = [("Test", "Sample", 25), ("Test", "Dump", 25)]
    ids = [str(i + 1) for i in range(25)] * 2
    (oi, ol), ob = _oracle_images(oracle_mod, pages, ids=ids, docs=docs)
    (gi, gl), gb, st = _gpu_images(pages, ids=ids, docs=docs, encoding=docodo_amd.DCD_ENC_UTF16LE)
    assert st.n_postings == ob.total_count and st.n_terms == ob.n_terms
    assert gl == ol
    assert gi == oi


def test_word_length_limits_and_empty(oracle_mod):
    import docodo_amd
    pages = ["", "a bb ccc dddd " * 50, "", "    ", "zzz"]
    (gi, gl), gb, st = _gpu_images(pages, docs=[("s", "d", 5)], ids=["1", "2", "3", "4", "5"])
    (oi, ol), ob = _oracle_images(oracle_mod, pages, docs=[("s", "d", 5)], ids=["1", "2", "3", "4", "5"])
    assert gi == oi and gl == ol
    # empty build: only the MaxCoord header
    b = docodo_amd.Builder()
    b.AddDoc("s", "d")
    b.Finish()
    assert b.index_image() == bytes(8)


def test_min_word_length_one_dense_tokens(oracle_mod):
    """MIN_WORD_LENGTH = 1 and alternating letter / digit classes: every character is a token of its own (the regex
    `\\p{L}+|\\p{N}+` splits at a class change without a separator), i.e. the densest token stream a tile can hold."""
    pages = ["a1" * 3000, "x9y8z7" * 700 + " tail", "b2" * 5, "", "a1b" * 1500]
    ids = ["1", "2", "3", "4", "5"]
    docs = [("s", "d", 5)]
    ix = oracle_mod.Index(stemmers=[])
    ix.set_limits(min_len=1)
    ob = oracle_mod.Builder(ix)
    ob.add_doc("s", "d")
    for i, t in zip(ids, pages):
        ob.index_page(i, t)
    oi, ol = ob.save()
    (gi, gl), gb, st = _gpu_images(pages, docs=docs, ids=ids, min_word_length=1)
    assert st.n_postings == ob.total_count
    assert gi == oi and gl == ol


@pytest.mark.parametrize("cap_log2", [20, 23, 24, 26, 27])
def test_vocabulary_capacity_does_not_change_the_index(cap_log2):
    """The sort's first pass reads slot -> rank through a small LDS cache whose entries pack (slot tag, rank) into one
    word: table sizes on both sides of the packing limit (2^23 slots: the highest tag equals the empty marker's) and
    beyond it (no cache) must give the same bytes as the default table.  2^27 slots is also past the size up to which
    the tokeniser keeps a string pool beside the table (hits are then verified against the text only)."""
    import docodo_amd
    from docodo_amd import synth
    n_docs = 3000
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    images = []
    for cap in (0, cap_log2):
        b = docodo_amd.Builder(vocab_capacity_log2=cap)
        b.AddPages(text, off, docs_per_page=True)
        st = b.Finish()
        if cap:
            assert st.vocab_capacity_log2 == cap
        images.append((b.index_image(), b.list_image()))
        b.close()
    assert images[0] == images[1]


def test_tile_population_extremes(oracle_mod):
    """Plain text builds write tile-strided tuples (no counting pass) that the first sort pass reads in place: tiles at
    the population limit (3-character tokens back to back, a class change and no separator between them: 314 per
    944-character tile), empty tiles (long runs of blanks / punctuation), tiles of one token, and a mix."""
    dense = "abc123" * 4000                                  # 24 000 characters, one token per 3 characters
    blanks = " " * 5000 + "." * 3000
    sparse = ("x" * 31 + " " * 900) * 8                      # one long token per tile or so
    pages = [dense, blanks, "word " * 10, blanks + "tail", sparse, dense[:1000] + blanks[:2000] + dense[:1001], "", "end"]
    ids = [str(i + 1) for i in range(len(pages))]
    docs = [("s", "d", len(pages))]
    (gi, gl), gb, st = _gpu_images(pages, docs=docs, ids=ids)
    (oi, ol), ob = _oracle_images(oracle_mod, pages, docs=docs, ids=ids)
    assert st.n_postings == ob.total_count
    assert gi == oi and gl == ol


@pytest.mark.parametrize("env", [{"DOCODO_TOK_TWO_PASS": "1"}, {"DOCODO_VOCAB_POOL": "0"},
                                 {"DOCODO_TOK_TWO_PASS": "1", "DOCODO_VOCAB_POOL": "0"}])
def test_alternative_build_paths_give_the_same_bytes(env):
    """The default build tokenises once (tile-strided tuples, string pool beside the vocabulary table); the counting-pass
    form and the text-only verification stay available behind environment switches and must produce the same files."""
    import hashlib
    import os
    import subprocess
    import sys
    import docodo_amd
    from docodo_amd import synth
    n_docs = 20000
    text = synth.corpus_host(n_docs)
    b = docodo_amd.Builder()
    b.AddPages(text, np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES, docs_per_page=True)
    b.Finish()
    h = hashlib.sha256(b.index_image())
    h.update(b.list_image())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "helpers", "build_digest_worker.py"), str(n_docs)],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "digest " + h.hexdigest() in r.stdout


def test_explicit_postings_builder_api(oracle_mod, known):
    """Index.Builder standalone path (BuilderTest, IndexTest.cs:272-313): AddWord(token, offset) for the
    \\b\\w+\\b tokens of TestText1, EndPage("1"); identity codes (no stemmer / vocab)."""
    import docodo_amd
    ka = known["D13_builder"]
    oix = oracle_mod.Index(stemmers=[])
    ob = oracle_mod.Builder(oix)
    ob.add_doc("A", "")
    for off, tok in ka["tokens"]:
        ob.add_word(tok, off)
    ob.end_page("1")
    oi, ol = ob.save()
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE)
    b.AddDoc("A", "")
    for off, tok in ka["tokens"]:
        b.AddWord(tok, off)
    b.EndPage("1")
    b.Finish()
    assert b.list_image() == ol
    assert b.index_image() == oi


def test_vocabulary_table_growth(oracle_mod):
    """Every token distinct: the vocabulary hash table starts far too small (one slot per 64 postings) and
    must be re-grown (x4 per round, emit re-run) until the load drops below 1/2 -- results stay exact."""
    import docodo_amd

    def word(k):
        s = ""
        for _ in range(5):
            s += chr(97 + k % 26)
            k //= 26
        return s

    n = 150_000
    text = " ".join(word(k) for k in range(n))
    pages = [text[i:i + 50_000] for i in range(0, len(text), 50_000)]
    (oi, ol), ob = _oracle_images(oracle_mod, pages, docs=[("s", "d", len(pages))], ids=[str(i + 1) for i in range(len(pages))])
    (gi, gl), gb, st = _gpu_images(pages, docs=[("s", "d", len(pages))], ids=[str(i + 1) for i in range(len(pages))])
    assert st.n_terms == ob.n_terms and st.n_terms > 100_000
    assert st.vocab_capacity_log2 >= 18
    assert gi == oi and gl == ol


@pytest.mark.gpu
def test_device_cache_is_stream_ordered_and_can_be_trimmed():
    """Released blocks stay in the library's per-device cache (no hipMalloc in repeated builds) and go back to HIP with
    dcd_pool_trim; two builders working from different threads (= different streams, Index.cs:464-465) that recycle each
    other's blocks still produce the images of a build that ran alone."""
    import threading
    import docodo_amd
    from docodo_amd import synth
    n_docs = 3000
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES

    def build(first, count):
        b = docodo_amd.Builder(coord_base=first * synth.DOC_BYTES)
        b.AddPages(text[first * synth.DOC_BYTES:(first + count) * synth.DOC_BYTES], off[:count + 1], docs_per_page=True, first_doc=first)
        b.Finish()
        img = (b.index_image(), b.list_image())
        b.close()
        return img

    want = [build(0, 1500), build(1500, 1500)]
    docodo_amd.pool_trim(0)
    assert docodo_amd.pool_cached_bytes(0) == 0
    got = [[], []]

    def worker(k):
        for _ in range(6):
            got[k].append(build(1500 * k, 1500))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for k in range(2):
        assert all(g == want[k] for g in got[k])
    assert docodo_amd.pool_cached_bytes(0) > 0
    docodo_amd.pool_trim(0)
    assert docodo_amd.pool_cached_bytes(0) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_docs", [3, 5000, 60000])
def test_id_keyed_and_rank_keyed_builds_agree(n_docs, tmp_path):
    """Plain text builds sort the tuples by dense vocabulary id and run the ordinal vocabulary sort on a second stream
    (term order is restored when the units are copied out); DOCODO_BUILD_RANK_KEYS=1 keeps the ordinal rank as the sort
    key.  Same images, same index arrays (units, starts, first / last coordinate per term through the shard export)."""
    import subprocess
    import sys
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
import docodo_amd
from docodo_amd import synth, dist
n = %d
text = synth.corpus_host(n)
off = np.arange(n + 1, dtype=np.uint64) * synth.DOC_BYTES
b = docodo_amd.Builder(coord_base=7 * 1024)
b.AddPages(text, off, docs_per_page=True)
st = b.Finish()
h = hashlib.sha256()
h.update(b.index_image()); h.update(b.list_image())
e = dist.export_shard(b, 3)
for t in (e.meta, e.keys, e.units):
    h.update(t.cpu().numpy().tobytes())
h.update(np.concatenate([e.term_count, e.key_count, e.unit_count]).tobytes())
ix = b.Build()
for w in ("aaaaaaa", synth.term(17), synth.term(4000)):
    t = ix.find(w)
    if t is not None:
        h.update(ix[w].units.tobytes()); h.update(str(ix.term_count(t)).encode())
print("DIGEST", h.hexdigest(), st.n_postings, st.n_terms, st.n_units)
''' % (ROOT, n_docs)
    outs = []
    # ordinal-rank keys; dense-id keys from a scan of the finished table without the hot / cold split, with a split of at
    # most 5 / 255 hot terms; ids handed out by the table while it is filled: default (hot set from a sampling launch of 64
    # groups), a hot set of at most 5 terms, of a sample of 2 groups, none; without any event records
    E0 = {"DOCODO_BUILD_EARLY_IDS": "0"}
    for extra in ({"DOCODO_BUILD_RANK_KEYS": "1"}, dict(E0, DOCODO_BUILD_HOT="0"), dict(E0, DOCODO_BUILD_HOT="5"), E0, {},
                  {"DOCODO_BUILD_HOT": "5"}, {"DOCODO_TOK_SAMPLE_GROUPS": "2"}, {"DOCODO_BUILD_HOT": "0"}, {"DOCODO_BUILD_EVENTS": "0"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0])
    assert len(set(outs)) == 1, outs


@pytest.mark.gpu
def test_hot_cold_split_with_a_vocabulary_beyond_the_remap_cache(oracle_mod):
    """1.1 M distinct words (dense ids beyond the 2^20 values the LDS remap cache of the first radix pass can hold: the
    uncached gather runs; 21 cold key bits = three cold passes) mixed with a few very frequent ones (hot terms, final
    after pass A): the images stay byte-identical to the oracle's."""
    import docodo_amd

    def word(k):
        s = ""
        for _ in range(6):
            s += chr(97 + k % 26)
            k //= 26
        return s

    rng = np.random.default_rng(11)
    n = 1_100_000
    hot = ["the", "and", "pickwick", "tupman", "winkle", "snodgrass", "wardle", "jingle"]
    picks = rng.choice(len(hot), size=n, p=np.array([0.4, 0.25, 0.1, 0.08, 0.07, 0.05, 0.03, 0.02]))
    parts = []
    for k in range(n):
        parts.append(word(k))
        parts.append(hot[picks[k]])
        if k % 3 == 0:
            parts.append(hot[picks[(k * 7) % n]])
    text = " ".join(parts)
    pages = [text[i:i + 1_000_000] for i in range(0, len(text), 1_000_000)]
    docs = [("s", "d", len(pages))]
    ids = [str(i + 1) for i in range(len(pages))]
    (oi, ol), ob = _oracle_images(oracle_mod, pages, docs=docs, ids=ids)
    (gi, gl), gb, st = _gpu_images(pages, docs=docs, ids=ids)
    assert st.n_terms == ob.n_terms and st.n_terms > (1 << 20)
    # the eight hot words are every second token: whichever way the hot set was chosen (sample counts of the filled table,
    # id_keys 1; the sampling launch of the numbering table, id_keys 2) it has to hold them; one pass for the hot terms + three
    # for the 21 key bits of the others
    assert st.id_keys in (1, 2)
    if os.environ.get("DOCODO_BUILD_HOT", "255") != "0":       # (0 in the environment of the test run: no hot set)
        assert 1 <= st.n_hot_terms <= 255 and st.n_hot_postings > st.n_postings // 2 and st.sort_passes == 4
    assert gi == oi and gl == ol


@pytest.mark.gpu
def test_utf16_and_latin1_builds_of_one_corpus_agree():
    """The synthetic corpus as 16-bit code units (what a C# string pins; 64-byte vocabulary records) gives the images of its
    Latin-1 form."""
    import docodo_amd
    from docodo_amd import synth
    n_docs = 20000
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    a = docodo_amd.Builder()
    a.AddPages(text, off, docs_per_page=True)
    sa = a.Finish()
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE)
    b.AddPages(text.astype(np.uint16), off, docs_per_page=True)
    sb = b.Finish()
    assert sa.id_keys == sb.id_keys and sb.id_keys in (1, 2)
    assert (sa.n_postings, sa.n_terms, sa.n_units) == (sb.n_postings, sb.n_terms, sb.n_units)
    assert a.index_image() == b.index_image() and a.list_image() == b.list_image()


def test_randomised_build_differential_small():
    """tests/stress_build.py (random vocabularies, Zipf exponents, word classes incl. Latin-1 letters, separators or none at a class
    change, ragged documents, word-length limits, both encodings; ".index" and ".index.list" byte for byte against the oracle) at a
    small size: 10 cases of one seed.  The long form (320 cases over six seeds and two build variants) stays a manual run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_build.py"), "10", "4"], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ", 0 mismatches" in r.stdout.strip().splitlines()[-1], r.stdout[-2000:]

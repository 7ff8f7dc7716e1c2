"""Index / Index.Builder mirror over the C ABI (reference: Docodo.NET/Index.cs, Build.cs, Search.cs).

This is the host-side stand-in for the C# shim described in INTEGRATION.md: same method names and
argument meaning as the reference where the hot path is concerned; everything numeric happens in
libdocodo_hip.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BuildOpts, BuildStats, QueryOpts, ResultsView, check, lib
from .sequence import IndexSequence


def _u16(s):
    a = np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()
    return a, (a.ctypes.data if a.size else None), a.size


class SearchOptions:  # Search.cs:262-276
    def __init__(self, dist=0):
        self.dist = dist


class Builder:
    """Index.Builder (Build.cs:258-437) fused with the IndexTask page loop (Build.cs:445-564)."""

    def __init__(self, encoding=_lib.DCD_ENC_LATIN1, coord_base=0, device=0, min_word_length=3, max_word_length=32,
                 vocab_capacity_log2=0, get_word_codes=None, shard_only=False):
        """get_word_codes: callable word -> list of index codes (GetWordCodes, Build.cs:175-247; see
        docodo_amd.wordcodes.WordCoder).  None = identity codes (Stemmers.Clear(), no vocabulary, no stop words)."""
        self.get_word_codes = get_word_codes
        o = BuildOpts()
        lib().dcd_build_default_opts(C.byref(o))
        o.device, o.encoding, o.coord_base = device, encoding, coord_base
        o.min_word_length, o.max_word_length = min_word_length, max_word_length
        o.vocab_capacity_log2 = vocab_capacity_log2
        # shard_only: one doc-range shard of a distributed build -- exported (dist.export_shard) and merged elsewhere, never
        # saved or queried on its own: Finish then skips the ".index" image and the term-ordered posting array
        o.flags = _lib.DCD_BUILD_SHARD_ONLY if shard_only else 0
        self.encoding = encoding
        self.device = device
        self.h = C.c_void_p()
        check(lib().dcd_build_create(C.byref(o), C.byref(self.h)))
        self._keep = []

    def close(self):
        if self.h:
            lib().dcd_build_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def AddDoc(self, sourceid, name):  # Build.cs:348
        a, p, n = _u16(sourceid)
        b, q, m = _u16(name)
        check(lib().dcd_build_add_doc(self.h, p, n, q, m))

    def _text_array(self, text):
        if isinstance(text, str):
            if self.encoding == _lib.DCD_ENC_UTF16LE:
                return np.frombuffer(text.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()
            return np.frombuffer(text.encode("latin-1"), dtype=np.uint8).copy()
        dt = np.uint16 if self.encoding == _lib.DCD_ENC_UTF16LE else np.uint8
        return np.ascontiguousarray(text, dtype=dt)

    def AddPages(self, text, page_off=None, page_ids=None, docs_per_page=False, first_doc=0, source="syn"):
        """Body pages back to back (the per-page loop of IndexTask, Build.cs:477-541)."""
        t = self._text_array(text)
        off = np.ascontiguousarray(page_off if page_off is not None else [0, t.size], dtype=np.uint64)
        ids_p = idoff_p = None
        if page_ids is not None:
            joined = "".join(page_ids)
            ids_a, ids_p, _ = _u16(joined)
            idoff = np.zeros(len(page_ids) + 1, np.uint64)
            idoff[1:] = np.cumsum([len(x.encode("utf-16-le")) // 2 for x in page_ids])
            idoff_p = idoff.ctypes.data
            self._keep += [ids_a, idoff]
        s, sp, sn = _u16(source)
        check(lib().dcd_build_add_pages(self.h, t.ctypes.data if t.size else None, t.size, off.ctypes.data, off.size - 1,
                                        ids_p, idoff_p, 1 if docs_per_page else 0, first_doc, sp, sn))

    def AddPagesDev(self, text_dev_ptr, n_chars, page_off, docs_per_page=True, first_doc=0, source="syn"):
        off = np.ascontiguousarray(page_off, dtype=np.uint64)
        s, sp, sn = _u16(source)
        check(lib().dcd_build_add_pages_dev(self.h, text_dev_ptr, n_chars, off.ctypes.data, off.size - 1,
                                            1 if docs_per_page else 0, first_doc, sp, sn))

    def AddUniformPagesDev(self, text_dev_ptr, n_pages, page_chars, first_doc=0, source="syn"):
        s, sp, sn = _u16(source)
        check(lib().dcd_build_add_uniform_pages_dev(self.h, text_dev_ptr, n_pages, page_chars, first_doc, sp, sn))

    def Add(self, code, coord):  # Build.cs:324
        a, p, n = _u16(code)
        check(lib().dcd_build_add(self.h, p, n, int(coord)))

    def AddWord(self, word, coord):
        """Build.cs:302-316: the word is indexed under every code GetWordCodes gives it (see Finish)."""
        a, p, n = _u16(word)
        check(lib().dcd_build_add_word(self.h, p, n, int(coord)))

    def Words(self):
        """The distinct words of the build in ordinal order (runs tokenise + vocabulary on the device)."""
        nw, nc = C.c_size_t(0), C.c_size_t(0)
        check(lib().dcd_build_words(self.h, C.byref(nw), C.byref(nc)))
        chars = np.empty(max(1, nc.value), np.uint16)
        off = np.empty(nw.value + 1, np.uint32)
        check(lib().dcd_build_words_get(self.h, chars.ctypes.data, chars.size, off.ctypes.data, off.size))
        raw = chars[:nc.value].tobytes()
        return [raw[2 * int(off[i]):2 * int(off[i + 1])].decode("utf-16-le", "surrogatepass") for i in range(nw.value)]

    def SetCodes(self, codes_per_word):
        """codes_per_word[w] = list of code strings of word w of Words() ([] / None = dropped)."""
        code_off = np.zeros(len(codes_per_word) + 1, np.uint32)
        strs = []
        for w, cs in enumerate(codes_per_word):
            cs = cs or []
            strs += cs
            code_off[w + 1] = code_off[w] + len(cs)
        str_off = np.zeros(len(strs) + 1, np.uint32)
        if strs:
            str_off[1:] = np.cumsum([len(s.encode("utf-16-le", "surrogatepass")) // 2 for s in strs])
        chars = np.frombuffer("".join(strs).encode("utf-16-le", "surrogatepass"), np.uint16)
        check(lib().dcd_build_set_codes(self.h, code_off.ctypes.data, len(codes_per_word), str_off.ctypes.data,
                                        chars.ctypes.data if chars.size else None, len(strs)))

    def EndPage(self, id):  # Build.cs:359
        a, p, n = _u16(id)
        check(lib().dcd_build_end_page(self.h, p, n))

    def Finish(self):
        if self.get_word_codes is not None:
            # GetWordCodes once per DISTINCT word on the host; one posting per (token, code) on the device
            self.SetCodes([self.get_word_codes(w) for w in self.Words()])
        check(lib().dcd_build_finish(self.h))
        return self.stats

    @property
    def stats(self):
        s = BuildStats()
        check(lib().dcd_build_get_stats(self.h, C.byref(s)))
        return s

    def _image(self, fn):
        n = C.c_size_t(0)
        fn(self.h, None, 0, C.byref(n))
        out = np.empty(max(1, n.value), np.uint8)
        check(fn(self.h, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value].tobytes()

    def index_image(self):
        return self._image(lib().dcd_build_index_image)

    def list_image(self):
        return self._image(lib().dcd_build_list_image)

    def Save(self, dirpath):
        check(lib().dcd_build_save(self.h, dirpath.encode("utf-8")))

    def Build(self):  # Build.cs:407-434 -> searchable Index
        return Index.from_build(self)


class QueryBatchResult:
    """Numpy copy of a dcd_results batch (CSR layout of include/docodo_hip.h)."""

    def __init__(self, view):
        nq = view.n_queries

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.empty(0, dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

        self.n_queries = nq
        self.status = arr(view.status, nq, np.int32)
        self.page_off = arr(view.page_off, nq + 1, np.uint64)
        npages = int(self.page_off[-1]) if nq else 0
        self.page_index = arr(view.page_index, npages, np.uint32)
        self.page_rank = arr(view.page_rank, npages, np.float32)
        self.pos_off = arr(view.pos_off, npages + 1, np.uint64)
        npos = int(self.pos_off[-1]) if npages else 0
        self.pos = arr(view.pos, npos, np.int32)
        self.doc_off = arr(view.doc_off, nq + 1, np.uint64)
        ndocs = int(self.doc_off[-1]) if nq else 0
        self.doc_index = arr(view.doc_index, ndocs, np.uint32)
        self.doc_rank = arr(view.doc_rank, ndocs, np.float32)
        self.doc_n_pages = arr(view.doc_n_pages, ndocs, np.uint32)
        self.seq_off = arr(view.seq_off, nq + 1, np.uint64) if view.seq_off else None
        if self.seq_off is not None:
            self.seq_units = arr(view.seq_units, int(self.seq_off[-1]), np.uint16)
            self.seq_R = arr(view.seq_R, nq, np.int32)
        self.ms_device = view.ms_device
        # common sub-expressions (dcd_results_view, ABI 2): operators named by the programs / evaluated, units of their index terms
        self.ops_named, self.ops_evaluated = int(view.ops_named), int(view.ops_evaluated)
        self.units_named, self.units_evaluated = int(view.units_named), int(view.units_evaluated)
        self.sub_batches = int(view.sub_batches)

    def pages(self, q):
        a, b = int(self.page_off[q]), int(self.page_off[q + 1])
        return [(int(self.page_index[i]), self.pos[int(self.pos_off[i]):int(self.pos_off[i + 1])], float(self.page_rank[i]))
                for i in range(a, b)]

    def docs(self, q):
        a, b = int(self.doc_off[q]), int(self.doc_off[q + 1])
        return [(int(self.doc_index[i]), float(self.doc_rank[i]), int(self.doc_n_pages[i])) for i in range(a, b)]


class Index:
    """Docodo.Index as far as the hot path goes: Load, Count, MaxCoord, this[key], batched Search."""

    def __init__(self, handle, device=0):
        self.h = handle
        self.device = device

    @classmethod
    def from_build(cls, builder):
        h = C.c_void_p()
        check(lib().dcd_index_from_build(builder.h, C.byref(h)))
        return cls(h, getattr(builder, "device", 0))

    @classmethod
    def Load(cls, dirpath, device=0):  # Index.cs:312
        h = C.c_void_p()
        check(lib().dcd_index_open(dirpath.encode("utf-8"), device, C.byref(h)))
        return cls(h, device)

    @classmethod
    def LoadOnDisk(cls, dirpath, device=0, resident_bytes=0):
        """new Index(path, InMemory: false) (Index.cs:124-136, 346-348): dictionary + page table loaded, posting lists read from
        ".index" when a key / a query batch asks for them (LoadSequence, Index.cs:198-216)."""
        h = C.c_void_p()
        check(lib().dcd_index_open_lazy(dirpath.encode("utf-8"), device, int(resident_bytes), C.byref(h)))
        return cls(h, device)

    def lazy_stats(self):
        """(lists read, bytes read, terms resident now) of an on-disk index."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().dcd_index_lazy_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    @classmethod
    def from_images(cls, index_bytes, list_bytes, device=0):
        ib = np.frombuffer(index_bytes, np.uint8)
        lb = np.frombuffer(list_bytes, np.uint8)
        h = C.c_void_p()
        check(lib().dcd_index_from_images(ib.ctypes.data, ib.size, lb.ctypes.data if lb.size else None, lb.size, device, C.byref(h)))
        return cls(h, device)

    def close(self):
        if self.h:
            lib().dcd_index_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self):
        """dcd_index_prepare: makes the resident coordinates + cell directories of the query side now; -> device ms (0 if present)."""
        ms = C.c_float(0)
        check(lib().dcd_index_prepare(self.h, C.byref(ms)))
        return float(ms.value)

    def trim(self):
        """Releases the work arena of the query batches (and the resident lists of an on-disk index)."""
        check(lib().dcd_index_trim(self.h))

    def _u64(self, fn):
        v = C.c_uint64(0)
        check(fn(self.h, C.byref(v)))
        return v.value

    # -- the reference's public surface (Index.cs:124-263, Search.cs:430-437) on top of the handle ------------------
    @property
    def CanSearch(self):          # Index.cs: true once an index is loaded / built
        return bool(self.h)

    def Dispose(self):            # Index.cs:247-251
        """Releases the handle and hands the library's cached device blocks back to HIP (dcd_pool_trim): after Dispose
        the memory is available to every other allocator of the process."""
        dev = getattr(self, "device", 0)
        self.close()
        from ._lib import pool_trim
        pool_trim(dev)

    def Search(self, req, opt=None, get_word_codes=None):   # Index.Search(string, SearchOptions), Search.cs:437
        from .search import Search
        if getattr(self, "_search", None) is None or get_word_codes is not None:
            self._search = Search(self, get_word_codes=get_word_codes)
        return self._search(req, opt)

    @property
    def Count(self):
        return self._u64(lib().dcd_index_count)

    @property
    def MaxCoord(self):
        return self._u64(lib().dcd_index_max_coord)

    @property
    def n_units(self):
        """Posting units of all lists together."""
        return self._u64(lib().dcd_index_n_units)

    @property
    def n_pages(self):
        return self._u64(lib().dcd_index_n_pages)

    @property
    def n_docs(self):
        return self._u64(lib().dcd_index_n_docs)

    def find(self, key):
        a, p, n = _u16(key)
        t = C.c_uint32(0)
        code = lib().dcd_index_find(self.h, p, n, C.byref(t))
        if code == _lib.DCD_E_NOTFOUND:
            return None
        check(code)
        return t.value

    def key(self, term):
        out = np.empty(64, np.uint16)
        n = C.c_size_t(0)
        check(lib().dcd_index_key(self.h, term, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value].tobytes().decode("utf-16-le", "surrogatepass")

    def get(self, term):
        n = C.c_size_t(0)
        lib().dcd_index_get(self.h, term, None, 0, C.byref(n))
        out = np.empty(max(1, n.value), np.uint16)
        check(lib().dcd_index_get(self.h, term, out.ctypes.data, out.size, C.byref(n)))
        return IndexSequence(out[:n.value].copy())

    def __getitem__(self, key):  # IndexSequence this[string key], Index.cs:255-263 (SortedList indexer: a missing key throws)
        t = self.find(key)
        if t is None:
            raise KeyError(key)
        return self.get(t)

    def term_count(self, term):
        v = C.c_uint64(0)
        check(lib().dcd_index_term_count(self.h, term, C.byref(v)))
        return v.value

    def term_units(self, term):
        """IndexSequence.Count of this[key]: the number of ushort UNITS of the posting list (IndexSequence.cs:43-45)."""
        n = C.c_size_t(0)
        lib().dcd_index_get(self.h, term, None, 0, C.byref(n))
        return n.value

    def page_info(self, page):
        end, doc, n = C.c_uint64(0), C.c_uint32(0), C.c_size_t(0)
        out = np.empty(256, np.uint16)
        check(lib().dcd_index_page_info(self.h, page, C.byref(end), C.byref(doc), out.ctypes.data, out.size, C.byref(n)))
        return end.value, doc.value, out[:n.value].tobytes().decode("utf-16-le", "surrogatepass")

    def _term_list(self, fn, text, n):
        a, p, ln = _u16(text)
        out = np.empty(max(1, n), np.uint32)
        k = C.c_uint32(0)
        check(fn(self.h, p, ln, n, out.ctypes.data, C.byref(k)))
        return [int(t) for t in out[:k.value]]

    def close_words(self, word, n=10):
        """GetCloseWords (Search.cs:168-174): terms of the n keys closest to `word` by Levenshtein distance."""
        return self._term_list(lib().dcd_index_close_words, word, n)

    def like_words(self, pattern, n=100):
        """GetLikeWords (Search.cs:158-167): terms of the first n keys that start with a letter and match `pattern` ('_' = ".*",
        unanchored) -- a dictionary scan on the device."""
        return self._term_list(lib().dcd_index_like_words, pattern, n)

    def suggest(self, prefix, n=10):
        """Term selection of GetSuggessions (Search.cs:176-188)."""
        return self._term_list(lib().dcd_index_suggest, prefix, n)

    def _image(self, fn):
        n = C.c_size_t(0)
        fn(self.h, None, 0, C.byref(n))
        out = np.empty(max(1, n.value), np.uint8)
        check(fn(self.h, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value].tobytes()

    def index_image(self):
        """Exact ".index" bytes of this (possibly merged / gathered) index."""
        return self._image(lib().dcd_index_image)

    def list_image(self):
        return self._image(lib().dcd_index_list_image)

    def set_list_image(self, list_bytes):
        """Attach a page table given as ".index.list" bytes (e.g. the shards' lists back to back)."""
        lb = np.frombuffer(list_bytes, np.uint8)
        check(lib().dcd_index_set_list_image(self.h, lb.ctypes.data if lb.size else None, lb.size))

    def Save(self, dirpath):
        check(lib().dcd_index_save(self.h, dirpath.encode("utf-8")))

    def doc_name(self, doc):
        out = np.empty(1024, np.uint16)
        n = C.c_size_t(0)
        check(lib().dcd_index_doc_name(self.h, doc, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value].tobytes().decode("utf-16-le", "surrogatepass")

    def query_batch(self, prog, prog_off, max_found_pages=30000, max_found_docs=500, doc_rank_multiply=10.0,
                    want_sequences=False, keep_handle=False, doc_match=None, totals=False):
        """doc_match: optional uint8 array, one entry per document: 1 = the document passes the request's -filter:.
        keep_handle: only the device time comes back; totals: {ms, n_pages, n_pos, n_docs} of the batch (measurements)."""
        prog = np.ascontiguousarray(prog, np.int32)
        off = np.ascontiguousarray(prog_off, np.uint64)
        o = QueryOpts()
        lib().dcd_query_default_opts(C.byref(o))
        o.max_found_pages, o.max_found_docs = max_found_pages, max_found_docs
        o.doc_rank_multiply, o.want_sequences = doc_rank_multiply, 1 if want_sequences else 0
        if doc_match is not None:
            dm = np.ascontiguousarray(doc_match, np.uint8)
            o.doc_match, o.n_doc_match = dm.ctypes.data, dm.size
        if totals:
            o.no_download = 1          # (the result arrays are produced on the device; only the counts come back)
        r = C.c_void_p()
        check(lib().dcd_query_batch(self.h, prog.ctypes.data if prog.size else None, off.ctypes.data, off.size - 1,
                                    C.byref(o), C.byref(r)))
        try:
            v = ResultsView()
            check(lib().dcd_results_get(r, C.byref(v)))
            if totals:
                return {"ms": float(v.ms_device), "n_pages": int(v.n_pages_total), "n_pos": int(v.n_pos_total), "n_docs": int(v.n_docs_total),
                        "ops_named": int(v.ops_named), "ops_evaluated": int(v.ops_evaluated),
                        "units_named": int(v.units_named), "units_evaluated": int(v.units_evaluated),
                        "sub_batches": int(v.sub_batches)}
            if keep_handle:
                return v.ms_device
            return QueryBatchResult(v)
        finally:
            lib().dcd_results_free(r)

"""CPU oracle for the Docodo hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
nothing under docodo_amd/ does.  The arithmetic lives in oracle/oracle.cpp (plain C++ compiled
with g++, `make -C oracle`), every function citing the reference file:line it restates.  This
module adds (a) ctypes bindings and (b) a literal restatement of the reference's host-side
request rewriting (Search.cs:319-363, 437-617), which is string/regex work that is only needed
to pin the oracle against the reference's known-answer tests.

Pinning status: the C# reference cannot be built or run in this image (no .NET toolchain), so
the oracle is pinned by the reference's own xUnit known answers (tests/golden/known_answers.json)
and codec round-trip properties; posting *bytes* are pinned by the format definition only.
"""
import ctypes as C
import os
import subprocess
import unicodedata

import numpy as np
import regex

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STEM_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint16), C.c_int, C.POINTER(C.c_uint16), C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u16p, u64p, u8p = C.POINTER(C.c_uint16), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
        sz, vp, i32p = C.c_size_t, C.c_void_p, C.POINTER(C.c_int)

        def sig(name, res, *args):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = list(args)

        sig("orc_set_char_tables", None, vp, vp)
        sig("orc_seq_encode", sz, vp, sz, vp, sz)
        sig("orc_seq_decode", sz, vp, sz, vp, sz)
        sig("orc_seq_shift", sz, vp, sz, C.c_uint64, vp, sz)
        sig("orc_seq_and", sz, vp, sz, C.c_int, vp, sz, C.c_int, vp, sz, i32p)
        sig("orc_seq_or", sz, vp, sz, C.c_int, vp, sz, C.c_int, vp, sz, i32p)
        sig("orc_index_new", vp)
        sig("orc_index_free", None, vp)
        sig("orc_index_set_limits", None, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
        sig("orc_index_add_stop_word", None, vp, vp, sz)
        sig("orc_index_add_voc", C.c_int, vp, C.c_uint16, C.c_uint16, STEM_FN)
        sig("orc_index_voc_add", None, vp, C.c_int, vp, sz, C.c_int)
        sig("orc_index_add_stemmer", None, vp, vp, sz, STEM_FN)
        sig("orc_index_clear_stemmers", None, vp)
        sig("orc_index_load", C.c_int, vp, vp, sz, vp, sz)
        sig("orc_index_count", sz, vp)
        sig("orc_index_max_coord", C.c_uint64, vp)
        sig("orc_index_n_pages", sz, vp)
        sig("orc_index_n_docs", sz, vp)
        sig("orc_index_key", sz, vp, sz, vp, sz)
        sig("orc_index_get", sz, vp, vp, sz, vp, sz)
        sig("orc_index_doc_name", sz, vp, sz, vp, sz)
        sig("orc_index_page_id", sz, vp, sz, vp, sz)
        sig("orc_index_page_end", C.c_uint64, vp, sz)
        sig("orc_index_page_doc", C.c_int, vp, sz)
        sig("orc_get_word_codes", sz, vp, vp, sz, vp, sz)
        sig("orc_search_word", sz, vp, vp, sz, vp, sz, i32p)
        sig("orc_close_words", sz, vp, vp, sz, sz, vp, sz, C.POINTER(sz))
        sig("orc_suggest", sz, vp, vp, sz, sz, vp, sz, C.POINTER(sz))
        sig("orc_like_words", sz, vp, vp, sz, vp, sz, C.POINTER(sz))
        sig("orc_builder_new", vp, vp)
        sig("orc_builder_free", None, vp)
        sig("orc_builder_add_doc", None, vp, vp, sz, vp, sz)
        sig("orc_builder_add_word", None, vp, vp, sz, C.c_uint64)
        sig("orc_builder_add", None, vp, vp, sz, C.c_uint64)
        sig("orc_builder_end_page", None, vp, vp, sz)
        sig("orc_builder_index_page", None, vp, vp, sz, vp, sz)
        sig("orc_builder_index_docs_latin1", None, vp, vp, vp, sz, C.c_uint64, vp, sz, vp, sz)
        sig("orc_builder_total_count", C.c_uint64, vp)
        sig("orc_builder_max_coord", C.c_uint64, vp)
        sig("orc_builder_coord", C.c_uint64, vp)
        sig("orc_builder_n_terms", sz, vp)
        sig("orc_builder_save", None, vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz))
        sig("orc_free", None, vp)
        sig("orc_builder_set_max_items", None, vp, C.c_longlong)
        sig("orc_builder_n_tmp", sz, vp)
        sig("orc_builder_save_merged", None, vp, C.POINTER(vp), C.POINTER(sz))
        sig("orc_builder_tmp", vp, vp, sz, C.POINTER(sz))
        sig("orc_merge_files", None, vp, vp, sz, C.c_int, vp, C.POINTER(vp), C.POINTER(sz))
        sig("orc_prepare_result", vp, vp, vp, sz, vp)
        sig("orc_result_intersect_docs", None, vp, vp, vp)
        sig("orc_result_finish", None, vp, vp)
        sig("orc_result_free", None, vp)
        sig("orc_result_error", C.c_int, vp)
        sig("orc_result_n_pages", sz, vp)
        sig("orc_result_page_index", C.c_long, vp, sz)
        sig("orc_result_page_npos", sz, vp, sz)
        sig("orc_result_page_pos", None, vp, sz, vp)
        sig("orc_result_page_rank", C.c_float, vp, sz)
        sig("orc_result_n_found_docs", sz, vp)
        sig("orc_result_found_doc", C.c_int, vp, sz)
        sig("orc_result_found_doc_rank", C.c_float, vp, sz)
        sig("orc_result_found_doc_npages", sz, vp, sz)
        sig("orc_query_batch", None, vp, vp, vp, sz, vp, vp, vp, vp)
        sig("orc_query_batch_mt", None, vp, vp, vp, sz, C.c_int, vp, vp, vp, vp)
        _LIB = L
    return _LIB


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _u16(s):
    """str -> (numpy uint16 array of UTF-16 code units, pointer, length)."""
    a = np.frombuffer(s.encode("utf-16-le", "surrogatepass"), dtype=np.uint16).copy()
    return a, a.ctypes.data, a.size


def _from_u16(a, n):
    return a[:n].tobytes().decode("utf-16-le", "surrogatepass")


def bmp_tables():
    """(cls[65536] u8, lower[65536] u16) as .NET defines \\p{L}, \\p{N}, \\w and String.ToLower per
    UTF-16 code unit -- derived from Python's unicodedata (Unicode 13 here; the reference's runtime
    ships an older UCD, so code points assigned since then are "parity unpinned")."""
    cls = np.zeros(65536, np.uint8)
    lower = np.arange(65536, dtype=np.uint16)
    for cp in range(65536):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        ch = chr(cp)
        cat = unicodedata.category(ch)
        v = 0
        if cat[0] == "L":
            v |= 1 | 4
        if cat[0] == "N":
            v |= 2
        if cat in ("Mn", "Nd", "Pc"):
            v |= 4
        cls[cp] = v
        lo = ch.lower()
        if len(lo) == 1 and ord(lo) < 65536:
            lower[cp] = ord(lo)
        elif cp == 0x130:
            lower[cp] = 0x69
    return cls, lower


def install_bmp_tables():
    cls, lower = bmp_tables()
    lib().orc_set_char_tables(cls.ctypes.data, lower.ctypes.data)


# --------------------------------------------------------------------------------------------
# IndexSequence algebra on numpy unit arrays
# --------------------------------------------------------------------------------------------
def seq_encode(values):
    v = np.ascontiguousarray(values, dtype=np.uint64)
    out = np.empty(max(1, 5 * v.size), np.uint16)
    n = lib().orc_seq_encode(v.ctypes.data, v.size, out.ctypes.data, out.size)
    return out[:n].copy()


def seq_decode(units):
    u = np.ascontiguousarray(units, dtype=np.uint16)
    out = np.empty(max(1, u.size), np.uint64)
    n = lib().orc_seq_decode(u.ctypes.data, u.size, out.ctypes.data, out.size)
    return out[:n].copy()


def seq_shift(units, shift):
    u = np.ascontiguousarray(units, dtype=np.uint16)
    out = np.empty(u.size + 8, np.uint16)
    n = lib().orc_seq_shift(u.ctypes.data, u.size, int(shift), out.ctypes.data, out.size)
    return out[:n].copy()


def _binop(fn, a, Ra, b, Rb):
    a = np.ascontiguousarray(a, dtype=np.uint16)
    b = np.ascontiguousarray(b, dtype=np.uint16)
    out = np.empty(max(1, 5 * (a.size + b.size)), np.uint16)
    R = C.c_int(0)
    n = fn(a.ctypes.data, a.size, int(Ra), b.ctypes.data, b.size, int(Rb), out.ctypes.data, out.size, C.byref(R))
    return out[:n].copy(), R.value


def seq_and(a, Ra, b, Rb):
    return _binop(lib().orc_seq_and, a, Ra, b, Rb)


def seq_or(a, Ra, b, Rb):
    return _binop(lib().orc_seq_or, a, Ra, b, Rb)


def merge_files(images, shift_coords=False, shifts=None):
    """MergeFiles (Index.cs:577-689) over ".tmpind" / ".index" byte images, literally -- INCLUDING its quirk with three or more
    inputs (see oracle.cpp).  shifts=None: the reference's own shifts (running sum of the inputs' MaxCoord, :589-593)."""
    n = len(images)
    bufs = [np.frombuffer(b, np.uint8) for b in images]
    ptrs = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (C.c_size_t * n)(*[b.size for b in bufs])
    sh = np.ascontiguousarray(shifts, np.uint64) if shifts is not None else None
    ob, on = C.c_void_p(), C.c_size_t()
    lib().orc_merge_files(ptrs, sizes, n, 1 if shift_coords else 0, sh.ctypes.data if sh is not None else None, C.byref(ob), C.byref(on))
    try:
        return C.string_at(ob.value, on.value)
    finally:
        lib().orc_free(ob)


def merge_pairwise(images, shifts=None):
    """A correct k-way merge out of the reference's own two-input MergeFiles (which is unaffected by the >= 3-input quirk):
    neighbours are merged pairwise, round by round, in input order.  shifts: coordinate base of every input (the
    multi-GPU / multi-task canonical mode: text length of everything in front of it); None = the inputs
    already share one coordinate space (MergeAll, Index.cs:537-573)."""
    items = [(img, 0 if shifts is None else int(shifts[i])) for i, img in enumerate(images)]
    if shifts is not None:                  # bring every input into the global space first: Shift is per input (Index.cs:671)
        items = [(merge_files([img], True, [sh]), 0) for img, sh in items]
    imgs = [img for img, _ in items]
    while len(imgs) > 1:
        nxt = [merge_files(imgs[i:i + 2], False) if i + 1 < len(imgs) else imgs[i] for i in range(0, len(imgs), 2)]
        imgs = nxt
    return imgs[0]


# --------------------------------------------------------------------------------------------
# Index / Builder / Search
# --------------------------------------------------------------------------------------------
class SearchOptions:  # Search.cs:262-276
    def __init__(self, dist=0):
        self.dist = dist


class ResultPage:
    def __init__(self, page_index, id, pos, rank):
        self.page_index, self.id, self.pos, self.rank = page_index, id, pos, rank


class ResultDoc:
    def __init__(self, doc_index, name, rank, n_pages):
        self.doc_index, self.name, self.rank, self.n_pages = doc_index, name, rank, n_pages


class SearchResult:
    def __init__(self, success=True, error=""):
        self.Success, self.Error = success, error
        self.foundPages, self.foundDocs, self.words = [], [], []


_W = r"\p{L}\p{Mn}\p{Nd}\p{Pc}"                                        # .NET \w
_B = rf"(?:(?<=[{_W}])(?![{_W}])|(?<![{_W}])(?=[{_W}]))"               # .NET \b


def _net_upper(s):
    return "".join(c.upper() if len(c.upper()) == 1 else c for c in s)


def _net_lower(s):
    return "".join(c.lower() if len(c.lower()) == 1 else ("i" if c == "İ" else c) for c in s)


class Index:
    """Restatement of Docodo.Index as far as the hot path needs it (config + loaded index + search)."""

    def __init__(self, stemmers=None):
        self.L = lib()
        self.h = self.L.orc_index_new()
        self._keep = []          # ctypes callbacks must outlive the index
        self.stop_words = set()
        self._keys = None
        # Index.cs:175-183 default stemmer table: only ranges are needed when no stemmer callback is
        # supplied; tests that mirror `index.Stemmers.Clear()` pass stemmers=[].
        if stemmers is not None:
            self.set_stemmers(stemmers)

    def set_limits(self, min_len=3, max_len=32, max_pages=30001, max_docs=500, keep_forms=False):
        """MIN_WORD_LENGTH / MAX_WORD_LENGTH (Index.cs:97,113), MAX_FOUND_PAGES / MAX_FOUND_DOCS, bKeepForms."""
        self.L.orc_index_set_limits(self.h, int(min_len), int(max_len), int(max_pages), int(max_docs), 1 if keep_forms else 0)

    def __del__(self):
        try:
            self.L.orc_index_free(self.h)
        except Exception:
            pass

    # -- configuration ---------------------------------------------------------------------
    def _wrap_stem(self, fn):
        if fn is None:
            return C.cast(None, STEM_FN)

        def cb(w, n, out, cap):
            s = bytes(C.cast(w, C.POINTER(C.c_uint16 * n)).contents) if n else b""
            r = fn(s.decode("utf-16-le"))
            e = np.frombuffer(r.encode("utf-16-le"), dtype=np.uint16)
            m = min(len(e), cap)
            for i in range(m):
                out[i] = int(e[i])
            return m

        c = STEM_FN(cb)
        self._keep.append(c)
        return c

    def set_stemmers(self, stemmers):
        """stemmers: list of (range string as in Index.cs:175-181, python callable or None)."""
        self.L.orc_index_clear_stemmers(self.h)
        for rng, fn in stemmers:
            a, p, n = _u16(rng)
            self.L.orc_index_add_stemmer(self.h, p, n, self._wrap_stem(fn))

    def add_stop_words(self, words):
        for w in words:
            self.stop_words.add(w)
            a, p, n = _u16(w)
            self.L.orc_index_add_stop_word(self.h, p, n)

    def add_voc(self, words, rng, stemmer=None):
        v = self.L.orc_index_add_voc(self.h, ord(rng[0]), ord(rng[1]), self._wrap_stem(stemmer))
        for w, g in words.items():
            a, p, n = _u16(w)
            self.L.orc_index_voc_add(self.h, v, p, n, g)
        return v

    # -- loaded index ------------------------------------------------------------------------
    def load(self, index_bytes, list_bytes):
        ib = np.frombuffer(index_bytes, np.uint8)
        lb = np.frombuffer(list_bytes, np.uint8)
        self._keys = None
        return self.L.orc_index_load(self.h, ib.ctypes.data, ib.size, lb.ctypes.data, lb.size) == 0

    @property
    def count(self):
        return self.L.orc_index_count(self.h)

    @property
    def max_coord(self):
        return self.L.orc_index_max_coord(self.h)

    def keys(self):
        if self._keys is None:
            buf = np.empty(4096, np.uint16)
            self._keys = []
            for i in range(self.count):
                n = self.L.orc_index_key(self.h, i, buf.ctypes.data, buf.size)
                self._keys.append(_from_u16(buf, n))
        return self._keys

    def get(self, key):
        a, p, n = _u16(key)
        m = self.L.orc_index_get(self.h, p, n, None, 0)
        if m == C.c_size_t(-1).value:
            return None
        out = np.empty(max(1, m), np.uint16)
        self.L.orc_index_get(self.h, p, n, out.ctypes.data, out.size)
        return out[:m].copy()

    def n_pages(self):
        return self.L.orc_index_n_pages(self.h)

    def n_docs(self):
        return self.L.orc_index_n_docs(self.h)

    def doc_name(self, d):
        buf = np.empty(4096, np.uint16)
        return _from_u16(buf, self.L.orc_index_doc_name(self.h, d, buf.ctypes.data, buf.size))

    def page_id(self, p):
        buf = np.empty(4096, np.uint16)
        return _from_u16(buf, self.L.orc_index_page_id(self.h, p, buf.ctypes.data, buf.size))

    def page_end(self, p):
        return self.L.orc_index_page_end(self.h, p)

    def page_doc(self, p):
        return self.L.orc_index_page_doc(self.h, p)

    def get_word_codes(self, word):
        a, p, n = _u16(word)
        out = np.empty(1024, np.uint16)
        m = self.L.orc_get_word_codes(self.h, p, n, out.ctypes.data, out.size)
        if m == C.c_size_t(-1).value:
            return None
        if m == 0:
            return []
        return _from_u16(out, m).split("\0")

    def _scan(self, fn, word, *take):
        a, p, n = _u16(word)
        cnt = C.c_size_t(0)
        m = fn(self.h, p, n, *take, None, 0, C.byref(cnt))
        out = np.empty(max(1, m), np.uint16)
        fn(self.h, p, n, *take, out.ctypes.data, out.size, C.byref(cnt))
        if cnt.value == 0:
            return []
        return _from_u16(out, m).split("\0")

    def close_words(self, word, n=10):
        """GetCloseWords, Search.cs:168-174 (Levenshtein, Index.cs:46-89)."""
        return self._scan(self.L.orc_close_words, word, n)

    def suggest_tails(self, lastword, n=10):
        """Term selection of GetSuggessions, Search.cs:176-188: the completions of `lastword` (already lower-cased)."""
        return self._scan(self.L.orc_suggest, lastword, n)

    def like_words(self, pattern):
        """GetLikeWords, Search.cs:158-167, non-trivial branch ('_' = ".*", unanchored, first 100 keys that start with a letter)."""
        return self._scan(self.L.orc_like_words, pattern)

    def search_word(self, word):
        """SearchWord (Search.cs:192-260) -> (units, R)."""
        a, p, n = _u16(word)
        R = C.c_int(0)
        m = self.L.orc_search_word(self.h, p, n, None, 0, C.byref(R))
        out = np.empty(max(1, m), np.uint16)
        self.L.orc_search_word(self.h, p, n, out.ctypes.data, out.size, C.byref(R))
        return out[:m].copy(), R.value

    # -- request rewriting: Search.cs:319-363 (literal regex pipeline) ------------------------
    def _prepare_search_request(self, req, funcs, searchfunc=None, keepshort=False):
        req = regex.sub(rf'[^{_W}(){{}}=~?|"]|_+', " ", req)                         # :325
        f = [""]

        def field_cb(m):                                                             # :328-333
            name = m.group(1)
            l, _ = self._prepare_search_request(m.group(2), funcs, (lambda s, name=name: self._search_field(name, s)), True)
            f[0] += ("*" if f[0] else "") + "(" + l + ")"
            return ""

        req = regex.sub(rf"{{*([{_W}]+)[ ]*=([{_W}|() ]+)}}", field_cb, req)
        fields = f[0]
        req = regex.sub(r"{.*}", "", req)                                            # :337
        req = req.replace("?", "_")                                                  # :340
        if not keepshort:
            req = regex.sub(rf"{_B}[{_W}]{{1,2}}{_B}", " ", req)                     # :344
        for st in self.stop_words:
            req = regex.sub(rf"{_B}{st}{_B}", "", req)                               # :346
        req = regex.sub(r'"(.*)"', lambda m: "(" + _net_upper(m.group(1)) + ")", req)  # :349
        req = req.replace("|", "+")                                                  # :351
        req = regex.sub(rf"({_B}|\))(\s+)({_B}|\()",                                 # :353
                        lambda m: regex.sub(m.group(2), "*", m.group(0)), req)

        def var_cb(m):                                                               # :357
            funcs.append({"word": m.group(0), "func": searchfunc or self.search_word, "nFound": 0})
            return "\x00%d\x00" % (len(funcs) - 1)

        req = regex.sub(rf"{_B}([{_W}]+){_B}", var_cb, req)
        return req, fields

    def _search_field(self, field, value):                                           # Search.cs:126-140
        seq = self.get("&" + _net_lower(field))
        if seq is None:
            return np.empty(0, np.uint16), 0
        new, Rn = self.search_word(_net_lower(value))
        return seq_and(seq, -1, new, Rn)

    def _d(self, fn, dist):                                                          # Search.cs:302-312
        units, R = fn["func"](fn["word"])
        fn["nFound"] = int(seq_decode(units).size)
        wl = len(fn["word"])
        return units, (-wl - 4 if R < 0 else dist + wl)

    def _eval(self, expr, funcs, dist):
        """C# expression semantics as DynamicExpresso applies them: * before +, left-assoc, parens."""
        toks = regex.findall(r"\x00\d+\x00|[*+()]|\S", expr)
        pos = [0]

        def peek():
            return toks[pos[0]] if pos[0] < len(toks) else None

        def factor():
            t = peek()
            if t is None:
                raise SyntaxError("eof")
            pos[0] += 1
            if t == "(":
                v = expr_()
                if peek() != ")":
                    raise SyntaxError(")")
                pos[0] += 1
                return v
            if t[0] == "\x00":
                return self._d(funcs[int(t[1:-1])], dist)
            raise SyntaxError(t)

        def term():
            v = factor()
            while peek() == "*":
                pos[0] += 1
                r = factor()
                v = seq_and(v[0], v[1], r[0], r[1])
            return v

        def expr_():
            v = term()
            while peek() == "+":
                pos[0] += 1
                r = term()
                v = seq_or(v[0], v[1], r[0], r[1])
            return v

        v = expr_()
        if pos[0] != len(toks):
            raise SyntaxError("trailing")
        return v

    def _prepare_result(self, units, filt):
        mask = None
        if filt:
            m = np.zeros(self.n_docs(), np.uint8)
            for d in range(self.n_docs()):
                name = self.doc_name(d)
                m[d] = 1 if any(regex.search(f, name) for f in filt) else 0
            mask = m
        u = np.ascontiguousarray(units, np.uint16)
        return self.L.orc_prepare_result(self.h, u.ctypes.data, u.size, mask.ctypes.data if mask is not None else None)

    def prepare_result_masked(self, units, mask):
        """PrepareSearchResult (Search.cs:365-420) with an explicit per-document filter mask, then the doc ranks /
        order of Search.cs:552-557, 599 -> [(doc, rank, n_pages)]."""
        u = np.ascontiguousarray(units, np.uint16)
        m = np.ascontiguousarray(mask, np.uint8)
        r = self.L.orc_prepare_result(self.h, u.ctypes.data, u.size, m.ctypes.data)
        try:
            self.L.orc_result_finish(self.h, r)
            return [(self.L.orc_result_found_doc(r, i), self.L.orc_result_found_doc_rank(r, i), self.L.orc_result_found_doc_npages(r, i))
                    for i in range(self.L.orc_result_n_found_docs(r))]
        finally:
            self.L.orc_result_free(r)

    def search(self, req, opt=None):
        """Index.Search, Search.cs:437-617 (snippets/headers omitted: out of scope)."""
        try:
            req = _net_lower(req)                                                        # :453
            filt = []
            fpat = r"\B-filter:([\w\*\?\\.()+{}/]+,?)+"
            m = regex.search(fpat, req)
            if m:
                filt = [c.strip(",") for c in m.captures(1)]
            req = regex.sub(fpat, " ", req)
            funcs = []
            req, fields = self._prepare_search_request(req, funcs)
            R = 255 if opt is None else opt.dist                                         # :472-473
            res = resf = None
            try:
                if len(req) > 0:
                    res = self._eval(req, funcs, R)
                if len(fields) > 0:
                    resf = self._eval(fields, funcs, R)
            except SyntaxError:
                return SearchResult(False, "Syntax Error in search request")
            if res is None:
                res = resf
            if res is None:
                return SearchResult()
            r = self._prepare_result(res[0], filt)
            r2 = None
            try:
                if self.L.orc_result_error(r):
                    return SearchResult(False, "Error: coordinate beyond the last page")
                if resf is not None:
                    r2 = self._prepare_result(resf[0], [])
                    if self.L.orc_result_error(r2):
                        return SearchResult(False, "Error: coordinate beyond the last page")
                    self.L.orc_result_intersect_docs(self.h, r, r2)
                self.L.orc_result_finish(self.h, r)
                out = SearchResult()
                for i in range(self.L.orc_result_n_pages(r)):
                    n = self.L.orc_result_page_npos(r, i)
                    pos = np.empty(n, np.int32)
                    self.L.orc_result_page_pos(r, i, pos.ctypes.data)
                    pi = self.L.orc_result_page_index(r, i)
                    out.foundPages.append(ResultPage(pi, self.page_id(pi), pos, self.L.orc_result_page_rank(r, i)))
                for i in range(self.L.orc_result_n_found_docs(r)):
                    d = self.L.orc_result_found_doc(r, i)
                    out.foundDocs.append(ResultDoc(d, self.doc_name(d), self.L.orc_result_found_doc_rank(r, i),
                                                   self.L.orc_result_found_doc_npages(r, i)))
                out.words = [(f["word"], f["nFound"]) for f in funcs]
                return out
            finally:
                self.L.orc_result_free(r)
                if r2:
                    self.L.orc_result_free(r2)
        except Exception as e:  # Search.cs:609-613
            return SearchResult(False, "Error: %s" % e)

    def query_batch(self, prog, prog_off):
        """CPU-baseline batch: postfix programs over ordinal term indices (see oracle.cpp)."""
        prog = np.ascontiguousarray(prog, np.int32)
        off = np.ascontiguousarray(prog_off, np.uint64)
        nq = off.size - 1
        out = [np.zeros(nq, np.uint64) for _ in range(4)]
        self.L.orc_query_batch(self.h, prog.ctypes.data, off.ctypes.data, nq, *[o.ctypes.data for o in out])
        return out

    def query_batch_mt(self, prog, prog_off, n_threads):
        """The same batch sharded by query over n_threads threads that share this index's posting lists and page table (each
        with a page cursor of its own): the all-cores baseline of SURVEY 8(d)."""
        prog = np.ascontiguousarray(prog, np.int32)
        off = np.ascontiguousarray(prog_off, np.uint64)
        nq = off.size - 1
        out = [np.zeros(nq, np.uint64) for _ in range(4)]
        self.L.orc_query_batch_mt(self.h, prog.ctypes.data, off.ctypes.data, nq, int(n_threads), *[o.ctypes.data for o in out])
        return out


class Builder:
    """Index.Builder (Build.cs:258-437) + the IndexTask page loop (Build.cs:445-564)."""

    def __init__(self, index):
        self.index = index
        self.L = index.L
        self.h = self.L.orc_builder_new(index.h)

    def __del__(self):
        try:
            self.L.orc_builder_free(self.h)
        except Exception:
            pass

    def add_doc(self, source, name):
        a, p, n = _u16(source)
        b, q, m = _u16(name)
        self.L.orc_builder_add_doc(self.h, p, n, q, m)

    def add_word(self, word, coord):
        a, p, n = _u16(word)
        self.L.orc_builder_add_word(self.h, p, n, int(coord))

    def add(self, code, coord):
        a, p, n = _u16(code)
        self.L.orc_builder_add(self.h, p, n, int(coord))

    def end_page(self, id):
        a, p, n = _u16(id)
        self.L.orc_builder_end_page(self.h, p, n)

    def index_page(self, id, text):
        a, p, n = _u16(id)
        b, q, m = _u16(text)
        self.L.orc_builder_index_page(self.h, p, n, q, m)

    def index_docs_latin1(self, text, offsets, first_doc=0, source="syn", page_id="1"):
        t = np.ascontiguousarray(text, np.uint8)
        o = np.ascontiguousarray(offsets, np.uint64)
        a, p, n = _u16(source)
        b, q, m = _u16(page_id)
        self.L.orc_builder_index_docs_latin1(self.h, t.ctypes.data, o.ctypes.data, o.size - 1, int(first_doc), p, n, q, m)

    @property
    def total_count(self):
        return self.L.orc_builder_total_count(self.h)

    @property
    def max_coord(self):
        return self.L.orc_builder_max_coord(self.h)

    @property
    def coord(self):
        return self.L.orc_builder_coord(self.h)

    @property
    def n_terms(self):
        return self.L.orc_builder_n_terms(self.h)

    def set_max_items(self, n):
        """Index.MaxTmpIndexItems (Index.cs:120): a temp index is flushed whenever TotalCount exceeds n (Build.cs:335-340)."""
        self.L.orc_builder_set_max_items(self.h, int(n))

    @property
    def n_tmp(self):
        return self.L.orc_builder_n_tmp(self.h)

    def tmp_images(self):
        """The flushed N.tmpind files, in creation order."""
        out = []
        for i in range(self.n_tmp):
            n = C.c_size_t()
            p = self.L.orc_builder_tmp(self.h, i, C.byref(n))
            out.append(C.string_at(p, n.value))
        return out

    def save_merged(self):
        """".index" bytes of a task that flushed temp indexes: Save of the rest + MergeAll (Build.cs:541, Index.cs:488, 537-573)."""
        ib, inn = C.c_void_p(), C.c_size_t()
        self.L.orc_builder_save_merged(self.h, C.byref(ib), C.byref(inn))
        try:
            return C.string_at(ib.value, inn.value)
        finally:
            self.L.orc_free(ib)

    def save(self):
        """-> (bytes of .index, bytes of .index.list)  (Build.cs:370-386, 99-118)."""
        ib, lb = C.c_void_p(), C.c_void_p()
        inn, ln = C.c_size_t(), C.c_size_t()
        self.L.orc_builder_save(self.h, C.byref(ib), C.byref(inn), C.byref(lb), C.byref(ln))
        try:
            return C.string_at(ib.value, inn.value), C.string_at(lb.value, ln.value)
        finally:
            self.L.orc_free(ib)
            self.L.orc_free(lb)

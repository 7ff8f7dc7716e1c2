"""Index.Search front end (reference: Docodo.NET/Search.cs:192-260, 302-363, 437-617).

Host-side string work that stays with the host (the C# shim keeps its own PrepareSearchRequest):
the request string is rewritten exactly as the reference does, every word becomes a term lookup
with the R of SearchSequence.d(), and the expression is compiled to ONE postfix program that the
GPU evaluates (`dcd_query_batch`).  GetWordCodes is the identity (no stemmer / vocabulary / stop
words: the parity configuration); a host with linguistics supplies `get_word_codes`.
Snippets/headers (Search.cs:560-595) are out of scope.
"""
import numpy as np
import regex

from . import _lib

_W = r"\p{L}\p{Mn}\p{Nd}\p{Pc}"                                        # .NET \w
_B = rf"(?:(?<=[{_W}])(?![{_W}])|(?<![{_W}])(?=[{_W}]))"               # .NET \b
MAX_LIKE_WORDS = 100                                                   # Search.cs:158


def _upper(s):
    return "".join(c.upper() if len(c.upper()) == 1 else c for c in s)


def _net_regex(pat):
    """A -filter: item is a .NET regex (Regex.Match(doc.Name, filt), Search.cs:391); the character class the request
    syntax admits ([\\w*?\\\\.()+{}/]) is spelled the same way for the `regex` module."""
    return pat


class _TwoResults:
    """Main query evaluated with the document filter, field query without (Search.cs:543-547)."""

    def __init__(self, a, b):
        self.parts = [a] + ([b] if b is not None else [])
        self.n_queries = len(self.parts)
        self.status = [p.status[0] for p in self.parts]

    def pages(self, q):
        return self.parts[q].pages(0)

    def docs(self, q):
        return self.parts[q].docs(0)


def _lower(s):
    return "".join(c.lower() if len(c.lower()) == 1 else ("i" if c == "İ" else c) for c in s)


class SearchResult:  # Search.cs:20-56
    def __init__(self, success=True, error=""):
        self.Success, self.Error = success, error
        self.foundPages = []   # (page index, page id, positions ndarray, rank)
        self.foundDocs = []    # (doc index, doc name, rank, n pages), ascending rank
        self.words = []        # (word, nFound)


class Search:
    def __init__(self, index, get_word_codes=None, stop_words=(), keep_forms=True):
        self.ix = index
        self.get_word_codes = get_word_codes or (lambda w: [w] if w else None)
        self.stop_words = list(stop_words)
        self.keep_forms = keep_forms
        self._keys = None

    # -- SearchWord, Search.cs:192-260: returns (postfix program for the OR of the codes, exact flag)
    def _keys_list(self):
        if self._keys is None:
            self._keys = [self.ix.key(t) for t in range(self.ix.Count)]
        return self._keys

    def _search_word(self, word):
        exact = _upper(word) == word and self.keep_forms
        word = _lower(word)
        words = [word]
        if "_" in word:
            if not self.keep_forms:
                return [], False, 0
            exact = True
            words = []
            if len(word) >= 2:                                             # GetLikeWords, Search.cs:160-167: dictionary scan on the device
                words = [self.ix.key(t) for t in self.ix.like_words(word, MAX_LIKE_WORDS)]
        prog, n, nfound = [], 0, 0
        for w in words:
            codes = self.get_word_codes(w)
            if not codes:
                continue
            selfcodes = [c for c in codes if regex.match(rf"[{_W}]", c[0])]
            know = [c for c in dict.fromkeys(codes) if c not in selfcodes]
            use = know if (not exact and know) else selfcodes[:1]
            for code in use:
                t = self.ix.find(code)
                if t is None:
                    continue
                prog += [_lib.DCD_OP_TERM, t, 0]
                if n:
                    prog.append(_lib.DCD_OP_OR)
                n += 1
        return prog, exact, n

    # -- PrepareSearchRequest, Search.cs:319-363 (literal regex pipeline)
    def _prepare(self, req, funcs, field=None, keepshort=False):
        req = regex.sub(rf'[^{_W}(){{}}=~?|"]|_+', " ", req)
        f = [""]

        def field_cb(m):
            l, _ = self._prepare(m.group(2), funcs, field=m.group(1), keepshort=True)
            f[0] += ("*" if f[0] else "") + "(" + l + ")"
            return ""

        req = regex.sub(rf"{{*([{_W}]+)[ ]*=([{_W}|() ]+)}}", field_cb, req)
        fields = f[0]
        req = regex.sub(r"{.*}", "", req)
        req = req.replace("?", "_")
        if not keepshort:
            req = regex.sub(rf"{_B}[{_W}]{{1,2}}{_B}", " ", req)
        for st in self.stop_words:
            req = regex.sub(rf"{_B}{st}{_B}", "", req)
        req = regex.sub(r'"(.*)"', lambda m: "(" + _upper(m.group(1)) + ")", req)
        req = req.replace("|", "+")
        req = regex.sub(rf"({_B}|\))(\s+)({_B}|\()", lambda m: regex.sub(m.group(2), "*", m.group(0)), req)

        def var_cb(m):
            funcs.append({"word": m.group(0), "field": field})
            return "\x00%d\x00" % (len(funcs) - 1)

        req = regex.sub(rf"{_B}([{_W}]+){_B}", var_cb, req)
        return req, fields

    # -- SearchSequence.d(), Search.cs:302-312, compiled to postfix
    def _d(self, fn, dist):
        word = fn["word"]
        if fn["field"] is None:
            prog, exact, n = self._search_word(word)
            R0 = -1 if exact else 0
        else:                                                                  # SearchField, Search.cs:126-140
            t = self.ix.find("&" + _lower(fn["field"]))
            vprog, vexact, n = self._search_word(_lower(word))
            if t is None or n == 0:
                prog, n, R0 = [], 0, 0
            else:
                vR = -1 if vexact else 0
                prog = [_lib.DCD_OP_TERM, t, -1] + vprog + [_lib.DCD_OP_SETR, vR, _lib.DCD_OP_AND]
                R0 = -1 if vR < 0 else 1
        wl = len(word)
        R = (-wl - 4) if R0 < 0 else dist + wl
        fn["n"] = n
        if n == 0:
            return None, R                                                     # empty IndexSequence
        return prog + [_lib.DCD_OP_SETR, R], R

    def _compile(self, expr, funcs, dist):
        toks = regex.findall(r"\x00\d+\x00|[*+()]|\S", expr)
        pos = [0]

        def peek():
            return toks[pos[0]] if pos[0] < len(toks) else None

        # a value is (program or None for the empty sequence, R)
        def combine(a, b, op):
            absR = max(abs(a[1]), abs(b[1]))
            R = -absR if (a[1] < 0 and b[1] < 0) else absR
            if op == _lib.DCD_OP_AND:
                if a[0] is None or b[0] is None:
                    return None, R                                             # nothing can contain both operands
                return a[0] + b[0] + [op], R
            if a[0] is None and b[0] is None:
                return None, R
            if a[0] is None:
                return b[0] + [_lib.DCD_OP_SETR, R], R
            if b[0] is None:
                return a[0] + [_lib.DCD_OP_SETR, R], R
            return a[0] + b[0] + [op], R

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
                v = combine(v, factor(), _lib.DCD_OP_AND)
            return v

        def expr_():
            v = term()
            while peek() == "+":
                pos[0] += 1
                v = combine(v, term(), _lib.DCD_OP_OR)
            return v

        v = expr_()
        if pos[0] != len(toks):
            raise SyntaxError("trailing")
        return v

    def _query(self, flat, off, doc_match, nprog):
        if doc_match is None or len(off) == 2:
            return self.ix.query_batch(flat, off, doc_match=doc_match)
        a = self.ix.query_batch(flat[:off[1]], [0, off[1]], doc_match=doc_match)
        b = self.ix.query_batch(flat[off[1]:off[2]], [0, off[2] - off[1]])
        return _TwoResults(a, b)

    def GetCloseWords(self, word):
        """Search.cs:168-174: the 10 keys closest to `word` (Levenshtein; a dictionary scan on the device)."""
        return [self.ix.key(t) for t in self.ix.close_words(word, 10)]

    def GetSuggessions(self, req, n=10):
        """Search.cs:176-188: completions of the last word of `req`, most frequent (longest posting lists) first."""
        if len(req) < 2:
            return []
        parts = [p for p in regex.split(r"\b", req) if len(p) > 0]
        if not parts:
            return []
        last = _lower(parts[-1])
        if len(last) < 2:
            return []
        return [self.ix.key(t)[len(last):] for t in self.ix.suggest(last, n)]

    # -- one request = plan (host string work) + evaluation (device) + result assembly (host) ---------------------
    def _plan(self, req, opt):
        """-> SearchResult (nothing to evaluate: empty / syntax error) or a plan for _finish."""
        req = _lower(req)
        # search filter, Search.cs:455-466: every "item," of -filter: is a regex over the document names; the host
        # evaluates the regexes (one byte per document), the device keeps filtered-out documents out of foundDocs
        fpat = r"\B-filter:([\w\*\?\\.()+{}/]+,?)+"
        doc_match = None
        m = regex.search(fpat, req)
        if m:
            filters = [c.strip(",") for c in m.captures(1)]
            req = regex.sub(fpat, " ", req)
            pats = [regex.compile(_net_regex(f)) for f in filters]
            nd = self.ix.n_docs
            doc_match = np.fromiter((1 if any(p.search(self.ix.doc_name(d)) for p in pats) else 0 for d in range(nd)), np.uint8, nd)
        funcs = []
        req, fields = self._prepare(req, funcs)
        dist = 255 if opt is None else opt.dist                            # Search.cs:472-473
        try:
            res = self._compile(req, funcs, dist) if len(req) > 0 else None
            resf = self._compile(fields, funcs, dist) if len(fields) > 0 else None
        except SyntaxError:
            return SearchResult(False, "Syntax Error in search request")
        if res is None:
            res = resf
        if res is None:
            return SearchResult()
        out = SearchResult()
        out.words = [(f["word"], None) for f in funcs]
        progs = [p[0] for p in (res, resf) if p is not None]
        if progs[0] is None:
            return out                                                     # empty result sequence
        flat, off = [], [0]
        for p in progs:
            flat += (p if p is not None else [])
            off.append(len(flat))
        if len(progs) == 2 and progs[1] is None:
            off = off[:2]
        return {"flat": flat, "off": off, "doc_match": doc_match, "out": out, "fields": resf is not None}

    def _finish(self, plan, r, q0, nq):
        """r: batch result holding this request's nq (1 or 2) queries from index q0."""
        out = plan["out"]
        if r.status[q0] != 0:
            return SearchResult(False, "Error: coordinate beyond the last page")
        keep = None
        if plan["fields"]:                                                 # CombineSearchResults, Search.cs:423-428
            keep = set()
            if nq == 2 and r.status[q0 + 1] == 0:
                keep = {self.ix.doc_name(d) for d, _, _ in r.docs(q0 + 1)}
        for pi, pos, rank in r.pages(q0):
            out.foundPages.append((pi, self.ix.page_info(pi)[2], pos, rank))
        for d, rank, npg in r.docs(q0):
            name = self.ix.doc_name(d)
            if keep is None or name in keep:
                out.foundDocs.append((d, name, rank, npg))
        return out

    def __call__(self, req, opt=None):
        """Index.Search(req, opt), Search.cs:437-617."""
        try:
            plan = self._plan(req, opt)
            if isinstance(plan, SearchResult):
                return plan
            r = self._query(plan["flat"], plan["off"], plan["doc_match"], len(plan["off"]) - 1)
            return self._finish(plan, r, 0, r.n_queries)
        except _lib.DocodoError as e:
            return SearchResult(False, "Error: %s" % e)

    def search_many(self, requests, opt=None):
        """Many requests, ONE dcd_query_batch: the reference serialises Search under a lock and serves requests from a
        thread pool (Search.cs:448, server.cs:16-30); batching them is what feeds the GPU.  Requests with a -filter:
        carry their own per-document mask and are evaluated on their own.  Returns the SearchResults in request order."""
        results = [None] * len(requests)
        plans = []
        flat, off = [], [0]
        for i, req in enumerate(requests):
            try:
                plan = self._plan(req, opt)
            except _lib.DocodoError as e:
                plan = SearchResult(False, "Error: %s" % e)
            if isinstance(plan, SearchResult):
                results[i] = plan
            elif plan["doc_match"] is not None:
                results[i] = self(req, opt)
            else:
                q0 = len(off) - 1
                base = len(flat)
                flat += plan["flat"]
                off += [base + o for o in plan["off"][1:]]
                plans.append((i, plan, q0, len(plan["off"]) - 1))
        if plans:
            try:
                r = self.ix.query_batch(flat, off)
                for i, plan, q0, nq in plans:
                    results[i] = self._finish(plan, r, q0, nq)
            except _lib.DocodoError as e:
                for i, _, _, _ in plans:
                    results[i] = SearchResult(False, "Error: %s" % e)
        return results


class MicroBatcher:
    """Batched serving entry: callers on any thread submit requests and get futures; a worker thread
    collects what arrives within `max_wait_ms` (or up to `max_batch` requests) and evaluates it with ONE
    Search.search_many call -- the role of the REST server's request threads + DoSearchLock in the reference
    (server.cs:24-34, Search.cs:448), turned into GPU-sized batches."""

    def __init__(self, search, max_batch=4096, max_wait_ms=2.0):
        import queue
        import threading
        self.search, self.max_batch, self.max_wait = search, max_batch, max_wait_ms / 1e3
        self.q = queue.Queue()
        self.batches = 0
        self._stop = False
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def submit(self, req, opt=None):
        from concurrent.futures import Future
        f = Future()
        self.q.put((req, opt, f))
        return f

    def close(self):
        self._stop = True
        self.q.put(None)
        self.t.join()

    def _run(self):
        import queue
        import time
        while not self._stop:
            item = self.q.get()
            if item is None:
                break
            batch = [item]
            deadline = time.perf_counter() + self.max_wait
            while len(batch) < self.max_batch:
                left = deadline - time.perf_counter()
                if left <= 0:
                    break
                try:
                    nxt = self.q.get(timeout=left)
                except queue.Empty:
                    break
                if nxt is None:
                    self._stop = True
                    break
                batch.append(nxt)
            # requests with the same options share a call
            groups = {}
            for req, opt, f in batch:
                groups.setdefault(None if opt is None else opt.dist, []).append((req, opt, f))
            for _, items in groups.items():
                try:
                    res = self.search.search_many([r for r, _, _ in items], items[0][1])
                    for (_, _, f), x in zip(items, res):
                        f.set_result(x)
                except Exception as e:   # noqa: BLE001 -- a failed batch must not kill the worker
                    for _, _, f in items:
                        f.set_exception(e)
            self.batches += 1

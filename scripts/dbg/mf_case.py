"""debug: the operator cases of tests/test_gpu_query.py through k_pair_stream, first difference against the oracle"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import docodo_amd, oracle as o
from test_gpu_query import _ops_cases
for Ra, Rb in [(258, 259), (-7, -9), (0, 0)]:
    for a, b in _ops_cases():
        ua, ub = o.seq_encode(a), o.seq_encode(b)
        for name, fo, fg in (("and", o.seq_and, lambda x, y: x * y), ("or", o.seq_or, lambda x, y: x + y)):
            want, _ = fo(ua, Ra, ub, Rb)
            got = fg(docodo_amd.IndexSequence(ua, Ra), docodo_amd.IndexSequence(ub, Rb)).units
            if np.array_equal(got, want):
                continue
            w, g = o.seq_decode(want), o.seq_decode(got)
            n = min(w.size, g.size)
            d = int(np.argmax(w[:n] != g[:n])) if (w[:n] != g[:n]).any() else n
            print(name, Ra, Rb, a.size, b.size, "want", w.size, "got", g.size, "first diff at", d)
            print("  want", w[max(0, d - 3):d + 6])
            print("  got ", g[max(0, d - 3):d + 6])
            v = int(w[d]) if d < w.size else int(g[d])
            ia, ib = int(np.searchsorted(a, v)), int(np.searchsorted(b, v))
            print("  a around", ia, a[max(0, ia - 3):ia + 4])
            print("  b around", ib, b[max(0, ib - 3):ib + 4])
            # position in the union (tile-relative)
            print("  union index", ia + ib, "lane", ((ia + ib) % 256) // 4, "k", (ia + ib) % 4)

"""debug: dcd_seq_and / dcd_seq_or on small operands through k_pair_stream vs the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle as o
import docodo_amd
rng = np.random.default_rng(1)
def seq(n, g, s=0):
    return (np.cumsum(rng.integers(1, g, n, dtype=np.uint64)) + np.uint64(s)).astype(np.uint64) if n else np.array([], np.uint64)
cases = [(0, 5, 10, 10), (5, 0, 10, 10), (1, 1, 5, 5), (3, 4, 5, 5), (50, 60, 40, 30), (300, 280, 40, 30), (1000, 1200, 300, 250), (3000, 3000, 3, 3), (20000, 17000, 700, 900)]
bad = 0
for na, nb, ga, gb in cases:
    a, b = seq(na, ga), seq(nb, gb)
    ua, ub = o.seq_encode(a), o.seq_encode(b)
    for Ra, Rb in [(258, 259), (-7, -9), (0, 0), (40, -3)]:
        sa, sb = docodo_amd.IndexSequence(ua, Ra), docodo_amd.IndexSequence(ub, Rb)
        for name, f, g in (("and", lambda: sa * sb, o.seq_and), ("or", lambda: sa + sb, o.seq_or)):
            want, wR = g(ua, Ra, ub, Rb)
            try:
                got = f()
                ok = got.R == wR and np.array_equal(got.units, want)
                msg = "" if ok else " got %d units, want %d; first diff at %s" % (got.units.size, want.size, next((i for i in range(min(got.units.size, want.size)) if got.units[i] != want[i]), None))
            except Exception as e:
                ok, msg = False, " EXC " + str(e)[-80:]
            if not ok:
                bad += 1
                print("MISMATCH", name, na, nb, Ra, Rb, msg, flush=True)
print("done, bad =", bad)

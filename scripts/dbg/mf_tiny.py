import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import docodo_amd, oracle as o
def run(a, b, Ra=258, Rb=259):
    a, b = np.array(a, np.uint64), np.array(b, np.uint64)
    ua, ub = o.seq_encode(a), o.seq_encode(b)
    for name, fo, fg in (("or", o.seq_or, lambda x, y: x + y), ("and", o.seq_and, lambda x, y: x * y)):
        want, _ = fo(ua, Ra, ub, Rb)
        got = fg(docodo_amd.IndexSequence(ua, Ra), docodo_amd.IndexSequence(ub, Rb)).units
        print(name, list(a), list(b), "want", list(o.seq_decode(want)), "got", list(o.seq_decode(got)), "OK" if np.array_equal(got, want) else "BAD")
run([5], [5])
run([1, 5], [5])
run([1, 2, 5], [5])
run([1, 2, 3, 5], [5])
run([1, 2, 3, 4, 5], [5])
run([5, 6, 7], [5, 6, 7])
run([5, 9], [5, 9, 13])
run(list(range(10, 100, 10)), list(range(10, 100, 10)))

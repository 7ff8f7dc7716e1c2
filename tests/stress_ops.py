"""Randomised differential test of the '*' / '+' operators (GPU through the C ABI vs the CPU oracle): random sizes,
densities, shared values, R combinations and partition sizes.  Not collected by pytest (long); run on a GPU box:
    python tests/stress_ops.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as o
import docodo_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)


def seq(n, max_gap, start):
    if n == 0:
        return np.array([], np.uint64)
    return (np.cumsum(rng.integers(1, max_gap, n, dtype=np.uint64)) + np.uint64(start)).astype(np.uint64)


t0 = time.time()
bad = 0
for c in range(n_cases):
    na = int(rng.choice([0, 1, 3, 50, 700, 5000, 40000, 150000]) * rng.uniform(0.5, 1.5))
    nb = int(rng.choice([0, 1, 3, 50, 700, 5000, 40000, 150000]) * rng.uniform(0.5, 1.5))
    ga, gb = int(rng.choice([2, 5, 40, 300, 3000, 50000, 5_000_000])), int(rng.choice([2, 5, 40, 300, 3000, 50000, 5_000_000]))
    a, b = seq(na, ga, int(rng.integers(0, 1000))), seq(nb, gb, int(rng.integers(0, 1000)))
    if na and nb and rng.random() < 0.5:       # shared values (double-tagged elements)
        k = max(1, min(na, nb) // int(rng.choice([2, 10, 100])))
        b = np.unique(np.concatenate([b, a[rng.integers(0, na, k)]])).astype(np.uint64)
    Ra = int(rng.choice([258, 0, -7, 40, -1, 100000, 3, -300]))
    Rb = int(rng.choice([259, 0, -9, 3, -1, 1, -300, 270]))
    part = rng.choice([None, None, 64, 300, 1024, 4096])
    if part is None:
        os.environ.pop("DOCODO_QUERY_PART", None)
    else:
        os.environ["DOCODO_QUERY_PART"] = str(int(part))
    ua, ub = o.seq_encode(a), o.seq_encode(b)
    sa, sb = docodo_amd.IndexSequence(ua, Ra), docodo_amd.IndexSequence(ub, Rb)
    for name, want, got in (("and", o.seq_and(ua, Ra, ub, Rb), sa * sb), ("or", o.seq_or(ua, Ra, ub, Rb), sa + sb)):
        if got.R != want[1] or not np.array_equal(got.units, want[0]):
            bad += 1
            print("MISMATCH", name, dict(case=c, na=a.size, nb=b.size, ga=ga, gb=gb, Ra=Ra, Rb=Rb, part=part, seed=seed), flush=True)
    if c % 50 == 49:
        print("%d cases, %d mismatches, %.0f s" % (c + 1, bad, time.time() - t0), flush=True)
print("done: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)

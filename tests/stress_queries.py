"""Randomised differential test of batched query programs on a synthetic index (GPU through the C ABI vs the CPU oracle):
2-4 terms per query drawn from all frequencies, random mix of '*' / '+', proximity and ordered R values.  The final
IndexSequence of every query (units + R) must match the oracle's operators applied to the same posting lists.  Not
collected by pytest (long); run on a GPU box:  python tests/stress_queries.py [n_docs] [n_queries] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as o
import docodo_amd
from docodo_amd import synth

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_q = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rng = np.random.default_rng(seed)
text = synth.corpus_host(n_docs)
b = docodo_amd.Builder()
b.AddPages(text, np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES, docs_per_page=True)
b.Finish()
ix = b.Build()
T, A, O = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND, docodo_amd.DCD_OP_OR
units_cache = {}
def units(t):
    if t not in units_cache:
        units_cache[t] = ix.get(t).units
    return units_cache[t]
prog, off, plans = [], [0], []
for q in range(n_q):
    k = int(rng.integers(2, 5))
    ranks = [int(r) for r in np.minimum(99999, (rng.pareto(0.7, k) * 3).astype(np.int64) + 1)]   # heavy head, long tail
    ts = [ix.find(synth.term(r)) for r in ranks]
    if any(t is None for t in ts):
        continue
    mode = rng.integers(0, 3)
    plan = []
    for i, (r, t) in enumerate(zip(ranks, ts)):
        L = len(synth.term(r))
        R = (255 + L) if mode == 0 else (-(L + 4) if mode == 1 else int(rng.choice([40, 3, 0, -7, 1000])))
        prog += [T, t, R]
        plan.append(("t", t, R))
        if i:
            op = A if rng.random() < 0.75 else O
            prog.append(op)
            plan.append(("o", op))
    off.append(len(prog))
    plans.append(plan)
t0 = time.time()
res = ix.query_batch(prog, off, want_sequences=True)
print("GPU batch of %d queries: %.1f ms device" % (len(plans), res.ms_device), flush=True)
bad = 0
for q, plan in enumerate(plans):
    st = []
    for e in plan:
        if e[0] == "t":
            st.append((units(e[1]), e[2]))
        else:
            (ub, Rb), (ua, Ra) = st.pop(), st.pop()
            st.append(o.seq_and(ua, Ra, ub, Rb) if e[1] == A else o.seq_or(ua, Ra, ub, Rb))
    want_u, want_R = st[0]
    got_u = res.seq_units[int(res.seq_off[q]):int(res.seq_off[q + 1])]
    if int(res.seq_R[q]) != want_R or not np.array_equal(got_u, want_u):
        bad += 1
        print("MISMATCH query", q, plan, flush=True)
    if q % 250 == 249:
        print("%d queries checked, %d mismatches, %.0f s" % (q + 1, bad, time.time() - t0), flush=True)
print("done: %d queries, %d mismatches" % (len(plans), bad))
sys.exit(1 if bad else 0)

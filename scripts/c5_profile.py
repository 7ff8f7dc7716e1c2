"""One C5 sub-batch (exact 2- / 3-word phrases sampled from corpus positions) on the 10 M-doc index, for rocprofv3 --kernel-trace --stats.
usage: python scripts/c5_profile.py [docs] [queries] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth, dist
import bench
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 2_500
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = torch.empty(n_docs * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n_docs)
sb = dist.SegmentedBuilder(1024, max_segment_chars=3_584_000_000)
sb.add_pages_dev(d.data_ptr(), n_docs)
ix = sb.finish()
del d
torch.cuda.empty_cache()
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
prog, off = [], [0]
sizes = []
for q in bench.phrase_queries(nq, n_docs, 77):
    ts = [ix.find(w) for w, _ in q]
    if any(t is None for t in ts):
        continue
    for k, ((w, r), t) in enumerate(zip(q, ts)):
        prog += [T, t, r]
        if k:
            prog.append(A)
    sizes.append(sorted(ix.term_count(t) for t in ts))
    off.append(len(prog))
pa, oa = np.array(prog, np.int32), np.array(off, np.uint64)
tot = sum(sum(s) for s in sizes)
print("queries %d, postings in the term lists %.3e (%.1f M per query); largest list per query: median %.2f M, mean %.2f M" %
      (len(sizes), tot, tot / len(sizes) / 1e6, np.median([s[-1] for s in sizes]) / 1e6, np.mean([s[-1] for s in sizes]) / 1e6), flush=True)
ix.query_batch(pa[:oa[8]], oa[:9], totals=True)
for r in range(reps):
    t0 = time.perf_counter()
    res = ix.query_batch(pa, oa, totals=True)
    print("rep %d: %d queries device %.2f ms wall %.2f ms -> %.0f q/s  (%s)" % (r, oa.size - 1, res["ms"], (time.perf_counter() - t0) * 1e3,
                                                                            (oa.size - 1) / res["ms"] * 1e3, {k: v for k, v in res.items() if k != "ms"}), flush=True)

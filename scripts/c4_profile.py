"""One C4 sub-batch (20 k three-term AND queries, Dist = 255) on the 10 M-doc index, for rocprofv3 --kernel-trace --stats.
usage: python scripts/c4_profile.py [docs] [queries] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth, dist
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
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
for row in synth.query_ranks(nq, 3):
    ws = [synth.term(int(k)) for k in row]
    ts = [ix.find(w) for w in ws]
    if any(t is None for t in ts):
        continue
    for k, (w, t) in enumerate(zip(ws, ts)):
        prog += [T, t, 255 + len(w)]
        if k:
            prog.append(A)
    off.append(len(prog))
pa, oa = np.array(prog, np.int32), np.array(off, np.uint64)
ix.query_batch(pa[:oa[8]], oa[:9], keep_handle=True)
for r in range(reps):
    t0 = time.perf_counter()
    ms = ix.query_batch(pa, oa, keep_handle=True)
    print("rep %d: %d queries device %.2f ms wall %.2f ms -> %.0f q/s" % (r, oa.size - 1, ms, (time.perf_counter() - t0) * 1e3, (oa.size - 1) / ms * 1e3), flush=True)
if os.environ.get("C4_SIZES"):
    cnt = np.array([[ix.term_count(int(pa[int(o) + 1])), ix.term_count(int(pa[int(o) + 4])), ix.term_count(int(pa[int(o) + 8]))] for o in oa[:-1]], dtype=np.float64)
    ns, nl = np.minimum(cnt[:, 0], cnt[:, 1]), np.maximum(cnt[:, 0], cnt[:, 1])
    g = nl >= 2 * ns
    print("round 0: sum short %.3e long %.3e; gallop ops %.1f %% with short %.3e long %.3e; stream ops short+long %.3e; third term sum %.3e" %
          (ns.sum(), nl.sum(), 100 * g.mean(), ns[g].sum(), nl[g].sum(), (ns[~g] + nl[~g]).sum(), cnt[:, 2].sum()))

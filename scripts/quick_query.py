import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import docodo_amd
from docodo_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq1 = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
nq3 = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n)
b = docodo_amd.Builder(); b.AddUniformPagesDev(d.data_ptr(), n, 1024); st = b.Finish()
ix = b.Build()
print("built: N=%d V=%d" % (st.n_postings, st.n_terms), flush=True)
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
def plans(ranks):
    prog, off = [], [0]
    npost = 0
    for row in ranks:
        ts = [ix.find(synth.term(int(k))) for k in row]
        if any(t is None for t in ts): continue
        for k, t in enumerate(ts):
            prog += [T, t, 255 + len(synth.term(int(row[k])))]
            npost += ix.term_count(t)
            if k: prog.append(A)
        off.append(len(prog))
    return np.array(prog, np.int32), np.array(off, np.uint64), npost
for name, ranks in (("single", synth.query_ranks(nq1, 1)), ("and3", synth.query_ranks(nq3, 3))):
    prog, off, npost = plans(ranks)
    for rep in range(3):
        t = time.time()
        ms = ix.query_batch(prog, off, keep_handle=True)
        wall = time.time() - t
        print("%s: %d queries, %d postings in: dev %.2f ms wall %.1f ms -> %.0f q/s (dev), %.2f Gpostings/s" % (
            name, off.size - 1, npost, ms, wall * 1e3, (off.size - 1) / (ms / 1e3), npost / 1e9 / (ms / 1e3)), flush=True)

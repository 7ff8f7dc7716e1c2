"""Wall time of the pieces of one bench step (Builder construction, add, Finish, close) -- where the gap between the
device time and the wall time of a step comes from."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import docodo_amd
from docodo_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n)
for r in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); b = docodo_amd.Builder()
    t1 = time.perf_counter(); b.AddUniformPagesDev(d.data_ptr(), n, 1024)
    t2 = time.perf_counter(); st = b.Finish()
    t3 = time.perf_counter(); b.close()
    t4 = time.perf_counter()
    print("rep %d new %.3f add %.3f finish %.3f (device %.3f) close %.3f | total %.3f ms" % (
        r, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, st.ms_total, (t4 - t3) * 1e3, (t4 - t0) * 1e3), flush=True)

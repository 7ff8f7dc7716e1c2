"""Where the wall time of a build with explicit page offsets goes (dcd_build_add_pages_dev): host calls timed separately.
usage: [DOCODO_TRACE=1] python scripts/general_pages_time.py [docs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n)
off = np.arange(n + 1, dtype=np.uint64) * 1024
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = docodo_amd.Builder()
    t1 = time.perf_counter()
    b.AddPagesDev(d.data_ptr(), n * 1024, off, docs_per_page=True)
    t2 = time.perf_counter()
    st = b.Finish()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    b.close()
    t4 = time.perf_counter()
    print("rep %d: create %.2f ms, add_pages_dev %.2f ms, finish %.2f ms (device %.2f), close %.2f ms" %
          (rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, st.ms_total, (t4 - t3) * 1e3), flush=True)
u = docodo_amd.Builder()
t0 = time.perf_counter(); u.AddUniformPagesDev(d.data_ptr(), n, 1024); st = u.Finish(); torch.cuda.synchronize()
print("uniform: %.2f ms wall, %.2f device" % ((time.perf_counter() - t0) * 1e3, st.ms_total))

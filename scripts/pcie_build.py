"""PCIe-inclusive build rate: the text starts in HOST memory (what dcd_build_add_pages receives from a C# host)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
text = synth.corpus_host(n)                       # pageable numpy buffer
off = np.arange(n + 1, dtype=np.uint64) * synth.DOC_BYTES
pinned = torch.from_numpy(text).pin_memory().numpy()
for label, buf in (("pageable", text), ("pinned", pinned)):
    for rep in range(3):
        t0 = time.perf_counter()
        b = docodo_amd.Builder()
        b.AddPages(buf, off, docs_per_page=True)
        t1 = time.perf_counter()
        st = b.Finish()
        t2 = time.perf_counter()
        b.close()
    print("%s: add_pages (H2D + page table) %.1f ms, finish %.1f ms wall (%.2f ms device) -> %.1f GB/s end to end" % (
        label, (t1 - t0) * 1e3, (t2 - t1) * 1e3, st.ms_total, n * 1024 / 1e9 / (t2 - t0)), flush=True)

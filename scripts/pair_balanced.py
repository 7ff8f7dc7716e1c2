"""Balanced two-term AND pairs on a 3 M-doc index (streaming-merge micro-benchmark for PMC runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth
n = 3_000_000
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda"); synth.corpus_dev(d.data_ptr(), n)
b = docodo_amd.Builder(); b.AddUniformPagesDev(d.data_ptr(), n, 1024); b.Finish(); ix = b.Build()
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
rng = np.random.default_rng(5)
prog, off, npost = [], [0], 0
for r in rng.integers(200, 3000, 3000):
    wa, wb = synth.term(int(r)), synth.term(int(r) + 1)
    ta, tb = ix.find(wa), ix.find(wb)
    prog += [T, ta, 255 + len(wa), T, tb, 255 + len(wb), A]; npost += ix.term_count(ta) + ix.term_count(tb); off.append(len(prog))
prog, off = np.array(prog, np.int32), np.array(off, np.uint64)
ix.query_batch(prog[:off[2]], off[:3], keep_handle=True)
for _ in range(2):
    ms = ix.query_batch(prog, off, keep_handle=True)
print("balanced", npost, "postings", round(ms, 2), "ms", round(npost / 1e9 / (ms / 1e3), 1), "G/s")
if os.environ.get("DOCODO_PROF"):
    import ctypes as C
    from docodo_amd._lib import lib
    L = lib()
    L.dcd_debug_prof.argtypes = [C.c_void_p, C.c_int32]
    z = np.zeros(16, np.uint64)
    L.dcd_debug_prof(z.ctypes.data, 1)
    ix.query_batch(prog, off, keep_handle=True)
    L.dcd_debug_prof(z.ctypes.data, 0)
    names = ["op lookup", "operands/mode", "HBM split+fill", "tile consumption+prefetch", "lane split", "merge", "refill store", "ownership", "fast path", "general path", "end"]
    tot = float(z[:11].sum())
    print({n: round(float(v) / tot, 3) for n, v in zip(names, z[:11])})

"""Skewed (3:1) two-term AND pairs on a 3 M-doc index (gallop micro-benchmark for PMC runs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth
n = 3_000_000
mult = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda"); synth.corpus_dev(d.data_ptr(), n)
b = docodo_amd.Builder(); b.AddUniformPagesDev(d.data_ptr(), n, 1024); b.Finish(); ix = b.Build()
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
rng = np.random.default_rng(5)
prog, off, npost, ns = [], [0], 0, 0
for r in rng.integers(200, 3000, 3000):
    wa, wb = synth.term(int(r)), synth.term(int(r) * mult)
    ta, tb = ix.find(wa), ix.find(wb)
    if ta is None or tb is None: continue
    prog += [T, ta, 255 + len(wa), T, tb, 255 + len(wb), A]
    ca, cb = ix.term_count(ta), ix.term_count(tb); npost += ca + cb; ns += min(ca, cb); off.append(len(prog))
prog, off = np.array(prog, np.int32), np.array(off, np.uint64)
ix.query_batch(prog[:off[2]], off[:3], keep_handle=True)
for _ in range(2):
    ms = ix.query_batch(prog, off, keep_handle=True)
print("skewed x%d" % mult, npost, "postings", ns, "short", round(ms, 2), "ms", round(ns / 1e9 / (ms / 1e3), 1), "G short/s")

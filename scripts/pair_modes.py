"""Micro-benchmark of the '*' operator by operand shape on the 10 M-doc index: balanced pairs (streaming merge) vs skewed
pairs (galloping), two-term AND queries, Dist = 255."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth, dist
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = torch.empty(n_docs * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n_docs)
shard = 3_500_000
bounds = list(range(0, n_docs, shard)) + [n_docs]
exports = []
for g in range(len(bounds) - 1):
    a, b_ = bounds[g], bounds[g + 1]
    sb = docodo_amd.Builder(coord_base=a * 1024)
    sb.AddUniformPagesDev(d.data_ptr() + a * 1024, b_ - a, 1024, first_doc=a)
    sb.Finish(); exports.append(dist.export_shard(sb, 1)); sb.close()
ix = dist.local_merge(exports, 0); del exports
dist.set_uniform_pages(ix, n_docs, 1024)
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
def run(pairs, label):
    prog, off, npost, ns = [], [0], 0, 0
    for ra, rb in pairs:
        wa, wb = synth.term(ra), synth.term(rb)
        ta, tb = ix.find(wa), ix.find(wb)
        if ta is None or tb is None: continue
        prog += [T, ta, 255 + len(wa), T, tb, 255 + len(wb), A]
        ca, cb = ix.term_count(ta), ix.term_count(tb)
        npost += ca + cb; ns += min(ca, cb)
        off.append(len(prog))
    prog, off = np.array(prog, np.int32), np.array(off, np.uint64)
    ix.query_batch(prog[:off[2]], off[:3], keep_handle=True)
    ms = min(ix.query_batch(prog, off, keep_handle=True) for _ in range(2))
    print(label, {"pairs": off.size - 1, "postings": npost, "short_side": ns, "ms": round(ms, 2), "Gpostings_per_s": round(npost / 1e9 / (ms / 1e3), 1),
                  "G_short_elems_per_s": round(ns / 1e9 / (ms / 1e3), 2)}, flush=True)
rng = np.random.default_rng(5)
base = rng.integers(200, 3000, 3000)
run([(int(r), int(r) + 1) for r in base], "balanced(r,r+1)")
for mult in (3, 8, 32, 128):
    run([(int(r), int(r) * mult) for r in base if r * mult < 100000], "skewed(r,%dr)" % mult)

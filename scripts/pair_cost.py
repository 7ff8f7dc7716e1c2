"""Cost of ONE ordered '*' (a two-word phrase, R = -(len + 4)) by operand sizes, on the 10 M-doc index: the same pair evaluated
`copies` times in one batch (terms of neighbouring ranks, so the copies do not share cache lines), device ms per operator.
Run once per DOCODO_QUERY_GAL_RATIO setting to compare the streaming merge with the galloping path.
usage: python scripts/pair_cost.py [docs] [copies]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth, dist
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = torch.empty(n_docs * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n_docs)
sb = dist.SegmentedBuilder(1024, max_segment_chars=3_584_000_000)
sb.add_pages_dev(d.data_ptr(), n_docs)
ix = sb.finish()
del d
torch.cuda.empty_cache()
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
def batch(ra, rb):
    prog, off = [], [0]
    for c in range(copies):
        wa, wb = synth.term(ra + c), synth.term(rb + c)
        ta, tb = ix.find(wa), ix.find(wb)
        prog += [T, ta, -(len(wa) + 4), T, tb, -(len(wb) + 4), A]
        off.append(len(prog))
    return np.array(prog, np.int32), np.array(off, np.uint64)
pa, oa = batch(1000, 2000)
ix.query_batch(pa, oa, keep_handle=True)
print("gal_ratio", os.environ.get("DOCODO_QUERY_GAL_RATIO", "default"), flush=True)
pairs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[3:]]
for ra, rb in pairs or [(1, 10), (1, 20), (1, 40), (1, 100), (1, 300), (1, 1000), (1, 10000), (10, 20), (10, 40), (10, 100), (10, 1000), (100, 200), (100, 400),
               (100, 1000), (100, 10000), (1000, 2000), (1000, 10000)]:
    pa, oa = batch(ra, rb)
    na = sum(ix.term_count(int(pa[7 * c + 1])) for c in range(copies)) / copies
    nb = sum(ix.term_count(int(pa[7 * c + 4])) for c in range(copies)) / copies
    ix.query_batch(pa, oa, totals=True)
    best = min(ix.query_batch(pa, oa, totals=True)["ms"] for _ in range(3))
    print("ranks %6d %6d  |A| %10.0f |B| %10.0f  ratio %7.1f  %8.3f ms/op  %7.2f G union elements/s  %7.2f G short/s" %
          (ra, rb, na, nb, max(na, nb) / min(na, nb), best / copies, (na + nb) * copies / best / 1e6, min(na, nb) * copies / best / 1e6), flush=True)

"""BASELINE.json configs[2..4] on ONE MI355X: 10 M synthetic docs (10.24 GB text) built as doc-range shards of
< 2^32 characters with global coordinates, merged on the device (dcd_merge_shards), then
  C2-like: 10 k single-term queries, C4: 100 k three-term AND queries (Dist = 255), C5: exact phrases.
Prints one JSON line.  (Measurement script; the grading bench is bench.py.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth, dist

n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
n_and3 = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
n_phr = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
chunk = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
out = {"docs": n_docs, "text_bytes": n_docs * 1024}
d = torch.empty(n_docs * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n_docs)
torch.cuda.synchronize(); t0 = time.perf_counter()
sb = dist.SegmentedBuilder(1024, max_segment_chars=3_584_000_000)
sb.add_pages_dev(d.data_ptr(), n_docs)
n_seg, dev_ms, N = len(sb.exports), sb.device_ms, sb.n_postings
torch.cuda.synchronize(); t1 = time.perf_counter()
ix = sb.finish()
torch.cuda.synchronize(); t2 = time.perf_counter()
out["build"] = {"shards": n_seg, "postings": int(N), "terms": ix.Count, "shard_build_device_ms": round(dev_ms, 2),
                "build_plus_export_wall_ms": round((t1 - t0) * 1e3, 1), "merge_wall_ms": round((t2 - t1) * 1e3, 1),
                "text_GB_per_s_incl_merge": round(n_docs * 1024 / 1e9 / (t2 - t0), 2)}
print(json.dumps(out), flush=True)
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
term_cache = {}
def tid(w):
    if w not in term_cache: term_cache[w] = ix.find(w)
    return term_cache[w]
def run(queries, label, reps=1):
    """queries: list of list of (word, R)"""
    tot_ms, nq, npost, per = 0.0, 0, 0, []
    for c0 in range(0, len(queries), chunk):
        prog, off = [], [0]
        for q in queries[c0:c0 + chunk]:
            ts = [tid(w) for w, _ in q]
            if any(t is None for t in ts): continue
            for k, ((w, R), t) in enumerate(zip(q, ts)):
                prog += [T, t, R]; npost += ix.term_count(t)
                if k: prog.append(A)
            off.append(len(prog))
        if len(off) == 1: continue
        pa, oa = np.array(prog, np.int32), np.array(off, np.uint64)
        ms = min(ix.query_batch(pa, oa, keep_handle=True) for _ in range(reps))   # the first evaluation allocates the result buffers
        tot_ms += ms; nq += len(off) - 1; per.append(round(ms, 1))
    out[label] = {"queries": nq, "postings_in": int(npost), "device_ms": round(tot_ms, 2), "queries_per_s": round(nq / (tot_ms / 1e3), 0),
                  "gpostings_per_s": round(npost / 1e9 / (tot_ms / 1e3), 2), "chunk": chunk, "chunk_ms": per[:8]}
    print(label, out[label], flush=True)
ix.query_batch([T, 0, 258], [0, 3], keep_handle=True)   # decodes the index once (auxiliary structure)
r1 = synth.query_ranks(10000, 1)
run([[(synth.term(int(k)), 255 + len(synth.term(int(k)))) for k in row] for row in r1], "single_term_10k", reps=2)
r3 = synth.query_ranks(n_and3, 3)
run([[(synth.term(int(k)), 255 + len(synth.term(int(k)))) for k in row] for row in r3], "and3")
# C5: exact phrases sampled from corpus positions (2 words for even query ids, 3 for odd)
rng = np.random.default_rng(0x5EA2C4)
phr = []
for q in range(n_phr):
    doc = int(rng.integers(0, n_docs))
    words = bytes(synth.corpus_host(1, first_doc=doc)).decode("ascii").split()
    k = 2 + (q & 1)
    if len(words) < k: continue
    s = int(rng.integers(0, len(words) - k + 1))
    phr.append([(w, -(len(w) + 4)) for w in words[s:s + k]])
run(phr, "phrase") if phr else None
print(json.dumps(out))

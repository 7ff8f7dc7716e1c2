import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import docodo_amd
from docodo_amd import synth, dist
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n)
for rep in range(3):
    b = docodo_amd.Builder(); b.AddUniformPagesDev(d.data_ptr(), n, 1024); st = b.Finish()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = dist.export_shard(b, G)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    # what rank 0 would receive at world size G if all shards looked like this one: G sources x its slice
    mo, ko, uo = (dist._split_offsets(x) for x in e.byte_splits())
    meta = torch.cat([e.meta[mo[0]:mo[1]]] * G); keys = torch.cat([e.keys[ko[0]:ko[1]]] * G); units = torch.cat([e.units[uo[0]:uo[1]]] * G)
    tc = np.full(G, e.term_count[0], np.uint64); kc = np.full(G, e.key_count[0], np.uint64); uc = np.full(G, e.unit_count[0], np.uint64)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ix = dist.merge_received(tc, kc, uc, meta, keys, units, st.max_coord)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("rep %d: build dev %.2f ms | export(G=%d) %.2f ms | merge of %d segs, %d units %.2f ms" % (
        rep, st.ms_total, G, (t1 - t0) * 1e3, int(tc.sum()), int(uc.sum()), (t3 - t2) * 1e3), flush=True)
    ix.close(); b.close()

import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, docodo_amd
from docodo_amd import synth
n = 1_000_000
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda"); synth.corpus_dev(d.data_ptr(), n)
b = docodo_amd.Builder(); b.AddUniformPagesDev(d.data_ptr(), n, 1024); b.Finish()
os.makedirs("/tmp/ixsave", exist_ok=True)
t = time.time(); b.Save("/tmp/ixsave"); print("save %.2f s" % (time.time() - t))
t = time.time(); ix = docodo_amd.Index.Load("/tmp/ixsave"); print("load %.2f s, %d terms" % (time.time() - t, ix.Count))
t = time.time(); r = ix.query_batch([docodo_amd.DCD_OP_TERM, 5, 260], [0, 3]); print("first query (decode + directories) %.2f s" % (time.time() - t))

"""Timing-only variants of k_tok_emit (library built with -DDCD_DIAG_TOK=n, scripts/build_variant.sh): what the table
gathers cost.  usage (GPU box): DOCODO_HIP_LIB=variants/libdocodo_hip_diagN.so python scripts/diag_tok.py [docs] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import docodo_amd
from docodo_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n)
for r in range(reps):
    b = docodo_amd.Builder()
    b.AddUniformPagesDev(d.data_ptr(), n, 1024)
    try:
        st = b.Finish()
        print("rep %d tok %.3f ms (k_tok_emit %.3f)" % (r, st.ms_tokenize, list(st.ms_kernel)[1]), flush=True)
    except docodo_amd.DocodoError as e:
        pass
    b.close()

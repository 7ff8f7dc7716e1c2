import sys, os
sys.path.insert(0, os.getcwd())
import torch, docodo_amd
from docodo_amd import synth
n=1000000
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda"); synth.corpus_dev(d.data_ptr(), n)
for r in range(4):
    b = docodo_amd.Builder(); b.AddUniformPagesDev(d.data_ptr(), n, 1024); st = b.Finish()
    if r==3: print("tok %.3f | count %.3f emit %.3f hist %.3f scat %.3f vocab %.3f" % (st.ms_tokenize, st.ms_kernel[0], st.ms_kernel[1], st.ms_kernel[2], st.ms_kernel[3], st.ms_kernel[7]))
    b.close()

"""Build throughput for UTF-16LE text (what a C# host pins): the same synthetic corpus widened to 16-bit units."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import docodo_amd
from docodo_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d8 = torch.empty(n * 1024, dtype=torch.uint8, device="cuda"); synth.corpus_dev(d8.data_ptr(), n)
d16 = d8.to(torch.int16)
torch.cuda.synchronize()
for rep in range(4):
    t = time.time()
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE)
    b.AddUniformPagesDev(d16.data_ptr(), n, 1024)
    st = b.Finish()
    wall = time.time() - t
    print("rep %d wall %.2f ms dev %.2f ms | tok %.2f voc %.2f sort %.2f pack %.2f img %.2f -> %.1f G chars/s, %.1f GB/s of UTF-16" % (
        rep, wall * 1e3, st.ms_total, st.ms_tokenize, st.ms_vocab, st.ms_sort, st.ms_pack, st.ms_image, n * 1024 / 1e9 / (st.ms_total / 1e3),
        2 * n * 1024 / 1e9 / (st.ms_total / 1e3)), flush=True)
    b.close()

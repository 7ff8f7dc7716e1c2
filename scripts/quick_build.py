import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import docodo_amd
from docodo_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = torch.empty(n * 1024, dtype=torch.uint8, device="cuda")
t = time.time(); synth.corpus_dev(d.data_ptr(), n); print("gen %.3fs" % (time.time() - t), flush=True)
for r in range(reps):
    t = time.time()
    b = docodo_amd.Builder(vocab_capacity_log2=int(os.environ.get("QB_CAP", "0")))
    b.AddUniformPagesDev(d.data_ptr(), n, 1024)
    st = b.Finish()
    wall = time.time() - t
    print("rep %d wall %.2f ms dev %.2f ms | tok %.2f voc %.2f sort %.2f pack %.2f img %.2f | N=%d V=%d U=%d passes=%d cap=2^%d img=%.1f MB hot=%d/%.1f%% -> %.1f GB/s" % (
        r, wall * 1e3, st.ms_total, st.ms_tokenize, st.ms_vocab, st.ms_sort, st.ms_pack, st.ms_image, st.n_postings, st.n_terms, st.n_units,
        st.sort_passes, st.vocab_capacity_log2, st.index_bytes / 1e6, st.n_hot_terms, 100.0 * st.n_hot_postings / max(1, st.n_postings), n * 1024 / 1e9 / ((st.ms_total or wall * 1e3) / 1e3)), flush=True)
    b.close()

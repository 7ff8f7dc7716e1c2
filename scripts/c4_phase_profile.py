"""Phase profile of k_pair_parts on configs[3] with a -DDCD_PROF build of the library (scripts/build_variant.sh prof "-DDCD_PROF"):
wave-residency cycles per phase (s_memtime deltas summed over all waves).
usage: DOCODO_HIP_LIB=variants/libdocodo_hip_prof.so python scripts/c4_phase_profile.py [docs] [queries]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import docodo_amd
from docodo_amd import synth, dist
n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
d = torch.empty(n_docs * 1024, dtype=torch.uint8, device="cuda")
synth.corpus_dev(d.data_ptr(), n_docs)
sb = dist.SegmentedBuilder(1024)
sb.add_pages_dev(d.data_ptr(), n_docs)
ix = sb.finish()
del d
torch.cuda.empty_cache()
T, A = docodo_amd.DCD_OP_TERM, docodo_amd.DCD_OP_AND
prog, off = [], [0]
phrases = len(sys.argv) > 3 and sys.argv[3] == "phrase"
if phrases:           # configs[4]: exact phrases sampled from corpus positions
    rng = np.random.default_rng(0x5EA2C4)
    for q in range(nq):
        doc = int(rng.integers(0, n_docs))
        words = bytes(synth.corpus_host(1, first_doc=doc)).decode("ascii").split()
        k = 2 + (q & 1)
        s0 = int(rng.integers(0, len(words) - k + 1))
        for j, w in enumerate(words[s0:s0 + k]):
            prog += [T, ix.find(w), -(len(w) + 4)]
            if j:
                prog.append(A)
        off.append(len(prog))
else:
    for row in synth.query_ranks(nq, 3):
        ws = [synth.term(int(k)) for k in row]
        ts = [ix.find(w) for w in ws]
        if any(t is None for t in ts):
            continue
        for k, (w, t) in enumerate(zip(ws, ts)):
            prog += [T, t, 255 + len(w)]
            if k:
                prog.append(A)
        off.append(len(prog))
pa, oa = np.array(prog, np.int32), np.array(off, np.uint64)
L = docodo_amd.lib()
L.dcd_debug_prof.argtypes = [C.c_void_p, C.c_int32]
z = np.zeros(24, np.uint64)
ix.query_batch(pa, oa, totals=True)
L.dcd_debug_prof(z.ctypes.data, 1)
r = ix.query_batch(pa, oa, totals=True)
L.dcd_debug_prof(z.ctypes.data, 0)
names = ["operator lookup", "gallop (everything)", "stream: split search + first fill", "half-cleaner + refill request", "bitonic sort", "merge flags",
         "refill store (waits for HBM)", "ownership", "fast path", "general path", "epilogue",
         "  gallop: short elems + directory + search", "  gallop: nearest neighbours + candidate list", "  gallop: candidates resolved in registers (dense passes)", "  gallop: walks (unresolved chains)",
         "  gallop: scan + emission"]
tot = float(z[:11].sum())
print("device %.2f ms for %d queries" % (r["ms"], oa.size - 1))
for i, n in enumerate(names):
    print("  %-36s %6.2f %%" % (n, 100.0 * float(z[i]) / tot))
print("galloping: %d short-list elements, %.2f %% with a neighbour within absR, %.2f %% resolved + emitting, %.2f %% unresolved (walk)" %
      (int(z[16]), 100.0 * float(z[17]) / max(1, int(z[16])), 100.0 * float(z[18]) / max(1, int(z[16])), 100.0 * float(z[19]) / max(1, int(z[16]))))

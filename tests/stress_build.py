"""Randomised differential test of the build path (GPU through the C ABI vs the CPU oracle): random vocabularies (size, Zipf
exponent, word lengths, letters / digits / upper case / Latin-1 letters), separators (also none at a class change), ragged
documents, word-length limits, both encodings, texts large enough for the sampling launch and the hot / cold sort as well as
tiny ones.  ".index" and ".index.list" byte for byte.  Not collected by pytest (long); run on a GPU box:
    python tests/stress_build.py [n_cases] [seed] [size_scale]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as o
import docodo_amd

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # multiplies the text sizes (8: up to 48 MB per case)
rng = np.random.default_rng(seed)
LETTERS = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ" + bytes([0xE9, 0xC9, 0xF1, 0xDF, 0xB5, 0xD7, 0xF7]), np.uint8)
DIGITS = np.frombuffer(b"0123456789" + bytes([0xB2, 0xBC]), np.uint8)
SEPS = [b" ", b" ", b" ", b", ", b". ", b"\n", b"-", b"  ", b"_", b"'"]


def make_vocab(V, max_len):
    words = []
    for _ in range(V):
        n = int(rng.integers(1, max_len + 1))
        src = DIGITS if rng.random() < 0.1 else LETTERS
        words.append(src[rng.integers(0, src.size, n)].tobytes())
    return words


def make_text(total, V, s, max_len):
    words = make_vocab(V, max_len)
    p = 1.0 / np.arange(1, V + 1) ** s
    p /= p.sum()
    parts, size = [], 0
    picks = rng.choice(V, size=max(16, total // 4), p=p)
    seps = rng.integers(0, len(SEPS), picks.size)
    glue = rng.random(picks.size) < 0.03                 # no separator: tokens still split where the class changes
    for k in range(picks.size):
        w = words[picks[k]]
        parts.append(w)
        size += len(w)
        if not glue[k]:
            parts.append(SEPS[seps[k]])
            size += len(SEPS[seps[k]])
        if size >= total:
            break
    return b"".join(parts)


bad = 0
t_start = time.time()
for case in range(n_cases):
    total = int(rng.choice([300, 20_000, 1_500_000, 3_000_000, 6_000_000])) * scale
    V = int(rng.choice([3, 40, 300, 5000, 100_000, 400_000]))
    s = float(rng.choice([0.0, 0.8, 1.1, 1.6]))
    wl = int(rng.choice([4, 9, 20, 40]))
    min_len, max_len = int(rng.choice([1, 2, 3, 5])), int(rng.choice([6, 16, 32]))
    wide = bool(rng.random() < 0.3)
    text = make_text(total, V, s, wl)
    cuts = np.sort(rng.choice(np.arange(1, max(2, len(text))), size=min(len(text) - 1, int(rng.integers(0, 400))), replace=False)) if len(text) > 2 else np.array([], int)
    off = np.concatenate([[0], cuts, [len(text)]]).astype(np.uint64)
    # oracle
    oix = o.Index(stemmers=[])
    oix.set_limits(min_len=min_len, max_len=max_len)
    ob = o.Builder(oix)
    arr = np.frombuffer(text, np.uint8)
    ob.index_docs_latin1(arr, off)
    oi, ol = ob.save()
    # device
    b = docodo_amd.Builder(encoding=docodo_amd.DCD_ENC_UTF16LE if wide else docodo_amd.DCD_ENC_LATIN1, min_word_length=min_len,
                           max_word_length=max_len)
    if wide:
        b.AddPages(text.decode("latin-1"), off, docs_per_page=True)
    else:
        b.AddPages(arr, off, docs_per_page=True)
    st = b.Finish()
    gi, gl = b.index_image(), b.list_image()
    ok = gi == oi and gl == ol and st.n_postings == ob.total_count
    if not ok:
        bad += 1
        print("MISMATCH case %d: total %d V %d s %.1f wl %d limits %d..%d wide %s docs %d: postings %d / %d, image %d / %d bytes" % (
            case, total, V, s, wl, min_len, max_len, wide, off.size - 1, st.n_postings, ob.total_count, len(gi), len(oi)), flush=True)
    b.close()
    if (case + 1) % 6 == 0:
        print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t_start), flush=True)
print("done: %d cases, %d mismatches" % (n_cases, bad))
sys.exit(1 if bad else 0)

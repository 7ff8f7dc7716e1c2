"""Worker of tests/test_gpu_build.py::test_alternative_build_paths_give_the_same_bytes: builds the synthetic corpus in a
fresh process (the library reads its environment switches once) and prints the SHA-256 of ".index" + ".index.list"."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import docodo_amd
    from docodo_amd import synth
    n_docs = int(sys.argv[1])
    text = synth.corpus_host(n_docs)
    off = np.arange(n_docs + 1, dtype=np.uint64) * synth.DOC_BYTES
    b = docodo_amd.Builder()
    b.AddPages(text, off, docs_per_page=True)
    b.Finish()
    h = hashlib.sha256(b.index_image())
    h.update(b.list_image())
    print("digest " + h.hexdigest())


if __name__ == "__main__":
    main()

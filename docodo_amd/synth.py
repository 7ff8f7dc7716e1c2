"""Synthetic Zipf corpus / query generator of SURVEY.md Appendix F (measurement support)."""
import ctypes as C

import numpy as np

from . import _lib

SEED_CORPUS = 0xD0C0D0
SEED_QUERY = 0x5EA2C4
VOCAB = 100_000
ZIPF_S = 1.1
DOC_BYTES = 1024


def corpus_host(n_docs, first_doc=0, doc_bytes=DOC_BYTES, vocab=VOCAB, s=ZIPF_S, seed=SEED_CORPUS):
    out = np.empty(n_docs * doc_bytes, np.uint8)
    _lib.check(_lib.lib().dcd_synth_corpus_host(out.ctypes.data, first_doc, n_docs, doc_bytes, vocab, s, seed))
    return out


def corpus_dev(dev_ptr, n_docs, first_doc=0, doc_bytes=DOC_BYTES, vocab=VOCAB, s=ZIPF_S, seed=SEED_CORPUS, device=0):
    _lib.check(_lib.lib().dcd_synth_corpus_dev(dev_ptr, first_doc, n_docs, doc_bytes, vocab, s, seed, device))


def term(rank, vocab=VOCAB):
    buf = C.create_string_buffer(16)
    n = C.c_size_t(0)
    _lib.check(_lib.lib().dcd_synth_term(rank, vocab, buf, 16, C.byref(n)))
    return buf.raw[:n.value].decode("ascii")


def query_ranks(n_queries, terms_per_query, distinct=True, lo_rank=101, vocab=VOCAB, s=ZIPF_S, seed=SEED_QUERY):
    out = np.empty(n_queries * terms_per_query, np.uint32)
    _lib.check(_lib.lib().dcd_synth_query_ranks(out.ctypes.data, n_queries, terms_per_query, 1 if distinct else 0,
                                                lo_rank, vocab, s, seed))
    return out.reshape(n_queries, terms_per_query)

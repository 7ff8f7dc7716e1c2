"""ctypes binding of libdocodo_hip.so (C ABI declared in include/docodo_hip.h).

There is no CPU fallback: importing this module without the built library raises, and every
compute entry point fails with DCD_E_HIP when no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DOCODO_HIP_LIB") or os.path.join(_HERE, "libdocodo_hip.so")   # override: A/B builds of the library

DCD_OK = 0
DCD_E_BADARG, DCD_E_OVERFLOW, DCD_E_IO, DCD_E_HIP, DCD_E_RCCL = -1, -2, -3, -4, -5
DCD_E_UNSUPPORTED, DCD_E_NOTFOUND, DCD_E_NOMEM, DCD_E_STATE = -6, -7, -8, -9
DCD_ENC_LATIN1, DCD_ENC_UTF16LE = 1, 2
DCD_OP_TERM, DCD_OP_AND, DCD_OP_OR, DCD_OP_SETR = 0, 1, 2, 3
DCD_COMM_RCCL, DCD_COMM_HOST, DCD_COMM_ID_BYTES = 0, 1, 128


class DocodoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdocodo_hip error %d: %s" % (code, msg))
        self.code = code


DCD_BUILD_SHARD_ONLY = 1


class BuildOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("min_word_length", C.c_int32), ("max_word_length", C.c_int32),
                ("encoding", C.c_int32), ("coord_base", C.c_uint64), ("vocab_capacity_log2", C.c_uint32),
                ("flags", C.c_uint32)]


class BuildStats(C.Structure):
    _fields_ = [("n_chars", C.c_uint64), ("n_pages", C.c_uint64), ("n_docs", C.c_uint64), ("n_postings", C.c_uint64),
                ("n_terms", C.c_uint64), ("n_units", C.c_uint64), ("max_coord", C.c_uint64), ("index_bytes", C.c_uint64),
                ("list_bytes", C.c_uint64), ("ms_tokenize", C.c_float), ("ms_vocab", C.c_float), ("ms_sort", C.c_float),
                ("ms_pack", C.c_float), ("ms_image", C.c_float), ("ms_total", C.c_float), ("sort_passes", C.c_uint32),
                ("vocab_capacity_log2", C.c_uint32), ("ms_kernel", C.c_float * 8), ("n_launch", C.c_uint32 * 8),
                ("id_keys", C.c_uint32), ("n_hot_terms", C.c_uint32), ("n_hot_postings", C.c_uint64)]


KERNEL_NAMES = ["k_tok_count", "k_tok_emit", "k_rs_hist", "k_rs_scatter", "k_enc_reduce", "k_enc_write", "k_img_units",
                "vocab_sort"]


class TermMeta(C.Structure):
    """dcd_term_meta: one record of a shard export (include/docodo_hip.h); dist.py views the device buffer as 4 x u64."""
    _fields_ = [("first", C.c_uint64), ("last", C.c_uint64), ("n_values", C.c_uint64), ("n_units", C.c_uint32),
                ("key_len", C.c_uint32)]


class QueryOpts(C.Structure):
    _fields_ = [("max_found_pages", C.c_int32), ("max_found_docs", C.c_int32), ("doc_rank_multiply", C.c_float),
                ("want_sequences", C.c_int32), ("doc_match", C.c_void_p), ("n_doc_match", C.c_uint64),
                ("no_download", C.c_int32), ("reserved", C.c_int32)]


class ResultsView(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("status", C.POINTER(C.c_int32)), ("page_off", C.POINTER(C.c_uint64)),
                ("page_index", C.POINTER(C.c_uint32)), ("page_rank", C.POINTER(C.c_float)),
                ("pos_off", C.POINTER(C.c_uint64)), ("pos", C.POINTER(C.c_int32)), ("doc_off", C.POINTER(C.c_uint64)),
                ("doc_index", C.POINTER(C.c_uint32)), ("doc_rank", C.POINTER(C.c_float)),
                ("doc_n_pages", C.POINTER(C.c_uint32)), ("seq_off", C.POINTER(C.c_uint64)),
                ("seq_units", C.POINTER(C.c_uint16)), ("seq_R", C.POINTER(C.c_int32)), ("ms_device", C.c_float),
                ("n_pages_total", C.c_uint64), ("n_pos_total", C.c_uint64), ("n_docs_total", C.c_uint64),
                ("ops_named", C.c_uint64), ("ops_evaluated", C.c_uint64), ("units_named", C.c_uint64), ("units_evaluated", C.c_uint64),
                ("sub_batches", C.c_uint64)]


# name -> (restype, argtypes); mirrors include/docodo_hip.h one to one
_vp, _sz, _i32, _u64, _u32 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint64, C.c_uint32
_pz, _pi32, _pu64, _pu32 = C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
SIGNATURES = {
    "dcd_version": (_i32, []),
    "dcd_struct_size": (_u64, [_i32]),
    "dcd_last_error": (_i32, [C.c_char_p, _sz]),
    "dcd_device_count": (_i32, [_pi32]),
    "dcd_pool_trim": (_i32, [_i32]),
    "dcd_set_build_events": (_i32, [_i32, C.c_uint32]),
    "dcd_pool_cached_bytes": (_i32, [_i32, _pu64]),
    "dcd_seq_encode": (_i32, [_vp, _sz, _vp, _sz, _pz]),
    "dcd_seq_decode": (_i32, [_vp, _sz, _vp, _sz, _pz]),
    "dcd_seq_count": (_i32, [_vp, _sz, _pz]),
    "dcd_seq_shift": (_i32, [_vp, _sz, _u64, _vp, _sz, _pz]),
    "dcd_seq_and": (_i32, [_vp, _sz, _i32, _vp, _sz, _i32, _vp, _sz, _pz, _pi32]),
    "dcd_seq_or": (_i32, [_vp, _sz, _i32, _vp, _sz, _i32, _vp, _sz, _pz, _pi32]),
    "dcd_build_default_opts": (None, [C.POINTER(BuildOpts)]),
    "dcd_build_create": (_i32, [C.POINTER(BuildOpts), C.POINTER(_vp)]),
    "dcd_build_free": (None, [_vp]),
    "dcd_build_add_doc": (_i32, [_vp, _vp, _sz, _vp, _sz]),
    "dcd_build_add_pages": (_i32, [_vp, _vp, _u64, _vp, _sz, _vp, _vp, _i32, _u64, _vp, _sz]),
    "dcd_build_add_pages_dev": (_i32, [_vp, _vp, _u64, _vp, _sz, _i32, _u64, _vp, _sz]),
    "dcd_build_add_uniform_pages_dev": (_i32, [_vp, _vp, _u64, _u32, _u64, _vp, _sz]),
    "dcd_build_add": (_i32, [_vp, _vp, _sz, _u64]),
    "dcd_build_end_page": (_i32, [_vp, _vp, _sz]),
    "dcd_build_add_word": (_i32, [_vp, _vp, _sz, _u64]),
    "dcd_build_finish": (_i32, [_vp]),
    "dcd_build_words": (_i32, [_vp, _vp, _vp]),
    "dcd_build_words_get": (_i32, [_vp, _vp, _sz, _vp, _sz]),
    "dcd_build_set_codes": (_i32, [_vp, _vp, _sz, _vp, _vp, _sz]),
    "dcd_build_get_stats": (_i32, [_vp, C.POINTER(BuildStats)]),
    "dcd_build_index_image": (_i32, [_vp, _vp, _sz, _pz]),
    "dcd_build_list_image": (_i32, [_vp, _vp, _sz, _pz]),
    "dcd_build_save": (_i32, [_vp, C.c_char_p]),
    "dcd_index_from_build": (_i32, [_vp, C.POINTER(_vp)]),
    "dcd_index_open": (_i32, [C.c_char_p, _i32, C.POINTER(_vp)]),
    "dcd_index_from_images": (_i32, [_vp, _sz, _vp, _sz, _i32, C.POINTER(_vp)]),
    "dcd_index_free": (None, [_vp]),
    "dcd_index_trim": (_i32, [_vp]),
    "dcd_index_prepare": (_i32, [_vp, C.POINTER(C.c_float)]),
    "dcd_debug_sk_network": (_i32, [_vp]),
    "dcd_index_count": (_i32, [_vp, _pu64]),
    "dcd_index_max_coord": (_i32, [_vp, _pu64]),
    "dcd_index_n_units": (_i32, [_vp, _pu64]),
    "dcd_merge_index_files": (_i32, [_vp, _sz, C.c_char_p]),
    "dcd_index_open_lazy": (_i32, [C.c_char_p, _i32, C.c_uint64, C.POINTER(_vp)]),
    "dcd_index_lazy_stats": (_i32, [_vp, _pu64, _pu64, _pu64]),
    "dcd_index_n_pages": (_i32, [_vp, _pu64]),
    "dcd_index_n_docs": (_i32, [_vp, _pu64]),
    "dcd_index_find": (_i32, [_vp, _vp, _sz, _pu32]),
    "dcd_index_key": (_i32, [_vp, _u32, _vp, _sz, _pz]),
    "dcd_index_get": (_i32, [_vp, _u32, _vp, _sz, _pz]),
    "dcd_index_term_count": (_i32, [_vp, _u32, _pu64]),
    "dcd_index_page_info": (_i32, [_vp, _u64, _pu64, _pu32, _vp, _sz, _pz]),
    "dcd_index_doc_name": (_i32, [_vp, _u32, _vp, _sz, _pz]),
    "dcd_query_default_opts": (None, [C.POINTER(QueryOpts)]),
    "dcd_query_batch": (_i32, [_vp, _vp, _vp, _sz, C.POINTER(QueryOpts), C.POINTER(_vp)]),
    "dcd_results_get": (_i32, [_vp, C.POINTER(ResultsView)]),
    "dcd_results_free": (None, [_vp]),
    "dcd_build_export_plan": (_i32, [_vp, _u32, _vp, _vp, _vp]),
    "dcd_build_export_fill": (_i32, [_vp, _vp, _vp, _vp]),
    "dcd_build_export_heavy": (_i32, [_vp, C.c_uint64, _vp, _vp, C.c_size_t, _vp]),
    "dcd_build_export_plan_balanced": (_i32, [_vp, _u32, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp]),
    "dcd_merge_shards": (_i32, [_i32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u64, C.POINTER(_vp)]),
    "dcd_merge_begin": (_i32, [_i32, _u32, _vp, _vp, _vp, _vp, _vp, _u64, C.POINTER(_vp)]),
    "dcd_merge_finish": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "dcd_merge_free": (None, [_vp]),
    "dcd_index_close_words": (_i32, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    "dcd_index_like_words": (_i32, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    "dcd_index_suggest": (_i32, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    "dcd_index_export_plan": (_i32, [_vp, C.c_uint32, _vp, _vp, _vp]),
    "dcd_index_export_fill": (_i32, [_vp, _vp, _vp, _vp]),
    "dcd_index_export_heavy": (_i32, [_vp, C.c_uint64, _vp, _vp, C.c_size_t, _vp]),
    "dcd_index_export_plan_balanced": (_i32, [_vp, C.c_uint32, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp]),
    "dcd_index_image": (_i32, [_vp, _vp, _sz, _vp]),
    "dcd_index_list_image": (_i32, [_vp, _vp, _sz, _vp]),
    "dcd_index_set_list_image": (_i32, [_vp, _vp, _sz]),
    "dcd_index_save": (_i32, [_vp, C.c_char_p]),
    "dcd_index_set_uniform_pages": (_i32, [_vp, _u64, _u32, _u64, _vp, _sz]),
    "dcd_comm_unique_id": (_i32, [_i32, _vp]),
    "dcd_comm_create": (_i32, [_i32, _i32, _vp, _i32, _i32, C.POINTER(_vp)]),
    "dcd_comm_create_all": (_i32, [_i32, _vp, _i32, _vp]),
    "dcd_comm_free": (None, [_vp]),
    "dcd_comm_rank": (_i32, [_vp, _pi32, _pi32]),
    "dcd_comm_allgather": (_i32, [_vp, _vp, _sz, _vp]),
    "dcd_build_merge_distributed": (_i32, [_vp, _vp, _i32, C.POINTER(_vp)]),
    "dcd_index_merge_distributed": (_i32, [_vp, _vp, _i32, C.POINTER(_vp)]),
    "dcd_index_allgather": (_i32, [_vp, _vp, C.POINTER(_vp)]),
    "dcd_index_gather": (_i32, [_vp, _vp, _i32, C.POINTER(_vp)]),
    "dcd_balance_owners": (_i32, [_u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _pz, _vp]),
    "dcd_synth_corpus_host": (_i32, [_vp, _u64, _u64, _u32, _u32, C.c_double, _u64]),
    "dcd_synth_corpus_dev": (_i32, [_vp, _u64, _u64, _u32, _u32, C.c_double, _u64, _i32]),
    "dcd_synth_term": (_i32, [_u32, _u32, C.c_char_p, _sz, _pz]),
    "dcd_synth_query_ranks": (_i32, [_vp, _u64, _u32, _i32, _u32, _u32, C.c_double, _u64]),
}

_LIB = None


def lib():
    """Loads libdocodo_hip.so; raises loudly when it was not built (no fallback path exists)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libdocodo_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "or `make -C docodo_amd/csrc`. docodo_amd has no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.
        # If torch is going to be used in this process (bench.py, tests: device buffers, RCCL), it must
        # be loaded first so that our NEEDED libamdhip64.so.7 binds to the same copy; two runtimes in
        # one process leave the second without a GPU.  A host without torch (the C# shim) simply uses
        # the system ROCm the library was linked against.
        if os.environ.get("DOCODO_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError here == header/library mismatch
            f.restype = res
            f.argtypes = args
        _LIB = L
    return _LIB


def last_error():
    buf = C.create_string_buffer(2048)
    lib().dcd_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def set_build_events(level=2, kernel_mask=0):
    """dcd_set_build_events: which HIP events builds record for their stats (2 = everything, 1 = stage timers, 0 = none;
    kernel_mask: DCD_K_* bits measured whatever the level)."""
    check(lib().dcd_set_build_events(level, kernel_mask))


def pool_trim(device=0):
    """Returns the library's cached device blocks to HIP (dcd_pool_trim): call it when another allocator in the process
    (torch) needs the memory, or after the last handle of a large build / index was closed."""
    check(lib().dcd_pool_trim(device))


def pool_cached_bytes(device=0):
    n = C.c_uint64(0)
    check(lib().dcd_pool_cached_bytes(device, C.byref(n)))
    return int(n.value)


def check(code):
    if code != DCD_OK:
        raise DocodoError(code, last_error())
    return code

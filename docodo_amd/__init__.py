"""docodo_amd -- MI355X-native inverted-index build + query evaluation behind Docodo's
Index / IndexSequence / Search surface.  Python mirror of the host-side interface (stand-in for
the C# shim of INTEGRATION.md); all compute goes through the C ABI of libdocodo_hip.so."""
from ._lib import (DCD_E_NOMEM, DCD_ENC_LATIN1, DCD_ENC_UTF16LE, DCD_OP_AND, DCD_OP_OR, DCD_OP_SETR, DCD_OP_TERM, DocodoError, lib, pool_cached_bytes, pool_trim, set_build_events)  # noqa: F401
from .sequence import IndexSequence  # noqa: F401
from .index import Builder, Index, SearchOptions, QueryBatchResult  # noqa: F401
from . import synth  # noqa: F401
from .search import MicroBatcher, Search, SearchResult  # noqa: F401
from .wordcodes import WordCoder  # noqa: F401

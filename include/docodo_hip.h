/*
 * docodo_hip.h -- C ABI of libdocodo_hip.so: the MI355X (gfx950) inverted-index build and
 * query-evaluation path that drops in behind Docodo's Index / IndexSequence / Search surface.
 *
 * The reference (alexarchen/Docodo, 100 % C#) has no FFI boundary; this header IS the boundary a
 * maintainer would P/Invoke (INTEGRATION.md shows the [DllImport] stubs).  Each entry point cites
 * the reference code it replaces (paths relative to Docodo.NET/ in the reference tree).
 *
 * Conventions
 *   - every function returns int32 status: DCD_OK or a negative DCD_E_* class; no C++ exception
 *     or HIP error crosses the ABI; dcd_last_error() returns the calling thread's last message;
 *   - inputs are caller-owned, only read during the call (C# pins string / ushort[] with `fixed`);
 *   - outputs go to caller-provided buffers (cap in elements; the needed size is always written,
 *     DCD_E_OVERFLOW if cap is too small -> call again) or to library-owned objects released by
 *     the matching *_free; handles are opaque, independent, and usable from any thread
 *     (one call at a time per handle);
 *   - strings are UTF-16LE code units + length (what a C# `string` pins to) unless named *_utf8;
 *   - "dev" pointers are HIP device pointers on the handle's device (hipMalloc / torch .data_ptr()).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with
 *     DCD_E_HIP.
 */
#ifndef DOCODO_HIP_H
#define DOCODO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCD_OK 0
#define DCD_E_BADARG (-1)
#define DCD_E_OVERFLOW (-2)     /* output buffer too small / value does not fit */
#define DCD_E_IO (-3)
#define DCD_E_HIP (-4)          /* HIP runtime error or no device */
#define DCD_E_RCCL (-5)
#define DCD_E_UNSUPPORTED (-6)  /* e.g. unsupported text encoding */
#define DCD_E_NOTFOUND (-7)
#define DCD_E_NOMEM (-8)
#define DCD_E_STATE (-9)        /* call order violated (e.g. query before finish) */

#define DCD_ENC_LATIN1 1        /* 1 byte = 1 UTF-16 code unit (ASCII / ISO-8859-1) */
#define DCD_ENC_UTF16LE 2       /* raw C# string memory */

typedef struct dcd_build dcd_build;
typedef struct dcd_index dcd_index;
typedef struct dcd_results dcd_results;

/* ABI version.  2 (round 4): dcd_results_view ends with ops_named .. units_evaluated; dcd_index_trim, dcd_struct_size.
 * A host compiled against version 1 must be rebuilt (dcd_results_get writes sizeof(dcd_results_view) bytes). */
#define DCD_ABI_VERSION 2
int32_t dcd_version(void);
/* sizeof of the ABI's structs as this library was built, for hosts that declare them by hand (P/Invoke, ctypes):
 * which = 0 dcd_build_opts, 1 dcd_build_stats, 2 dcd_query_opts, 3 dcd_results_view, 4 dcd_term_meta; 0 for anything else */
uint64_t dcd_struct_size(int32_t which);
/* copies the calling thread's last error message (UTF-8, NUL-terminated); returns its length */
int32_t dcd_last_error(char* buf, size_t cap);
int32_t dcd_device_count(int32_t* n);
/* The library keeps released device blocks in a per-device cache (stream-ordered: a block is reused by another
 * stream only after the work queued on it at release time has completed).  dcd_pool_trim returns the cached blocks
 * to HIP -- call it when another allocator in the process (the host's own, torch, ...) needs the memory, e.g. from
 * Index.Dispose(); the cache never holds more than DOCODO_POOL_MAX_CACHED bytes (default: half of the device memory).
 * No reference counterpart: the C# reference leaves this to the garbage collector. */
int32_t dcd_pool_trim(int32_t device);
/* How many HIP events the builds of this process record for dcd_build_stats (every record is a marker packet the build's
 * stream stops at: ~30 of them cost 0.12 ms of a 4 ms build).  level 2 (default; DOCODO_BUILD_EVENTS): the stage timers
 * ms_tokenize .. ms_total and ms_kernel[] / n_launch[] of every kernel; 1: the stage timers only; 0: neither (the fields are
 * 0).  kernel_mask: bit DCD_K_* k set = ms_kernel[k] is measured whatever the level.  A build takes the setting that is in
 * force when dcd_build_finish starts. */
int32_t dcd_set_build_events(int32_t level, uint32_t kernel_mask);
int32_t dcd_pool_cached_bytes(int32_t device, uint64_t* bytes);

/* ------------------------------------------------------------------------------------------
 * IndexSequence codec and algebra on host buffers (IndexSequence.cs).  These are the per-object
 * entry points a C# IndexSequence shim calls; batched work should use the build / query handles.
 * ------------------------------------------------------------------------------------------ */
/* IndexSequence.Builder.Add / AddRange, IndexSequence.cs:63-90: ascending u64 -> 15-bit var-ushort deltas */
int32_t dcd_seq_encode(const uint64_t* values, size_t n, uint16_t* out, size_t cap, size_t* n_units);
/* SeqEnumerator.MoveNext over the whole sequence, IndexSequence.cs:114-136 */
int32_t dcd_seq_decode(const uint16_t* units, size_t n_units, uint64_t* out, size_t cap, size_t* n);
/* LINQ Count() of the enumerator (Search.cs:307): number of VALUES, not units */
int32_t dcd_seq_count(const uint16_t* units, size_t n_units, size_t* n_values);
/* IndexSequence.Shift, IndexSequence.cs:191-202 */
int32_t dcd_seq_shift(const uint16_t* units, size_t n_units, uint64_t shift, uint16_t* out, size_t cap, size_t* n_out);
/* operator * / operator &, IndexSequence.cs:205-284 (proximity / ordered AND) */
int32_t dcd_seq_and(const uint16_t* a, size_t na, int32_t Ra, const uint16_t* b, size_t nb, int32_t Rb,
                    uint16_t* out, size_t cap, size_t* n_out, int32_t* R_out);
/* operator +, IndexSequence.cs:286-322 (OR) */
int32_t dcd_seq_or(const uint16_t* a, size_t na, int32_t Ra, const uint16_t* b, size_t nb, int32_t Rb,
                   uint16_t* out, size_t cap, size_t* n_out, int32_t* R_out);

/* ------------------------------------------------------------------------------------------
 * Build: replaces the IndexTask page loop (Build.cs:465-553), Index.Builder (Build.cs:258-437)
 * and the merge functions (Index.cs:537-752): everything between "pages arrive" and
 * ".index / .index.list exist".
 * ------------------------------------------------------------------------------------------ */
typedef struct dcd_build_opts {
  int32_t device;           /* HIP device ordinal */
  int32_t min_word_length;  /* Index.MIN_WORD_LENGTH, Index.cs:113 (default 3) */
  int32_t max_word_length;  /* Index.MAX_WORD_LENGTH, Index.cs:97  (default 32, max 32) */
  int32_t encoding;         /* DCD_ENC_* of every text buffer handed to this handle */
  uint64_t coord_base;      /* global coordinate of the first character (doc-range shards: text
                               length of all previous shards; 0 for a single builder) */
  uint32_t vocab_capacity_log2; /* 0 = auto; hash-table slots = 2^n (grown automatically on overflow) */
  uint32_t flags;           /* DCD_BUILD_* (was: reserved, 0) */
} dcd_build_opts;
/* The build is one doc-range shard of a distributed build: it will be exported (dcd_build_export_plan / _fill) and merged
 * elsewhere, never serialised or queried on its own.  dcd_build_finish then stops in front of the ".index" image and the
 * term-ordered posting array (a third of a gigabyte read and two written at 1 M documents) and the export reads the posting
 * units where the pack kernel left them; dcd_build_index_image, dcd_build_save and dcd_index_from_build fail with
 * DCD_E_STATE on such a handle. */
#define DCD_BUILD_SHARD_ONLY 1u

void dcd_build_default_opts(dcd_build_opts* o);
int32_t dcd_build_create(const dcd_build_opts* opts, dcd_build** out);
void dcd_build_free(dcd_build* h);

/* Builder.AddDoc(sourceid, name), Build.cs:348-356: opens a document; following pages belong to it */
int32_t dcd_build_add_doc(dcd_build* h, const uint16_t* source, size_t source_len, const uint16_t* name, size_t name_len);
/* The per-page body of IndexTask, Build.cs:477-541, for n_pages BODY pages at once: text holds the
 * pages back to back (page p = [page_off[p], page_off[p+1]) in characters, page_off[0] = 0), the
 * page ids are given as one UTF-16 string with ids_off[n_pages+1] offsets (ids_off == NULL: every
 * page has id "1").  If docs_per_page != 0 each page opens its own document named by its decimal
 * index first_doc + p under `source` (the synthetic corpora of SURVEY.md Appendix F) and no prior
 * dcd_build_add_doc is needed.  Pages with id "0" (header pages, Build.cs:485-524) are tokenised
 * on the host and enter the device pipeline as explicit postings.  Text is copied to the device
 * by this call (asynchronously staged); the kernels run in dcd_build_finish. */
int32_t dcd_build_add_pages(dcd_build* h, const void* text, uint64_t n_chars, const uint64_t* page_off,
                            size_t n_pages, const uint16_t* page_ids, const uint64_t* ids_off,
                            int32_t docs_per_page, uint64_t first_doc, const uint16_t* source, size_t source_len);
/* Same, for text that already lives in device memory (bench / pipelines that tokenise where the
 * text was produced).  The buffer must stay valid until dcd_build_finish returns. */
int32_t dcd_build_add_pages_dev(dcd_build* h, const void* text_dev, uint64_t n_chars, const uint64_t* page_off_host,
                                size_t n_pages, int32_t docs_per_page, uint64_t first_doc,
                                const uint16_t* source, size_t source_len);
/* Device-resident text made of n_pages pages of exactly page_chars characters each, one document per
 * page (the synthetic corpora of SURVEY.md Appendix F).  No per-page host metadata is touched: page
 * boundaries are computed arithmetically on the device and ".index.list" is generated on demand. */
int32_t dcd_build_add_uniform_pages_dev(dcd_build* h, const void* text_dev, uint64_t n_pages, uint32_t page_chars,
                                        uint64_t first_doc, const uint16_t* source, size_t source_len);
/* Builder.Add(code, coord), Build.cs:324-341: explicit posting for an index code */
int32_t dcd_build_add(dcd_build* h, const uint16_t* code, size_t len, uint64_t coord);
/* Builder.AddWord(word, coord), Build.cs:302-316: explicit posting for a WORD; it receives the word's index codes
 * (identity unless the host supplies codes through dcd_build_words / dcd_build_set_codes) */
int32_t dcd_build_add_word(dcd_build* h, const uint16_t* word, size_t len, uint64_t coord);
/* Builder.EndPage(id) / EndPage(id, maxcoord), Build.cs:359-367 (explicit-posting API only) */
int32_t dcd_build_end_page(dcd_build* h, const uint16_t* id, size_t len);
/* Runs the device pipeline: tokenise -> vocabulary -> stable radix sort by term -> var-ushort pack
 * -> ".index" byte image, all resident in HBM.  Equivalent of Builder.Save + MergeAll/MergeIndexes
 * for one coordinate space (Build.cs:370-404, Index.cs:537-752). */
int32_t dcd_build_finish(dcd_build* h);

/* Word codes -- GetWordCodes, Build.cs:175-247 (stop words, "#HEX" vocabulary groups, "$stem" codes).  The
 * stemmers and vocabularies stay on the host (StemmersNet / Dict.cs in the reference); the device does the
 * per-posting work.  Optional two-step protocol between the last page and dcd_build_finish:
 *   dcd_build_words      runs tokenise + vocabulary on the device and returns the number of DISTINCT words of
 *                        the build (text tokens, header-page value words, dcd_build_add_word) and their total
 *                        length; dcd_build_words_get copies them out in ordinal order (UTF-16, off[n_words+1]);
 *   dcd_build_set_codes  gives, for word w, its codes = entries code_off[w] .. code_off[w+1], entry e being the
 *                        string chars[str_off[e] .. str_off[e+1]).  No entries = the word is dropped (stop word).
 *                        (GetWordCodes' `suff` member is always null in the reference, Build.cs:203/233, so the
 *                        bKeepForms branch of AddWord, Build.cs:310-313, never adds anything.)
 * dcd_build_finish then emits one posting per (token, code), Builder.AddWord, at the token's coordinate;
 * postings added with dcd_build_add (explicit CODES, e.g. the "&field" markers of header pages) are unaffected.
 * Without dcd_build_set_codes every word keeps the identity code list [word]. */
int32_t dcd_build_words(dcd_build* h, size_t* n_words, size_t* n_chars);
int32_t dcd_build_words_get(dcd_build* h, uint16_t* chars, size_t chars_cap, uint32_t* off, size_t off_cap);
int32_t dcd_build_set_codes(dcd_build* h, const uint32_t* code_off, size_t n_words, const uint32_t* str_off,
                            const uint16_t* chars, size_t n_entries);

typedef struct dcd_build_stats {
  uint64_t n_chars, n_pages, n_docs;
  uint64_t n_postings;      /* Builder.TotalCount */
  uint64_t n_terms;         /* distinct index codes */
  uint64_t n_units;         /* total ushort units in all posting lists */
  uint64_t max_coord;       /* Builder.maxCoord = coordinate of the last posting added, Build.cs:326 */
  uint64_t index_bytes;     /* size of the .index image */
  uint64_t list_bytes;      /* size of the .index.list image */
  float ms_tokenize, ms_vocab, ms_sort, ms_pack, ms_image, ms_total; /* device time per stage (HIP events) */
  uint32_t sort_passes, vocab_capacity_log2;
  /* device time of individual kernels (HIP events on the build stream), summed over their launches:
   * [0] k_tok_count  [1] k_tok_emit  [2] k_rs_hist (tuple sort)  [3] k_rs_scatter (tuple sort)
   * [4] k_enc_reduce  [5] k_enc_write  [6] k_img_units  [7] vocabulary string sort (all kernels) */
  float ms_kernel[8];
  uint32_t n_launch[8];
  /* how the tuples were sorted: id_keys = 2: by the ids the vocabulary table hands out while the tokeniser fills it (the tuples
   * carry their sort keys from the start), 1: by dense ids taken from a scan of the table afterwards; either way the ordinal
   * vocabulary sort ran beside the tuple sort ([7] is then overlapped time) and k_img_units also copies the units into term
   * order; 0: by ordinal rank.  n_hot_terms > 0 (id_keys = 1: chosen from sample counts of the filled table; id_keys = 2: by a sampling launch in front of the tokeniser, whose most frequent terms get the ids 0 .. n_hot_terms - 1): the first radix
   * pass gave that many frequent terms a bucket of their own (n_hot_postings tuples, final after one pass) and only the
   * other n_postings - n_hot_postings tuples took the remaining sort_passes - 1 passes */
  uint32_t id_keys, n_hot_terms;
  uint64_t n_hot_postings;
} dcd_build_stats;
#define DCD_K_TOK_COUNT 0
#define DCD_K_TOK_EMIT 1
#define DCD_K_RS_HIST 2
#define DCD_K_RS_SCATTER 3
#define DCD_K_ENC_REDUCE 4
#define DCD_K_ENC_WRITE 5
#define DCD_K_IMG_UNITS 6
#define DCD_K_VOCAB_SORT 7
int32_t dcd_build_get_stats(dcd_build* h, dcd_build_stats* out);

/* Exact bytes of ".index" (Build.cs:375-386 / Appendix A.3) and ".index.list" (Build.cs:99-118) */
int32_t dcd_build_index_image(dcd_build* h, uint8_t* out, size_t cap, size_t* n);
int32_t dcd_build_list_image(dcd_build* h, uint8_t* out, size_t cap, size_t* n);
/* Writes <dir>/.index and <dir>/.index.list (what Builder.Build() leaves behind, Build.cs:407-434) */
int32_t dcd_build_save(dcd_build* h, const char* dir_utf8);

/* ------------------------------------------------------------------------------------------
 * Index + query: replaces Index.Load (Index.cs:312-380), this[key] / LoadSequence
 * (Index.cs:198-216, 255-263), the IndexSequence operators as used by Search, Count()
 * (Search.cs:307), PrepareSearchResult (Search.cs:365-420) and the ranks (Search.cs:99-111,
 * 552-557, 599).
 * ------------------------------------------------------------------------------------------ */
/* adopt the HBM-resident index of a finished build.  (The build itself leaves the ".index" image and the posting units in its
 * sort order; the term-ordered posting array a queryable index needs is made here, once, on the device.) */
int32_t dcd_index_from_build(dcd_build* b, dcd_index** out);
int32_t dcd_index_open(const char* dir_utf8, int32_t device, dcd_index** out);   /* Index.Load */
int32_t dcd_index_from_images(const uint8_t* index_img, size_t n_index, const uint8_t* list_img, size_t n_list,
                              int32_t device, dcd_index** out);
/* Index.Load with InMemory == false (Index.cs:133-136, 346-348): the term dictionary and the page table are loaded (host +
 * device), the posting lists STAY in "<dir>/.index" -- one stub per key {n_units, file position}.  dcd_index_get reads a list
 * when it is asked for (LoadSequence, Index.cs:198-216); dcd_query_batch reads the lists its programs name, keeps them as a
 * small resident index on the device (at most ~resident_bytes of device memory, 0 = 1 GiB; the union with the previous
 * batches' lists while it fits) and evaluates there -- same results as the in-memory handle.  Dictionary calls (find, key,
 * close_words, suggest), the page calls and dcd_index_term_count work as on any handle; export / image / save need the
 * postings in memory and fail with DCD_E_STATE.  dcd_index_lazy_stats: lists and bytes read from the file so far, terms
 * resident now. */
int32_t dcd_index_open_lazy(const char* dir_utf8, int32_t device, uint64_t resident_bytes, dcd_index** out);
int32_t dcd_index_lazy_stats(dcd_index* h, uint64_t* lists_read, uint64_t* bytes_read, uint64_t* resident_terms);
void dcd_index_free(dcd_index* h);
/* Releases what the handle keeps between query batches -- the work arena of dcd_query_batch (at most an eighth of the device
 * memory) and, for an on-disk index, the resident posting lists -- and returns the library's cached device blocks to the device.
 * The index stays usable; the next batch allocates again.  For processes that keep several indexes open on one device. */
int32_t dcd_index_trim(dcd_index* h);
/* The query side keeps every posting list decoded (8-byte absolute coordinates) plus a cell directory per long list: auxiliary
 * structures made from the units in two passes, by the first dcd_query_batch of a handle or -- explicitly -- here (a serving host
 * calls it right after Load / Build).  *ms_device (may be NULL): device time of that work, 0 if it was done before. */
int32_t dcd_index_prepare(dcd_index* h, float* ms_device);
/* tests: the 256-key merge network of the streaming operator kernel (k_pair_stream) on a bitonic sequence in host memory, in place */
int32_t dcd_debug_sk_network(uint32_t* keys256);
int32_t dcd_index_count(dcd_index* h, uint64_t* n_terms);              /* Index.Count */
int32_t dcd_index_max_coord(dcd_index* h, uint64_t* v);                /* Index.MaxCoord */
int32_t dcd_index_n_units(dcd_index* h, uint64_t* n);                  /* ushort units of all posting lists together */
int32_t dcd_index_n_pages(dcd_index* h, uint64_t* n);
int32_t dcd_index_n_docs(dcd_index* h, uint64_t* n);
/* ordinal position of a key in the sorted dictionary (SortedList keys, Index.cs:151); DCD_E_NOTFOUND */
int32_t dcd_index_find(dcd_index* h, const uint16_t* key, size_t len, uint32_t* term);
int32_t dcd_index_key(dcd_index* h, uint32_t term, uint16_t* out, size_t cap, size_t* len);
/* this[key]: the posting list's units (Index.cs:255-263) */
int32_t dcd_index_get(dcd_index* h, uint32_t term, uint16_t* out, size_t cap, size_t* n_units);
int32_t dcd_index_term_count(dcd_index* h, uint32_t term, uint64_t* n_values);   /* Count() */
/* GetCloseWords, Search.cs:168-174 (Levenshtein, Index.cs:46-95): the n keys closest to `word`, ties in dictionary
 * order -- a dictionary scan on the device (one thread per key, stable radix sort by distance) */
int32_t dcd_index_close_words(dcd_index* h, const uint16_t* word, size_t len, uint32_t n, uint32_t* out_terms, uint32_t* n_out);
/* GetLikeWords, Search.cs:158-167 (wildcard expansion of SearchWord): the first n keys, in dictionary order, whose first unit is a
 * letter and that match `pattern` as an unanchored regex in which '_' stands for ".*" -- a dictionary scan on the device */
int32_t dcd_index_like_words(dcd_index* h, const uint16_t* pattern, size_t len, uint32_t n, uint32_t* out_terms, uint32_t* n_out);
/* term selection of GetSuggessions, Search.cs:176-188: keys starting with `prefix`, longer than it, first unit >= 'A',
 * by descending IndexSequence.Count (ushort units), the first n */
int32_t dcd_index_suggest(dcd_index* h, const uint16_t* prefix, size_t len, uint32_t n, uint32_t* out_terms, uint32_t* n_out);
int32_t dcd_index_page_info(dcd_index* h, uint64_t page, uint64_t* end_coord, uint32_t* doc,
                            uint16_t* id_out, size_t id_cap, size_t* id_len);
int32_t dcd_index_doc_name(dcd_index* h, uint32_t doc, uint16_t* out, size_t cap, size_t* len);

/* A query is a postfix program of int32 words over ordinal term ids:
 *   DCD_OP_TERM, term, R   push this[term] with IndexSequence.R = R (Search.cs:302-312 computes R)
 *   DCD_OP_AND             pop b, pop a, push a * b   (IndexSequence.cs:218-284)
 *   DCD_OP_OR              pop b, pop a, push a + b   (IndexSequence.cs:286-322)
 * prog_off[q] .. prog_off[q+1] delimit query q.  The batch is evaluated on the device; per query
 * the final sequence goes through PrepareSearchResult + ranks with doc filter "accept all". */
#define DCD_OP_TERM 0
#define DCD_OP_AND 1
#define DCD_OP_OR 2
#define DCD_OP_SETR 3   /* DCD_OP_SETR, R: overwrite IndexSequence.R of the top of the stack (what
                           SearchSequence.d() does to the result of SearchWord / SearchField, Search.cs:308) */
typedef struct dcd_query_opts {
  int32_t max_found_pages;   /* Index.MAX_FOUND_PAGES, Index.cs:101 (30000) */
  int32_t max_found_docs;    /* Index.MAX_FOUND_DOCS, Index.cs:102 (500) */
  float doc_rank_multiply;   /* Index.DOC_RANK_MULTIPLY, Index.cs:115 (10) */
  int32_t want_sequences;    /* also return the final IndexSequence units per query */
  /* "-filter:" of the request (Search.cs:455-466): one byte per document of the index, 1 = the document's name matches
   * one of the filter regexes (the host evaluates them: Regex.Match(doc.Name, filt), Search.cs:389-393).  A document with
   * 0 is never added to foundDocs (its pages stay in foundPages), PrepareSearchResult Search.cs:386-396.  NULL = no filter. */
  const uint8_t* doc_match;
  uint64_t n_doc_match;      /* must equal the number of documents when doc_match != NULL */
  /* 1: the batch is evaluated and every result array (pages, positions, ranks, documents) is PRODUCED on the device as
   * always, but only the per-query counts (status, page_off, doc_off) and the totals are copied back: the page / pos / doc
   * arrays of the view are NULL.  For throughput measurements and for callers that only need counts (WordInfo.nFound-style
   * statistics): the D2H copy of the CSR arrays of 100 k phrase queries is tens of gigabytes. */
  int32_t no_download;
  int32_t reserved;          /* 0 */
} dcd_query_opts;
void dcd_query_default_opts(dcd_query_opts* o);
int32_t dcd_query_batch(dcd_index* h, const int32_t* prog, const uint64_t* prog_off, size_t n_queries,
                        const dcd_query_opts* opts, dcd_results** out);

/* Flat (CSR) view of a result batch; pointers stay valid until dcd_results_free. */
typedef struct dcd_results_view {
  uint64_t n_queries;
  const int32_t* status;        /* per query: DCD_OK, or DCD_E_NOTFOUND if a coordinate lies beyond the
                                   last page (the reference returns ErrorSearchResult, Search.cs:609) */
  const uint64_t* page_off;     /* [n_queries+1] into the page arrays (foundPages, corpus order) */
  const uint32_t* page_index;   /* index of the page in the page table */
  const float* page_rank;       /* ResultDocPage.rank, Search.cs:99-111 */
  const uint64_t* pos_off;      /* [total pages + 1] into pos */
  const int32_t* pos;           /* page-relative positions, Search.cs:381,407 */
  const uint64_t* doc_off;      /* [n_queries+1] into the doc arrays (foundDocs, ascending rank) */
  const uint32_t* doc_index;
  const float* doc_rank;        /* Search.cs:552-557 */
  const uint32_t* doc_n_pages;  /* doc.pages.Count */
  const uint64_t* seq_off;      /* [n_queries+1] into seq_units (only if want_sequences) */
  const uint16_t* seq_units;
  const int32_t* seq_R;
  float ms_device;              /* device time of the batch (HIP events) */
  uint64_t n_pages_total, n_pos_total, n_docs_total;   /* sizes of the page / position / document arrays (also with no_download) */
  /* ABI version 2.  Common sub-expressions: operators (* / +) the programs of the batch name, and how many were evaluated -- an
   * operator with the same operands, R values and opcode as an earlier one of the (sub-)batch refers to that one's result;
   * units_*: ushort units of the index-term operands of those operators (2 bytes each: the bytes the operators had to read) */
  uint64_t ops_named, ops_evaluated, units_named, units_evaluated;
  uint64_t sub_batches;         /* 1, or more when the results of the batch's operators did not fit the work pool together */
} dcd_results_view;
int32_t dcd_results_get(dcd_results* r, dcd_results_view* out);
void dcd_results_free(dcd_results* r);

/* ------------------------------------------------------------------------------------------
 * Doc-range shard merge ("reduce by key"): the device-side equivalent of MergeFiles/MergeIndexes
 * with shift_coords (Index.cs:577-752): equal terms' sequences are concatenated in shard order and
 * the first value of every appended sequence is re-coded against the previous shard's last value
 * (IndexSequence.Shift + AddRange semantics, Index.cs:665-674).  Shards were built with
 * dcd_build_opts.coord_base = text length of all previous shards, so coordinates are already global.
 * A term is owned by shard-merge participant  hash(key) % n_owners;  the caller moves the three
 * export buffers between GPUs itself (the functions of this section never communicate), or lets the library do it:
 * dcd_build_merge_distributed below.
 * ------------------------------------------------------------------------------------------ */
typedef struct dcd_term_meta {
  uint64_t first, last;   /* first / last absolute coordinate of this shard's postings of the term */
  uint64_t n_values;
  uint32_t n_units;
  uint32_t key_len;       /* UTF-16 code units */
} dcd_term_meta;
/* Sizes of the export, per owner (host arrays of n_owners entries each). */
int32_t dcd_build_export_plan(dcd_build* h, uint32_t n_owners, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count);
/* Balance by POSTINGS, not by terms.  hash(key) % n_owners spreads the terms evenly, but a Zipf vocabulary puts 13 % of all
 * posting units into its first term: at 8 owners the owner of that term receives 1.9 x the mean.  dcd_build_export_heavy
 * lists the terms of this shard with at least min_units units (64-bit key fingerprint + units, sorted by fingerprint; n_out =
 * how many there are, DCD_E_OVERFLOW if more than cap); the caller all-gathers those lists, assigns the heavy terms to owners
 * (largest first, to the least loaded owner -- every rank computes the same table from the same data) and exports with
 * dcd_build_export_plan_balanced: a term whose fingerprint is in over_fp goes to over_owner; every other (light) term goes to
 * hash(key) % n_owners as before when light_thr == NULL, else to the first owner o with hash(key) <= light_thr[o] (n_owners
 * ascending thresholds, the last one 2^32 - 1): the slice of the hash space an owner gets shrinks by what its heavy terms already
 * weigh, so that every owner ends near the mean.  (No reference counterpart: the reference merges on one host, Index.cs:577-752.) */
int32_t dcd_build_export_heavy(dcd_build* h, uint64_t min_units, uint64_t* fp, uint64_t* n_units, size_t cap, size_t* n_out);
int32_t dcd_build_export_plan_balanced(dcd_build* h, uint32_t n_owners, const uint64_t* over_fp, const uint32_t* over_owner, size_t n_over,
                                       const uint32_t* light_thr, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count);
/* Fills caller-allocated DEVICE buffers: meta (dcd_term_meta per term), keys (UTF-16 units) and
 * posting units, all grouped by owner (owner 0 first) and in ordinal term order inside an owner. */
int32_t dcd_build_export_fill(dcd_build* h, void* meta_dev, uint16_t* keys_dev, uint16_t* units_dev);
/* Merges what n_sources shards sent to this owner.  The three device buffers hold the sources'
 * contributions back to back in source (= coordinate) order; *_count are host arrays [n_sources].
 * The result is an index over the owned terms (hash-partitioned sub-index). */
int32_t dcd_merge_shards(int32_t device, uint32_t n_sources, const uint64_t* term_count, const uint64_t* key_count,
                         const uint64_t* unit_count, const void* meta_dev, const uint16_t* keys_dev, const uint16_t* units_dev,
                         uint64_t max_coord, dcd_index** out);
/* The same merge in two halves, so that the first one can run while the posting units are still travelling:
 * dcd_merge_begin needs only the segment table and the keys (vocabulary, ordinal ranks, output layout, term strings);
 * dcd_merge_finish copies the units (re-coding the first delta of every appended segment) and returns the index.
 * meta_dev must stay valid until dcd_merge_finish; finish consumes the handle (also on error), dcd_merge_free drops a
 * merge that will not be finished.  dcd_merge_shards == begin + finish. */
typedef struct dcd_merge dcd_merge;
int32_t dcd_merge_begin(int32_t device, uint32_t n_sources, const uint64_t* term_count, const uint64_t* key_count,
                        const uint64_t* unit_count, const void* meta_dev, const uint16_t* keys_dev, uint64_t max_coord,
                        dcd_merge** out);
int32_t dcd_merge_finish(dcd_merge* m, const uint16_t* units_dev, dcd_index** out);
void dcd_merge_free(dcd_merge* m);
/* Page table of an index whose pages are uniform (synthetic corpora): n_pages pages of page_chars
 * characters starting at coordinate 0, one document per page named first_doc + p under `source`. */
/* The owned partition after dcd_merge_shards can be exported again (same buffers as dcd_build_export_*), e.g. with
 * n_owners = 1 to GATHER all partitions on one GPU: dcd_merge_shards over the gathered buffers (every term now has
 * exactly one segment) gives the complete index, dcd_index_image / dcd_index_save its ".index" -- byte-identical to the
 * one a single builder writes for the whole corpus.  The page table of a merged index is attached with
 * dcd_index_set_list_image (concatenation of the shards' ".index.list" images, shards split at document borders) or
 * dcd_index_set_uniform_pages. */
int32_t dcd_index_export_plan(dcd_index* h, uint32_t n_owners, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count);
int32_t dcd_index_export_fill(dcd_index* h, void* meta_dev, uint16_t* keys_dev, uint16_t* units_dev);
/* (as dcd_build_export_heavy / dcd_build_export_plan_balanced, for a shard that is itself a merged index) */
int32_t dcd_index_export_heavy(dcd_index* h, uint64_t min_units, uint64_t* fp, uint64_t* n_units, size_t cap, size_t* n_out);
int32_t dcd_index_export_plan_balanced(dcd_index* h, uint32_t n_owners, const uint64_t* over_fp, const uint32_t* over_owner, size_t n_over,
                                       const uint32_t* light_thr, uint64_t* term_count, uint64_t* key_count, uint64_t* unit_count);
int32_t dcd_index_image(dcd_index* h, uint8_t* out, size_t cap, size_t* n);
int32_t dcd_index_list_image(dcd_index* h, uint8_t* out, size_t cap, size_t* n);
int32_t dcd_index_set_list_image(dcd_index* h, const uint8_t* list_img, size_t n_list);
int32_t dcd_index_save(dcd_index* h, const char* dir_utf8);
int32_t dcd_index_set_uniform_pages(dcd_index* h, uint64_t n_pages, uint32_t page_chars, uint64_t first_doc,
                                    const uint16_t* source, size_t source_len);
/* Out-of-core tail of a build whose index does not fit the device (the reference's temp-index flush + MergeAll, Build.cs:335-340,
 * Index.cs:537-573): the doc-range segments are exported to K hash partitions and spilled (host memory or files), every
 * partition is merged on the device by itself (dcd_merge_shards) and saved as a sorted run (dcd_index_save), and this call
 * merges the runs -- ".index" files whose key sets are DISJOINT -- into one ".index" in ordinal key order: a streaming k-way
 * merge of records on the host (MergeFiles, Index.cs:577-689, for inputs that never share a key: nothing is decoded).
 * MaxCoord = max of the inputs (Index.cs:596).  A key in two inputs is DCD_E_BADARG.  The result is opened with
 * dcd_index_open or, when it does not fit, dcd_index_open_lazy. */
int32_t dcd_merge_index_files(const char* const* paths_utf8, size_t n, const char* out_path_utf8);

/* ------------------------------------------------------------------------------------------
 * The exchange INSIDE the library (round 4): a host that drives several GPUs -- one process per GPU, or one process that owns all
 * of them -- builds one doc-range shard per GPU (dcd_build_opts.coord_base = text length of all previous shards,
 * DCD_BUILD_SHARD_ONLY) and calls dcd_build_merge_distributed on every rank: heavy-term all-gather -> balanced owners ->
 * all-to-all-v of meta / keys / units (RCCL ncclSend / ncclRecv groups over xGMI, on the communicator's own stream) ->
 * dcd_merge_begin while the units travel -> dcd_merge_finish.  This is what CreateAsync does with its index tasks and MergeAll
 * on one host (Index.cs:464-488, 537-573).  Every call below is COLLECTIVE: all ranks of the communicator call it, in the same
 * order; in a single process that owns several communicators (dcd_comm_create_all) each communicator is driven by its own host
 * thread (as the reference's index tasks are, Index.cs:464-465).
 * ------------------------------------------------------------------------------------------ */
typedef struct dcd_comm dcd_comm;
#define DCD_COMM_RCCL 0   /* RCCL (librccl.so.1, loaded on first use) over xGMI / PCIe */
#define DCD_COMM_HOST 1   /* buffers staged through POSIX shared memory: tests and rehearsals (several ranks on one GPU), not for speed */
#define DCD_COMM_ID_BYTES 128
/* rank 0 makes the id and hands it to the other ranks by whatever means the host has (the 128 bytes of an ncclUniqueId) */
int32_t dcd_comm_unique_id(int32_t transport, uint8_t id[DCD_COMM_ID_BYTES]);
/* device < 0 with DCD_COMM_HOST: a host-only communicator (dcd_comm_allgather works, the merge calls need a device) */
int32_t dcd_comm_create(int32_t n_ranks, int32_t rank, const uint8_t id[DCD_COMM_ID_BYTES], int32_t device, int32_t transport, dcd_comm** out);
/* one process, n devices: out[i] is the communicator of devices[i] with rank i (ncclCommInitAll) */
int32_t dcd_comm_create_all(int32_t n, const int32_t* devices, int32_t transport, dcd_comm** out);
void dcd_comm_free(dcd_comm* c);
int32_t dcd_comm_rank(dcd_comm* c, int32_t* rank, int32_t* n_ranks);
/* all[r * bytes ..] = rank r's `mine` (host memory, small: e.g. the shards' text lengths, whose prefix sums are the coord_base values) */
int32_t dcd_comm_allgather(dcd_comm* c, const void* mine, size_t bytes, void* all);
/* this rank's shard -> this rank's hash partition of the global index (complete posting lists of the terms it owns).
 * balanced != 0: owners balanced by posting units (heavy terms placed one by one, light terms over weighted hash slices). */
int32_t dcd_build_merge_distributed(dcd_build* b, dcd_comm* c, int32_t balanced, dcd_index** owned);
int32_t dcd_index_merge_distributed(dcd_index* shard, dcd_comm* c, int32_t balanced, dcd_index** owned);   /* shard = several segments merged on the device */
/* every rank ends with a REPLICA of the complete index (query batches then shard by query, no routing); its ".index" is what a
 * single builder writes for the whole corpus.  Attach the page table with dcd_index_set_list_image / dcd_index_set_uniform_pages. */
int32_t dcd_index_allgather(dcd_index* owned, dcd_comm* c, dcd_index** replica);
/* the complete index on rank dst only (*out = NULL elsewhere) */
int32_t dcd_index_gather(dcd_index* owned, dcd_comm* c, int32_t dst, dcd_index** out);
/* The owner table of a balanced export as a pure function (every rank computes it from the same all-gathered lists): list l =
 * fp / units [list_off[l], list_off[l + 1]) are the heavy terms of shard l, totals[l] its posting units.  Out: the heavy terms'
 * fingerprints (ascending) with their owners, and light_thr[n_owners] for dcd_build_export_plan_balanced. */
int32_t dcd_balance_owners(uint32_t n_owners, uint32_t n_lists, const uint64_t* list_off, const uint64_t* fp, const uint64_t* units,
                           const uint64_t* totals, uint64_t* out_fp, uint32_t* out_owner, size_t cap, size_t* n_out, uint32_t* light_thr);

/* ------------------------------------------------------------------------------------------
 * Measurement support (bench.py / tests): the synthetic corpus of SURVEY.md Appendix F.
 * ------------------------------------------------------------------------------------------ */
/* n_docs documents of exactly doc_bytes ASCII bytes each, Zipf(s) over vocab terms, per-doc
 * splitmix64 stream whose state starts at mix(seed + doc * 0xD1B54A32D192ED03) (the
 * survey's seed + doc * GOLDEN would make consecutive documents shifted copies).  Host and device
 * variants produce identical bytes. */
int32_t dcd_synth_corpus_host(uint8_t* out, uint64_t first_doc, uint64_t n_docs, uint32_t doc_bytes,
                              uint32_t vocab, double zipf_s, uint64_t seed);
int32_t dcd_synth_corpus_dev(void* out_dev, uint64_t first_doc, uint64_t n_docs, uint32_t doc_bytes,
                             uint32_t vocab, double zipf_s, uint64_t seed, int32_t device);
/* term string of Zipf rank k (1-based) -> ASCII bytes, returns length via *len */
int32_t dcd_synth_term(uint32_t rank, uint32_t vocab, char* out, size_t cap, size_t* len);
/* Zipf ranks restricted to [lo_rank, vocab]; query i draws from its own splitmix64 stream */
int32_t dcd_synth_query_ranks(uint32_t* out, uint64_t n_queries, uint32_t terms_per_query, int32_t distinct,
                              uint32_t lo_rank, uint32_t vocab, double zipf_s, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* DOCODO_HIP_H */

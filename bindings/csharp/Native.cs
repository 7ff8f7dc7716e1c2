// P/Invoke surface of libdocodo_hip.so for Docodo.NET (see INTEGRATION.md).  Source only: the build image has no
// .NET toolchain, so this file is not compiled here; it mirrors include/docodo_hip.h one to one.
using System;
using System.Runtime.InteropServices;

namespace Docodo
{
    internal static unsafe class Native
    {
        const string Lib = "docodo_hip";   // libdocodo_hip.so next to Docodo.NET.dll or on LD_LIBRARY_PATH

        [StructLayout(LayoutKind.Sequential)]
        public struct BuildOpts { public int device, min_word_length, max_word_length, encoding;
                                  public ulong coord_base; public uint vocab_capacity_log2, flags; }   // flags: 1 = DCD_BUILD_SHARD_ONLY
        [StructLayout(LayoutKind.Sequential)]
        public struct QueryOpts { public int max_found_pages, max_found_docs; public float doc_rank_multiply; public int want_sequences;
                                  public byte* doc_match; public ulong n_doc_match; public int no_download, reserved; }   // -filter: one byte per document, null = none
        [StructLayout(LayoutKind.Sequential)]
        public struct ResultsView { public ulong n_queries; public int* status; public ulong* page_off; public uint* page_index;
                                    public float* page_rank; public ulong* pos_off; public int* pos; public ulong* doc_off;
                                    public uint* doc_index; public float* doc_rank; public uint* doc_n_pages;
                                    public ulong* seq_off; public ushort* seq_units; public int* seq_R; public float ms_device;
                                    public ulong n_pages_total, n_pos_total, n_docs_total;
                                    // ABI 2: common sub-expressions of the batch (operators named / evaluated, units of their index terms)
                                    public ulong ops_named, ops_evaluated, units_named, units_evaluated, sub_batches; }
        public const int DCD_COMM_RCCL = 0, DCD_COMM_HOST = 1, DCD_COMM_ID_BYTES = 128, DCD_ABI_VERSION = 2;

        [DllImport(Lib)] public static extern int dcd_last_error(byte* buf, UIntPtr cap);
        [DllImport(Lib)] public static extern int dcd_version();
        // sizeof of the ABI structs as the library was built (0 BuildOpts, 1 build stats, 2 QueryOpts, 3 ResultsView, 4 term meta):
        // IndexGpu's static constructor compares them with Marshal.SizeOf of the declarations above
        [DllImport(Lib)] public static extern ulong dcd_struct_size(int which);
        [DllImport(Lib)] public static extern int dcd_device_count(out int n);
        // device-memory cache of the library: Index.Dispose() trims it so that the memory goes back to the driver
        [DllImport(Lib)] public static extern int dcd_pool_trim(int device);
        [DllImport(Lib)] public static extern int dcd_set_build_events(int level, uint kernel_mask);
        [DllImport(Lib)] public static extern int dcd_pool_cached_bytes(int device, out ulong bytes);

        // IndexSequence.cs
        [DllImport(Lib)] public static extern int dcd_seq_encode(ulong* values, UIntPtr n, ushort* outp, UIntPtr cap, out UIntPtr nUnits);
        [DllImport(Lib)] public static extern int dcd_seq_decode(ushort* units, UIntPtr nUnits, ulong* outp, UIntPtr cap, out UIntPtr n);
        [DllImport(Lib)] public static extern int dcd_seq_count(ushort* units, UIntPtr nUnits, out UIntPtr nValues);
        [DllImport(Lib)] public static extern int dcd_seq_shift(ushort* units, UIntPtr nUnits, ulong shift, ushort* outp, UIntPtr cap, out UIntPtr nOut);
        [DllImport(Lib)] public static extern int dcd_seq_and(ushort* a, UIntPtr na, int Ra, ushort* b, UIntPtr nb, int Rb,
                                                              ushort* outp, UIntPtr cap, out UIntPtr nOut, out int R);
        [DllImport(Lib)] public static extern int dcd_seq_or(ushort* a, UIntPtr na, int Ra, ushort* b, UIntPtr nb, int Rb,
                                                             ushort* outp, UIntPtr cap, out UIntPtr nOut, out int R);
        // Build.cs
        [DllImport(Lib)] public static extern void dcd_build_default_opts(out BuildOpts o);
        [DllImport(Lib)] public static extern int dcd_build_create(ref BuildOpts o, out IntPtr h);
        [DllImport(Lib)] public static extern void dcd_build_free(IntPtr h);
        [DllImport(Lib)] public static extern int dcd_build_add_doc(IntPtr h, char* src, UIntPtr srcLen, char* name, UIntPtr nameLen);
        [DllImport(Lib)] public static extern int dcd_build_add_pages(IntPtr h, void* text, ulong nChars, ulong* pageOff, UIntPtr nPages,
                                                                      char* pageIds, ulong* idsOff, int docsPerPage, ulong firstDoc,
                                                                      char* src, UIntPtr srcLen);
        [DllImport(Lib)] public static extern int dcd_build_add(IntPtr h, char* code, UIntPtr len, ulong coord);
        [DllImport(Lib)] public static extern int dcd_build_add_word(IntPtr h, char* word, UIntPtr len, ulong coord);
        [DllImport(Lib)] public static extern int dcd_build_end_page(IntPtr h, char* id, UIntPtr len);
        // GetWordCodes (Build.cs:175-247) once per DISTINCT word: words out, codes in, then finish
        [DllImport(Lib)] public static extern int dcd_build_words(IntPtr h, out UIntPtr nWords, out UIntPtr nChars);
        [DllImport(Lib)] public static extern int dcd_build_words_get(IntPtr h, char* chars, UIntPtr charsCap, uint* off, UIntPtr offCap);
        [DllImport(Lib)] public static extern int dcd_build_set_codes(IntPtr h, uint* codeOff, UIntPtr nWords, uint* strOff, char* chars, UIntPtr nEntries);
        [DllImport(Lib)] public static extern int dcd_build_finish(IntPtr h);
        [DllImport(Lib)] public static extern int dcd_build_save(IntPtr h, [MarshalAs(UnmanagedType.LPUTF8Str)] string dir);
        [DllImport(Lib)] public static extern int dcd_build_add_pages_dev(IntPtr h, void* textDev, ulong nChars, ulong* pageOffHost, UIntPtr nPages,
                                                                          int docsPerPage, ulong firstDoc, char* src, UIntPtr srcLen);
        [DllImport(Lib)] public static extern int dcd_build_add_uniform_pages_dev(IntPtr h, void* textDev, ulong nPages, uint pageChars, ulong firstDoc,
                                                                                  char* src, UIntPtr srcLen);
        [DllImport(Lib)] public static extern int dcd_build_get_stats(IntPtr h, void* stats);   // dcd_build_stats, include/docodo_hip.h
        [DllImport(Lib)] public static extern int dcd_build_index_image(IntPtr h, byte* outp, UIntPtr cap, out UIntPtr n);
        [DllImport(Lib)] public static extern int dcd_build_list_image(IntPtr h, byte* outp, UIntPtr cap, out UIntPtr n);
        // doc-range shards (Index.cs:537-752), piece by piece: export -> (the host moves the buffers) -> merge; dcd_build_merge_distributed does all of it
        [DllImport(Lib)] public static extern int dcd_build_export_plan(IntPtr h, uint nOwners, ulong* termCount, ulong* keyCount, ulong* unitCount);
        [DllImport(Lib)] public static extern int dcd_build_export_fill(IntPtr h, void* metaDev, ushort* keysDev, ushort* unitsDev);
        [DllImport(Lib)] public static extern int dcd_build_export_heavy(IntPtr h, ulong minUnits, ulong* fp, ulong* nUnits, UIntPtr cap, out UIntPtr nOut);
        [DllImport(Lib)] public static extern int dcd_build_export_plan_balanced(IntPtr h, uint nOwners, ulong* overFp, uint* overOwner, UIntPtr nOver, uint* lightThr,
                                                                                 ulong* termCount, ulong* keyCount, ulong* unitCount);
        [DllImport(Lib)] public static extern int dcd_index_export_plan(IntPtr h, uint nOwners, ulong* termCount, ulong* keyCount, ulong* unitCount);
        [DllImport(Lib)] public static extern int dcd_index_export_fill(IntPtr h, void* metaDev, ushort* keysDev, ushort* unitsDev);
        [DllImport(Lib)] public static extern int dcd_index_export_heavy(IntPtr h, ulong minUnits, ulong* fp, ulong* nUnits, UIntPtr cap, out UIntPtr nOut);
        [DllImport(Lib)] public static extern int dcd_index_export_plan_balanced(IntPtr h, uint nOwners, ulong* overFp, uint* overOwner, UIntPtr nOver, uint* lightThr,
                                                                                 ulong* termCount, ulong* keyCount, ulong* unitCount);
        [DllImport(Lib)] public static extern int dcd_merge_shards(int device, uint nSources, ulong* termCount, ulong* keyCount, ulong* unitCount,
                                                                   void* metaDev, ushort* keysDev, ushort* unitsDev, ulong maxCoord, out IntPtr index);
        // the same merge in two halves: begin needs only meta + keys and can run while the units are still in flight
        [DllImport(Lib)] public static extern int dcd_merge_begin(int device, uint nSources, ulong* termCount, ulong* keyCount, ulong* unitCount,
                                                                  void* metaDev, ushort* keysDev, ulong maxCoord, out IntPtr merge);
        [DllImport(Lib)] public static extern int dcd_merge_finish(IntPtr merge, ushort* unitsDev, out IntPtr index);
        [DllImport(Lib)] public static extern void dcd_merge_free(IntPtr merge);
        // Index.cs / Search.cs
        [DllImport(Lib)] public static extern int dcd_index_from_images(byte* indexImg, UIntPtr nIndex, byte* listImg, UIntPtr nList, int device, out IntPtr h);
        [DllImport(Lib)] public static extern int dcd_index_count(IntPtr h, out ulong nTerms);       // Index.Count
        [DllImport(Lib)] public static extern int dcd_index_max_coord(IntPtr h, out ulong v);        // Index.MaxCoord
        [DllImport(Lib)] public static extern int dcd_index_n_units(IntPtr h, out ulong n);
        [DllImport(Lib)] public static extern int dcd_index_n_pages(IntPtr h, out ulong n);
        [DllImport(Lib)] public static extern int dcd_index_n_docs(IntPtr h, out ulong n);
        [DllImport(Lib)] public static extern int dcd_index_key(IntPtr h, uint term, char* outp, UIntPtr cap, out UIntPtr len);
        [DllImport(Lib)] public static extern int dcd_index_page_info(IntPtr h, ulong page, out ulong endCoord, out uint doc, char* id, UIntPtr cap, out UIntPtr len);
        [DllImport(Lib)] public static extern int dcd_index_doc_name(IntPtr h, uint doc, char* outp, UIntPtr cap, out UIntPtr len);
        [DllImport(Lib)] public static extern int dcd_index_image(IntPtr h, byte* outp, UIntPtr cap, out UIntPtr n);
        [DllImport(Lib)] public static extern int dcd_index_list_image(IntPtr h, byte* outp, UIntPtr cap, out UIntPtr n);
        [DllImport(Lib)] public static extern int dcd_index_set_list_image(IntPtr h, byte* listImg, UIntPtr nList);
        [DllImport(Lib)] public static extern int dcd_index_set_uniform_pages(IntPtr h, ulong nPages, uint pageChars, ulong firstDoc, char* src, UIntPtr srcLen);
        [DllImport(Lib)] public static extern int dcd_index_open([MarshalAs(UnmanagedType.LPUTF8Str)] string dir, int device, out IntPtr h);
        [DllImport(Lib)] public static extern int dcd_index_from_build(IntPtr build, out IntPtr h);
        // out-of-core tail (Index.cs:537-689): k-way merge of the sorted ".index" runs of disjoint partitions
        [DllImport(Lib)] public static extern int dcd_merge_index_files(IntPtr* pathsUtf8, UIntPtr n, [MarshalAs(UnmanagedType.LPUTF8Str)] string outPath);
        // Index(path, InMemory: false), Index.cs:133-136, 346-348: postings stay in the file
        [DllImport(Lib)] public static extern int dcd_index_open_lazy([MarshalAs(UnmanagedType.LPUTF8Str)] string dir, int device, ulong residentBytes, out IntPtr h);
        [DllImport(Lib)] public static extern int dcd_index_lazy_stats(IntPtr h, out ulong listsRead, out ulong bytesRead, out ulong residentTerms);
        [DllImport(Lib)] public static extern int dcd_index_close_words(IntPtr h, char* word, UIntPtr len, uint n, uint* terms, out uint nOut);   // GetCloseWords
        [DllImport(Lib)] public static extern int dcd_index_like_words(IntPtr h, char* pattern, UIntPtr len, uint n, uint* terms, out uint nOut);   // GetLikeWords
        [DllImport(Lib)] public static extern int dcd_index_suggest(IntPtr h, char* prefix, UIntPtr len, uint n, uint* terms, out uint nOut);      // GetSuggessions
        [DllImport(Lib)] public static extern int dcd_index_save(IntPtr h, [MarshalAs(UnmanagedType.LPUTF8Str)] string dir);
        [DllImport(Lib)] public static extern void dcd_index_free(IntPtr h);
        [DllImport(Lib)] public static extern int dcd_index_prepare(IntPtr h, out float msDevice);   // resident coordinates + cell directories now (else: first batch)
        [DllImport(Lib)] public static extern int dcd_index_trim(IntPtr h);          // gives the query work arena back (several indexes on one device)
        // the exchange inside the library (round 4): one communicator per GPU; every call below is collective
        [DllImport(Lib)] public static extern int dcd_comm_unique_id(int transport, byte* id128);
        [DllImport(Lib)] public static extern int dcd_comm_create(int nRanks, int rank, byte* id128, int device, int transport, out IntPtr comm);
        [DllImport(Lib)] public static extern int dcd_comm_create_all(int n, int* devices, int transport, IntPtr* comms);
        [DllImport(Lib)] public static extern void dcd_comm_free(IntPtr comm);
        [DllImport(Lib)] public static extern int dcd_comm_rank(IntPtr comm, out int rank, out int nRanks);
        [DllImport(Lib)] public static extern int dcd_comm_allgather(IntPtr comm, void* mine, UIntPtr bytes, void* all);
        [DllImport(Lib)] public static extern int dcd_build_merge_distributed(IntPtr build, IntPtr comm, int balanced, out IntPtr owned);
        [DllImport(Lib)] public static extern int dcd_index_merge_distributed(IntPtr shard, IntPtr comm, int balanced, out IntPtr owned);
        [DllImport(Lib)] public static extern int dcd_index_allgather(IntPtr owned, IntPtr comm, out IntPtr replica);
        [DllImport(Lib)] public static extern int dcd_index_gather(IntPtr owned, IntPtr comm, int dst, out IntPtr index);
        [DllImport(Lib)] public static extern int dcd_balance_owners(uint nOwners, uint nLists, ulong* listOff, ulong* fp, ulong* units, ulong* totals,
                                                                     ulong* outFp, uint* outOwner, UIntPtr cap, out UIntPtr nOut, uint* lightThr);
        [DllImport(Lib)] public static extern int dcd_index_find(IntPtr h, char* key, UIntPtr len, out uint term);
        [DllImport(Lib)] public static extern int dcd_index_get(IntPtr h, uint term, ushort* outp, UIntPtr cap, out UIntPtr nUnits);
        [DllImport(Lib)] public static extern int dcd_index_term_count(IntPtr h, uint term, out ulong nValues);
        [DllImport(Lib)] public static extern void dcd_query_default_opts(out QueryOpts o);
        [DllImport(Lib)] public static extern int dcd_query_batch(IntPtr h, int* prog, ulong* progOff, UIntPtr nQueries, ref QueryOpts o, out IntPtr results);
        [DllImport(Lib)] public static extern int dcd_results_get(IntPtr r, out ResultsView v);
        [DllImport(Lib)] public static extern void dcd_results_free(IntPtr r);

        public static void Check(int status) {
            if (status == 0) return;
            byte* buf = stackalloc byte[1024];
            dcd_last_error(buf, (UIntPtr)1024);
            throw new InvalidOperationException("docodo_hip " + status + ": " + Marshal.PtrToStringUTF8((IntPtr)buf));
        }
    }
}

// IndexGpuMulti.cs -- the members of the drop-in that round 4 added (partial class Index, continues IndexGpu.cs):
//
//   reference member                                     here
//   --------------------------------------------------   -----------------------------------------------------------------------
//   Index(path, InMemory: false)    Index.cs:133-136,    LoadGpu() chooses dcd_index_open_lazy when InMemory == false (postings stay
//     LoadSequence                  198-216, 346-348     in ".index"; this[key] / a query batch reads the lists it names)
//   GetLikeWords / GetCloseWords /  Search.cs:158-188    GetLikeWordsGpu / GetCloseWordsGpu / GetSuggessionsGpu (dictionary scans on the
//     GetSuggessions                                     device: dcd_index_like_words / close_words / suggest)
//   Builder.Add temp flush          Build.cs:335-340     GpuSegmentedBuilder: a segment (DCD_BUILD_SHARD_ONLY build with global
//     + MergeAll                    Index.cs:537-573     coordinates) per MaxTmpIndexItems postings, dcd_merge_shards at the end
//   out-of-core tail                Index.cs:577-689     GpuSegmentedBuilder.FinishOutOfCore: K hash partitions, one merge + one sorted
//                                                        run per partition, dcd_merge_index_files, Load with InMemory = false
//   CreateAsync: parallel index     Index.cs:424-534     CreateAsyncGpu(devices): one index task per GPU over a document range, the
//     tasks + MergeAll                                   exchange + merge inside the library (dcd_comm_create_all,
//     + MergeIndexes                                     dcd_build_merge_distributed, dcd_index_gather), ".index" written once
//
// SOURCE ONLY (no .NET toolchain in the build image): written against the reference's declarations and Native.cs; the static
// constructor checks the struct layouts against the library at start-up (dcd_struct_size), which is where a mismatch would show.
using System;
using System.Collections.Generic;
using System.IO;
using System.Linq;
using System.Runtime.InteropServices;
using System.Text.RegularExpressions;
using System.Threading;
using System.Threading.Tasks;

namespace Docodo
{
    public unsafe partial class Index
    {
        /// <summary>Device bytes the lists of an on-disk index may occupy between batches (dcd_index_open_lazy); 0 = the library's default.</summary>
        public ulong GpuResidentBytes = 0;

        static Index()
        {
            // ABI check: a host compiled against another header would read / write the structs at the wrong offsets
            if (Native.dcd_version() != Native.DCD_ABI_VERSION) throw new InvalidOperationException("libdocodo_hip: ABI version " + Native.dcd_version());
            if (Native.dcd_struct_size(0) != (ulong)Marshal.SizeOf<Native.BuildOpts>() || Native.dcd_struct_size(2) != (ulong)Marshal.SizeOf<Native.QueryOpts>() ||
                Native.dcd_struct_size(3) != (ulong)Marshal.SizeOf<Native.ResultsView>())
                throw new InvalidOperationException("libdocodo_hip: struct layout differs from Native.cs");
        }

        // ---------------------------------------------------------------------------------------------------------------
        // Index.Load with both InMemory settings (Index.cs:133-136, 312-380): InMemory -> the posting lists live on the device;
        // !InMemory -> stubs only (Index.cs:346-348), LoadSequence reads a list when it is asked for (Index.cs:198-216)
        // ---------------------------------------------------------------------------------------------------------------
        public bool LoadGpuLazyAware()
        {
            if (!File.Exists(Path.Combine(WorkPath, ".index")) || !File.Exists(Path.Combine(WorkPath, ".index.list"))) return false;
            CanSearch = false;
            DisposeGpu();
            try
            {
                if (InMemory) Native.Check(Native.dcd_index_open(WorkPath, GpuDevice, out gpuIndex));
                else Native.Check(Native.dcd_index_open_lazy(WorkPath, GpuDevice, GpuResidentBytes, out gpuIndex));
                ulong mc; Native.Check(Native.dcd_index_max_coord(gpuIndex, out mc));
                MaxCoord = mc;
                CanSearch = true;
                return true;
            }
            catch (Exception e)
            {
                Console.WriteLine("Can't load: " + e.Message);     // as Index.cs:373-376
                return false;
            }
        }

        // ---------------------------------------------------------------------------------------------------------------
        // dictionary scans, Search.cs:158-188
        // ---------------------------------------------------------------------------------------------------------------
        string[] TermStrings(uint[] terms, uint n)
        {
            var res = new string[n];
            char* buf = stackalloc char[256];
            for (uint i = 0; i < n; i++)
            {
                UIntPtr len;
                Native.Check(Native.dcd_index_key(gpuIndex, terms[i], buf, (UIntPtr)256, out len));
                res[i] = new string(buf, 0, (int)len);
            }
            return res;
        }

        /// GetLikeWords, Search.cs:158-167: keys whose first unit is a letter and that match the pattern ('_' = ".*", unanchored),
        /// the first MAX_LIKE_WORDS in dictionary order
        public string[] GetLikeWordsGpu(string word)
        {
            if ((word.IndexOf('_') < 0) || (!bKeepForms)) return new string[] { word };
            if (word.Length < 2) return new string[] { };
            var terms = new uint[MAX_LIKE_WORDS];
            uint n;
            fixed (char* w = word) fixed (uint* t = terms)
                Native.Check(Native.dcd_index_like_words(gpuIndex, w, (UIntPtr)word.Length, (uint)MAX_LIKE_WORDS, t, out n));
            return TermStrings(terms, n);
        }

        /// GetCloseWords, Search.cs:168-174: the 10 keys closest to `word` (Levenshtein, Index.cs:46-95), ties in dictionary order
        public string[] GetCloseWordsGpu(string word)
        {
            var terms = new uint[10];
            uint n;
            fixed (char* w = word) fixed (uint* t = terms)
                Native.Check(Native.dcd_index_close_words(gpuIndex, w, (UIntPtr)word.Length, 10, t, out n));
            return TermStrings(terms, n);
        }

        /// GetSuggessions, Search.cs:176-188: completions of the request's last word by descending IndexSequence.Count
        public Task<string[]> GetSuggessionsGpu(string req, int n = 10)
        {
            if (req.Length < 2) return Task.FromResult(new string[] { });
            string lastword = Regex.Split(req, @"\b").Last((s) => s.Length > 0).ToLower();
            if (lastword.Length < 2) return Task.FromResult(new string[] { });
            return Task.Factory.StartNew<string[]>(() =>
            {
                var terms = new uint[Math.Max(1, n)];
                uint got;
                fixed (char* w = lastword) fixed (uint* t = terms)
                    Native.Check(Native.dcd_index_suggest(gpuIndex, w, (UIntPtr)lastword.Length, (uint)n, t, out got));
                return TermStrings(terms, got).Select((s) => s.Substring(lastword.Length)).ToArray();
            });
        }

        // ---------------------------------------------------------------------------------------------------------------
        // Builder.Add's temp-index flush + MergeAll (Build.cs:335-340, Index.cs:537-573) with the device as the merge engine:
        // a SEGMENT = a DCD_BUILD_SHARD_ONLY build whose coord_base is the text length of everything before it, closed every
        // MaxTmpIndexItems postings; Finish() merges the segments' exports on the device (dcd_merge_shards re-codes the first delta
        // of every appended segment: IndexSequence.Shift + AddRange, Index.cs:665-674) and writes ".index" / ".index.list" once.
        // ---------------------------------------------------------------------------------------------------------------
        public class GpuSegmentedBuilder : IDisposable
        {
            readonly Index Parent;
            readonly int Device;
            readonly long MaxItems;                               // Index.MaxTmpIndexItems, Build.cs:335
            ulong coordBase = 0;
            GpuBuilder cur = null;
            long curItems = 0;
            struct Export { public ulong nTerms, nKeys, nUnits, maxCoord; public IntPtr meta, keys, units; public byte[] list; }
            readonly List<Export> exports = new List<Export>();
            [DllImport("amdhip64")] static extern int hipMalloc(out IntPtr p, UIntPtr bytes);
            [DllImport("amdhip64")] static extern int hipFree(IntPtr p);

            public GpuSegmentedBuilder(Index parent, long maxItems = 0)
            {
                Parent = parent; Device = parent.GpuDevice;
                MaxItems = maxItems > 0 ? maxItems : parent.MaxTmpIndexItems;
            }

            GpuBuilder Cur() { if (cur == null) { cur = new GpuBuilder(Parent, coordBase, shardOnly: true); curItems = 0; } return cur; }

            public void AddDoc(string sourceid, string name) { Cur().AddDoc(sourceid, name); }

            /// pages of one document (Build.cs:477-541); a segment is closed once it holds MaxItems postings (estimated, as the
            /// reference counts them, by the tokens added so far: one posting per kept token)
            public void AddPages(IList<string> ids, IList<string> texts)
            {
                Cur().AddPages(ids, texts);
                foreach (string t in texts) { coordBase += (ulong)t.Length; curItems += t.Length / 6; }
                if (curItems > MaxItems) Flush();                                       // Build.cs:335-340
            }

            void Flush()
            {
                if (cur == null) return;
                cur.Finish();
                var e = new Export();
                ulong t, k, u;
                Native.Check(Native.dcd_build_export_plan(cur.Handle, 1, &t, &k, &u));
                e.nTerms = t; e.nKeys = k; e.nUnits = u;
                if (hipMalloc(out e.meta, (UIntPtr)Math.Max(1UL, t * 32)) != 0 || hipMalloc(out e.keys, (UIntPtr)Math.Max(1UL, k * 2)) != 0 ||
                    hipMalloc(out e.units, (UIntPtr)Math.Max(1UL, u * 2)) != 0) throw new OutOfMemoryException("hipMalloc");
                Native.Check(Native.dcd_build_export_fill(cur.Handle, (void*)e.meta, (ushort*)e.keys, (ushort*)e.units));
                e.maxCoord = cur.MaxCoord;
                e.list = cur.ListImage();                                               // this segment's part of ".index.list"
                exports.Add(e);
                cur.Dispose(); cur = null;
            }

            /// MergeAll + MergeIndexes: one merge on the device, files written once
            public Index Finish()
            {
                Flush();
                int S = exports.Count;
                ulong[] tc = exports.Select(e => e.nTerms).ToArray(), kc = exports.Select(e => e.nKeys).ToArray(), uc = exports.Select(e => e.nUnits).ToArray();
                // the three buffers back to back, in segment (= coordinate) order
                IntPtr meta, keys, units;
                ulong tm = (ulong)tc.Sum(x => (decimal)x) * 32, tk = (ulong)kc.Sum(x => (decimal)x) * 2, tu = (ulong)uc.Sum(x => (decimal)x) * 2;
                if (hipMalloc(out meta, (UIntPtr)Math.Max(1UL, tm)) != 0 || hipMalloc(out keys, (UIntPtr)Math.Max(1UL, tk)) != 0 ||
                    hipMalloc(out units, (UIntPtr)Math.Max(1UL, tu)) != 0) throw new OutOfMemoryException("hipMalloc");
                ulong om = 0, ok = 0, ou = 0;
                foreach (var e in exports)
                {
                    HipCopy(meta + (int)om, e.meta, e.nTerms * 32); HipCopy(keys + (int)ok, e.keys, e.nKeys * 2); HipCopy(units + (int)ou, e.units, e.nUnits * 2);
                    om += e.nTerms * 32; ok += e.nKeys * 2; ou += e.nUnits * 2;
                    hipFree(e.meta); hipFree(e.keys); hipFree(e.units);
                }
                IntPtr merged;
                fixed (ulong* a = tc, b = kc, c = uc)
                    Native.Check(Native.dcd_merge_shards(Device, (uint)S, a, b, c, (void*)meta, (ushort*)keys, (ushort*)units, exports.Max(e => e.maxCoord), out merged));
                hipFree(meta); hipFree(keys); hipFree(units);
                byte[] list = exports.SelectMany(e => e.list).ToArray();                 // segments split at document borders: the images concatenate
                fixed (byte* l = list) Native.Check(Native.dcd_index_set_list_image(merged, l, (UIntPtr)list.Length));
                Native.Check(Native.dcd_index_save(merged, Parent.WorkPath));
                Native.dcd_index_free(merged);
                exports.Clear();
                Parent.LoadGpuLazyAware();
                return Parent;
            }

            [DllImport("amdhip64")] static extern int hipMemcpy(IntPtr dst, IntPtr src, UIntPtr bytes, int kind);
            static void HipCopy(IntPtr dst, IntPtr src, ulong bytes) { if (bytes > 0 && hipMemcpy(dst, src, (UIntPtr)bytes, 3 /* hipMemcpyDeviceToDevice */) != 0) throw new InvalidOperationException("hipMemcpy"); }

            public void Dispose() { if (cur != null) { cur.Dispose(); cur = null; } foreach (var e in exports) { hipFree(e.meta); hipFree(e.keys); hipFree(e.units); } exports.Clear(); }
        }

        // ---------------------------------------------------------------------------------------------------------------
        // CreateAsync on several GPUs (Index.cs:424-534).  The reference starts MaxDegreeOfParallelism index tasks per source, each
        // with its own Builder and coordinate space, then MergeAll + MergeIndexes shift the coordinates (Index.cs:587-593).  Here
        // every GPU gets ONE index task over a contiguous document range; the ranges' text lengths are known after the
        // documents were read, so every shard encodes GLOBAL coordinates from the start (coord_base = text length of the ranges in
        // front of it) and the merge is the library's collective:
        //     dcd_comm_create_all -> per device thread: GpuBuilder(shardOnly) ... Finish -> dcd_build_merge_distributed
        //                         -> dcd_index_gather(dst: 0) -> rank 0 attaches the page lists and writes ".index" / ".index.list".
        // ---------------------------------------------------------------------------------------------------------------
        public async Task CreateAsyncGpu(int[] devices)
        {
            if ((sources == null) || (sources.Length == 0) || (sources[0] == null) || status != Status.Idle) { CreationDone?.Invoke(this, new EventArgs()); return; }
            status = Status.Nav;
            await Task.Run(() =>
            {
                try
                {
                    int G = devices.Length;
                    // 1. read the documents once (the text cache does the same, DataSources.cs:579-610) and cut them into G contiguous ranges
                    //    of about equal text length
                    var docs = new List<(string src, string name, List<string> ids, List<string> texts, ulong chars)>();
                    foreach (IIndexDataSource source in sources)
                    {
                        source.Reset();
                        for (IIndexDocument doc = source.Next(); doc != null; doc = source.Next())
                            using (doc)
                            {
                                var ids = new List<string>(); var texts = new List<string>(); ulong n = 0;
                                foreach (IndexPage page in doc) { if (page.text.Length == 0) continue; ids.Add(page.id); texts.Add(page.text); n += (ulong)page.text.Length; }
                                docs.Add((source.Name, doc.Name, ids, texts, n));
                            }
                    }
                    ulong total = 0; foreach (var d in docs) total += d.chars;
                    var first = new int[G + 1]; var baseOf = new ulong[G + 1];
                    { int g = 0; ulong acc = 0; for (int i = 0; i < docs.Count; i++) { while (g + 1 < G && acc >= total * (ulong)(g + 1) / (ulong)G) { g++; first[g] = i; baseOf[g] = acc; } acc += docs[i].chars; } for (int q = g + 1; q <= G; q++) { first[q] = docs.Count; baseOf[q] = acc; } }
                    status = Status.Index;
                    // 2. one communicator and one thread per device (every dcd_* call below is collective: same order on every thread)
                    var comms = new IntPtr[G];
                    fixed (int* dv = devices) fixed (IntPtr* cp = comms) Native.Check(Native.dcd_comm_create_all(G, dv, Native.DCD_COMM_RCCL, cp));
                    var lists = new byte[G][];
                    IntPtr full = IntPtr.Zero;
                    var errors = new List<Exception>();
                    var threads = new Thread[G];
                    for (int g = 0; g < G; g++)
                    {
                        int rank = g;
                        threads[g] = new Thread(() =>
                        {
                            try
                            {
                                var shardParent = this;
                                using (var b = new GpuBuilder(shardParent, baseOf[rank], shardOnly: true, device: devices[rank]))
                                {
                                    for (int i = first[rank]; i < first[rank + 1]; i++)
                                    {
                                        b.AddDoc(docs[i].src, docs[i].name);                        // Build.cs:474
                                        if (docs[i].texts.Count > 0) b.AddPages(docs[i].ids, docs[i].texts);
                                    }
                                    b.Finish();
                                    lists[rank] = b.ListImage();
                                    status = Status.Merge;
                                    IntPtr owned, got;
                                    Native.Check(Native.dcd_build_merge_distributed(b.Handle, comms[rank], 1, out owned));
                                    Native.Check(Native.dcd_index_gather(owned, comms[rank], 0, out got));
                                    Native.dcd_index_free(owned);
                                    if (rank == 0) full = got;
                                }
                            }
                            catch (Exception e) { lock (errors) errors.Add(e); }
                        });
                        threads[g].Start();
                    }
                    foreach (var t in threads) t.Join();
                    foreach (var c in comms) Native.dcd_comm_free(c);
                    if (errors.Count > 0) throw errors[0];
                    // 3. rank 0 holds the complete index: page lists of the shards back to back (shards split at document borders), files once
                    byte[] list = lists.SelectMany(l => l).ToArray();
                    fixed (byte* l = list) Native.Check(Native.dcd_index_set_list_image(full, l, (UIntPtr)list.Length));
                    Native.Check(Native.dcd_index_save(full, WorkPath));
                    Native.dcd_index_free(full);
                    LoadGpuLazyAware();                                                 // Load(); CanSearch = true  (Index.cs:512-513)
                    status = Status.Idle;
                }
                catch (Exception e)
                {
                    Console.WriteLine("Error: " + e.Message);                           // Index.cs:521-526
                    status = Status.Idle;
                    CanSearch = false;
                }
            });
            CreationDone?.Invoke(this, new EventArgs());
        }
    }
}

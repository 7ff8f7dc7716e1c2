// IndexGpu.cs -- the C# side of the drop-in: partial-class members that keep the public surface of Docodo.Index /
// Index.Builder / IndexSequence / Index.Search while the numeric work runs in libdocodo_hip.so (Native.cs).
//
// SOURCE ONLY: the build image has no .NET toolchain, so this file has never been compiled; it is written against the
// reference's own declarations (paths relative to Docodo.NET/) and follows INTEGRATION.md section 2 row by row:
//
//   reference member                                   here
//   ------------------------------------------------   ------------------------------------------------------------------
//   Index.Load()                    Index.cs:312-380   LoadGpu()                -> dcd_index_open / dcd_index_open_lazy (InMemory == false)
//   Index.this[key] / LoadSequence  Index.cs:198-263   GpuSequence(key)         -> dcd_index_find + dcd_index_get
//   Index.Count / MaxCoord          Index.cs:255, 310  GpuCount / GpuMaxCoord
//   IndexTask(source)               Build.cs:445-564   IndexTaskGpu(source)     -> GpuBuilder.AddPages + Finish + Save
//   Index.Builder                   Build.cs:258-437   Index.GpuBuilder (same method names and argument meaning)
//   GetWordCodes per token          Build.cs:175-247   GpuBuilder.Finish: once per DISTINCT word (dcd_build_words / set_codes)
//   MergeAll / MergeIndexes         Index.cs:537-752   not called (one build per coordinate space; shards: dcd_merge_shards)
//   Index.Search(req, opt)          Search.cs:437-617  SearchGpu(req, opt)      -> postfix plan + dcd_query_batch
//   IndexSequence operators * + &   IndexSequence.cs:205-322   IndexSequenceGpu.Mul / Add (dcd_seq_and / dcd_seq_or)
//   IndexSequence.Builder / enumerator / Shift         IndexSequenceGpu.Encode / Decode / CountValues / Shift
//   Index.Dispose()                 Index.cs:247-251   DisposeGpu()             -> dcd_index_free + dcd_pool_trim
//
// To switch the reference over, a maintainer (1) adds Native.cs and this file to Docodo.NET.csproj with
// <AllowUnsafeBlocks>true</AllowUnsafeBlocks>, (2) replaces the bodies of Load(), IndexTask(), Search() and this[key] by
// calls to the *Gpu members below (one line each; the old bodies stay available as the CPU path when UseGpu == false),
// (3) in IndexSequence.cs forwards operator * / + / & to IndexSequenceGpu when IndexSequenceGpu.Enabled.
// Nothing else in the reference changes: data sources, text cache, snippets, REST server keep calling the same members.
using System;
using System.Collections.Generic;
using System.IO;
using System.Linq;
using System.Text;
using System.Text.RegularExpressions;

namespace Docodo
{
    /// <summary>IndexSequence algebra on the device (per-object path: keeps DynamicExpresso evaluation unchanged).</summary>
    public static unsafe class IndexSequenceGpu
    {
        public static bool Enabled = true;

        static ushort[] Units(IndexSequence s)
        {
            // IndexSequence exposes its List<ushort> through this[int] and Count (IndexSequence.cs:43-45, 158)
            var a = new ushort[s.Count];
            for (int i = 0; i < a.Length; i++) a[i] = s[i];
            return a;
        }

        /// IndexSequence.Builder.AddRange(values).build(), IndexSequence.cs:63-90
        public static IndexSequence Encode(ulong[] values)
        {
            var outp = new ushort[Math.Max(1, 5 * values.Length)];
            UIntPtr n;
            fixed (ulong* v = values) fixed (ushort* o = outp)
                Native.Check(Native.dcd_seq_encode(v, (UIntPtr)values.Length, o, (UIntPtr)outp.Length, out n));
            return new IndexSequence(new ArraySegment<ushort>(outp, 0, (int)n));
        }

        /// the enumerator over the whole sequence, IndexSequence.cs:114-136
        public static ulong[] Decode(IndexSequence s)
        {
            var u = Units(s);
            var outp = new ulong[Math.Max(1, u.Length)];
            UIntPtr n;
            fixed (ushort* p = u) fixed (ulong* o = outp)
                Native.Check(Native.dcd_seq_decode(p, (UIntPtr)u.Length, o, (UIntPtr)outp.Length, out n));
            Array.Resize(ref outp, (int)n);
            return outp;
        }

        /// LINQ Count() (Search.cs:307): number of VALUES
        public static int CountValues(IndexSequence s)
        {
            var u = Units(s);
            UIntPtr n;
            fixed (ushort* p = u) Native.Check(Native.dcd_seq_count(p, (UIntPtr)u.Length, out n));
            return (int)n;
        }

        /// IndexSequence.Shift, IndexSequence.cs:191-202 (returns the shifted sequence; the reference mutates in place)
        public static IndexSequence Shift(IndexSequence s, ulong shift)
        {
            var u = Units(s);
            var outp = new ushort[u.Length + 8];
            UIntPtr n;
            fixed (ushort* p = u) fixed (ushort* o = outp)
                Native.Check(Native.dcd_seq_shift(p, (UIntPtr)u.Length, shift, o, (UIntPtr)outp.Length, out n));
            return new IndexSequence(new ArraySegment<ushort>(outp, 0, (int)n)) { R = s.R };
        }

        delegate int BinOp(ushort* a, UIntPtr na, int Ra, ushort* b, UIntPtr nb, int Rb, ushort* outp, UIntPtr cap, out UIntPtr nOut, out int R);

        static IndexSequence Bin(BinOp f, IndexSequence a, IndexSequence b)
        {
            var ua = Units(a); var ub = Units(b);
            var outp = new ushort[Math.Max(1, 5 * (ua.Length + ub.Length))];
            UIntPtr n; int R;
            fixed (ushort* pa = ua) fixed (ushort* pb = ub) fixed (ushort* o = outp)
                Native.Check(f(pa, (UIntPtr)ua.Length, a.R, pb, (UIntPtr)ub.Length, b.R, o, (UIntPtr)outp.Length, out n, out R));
            return new IndexSequence(new ArraySegment<ushort>(outp, 0, (int)n)) { R = R };
        }

        /// operator * and operator &, IndexSequence.cs:205-284
        public static IndexSequence Mul(IndexSequence a, IndexSequence b) { return Bin(Native.dcd_seq_and, a, b); }
        /// operator +, IndexSequence.cs:286-322
        public static IndexSequence Add(IndexSequence a, IndexSequence b) { return Bin(Native.dcd_seq_or, a, b); }
    }

    public unsafe partial class Index
    {
        /// <summary>false = the reference's CPU bodies; true = libdocodo_hip.  There is no silent fallback: with UseGpu set, a
        /// missing library or device surfaces as the exception Native.Check throws.</summary>
        public bool UseGpu = true;
        public int GpuDevice = 0;
        IntPtr gpuIndex = IntPtr.Zero;

        const int DCD_ENC_UTF16LE = 2, DCD_E_NOTFOUND = -7;
        const int DCD_OP_TERM = 0, DCD_OP_AND = 1, DCD_OP_OR = 2, DCD_OP_SETR = 3;

        // ---------------------------------------------------------------------------------------------------------------
        // Index.Load, Index.cs:312-380 (InMemory: the device holds the postings; !InMemory: they stay in ".index" and are read on
        // demand, Index.cs:198-216, 346-348; the files stay loadable by the unmodified reference)
        // ---------------------------------------------------------------------------------------------------------------
        public bool LoadGpu() { return LoadGpuLazyAware(); }          // IndexGpuMulti.cs: dcd_index_open, or dcd_index_open_lazy when InMemory == false

        /// Index.Count, Index.cs:253
        public int GpuCount { get { ulong n; Native.Check(Native.dcd_index_count(gpuIndex, out n)); return (int)n; } }

        /// this[key], Index.cs:255-263 (a missing key throws KeyNotFoundException like the SortedList indexer)
        public IndexSequence GpuSequence(string key)
        {
            uint term;
            fixed (char* k = key)
            {
                int st = Native.dcd_index_find(gpuIndex, k, (UIntPtr)key.Length, out term);
                if (st == DCD_E_NOTFOUND) throw new KeyNotFoundException(key);
                Native.Check(st);
            }
            UIntPtr n;
            Native.dcd_index_get(gpuIndex, term, null, UIntPtr.Zero, out n);          // size query (DCD_E_OVERFLOW by design)
            var units = new ushort[Math.Max(1, (int)n)];
            fixed (ushort* u = units) Native.Check(Native.dcd_index_get(gpuIndex, term, u, (UIntPtr)units.Length, out n));
            return new IndexSequence(new ArraySegment<ushort>(units, 0, (int)n));
        }

        /// Index.Dispose(), Index.cs:247-251
        public void DisposeGpu()
        {
            if (gpuIndex != IntPtr.Zero) { Native.dcd_index_free(gpuIndex); gpuIndex = IntPtr.Zero; Native.dcd_pool_trim(GpuDevice); }
        }

        // ---------------------------------------------------------------------------------------------------------------
        // Index.Builder, Build.cs:258-437 -- same members, same call order (AddDoc ... EndPage ... Build / Save)
        // ---------------------------------------------------------------------------------------------------------------
        public class GpuBuilder : IDisposable
        {
            readonly Index Parent;
            IntPtr h;
            public string Path { get; }
            public ulong maxCoord { get; private set; } = 0;
            public int TotalCount { get; private set; } = 0;
            bool finished = false;

            /// the native handle (IndexGpuMulti.cs: export / dcd_build_merge_distributed)
            internal IntPtr Handle { get { return h; } }
            /// maxCoord of the finished build (Build.cs:326, 377)
            internal ulong MaxCoord { get { ulong* st = stackalloc ulong[32]; Native.Check(Native.dcd_build_get_stats(h, st)); return st[6]; } }   // dcd_build_stats.max_coord
            /// this build's ".index.list" bytes (IndexPageList.Save, Build.cs:99-118)
            internal byte[] ListImage()
            {
                UIntPtr n;
                Native.dcd_build_list_image(h, null, UIntPtr.Zero, out n);               // size query (DCD_E_OVERFLOW by design)
                var img = new byte[Math.Max(1, (int)n)];
                fixed (byte* p = img) Native.Check(Native.dcd_build_list_image(h, p, (UIntPtr)img.Length, out n));
                Array.Resize(ref img, (int)n);
                return img;
            }

            public GpuBuilder(Index parent, ulong coordBase = 0, bool shardOnly = false, int device = -1)
            {
                Parent = parent;
                Path = parent.WorkPath;
                Native.BuildOpts o;
                Native.dcd_build_default_opts(out o);
                o.device = device >= 0 ? device : parent.GpuDevice;
                o.encoding = DCD_ENC_UTF16LE;                          // a C# string pins to UTF-16 code units
                o.min_word_length = parent.MIN_WORD_LENGTH;            // Index.cs:113
                o.max_word_length = Index.MAX_WORD_LENGTH;             // Index.cs:97
                o.coord_base = coordBase;
                o.flags = shardOnly ? 1u : 0u;                         // DCD_BUILD_SHARD_ONLY
                Native.Check(Native.dcd_build_create(ref o, out h));
            }

            /// Builder.AddDoc(sourceid, name), Build.cs:348-356
            public void AddDoc(string sourceid, string name)
            {
                if (sourceid.Length == 0) throw new Exception("sourceid must not be empty");
                fixed (char* s = sourceid) fixed (char* n = name)
                    Native.Check(Native.dcd_build_add_doc(h, s, (UIntPtr)sourceid.Length, n, (UIntPtr)name.Length));
            }

            /// The per-page body of IndexTask (Build.cs:477-541) for all pages of one document in one call: lower-casing,
            /// Regex.Matches(c, @"\p{L}+|\p{N}+"), the length filter, AddWord(m.Value, coord + m.Index), coord += c.Length and
            /// EndPage(id, coord) run on the device; header pages (id "0", Build.cs:485-524) are recognised by their id.
            public void AddPages(IList<string> ids, IList<string> texts)
            {
                var sb = new StringBuilder();
                var off = new ulong[texts.Count + 1];
                var idsb = new StringBuilder();
                var idOff = new ulong[texts.Count + 1];
                for (int p = 0; p < texts.Count; p++)
                {
                    sb.Append(texts[p]); off[p + 1] = (ulong)sb.Length;
                    idsb.Append(ids[p]); idOff[p + 1] = (ulong)idsb.Length;
                }
                string all = sb.ToString(), allIds = idsb.ToString();
                fixed (char* t = all) fixed (char* i = allIds) fixed (ulong* po = off) fixed (ulong* io = idOff)
                    Native.Check(Native.dcd_build_add_pages(h, t, (ulong)all.Length, po, (UIntPtr)texts.Count, i, io, 0, 0, null, UIntPtr.Zero));
            }

            /// Builder.Add(code, coord), Build.cs:324-341
            public void Add(string code, ulong coord)
            {
                fixed (char* c = code) Native.Check(Native.dcd_build_add(h, c, (UIntPtr)code.Length, coord));
                maxCoord = coord; TotalCount++;
            }

            /// Builder.AddWord(word, coord), Build.cs:302-316 (the codes are attached in Finish, once per distinct word)
            public void AddWord(string word, ulong coord)
            {
                fixed (char* w = word) Native.Check(Native.dcd_build_add_word(h, w, (UIntPtr)word.Length, coord));
                maxCoord = coord; TotalCount++;
            }

            /// Builder.EndPage(id), Build.cs:359-362
            public void EndPage(string id)
            {
                fixed (char* i = id) Native.Check(Native.dcd_build_end_page(h, i, (UIntPtr)id.Length));
            }

            /// Runs the device pipeline.  GetWordCodes (Build.cs:175-247: stop words, "#HEX" vocabulary groups, "$stem" codes) is
            /// called ONCE PER DISTINCT WORD here instead of once per token in AddWord; skipped when the index has neither
            /// stemmers nor vocabularies nor stop words (code == word).
            public void Finish()
            {
                if (finished) return;
                if (Parent.Stemmers.Count > 0 || Parent.vocs.Count > 0 || Parent.stopWords.Count > 0)
                {
                    UIntPtr nWords, nChars;
                    Native.Check(Native.dcd_build_words(h, out nWords, out nChars));
                    var chars = new char[Math.Max(1, (int)nChars)];
                    var woff = new uint[(int)nWords + 1];
                    fixed (char* c = chars) fixed (uint* o = woff)
                        Native.Check(Native.dcd_build_words_get(h, c, (UIntPtr)chars.Length, o, (UIntPtr)woff.Length));
                    var codeOff = new uint[(int)nWords + 1];
                    var strOff = new List<uint> { 0 };
                    var codeChars = new StringBuilder();
                    for (int w = 0; w < (int)nWords; w++)
                    {
                        string word = new string(chars, (int)woff[w], (int)(woff[w + 1] - woff[w]));
                        var codes = Parent.GetWordCodes(word);                 // unchanged reference code
                        uint k = 0;
                        if (codes != null)
                            foreach (var code in codes) { codeChars.Append(code.code); strOff.Add((uint)codeChars.Length); k++; }
                        codeOff[w + 1] = codeOff[w] + k;
                    }
                    string cc = codeChars.ToString();
                    var so = strOff.ToArray();
                    fixed (uint* co = codeOff) fixed (uint* s = so) fixed (char* c = cc)
                        Native.Check(Native.dcd_build_set_codes(h, co, nWords, s, c, (UIntPtr)(so.Length - 1)));
                }
                Native.Check(Native.dcd_build_finish(h));
                finished = true;
            }

            /// Builder.Save(), Build.cs:370-404: leaves .index / .index.list in the work path
            public void Save() { Finish(); Native.Check(Native.dcd_build_save(h, Parent.WorkPath)); }

            /// Builder.Build(), Build.cs:407-434: the parent becomes searchable on the index that is already resident in HBM
            public Index Build()
            {
                lock (Parent.DoSearchLock)
                {
                    Save();
                    Parent.DisposeGpu();
                    Native.Check(Native.dcd_index_from_build(h, out Parent.gpuIndex));
                    ulong mc; Native.Check(Native.dcd_index_max_coord(Parent.gpuIndex, out mc));
                    Parent.MaxCoord = mc;
                    Parent.CanSearch = true;
                }
                return Parent;
            }

            public void Dispose() { if (h != IntPtr.Zero) { Native.dcd_build_free(h); h = IntPtr.Zero; } }
        }

        public GpuBuilder GetGpuBuilder() { return new GpuBuilder(this); }

        // ---------------------------------------------------------------------------------------------------------------
        // IndexTask, Build.cs:445-564: the document loop stays, the per-token loop moves to the device
        // ---------------------------------------------------------------------------------------------------------------
        private void IndexTaskGpu(IIndexDataSource source)
        {
            try
            {
                using (GpuBuilder index = GetGpuBuilder())
                {
                    do
                    {
                        using (IIndexDocument doc = source.Next())
                        {
                            if (doc == null) break;
                            index.AddDoc(source.Name, doc.Name);                        // Build.cs:474
                            var ids = new List<string>(); var texts = new List<string>();
                            foreach (IndexPage page in doc)                             // Build.cs:477
                            {
                                if (page.text.Length == 0) continue;                    // Build.cs:483 (c.Length == 0: page skipped)
                                ids.Add(page.id); texts.Add(page.text);
                            }
                            if (texts.Count > 0) index.AddPages(ids, texts);
                        }
                    } while (true);
                    index.Save();                                                       // Build.cs:553
                }
            }
            catch (Exception e)
            {
                Console.WriteLine("Error: " + e.Message);                               // Build.cs:556-560
                status = Status.Idle; CanSearch = false;
            }
        }

        // ---------------------------------------------------------------------------------------------------------------
        // Index.Search, Search.cs:437-617 -- PrepareSearchRequest (string work) stays; the expression it returns is compiled
        // into a postfix plan instead of being handed to DynamicExpresso; the device returns pages / positions / ranks.
        // ---------------------------------------------------------------------------------------------------------------
        struct PlanWord { public string Word; public bool Field; public string FieldName; }

        /// SearchWord (Search.cs:192-260) as a plan fragment: the keys that are OR-ed, whether the result is "exact" (R = -1), and
        /// the number of postings found (WordInfo.nFound = Count() of the sequence, Search.cs:307).  null = new IndexSequence().
        List<int> PlanSearchWord(string word, out bool exact, out ulong nFound)
        {
            exact = false; nFound = 0;
            if (word.ToUpper().Equals(word) && bKeepForms) exact = true;
            word = word.ToLower();
            string[] words = { word };
            if (word.IndexOf('_') >= 0)
            {
                if (!bKeepForms) return null;
                exact = true;
                words = GetLikeWordsGpu(word);                              // dictionary scan on the device (IndexGpuMulti.cs)
            }
            var prog = new List<int>();
            int pushed = 0;
            foreach (string wword in words)
            {
                var codes = GetWordCodes(wword)?.Select((c) => c.code).ToList();
                if (codes == null || codes.Count == 0) continue;
                var selfcodes = codes.Where((s) => Regex.IsMatch(s.Substring(0, 1), @"\w")).ToList();
                var knowcodes = codes.Except(selfcodes).ToList();
                foreach (string code in (!exact ? (knowcodes.Count > 0 ? knowcodes : selfcodes.Take(1).ToList()) : selfcodes.Take(1).ToList()))
                {
                    uint term;
                    fixed (char* k = code)
                        if (Native.dcd_index_find(gpuIndex, k, (UIntPtr)code.Length, out term) != 0) continue;   // self.Keys.Contains(code)
                    ulong n; Native.Check(Native.dcd_index_term_count(gpuIndex, term, out n));
                    nFound += n;                                                                              // (an upper bound when codes overlap)
                    prog.Add(DCD_OP_TERM); prog.Add((int)term); prog.Add(0);
                    if (pushed++ > 0) prog.Add(DCD_OP_OR);                                                    // total += res
                }
            }
            return pushed > 0 ? prog : null;
        }

        /// A compiled operand: its postfix program (null = the empty IndexSequence) and its R.
        struct PlanVal { public List<int> Prog; public int R; }

        /// SearchSequence.d(), Search.cs:302-312, for a plain word or a {field=value} word (SearchField, Search.cs:126-140)
        PlanVal PlanD(SearchSequence s, int dist, Dictionary<string, string> fieldOf)
        {
            bool exact; ulong n;
            List<int> prog;
            int R0;
            string fname;
            if (fieldOf.TryGetValue(s.Name, out fname))
            {
                uint fterm; string key = FIELD_NAME_CHAR + fname.ToLower();
                bool haveField;
                fixed (char* k = key) haveField = Native.dcd_index_find(gpuIndex, k, (UIntPtr)key.Length, out fterm) == 0;
                var v = PlanSearchWord(s.Word.ToLower(), out exact, out n);
                if (!haveField || v == null) { prog = null; n = 0; R0 = 0; }
                else
                {
                    int vR = exact ? -1 : 0;
                    prog = new List<int> { DCD_OP_TERM, (int)fterm, -1 };            // seq.R = -1
                    prog.AddRange(v); prog.Add(DCD_OP_SETR); prog.Add(vR); prog.Add(DCD_OP_AND);
                    R0 = vR < 0 ? -1 : 1;                                           // R of (-1) * x, IndexSequence.cs:221-222
                }
            }
            else
            {
                prog = PlanSearchWord(s.Word, out exact, out n);
                R0 = exact ? -1 : 0;
            }
            s.wordInfo.nFound = (int)n;
            int R = R0 < 0 ? -s.Word.Length - 4 : dist + s.Word.Length;             // Search.cs:308
            if (prog != null) { prog.Add(DCD_OP_SETR); prog.Add(R); }
            return new PlanVal { Prog = prog, R = R };
        }

        static PlanVal Combine(PlanVal a, PlanVal b, int op)
        {
            int absR = Math.Max(Math.Abs(a.R), Math.Abs(b.R));
            int R = (a.R < 0 && b.R < 0) ? -absR : absR;                            // IndexSequence.cs:221-222, 288-289
            if (op == DCD_OP_AND && (a.Prog == null || b.Prog == null)) return new PlanVal { Prog = null, R = R };   // nothing holds both operands
            if (a.Prog == null && b.Prog == null) return new PlanVal { Prog = null, R = R };
            if (a.Prog == null || b.Prog == null)
            {
                var one = new List<int>(a.Prog ?? b.Prog); one.Add(DCD_OP_SETR); one.Add(R);   // x + empty = x with the combined R
                return new PlanVal { Prog = one, R = R };
            }
            var p = new List<int>(a.Prog); p.AddRange(b.Prog); p.Add(op);
            return new PlanVal { Prog = p, R = R };
        }

        /// Postfix plan of one rewritten expression ("A.d()*(B.d()+C.d())") by recursive descent: C# precedence (* before +),
        /// left-associative, parentheses (Search.cs:349-357).  Empty operands are folded on the host exactly as the operators
        /// treat them (a * empty = empty, a + empty = a).  Throws FormatException on a syntax error.
        PlanVal CompilePlan(string expr, List<SearchSequence> seq, int dist, Dictionary<string, string> fieldOf)
        {
            int i = 0;
            Action skip = () => { while (i < expr.Length && char.IsWhiteSpace(expr[i])) i++; };
            Func<PlanVal> parseExpr = null, parseTerm = null, parseFactor = null;
            parseFactor = () =>
            {
                skip();
                if (i >= expr.Length) throw new FormatException("eof");
                if (expr[i] == '(')
                {
                    i++;
                    PlanVal v = parseExpr();
                    skip();
                    if (i >= expr.Length || expr[i] != ')') throw new FormatException(")");
                    i++;
                    return v;
                }
                int j = expr.IndexOf(".d()", i, StringComparison.Ordinal);
                if (j < 0) throw new FormatException("operand");
                string name = expr.Substring(i, j - i);
                SearchSequence s = seq.FirstOrDefault((q) => q.Name == name);
                if (s == null) throw new FormatException(name);
                i = j + 4;
                return PlanD(s, dist, fieldOf);
            };
            parseTerm = () =>
            {
                PlanVal v = parseFactor();
                for (skip(); i < expr.Length && expr[i] == '*'; skip()) { i++; v = Combine(v, parseFactor(), DCD_OP_AND); }
                return v;
            };
            parseExpr = () =>
            {
                PlanVal v = parseTerm();
                for (skip(); i < expr.Length && expr[i] == '+'; skip()) { i++; v = Combine(v, parseTerm(), DCD_OP_OR); }
                return v;
            };
            PlanVal r = parseExpr();
            skip();
            if (i != expr.Length) throw new FormatException("trailing");
            return r;
        }

        public SearchResult SearchGpu(string req, SearchOptions opt = null)
        {
            if (!CanSearch) return new ErrorSearchResult("Index is not built");
            try
            {
                lock (DoSearchLock)
                {
                    req = req.ToLower();
                    var filter = new List<string>();
                    const string fpat = @"\B-filter:([\w\*\?\\.()+{}/]+,?)+";
                    var matches = Regex.Match(req, fpat);                               // Search.cs:455-466
                    if (matches.Groups.Count > 1)
                        foreach (var cap in matches.Groups[1].Captures) filter.Add(cap.ToString().Trim(','));
                    req = Regex.Replace(req, fpat, " ");

                    string fields;
                    var seq = new List<SearchSequence>();
                    var fieldOf = new Dictionary<string, string>();
                    req = PrepareSearchRequestGpu(req, seq, out fields, fieldOf);
                    int R = opt != null ? opt.dist : 255;                               // Search.cs:472-473
                    List<int> plan = null, planf = null;
                    bool haveRes = req.Trim().Length > 0, haveResf = fields.Length > 0;
                    try
                    {
                        if (haveRes) plan = CompilePlan(req, seq, R, fieldOf).Prog;
                        if (haveResf) planf = CompilePlan(fields, seq, R, fieldOf).Prog;
                    }
                    catch (FormatException)
                    {
                        return new ErrorSearchResult("Syntax Error in search request");  // Search.cs:505-510
                    }
                    if (!haveRes) { plan = planf; haveRes = haveResf; planf = null; haveResf = false; }   // if (res == null) res = resf
                    if (!haveRes) return new SearchResult();
                    if (plan == null)                                                    // an empty result sequence: no pages, no documents
                    {
                        var empty = new SearchResult();
                        empty.words = seq.Select((s) => s.wordInfo).ToArray();
                        return empty;
                    }

                    // filter: Regex.Match(doc.Name, filt) per document (Search.cs:389-393) -> one byte per document
                    byte[] docMatch = null;
                    if (filter.Count > 0)
                    {
                        ulong nDocs; Native.Check(Native.dcd_index_n_docs(gpuIndex, out nDocs));
                        docMatch = new byte[nDocs];
                        for (uint d = 0; d < nDocs; d++) { string name = GpuDocName(d); docMatch[d] = (byte)(filter.Any((f) => Regex.Match(name, f).Success) ? 1 : 0); }
                    }
                    SearchResult result = RunPlan(plan, docMatch);
                    if (!result.Success) return result;
                    if (haveResf)
                    {
                        SearchResult rf = planf != null ? RunPlan(planf, null) : new SearchResult();   // CombineSearchResults, Search.cs:423-428
                        if (!rf.Success) return rf;
                        result.foundDocs.IntersectWith(rf.foundDocs);
                    }
                    // ranks come from the device (Search.cs:552-557); headers / snippets / summary stay as in Search.cs:559-597
                    result.foundDocs = new HashSet<ResultDocument>((from doc in result.foundDocs orderby doc.rank select doc).ToArray());
                    result.words = seq.Select((s) => s.wordInfo).ToArray();
                    return result;
                }
            }
            catch (Exception e)
            {
                Console.WriteLine("Error: " + e.Message);
                return new ErrorSearchResult("Error: " + e.Message);                    // Search.cs:609-613
            }
        }

        /// PrepareSearchRequest (Search.cs:319-363) unchanged, plus a note of which variables came from {field=value} groups
        string PrepareSearchRequestGpu(string req, List<SearchSequence> funcs, out string fields, Dictionary<string, string> fieldOf)
        {
            int before = funcs.Count;
            string r = PrepareSearchRequest(req, funcs, out fields);
            // variables created by the recursive field call carry a SearchField closure; recover the field name from the request
            foreach (Match m in Regex.Matches(Regex.Replace(req, @"[^\w(){}=~?|""]|_+", " "), @"{*(\w+)[ ]*=([\w|() ]+)}"))
                foreach (Match w in Regex.Matches(m.Groups[2].Value, @"\b(\w+)\b"))
                {
                    var s = funcs.Skip(before).FirstOrDefault((q) => q.Word == w.Value && !fieldOf.ContainsKey(q.Name));
                    if (s != null) fieldOf[s.Name] = m.Groups[1].Value;
                }
            return r;
        }

        string GpuDocName(uint d)
        {
            var buf = new char[1024]; UIntPtr n;
            fixed (char* b = buf) Native.Check(Native.dcd_index_doc_name(gpuIndex, d, b, (UIntPtr)buf.Length, out n));
            return new string(buf, 0, (int)n);
        }

        /// One query through dcd_query_batch -> SearchResult (PrepareSearchResult, Search.cs:365-420; ranks :99-111, 552-557)
        SearchResult RunPlan(List<int> plan, byte[] docMatch)
        {
            var prog = plan.ToArray();
            var off = new ulong[] { 0, (ulong)prog.Length };
            Native.QueryOpts o; Native.dcd_query_default_opts(out o);
            o.max_found_pages = MAX_FOUND_PAGES; o.max_found_docs = MAX_FOUND_DOCS; o.doc_rank_multiply = DOC_RANK_MULTIPLY;
            IntPtr r;
            fixed (int* p = prog) fixed (ulong* po = off) fixed (byte* dm = docMatch)
            {
                o.doc_match = dm; o.n_doc_match = docMatch != null ? (ulong)docMatch.Length : 0;
                Native.Check(Native.dcd_query_batch(gpuIndex, p, po, (UIntPtr)1, ref o, out r));
            }
            try
            {
                Native.ResultsView v; Native.Check(Native.dcd_results_get(r, out v));
                if (v.status[0] != 0) return new ErrorSearchResult("Error: coordinate beyond the last page");   // Build.cs:96, Search.cs:609
                var result = new SearchResult();
                var pageOfIndex = new Dictionary<uint, ResultDocPage>();
                var pagesOfDoc = new Dictionary<uint, List<ResultDocPage>>();
                for (ulong i = v.page_off[0]; i < v.page_off[1]; i++)
                {
                    ulong end; uint doc; var id = new char[64]; UIntPtr idLen;
                    fixed (char* ip = id) Native.Check(Native.dcd_index_page_info(gpuIndex, v.page_index[i], out end, out doc, ip, (UIntPtr)id.Length, out idLen));
                    var page = new ResultDocPage(new DocPage(null, new string(id, 0, (int)idLen)));
                    for (ulong q = v.pos_off[i]; q < v.pos_off[i + 1]; q++) page.pos.Add(v.pos[q]);
                    result.foundPages.Add(page);
                    List<ResultDocPage> l; if (!pagesOfDoc.TryGetValue(doc, out l)) pagesOfDoc[doc] = l = new List<ResultDocPage>();
                    l.Add(page);
                }
                for (ulong i = v.doc_off[0]; i < v.doc_off[1]; i++)
                {
                    var d = new ResultDocument(GpuDocName(v.doc_index[i]));
                    d.rank = v.doc_rank[i];
                    foreach (var pg in pagesOfDoc[v.doc_index[i]]) d.pages.Add(pg);
                    result.foundDocs.Add(d);
                }
                return result;
            }
            finally { Native.dcd_results_free(r); }
        }
    }
}

// oracle/oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Single-threaded, obviously-correct CPU restatement of the Docodo hot path
// (reference: alexarchen/Docodo, C#).  It is the parity checker for the HIP path and the
// "port" CPU baseline of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; nothing under docodo_amd/ may.
//
// Pinning: the reference cannot be compiled here (no .NET toolchain in the image), so this
// restatement is pinned by the reference's own known-answer tests (SURVEY.md Appendix D,
// tests/golden/known_answers.json, exercised by tests/test_oracle_known_answers.py).  The
// reference holds no golden *bytes* for postings/.index, so byte-level parity is pinned by
// the format definitions cited below plus the round-trip properties of IndexSequenceTest.cs.
//
// Every function cites the reference file:line it follows (paths relative to Docodo.NET/).
//
// Data-structure choices deliberately mirror the reference "in spirit" so that timing this
// code is a fair CPU baseline: ordered map keyed by term with ordinal compare
// (SortedDictionary, Build.cs:258), incremental var-ushort append per posting
// (IndexSequence.cs:63), enumerator-style two-pointer merges (IndexSequence.cs:218-322),
// forward-cursor page lookup (Build.cs:74-97).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <vector>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef std::u16string ustr;  // UTF-16 code units; operator< is String.CompareOrdinal (Index.cs:142-149)

namespace {

// ---------------------------------------------------------------------------------------------
// Character tables (UTF-16 code unit -> class / lower case).  Defaults cover ASCII + Latin-1;
// tests install full BMP tables derived from Python's unicodedata (orc_set_char_tables).
// class bits: 1 = \p{L}, 2 = \p{N}, 4 = \w (L | Mn | Nd | Pc), as .NET Regex defines them.
// ---------------------------------------------------------------------------------------------
u8 g_cls[65536];
u16 g_lower[65536];
bool g_tables_init = false;
u16 g_upper[65536];
void rebuild_upper() {  // String.ToUpper(): inverse of the lower table (exact for ASCII / Latin-1)
  for (int c = 0; c < 65536; c++) g_upper[c] = (u16)c;
  for (int c = 65535; c >= 0; c--) if (g_lower[c] != c) g_upper[g_lower[c]] = (u16)c;
}

void init_default_tables() {
  if (g_tables_init) return;
  for (int c = 0; c < 65536; c++) { g_cls[c] = 0; g_lower[c] = (u16)c; }
  for (int c = 'a'; c <= 'z'; c++) g_cls[c] = 1 | 4;
  for (int c = 'A'; c <= 'Z'; c++) { g_cls[c] = 1 | 4; g_lower[c] = (u16)(c + 32); }
  for (int c = '0'; c <= '9'; c++) g_cls[c] = 2 | 4;
  g_cls['_'] = 4;
  g_cls[0xAA] = g_cls[0xB5] = g_cls[0xBA] = 1 | 4;
  for (int c = 0xC0; c <= 0xFF; c++) if (c != 0xD7 && c != 0xF7) g_cls[c] = 1 | 4;
  for (int c = 0xC0; c <= 0xDE; c++) if (c != 0xD7) g_lower[c] = (u16)(c + 32);
  g_cls[0xB2] = g_cls[0xB3] = g_cls[0xB9] = 2;           // No
  g_cls[0xBC] = g_cls[0xBD] = g_cls[0xBE] = 2;           // No
  g_tables_init = true;
  rebuild_upper();
}
inline bool is_L(u16 c) { return g_cls[c] & 1; }
inline bool is_N(u16 c) { return g_cls[c] & 2; }
inline bool is_w(u16 c) { return g_cls[c] & 4; }

ustr to_lower(const ustr& s) {  // String.ToLower(): per-code-unit simple mapping, length preserving
  ustr r(s);
  for (auto& c : r) c = g_lower[(u16)c];
  return r;
}

// ---------------------------------------------------------------------------------------------
// IndexSequence (IndexSequence.cs)
// ---------------------------------------------------------------------------------------------
const u16 OVERFLOW_ = 0x8000;  // IndexSequence.cs:13-15
const u16 MASK_ = 0x7FFF;
const int BITS_ = 15;

struct Seq {
  std::vector<u16> self;
  int R = 0;  // IndexSequence.cs:163
};

// IndexSequence.Builder.Add, IndexSequence.cs:63-84 (min == 0 on the hot path)
struct SeqBuilder {
  Seq seq;
  u64 Last = 0;
  void Add(u64 l) {
    u64 diff = l - Last;
    do {
      if (diff > MASK_) {
        seq.self.push_back((u16)(OVERFLOW_ | (diff & MASK_)));
        diff >>= BITS_;
      } else {
        seq.self.push_back((u16)diff);
        break;
      }
    } while (diff > 0);
    Last = l;
  }
};

// SeqEnumerator.MoveNext, IndexSequence.cs:114-136.  C# masks a ulong shift count to 6 bits.
struct SeqEnum {
  const u16* p;
  size_t n, i = 0;
  u64 Last = 0;
  SeqEnum(const Seq& s) : p(s.self.data()), n(s.self.size()) {}
  SeqEnum(const u16* p_, size_t n_) : p(p_), n(n_) {}
  bool MoveNext() {
    bool more = false;
    int shift = 0;
    do {
      if (i < n) {
        u16 value = p[i++];
        more = (value & OVERFLOW_) != 0;
        Last += ((u64)(value & MASK_)) << (shift & 63);
        shift += BITS_;
      } else {
        return false;
      }
    } while (more);
    return true;
  }
  u64 Current() const { return Last; }
};

// IndexSequence.Shift, IndexSequence.cs:191-202
void seq_shift(Seq& s, u64 shift) {
  if (s.self.empty() || shift == 0) return;
  SeqEnum en(s);
  en.MoveNext();
  SeqBuilder b;
  b.Add(shift + en.Current());
  size_t q = 0;
  do { } while (q < s.self.size() && (s.self[q++] & OVERFLOW_) != 0);
  s.self.erase(s.self.begin(), s.self.begin() + q);
  s.self.insert(s.self.begin(), b.seq.self.begin(), b.seq.self.end());
}

// operator *, IndexSequence.cs:218-284 (operator & is the same, :205-215) -- literal transcription.
Seq seq_mul(const Seq& seq1, const Seq& seq2) {
  int absR = std::max(std::abs(seq1.R), std::abs(seq2.R));
  int R = ((seq1.R < 0) && (seq2.R < 0)) ? -absR : absR;
  SeqBuilder newSeq;
  newSeq.seq.R = R;
  std::vector<u64> newGroup;
  SeqEnum enums[2] = {SeqEnum(seq1), SeqEnum(seq2)};
  bool move[2] = {true, true};
  bool can[2] = {true, true};
  bool IsInGr[2] = {false, false};
  do {
    u64 valToAdd = 0;
    if (!move[0] && !move[1]) break;
    if (move[0] && can[0]) can[0] = enums[0].MoveNext();
    if (move[1] && can[1]) can[1] = enums[1].MoveNext();
    move[0] = move[1] = false;
    if (!can[0] && !can[1]) break;

    if (!can[0]) { valToAdd = enums[1].Current(); move[1] = true; }
    else if (!can[1]) { valToAdd = enums[0].Current(); move[0] = true; }
    else if (enums[0].Current() < enums[1].Current()) { valToAdd = enums[0].Current(); move[0] = true; }
    else if (enums[0].Current() > enums[1].Current()) { valToAdd = enums[1].Current(); move[1] = true; }
    else { move[0] = move[1] = true; valToAdd = enums[1].Current(); }

    if (!newGroup.empty()) {
      bool bfinish = false;
      if ((absR != 0) && (valToAdd - newGroup.back() > (u64)absR)) bfinish = true;
      else if ((R < 0) && move[0] && !IsInGr[0]) bfinish = true;
      if (bfinish) {
        if (IsInGr[0] && IsInGr[1])
          for (u64 v : newGroup) newSeq.Add(v);
        IsInGr[0] = IsInGr[1] = false;
        newGroup.clear();
        if (!can[0] || !can[1]) break;
      }
    }
    if (move[0]) IsInGr[0] = true;
    if (move[1]) IsInGr[1] = true;
    newGroup.push_back(valToAdd);
  } while (can[0] || can[1]);

  if (IsInGr[0] && IsInGr[1])
    for (u64 v : newGroup) newSeq.Add(v);
  return newSeq.seq;
}

// operator +, IndexSequence.cs:286-322 -- literal transcription.
Seq seq_add(const Seq& seq1, const Seq& seq2) {
  int absR = std::max(std::abs(seq1.R), std::abs(seq2.R));
  int R = ((seq1.R < 0) && (seq2.R < 0)) ? -absR : absR;
  SeqBuilder res;
  res.seq.R = R;
  SeqEnum seqe[2] = {SeqEnum(seq1), SeqEnum(seq2)};
  bool Move[2] = {true, true};
  bool Exists[2] = {false, false};
  do {
    for (int q = 0; q < 2; q++) {
      if (Move[q]) Exists[q] = seqe[q].MoveNext();
      Move[q] = false;
    }
    if (Exists[0] || Exists[1]) {
      if (!Exists[1]) { res.Add(seqe[0].Current()); Move[0] = true; }
      else if (!Exists[0]) { res.Add(seqe[1].Current()); Move[1] = true; }
      else {
        if (seqe[0].Current() <= seqe[1].Current()) Move[0] = true;
        if (seqe[1].Current() <= seqe[0].Current()) Move[1] = true;
        if (Move[0]) res.Add(seqe[0].Current());
        else res.Add(seqe[1].Current());
      }
    }
  } while (Exists[0] || Exists[1]);
  return res.seq;
}

size_t seq_count(const Seq& s) {  // LINQ Count() over the enumerator (Search.cs:307)
  SeqEnum e(s);
  size_t n = 0;
  while (e.MoveNext()) n++;
  return n;
}

// ---------------------------------------------------------------------------------------------
// BinaryWriter/BinaryReader helpers (.NET): 7-bit length-prefixed UTF-8 strings, LE integers.
// ---------------------------------------------------------------------------------------------
void utf16_to_utf8(const ustr& s, std::string& out) {
  for (size_t i = 0; i < s.size(); i++) {
    u32 c = (u16)s[i];
    if (c >= 0xD800 && c <= 0xDBFF && i + 1 < s.size() && (u16)s[i + 1] >= 0xDC00 && (u16)s[i + 1] <= 0xDFFF) {
      c = 0x10000 + ((c - 0xD800) << 10) + ((u16)s[i + 1] - 0xDC00);
      i++;
    } else if (c >= 0xD800 && c <= 0xDFFF) {
      c = 0xFFFD;  // lone surrogate: UTF8Encoding replacement fallback
    }
    if (c < 0x80) out.push_back((char)c);
    else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) {
      out.push_back((char)(0xE0 | (c >> 12))); out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (c & 0x3F)));
    } else {
      out.push_back((char)(0xF0 | (c >> 18))); out.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
      out.push_back((char)(0x80 | ((c >> 6) & 0x3F))); out.push_back((char)(0x80 | (c & 0x3F)));
    }
  }
}
bool utf8_to_utf16(const u8* p, size_t n, ustr& out) {
  size_t i = 0;
  while (i < n) {
    u32 c = p[i];
    int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0;
    if (!len || i + len > n) { out.push_back(0xFFFD); i++; continue; }
    if (len == 2) c = ((c & 0x1F) << 6) | (p[i + 1] & 0x3F);
    else if (len == 3) c = ((c & 0x0F) << 12) | ((p[i + 1] & 0x3F) << 6) | (p[i + 2] & 0x3F);
    else if (len == 4) c = ((c & 0x07) << 18) | ((p[i + 1] & 0x3F) << 12) | ((p[i + 2] & 0x3F) << 6) | (p[i + 3] & 0x3F);
    i += len;
    if (c >= 0x10000) { c -= 0x10000; out.push_back((char16_t)(0xD800 + (c >> 10))); out.push_back((char16_t)(0xDC00 + (c & 0x3FF))); }
    else out.push_back((char16_t)c);
  }
  return true;
}
struct BinW {
  std::vector<u8> b;
  void u64le(u64 v) { for (int i = 0; i < 8; i++) b.push_back((u8)(v >> (8 * i))); }
  void i32le(int32_t v) { for (int i = 0; i < 4; i++) b.push_back((u8)((u32)v >> (8 * i))); }
  void str(const ustr& s) {  // BinaryWriter.Write(string)
    std::string u; utf16_to_utf8(s, u);
    u32 n = (u32)u.size();
    while (n >= 0x80) { b.push_back((u8)(n | 0x80)); n >>= 7; }
    b.push_back((u8)n);
    b.insert(b.end(), u.begin(), u.end());
  }
  void seq(const Seq& s) {  // IndexSequence.Write, IndexSequence.cs:167-173
    i32le((int32_t)s.self.size());
    for (u16 v : s.self) { b.push_back((u8)v); b.push_back((u8)(v >> 8)); }
  }
};
struct BinR {
  const u8* p; size_t n, i = 0;
  bool eof = false;
  BinR(const u8* p_, size_t n_) : p(p_), n(n_) {}
  u64 u64le() { if (i + 8 > n) { eof = true; i = n; return 0; } u64 v = 0; for (int k = 0; k < 8; k++) v |= (u64)p[i + k] << (8 * k); i += 8; return v; }
  int32_t i32le() { if (i + 4 > n) { eof = true; i = n; return 0; } u32 v = 0; for (int k = 0; k < 4; k++) v |= (u32)p[i + k] << (8 * k); i += 4; return (int32_t)v; }
  ustr str() {
    u32 len = 0; int sh = 0;
    while (true) { if (i >= n) { eof = true; return ustr(); } u8 c = p[i++]; len |= (u32)(c & 0x7F) << sh; sh += 7; if (!(c & 0x80)) break; }
    if (i + len > n) { eof = true; i = n; return ustr(); }
    ustr s; utf8_to_utf16(p + i, len, s); i += len; return s;
  }
};

// ---------------------------------------------------------------------------------------------
// Index configuration shared by build and search (Index.cs:96-115, 151-183, 219-235)
// ---------------------------------------------------------------------------------------------
typedef int (*stem_fn)(const u16* word, int len, u16* out, int cap);  // returns out length

struct Vocab {                    // Dict.cs:13-104
  std::map<ustr, int> words;
  u16 range_begin = 0, range_end = 0;
  stem_fn stemmer = nullptr;      // Vocab.Stem: identity when null (Dict.cs:65)
};
struct Stemmer { ustr range; stem_fn fn; };  // Index.cs:175-183 (lang, stemmer, range)

struct PageEntry { u64 end; ustr id; int doc; u64 coord; };  // Dictionary<ulong, DocPage> in insertion order

struct Index {
  // config
  int MIN_WORD_LENGTH = 3, MAX_WORD_LENGTH = 32;          // Index.cs:97,113
  int MAX_FOUND_PAGES = 30000, MAX_FOUND_DOCS = 500;      // Index.cs:101-102
  float DOC_RANK_MULTIPLY = 10;                           // Index.cs:115
  bool bKeepForms = true;                                 // Index.cs:154
  std::set<ustr> stopWords;
  std::vector<Vocab> vocs;
  std::vector<Stemmer> stemmers;
  // loaded index (Index.Load, Index.cs:312-380), always "InMemory" here
  std::map<ustr, Seq> self;
  u64 MaxCoord = 0;
  // IndexPageList (Build.cs:41-149)
  std::vector<ustr> docs;
  std::vector<PageEntry> pages;
  u64 prevc = 0xFFFFFFFFull;  // Build.cs:51
  size_t cur = 0;             // _enum position; pages.size() means "past the end"
};

bool range_allows(const ustr& range, const ustr& word) {
  // !Regex.IsMatch(word, "[^" + range + "]")  (Build.cs:216): every char inside the class
  for (char16_t ch : word) {
    bool ok = false;
    for (size_t i = 0; i < range.size() && !ok; i++) {
      if (i + 2 < range.size() && range[i + 1] == u'-') { if (ch >= range[i] && ch <= range[i + 2]) ok = true; i += 2; }
      else if (ch == range[i]) ok = true;
    }
    if (!ok) return false;
  }
  return true;
}
ustr call_stem(stem_fn fn, const ustr& w) {
  if (!fn) return w;
  u16 buf[256];
  int n = fn((const u16*)w.data(), (int)w.size(), buf, 256);
  return ustr((const char16_t*)buf, (size_t)std::max(0, n));
}
ustr hex_upper(u32 v) { char b[16]; snprintf(b, sizeof b, "%X", v); ustr r; for (char* p = b; *p; p++) r.push_back((char16_t)*p); return r; }

// GetWordCodes, Build.cs:175-247.  Returns false for "null" (empty word).
bool GetWordCodes(const Index& ix, const ustr& word, std::vector<ustr>& l) {
  l.clear();
  if (word.empty()) return false;
  if (word[0] <= u'9' && word[0] >= u'0') { l.push_back(word); return true; }
  ustr stemmed = word;
  int nG = 0, nVoc = 0;
  if (ix.stopWords.count(word)) return true;  // empty array
  ustr firststemmed;
  l.push_back(word);
  for (const Vocab& voc : ix.vocs) {
    if (word[0] >= voc.range_begin && word[0] <= voc.range_end) {
      stemmed = call_stem(voc.stemmer, word);
      auto it = voc.words.find(stemmed);
      nG = it == voc.words.end() ? 0 : it->second;
      if (nG != 0) l.push_back(u"#" + hex_upper(((u32)nVoc << 24) | ((u32)nG & 0xFFFFFF)));
    }
    if (firststemmed.empty()) firststemmed = stemmed;
    nVoc++;
  }
  if (nG == 0) {
    stemmed = firststemmed;
    if (ix.vocs.empty()) {
      for (const Stemmer& st : ix.stemmers) {
        if (range_allows(st.range, word)) {
          if (st.fn) stemmed = call_stem(st.fn, word);
          break;
        }
      }
    }
    if (!stemmed.empty() && stemmed != word) l.push_back(u"$" + stemmed);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Index.Builder (Build.cs:258-437) + IndexTask page loop (Build.cs:445-564)
// ---------------------------------------------------------------------------------------------
struct Builder {
  Index* Parent;
  std::map<ustr, SeqBuilder> dict;                      // SortedDictionary<string, IndexSequence.Builder>
  std::vector<std::pair<ustr, u64>> pages;              // Build.cs:343
  u64 maxCoord = 0;                                     // Build.cs:294
  u64 TotalCount = 0;
  u64 coord = 0;                                        // IndexTask local, Build.cs:459
  int nTmpIndex = 0;
  long long MaxItems = (1ll << 62);                     // parity mode: no temp flush

  std::vector<std::vector<u8>> tmpind;                  // the N.tmpind files of Path, in creation order (Build.cs:370-386)

  void Add(const ustr& code, u64 c) {                   // Build.cs:324-341
    maxCoord = c;
    dict[code].Add(c);
    TotalCount++;
    if ((long long)TotalCount > MaxItems) {             // :335-340: Save(false); Clear(); TotalCount = 0
      SaveTmp();
      dict.clear();                                     // every term's Builder restarts with Last = 0
      TotalCount = 0;
    }
  }
  void SaveTmp();                                       // Builder.Save, Build.cs:370-386 (defined behind SaveIndex)
  void AddWord(const ustr& word, u64 c) {               // Build.cs:302-316 (suff is always null, :198,:237)
    std::vector<ustr> codes;
    if (GetWordCodes(*Parent, word, codes))
      for (const ustr& code : codes) Add(code, c);
  }
  void AddDoc(const ustr& sourceid, const ustr& name) { // Build.cs:348-356
    pages.emplace_back(sourceid + u":" + name, maxCoord);
  }
  void EndPage(const ustr& id, u64 maxcoord) {          // Build.cs:364-367
    pages.emplace_back(u":" + id, maxcoord);
  }
  void EndPage(const ustr& id) { EndPage(id, maxCoord); }  // Build.cs:359-362

  // Body of the per-page loop in IndexTask, Build.cs:477-541.
  void IndexPage(const ustr& id, const ustr& text) {
    ustr c = to_lower(text);                            // :482
    if (c.empty()) return;                              // :483
    if (id == u"0") {                                   // :485-524 header page
      // StringReader.ReadLine over the ORIGINAL text; each line lower-cased (:498)
      size_t p = 0, n = text.size();
      while (p < n) {                                   // ReadLine returns null at end of text
        size_t e = p;
        while (e < n && text[e] != u'\n' && text[e] != u'\r') e++;
        ustr line = to_lower(text.substr(p, e - p));
        p = e;
        if (p < n) { if (text[p] == u'\r' && p + 1 < n && text[p + 1] == u'\n') p += 2; else p += 1; }
        // fields = line.Trim('\n').Split('=')
        std::vector<ustr> fields;
        { size_t s = 0; while (true) { size_t q = line.find(u'=', s); if (q == ustr::npos) { fields.push_back(line.substr(s)); break; } fields.push_back(line.substr(s, q - s)); s = q + 1; } }
        if ((int)fields[0].size() >= Parent->MIN_WORD_LENGTH) {
          if (fields.size() < 2) return;                // fields[1] throws -> page aborted, no EndPage (:537-540)
          const ustr& v = fields[1];
          int dc = (int)fields[0].size() + 1;
          // Regex.Split(v, "\\b"): maximal runs of \w / non-\w characters (:502)
          size_t i = 0;
          while (i < v.size()) {
            size_t j = i; bool w = is_w(v[i]);
            while (j < v.size() && is_w(v[j]) == w) j++;
            if (w) {                                    // (match.Length>=1) && IsLetterOrDigit(match[0])  (:506)
              Add(u"&" + fields[0], (u64)(u32)(coord + (u64)(dc - 1)));   // (uint) cast, :509
              AddWord(to_lower(v.substr(i, j - i)), coord + (u64)dc);     // :510
            }
            dc += (int)(j - i);
            i = j;
          }
        }
        coord += (u64)line.size() + 1;                  // :515
      }
      EndPage(id, coord);                               // :521
      return;
    }
    // Regex.Matches(c, @"\p{L}+|\p{N}+"), length filter, :526-530
    size_t n = c.size(), i = 0;
    while (i < n) {
      if (is_L(c[i])) {
        size_t j = i; while (j < n && is_L(c[j])) j++;
        if ((int)(j - i) >= Parent->MIN_WORD_LENGTH && (int)(j - i) <= Parent->MAX_WORD_LENGTH) AddWord(c.substr(i, j - i), coord + (u32)i);
        i = j;
      } else if (is_N(c[i])) {
        size_t j = i; while (j < n && is_N(c[j]) && !is_L(c[j])) j++;
        if ((int)(j - i) >= Parent->MIN_WORD_LENGTH && (int)(j - i) <= Parent->MAX_WORD_LENGTH) AddWord(c.substr(i, j - i), coord + (u32)i);
        i = j;
      } else i++;
    }
    coord += (u32)c.size();                             // :531
    EndPage(id, coord);                                 // :532
  }

  // Builder.Save, Build.cs:370-386 -> bytes of "N.tmpind" == ".index" after Build()/single-task merge
  void SaveIndex(BinW& w) const {
    w.u64le(maxCoord);
    for (const auto& kv : dict) { w.str(kv.first); w.seq(kv.second.seq); }
  }
  // IndexPageList.AddFromList(pages, 0) + Save, Build.cs:53-72, 99-118 -> ".index.list"
  void SaveList(BinW& w) const {
    const ustr* curdoc = nullptr;   // AddFromList: a non-':' key opens a new Document
    const ustr* written = nullptr;  // Save: the doc name is written when the Document object changes
    static const ustr none;
    for (const auto& d : pages) {
      if (d.first.empty() || d.first[0] != u':') { curdoc = &d.first; continue; }
      w.u64le(d.second);
      if (curdoc != written || written == nullptr) {
        written = curdoc;
        w.str(curdoc ? *curdoc : none);
        w.u64le(d.second);
      }
      w.str(d.first);  // DOC_SEP + id
    }
  }
};

void Builder::SaveTmp() {                               // Build.cs:370-386: Interlocked.Increment(ref nTmpIndex); "{nTmpIndex}.tmpind"
  nTmpIndex++;
  BinW w; SaveIndex(w);
  tmpind.push_back(std::move(w.b));
}

// MergeFiles, Index.cs:577-689 -- literal: k-way merge of ".tmpind" images by ordinal key; equal keys are concatenated in FILE
// order through a fresh IndexSequence.Builder (AddRange re-codes every value, so the first delta of an appended input is
// taken against the previous input's last value); shift_coords: every input's sequence is Shift()ed by shifts[q] first.
// shifts[q] = shifts[q-1] + maxCoords[q-1] (:589-593) unless `shifts_override` is given (the multi-task quirk of SURVEY
// Appendix E.3: canonical parity uses text-length bases).  Output header = max of the inputs' MaxCoord (:596).
std::vector<u8> merge_files(const std::vector<std::pair<const u8*, size_t>>& files, bool shift_coords, const u64* shifts_override) {
  const size_t F = files.size();
  std::vector<BinR> bins; bins.reserve(F);
  std::vector<u64> maxCoords(F), shifts(F, 0);
  for (size_t q = 0; q < F; q++) {
    bins.emplace_back(files[q].first, files[q].second);
    maxCoords[q] = bins[q].u64le();
    if (q > 0) shifts[q] = shifts[q - 1] + maxCoords[q - 1];
  }
  if (shifts_override) for (size_t q = 0; q < F; q++) shifts[q] = shifts_override[q];
  BinW wr;
  u64 mx = 0; for (u64 m : maxCoords) mx = std::max(mx, m);
  wr.u64le(mx);
  std::vector<ustr> s(F, ustr(u" "));                  // next words
  std::vector<Seq> arr(F);
  std::vector<char> readnext(F, 1);
  auto s_end = [&]() { size_t n = 0; for (auto& x : s) n += !x.empty(); return n; };
  do {
    for (size_t q = 0; q < F; q++)
      if (readnext[q] && !s[q].empty()) {             // read next string & array if previous was read
        ustr k = bins[q].str();
        if (bins[q].eof) { s[q].clear(); continue; }   // EndOfStreamException -> s[q] = ""
        int32_t n = bins[q].i32le();
        if (bins[q].eof) { s[q].clear(); continue; }
        s[q] = k;
        BinR& r = bins[q];
        if (r.i + 2ull * (size_t)n > r.n) n = (int32_t)((r.n - r.i) / 2);
        arr[q].self.resize((size_t)n);
        memcpy(arr[q].self.data(), r.p + r.i, 2ull * (size_t)n);
        r.i += 2ull * (size_t)n;
      }
    std::vector<std::pair<const ustr*, size_t>> list;
    for (size_t q = 0; q < F; q++) if (!s[q].empty()) list.emplace_back(&s[q], q);
    std::stable_sort(list.begin(), list.end(), [&](const std::pair<const ustr*, size_t>& a, const std::pair<const ustr*, size_t>& b) {
      int i = a.first->compare(*b.first);               // String.CompareOrdinal (UTF-16 code units)
      if (i == 0) return shift_coords ? maxCoords[a.second] < maxCoords[b.second] : shifts[a.second] < shifts[b.second];
      return i < 0;
    });
    std::fill(readnext.begin(), readnext.end(), 0);
    if (!list.empty()) {
      readnext[list[0].second] = 1;
      for (size_t q = 1; q < list.size(); q++)
        if (*list[q].first == *list[q - 1].first) readnext[list[q].second] = 1;
      // QUIRK kept literal (Index.cs:655-660): list[q] is marked whenever it equals list[q-1], wherever q is -- with three or
      // more inputs whose heads are  a < b == b  the second "b" is appended under key "a" and lost for "b".  Merges of TWO
      // inputs are unaffected (the reference's default MaxDegreeOfParallelism = 2 and its tests never merge more); callers
      // that want a correct k-way result merge pairwise (see oracle.merge_pairwise).
      wr.str(s[list[0].second]);
      SeqBuilder bldr;
      for (size_t q = 0; q < F; q++)
        if (readnext[q]) {
          if (shift_coords) seq_shift(arr[q], shifts[q]);
          SeqEnum en(arr[q]);
          while (en.MoveNext()) bldr.Add(en.Current());  // AddRange, IndexSequence.cs:86-90
        }
      wr.seq(bldr.seq);
    }
  } while (s_end() > 0);
  return std::move(wr.b);
}

// MergeAll, Index.cs:537-573: while more than one ".tmpind" is left, merge groups of nMax = 5 consecutive files (creation
// order: SURVEY Appendix E.2) into the first file of the group.
std::vector<u8> merge_all(std::vector<std::vector<u8>> files) {
  while (files.size() > 1) {
    std::vector<std::vector<u8>> next;
    const size_t nMax = 5;
    size_t q = 0;
    for (; q + 1 < files.size(); q += nMax) {           // for (q = 0; q < files.Length - 1; q += nMax)
      std::vector<std::pair<const u8*, size_t>> grp;
      for (size_t k = q; k < std::min(q + nMax, files.size()); k++) grp.emplace_back(files[k].data(), files[k].size());
      next.push_back(merge_files(grp, false, nullptr));
    }
    for (; q < files.size(); q++) next.push_back(std::move(files[q]));   // (a last file left alone stays as it is)
    files = std::move(next);
  }
  return files.empty() ? std::vector<u8>() : std::move(files[0]);
}

// IndexPageList.Load (Build.cs:119-148) and Index.Load InMemory branch (Index.cs:312-343)
bool index_load(Index& ix, const u8* ib, size_t in, const u8* lb, size_t ln) {
  ix.self.clear(); ix.docs.clear(); ix.pages.clear();
  BinR r(ib, in);
  ix.MaxCoord = r.u64le();
  while (true) {
    ustr s = r.str();
    if (r.eof) break;
    int32_t n = r.i32le();
    if (r.eof) break;
    Seq q;
    if (r.i + 2ull * (size_t)n > r.n) n = (int32_t)((r.n - r.i) / 2);  // ReadBytes returns short; Read swallows (IndexSequence.cs:175-188)
    q.self.resize((size_t)n);
    memcpy(q.self.data(), r.p + r.i, 2ull * (size_t)n);
    r.i += 2ull * (size_t)n;
    ix.self[s] = q;
  }
  BinR l(lb, ln);
  int doc = -1;
  while (true) {
    u64 n = l.u64le();
    if (l.eof) break;
    ustr s = l.str();
    if (l.eof) break;
    if (s.empty() || s[0] != u':') { ix.docs.push_back(s); doc = (int)ix.docs.size() - 1; }
    else ix.pages.push_back(PageEntry{n, s.substr(1), doc, 0});
  }
  ix.prevc = 0xFFFFFFFFull;  // Load re-creates the enumerator but leaves prevc at its sentinel (Build.cs:146-147)
  ix.cur = 0;
  return true;
}

// IndexPageList.GetPage, Build.cs:74-97.  Returns page index or -1 for "new DocPage()".
// The enumerator position (_enum, prevc) is the only state a search changes in a loaded index.  `cs` = that state: the index's own
// (the reference: one cursor per IndexPageList object, every Search under one lock, Search.cs:448), or a thread's private one in the
// query-sharded CPU baseline (orc_query_batch_mt), where all threads read ONE set of posting lists and page entries.  (A page's
// base coordinate `coord` is written on arrival; every cursor writes the same value -- the end of the page before it.)
struct PageCursor { u64 prevc = 0xFFFFFFFFull; size_t cur = 0; };
long get_page(Index& ix, u64 coord, PageCursor* pc = nullptr) {
  u64& prevc = pc ? pc->prevc : ix.prevc;
  size_t& cur = pc ? pc->cur : ix.cur;
  auto key = [&](size_t c) -> u64 { return c < ix.pages.size() ? ix.pages[c].end : 0; };  // default(KeyValuePair) past the end
  if (prevc == 0xFFFFFFFFull) { cur = 0; prevc = 0; }
  else {
    if (key(cur) > coord && prevc <= coord) return cur < ix.pages.size() ? (long)cur : -1;
    if (key(cur) > coord) { cur = 0; prevc = 0; }
  }
  if (ix.pages.empty()) return -1;
  do {
    if (key(cur) > coord) { if (ix.pages[cur].coord != prevc) ix.pages[cur].coord = prevc; return (long)cur; }
    prevc = key(cur);
    if (cur < ix.pages.size()) cur++;
  } while (cur < ix.pages.size());
  return -1;
}

// ---------------------------------------------------------------------------------------------
// Search side
// ---------------------------------------------------------------------------------------------
ustr to_upper(const ustr& s) { ustr r(s); for (auto& c : r) c = g_upper[(u16)c]; return r; }

// GetLikeWords, Search.cs:160-167: unanchored regex with "_" -> ".*" over keys whose first char is a letter
bool like_match(const ustr& key, const ustr& pat) {
  size_t pos = 0, i = 0;
  while (i <= pat.size()) {
    size_t j = pat.find(u'_', i);
    if (j == ustr::npos) j = pat.size();
    ustr lit = pat.substr(i, j - i);
    if (!lit.empty()) {
      size_t f = key.find(lit, pos);
      if (f == ustr::npos) return false;
      pos = f + lit.size();
    }
    i = j + 1;
  }
  return true;
}

// Utils.Levenshtein, Index.cs:46-89 -- literal (full matrix)
int levenshtein(const ustr& s, const ustr& t) {
  const int n = (int)s.size(), m = (int)t.size();
  if (n == 0) return m;
  if (m == 0) return n;
  std::vector<std::vector<int>> d(n + 1, std::vector<int>(m + 1, 0));
  for (int i = 0; i <= n; i++) d[i][0] = i;
  for (int j = 0; j <= m; j++) d[0][j] = j;
  for (int i = 1; i <= n; i++)
    for (int j = 1; j <= m; j++) {
      int cost = (t[j - 1] == s[i - 1]) ? 0 : 1;
      d[i][j] = std::min(std::min(d[i - 1][j] + 1, d[i][j - 1] + 1), d[i - 1][j - 1] + cost);
    }
  return d[n][m];
}
// GetCloseWords, Search.cs:168-174: self.Keys.OrderBy(Levenshtein).Take(10) -- OrderBy is a STABLE sort over the dictionary order
std::vector<ustr> GetCloseWords(const Index& ix, const ustr& word, size_t n) {
  std::vector<std::pair<int, const ustr*>> v;
  for (const auto& kv : ix.self) v.emplace_back(levenshtein(kv.first, word), &kv.first);
  std::stable_sort(v.begin(), v.end(), [](const std::pair<int, const ustr*>& a, const std::pair<int, const ustr*>& b) { return a.first < b.first; });
  std::vector<ustr> r;
  for (size_t i = 0; i < v.size() && i < n; i++) r.push_back(*v[i].second);
  return r;
}
// the term selection of GetSuggessions, Search.cs:176-188: keys with Key[0] >= 'A' that start with lastword (ordinal) and are
// longer than it, OrderBy(-Value.Count) (stable; Count = ushort units), Select(Key.Substring(lastword.Length)), Take(n)
std::vector<ustr> SuggestTails(const Index& ix, const ustr& lastword, size_t n) {
  std::vector<std::pair<long long, const ustr*>> v;
  for (const auto& kv : ix.self) {
    const ustr& k = kv.first;
    if (k.empty() || k[0] < u'A') continue;
    if (k.size() <= lastword.size() || k.compare(0, lastword.size(), lastword) != 0) continue;
    v.emplace_back(-(long long)kv.second.self.size(), &k);
  }
  std::stable_sort(v.begin(), v.end(), [](const std::pair<long long, const ustr*>& a, const std::pair<long long, const ustr*>& b) { return a.first < b.first; });
  std::vector<ustr> r;
  for (size_t i = 0; i < v.size() && i < n; i++) r.push_back(v[i].second->substr(lastword.size()));
  return r;
}
// GetLikeWords, Search.cs:158-167 (the non-trivial branch: word contains '_', bKeepForms, Length >= 2)
std::vector<ustr> GetLikeWords(const Index& ix, const ustr& word) {
  std::vector<ustr> r;
  for (const auto& kv : ix.self)
    if (!kv.first.empty() && is_L(kv.first[0]) && like_match(kv.first, word)) {
      r.push_back(kv.first);
      if (r.size() >= 100) break;                              // MAX_LIKE_WORDS
    }
  return r;
}

// SearchWord, Search.cs:192-260
Seq SearchWord(Index& ix, ustr word) {
  bool bExact = false;
  if (to_upper(word) == word && ix.bKeepForms) bExact = true;  // :199
  word = to_lower(word);
  bool have_total = false;
  Seq total;
  std::vector<ustr> words{word};
  if (word.find(u'_') != ustr::npos) {                         // :206-217
    if (ix.bKeepForms) {
      bExact = true;
      words.clear();
      if (word.size() >= 2) {
        for (const auto& kv : ix.self) {
          if (!kv.first.empty() && is_L(kv.first[0]) && like_match(kv.first, word)) {
            words.push_back(kv.first);
            if (words.size() >= 100) break;                    // MAX_LIKE_WORDS, :158
          }
        }
      }
    } else return Seq();
  }
  for (const ustr& wword : words) {
    std::vector<ustr> codes;
    if (!GetWordCodes(ix, wword, codes)) continue;             // null -> exception -> caught per word (:251)
    if (codes.empty()) continue;
    std::vector<ustr> selfcodes, knowcodes;
    for (const ustr& s : codes) (is_w(s[0]) ? selfcodes : knowcodes).push_back(s);  // :230-231
    // Enumerable.Except yields distinct elements
    { std::vector<ustr> d; for (auto& k : knowcodes) if (std::find(d.begin(), d.end(), k) == d.end()) d.push_back(k); knowcodes.swap(d); }
    std::vector<ustr> use;
    if (!bExact && !knowcodes.empty()) use = knowcodes;
    else if (!selfcodes.empty()) use.push_back(selfcodes[0]);
    for (const ustr& code : use) {
      auto it = ix.self.find(code);
      if (it != ix.self.end()) {
        if (!have_total) { total = it->second; total.R = 0; have_total = true; }   // this[code] -> fresh IndexSequence, R = 0
        else { Seq rhs = it->second; rhs.R = 0; total = seq_add(total, rhs); }
      }
    }
  }
  if (bExact) total.R = -1;                                    // :258
  return total;
}

// ResultDocPage.rank, Search.cs:99-111
float page_rank(const std::vector<int32_t>& pos) {
  float bonus = 0;
  if (pos.size() > 1)
    for (size_t q = 1; q < pos.size(); q++) bonus += (float)(30 / std::max(5, pos[q] - pos[q - 1]));
  return 1 + bonus + (float)std::log((double)pos.size());
}

struct RPage { long page; std::vector<int32_t> pos; };
struct RDoc { int doc; std::vector<size_t> pages; bool found; float rank; };
struct Result {
  bool error = false;
  std::vector<RPage> foundPages;
  std::vector<RDoc> docs;            // every ResultDocument created, in order
  std::vector<size_t> foundDocs;     // indices into docs (HashSet by Name, insertion order), later rank-ordered
};

// PrepareSearchResult, Search.cs:365-420.  doc_mask[d] != 0 <=> the -filter: regexes accept doc d
// (filter.Length == 0 -> all ones); regex evaluation itself is host string work.
void PrepareSearchResult(Index& ix, const Seq& res, const u8* doc_mask, Result& result, PageCursor* pc = nullptr) {
  long prevp = -2; int prevd = -2;
  long lastPage = -1; long lastDoc = -1;
  SeqEnum en(res);
  while (en.MoveNext()) {
    u64 coord = en.Current();
    long _p = get_page(ix, coord, pc);
    if (_p < 0) { result.error = true; return; }               // null dereference in the reference
    if (_p != prevp) {
      result.foundPages.push_back(RPage{_p, {}});
      lastPage = (long)result.foundPages.size() - 1;
      result.foundPages[lastPage].pos.push_back((int32_t)(coord - ix.pages[_p].coord));
      if (prevd != ix.pages[_p].doc) {
        RDoc d{ix.pages[_p].doc, {}, false, 0};
        if ((int)result.foundDocs.size() < ix.MAX_FOUND_DOCS) {
          bool matched = doc_mask ? doc_mask[d.doc] != 0 : true;
          if (matched) {
            bool dup = false;   // HashSet<ResultDocument>: equality by Name
            for (size_t f : result.foundDocs) if (ix.docs[result.docs[f].doc] == ix.docs[d.doc]) { dup = true; break; }
            if (!dup) { d.found = true; result.foundDocs.push_back(result.docs.size()); }
          }
        }
        result.docs.push_back(d);
        lastDoc = (long)result.docs.size() - 1;
      }
      if (lastDoc < 0) { result.error = true; return; }
      result.docs[lastDoc].pages.push_back((size_t)lastPage);
      prevp = _p;
      prevd = ix.pages[_p].doc;
    } else {
      result.foundPages[lastPage].pos.push_back((int32_t)(coord - ix.pages[_p].coord));
    }
    if ((int)result.foundPages.size() > ix.MAX_FOUND_PAGES) break;
  }
}

// Doc rank + ordering, Search.cs:552-557, 599
void FinishRanks(Index& ix, Result& r) {
  for (size_t f : r.foundDocs) {
    RDoc& d = r.docs[f];
    double sum = 0;                                            // Enumerable.Sum(float selector): double accumulator
    for (size_t p : d.pages) sum += (double)page_rank(r.foundPages[p].pos);
    float rank = (float)sum;
    rank = 1 + (float)std::log((double)rank);
    if (!d.pages.empty() && ix.pages[r.foundPages[d.pages[0]].page].id == u"0") rank *= ix.DOC_RANK_MULTIPLY;
    d.rank = rank;
  }
  std::stable_sort(r.foundDocs.begin(), r.foundDocs.end(), [&](size_t a, size_t b) { return r.docs[a].rank < r.docs[b].rank; });
}

ustr mk(const u16* p, size_t n) { return ustr((const char16_t*)p, n); }

}  // namespace

// =============================================================================================
// C entry points (ctypes).  Strings are UTF-16LE code-unit arrays + length.
// =============================================================================================
extern "C" {

void orc_set_char_tables(const u8* cls, const u16* lower) {
  init_default_tables();
  memcpy(g_cls, cls, 65536);
  memcpy(g_lower, lower, 65536 * 2);
  rebuild_upper();
}

// ---- codec / algebra on raw unit arrays ------------------------------------------------------
size_t orc_seq_encode(const u64* v, size_t n, u16* out, size_t cap) {
  SeqBuilder b;
  for (size_t i = 0; i < n; i++) b.Add(v[i]);
  size_t m = b.seq.self.size();
  if (out && m <= cap) memcpy(out, b.seq.self.data(), m * 2);
  return m;
}
size_t orc_seq_decode(const u16* u, size_t nu, u64* out, size_t cap) {
  SeqEnum e(u, nu);
  size_t n = 0;
  while (e.MoveNext()) { if (out && n < cap) out[n] = e.Current(); n++; }
  return n;
}
size_t orc_seq_shift(const u16* u, size_t nu, u64 shift, u16* out, size_t cap) {
  Seq s; s.self.assign(u, u + nu);
  seq_shift(s, shift);
  if (out && s.self.size() <= cap) memcpy(out, s.self.data(), s.self.size() * 2);
  return s.self.size();
}
static size_t emit(const Seq& r, u16* out, size_t cap, int* R) {
  if (R) *R = r.R;
  if (out && r.self.size() <= cap && !r.self.empty()) memcpy(out, r.self.data(), r.self.size() * 2);
  return r.self.size();
}
size_t orc_seq_and(const u16* a, size_t na, int Ra, const u16* b, size_t nb, int Rb, u16* out, size_t cap, int* R) {
  Seq x, y; x.self.assign(a, a + na); x.R = Ra; y.self.assign(b, b + nb); y.R = Rb;
  return emit(seq_mul(x, y), out, cap, R);
}
size_t orc_seq_or(const u16* a, size_t na, int Ra, const u16* b, size_t nb, int Rb, u16* out, size_t cap, int* R) {
  Seq x, y; x.self.assign(a, a + na); x.R = Ra; y.self.assign(b, b + nb); y.R = Rb;
  return emit(seq_add(x, y), out, cap, R);
}

// ---- Index object ------------------------------------------------------------------------------
void* orc_index_new() { init_default_tables(); return new Index(); }
void orc_index_free(void* h) { delete (Index*)h; }
void orc_index_set_limits(void* h, int min_len, int max_len, int max_pages, int max_docs, int keep_forms) {
  Index* ix = (Index*)h; ix->MIN_WORD_LENGTH = min_len; ix->MAX_WORD_LENGTH = max_len;
  ix->MAX_FOUND_PAGES = max_pages; ix->MAX_FOUND_DOCS = max_docs; ix->bKeepForms = keep_forms != 0;
}
void orc_index_add_stop_word(void* h, const u16* w, size_t n) { ((Index*)h)->stopWords.insert(mk(w, n)); }
int orc_index_add_voc(void* h, u16 rb, u16 re, stem_fn fn) { Index* ix = (Index*)h; Vocab v; v.range_begin = rb; v.range_end = re; v.stemmer = fn; ix->vocs.push_back(v); return (int)ix->vocs.size() - 1; }
void orc_index_voc_add(void* h, int voc, const u16* w, size_t n, int group) { ((Index*)h)->vocs[voc].words[mk(w, n)] = group; }
void orc_index_add_stemmer(void* h, const u16* range, size_t n, stem_fn fn) { ((Index*)h)->stemmers.push_back(Stemmer{mk(range, n), fn}); }
void orc_index_clear_stemmers(void* h) { ((Index*)h)->stemmers.clear(); }

int orc_index_load(void* h, const u8* ib, size_t in, const u8* lb, size_t ln) { return index_load(*(Index*)h, ib, in, lb, ln) ? 0 : -1; }
size_t orc_index_count(void* h) { return ((Index*)h)->self.size(); }
u64 orc_index_max_coord(void* h) { return ((Index*)h)->MaxCoord; }
size_t orc_index_n_pages(void* h) { return ((Index*)h)->pages.size(); }
size_t orc_index_n_docs(void* h) { return ((Index*)h)->docs.size(); }
// key #i in ordinal order -> copies UTF-16 units, returns length
size_t orc_index_key(void* h, size_t i, u16* out, size_t cap) {
  Index* ix = (Index*)h; auto it = ix->self.begin(); std::advance(it, (long)i);
  size_t n = it->first.size(); if (out && n <= cap) memcpy(out, it->first.data(), n * 2); return n;
}
// this[key]; returns unit count or (size_t)-1 if absent
size_t orc_index_get(void* h, const u16* k, size_t kn, u16* out, size_t cap) {
  Index* ix = (Index*)h; auto it = ix->self.find(mk(k, kn));
  if (it == ix->self.end()) return (size_t)-1;
  size_t n = it->second.self.size(); if (out && n <= cap && n) memcpy(out, it->second.self.data(), n * 2); return n;
}
size_t orc_index_doc_name(void* h, size_t d, u16* out, size_t cap) { Index* ix = (Index*)h; size_t n = ix->docs[d].size(); if (out && n <= cap) memcpy(out, ix->docs[d].data(), n * 2); return n; }
size_t orc_index_page_id(void* h, size_t p, u16* out, size_t cap) { Index* ix = (Index*)h; size_t n = ix->pages[p].id.size(); if (out && n <= cap) memcpy(out, ix->pages[p].id.data(), n * 2); return n; }
u64 orc_index_page_end(void* h, size_t p) { return ((Index*)h)->pages[p].end; }
int orc_index_page_doc(void* h, size_t p) { return ((Index*)h)->pages[p].doc; }

// GetWordCodes -> codes joined by U+0000; returns total length or (size_t)-1 for null
size_t orc_get_word_codes(void* h, const u16* w, size_t n, u16* out, size_t cap) {
  std::vector<ustr> l;
  if (!GetWordCodes(*(Index*)h, mk(w, n), l)) return (size_t)-1;
  ustr j; for (size_t i = 0; i < l.size(); i++) { if (i) j.push_back(0); j += l[i]; }
  if (out && j.size() <= cap && !j.empty()) memcpy(out, j.data(), j.size() * 2);
  return l.empty() ? 0 : j.size();
}

// SearchWord (Search.cs:192-260); R returned through *R
size_t orc_search_word(void* h, const u16* w, size_t n, u16* out, size_t cap, int* R) {
  Seq s = SearchWord(*(Index*)h, mk(w, n));
  return emit(s, out, cap, R);
}

// dictionary scans: results joined by U+0000; returns the total length (call with out == NULL for the size)
static size_t join0(const std::vector<ustr>& v, u16* out, size_t cap) {
  ustr j;
  for (size_t i = 0; i < v.size(); i++) { if (i) j.push_back(u'\0'); j += v[i]; }
  if (out && cap >= j.size()) memcpy(out, j.data(), j.size() * 2);
  return j.size() + (v.empty() ? 0 : 0);
}
size_t orc_close_words(void* h, const u16* w, size_t n, size_t take, u16* out, size_t cap, size_t* count) {
  auto v = GetCloseWords(*(Index*)h, mk(w, n), take); *count = v.size(); return join0(v, out, cap);
}
size_t orc_suggest(void* h, const u16* w, size_t n, size_t take, u16* out, size_t cap, size_t* count) {
  auto v = SuggestTails(*(Index*)h, mk(w, n), take); *count = v.size(); return join0(v, out, cap);
}
size_t orc_like_words(void* h, const u16* w, size_t n, u16* out, size_t cap, size_t* count) {
  auto v = GetLikeWords(*(Index*)h, mk(w, n)); *count = v.size(); return join0(v, out, cap);
}

// ---- Builder ---------------------------------------------------------------------------------
void* orc_builder_new(void* index) { Builder* b = new Builder(); b->Parent = (Index*)index; return b; }
void orc_builder_free(void* b) { delete (Builder*)b; }
void orc_builder_add_doc(void* b, const u16* src, size_t sn, const u16* name, size_t nn) { ((Builder*)b)->AddDoc(mk(src, sn), mk(name, nn)); }
void orc_builder_add_word(void* b, const u16* w, size_t n, u64 coord) { ((Builder*)b)->AddWord(mk(w, n), coord); }
void orc_builder_add(void* b, const u16* code, size_t n, u64 coord) { ((Builder*)b)->Add(mk(code, n), coord); }
void orc_builder_end_page(void* b, const u16* id, size_t n) { ((Builder*)b)->EndPage(mk(id, n)); }
void orc_builder_index_page(void* b, const u16* id, size_t idn, const u16* text, size_t n) { ((Builder*)b)->IndexPage(mk(id, idn), mk(text, n)); }
// Bulk path for the CPU baseline: n_pages single-byte (Latin-1 == code unit) pages, all with the same id,
// one document per page named by its decimal index, source name `src`.
void orc_builder_index_docs_latin1(void* bp, const u8* text, const u64* off, size_t n_pages, u64 first_doc,
                                   const u16* src, size_t sn, const u16* id, size_t idn) {
  Builder* b = (Builder*)bp;
  ustr s = mk(src, sn), pid = mk(id, idn), page;
  for (size_t p = 0; p < n_pages; p++) {
    char num[32]; snprintf(num, sizeof num, "%llu", (unsigned long long)(first_doc + p));
    ustr name; for (char* q = num; *q; q++) name.push_back((char16_t)*q);
    b->AddDoc(s, name);
    page.assign((size_t)(off[p + 1] - off[p]), u' ');
    for (size_t i = 0; i < page.size(); i++) page[i] = (char16_t)text[off[p] + i];
    b->IndexPage(pid, page);
  }
}
u64 orc_builder_total_count(void* b) { return ((Builder*)b)->TotalCount; }
u64 orc_builder_max_coord(void* b) { return ((Builder*)b)->maxCoord; }
u64 orc_builder_coord(void* b) { return ((Builder*)b)->coord; }
size_t orc_builder_n_terms(void* b) { return ((Builder*)b)->dict.size(); }
// Serialise: returns malloc'd buffers the caller frees with orc_free
void orc_builder_save(void* bp, u8** ib, size_t* in, u8** lb, size_t* ln) {
  Builder* b = (Builder*)bp;
  BinW wi, wl; b->SaveIndex(wi); b->SaveList(wl);
  *in = wi.b.size(); *ib = (u8*)malloc(wi.b.size() + 1); memcpy(*ib, wi.b.data(), wi.b.size());
  *ln = wl.b.size(); *lb = (u8*)malloc(wl.b.size() + 1); memcpy(*lb, wl.b.data(), wl.b.size());
}
void orc_free(void* p) { free(p); }
// Index.MaxTmpIndexItems (Index.cs:120): the builder flushes a temp index whenever TotalCount exceeds it (Build.cs:335-340)
void orc_builder_set_max_items(void* b, long long n) { ((Builder*)b)->MaxItems = n; }
size_t orc_builder_n_tmp(void* b) { return ((Builder*)b)->tmpind.size(); }
// bytes of temp index #i (library-owned; valid until the builder is freed)
const u8* orc_builder_tmp(void* b, size_t i, size_t* n) { auto& v = ((Builder*)b)->tmpind[i]; *n = v.size(); return v.data(); }
// What CreateAsync leaves for ONE index task that flushed temp indexes (Build.cs:541 Save -> Index.cs:488 MergeAll): the
// remaining postings are saved as the last N.tmpind, then MergeAll.  Returns the merged ".index" bytes (malloc'd).
void orc_builder_save_merged(void* bp, u8** ib, size_t* in) {
  Builder* b = (Builder*)bp;
  std::vector<std::vector<u8>> files = b->tmpind;
  { BinW w; b->SaveIndex(w); files.push_back(std::move(w.b)); }
  std::vector<u8> out = merge_all(std::move(files));
  *in = out.size(); *ib = (u8*)malloc(out.size() + 1); memcpy(*ib, out.data(), out.size());
}
// MergeFiles over n images (Index.cs:577-689); shifts == NULL: the reference's own shifts (running sum of MaxCoord)
void orc_merge_files(const u8* const* imgs, const size_t* sizes, size_t n, int shift_coords, const u64* shifts, u8** ob, size_t* on) {
  std::vector<std::pair<const u8*, size_t>> files;
  for (size_t q = 0; q < n; q++) files.emplace_back(imgs[q], sizes[q]);
  std::vector<u8> out = merge_files(files, shift_coords != 0, shifts);
  *on = out.size(); *ob = (u8*)malloc(out.size() + 1); memcpy(*ob, out.data(), out.size());
}

// ---- Results ---------------------------------------------------------------------------------
// PrepareSearchResult + doc ranks for one result sequence.  Returns an opaque Result*.
void* orc_prepare_result(void* h, const u16* units, size_t nu, const u8* doc_mask) {
  Index* ix = (Index*)h;
  Seq s; s.self.assign(units, units + nu);
  Result* r = new Result();
  PrepareSearchResult(*ix, s, doc_mask, *r);
  return r;
}
// CombineSearchResults (Search.cs:423-428): keep r1.foundDocs whose Name is in r2.foundDocs
void orc_result_intersect_docs(void* h, void* r1p, void* r2p) {
  Index* ix = (Index*)h; Result* r1 = (Result*)r1p; Result* r2 = (Result*)r2p;
  std::vector<size_t> keep;
  for (size_t f : r1->foundDocs) {
    bool in = false;
    for (size_t g : r2->foundDocs) if (ix->docs[r2->docs[g].doc] == ix->docs[r1->docs[f].doc]) { in = true; break; }
    if (in) keep.push_back(f);
  }
  r1->foundDocs.swap(keep);
}
void orc_result_finish(void* h, void* r) { FinishRanks(*(Index*)h, *(Result*)r); }
void orc_result_free(void* r) { delete (Result*)r; }
int orc_result_error(void* r) { return ((Result*)r)->error ? 1 : 0; }
size_t orc_result_n_pages(void* r) { return ((Result*)r)->foundPages.size(); }
long orc_result_page_index(void* r, size_t i) { return ((Result*)r)->foundPages[i].page; }
size_t orc_result_page_npos(void* r, size_t i) { return ((Result*)r)->foundPages[i].pos.size(); }
void orc_result_page_pos(void* r, size_t i, int32_t* out) { auto& v = ((Result*)r)->foundPages[i].pos; memcpy(out, v.data(), v.size() * 4); }
float orc_result_page_rank(void* r, size_t i) { return page_rank(((Result*)r)->foundPages[i].pos); }
size_t orc_result_n_found_docs(void* r) { return ((Result*)r)->foundDocs.size(); }
int orc_result_found_doc(void* r, size_t i) { Result* R = (Result*)r; return R->docs[R->foundDocs[i]].doc; }
float orc_result_found_doc_rank(void* r, size_t i) { Result* R = (Result*)r; return R->docs[R->foundDocs[i]].rank; }
size_t orc_result_found_doc_npages(void* r, size_t i) { Result* R = (Result*)r; return R->docs[R->foundDocs[i]].pages.size(); }

// ---- Batched plan evaluation for the CPU baseline (single-threaded, like lock(DoSearchLock)) ---
// A plan is a postfix program of int32 words: [0, termIndex, R] = push this[key(termIndex)] with R;
// [1] = operator *; [2] = operator +.  Terms are addressed by ordinal key index.  Each query is then
// run through Count() per term (Search.cs:307), the operators, PrepareSearchResult and the ranks.
// Outputs per query: number of found pages, number of positions, number of found docs, and an
// FNV-1a checksum over (page index, positions, page rank bits, doc index, doc rank bits).
static void query_batch_range(Index* ix, const std::vector<const Seq*>& keys, const int32_t* prog, const u64* prog_off, size_t q0, size_t q1,
                              u64* n_pages, u64* n_pos, u64* n_docs, u64* checksum, PageCursor* pc) {
  for (size_t q = q0; q < q1; q++) {
    std::vector<Seq> st;
    for (u64 i = prog_off[q]; i < prog_off[q + 1];) {
      int op = prog[i];
      if (op == 0) { Seq s = *keys[(size_t)prog[i + 1]]; s.R = prog[i + 2]; (void)seq_count(s); st.push_back(std::move(s)); i += 3; }
      else { Seq b = std::move(st.back()); st.pop_back(); Seq a = std::move(st.back()); st.pop_back(); st.push_back(op == 1 ? seq_mul(a, b) : seq_add(a, b)); i += 1; }
    }
    Result r;
    PrepareSearchResult(*ix, st.back(), nullptr, r, pc);
    FinishRanks(*ix, r);
    u64 hsh = 1469598103934665603ull, np = 0;
    auto mix = [&](u64 v) { for (int k = 0; k < 8; k++) { hsh ^= (v >> (8 * k)) & 0xFF; hsh *= 1099511628211ull; } };
    for (auto& p : r.foundPages) {
      mix((u64)p.page); np += p.pos.size();
      for (int32_t x : p.pos) mix((u64)(u32)x);
      float pr = page_rank(p.pos); u32 bits; memcpy(&bits, &pr, 4); mix(bits);
    }
    for (size_t f : r.foundDocs) { mix((u64)r.docs[f].doc); u32 bits; memcpy(&bits, &r.docs[f].rank, 4); mix(bits); }
    if (n_pages) n_pages[q] = r.foundPages.size();
    if (n_pos) n_pos[q] = np;
    if (n_docs) n_docs[q] = r.foundDocs.size();
    if (checksum) checksum[q] = r.error ? ~0ull : hsh;
  }
}
void orc_query_batch(void* h, const int32_t* prog, const u64* prog_off, size_t nq,
                     u64* n_pages, u64* n_pos, u64* n_docs, u64* checksum) {
  Index* ix = (Index*)h;
  std::vector<const Seq*> keys; keys.reserve(ix->self.size());
  for (auto& kv : ix->self) keys.push_back(&kv.second);
  query_batch_range(ix, keys, prog, prog_off, 0, nq, n_pages, n_pos, n_docs, checksum, nullptr);
}
// The query-sharded baseline on ALL host cores: n_threads threads evaluate contiguous slices of the batch on ONE loaded
// index -- the posting lists and the page table are only read, each thread has its own page cursor (a private IndexPageList
// enumerator: what a private copy of the index per thread would give, without the copies).
void orc_query_batch_mt(void* h, const int32_t* prog, const u64* prog_off, size_t nq, int n_threads,
                        u64* n_pages, u64* n_pos, u64* n_docs, u64* checksum) {
  Index* ix = (Index*)h;
  std::vector<const Seq*> keys; keys.reserve(ix->self.size());
  for (auto& kv : ix->self) keys.push_back(&kv.second);
  if (n_threads < 1) n_threads = 1;
  std::vector<std::thread> ts;
  for (int t = 0; t < n_threads; t++) {
    const size_t a = nq * (size_t)t / (size_t)n_threads, b = nq * (size_t)(t + 1) / (size_t)n_threads;
    ts.emplace_back([=, &keys] {
      PageCursor pc;
      query_batch_range(ix, keys, prog, prog_off, a, b, n_pages, n_pos, n_docs, checksum, &pc);
    });
  }
  for (auto& t : ts) t.join();
}

}  // extern "C"

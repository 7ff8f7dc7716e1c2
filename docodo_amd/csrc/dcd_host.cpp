// dcd_host.cpp -- host-side format code: ".index.list" writer/reader (IndexPageList.Save/Load,
// Build.cs:99-148), BinaryWriter string helpers, UTF-8 <-> UTF-16.
#include <cstdio>
#include <memory>

#include "dcd_build_int.h"
#include "../../include/docodo_hip.h"

namespace dcd {

void utf16_to_utf8(const u16* s, size_t n, std::string& out) {
  for (size_t i = 0; i < n; i++) {
    u32 c = s[i];
    if (c >= 0xD800 && c <= 0xDBFF && i + 1 < n && s[i + 1] >= 0xDC00 && s[i + 1] <= 0xDFFF) {
      c = 0x10000 + ((c - 0xD800) << 10) + (s[i + 1] - 0xDC00);
      i++;
    } else if (c >= 0xD800 && c <= 0xDFFF) {
      c = 0xFFFD;  // UTF8Encoding replacement fallback for a lone surrogate
    }
    if (c < 0x80) out.push_back((char)c);
    else if (c < 0x800) { out.push_back((char)(0xC0 | (c >> 6))); out.push_back((char)(0x80 | (c & 0x3F))); }
    else if (c < 0x10000) {
      out.push_back((char)(0xE0 | (c >> 12)));
      out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (c & 0x3F)));
    } else {
      out.push_back((char)(0xF0 | (c >> 18)));
      out.push_back((char)(0x80 | ((c >> 12) & 0x3F)));
      out.push_back((char)(0x80 | ((c >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (c & 0x3F)));
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
    if (c >= 0x10000) {
      c -= 0x10000;
      out.push_back((char16_t)(0xD800 + (c >> 10)));
      out.push_back((char16_t)(0xDC00 + (c & 0x3FF)));
    } else out.push_back((char16_t)c);
  }
  return true;
}

ustr PageTable::doc_name(u64 d) const {
  if (!doc_per_page) return docs[d].name;
  char num[32];
  snprintf(num, sizeof num, "%llu", (unsigned long long)(first_doc + d));
  ustr r = source;
  r.push_back(u':');
  for (char* q = num; *q; q++) r.push_back((char16_t)*q);
  return r;
}

static void put_u64(std::vector<u8>& b, u64 v) { for (int i = 0; i < 8; i++) b.push_back((u8)(v >> (8 * i))); }
static void put_str(std::vector<u8>& b, const ustr& s) {  // BinaryWriter.Write(string)
  std::string u;
  utf16_to_utf8((const u16*)s.data(), s.size(), u);
  u32 n = (u32)u.size();
  while (n >= 0x80) { b.push_back((u8)(n | 0x80)); n >>= 7; }
  b.push_back((u8)n);
  b.insert(b.end(), u.begin(), u.end());
}

// IndexPageList.Save (Build.cs:99-118): per page u64 end; when the document changes: string doc name,
// u64 end again; then string ":" + page id.
void write_list_image(const PageTable& pt, std::vector<u8>& out) {
  out.clear();
  u64 np = pt.n_pages();
  out.reserve((size_t)np * 32);
  int64_t last_doc = -1;
  ustr sep(1, u':');
  for (u64 p = 0; p < np; p++) {
    put_u64(out, pt.end_at(p));
    int64_t d = (int64_t)pt.doc_of_page(p);
    if (d != last_doc) {
      last_doc = d;
      put_str(out, pt.doc_name((u64)d));
      put_u64(out, pt.end_at(p));
    }
    put_str(out, sep + pt.page_id(p));
  }
}

}  // namespace dcd

// ---------------------------------------------------------------------------------------------------------------------
// Out-of-core tail of a build whose index does not fit the device: every hash partition was merged on the device and saved
// as a sorted ".index" run; the final ".index" is the k-way merge of the runs by key.  It is MergeFiles (Index.cs:577-689)
// for inputs that never share a key -- records are copied, nothing is decoded -- streamed through bounded buffers.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct RunReader {
  FILE* f = nullptr;
  dcd::ustr key;          // current record's key (UTF-16, the ordinal compare of String.CompareOrdinal)
  std::string key_bytes;  // its length prefix + UTF-8 bytes, verbatim
  uint32_t n_units = 0;
  bool have = false;
  ~RunReader() { if (f) fclose(f); }
  // reads the head of the next record (key + count); the payload stays in the file
  int32_t next() {
    have = false;
    int c = fgetc(f);
    if (c == EOF) return DCD_OK;
    key_bytes.clear();
    uint32_t len = 0;
    int sh = 0;
    for (;;) {
      if (c == EOF) return DCD_E_IO;
      if (sh > 28) return DCD_E_IO;
      key_bytes.push_back((char)c);
      len |= (uint32_t)(c & 0x7F) << sh;
      sh += 7;
      if (!(c & 0x80)) break;
      c = fgetc(f);
    }
    if (len > (1u << 20)) return DCD_E_IO;              // (keys are <= 3 * 63 UTF-8 bytes: a damaged run, not a 4 GiB allocation)
    std::string kb(len, '\0');
    if (len && fread(&kb[0], 1, len, f) != len) return DCD_E_IO;
    key_bytes += kb;
    key.clear();
    dcd::utf8_to_utf16((const u8*)kb.data(), kb.size(), key);
    unsigned char nb[4];
    if (fread(nb, 1, 4, f) != 4) return DCD_E_IO;
    n_units = (uint32_t)nb[0] | ((uint32_t)nb[1] << 8) | ((uint32_t)nb[2] << 16) | ((uint32_t)nb[3] << 24);
    if ((int32_t)n_units < 0) return DCD_E_IO;
    have = true;
    return DCD_OK;
  }
};
}  // namespace

static int32_t merge_index_files_impl(const char* const* paths_utf8, size_t n, const char* out_path_utf8);
// The output is written as "<out>.tmp" and renamed on success: a failed merge leaves no file that looks like a valid, shorter
// index.  No C++ exception crosses the C ABI (std::bad_alloc from a damaged run would take a C# host down).
extern "C" int32_t dcd_merge_index_files(const char* const* paths_utf8, size_t n, const char* out_path_utf8) {
  using namespace dcd;
  if ((!paths_utf8 && n) || !out_path_utf8) DCD_FAIL(DCD_E_BADARG, "dcd_merge_index_files: NULL argument");
  const std::string tmp = std::string(out_path_utf8) + ".tmp";
  int32_t st;
  try {
    st = merge_index_files_impl(paths_utf8, n, tmp.c_str());
  } catch (const std::bad_alloc&) {
    set_error("dcd_merge_index_files: out of host memory");
    st = DCD_E_NOMEM;
  } catch (...) {
    set_error("dcd_merge_index_files: unexpected exception");
    st = DCD_E_IO;
  }
  if (st == DCD_OK && rename(tmp.c_str(), out_path_utf8) != 0) { set_error("cannot rename %s to %s", tmp.c_str(), out_path_utf8); st = DCD_E_IO; }
  if (st != DCD_OK) (void)remove(tmp.c_str());
  return st;
}
static int32_t merge_index_files_impl(const char* const* paths_utf8, size_t n, const char* out_path_utf8) {
  using namespace dcd;
  std::vector<std::unique_ptr<RunReader>> runs;
  u64 max_coord = 0;
  for (size_t q = 0; q < n; q++) {
    std::unique_ptr<RunReader> r(new RunReader());
    r->f = fopen(paths_utf8[q], "rb");
    if (!r->f) DCD_FAIL(DCD_E_IO, "cannot open %s", paths_utf8[q]);
    unsigned char h[8];
    if (fread(h, 1, 8, r->f) != 8) DCD_FAIL(DCD_E_IO, "%s: shorter than its MaxCoord header", paths_utf8[q]);
    u64 mc = 0;
    for (int i = 0; i < 8; i++) mc |= (u64)h[i] << (8 * i);
    max_coord = std::max(max_coord, mc);                       // wr.Write(maxCoords.Max()), Index.cs:596
    if (r->next() != DCD_OK) DCD_FAIL(DCD_E_IO, "%s: malformed record", paths_utf8[q]);
    runs.push_back(std::move(r));
  }
  FILE* out = fopen(out_path_utf8, "wb");
  if (!out) DCD_FAIL(DCD_E_IO, "cannot create %s", out_path_utf8);
  struct Closer { FILE* f; ~Closer() { if (f) fclose(f); } } closer{out};
  unsigned char h[8];
  for (int i = 0; i < 8; i++) h[i] = (unsigned char)(max_coord >> (8 * i));
  if (fwrite(h, 1, 8, out) != 8) DCD_FAIL(DCD_E_IO, "write to %s failed", out_path_utf8);
  std::vector<char> buf(1 << 20);
  ustr prev;
  bool first = true;
  for (;;) {
    int best = -1;
    for (size_t q = 0; q < runs.size(); q++) {
      if (!runs[q]->have) continue;
      if (best < 0 || runs[q]->key < runs[best]->key) best = (int)q;
      else if (runs[q]->key == runs[best]->key)
        DCD_FAIL(DCD_E_BADARG, "dcd_merge_index_files: a key occurs in two inputs (the inputs must be disjoint partitions)");
    }
    if (best < 0) break;
    RunReader& r = *runs[best];
    if (!first && !(prev < r.key)) DCD_FAIL(DCD_E_IO, "%s: keys are not strictly ascending", paths_utf8[best]);
    first = false;
    prev = r.key;
    unsigned char nb[4] = {(unsigned char)r.n_units, (unsigned char)(r.n_units >> 8), (unsigned char)(r.n_units >> 16), (unsigned char)(r.n_units >> 24)};
    if (fwrite(r.key_bytes.data(), 1, r.key_bytes.size(), out) != r.key_bytes.size() || fwrite(nb, 1, 4, out) != 4)
      DCD_FAIL(DCD_E_IO, "write to %s failed", out_path_utf8);
    u64 left = 2ull * r.n_units;
    while (left) {
      const size_t k = (size_t)std::min<u64>(left, buf.size());
      if (fread(buf.data(), 1, k, r.f) != k) DCD_FAIL(DCD_E_IO, "%s: truncated posting list", paths_utf8[best]);
      if (fwrite(buf.data(), 1, k, out) != k) DCD_FAIL(DCD_E_IO, "write to %s failed", out_path_utf8);
      left -= k;
    }
    if (r.next() != DCD_OK) DCD_FAIL(DCD_E_IO, "%s: malformed record", paths_utf8[best]);
  }
  closer.f = nullptr;
  if (fclose(out) != 0) DCD_FAIL(DCD_E_IO, "closing %s failed", out_path_utf8);
  return DCD_OK;
}

// dcd_host.cpp -- host-side format code: ".index.list" writer/reader (IndexPageList.Save/Load,
// Build.cs:99-148), BinaryWriter string helpers, UTF-8 <-> UTF-16.
#include "dcd_build_int.h"

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

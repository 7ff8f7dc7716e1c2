// dcd_unicode.cpp -- expands the generated BMP class / lower-case tables and keeps one device copy
// per GPU for the UTF-16 tokeniser.
#include <mutex>

#include "dcd_common.h"
#include "dcd_tokenize.h"
#include "dcd_unicode_data.h"

namespace dcd {

void host_char_tables16(const u8** cls, const u16** lower) {
  static std::vector<u8> c;
  static std::vector<u16> l;
  static std::once_flag once;
  std::call_once(once, []() {
    c.assign(65536, 0);
    l.resize(65536);
    for (u32 i = 0; i < 65536; i++) l[i] = (u16)i;
    // bit0 = \\p{L}, bit1 = \\p{N}, bit2 = \\w (L | Mn | Nd | Pc, as .NET Regex defines it)
    for (auto& r : kLetterRanges) for (u32 i = r[0]; i <= r[1]; i++) c[i] = 1 | 4;
    for (auto& r : kNumberRanges) for (u32 i = r[0]; i <= r[1]; i++) c[i] |= 2;
    for (auto& r : kWordExtraRanges) for (u32 i = r[0]; i <= r[1]; i++) c[i] |= 4;
    for (auto& p : kLowerPairs) l[p[0]] = p[1];
  });
  *cls = c.data();
  *lower = l.data();
}

int32_t get_char_tables16(int device, CharTables16* out) {
  static std::mutex mu;
  static CharTables16 cache[64];
  std::lock_guard<std::mutex> g(mu);
  CharTables16& t = cache[device & 63];
  if (!t.cls) {
    const u8* c; const u16* l;
    host_char_tables16(&c, &l);
    void *dc = nullptr, *dl = nullptr;
    DCD_HIP(hipSetDevice(device));
    DCD_HIP(hipMalloc(&dc, 65536));
    DCD_HIP(hipMalloc(&dl, 65536 * 2));
    DCD_HIP(hipMemcpy(dc, c, 65536, hipMemcpyHostToDevice));
    DCD_HIP(hipMemcpy(dl, l, 65536 * 2, hipMemcpyHostToDevice));
    t.cls = (const u8*)dc;
    t.lower = (const u16*)dl;
  }
  *out = t;
  return DCD_OK;
}

}  // namespace dcd

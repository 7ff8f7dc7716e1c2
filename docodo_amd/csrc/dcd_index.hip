// dcd_index.hip -- the loaded index: Index.Load (Index.cs:312-380), dictionary lookups
// (SortedList keys, Index.cs:151, 235), this[key] (Index.cs:255-263), page table accessors
// (IndexPageList, Build.cs:41-149).  Query evaluation lives in dcd_query.hip.
#include <algorithm>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "dcd_index_int.h"
#include "dcd_scan.h"
#include "dcd_sort.h"
#include "dcd_tokenize.h"

namespace dcd {
DeviceIndex* build_take_index(dcd_build* b);
DeviceIndex* build_peek_index(dcd_build* b);
int build_device(dcd_build* b);
hipStream_t build_stream(dcd_build* b);
}  // namespace dcd

namespace dcd {
// dictionary scan: Levenshtein distance (Index.cs:46-95) of every key to one word, one thread per key; the word sits in
// constant kernel arguments (<= 32 units), the DP row in registers / local memory.  Output pairs (distance, term) are
// stably sorted by distance afterwards (OrderBy(Levenshtein).Take(10), Search.cs:172).
struct LevWord { u16 c[32]; u32 len; };
__global__ void k_levenshtein(const u16* __restrict__ chars, const u32* __restrict__ str_off, u64 n_terms, LevWord w, uint2* __restrict__ out) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const u16* s = chars + str_off[t];
  const u32 n = str_off[t + 1] - str_off[t], m = w.len;
  u32 d = 0;
  if (n == 0) d = m;
  else if (m == 0) d = n;
  else {
    u32 row[33];
    for (u32 j = 0; j <= m; j++) row[j] = j;
    for (u32 i = 1; i <= n; i++) {
      u32 diag = row[0];
      row[0] = i;
      const u16 si = s[i - 1];
      for (u32 j = 1; j <= m; j++) {
        const u32 cost = (w.c[j - 1] == si) ? 0u : 1u;
        const u32 v = min(min(row[j] + 1u, row[j - 1] + 1u), diag + cost);
        diag = row[j];
        row[j] = v;
      }
    }
    d = row[m];
  }
  out[t] = make_uint2(d, (u32)t);
}
// dictionary scan of GetLikeWords (Search.cs:160-167): keys whose first unit is a letter and that match the pattern as an
// UNANCHORED regex in which every '_' stands for ".*" -- i.e. the pattern's literal segments occur in the key in order, anywhere.
// One thread per key, earliest-match scan (sufficient for an existence test); the pattern sits in constant kernel arguments.
struct LikePat { u16 c[64]; u32 len; };
__global__ void k_like_words(const u16* __restrict__ chars, const u32* __restrict__ str_off, u64 n_terms, LikePat pat, const u8* __restrict__ cls,
                             u32* __restrict__ flag) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_terms) return;
  const u16* s = chars + str_off[t];
  const u32 n = str_off[t + 1] - str_off[t];
  bool ok = n > 0 && (cls[s[0]] & 1u) != 0;                    // IsLetter(key[0]): \p{L}
  u32 at = 0;                                                  // the next segment may start at or behind `at`
  for (u32 i = 0; ok && i <= pat.len;) {
    u32 j = i;
    while (j < pat.len && pat.c[j] != u'_') j++;               // segment [i, j)
    const u32 m = j - i;
    if (m) {
      bool found = false;
      for (u32 p = at; p + m <= n; p++) {
        u32 q = 0;
        while (q < m && s[p + q] == pat.c[i + q]) q++;
        if (q == m) { found = true; at = p + m; break; }
      }
      ok = found;
    }
    i = j + 1;
  }
  flag[t] = ok ? 1u : 0u;
}
// the first `cap` flagged terms, in dictionary order (Take(MAX_LIKE_WORDS)): one workgroup walks the flags
__global__ __launch_bounds__(1024) void k_take_flagged(const u32* __restrict__ flag, u64 n_terms, u32 cap, u32* __restrict__ out, u32* __restrict__ n_out) {
  __shared__ u32 s_w[16];
  __shared__ u32 s_base;
  if (threadIdx.x == 0) s_base = 0;
  __syncthreads();
  for (u64 t0 = 0; t0 < n_terms && s_base < cap; t0 += 1024) {
    const u64 t = t0 + threadIdx.x;
    const u32 f = t < n_terms ? flag[t] : 0u;
    const u64 m = ballot64(f != 0);
    const u32 w = threadIdx.x >> 6;
    if (lane_id() == 0) s_w[w] = popc64(m);
    __syncthreads();
    u32 before = s_base;
    for (u32 k = 0; k < w; k++) before += s_w[k];
    const u32 pos = before + mbcnt(m);
    if (f && pos < cap) out[pos] = (u32)t;
    __syncthreads();
    if (threadIdx.x == 0) { u32 tot = 0; for (u32 k = 0; k < 16; k++) tot += s_w[k]; s_base += tot; }
    __syncthreads();
  }
  if (threadIdx.x == 0) *n_out = s_base < cap ? s_base : cap;
}
}  // namespace dcd

namespace dcd {
int32_t get_char_tables16(int device, CharTables16* out);
}

using namespace dcd;

// Parses a ".index" image into the DeviceIndex layout (units back to back + directory).
static int32_t parse_index_image(const u8* p, size_t n, DeviceIndex& ix, std::vector<u16>& units) {
  if (n < 8) DCD_FAIL(DCD_E_IO, ".index image shorter than its MaxCoord header");
  u64 mc = 0;
  for (int i = 0; i < 8; i++) mc |= (u64)p[i] << (8 * i);
  ix.max_coord = mc;
  size_t i = 8;
  ix.h_term_chars.clear();
  ix.h_term_str_off.assign(1, 0);
  ix.h_term_unit_start.assign(1, 0);
  ustr prev;
  bool first = true;
  while (i < n) {
    u32 len = 0;
    int sh = 0;
    for (;;) {
      if (i >= n) DCD_FAIL(DCD_E_IO, "truncated key length in .index");
      // BinaryReader.Read7BitEncodedInt: at most 5 length bytes, FormatException on a 6th (Index.Load then fails,
      // Index.cs:373-376)
      if (sh > 28) DCD_FAIL(DCD_E_IO, "malformed key length in .index (more than 5 length bytes)");
      u8 c = p[i++];
      len |= (u32)(c & 0x7F) << sh;
      sh += 7;
      if (!(c & 0x80)) break;
    }
    if ((int32_t)len < 0) DCD_FAIL(DCD_E_IO, "negative key length in .index");
    if (i + (size_t)len + 4 > n) DCD_FAIL(DCD_E_IO, "truncated record in .index");
    ustr key;
    utf8_to_utf16(p + i, len, key);
    i += len;
    u32 nu = 0;
    for (int k = 0; k < 4; k++) nu |= (u32)p[i + k] << (8 * k);
    i += 4;
    if ((int32_t)nu < 0 || i + 2ull * nu > n) DCD_FAIL(DCD_E_IO, "truncated posting list in .index");
    if (!first && !(prev < key)) DCD_FAIL(DCD_E_IO, "keys of .index are not strictly ascending (String.CompareOrdinal)");
    first = false;
    prev = key;
    ix.h_term_chars.insert(ix.h_term_chars.end(), (const u16*)key.data(), (const u16*)key.data() + key.size());
    ix.h_term_str_off.push_back((u32)ix.h_term_chars.size());
    size_t u0 = units.size();
    units.resize(u0 + nu);
    memcpy(units.data() + u0, p + i, 2ull * nu);
    i += 2ull * nu;
    ix.h_term_unit_start.push_back((u64)units.size());
  }
  ix.n_terms = ix.h_term_str_off.size() - 1;
  ix.n_units = units.size();
  return DCD_OK;
}

LazyPostings::~LazyPostings() {
  if (fd >= 0) close(fd);
  if (sub) { dcd_index* p = sub.release(); dcd_index_free(p); }
}

// The key / stub scan of Index.Load with InMemory == false (Index.cs:321-354): keys, unit counts and payload positions; the
// payloads are skipped (bin.BaseStream.Seek, :348).  Streams through the file with a bounded buffer.
static int32_t scan_index_file(int fd, DeviceIndex& ix, LazyPostings& lz) {
  struct stat stt;
  if (fstat(fd, &stt) != 0) DCD_FAIL(DCD_E_IO, "cannot stat .index");
  const u64 n = (u64)stt.st_size;
  if (n < 8) DCD_FAIL(DCD_E_IO, ".index image shorter than its MaxCoord header");
  auto rd = [&](u64 off, void* dst, size_t len) -> bool { return pread(fd, dst, len, (off_t)off) == (ssize_t)len; };
  u8 hdr[8];
  if (!rd(0, hdr, 8)) DCD_FAIL(DCD_E_IO, "short read from .index");
  u64 mc = 0;
  for (int i = 0; i < 8; i++) mc |= (u64)hdr[i] << (8 * i);
  ix.max_coord = mc;
  ix.h_term_chars.clear();
  ix.h_term_str_off.assign(1, 0);
  ix.h_term_unit_start.assign(1, 0);
  u64 i = 8;
  ustr prev;
  bool first = true;
  // ONE sliding window over the file: it is refilled only when the next record head (<= 5 length bytes + key + 4 count bytes)
  // is not completely inside it.  Most records of a dictionary's tail lie a few bytes apart, so the file is read about once;
  // (round 3 read a fresh 64 KiB window per record: V x 64 KiB of pread traffic, 6.5 GB for a 100 k-term vocabulary)
  constexpr u32 MAX_KEY_BYTES = 1u << 20;                 // (keys are <= 3 * 63 UTF-8 bytes; anything near this is a damaged file)
  std::vector<u8> buf(1 << 16);
  u64 w0 = 0, w1 = 0;                                     // the window holds the file bytes [w0, w1)
  auto need = [&](u64 from, u64 len) -> bool {            // makes [from, from + len) part of the window (len <= buf.size())
    if (from >= w0 && from + len <= w1) return true;
    if (from + len > n) return false;
    const size_t want = (size_t)std::min<u64>(buf.size(), n - from);
    if (!rd(from, buf.data(), want)) return false;
    w0 = from; w1 = from + want;
    return true;
  };
  while (i < n) {
    size_t k = 0;
    u32 len = 0;
    int sh = 0;
    for (;;) {
      if (i + k >= n) DCD_FAIL(DCD_E_IO, "truncated key length in .index");
      if (sh > 28) DCD_FAIL(DCD_E_IO, "malformed key length in .index (more than 5 length bytes)");
      if (!need(i + k, 1)) DCD_FAIL(DCD_E_IO, "short read from .index");
      u8 c = buf[(size_t)(i + k - w0)];
      k++;
      len |= (u32)(c & 0x7F) << sh;
      sh += 7;
      if (!(c & 0x80)) break;
    }
    if ((int32_t)len < 0 || i + k + (u64)len + 4 > n) DCD_FAIL(DCD_E_IO, "truncated record in .index");
    if (len > MAX_KEY_BYTES) DCD_FAIL(DCD_E_IO, "key of %u bytes in .index: damaged file", len);
    if ((size_t)len + 4 > buf.size()) buf.resize((size_t)len + 4 + (1 << 16)), w0 = w1 = 0;     // (a key longer than the window)
    if (!need(i + k, (u64)len + 4)) DCD_FAIL(DCD_E_IO, "short read from .index");
    const u8* rp = buf.data() + (size_t)(i + k - w0);
    ustr key;
    utf8_to_utf16(rp, len, key);
    u32 nu = 0;
    for (int b = 0; b < 4; b++) nu |= (u32)rp[len + b] << (8 * b);
    k += (size_t)len + 4;
    if ((int32_t)nu < 0 || i + k + 2ull * nu > n) DCD_FAIL(DCD_E_IO, "truncated posting list in .index");
    if (!first && !(prev < key)) DCD_FAIL(DCD_E_IO, "keys of .index are not strictly ascending (String.CompareOrdinal)");
    first = false;
    prev = key;
    ix.h_term_chars.insert(ix.h_term_chars.end(), (const u16*)key.data(), (const u16*)key.data() + key.size());
    ix.h_term_str_off.push_back((u32)ix.h_term_chars.size());
    lz.file_off.push_back(i + k);
    lz.n_units.push_back(nu);
    ix.h_term_unit_start.push_back(ix.h_term_unit_start.back() + nu);
    i += k + 2ull * nu;
  }
  ix.n_terms = ix.h_term_str_off.size() - 1;
  ix.n_units = ix.h_term_unit_start.back();
  return DCD_OK;
}

// LoadSequence, Index.cs:198-216: n_units ushorts at the list's file position
int32_t dcd::lazy_read_units(const LazyPostings& lz, u32 term, u16* dst) {
  const size_t bytes = 2ull * lz.n_units[term];
  size_t done = 0;
  while (done < bytes) {
    ssize_t r = pread(lz.fd, (u8*)dst + done, bytes - done, (off_t)(lz.file_off[term] + done));
    if (r <= 0) DCD_FAIL(DCD_E_IO, "short read of a posting list from .index");
    done += (size_t)r;
  }
  return DCD_OK;
}

// IndexPageList.Load, Build.cs:119-148
static int32_t parse_list_image(const u8* p, size_t n, PageTable& pt) {
  size_t i = 0;
  pt = PageTable();
  while (i + 8 <= n) {
    u64 v = 0;
    for (int k = 0; k < 8; k++) v |= (u64)p[i + k] << (8 * k);
    i += 8;
    u32 len = 0;
    int sh = 0;
    bool ok = true;
    for (;;) {
      if (i >= n || sh > 28) { ok = false; break; }   // EndOfStream / FormatException after 5 length bytes
      u8 c = p[i++];
      len |= (u32)(c & 0x7F) << sh;
      sh += 7;
      if (!(c & 0x80)) break;
    }
    // IndexPageList.Load swallows every exception and keeps what it has read so far (Build.cs:119-148)
    if (!ok || (int32_t)len < 0 || i + (size_t)len > n) break;
    ustr s;
    utf8_to_utf16(p + i, len, s);
    i += len;
    if (s.empty()) break;   // s[0] throws in the reference (Build.cs:131): the load ends at this record
    if (s[0] != u':') {
      DocMeta d;
      d.name = s;
      d.first_page = pt.end.size();
      pt.docs.push_back(d);
    } else {
      if (pt.docs.empty()) { DocMeta d; d.first_page = 0; pt.docs.push_back(d); }
      pt.ids.push_back(s.substr(1));
      pt.end.push_back(v);
      pt.page_doc.push_back((u32)(pt.docs.size() - 1));
    }
  }
  return DCD_OK;
}

__global__ void k_count_term_values(const u16* __restrict__ units, const u64* __restrict__ unit_start, u64 n_terms,
                                    u64* __restrict__ counts) {
  // one wave per term: number of terminator units
  u64 t = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= n_terms) return;
  u64 a = unit_start[t], b = unit_start[t + 1];
  u32 c = 0;
  for (u64 i = a + lane_id(); i < b; i += 64) c += (units[i] & 0x8000u) ? 0u : 1u;
  c = wave_sum(c);
  if (lane_id() == 0) counts[t] = c;
}

namespace dcd {
int32_t index_from_host_units(DeviceIndex& ix, const std::vector<u16>& units, int device);
}
static int32_t index_from_host(DeviceIndex& ix, const std::vector<u16>& units, int device) { return dcd::index_from_host_units(ix, units, device); }
int32_t dcd::index_from_host_units(DeviceIndex& ix, const std::vector<u16>& units, int device) {
  DCD_TRY(ensure_device(device));
  ix.device = device;
  u64 V = ix.n_terms, U = ix.n_units;
  DCD_TRY(ix.units.alloc(device, U ? U : 1));
  DCD_TRY(ix.term_unit_start.alloc(device, V + 1));
  DCD_TRY(ix.term_item_start.alloc(device, V + 1));
  DCD_TRY(ix.term_chars.alloc(device, ix.h_term_chars.size() ? ix.h_term_chars.size() : 1));
  DCD_TRY(ix.term_str_off.alloc(device, V + 1));
  if (U) DCD_HIP(hipMemcpy(ix.units.p, units.data(), U * 2, hipMemcpyHostToDevice));
  DCD_HIP(hipMemcpy(ix.term_unit_start.p, ix.h_term_unit_start.data(), (V + 1) * 8, hipMemcpyHostToDevice));
  if (!ix.h_term_chars.empty())
    DCD_HIP(hipMemcpy(ix.term_chars.p, ix.h_term_chars.data(), ix.h_term_chars.size() * 2, hipMemcpyHostToDevice));
  DCD_HIP(hipMemcpy(ix.term_str_off.p, ix.h_term_str_off.data(), (V + 1) * 4, hipMemcpyHostToDevice));
  // posting counts per term (Count()): terminator units, counted on the device
  DevBuf<u64> counts;
  DCD_TRY(counts.alloc(device, V + 1));
  std::vector<u64> hc(V + 1, 0);
  if (V) {
    hipLaunchKernelGGL(k_count_term_values, dim3((u32)div_up(V * 64, 256)), dim3(256), 0, 0, ix.units.p, ix.term_unit_start.p, V, counts.p);
    DCD_HIP(hipGetLastError());
    DCD_HIP(hipMemcpy(hc.data(), counts.p, V * 8, hipMemcpyDeviceToHost));
  }
  ix.h_term_item_start.assign(V + 1, 0);
  for (u64 t = 0; t < V; t++) ix.h_term_item_start[t + 1] = ix.h_term_item_start[t] + hc[t];
  ix.n_postings = ix.h_term_item_start[V];
  DCD_HIP(hipMemcpy(ix.term_item_start.p, ix.h_term_item_start.data(), (V + 1) * 8, hipMemcpyHostToDevice));
  if (!ix.pages.end.empty()) {
    DCD_TRY(ix.page_end.alloc(device, ix.pages.end.size()));
    DCD_HIP(hipMemcpy(ix.page_end.p, ix.pages.end.data(), ix.pages.end.size() * 8, hipMemcpyHostToDevice));
  }
  return DCD_OK;
}

static int32_t read_file(const std::string& path, std::vector<u8>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) DCD_FAIL(DCD_E_IO, "cannot open %s", path.c_str());
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize((size_t)n);
  bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  if (!ok) DCD_FAIL(DCD_E_IO, "short read from %s", path.c_str());
  return DCD_OK;
}

extern "C" {

int32_t dcd_index_from_build(dcd_build* b, dcd_index** out) {
  if (!b || !out) DCD_FAIL(DCD_E_BADARG, "dcd_index_from_build: NULL argument");
  // everything that can fail happens while the build still owns its index: a failed call leaves the builder usable
  DeviceIndex* peek = build_peek_index(b);
  if (!peek) DCD_FAIL(DCD_E_STATE, "build is not finished or its index was already adopted");
  if (peek->shard_only) DCD_FAIL(DCD_E_STATE, "a DCD_BUILD_SHARD_ONLY build can only be exported");
  DCD_TRY(index_materialize_units(*peek, build_stream(b)));   // (the build left its units in sort-key order)
  DCD_TRY(index_fetch_host(*peek));
  hipStream_t st = nullptr;
  DCD_TRY(stream_acquire(peek->device, &st));
  DeviceIndex* src = build_take_index(b);
  dcd_index* h = new dcd_index();
  h->ix = std::move(*src);
  h->st = st;
  *out = h;
  return DCD_OK;
}

int32_t dcd_index_from_images(const uint8_t* index_img, size_t n_index, const uint8_t* list_img, size_t n_list, int32_t device,
                              dcd_index** out) {
  if (!index_img || !out || (!list_img && n_list)) DCD_FAIL(DCD_E_BADARG, "dcd_index_from_images: NULL argument");
  std::unique_ptr<dcd_index> h(new dcd_index());
  std::vector<u16> units;
  DCD_TRY(parse_index_image(index_img, n_index, h->ix, units));   // host-side format checks first: DCD_E_IO needs no device
  DCD_TRY(parse_list_image(list_img, n_list, h->ix.pages));
  DCD_TRY(ensure_device(device));
  DCD_TRY(index_from_host(h->ix, units, device));
  DCD_TRY(stream_acquire(device, &h->st));
  *out = h.release();
  return DCD_OK;
}

int32_t dcd_index_open_lazy(const char* dir_utf8, int32_t device, uint64_t resident_bytes, dcd_index** out) {
  if (!dir_utf8 || !out) DCD_FAIL(DCD_E_BADARG, "dcd_index_open_lazy: NULL argument");
  std::string d(dir_utf8);
  if (!d.empty() && d.back() != '/') d.push_back('/');
  std::unique_ptr<dcd_index> h(new dcd_index());
  h->lazy.reset(new LazyPostings());
  h->lazy->resident_budget = resident_bytes ? resident_bytes : (1ull << 30);
  h->lazy->fd = open((d + ".index").c_str(), O_RDONLY);
  if (h->lazy->fd < 0) DCD_FAIL(DCD_E_IO, "cannot open %s.index", d.c_str());
  DCD_TRY(scan_index_file(h->lazy->fd, h->ix, *h->lazy));
  std::vector<u8> lst;
  DCD_TRY(read_file(d + ".index.list", lst));
  DCD_TRY(parse_list_image(lst.data(), lst.size(), h->ix.pages));
  DCD_TRY(ensure_device(device));
  // the dictionary and the page table live on the device (dictionary scans, PrepareSearchResult); no posting unit does
  DeviceIndex& ix = h->ix;
  ix.device = device;
  const u64 V = ix.n_terms;
  DCD_TRY(ix.term_chars.alloc(device, ix.h_term_chars.size() ? ix.h_term_chars.size() : 1));
  DCD_TRY(ix.term_str_off.alloc(device, V + 1));
  DCD_TRY(ix.term_unit_start.alloc(device, V + 1));
  if (!ix.h_term_chars.empty()) DCD_HIP(hipMemcpy(ix.term_chars.p, ix.h_term_chars.data(), ix.h_term_chars.size() * 2, hipMemcpyHostToDevice));
  DCD_HIP(hipMemcpy(ix.term_str_off.p, ix.h_term_str_off.data(), (V + 1) * 4, hipMemcpyHostToDevice));
  DCD_HIP(hipMemcpy(ix.term_unit_start.p, ix.h_term_unit_start.data(), (V + 1) * 8, hipMemcpyHostToDevice));
  if (!ix.pages.end.empty()) {
    DCD_TRY(ix.page_end.alloc(device, ix.pages.end.size()));
    DCD_HIP(hipMemcpy(ix.page_end.p, ix.pages.end.data(), ix.pages.end.size() * 8, hipMemcpyHostToDevice));
  }
  ix.h_term_item_start.clear();                // Count() of a list is known once it has been read (dcd_index_term_count)
  DCD_TRY(stream_acquire(device, &h->st));
  *out = h.release();
  return DCD_OK;
}

int32_t dcd_index_lazy_stats(dcd_index* h, uint64_t* lists_read, uint64_t* bytes_read, uint64_t* resident_terms) {
  if (!h) DCD_FAIL(DCD_E_BADARG, "dcd_index_lazy_stats: NULL handle");
  if (!h->lazy) DCD_FAIL(DCD_E_STATE, "not an on-disk index (dcd_index_open_lazy)");
  if (lists_read) *lists_read = h->lazy->fetches;
  if (bytes_read) *bytes_read = h->lazy->fetched_bytes;
  if (resident_terms) *resident_terms = h->lazy->terms.size();
  return DCD_OK;
}

int32_t dcd_index_open(const char* dir_utf8, int32_t device, dcd_index** out) {
  if (!dir_utf8 || !out) DCD_FAIL(DCD_E_BADARG, "dcd_index_open: NULL argument");
  std::string d(dir_utf8);
  if (!d.empty() && d.back() != '/') d.push_back('/');
  std::vector<u8> a, b;
  DCD_TRY(read_file(d + ".index", a));
  DCD_TRY(read_file(d + ".index.list", b));
  return dcd_index_from_images(a.data(), a.size(), b.data(), b.size(), device, out);
}

void dcd_index_free(dcd_index* h) {
  if (!h) return;
  (void)hipSetDevice(h->ix.device);
  hipStream_t st = h->st;
  StreamScope ss_(st);
  if (st) (void)hipStreamSynchronize(st);
  const int dev = h->ix.device;
  delete h;
  stream_release(dev, st);
}

int32_t dcd_index_count(dcd_index* h, uint64_t* n) { if (!h || !n) DCD_FAIL(DCD_E_BADARG, "NULL argument"); *n = h->ix.n_terms; return DCD_OK; }
int32_t dcd_index_max_coord(dcd_index* h, uint64_t* v) { if (!h || !v) DCD_FAIL(DCD_E_BADARG, "NULL argument"); *v = h->ix.max_coord; return DCD_OK; }
int32_t dcd_index_n_units(dcd_index* h, uint64_t* n) { if (!h || !n) DCD_FAIL(DCD_E_BADARG, "NULL argument"); *n = h->ix.n_units; return DCD_OK; }
int32_t dcd_index_n_pages(dcd_index* h, uint64_t* n) { if (!h || !n) DCD_FAIL(DCD_E_BADARG, "NULL argument"); *n = h->ix.pages.n_pages(); return DCD_OK; }
int32_t dcd_index_n_docs(dcd_index* h, uint64_t* n) { if (!h || !n) DCD_FAIL(DCD_E_BADARG, "NULL argument"); *n = h->ix.pages.n_docs(); return DCD_OK; }

int32_t dcd_index_find(dcd_index* h, const uint16_t* key, size_t len, uint32_t* term) {
  if (!h || (!key && len) || !term) DCD_FAIL(DCD_E_BADARG, "dcd_index_find: NULL argument");
  DCD_TRY(index_fetch_host(h->ix));     // (host mirrors of a merged index are fetched on first use)
  const DeviceIndex& ix = h->ix;
  // binary search over the ordinally sorted keys (String.CompareOrdinal = UTF-16 code-unit order)
  u64 a = 0, b = ix.n_terms;
  while (a < b) {
    u64 m = (a + b) >> 1;
    const u16* s = ix.h_term_chars.data() + ix.h_term_str_off[m];
    size_t sl = ix.h_term_str_off[m + 1] - ix.h_term_str_off[m];
    size_t k = 0;
    int cmp = 0;
    while (k < sl && k < len) { if (s[k] != key[k]) { cmp = s[k] < key[k] ? -1 : 1; break; } k++; }
    if (cmp == 0) cmp = sl < len ? -1 : (sl > len ? 1 : 0);
    if (cmp == 0) { *term = (u32)m; return DCD_OK; }
    if (cmp < 0) a = m + 1; else b = m;
  }
  return DCD_E_NOTFOUND;  // missing term: empty sequence in the reference (Search.cs:235,256); no message
}

int32_t dcd_index_key(dcd_index* h, uint32_t term, uint16_t* out, size_t cap, size_t* len) {
  if (!h || !len) DCD_FAIL(DCD_E_BADARG, "dcd_index_key: NULL argument");
  if (term >= h->ix.n_terms) DCD_FAIL(DCD_E_BADARG, "term %u out of range", term);
  DCD_TRY(index_fetch_host(h->ix));
  size_t n = h->ix.h_term_str_off[term + 1] - h->ix.h_term_str_off[term];
  *len = n;
  if (!out || cap < n) return DCD_E_OVERFLOW;
  memcpy(out, h->ix.h_term_chars.data() + h->ix.h_term_str_off[term], n * 2);
  return DCD_OK;
}

int32_t dcd_index_get(dcd_index* h, uint32_t term, uint16_t* out, size_t cap, size_t* n_units) {
  if (!h || !n_units) DCD_FAIL(DCD_E_BADARG, "dcd_index_get: NULL argument");
  if (term >= h->ix.n_terms) DCD_FAIL(DCD_E_BADARG, "term %u out of range", term);
  DCD_TRY(index_fetch_host(h->ix));
  u64 a = h->ix.h_term_unit_start[term], b = h->ix.h_term_unit_start[term + 1];
  *n_units = (size_t)(b - a);
  if (!out || cap < b - a) return DCD_E_OVERFLOW;
  if (h->lazy) {                               // LoadSequence, Index.cs:198-216
    if (b > a) { DCD_TRY(lazy_read_units(*h->lazy, term, out)); h->lazy->fetches++; h->lazy->fetched_bytes += 2 * (b - a); }
    return DCD_OK;
  }
  DCD_TRY(ensure_device(h->ix.device));
  if (b > a) DCD_HIP(hipMemcpy(out, h->ix.units.p + a, (b - a) * 2, hipMemcpyDeviceToHost));
  return DCD_OK;
}

int32_t dcd_index_term_count(dcd_index* h, uint32_t term, uint64_t* n_values) {
  if (!h || !n_values) DCD_FAIL(DCD_E_BADARG, "dcd_index_term_count: NULL argument");
  if (term >= h->ix.n_terms) DCD_FAIL(DCD_E_BADARG, "term %u out of range", term);
  DCD_TRY(index_fetch_host(h->ix));
  if (h->lazy) {
    // Count() enumerates the list (Search.cs:307): read it, count the terminator units on the device
    const u32 nu = h->lazy->n_units[term];
    *n_values = 0;
    if (!nu) return DCD_OK;
    std::vector<u16> u(nu);
    DCD_TRY(lazy_read_units(*h->lazy, term, u.data()));
    h->lazy->fetches++; h->lazy->fetched_bytes += 2ull * nu;
    const int dev = h->ix.device;
    DCD_TRY(ensure_device(dev));
    StreamScope ss_(h->st);
    DevBuf<u16> du; DevBuf<u64> us, cnt;
    DCD_TRY(du.alloc(dev, nu)); DCD_TRY(us.alloc(dev, 2)); DCD_TRY(cnt.alloc(dev, 1));
    const u64 se[2] = {0, nu};
    DCD_HIP(hipMemcpyAsync(du.p, u.data(), 2ull * nu, hipMemcpyHostToDevice, h->st));
    DCD_HIP(hipMemcpyAsync(us.p, se, 16, hipMemcpyHostToDevice, h->st));
    hipLaunchKernelGGL(k_count_term_values, dim3(1), dim3(64), 0, h->st, du.p, us.p, (u64)1, cnt.p);
    DCD_HIP(hipMemcpyAsync(n_values, cnt.p, 8, hipMemcpyDeviceToHost, h->st));
    DCD_HIP(hipStreamSynchronize(h->st));
    return DCD_OK;
  }
  *n_values = h->ix.h_term_item_start[term + 1] - h->ix.h_term_item_start[term];
  return DCD_OK;
}

int32_t dcd_index_page_info(dcd_index* h, uint64_t page, uint64_t* end_coord, uint32_t* doc, uint16_t* id_out, size_t id_cap,
                            size_t* id_len) {
  if (!h) DCD_FAIL(DCD_E_BADARG, "dcd_index_page_info: NULL handle");
  const PageTable& pt = h->ix.pages;
  if (page >= pt.n_pages()) DCD_FAIL(DCD_E_BADARG, "page out of range");
  if (end_coord) *end_coord = pt.end_at(page);
  if (doc) *doc = pt.doc_of_page(page);
  ustr id = pt.page_id(page);
  if (id_len) *id_len = id.size();
  if (id_out) {
    if (id_cap < id.size()) return DCD_E_OVERFLOW;
    memcpy(id_out, id.data(), id.size() * 2);
  }
  return DCD_OK;
}

int32_t dcd_index_doc_name(dcd_index* h, uint32_t doc, uint16_t* out, size_t cap, size_t* len) {
  if (!h || !len) DCD_FAIL(DCD_E_BADARG, "dcd_index_doc_name: NULL argument");
  if (doc >= h->ix.pages.n_docs()) DCD_FAIL(DCD_E_BADARG, "doc out of range");
  ustr s = h->ix.pages.doc_name(doc);
  *len = s.size();
  if (!out || cap < s.size()) return DCD_E_OVERFLOW;
  memcpy(out, s.data(), s.size() * 2);
  return DCD_OK;
}

}  // extern "C"

extern "C" {

// ".index" / ".index.list" of an index handle (an adopted build, a merged shard partition, a gathered index)
int32_t dcd_index_image(dcd_index* h, uint8_t* out, size_t cap, size_t* n) {
  if (!h || !n) DCD_FAIL(DCD_E_BADARG, "dcd_index_image: NULL argument");
  if (h && h->lazy) DCD_FAIL(DCD_E_STATE, "dcd_index_image: the postings of an on-disk index (dcd_index_open_lazy) are not in memory");
  std::lock_guard<std::mutex> guard(h->mu);
  if (!h->image_bytes) DCD_TRY(index_make_image(h->ix, h->st, h->image, &h->image_bytes));
  *n = (size_t)h->image_bytes;
  if (!out || cap < h->image_bytes) return DCD_E_OVERFLOW;
  DCD_TRY(ensure_device(h->ix.device));
  DCD_HIP(hipMemcpy(out, h->image.p, h->image_bytes, hipMemcpyDeviceToHost));
  return DCD_OK;
}

int32_t dcd_index_list_image(dcd_index* h, uint8_t* out, size_t cap, size_t* n) {
  if (!h || !n) DCD_FAIL(DCD_E_BADARG, "dcd_index_list_image: NULL argument");
  std::vector<u8> img;
  write_list_image(h->ix.pages, img);
  *n = img.size();
  if (!out || cap < img.size()) return DCD_E_OVERFLOW;
  if (!img.empty()) memcpy(out, img.data(), img.size());
  return DCD_OK;
}

// attaches a page table given as ".index.list" bytes (Build.cs:99-148), e.g. the concatenation of the shards' lists
int32_t dcd_index_set_list_image(dcd_index* h, const uint8_t* list_img, size_t n_list) {
  if (!h || (!list_img && n_list)) DCD_FAIL(DCD_E_BADARG, "dcd_index_set_list_image: NULL argument");
  std::lock_guard<std::mutex> guard(h->mu);
  PageTable pt;
  DCD_TRY(parse_list_image(list_img, n_list, pt));
  h->ix.pages = std::move(pt);
  h->ix.page_end.reset();   // rebuilt on the next query
  return DCD_OK;
}

int32_t dcd_index_save(dcd_index* h, const char* dir_utf8) {
  if (!h || !dir_utf8) DCD_FAIL(DCD_E_BADARG, "dcd_index_save: NULL argument");
  if (h && h->lazy) DCD_FAIL(DCD_E_STATE, "dcd_index_save: the postings of an on-disk index (dcd_index_open_lazy) are not in memory");
  size_t n = 0;
  (void)dcd_index_image(h, nullptr, 0, &n);
  std::vector<u8> img(n);
  DCD_TRY(dcd_index_image(h, img.data(), img.size(), &n));
  std::vector<u8> lst;
  write_list_image(h->ix.pages, lst);
  std::string d(dir_utf8);
  if (!d.empty() && d.back() != '/') d.push_back('/');
  const char* names[2] = {".index", ".index.list"};
  const std::vector<u8>* data[2] = {&img, &lst};
  for (int k = 0; k < 2; k++) {
    FILE* f = fopen((d + names[k]).c_str(), "wb");
    if (!f) DCD_FAIL(DCD_E_IO, "cannot create %s%s", d.c_str(), names[k]);
    bool ok = fwrite(data[k]->data(), 1, data[k]->size(), f) == data[k]->size();
    ok = fclose(f) == 0 && ok;
    if (!ok) DCD_FAIL(DCD_E_IO, "short write to %s%s", d.c_str(), names[k]);
  }
  return DCD_OK;
}

}  // extern "C"

extern "C" {

// GetCloseWords, Search.cs:168-174: the n keys closest to `word` by Levenshtein distance (ties: dictionary order)
int32_t dcd_index_close_words(dcd_index* h, const uint16_t* word, size_t len, uint32_t n, uint32_t* out_terms, uint32_t* n_out) {
  if (!h || (!word && len) || !out_terms || !n_out) DCD_FAIL(DCD_E_BADARG, "dcd_index_close_words: NULL argument");
  if (len > 32) DCD_FAIL(DCD_E_UNSUPPORTED, "words longer than 32 UTF-16 units are not indexed (MAX_WORD_LENGTH)");
  std::lock_guard<std::mutex> guard(h->mu);
  DeviceIndex& ix = h->ix;
  const u64 V = ix.n_terms;
  const u32 k = (u32)std::min<u64>(n, V);
  *n_out = k;
  if (!k) return DCD_OK;
  const int dev = ix.device;
  DCD_TRY(ensure_device(dev));
  hipStream_t st = h->st;
  StreamScope ss_(st);
  LevWord w;
  memset(&w, 0, sizeof w);
  for (size_t i = 0; i < len; i++) w.c[i] = word[i];
  w.len = (u32)len;
  DevBuf<uint2> pa, pb;
  DevBuf<u32> hist, sscr;
  DCD_TRY(pa.alloc(dev, V)); DCD_TRY(pb.alloc(dev, V));
  DCD_TRY(hist.alloc(dev, (u64)RS_MAX_BLOCKS << 10));
  DCD_TRY(sscr.alloc(dev, scan_scratch_elems((u64)RS_MAX_BLOCKS << 10)));
  hipLaunchKernelGGL(k_levenshtein, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ix.term_chars.p, ix.term_str_off.p, V, w, pa.p);
  uint2* sorted = rs_sort(pa.p, pb.p, nullptr, V, 7, hist.p, sscr.p, st);      // distances <= 64 + 32: 7 bits, stable
  std::vector<uint2> top(k);
  DCD_HIP(hipMemcpyAsync(top.data(), sorted, (size_t)k * sizeof(uint2), hipMemcpyDeviceToHost, st));
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  for (u32 i = 0; i < k; i++) out_terms[i] = top[i].y;
  return DCD_OK;
}

// GetLikeWords, Search.cs:158-167: the first n keys (dictionary order) that begin with a letter and match `pattern`, where '_'
// (what the request rewriter makes of '?', Search.cs:340) stands for ".*" and the match is unanchored.  The host keeps the
// trivial branches (no '_' or !bKeepForms: the word itself; fewer than 2 characters: nothing).
int32_t dcd_index_like_words(dcd_index* h, const uint16_t* pattern, size_t len, uint32_t n, uint32_t* out_terms, uint32_t* n_out) {
  if (!h || (!pattern && len) || !out_terms || !n_out) DCD_FAIL(DCD_E_BADARG, "dcd_index_like_words: NULL argument");
  if (len > 64) DCD_FAIL(DCD_E_UNSUPPORTED, "patterns longer than 64 UTF-16 units are not supported");
  std::lock_guard<std::mutex> guard(h->mu);
  DeviceIndex& ix = h->ix;
  const u64 V = ix.n_terms;
  *n_out = 0;
  if (!V || !n) return DCD_OK;
  const int dev = ix.device;
  DCD_TRY(ensure_device(dev));
  hipStream_t st = h->st;
  StreamScope ss_(st);
  CharTables16 tab{nullptr, nullptr};
  DCD_TRY(get_char_tables16(dev, &tab));
  LikePat pat;
  memset(&pat, 0, sizeof pat);
  for (size_t i = 0; i < len; i++) pat.c[i] = pattern[i];
  pat.len = (u32)len;
  DevBuf<u32> flag, out, cnt;
  DCD_TRY(flag.alloc(dev, V)); DCD_TRY(out.alloc(dev, n)); DCD_TRY(cnt.alloc(dev, 1));
  hipLaunchKernelGGL(k_like_words, dim3((u32)div_up(V, 256)), dim3(256), 0, st, ix.term_chars.p, ix.term_str_off.p, V, pat, tab.cls, flag.p);
  hipLaunchKernelGGL(k_take_flagged, dim3(1), dim3(1024), 0, st, flag.p, V, n, out.p, cnt.p);
  u32 k = 0;
  DCD_HIP(hipMemcpyAsync(&k, cnt.p, 4, hipMemcpyDeviceToHost, st));
  DCD_HIP(hipStreamSynchronize(st));
  DCD_HIP(hipGetLastError());
  if (k) DCD_HIP(hipMemcpy(out_terms, out.p, (size_t)k * 4, hipMemcpyDeviceToHost));
  *n_out = k;
  return DCD_OK;
}

// Term selection of GetSuggessions, Search.cs:176-188: keys that start with `prefix`, are longer than it and whose first
// unit is >= 'A', ordered by descending IndexSequence.Count (= number of ushort units; stable: dictionary order among
// equals), the first n.  Keys are sorted ordinally, so the candidates are one contiguous range of the dictionary.
int32_t dcd_index_suggest(dcd_index* h, const uint16_t* prefix, size_t len, uint32_t n, uint32_t* out_terms, uint32_t* n_out) {
  if (!h || (!prefix && len) || !out_terms || !n_out) DCD_FAIL(DCD_E_BADARG, "dcd_index_suggest: NULL argument");
  DCD_TRY(index_fetch_host(h->ix));
  const DeviceIndex& ix = h->ix;
  auto cmp_prefix = [&](u64 m) -> int {   // key m vs prefix, over the first len units
    const u16* s = ix.h_term_chars.data() + ix.h_term_str_off[m];
    const size_t sl = ix.h_term_str_off[m + 1] - ix.h_term_str_off[m];
    for (size_t k = 0; k < len; k++) {
      if (k >= sl) return -1;
      if (s[k] != prefix[k]) return s[k] < prefix[k] ? -1 : 1;
    }
    return 0;
  };
  u64 a = 0, b = ix.n_terms;
  while (a < b) { u64 m = (a + b) >> 1; if (cmp_prefix(m) < 0) a = m + 1; else b = m; }
  const u64 lo = a;
  b = ix.n_terms;
  while (a < b) { u64 m = (a + b) >> 1; if (cmp_prefix(m) <= 0) a = m + 1; else b = m; }
  const u64 hi = a;
  std::vector<std::pair<u64, u32>> cand;   // (unit count, term)
  for (u64 t = lo; t < hi; t++) {
    const size_t sl = ix.h_term_str_off[t + 1] - ix.h_term_str_off[t];
    if (sl <= len || ix.h_term_chars[ix.h_term_str_off[t]] < u'A') continue;
    cand.emplace_back(ix.h_term_unit_start[t + 1] - ix.h_term_unit_start[t], (u32)t);
  }
  std::stable_sort(cand.begin(), cand.end(), [](const std::pair<u64, u32>& x, const std::pair<u64, u32>& y) { return x.first > y.first; });
  const u32 k = (u32)std::min<u64>(n, cand.size());
  for (u32 i = 0; i < k; i++) out_terms[i] = cand[i].second;
  *n_out = k;
  return DCD_OK;
}

}  // extern "C"

// dcd_build_int.h -- internal layout of a finished build: the HBM-resident index that both the
// ".index" image writer and the query path consume.
#pragma once
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "dcd_common.h"

namespace dcd {

typedef std::u16string ustr;

// A vector that is SHARED by its copies until one of them writes (then that copy gets its own), and that may be LAZY: known size,
// contents made by a generator on first access.  The page table of a build is handed to the index the build leaves (DeviceIndex::pages)
// and stays with the builder for its ".index.list": a deep copy of a million-entry table was 1.3 ms of every build with explicit
// pages; and the END coordinates of a bulk batch (one u64 per page) are a function of the caller's offset array that the DEVICE
// needs, the host only if somebody asks for the list image or a page's coordinates.
template <typename T>
class CowVec {
  struct Rep {
    std::vector<T> v;
    std::function<void(std::vector<T>&)> gen;   // non-null: v is still to be made (n entries)
    size_t n = 0;
    std::once_flag once;
  };
  std::shared_ptr<Rep> r;
  const std::vector<T>& get() const {
    static const std::vector<T> none;
    if (!r) return none;
    if (r->gen) std::call_once(r->once, [&] { r->gen(r->v); r->gen = nullptr; });
    return r->v;
  }
  std::vector<T>& mut() {
    if (!r) r = std::make_shared<Rep>();
    else if (r.use_count() > 1 || r->gen) { auto n = std::make_shared<Rep>(); n->v = get(); r = n; }
    return r->v;
  }

 public:
  size_t size() const { return !r ? 0 : (r->gen ? r->n : r->v.size()); }
  bool empty() const { return size() == 0; }
  const T& operator[](size_t i) const { return get()[i]; }
  const T& back() const { return get().back(); }
  const T* data() const { return get().data(); }
  typename std::vector<T>::const_iterator begin() const { return get().begin(); }
  typename std::vector<T>::const_iterator end() const { return get().end(); }
  T* data() { return mut().data(); }
  void reserve(size_t n) { mut().reserve(n); }
  void resize(size_t n) { mut().resize(n); }
  void push_back(const T& x) { mut().push_back(x); }
  void clear() { r.reset(); }
  // n entries, made by gen on first access (only for an empty vector)
  void set_lazy(size_t n, std::function<void(std::vector<T>&)> gen) { r = std::make_shared<Rep>(); r->n = n; r->gen = std::move(gen); }
};

// Documents / pages as Index.Builder.pages keeps them (Build.cs:343-367): a document entry opens a
// doc, page entries carry the id and the END coordinate of the page.
struct DocMeta {
  ustr name;           // "source:name" (Build.cs:355)
  u64 first_page;      // index of its first page in the page table
};
struct PageTable {
  // Explicit pages (ids stored) ...
  CowVec<ustr> ids;               // per page id; empty vector when `uniform_id` is used
  ustr uniform_id;                // id shared by all pages (synthetic corpora: "1")
  CowVec<u64> end;                // END coordinate per page (IndexPageList key, Build.cs:66)
  CowVec<DocMeta> docs;           // explicit docs
  // ... or one document per page named by its decimal index
  bool doc_per_page = false;
  ustr source;                    // source name for doc_per_page
  u64 first_doc = 0;
  CowVec<u32> page_doc;           // doc index per page (explicit docs only)
  // ... or n uniform pages of `uni_chars` characters (implies doc_per_page, uniform id): no vectors at all
  bool uniform = false;
  u64 uni_n = 0, uni_chars = 0, uni_base = 0;
  u64 n_pages() const { return uniform ? uni_n : end.size(); }
  u64 n_docs() const { return doc_per_page ? n_pages() : docs.size(); }
  u64 end_at(u64 p) const { return uniform ? uni_base + (p + 1) * uni_chars : end[p]; }
  ustr doc_name(u64 d) const;
  ustr page_id(u64 p) const { return ids.empty() ? uniform_id : ids[p]; }
  u32 doc_of_page(u64 p) const { return doc_per_page ? (u32)p : page_doc[p]; }
};

// Device-resident index (term order = String.CompareOrdinal order of the keys, Index.cs:142-151)
struct DeviceIndex {
  int device = 0;
  u64 n_terms = 0, n_units = 0, n_postings = 0, max_coord = 0;
  DevBuf<u16> units;             // all posting lists back to back, term 0 first
  DevBuf<u64> term_unit_start;   // [n_terms + 1]
  DevBuf<u64> term_item_start;   // [n_terms + 1] posting (value) counts as differences
  DevBuf<u16> term_chars;        // keys, UTF-16 code units, back to back
  DevBuf<u32> term_str_off;      // [n_terms + 1]
  DevBuf<u64> page_end;          // [n_pages] END coordinates (device copy of PageTable::end)
  DevBuf<u64> term_first, term_last;  // [n_terms] first / last absolute coordinate per term (shard merge)
  // DCD_BUILD_SHARD_ONLY builds (export only): `units` is empty; the units lie where the pack kernel left them (by sort key)
  // and term r's run starts at units_by_key[unit_src_start[r]]
  DevBuf<u16> units_by_key;
  DevBuf<u64> unit_src_start;    // [n_terms]
  // ... round 3: EVERY id-keyed build leaves its units like that (the build's product is the ".index" image); `units` is made
  // from them when an index handle adopts the build (index_materialize_units).  shard_only: there is no image either, the
  // build can only be exported.
  bool shard_only = false;
  // host mirrors (small): keys for dcd_index_find, offsets
  std::vector<u16> h_term_chars;
  std::vector<u32> h_term_str_off;
  std::vector<u64> h_term_unit_start;
  std::vector<u64> h_term_item_start;
  bool host_ready = true;        // false: the mirrors above still have to be fetched (index_fetch_host)
  PageTable pages;
};
// downloads the host mirrors of an index a build left on the device (done on demand: a build that is only saved
// or exported never needs them)
int32_t index_fetch_host(DeviceIndex& ix);
int32_t index_materialize_units(DeviceIndex& ix, hipStream_t st);
// exact ".index" bytes (Build.cs:375-386 / Appendix A.3) of a device-resident index, generated on the device
int32_t index_make_image(const DeviceIndex& ix, hipStream_t st, DevBuf<u8>& image, u64* image_bytes);

void write_list_image(const PageTable& pt, std::vector<u8>& out);  // ".index.list", Build.cs:99-118
void utf16_to_utf8(const u16* s, size_t n, std::string& out);
bool utf8_to_utf16(const u8* p, size_t n, ustr& out);

}  // namespace dcd

// dcd_common.cpp -- error state, device check and the caching device allocator.
#include "dcd_common.h"
#include <atomic>

#include <cstdlib>
#include <chrono>
#include <map>
#include <mutex>

namespace dcd {

static thread_local std::string t_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  t_err = buf;
}
const std::string& last_error() { return t_err; }

int32_t ensure_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    DCD_FAIL(DCD_E_HIP, "no usable HIP device (hipGetDeviceCount: %s, count %d); libdocodo_hip has no CPU fallback",
             hipGetErrorString(e), n);
  }
  if (device < 0 || device >= n) DCD_FAIL(DCD_E_BADARG, "device %d out of range (have %d)", device, n);
  DCD_HIP(hipSetDevice(device));
  return DCD_OK;
}

// ---- stream-ordered caching allocator ------------------------------------------------------------
// Every handle of the library works on its own hipStreamNonBlocking stream, and the error unwinds
// (DCD_TRY / DCD_HIP early returns) destroy DevBufs whose kernels or copies may still be in flight.
// A released block therefore carries the stream it was last used on plus its position in that stream's
// release order; it is handed out again at once to the same stream (stream order covers it), and to any
// other stream only once the releasing stream has passed that point (see make_usable below).
static thread_local hipStream_t t_cur_stream = nullptr;
static thread_local bool t_have_stream = false;

StreamScope::StreamScope(hipStream_t s) : prev(t_cur_stream), prev_have(t_have_stream) {
  t_cur_stream = s;
  t_have_stream = true;
}
StreamScope::~StreamScope() {
  t_cur_stream = prev;
  t_have_stream = prev_have;
}

namespace {
struct FreeBlock {
  void* p;
  hipStream_t stream;   // stream the release was ordered on
  u64 ticket;           // position of the release in that stream's release order; 0 = nothing pending
};
struct LiveBlock {
  size_t size;
  hipStream_t stream;
  bool tagged;          // allocated inside a StreamScope
};
// Release order per stream.  A release costs no HIP call: the block only takes the next ticket of its stream.  Another
// stream gets the block once the releasing stream is known to have passed that ticket (it was seen idle, or an event
// recorded behind the ticket has completed); until then the block is passed over (see make_usable).
struct StreamState {
  u64 issued = 0;         // tickets handed out
  u64 completed = 0;      // every release with ticket <= completed has finished on the device
  hipEvent_t ev = nullptr;
  u64 ev_ticket = 0;      // ev covers the releases up to this ticket
};
struct Pool {
  std::mutex mu;
  std::multimap<size_t, FreeBlock> free_;   // rounded size -> block
  std::map<void*, LiveBlock> live;          // block -> rounded size + owner stream
  std::map<hipStream_t, StreamState> streams;
  size_t cached = 0;                        // bytes in free_
  size_t cap = 0;                           // 0 = not initialised yet
};
Pool& pool(int device) {
  static Pool pools[64];
  return pools[device & 63];
}
size_t round_size(size_t b) {
  const size_t big = 2ull << 20, small = 64ull << 10;
  size_t g = b >= (1ull << 20) ? big : small;
  return (b + g - 1) / g * g;
}
// Upper bound of the bytes the cache may hold on to (released blocks beyond it go back to HIP at once, so that a
// process that shares the GPU with another allocator -- torch in bench.py / dist.py -- is not starved):
// DOCODO_POOL_MAX_CACHED bytes, default half of the device memory.
size_t pool_cap(int device) {
  if (const char* e = getenv("DOCODO_POOL_MAX_CACHED")) {
    char* end = nullptr;
    unsigned long long v = strtoull(e, &end, 10);
    if (end != e) return (size_t)v;
  }
  size_t fr = 0, tot = 0;
  (void)hipSetDevice(device);
  if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return (size_t)64 << 30; }
  return tot / 2;
}
// Is the block usable by `want` right now?  allow_wait = false (allocation): never makes `want` wait for another stream --
// an event recorded NOW on the releasing stream would also cover everything that was queued there after the release (a
// build's side stream would wait for the whole tuple sort of the main stream); a block whose releasing stream is still
// busy is simply passed over.  allow_wait = true (release of a block that changed streams): orders `want` behind the
// other stream with an event.  Called with P.mu held.
bool make_usable(int device, Pool& P, const FreeBlock& b, hipStream_t want, bool have_want, bool allow_wait) {
  if (!b.ticket) return true;
  if (have_want && want == b.stream) return true;               // stream order covers it
  StreamState& S = P.streams[b.stream];
  if (S.completed >= b.ticket) return true;
  if (S.ev) {
    if (hipEventQuery(S.ev) == hipSuccess) {
      S.completed = S.ev_ticket;
      event_release(device, S.ev);
      S.ev = nullptr;
      if (S.completed >= b.ticket) return true;
    } else (void)hipGetLastError();
  }
  {
    const u64 issued = S.issued;                                 // an idle stream has finished everything released on it so far
    if (hipStreamQuery(b.stream) == hipSuccess) { S.completed = issued; return true; }
    (void)hipGetLastError();
  }
  if (!allow_wait) return false;
  if (!S.ev || S.ev_ticket < b.ticket) {                         // need an event that covers this release
    hipEvent_t ev = event_acquire(device);
    if (!ev || hipEventRecord(ev, b.stream) != hipSuccess) {     // (a destroyed stream: wait for the device instead)
      (void)hipGetLastError();
      if (ev) event_release(device, ev);
      if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return false; }
      S.completed = S.issued;
      return true;
    }
    if (S.ev) event_release(device, S.ev);                       // superseded: the newer event covers it
    S.ev = ev;
    S.ev_ticket = S.issued;
  }
  hipError_t w = (have_want && want) ? hipStreamWaitEvent(want, S.ev, 0) : hipEventSynchronize(S.ev);
  if (w != hipSuccess) { (void)hipGetLastError(); return false; }
  return true;
}
}  // namespace

int32_t dev_alloc(int device, size_t bytes, void** out) {
  Pool& P = pool(device);
  size_t r = round_size(bytes);
  const hipStream_t want = t_cur_stream;
  const bool have_want = t_have_stream;
  {
    std::lock_guard<std::mutex> g(P.mu);
    // accept a cached block up to 25 % larger than needed.  The pool's mutex is held while candidates are probed (an event /
    // stream query for a block whose releasing stream may still be busy): after four busy ones the allocation goes to hipMalloc
    int failed = 0;
    for (auto it = P.free_.lower_bound(r); it != P.free_.end() && it->first <= r + r / 4 && failed < 4; ++it) {
      if (!make_usable(device, P, it->second, want, have_want, false)) { failed++; continue; }
      *out = it->second.p;
      P.live[it->second.p] = LiveBlock{it->first, want, have_want};
      P.cached -= it->first;
      P.free_.erase(it);
      return DCD_OK;
    }
  }
  DCD_HIP(hipSetDevice(device));
  void* p = nullptr;
  const auto t_alloc = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&p, r);
  {
    // (a hipMalloc of gigabytes takes milliseconds on most boxes of the pool and SECONDS on some -- first_query_ready_ms of the
    //  bench line went from 23 ms to 2 s and back between runs; said on stderr so that such a figure explains itself)
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count();
    if (ms > 250.0) fprintf(stderr, "[docodo] hipMalloc(%.2f GB) took %.0f ms\n", (double)r / 1e9, ms);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    dev_pool_trim(device);
    e = hipMalloc(&p, r);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      DCD_FAIL(DCD_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", r, hipGetErrorString(e));
    }
  }
  std::lock_guard<std::mutex> g(P.mu);
  P.live[p] = LiveBlock{r, want, have_want};
  *out = p;
  return DCD_OK;
}

void dev_release(int device, void* p) {
  if (!p) return;
  Pool& P = pool(device);
  size_t size = 0;
  bool drop = false;
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.live.find(p);
    if (it == P.live.end()) return;  // double release: ignore (error, not UB)
    const LiveBlock lb = it->second;
    P.live.erase(it);
    size = lb.size;
    if (!P.cap) P.cap = pool_cap(device);
    if (P.cached + size > P.cap) drop = true;
    else {
      // The release is ordered on the releasing thread's stream when it declared one (that is the stream the block was
      // last used on), else on the stream it was allocated for.  A block that changed streams inside the library (a
      // build's index adopted by an index handle, ...) is ordered behind both.
      FreeBlock fb{p, nullptr, 0};
      const bool have_cur = t_have_stream;
      if (have_cur || lb.tagged) {
        fb.stream = have_cur ? t_cur_stream : lb.stream;
        if (have_cur && lb.tagged && lb.stream != fb.stream) {
          FreeBlock other{p, lb.stream, ++P.streams[lb.stream].issued};
          if (!make_usable(device, P, other, fb.stream, true, true)) (void)hipStreamSynchronize(lb.stream);
        }
        fb.ticket = ++P.streams[fb.stream].issued;
      }
      P.free_.emplace(size, fb);
      P.cached += size;
    }
  }
  if (drop) {   // over the cap: back to HIP (hipFree waits for the device, so pending work is safe)
    (void)hipSetDevice(device);
    (void)hipFree(p);
  }
}

size_t dev_pool_cached_bytes(int device) {
  Pool& P = pool(device);
  std::lock_guard<std::mutex> g(P.mu);
  return P.cached;
}

void dev_pool_trim(int device) {
  Pool& P = pool(device);
  std::vector<void*> v;
  {
    std::lock_guard<std::mutex> g(P.mu);
    for (auto& kv : P.free_) v.push_back(kv.second.p);
    P.free_.clear();
    P.cached = 0;
  }
  if (!v.empty()) {
    (void)hipSetDevice(device);
    for (void* p : v) (void)hipFree(p);   // implicit device synchronisation: in-flight users of the block have finished
  }
}

// ---- stream / event caches ---------------------------------------------------------------------
// Creating and destroying a HIP stream costs ~0.1-0.3 ms and an event a few microseconds: per-handle streams and the
// dozens of timing events of a build are recycled instead (idle streams per device, events per device).
namespace {
struct HandlePool {
  std::mutex mu;
  std::map<int, std::vector<hipStream_t>> streams;
  std::map<int, std::vector<hipStream_t>> streams_high;    // highest-priority streams (side chains of small kernels)
  std::map<int, std::vector<hipEvent_t>> events;
};
HandlePool& handles() { static HandlePool* h = new HandlePool(); return *h; }   // leaked on purpose (no teardown races)
}  // namespace

int32_t stream_acquire(int device, hipStream_t* out) {
  {
    HandlePool& H = handles();
    std::lock_guard<std::mutex> g(H.mu);
    auto& v = H.streams[device];
    if (!v.empty()) { *out = v.back(); v.pop_back(); return DCD_OK; }
  }
  DCD_HIP(hipSetDevice(device));
  DCD_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
  return DCD_OK;
}
// A stream that is destroyed (its synchronisation failed) leaves release tickets behind in the device cache: blocks released on
// it would be passed over for ever (hipStreamQuery on a dead handle fails) and held until dcd_pool_trim.  After a device-wide
// synchronisation everything released on the stream has finished: its state is dropped and its blocks lose their tickets.
static void pool_forget_stream(int device, hipStream_t st) {
  (void)hipDeviceSynchronize();
  (void)hipGetLastError();
  Pool& P = pool(device);
  std::lock_guard<std::mutex> g(P.mu);
  auto it = P.streams.find(st);
  if (it != P.streams.end()) {
    if (it->second.ev) event_release(device, it->second.ev);
    P.streams.erase(it);
  }
  for (auto& kv : P.free_) if (kv.second.stream == st) { kv.second.ticket = 0; kv.second.stream = nullptr; }
}
void stream_release(int device, hipStream_t st) {
  if (!st) return;
  if (hipStreamSynchronize(st) != hipSuccess) {          // never recycle a broken stream
    (void)hipGetLastError();
    pool_forget_stream(device, st);
    (void)hipStreamDestroy(st);
    return;
  }
  HandlePool& H = handles();
  std::lock_guard<std::mutex> g(H.mu);
  H.streams[device].push_back(st);
}
// ---- event configuration (dcd_common.h) ----------------------------------------------------------------------------------
static std::atomic<int> g_ev_level{-1};
static std::atomic<uint32_t> g_ev_mask{0};
EventCfg event_cfg_global() {
  int lv = g_ev_level.load(std::memory_order_relaxed);
  if (lv < 0) {
    const char* e = getenv("DOCODO_BUILD_EVENTS");
    lv = e ? atoi(e) : 2;
    if (lv < 0) lv = 0;
    g_ev_level.store(lv, std::memory_order_relaxed);
  }
  return EventCfg{lv, g_ev_mask.load(std::memory_order_relaxed)};
}
void event_cfg_set_global(int level, uint32_t mask) {
  g_ev_level.store(level < 0 ? 0 : level, std::memory_order_relaxed);
  g_ev_mask.store(mask, std::memory_order_relaxed);
}
// A few pinned host words per thread: read-backs into pinned memory are truly asynchronous (a D2H copy into pageable memory
// holds the calling thread until the data has arrived, i.e. until everything queued before it has run).
namespace {
struct PinnedCache {
  std::mutex mu;
  std::multimap<size_t, void*> free_;      // size -> block
  std::map<void*, size_t> live;
  size_t cached = 0;
};
PinnedCache& pinned_cache() { static PinnedCache c; return c; }
}  // namespace
void* pinned_stage_acquire(size_t bytes) {
  size_t r = 1 << 16;
  while (r < bytes) r <<= 1;
  PinnedCache& C = pinned_cache();
  {
    std::lock_guard<std::mutex> g(C.mu);
    auto it = C.free_.lower_bound(r);
    if (it != C.free_.end() && it->first <= 2 * r) {
      void* p = it->second;
      C.live[p] = it->first;
      C.cached -= it->first;
      C.free_.erase(it);
      return p;
    }
  }
  void* q = nullptr;
  if (hipHostMalloc(&q, r, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  std::lock_guard<std::mutex> g(C.mu);
  C.live[q] = r;
  return q;
}
void pinned_stage_release(void* p) {
  if (!p) return;
  PinnedCache& C = pinned_cache();
  size_t sz = 0;
  {
    std::lock_guard<std::mutex> g(C.mu);
    auto it = C.live.find(p);
    if (it == C.live.end()) return;
    sz = it->second;
    C.live.erase(it);
    if (C.cached + sz <= ((size_t)256 << 20)) { C.free_.emplace(sz, p); C.cached += sz; return; }   // keep at most 256 MB pinned
  }
  (void)hipHostFree(p);
}
u64* pinned_words() {
  static thread_local u64* p = nullptr;
  if (!p) {
    void* q = nullptr;
    if (hipHostMalloc(&q, 64 * sizeof(u64), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    p = (u64*)q;   // (kept for the life of the process)
  }
  return p;
}
int32_t stream_acquire_high(int device, hipStream_t* out) {
  {
    HandlePool& H = handles();
    std::lock_guard<std::mutex> g(H.mu);
    auto& v = H.streams_high[device];
    if (!v.empty()) { *out = v.back(); v.pop_back(); return DCD_OK; }
  }
  DCD_HIP(hipSetDevice(device));
  int least = 0, greatest = 0;
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
  if (const char* e = getenv("DOCODO_SIDE_PRIO")) if (atoi(e) == 0) greatest = 0;       // A/B: normal priority
  DCD_HIP(hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest));
  return DCD_OK;
}
void stream_release_high(int device, hipStream_t st) {
  if (!st) return;
  if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); pool_forget_stream(device, st); (void)hipStreamDestroy(st); return; }
  HandlePool& H = handles();
  std::lock_guard<std::mutex> g(H.mu);
  H.streams_high[device].push_back(st);
}
hipEvent_t event_acquire(int device) {
  {
    HandlePool& H = handles();
    std::lock_guard<std::mutex> g(H.mu);
    auto& v = H.events[device];
    if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}
void event_release(int device, hipEvent_t e) {
  if (!e) return;
  HandlePool& H = handles();
  std::lock_guard<std::mutex> g(H.mu);
  H.events[device].push_back(e);
}

}  // namespace dcd

extern "C" {
int32_t dcd_version(void) { return DCD_ABI_VERSION; }
uint64_t dcd_struct_size(int32_t which) {
  switch (which) {
    case 0: return sizeof(dcd_build_opts);
    case 1: return sizeof(dcd_build_stats);
    case 2: return sizeof(dcd_query_opts);
    case 3: return sizeof(dcd_results_view);
    case 4: return sizeof(dcd_term_meta);
    default: return 0;
  }
}
int32_t dcd_last_error(char* buf, size_t cap) {
  const std::string& e = dcd::last_error();
  if (buf && cap) {
    size_t n = e.size() < cap - 1 ? e.size() : cap - 1;
    memcpy(buf, e.data(), n);
    buf[n] = 0;
  }
  return (int32_t)e.size();
}
int32_t dcd_set_build_events(int32_t level, uint32_t kernel_mask) {
  if (level < 0 || level > 2 || (kernel_mask >> 8)) DCD_FAIL(DCD_E_BADARG, "dcd_set_build_events: level 0..2, mask of DCD_K_* bits");
  dcd::event_cfg_set_global(level, kernel_mask);
  return DCD_OK;
}
int32_t dcd_pool_trim(int32_t device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
  if (device < 0 || device >= n) DCD_FAIL(DCD_E_BADARG, "dcd_pool_trim: device %d out of range (have %d)", device, n);
  dcd::dev_pool_trim(device);
  return DCD_OK;
}
int32_t dcd_pool_cached_bytes(int32_t device, uint64_t* bytes) {
  if (!bytes) DCD_FAIL(DCD_E_BADARG, "bytes is NULL");
  if (device < 0 || device >= 64) DCD_FAIL(DCD_E_BADARG, "dcd_pool_cached_bytes: device %d out of range", device);
  *bytes = dcd::dev_pool_cached_bytes(device);
  return DCD_OK;
}
int32_t dcd_device_count(int32_t* n) {
  if (!n) DCD_FAIL(DCD_E_BADARG, "n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *n = 0;
    DCD_FAIL(DCD_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *n = c;
  return DCD_OK;
}
}

// dcd_common.cpp -- error state, device check and the caching device allocator.
#include "dcd_common.h"

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

namespace {
struct Pool {
  std::mutex mu;
  std::multimap<size_t, void*> free_;   // rounded size -> block
  std::map<void*, size_t> live;         // block -> rounded size
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
}  // namespace

int32_t dev_alloc(int device, size_t bytes, void** out) {
  Pool& P = pool(device);
  size_t r = round_size(bytes);
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.free_.lower_bound(r);
    // accept a cached block up to 25 % larger than needed
    if (it != P.free_.end() && it->first <= r + r / 4) {
      *out = it->second;
      P.live[it->second] = it->first;
      P.free_.erase(it);
      return DCD_OK;
    }
  }
  DCD_HIP(hipSetDevice(device));
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, r);
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
  P.live[p] = r;
  *out = p;
  return DCD_OK;
}

void dev_release(int device, void* p) {
  if (!p) return;
  Pool& P = pool(device);
  std::lock_guard<std::mutex> g(P.mu);
  auto it = P.live.find(p);
  if (it == P.live.end()) return;  // double release: ignore (error, not UB)
  P.free_.emplace(it->second, p);
  P.live.erase(it);
}

size_t dev_pool_cached_bytes(int device) {
  Pool& P = pool(device);
  std::lock_guard<std::mutex> g(P.mu);
  size_t t = 0;
  for (auto& kv : P.free_) t += kv.first;
  return t;
}

void dev_pool_trim(int device) {
  Pool& P = pool(device);
  std::vector<void*> v;
  {
    std::lock_guard<std::mutex> g(P.mu);
    for (auto& kv : P.free_) v.push_back(kv.second);
    P.free_.clear();
  }
  if (!v.empty()) {
    (void)hipSetDevice(device);
    for (void* p : v) (void)hipFree(p);
  }
}

// ---- stream / event caches ---------------------------------------------------------------------
// Creating and destroying a HIP stream costs ~0.1-0.3 ms and an event a few microseconds: per-handle streams and the
// dozens of timing events of a build are recycled instead (idle streams per device, events per device).
namespace {
struct HandlePool {
  std::mutex mu;
  std::map<int, std::vector<hipStream_t>> streams;
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
void stream_release(int device, hipStream_t st) {
  if (!st) return;
  if (hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamDestroy(st); return; }   // never recycle a broken stream
  HandlePool& H = handles();
  std::lock_guard<std::mutex> g(H.mu);
  H.streams[device].push_back(st);
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
int32_t dcd_version(void) { return 1; }
int32_t dcd_last_error(char* buf, size_t cap) {
  const std::string& e = dcd::last_error();
  if (buf && cap) {
    size_t n = e.size() < cap - 1 ? e.size() : cap - 1;
    memcpy(buf, e.data(), n);
    buf[n] = 0;
  }
  return (int32_t)e.size();
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

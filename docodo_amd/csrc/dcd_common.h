// dcd_common.h -- shared host-side plumbing of libdocodo_hip: status/error handling, a caching
// device allocator (no hipMalloc inside timed regions), per-handle streams, HIP-event timers.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/docodo_hip.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

namespace dcd {

// ---- thread-local last error --------------------------------------------------------------
void set_error(const char* fmt, ...);
const std::string& last_error();

struct Status {
  int32_t code;
};

#define DCD_FAIL(code_, ...)          \
  do {                                \
    ::dcd::set_error(__VA_ARGS__);    \
    return (code_);                   \
  } while (0)

#define DCD_HIP(expr)                                                                        \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      ::dcd::set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_),     \
                       __FILE__, __LINE__, #expr);                                           \
      return DCD_E_HIP;                                                                      \
    }                                                                                        \
  } while (0)

#define DCD_TRY(expr)            \
  do {                           \
    int32_t s_ = (expr);         \
    if (s_ != DCD_OK) return s_; \
  } while (0)

// ---- caching device allocator ---------------------------------------------------------------
// Blocks are rounded up to 2 MiB (>= 1 MiB requests) or 64 KiB and kept in a per-device free list
// when released, so that repeated builds / query batches of the same shape do not call hipMalloc.
// The allocator is stream-ordered: an entry point declares the stream its work runs on with a StreamScope; blocks
// allocated or released inside the scope are tagged with that stream, a released block carries an event recorded on
// it, and another stream gets the block only once that event has completed (see dcd_common.cpp).
struct StreamScope {
  hipStream_t prev;
  bool prev_have;
  explicit StreamScope(hipStream_t s);
  ~StreamScope();
  StreamScope(const StreamScope&) = delete;
  StreamScope& operator=(const StreamScope&) = delete;
};
int32_t dev_alloc(int device, size_t bytes, void** out);
void dev_release(int device, void* p);   // back to the pool
void dev_pool_trim(int device);          // hipFree everything cached
size_t dev_pool_cached_bytes(int device);   // bytes held by the cache (released blocks not yet returned to HIP)

// idle HIP streams (hipStreamNonBlocking) and events are recycled per device instead of created per handle
int32_t stream_acquire(int device, hipStream_t* out);
void stream_release(int device, hipStream_t st);     // synchronises the stream, then parks it
// the same for streams of the highest priority: a chain of small kernels that runs beside the big kernels of another
// stream (the vocabulary sort beside the tuple sort) and must not queue behind them
// Pinned staging for host -> device uploads of library-owned host data (page tables, ...).  hipMemcpyAsync from PAGEABLE memory
// makes the runtime lock the pages in place, and freeing such memory afterwards (a std::vector of 8 MB is an mmap region) costs a
// driver round trip that stalls the next upload: round 3 measured 20 ms in front of every build with explicit page offsets.
// pinned_stage_acquire hands out a cached hipHostMalloc block of at least `bytes`; release it once the copy has completed.
void* pinned_stage_acquire(size_t bytes);
void pinned_stage_release(void* p);
// host -> device copies of a scope through pinned staging; the blocks go back when the scope ends (the owner has synchronised
// the stream by then)
struct StagedUploads {
  std::vector<void*> blocks;
  StagedUploads() {}
  StagedUploads(const StagedUploads&) = delete;
  StagedUploads& operator=(const StagedUploads&) = delete;
  ~StagedUploads() { for (void* p : blocks) pinned_stage_release(p); }
  int32_t copy(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (!bytes) return DCD_OK;
    void* p = pinned_stage_acquire(bytes);
    if (!p) return DCD_E_NOMEM;
    blocks.push_back(p);
    memcpy(p, src, bytes);
    return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, st) == hipSuccess ? DCD_OK : DCD_E_HIP;
  }
};
u64* pinned_words();     // 64 pinned host words of the calling thread (nullptr on failure): targets of asynchronous read-backs
int32_t stream_acquire_high(int device, hipStream_t* out);
void stream_release_high(int device, hipStream_t st);
hipEvent_t event_acquire(int device);                // nullptr on failure
void event_release(int device, hipEvent_t e);        // the event must not be pending

// RAII device buffer bound to the pool
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;  // elements
  int device = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n), device(o.device) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { reset(); p = o.p; n = o.n; device = o.device; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { reset(); }
  void reset() {
    if (p) dev_release(device, p);
    p = nullptr;
    n = 0;
  }
  int32_t alloc(int dev, size_t count) {
    reset();
    device = dev;
    void* q = nullptr;
    int32_t s = dev_alloc(dev, (count ? count : 1) * sizeof(T), &q);
    if (s != DCD_OK) return s;
    p = (T*)q;
    n = count;
    return DCD_OK;
  }
  size_t bytes() const { return n * sizeof(T); }
};

// ---- HIP event timer ------------------------------------------------------------------------
// How many events a build records (every record is a marker packet the stream stops at: ~5 us each, ~30 per build with
// everything on: 0.12 ms of a 4 ms build).  Process-wide setting, dcd_set_build_events(level, kernel_mask) / DOCODO_BUILD_EVENTS:
// level 2 = stage timers + all per-kernel probes (default), 1 = stage timers, 0 = no stage timers (the ms_* stage fields of the
// stats are then 0); kernel_mask: the probes of kernel DCD_K_* k are recorded if level >= 2 or bit k of the mask is set.
// A build takes its setting once, when dcd_build_finish starts (EventScope), so that one build is consistent in itself.
struct EventCfg { int level; uint32_t mask; };
EventCfg event_cfg_global();
void event_cfg_set_global(int level, uint32_t mask);
inline EventCfg& event_cfg_tl() { static thread_local EventCfg c{-1, 0}; return c; }
struct EventScope {      // the calling thread uses a snapshot of the global setting until the scope ends
  EventCfg saved;
  EventScope() : saved(event_cfg_tl()) { event_cfg_tl() = event_cfg_global(); }
  ~EventScope() { event_cfg_tl() = saved; }
};
inline int event_level() { const EventCfg& c = event_cfg_tl(); return c.level >= 0 ? c.level : event_cfg_global().level; }
inline bool event_probe_on(int kernel) {
  const EventCfg& t = event_cfg_tl();
  const EventCfg c = t.level >= 0 ? t : event_cfg_global();
  return c.level >= 2 || ((c.mask >> kernel) & 1u);
}
struct EvTimer {
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s = nullptr;
  int device = 0;
  // A timer may share a boundary with another one instead of recording its own event (every record is a marker packet the
  // stream stops at): pa / pb point at the events ms() reads.
  const hipEvent_t* pa = &a;
  const hipEvent_t* pb = &b;
  EvTimer() = default;
  EvTimer(const EvTimer&) = delete;
  EvTimer& operator=(const EvTimer&) = delete;
  int32_t init(hipStream_t st, int dev = 0) {
    s = st;
    device = dev;
    if (!a) a = event_acquire(dev);
    if (!b) b = event_acquire(dev);
    if (!a || !b) DCD_FAIL(DCD_E_HIP, "hipEventCreate failed");
    return DCD_OK;
  }
  int32_t start() { pa = &a; if (event_level() >= 1) DCD_HIP(hipEventRecord(a, s)); return DCD_OK; }
  int32_t stop() { pb = &b; if (event_level() >= 1) DCD_HIP(hipEventRecord(b, s)); return DCD_OK; }
  void start_at_start_of(const EvTimer& o) { pa = o.pa; }
  void start_at_stop_of(const EvTimer& o) { pa = o.pb; }
  void stop_at_stop_of(const EvTimer& o) { pb = o.pb; }
  float ms() {  // call after the stream was synchronised (and while the timers it shares boundaries with are alive)
    if (event_level() < 1) return 0.f;
    float t = 0;
    if (hipEventElapsedTime(&t, *pa, *pb) != hipSuccess) return -1.f;
    return t;
  }
  ~EvTimer() {
    event_release(device, a);
    event_release(device, b);
  }
};

inline u64 div_up(u64 a, u64 b) { return (a + b - 1) / b; }
inline u32 log2_ceil(u64 v) { u32 r = 0; while ((1ull << r) < v) r++; return r; }

int32_t ensure_device(int device);  // DCD_E_HIP when no usable device

}  // namespace dcd

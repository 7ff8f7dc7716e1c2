// dcd_comm.cpp -- the exchange step of a doc-range sharded build INSIDE the library (round 4).
//
// The reference's CreateAsync runs its parallel index tasks and merges their temp indexes in one call (Index.cs:464-488,
// 537-573).  The multi-GPU form of that -- every GPU indexes a document range with global coordinates, per-term posting segments
// travel to the term's owner, the owner concatenates them in shard order (IndexSequence.Shift + AddRange, Index.cs:665-674) -- used
// to need a host that moved the export buffers itself (docodo_amd/dist.py over torch.distributed).  Here it is three C entry
// points a C# host can P/Invoke:
//   dcd_comm_create / dcd_comm_create_all     one communicator per GPU (per process, or all of them in one process)
//   dcd_build_merge_distributed               export -> balanced owners -> all-to-all-v -> merge: this rank's hash partition
//   dcd_index_allgather / dcd_index_gather    every rank (one rank) ends with the complete index
// Transport RCCL: ncclSend / ncclRecv groups over xGMI on the communicator's own stream (the library is loaded lazily with dlopen,
// so that a host without RCCL -- or one that brought its own copy, as PyTorch does -- still loads libdocodo_hip).  Transport HOST:
// the same calls with the buffers staged through POSIX shared memory -- for tests and rehearsals (several ranks on ONE GPU, which
// RCCL refuses), never for measurements.
// Per step: ONE small all-gather (split sizes + MaxCoord), then meta, keys and units as three all-to-all-v on the communication
// stream; dcd_merge_begin (vocabulary, ordinal ranks, output layout: everything that does not touch a posting unit) runs as soon
// as meta and keys have landed, i.e. while the units are still on the fabric; dcd_merge_finish is the one copy kernel behind it.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

#include "dcd_index_int.h"

using namespace dcd;

namespace {

// ---- RCCL through dlopen ------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(nccl_uid*) = nullptr;
  int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
  int (*CommInitAll)(nccl_comm*, int, const int*) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_UINT8 = 1;   // ncclUint8 (rccl.h: ncclInt8 = 0, ncclUint8 = 1)

RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // (a copy that is already in the process -- PyTorch loads its own -- is found first: same SONAME, one RCCL per process)
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) return;
    auto sym = [&](const char* n) { return dlsym(api.lib, n); };
    api.GetUniqueId = (int (*)(nccl_uid*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(nccl_comm*, int, nccl_uid, int))sym("ncclCommInitRank");
    api.CommInitAll = (int (*)(nccl_comm*, int, const int*))sym("ncclCommInitAll");
    api.CommDestroy = (int (*)(nccl_comm))sym("ncclCommDestroy");
    api.Send = (int (*)(const void*, size_t, int, int, nccl_comm, hipStream_t))sym("ncclSend");
    api.Recv = (int (*)(void*, size_t, int, int, nccl_comm, hipStream_t))sym("ncclRecv");
    api.AllGather = (int (*)(const void*, void*, size_t, int, nccl_comm, hipStream_t))sym("ncclAllGather");
    api.GroupStart = (int (*)())sym("ncclGroupStart");
    api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommInitAll || !api.CommDestroy || !api.Send || !api.Recv || !api.AllGather ||
        !api.GroupStart || !api.GroupEnd) { dlclose(api.lib); api.lib = nullptr; }
  });
  return api.lib ? &api : nullptr;
}

#define DCD_NCCL(expr)                                                                                      \
  do {                                                                                                      \
    int r_ = (expr);                                                                                        \
    if (r_ != 0) {                                                                                          \
      RcclApi* a_ = rccl_api();                                                                             \
      ::dcd::set_error("RCCL error %d (%s) at %s:%d: %s", r_, (a_ && a_->GetErrorString) ? a_->GetErrorString(r_) : "?", __FILE__, \
                       __LINE__, #expr);                                                                    \
      return DCD_E_RCCL;                                                                                    \
    }                                                                                                       \
  } while (0)

// ---- host transport: POSIX shared memory ---------------------------------------------------------------------------
constexpr int HOST_MAX_RANKS = 64;
constexpr double HOST_TIMEOUT_S = 180.0;
struct HostCtl {                       // "/dcdc_<id>": zero-filled by the kernel when the first rank creates it
  std::atomic<uint32_t> count, gen;
  std::atomic<uint64_t> size[HOST_MAX_RANKS];     // payload bytes a rank has published for the current operation
};
struct HostShm {
  std::string prefix;                  // "/dcdc_<24 hex digits>"
  HostCtl* ctl = nullptr;
  int n = 0, rank = 0;
  // this rank's payload segment "<prefix>_<rank>" (grown on demand)
  int fd = -1;
  u8* map = nullptr;
  size_t cap = 0;
  ~HostShm() {
    if (map) munmap(map, cap);
    if (fd >= 0) close(fd);
    if (ctl) munmap(ctl, sizeof(HostCtl));
  }
};

int32_t host_attach(HostShm& h) {
  const int fd = shm_open(h.prefix.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) DCD_FAIL(DCD_E_IO, "shm_open(%s) failed", h.prefix.c_str());
  if (ftruncate(fd, sizeof(HostCtl)) != 0) { close(fd); DCD_FAIL(DCD_E_IO, "ftruncate(%s) failed", h.prefix.c_str()); }
  void* p = mmap(nullptr, sizeof(HostCtl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) DCD_FAIL(DCD_E_IO, "mmap(%s) failed", h.prefix.c_str());
  h.ctl = static_cast<HostCtl*>(p);
  return DCD_OK;
}
int32_t host_barrier(HostShm& h) {
  const uint32_t g = h.ctl->gen.load(std::memory_order_acquire);
  if (h.ctl->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)h.n) {
    h.ctl->count.store(0, std::memory_order_relaxed);
    h.ctl->gen.store(g + 1, std::memory_order_release);
    return DCD_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (h.ctl->gen.load(std::memory_order_acquire) == g) {
    if (++spins < 200) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(200));
    if ((spins & 1023u) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > HOST_TIMEOUT_S)
      DCD_FAIL(DCD_E_RCCL, "host transport: rank %d waited %.0f s for its peers", h.rank, HOST_TIMEOUT_S);
  }
  return DCD_OK;
}
std::string host_seg_name(const HostShm& h, int r) { return h.prefix + "_" + std::to_string(r); }
// publishes `bytes` of this rank and lets visit(src, data, bytes) look at every rank's payload (collective)
int32_t host_exchange(HostShm& h, size_t bytes, const std::function<int32_t(u8* dst)>& fill,
                      const std::function<int32_t(int src, const u8* data, size_t n)>& visit) {
  if (bytes > h.cap || !h.map) {
    if (h.map) { munmap(h.map, h.cap); h.map = nullptr; }
    if (h.fd < 0) {
      h.fd = shm_open(host_seg_name(h, h.rank).c_str(), O_CREAT | O_RDWR, 0600);
      if (h.fd < 0) DCD_FAIL(DCD_E_IO, "shm_open(%s) failed", host_seg_name(h, h.rank).c_str());
    }
    size_t cap = 1 << 16;
    while (cap < bytes) cap <<= 1;
    if (ftruncate(h.fd, (off_t)cap) != 0) DCD_FAIL(DCD_E_IO, "host transport: no shared memory for %zu bytes", cap);
    void* p = mmap(nullptr, cap, PROT_READ | PROT_WRITE, MAP_SHARED, h.fd, 0);
    if (p == MAP_FAILED) DCD_FAIL(DCD_E_IO, "host transport: mmap of %zu bytes failed", cap);
    h.map = static_cast<u8*>(p);
    h.cap = cap;
  }
  int32_t st = bytes ? fill(h.map) : DCD_OK;
  h.ctl->size[h.rank].store(st == DCD_OK ? bytes : ~0ull, std::memory_order_release);
  DCD_TRY(host_barrier(h));                                    // every payload is in place
  for (int s = 0; s < h.n && st == DCD_OK; s++) {
    const uint64_t n = h.ctl->size[s].load(std::memory_order_acquire);
    if (n == ~0ull) { set_error("host transport: rank %d failed to publish", s); st = DCD_E_RCCL; break; }
    if (s == h.rank) { st = visit(s, h.map, n); continue; }
    if (!n) { st = visit(s, nullptr, 0); continue; }
    const int fd = shm_open(host_seg_name(h, s).c_str(), O_RDONLY, 0600);
    if (fd < 0) { set_error("host transport: cannot open the segment of rank %d", s); st = DCD_E_IO; break; }
    void* p = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { set_error("host transport: cannot map the segment of rank %d", s); st = DCD_E_IO; break; }
    st = visit(s, static_cast<const u8*>(p), n);
    munmap(p, n);
  }
  const int32_t sb = host_barrier(h);                          // every rank has read: the segments may be rewritten
  return st != DCD_OK ? st : sb;
}

}  // namespace

struct dcd_comm {
  int n = 1, rank = 0, device = 0, transport = DCD_COMM_RCCL;
  hipStream_t st = nullptr;            // communication stream
  nccl_comm nccl = nullptr;
  std::unique_ptr<HostShm> shm;
  std::mutex mu;                       // one collective at a time per communicator
};

namespace {

// ---- collectives ---------------------------------------------------------------------------------------------------
// small host blobs: all[r * bytes ...] = rank r's `mine`
int32_t comm_allgather_host(dcd_comm* c, const void* mine, size_t bytes, void* all) {
  if (c->n == 1) { memcpy(all, mine, bytes); return DCD_OK; }
  if (c->transport == DCD_COMM_HOST) {
    return host_exchange(*c->shm, bytes, [&](u8* dst) { memcpy(dst, mine, bytes); return DCD_OK; },
                         [&](int src, const u8* data, size_t n) {
                           if (n != bytes) DCD_FAIL(DCD_E_RCCL, "all-gather: rank %d published %zu bytes, %zu expected", src, n, bytes);
                           memcpy(static_cast<u8*>(all) + (size_t)src * bytes, data, bytes);
                           return DCD_OK;
                         });
  }
  RcclApi* api = rccl_api();
  StreamScope ss_(c->st);
  DevBuf<u8> d;
  DCD_TRY(d.alloc(c->device, (size_t)c->n * bytes));
  void* pin = pinned_stage_acquire((size_t)c->n * bytes);
  if (!pin) DCD_FAIL(DCD_E_NOMEM, "no pinned host memory");
  struct Rel { void* p; ~Rel() { pinned_stage_release(p); } } rel{pin};
  memcpy(static_cast<u8*>(pin) + (size_t)c->rank * bytes, mine, bytes);
  DCD_HIP(hipMemcpyAsync(d.p + (size_t)c->rank * bytes, static_cast<u8*>(pin) + (size_t)c->rank * bytes, bytes, hipMemcpyHostToDevice, c->st));
  DCD_NCCL(api->AllGather(d.p + (size_t)c->rank * bytes, d.p, bytes, NCCL_UINT8, c->nccl, c->st));
  DCD_HIP(hipMemcpyAsync(pin, d.p, (size_t)c->n * bytes, hipMemcpyDeviceToHost, c->st));
  DCD_HIP(hipStreamSynchronize(c->st));
  memcpy(all, pin, (size_t)c->n * bytes);
  return DCD_OK;
}

// device buffers: send[soff[p] .. + scnt[p]) goes to rank p, recv[roff[s] .. + rcnt[s]) comes from rank s (bytes); queued on the
// communication stream (RCCL) or completed on return (HOST)
int32_t comm_alltoallv(dcd_comm* c, const u8* send, const u64* soff, const u64* scnt, u8* recv, const u64* roff, const u64* rcnt) {
  if (c->transport == DCD_COMM_HOST && c->n > 1) {
    const u64 total = soff[c->n - 1] + scnt[c->n - 1];
    const size_t hdr = 2 * sizeof(u64) * (size_t)c->n;
    return host_exchange(*c->shm, hdr + total,
                         [&](u8* dst) {
                           memcpy(dst, soff, sizeof(u64) * c->n);
                           memcpy(dst + sizeof(u64) * c->n, scnt, sizeof(u64) * c->n);
                           if (total) DCD_HIP(hipMemcpy(dst + hdr, send, total, hipMemcpyDeviceToHost));
                           return DCD_OK;
                         },
                         [&](int src, const u8* data, size_t n) {
                           if (n < hdr) DCD_FAIL(DCD_E_RCCL, "all-to-all: short payload from rank %d", src);
                           const u64* so = reinterpret_cast<const u64*>(data);
                           const u64* sc = so + c->n;
                           if (sc[c->rank] != rcnt[src]) DCD_FAIL(DCD_E_RCCL, "all-to-all: rank %d sends %llu bytes, %llu expected", src,
                                                                   (unsigned long long)sc[c->rank], (unsigned long long)rcnt[src]);
                           if (rcnt[src]) DCD_HIP(hipMemcpy(recv + roff[src], data + hdr + so[c->rank], rcnt[src], hipMemcpyHostToDevice));
                           return DCD_OK;
                         });
  }
  if (scnt[c->rank]) DCD_HIP(hipMemcpyAsync(recv + roff[c->rank], send + soff[c->rank], scnt[c->rank], hipMemcpyDeviceToDevice, c->st));
  if (c->n == 1) return DCD_OK;
  RcclApi* api = rccl_api();
  DCD_NCCL(api->GroupStart());
  for (int p = 0; p < c->n; p++) {
    if (p == c->rank) continue;
    if (scnt[p]) DCD_NCCL(api->Send(send + soff[p], scnt[p], NCCL_UINT8, p, c->nccl, c->st));
    if (rcnt[p]) DCD_NCCL(api->Recv(recv + roff[p], rcnt[p], NCCL_UINT8, p, c->nccl, c->st));
  }
  DCD_NCCL(api->GroupEnd());
  return DCD_OK;
}

int32_t comm_create_common(dcd_comm* c) {
  if (c->device < 0) return DCD_OK;                           // host-only communicator (dcd_comm_allgather): no device, no stream
  DCD_TRY(ensure_device(c->device));
  DCD_HIP(hipSetDevice(c->device));
  DCD_TRY(stream_acquire(c->device, &c->st));
  return DCD_OK;
}

std::string host_prefix(const uint8_t id[DCD_COMM_ID_BYTES]) {
  static const char* hx = "0123456789abcdef";
  std::string s = "/dcdc_";
  for (int i = 0; i < 12; i++) { s += hx[id[i] >> 4]; s += hx[id[i] & 15]; }
  return s;
}

// ---- owner balance: pure integer function of the gathered heavy-term lists (every rank computes the same table) --------
struct HeavyTerm { u64 fp, units; };

}  // namespace

extern "C" {

// Heavy terms are taken largest first (ties: smaller fingerprint) and given to the owner with the least heavy units so far (ties:
// lower rank); the LIGHT terms -- everything else, spread by hash -- then fill every owner up to the mean: owner o gets the slice
// of the 32-bit hash space proportional to max(0, mean - heavy_o).  (docodo_amd.dist.assign_heavy is the same function in Python;
// tests/test_dist_cpu.py compares the two.)
int32_t dcd_balance_owners(uint32_t n_owners, uint32_t n_lists, const uint64_t* list_off, const uint64_t* fp, const uint64_t* units,
                           const uint64_t* totals, uint64_t* out_fp, uint32_t* out_owner, size_t cap, size_t* n_out, uint32_t* light_thr) {
  if (!n_owners || !list_off || !totals || !n_out || !light_thr) DCD_FAIL(DCD_E_BADARG, "dcd_balance_owners: bad argument");
  std::map<u64, u64> sum;
  for (uint32_t l = 0; l < n_lists; l++)
    for (u64 i = list_off[l]; i < list_off[l + 1]; i++) sum[fp[i]] += units[i];
  unsigned __int128 total = 0;
  for (uint32_t l = 0; l < n_lists; l++) total += totals[l];
  std::vector<HeavyTerm> hv;
  hv.reserve(sum.size());
  for (auto& kv : sum) hv.push_back(HeavyTerm{kv.first, kv.second});
  std::sort(hv.begin(), hv.end(), [](const HeavyTerm& a, const HeavyTerm& b) { return a.units != b.units ? a.units > b.units : a.fp < b.fp; });
  std::vector<unsigned __int128> load(n_owners, 0);
  std::map<u64, uint32_t> owner;
  for (const HeavyTerm& t : hv) {
    uint32_t o = 0;
    for (uint32_t i = 1; i < n_owners; i++) if (load[i] < load[o]) o = i;
    owner[t.fp] = o;
    load[o] += t.units;
  }
  // light capacity per owner: what is missing to the mean (an owner whose heavy terms already exceed it gets no light term)
  std::vector<unsigned __int128> capv(n_owners);
  unsigned __int128 tot = 0;
  for (uint32_t o = 0; o < n_owners; o++) { capv[o] = total > (unsigned __int128)n_owners * load[o] ? total - (unsigned __int128)n_owners * load[o] : 0; tot += capv[o]; }
  if (tot == 0) { for (auto& v : capv) v = 1; tot = n_owners; }
  unsigned __int128 cum = 0;
  for (uint32_t o = 0; o < n_owners; o++) {
    cum += capv[o];
    const unsigned __int128 q = (cum << 32) / tot;
    light_thr[o] = q > 0 ? (uint32_t)std::min<unsigned __int128>(q - 1, 0xFFFFFFFFu) : 0u;
  }
  light_thr[n_owners - 1] = 0xFFFFFFFFu;
  for (int o = (int)n_owners - 2; o >= 0; o--) light_thr[o] = std::min(light_thr[o], light_thr[o + 1]);   // (an owner without capacity: an empty slice)
  *n_out = owner.size();
  if (owner.size() > cap) return DCD_E_OVERFLOW;
  size_t k = 0;
  for (auto& kv : owner) { if (out_fp) out_fp[k] = kv.first; if (out_owner) out_owner[k] = kv.second; k++; }
  return DCD_OK;
}

int32_t dcd_comm_unique_id(int32_t transport, uint8_t id[DCD_COMM_ID_BYTES]) {
  if (!id) DCD_FAIL(DCD_E_BADARG, "dcd_comm_unique_id: NULL argument");
  if (transport == DCD_COMM_RCCL) {
    RcclApi* api = rccl_api();
    if (!api) DCD_FAIL(DCD_E_RCCL, "librccl.so.1 is not available");
    nccl_uid u;
    DCD_NCCL(api->GetUniqueId(&u));
    static_assert(sizeof(u) == DCD_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id, &u, sizeof u);
    return DCD_OK;
  }
  if (transport != DCD_COMM_HOST) DCD_FAIL(DCD_E_BADARG, "dcd_comm_unique_id: unknown transport %d", transport);
  memset(id, 0, DCD_COMM_ID_BYTES);
  FILE* f = fopen("/dev/urandom", "rb");
  size_t got = f ? fread(id, 1, 16, f) : 0;
  if (f) fclose(f);
  if (got != 16) {
    const u64 t = (u64)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((u64)getpid() << 32);
    memcpy(id, &t, 8);
  }
  return DCD_OK;
}

int32_t dcd_comm_create(int32_t n_ranks, int32_t rank, const uint8_t id[DCD_COMM_ID_BYTES], int32_t device, int32_t transport, dcd_comm** out) {
  if (!out || n_ranks < 1 || rank < 0 || rank >= n_ranks || (!id && n_ranks > 1)) DCD_FAIL(DCD_E_BADARG, "dcd_comm_create: bad argument");
  std::unique_ptr<dcd_comm> c(new dcd_comm());
  c->n = n_ranks; c->rank = rank; c->device = device; c->transport = transport;
  DCD_TRY(comm_create_common(c.get()));
  if (transport == DCD_COMM_RCCL) {
    if (n_ranks > 1 || id) {
      RcclApi* api = rccl_api();
      if (!api) DCD_FAIL(DCD_E_RCCL, "librccl.so.1 is not available");
      nccl_uid u;
      if (id) memcpy(&u, id, sizeof u); else DCD_NCCL(api->GetUniqueId(&u));
      DCD_NCCL(api->CommInitRank(&c->nccl, n_ranks, u, rank));
    }
  } else if (transport == DCD_COMM_HOST) {
    if (n_ranks > HOST_MAX_RANKS) DCD_FAIL(DCD_E_BADARG, "host transport: at most %d ranks", HOST_MAX_RANKS);
    if (n_ranks > 1) {
      c->shm.reset(new HostShm());
      c->shm->prefix = host_prefix(id);
      c->shm->n = n_ranks; c->shm->rank = rank;
      DCD_TRY(host_attach(*c->shm));
    }
  } else DCD_FAIL(DCD_E_BADARG, "dcd_comm_create: unknown transport %d", transport);
  *out = c.release();
  return DCD_OK;
}

int32_t dcd_comm_create_all(int32_t n, const int32_t* devices, int32_t transport, dcd_comm** out) {
  if (!out || n < 1 || !devices) DCD_FAIL(DCD_E_BADARG, "dcd_comm_create_all: bad argument");
  for (int i = 0; i < n; i++) out[i] = nullptr;
  auto fail = [&](int32_t st) { for (int i = 0; i < n; i++) { dcd_comm_free(out[i]); out[i] = nullptr; } return st; };
  if (transport == DCD_COMM_HOST) {
    uint8_t id[DCD_COMM_ID_BYTES];
    DCD_TRY(dcd_comm_unique_id(DCD_COMM_HOST, id));
    for (int i = 0; i < n; i++) { const int32_t st = dcd_comm_create(n, i, id, devices[i], DCD_COMM_HOST, &out[i]); if (st != DCD_OK) return fail(st); }
    return DCD_OK;
  }
  if (transport != DCD_COMM_RCCL) DCD_FAIL(DCD_E_BADARG, "dcd_comm_create_all: unknown transport %d", transport);
  RcclApi* api = rccl_api();
  if (!api && n > 1) DCD_FAIL(DCD_E_RCCL, "librccl.so.1 is not available");
  std::vector<nccl_comm> comms(n, nullptr);
  std::vector<int> devs(devices, devices + n);
  if (n > 1) DCD_NCCL(api->CommInitAll(comms.data(), n, devs.data()));
  for (int i = 0; i < n; i++) {
    std::unique_ptr<dcd_comm> c(new dcd_comm());
    c->n = n; c->rank = i; c->device = devices[i]; c->transport = DCD_COMM_RCCL; c->nccl = comms[i];
    const int32_t st = comm_create_common(c.get());
    out[i] = c.release();
    if (st != DCD_OK) return fail(st);
  }
  return DCD_OK;
}

void dcd_comm_free(dcd_comm* c) {
  if (!c) return;
  if (c->st) { (void)hipSetDevice(c->device); stream_release(c->device, c->st); }
  if (c->nccl) { RcclApi* api = rccl_api(); if (api) (void)api->CommDestroy(c->nccl); }
  if (c->shm) {
    // (every collective ends behind a barrier, so nobody reads this rank's segment any more; the barrier here only keeps rank 0
    // from removing the control block before a slow rank has attached)
    (void)host_barrier(*c->shm);
    shm_unlink(host_seg_name(*c->shm, c->rank).c_str());
    if (c->rank == 0) shm_unlink(c->shm->prefix.c_str());
  }
  delete c;
}

int32_t dcd_comm_rank(dcd_comm* c, int32_t* rank, int32_t* n_ranks) {
  if (!c) DCD_FAIL(DCD_E_BADARG, "dcd_comm_rank: NULL argument");
  if (rank) *rank = c->rank;
  if (n_ranks) *n_ranks = c->n;
  return DCD_OK;
}

int32_t dcd_comm_allgather(dcd_comm* c, const void* mine, size_t bytes, void* all) {
  if (!c || (!mine && bytes) || (!all && bytes)) DCD_FAIL(DCD_E_BADARG, "dcd_comm_allgather: NULL argument");
  std::lock_guard<std::mutex> g(c->mu);
  if (!bytes) return DCD_OK;
  if (c->device >= 0) DCD_HIP(hipSetDevice(c->device));
  else if (c->transport != DCD_COMM_HOST && c->n > 1) DCD_FAIL(DCD_E_BADARG, "dcd_comm_allgather: an RCCL communicator needs a device");
  return comm_allgather_host(c, mine, bytes, all);
}

}  // extern "C"

namespace {

// one shard = a finished build or an index handle (several segments of a rank merged on the device)
struct ShardSrc {
  dcd_build* b = nullptr;
  dcd_index* ix = nullptr;
  int32_t heavy(u64 min_units, u64* fp, u64* un, size_t cap, size_t* n) const {
    return b ? dcd_build_export_heavy(b, min_units, fp, un, cap, n) : dcd_index_export_heavy(ix, min_units, fp, un, cap, n);
  }
  int32_t plan(uint32_t G, u64* tc, u64* kc, u64* uc) const { return b ? dcd_build_export_plan(b, G, tc, kc, uc) : dcd_index_export_plan(ix, G, tc, kc, uc); }
  int32_t plan_balanced(uint32_t G, const u64* ofp, const uint32_t* oow, size_t n, const uint32_t* thr, u64* tc, u64* kc, u64* uc) const {
    return b ? dcd_build_export_plan_balanced(b, G, ofp, oow, n, thr, tc, kc, uc) : dcd_index_export_plan_balanced(ix, G, ofp, oow, n, thr, tc, kc, uc);
  }
  int32_t fill(void* meta, u16* keys, u16* units) const { return b ? dcd_build_export_fill(b, meta, keys, units) : dcd_index_export_fill(ix, meta, keys, units); }
  int32_t sizes(u64* n_units, u64* max_coord) const {
    if (b) { dcd_build_stats s; DCD_TRY(dcd_build_get_stats(b, &s)); *n_units = s.n_units; *max_coord = s.max_coord; return DCD_OK; }
    DCD_TRY(dcd_index_n_units(ix, n_units));
    return dcd_index_max_coord(ix, max_coord);
  }
};

constexpr u64 META_BYTES = sizeof(dcd_term_meta);

u64 sum_of(const std::vector<u64>& v) { u64 s = 0; for (u64 x : v) s += x; return s; }
void offsets_of(const std::vector<u64>& cnt, std::vector<u64>& off) { off.resize(cnt.size()); u64 a = 0; for (size_t i = 0; i < cnt.size(); i++) { off[i] = a; a += cnt[i]; } }

// export (per-owner groups) -> all-to-all-v -> merge; `single_dst` >= 0: everything goes to that rank (gather)
int32_t exchange_and_merge(const ShardSrc& src, dcd_comm* c, bool balanced, int single_dst, dcd_index** out) {
  const int G = c->n;
  DCD_HIP(hipSetDevice(c->device));
  StreamScope ss_(c->st);
  u64 my_units = 0, my_max = 0;
  DCD_TRY(src.sizes(&my_units, &my_max));
  std::vector<u64> tc(G, 0), kc(G, 0), uc(G, 0);
  if (single_dst >= 0) {
    u64 t1 = 0, k1 = 0, u1 = 0;
    DCD_TRY(src.plan(1, &t1, &k1, &u1));
    tc[single_dst] = t1; kc[single_dst] = k1; uc[single_dst] = u1;
  } else if (balanced && G > 1) {
    // ONE small all-gather: every rank's heavy terms (at least 1 / (8 G) of its shard's posting units: at most 8 G of them)
    const size_t cap = 8 * (size_t)G;
    std::vector<u64> fp(cap, 0), un(cap, 0);
    size_t nh = 0;
    DCD_TRY(src.heavy(std::max<u64>(1, div_up(my_units, cap)), fp.data(), un.data(), cap, &nh));
    std::vector<u64> mine(2 + 2 * cap, 0), all((2 + 2 * cap) * (size_t)G, 0);
    mine[0] = nh; mine[1] = my_units;
    std::copy(fp.begin(), fp.begin() + nh, mine.begin() + 2);
    std::copy(un.begin(), un.begin() + nh, mine.begin() + 2 + cap);
    DCD_TRY(comm_allgather_host(c, mine.data(), mine.size() * 8, all.data()));
    std::vector<u64> loff(G + 1, 0), lfp, lun, totals(G, 0);
    for (int r = 0; r < G; r++) {
      const u64* m = all.data() + (size_t)r * mine.size();
      const u64 n = std::min<u64>(m[0], cap);
      totals[r] = m[1];
      lfp.insert(lfp.end(), m + 2, m + 2 + n);
      lun.insert(lun.end(), m + 2 + cap, m + 2 + cap + n);
      loff[r + 1] = lfp.size();
    }
    std::vector<u64> ofp(lfp.size() + 1);
    std::vector<uint32_t> oow(lfp.size() + 1), thr(G);
    size_t n_over = 0;
    DCD_TRY(dcd_balance_owners((uint32_t)G, (uint32_t)G, loff.data(), lfp.data(), lun.data(), totals.data(), ofp.data(), oow.data(), ofp.size(), &n_over,
                               thr.data()));
    DCD_TRY(src.plan_balanced((uint32_t)G, n_over ? ofp.data() : nullptr, n_over ? oow.data() : nullptr, n_over, thr.data(), tc.data(), kc.data(), uc.data()));
  } else {
    DCD_TRY(src.plan((uint32_t)G, tc.data(), kc.data(), uc.data()));
  }
  DevBuf<u8> s_meta, s_keys, s_units;
  const u64 n_meta = sum_of(tc) * META_BYTES, n_keys = sum_of(kc) * 2, n_unit = sum_of(uc) * 2;
  DCD_TRY(s_meta.alloc(c->device, n_meta)); DCD_TRY(s_keys.alloc(c->device, n_keys)); DCD_TRY(s_units.alloc(c->device, n_unit));
  DCD_TRY(src.fill(n_meta ? s_meta.p : nullptr, n_keys ? reinterpret_cast<u16*>(s_keys.p) : nullptr, n_unit ? reinterpret_cast<u16*>(s_units.p) : nullptr));
  // ONE small all-gather tells every rank all the split sizes and the global MaxCoord
  std::vector<u64> mine(3 * (size_t)G + 1), all((3 * (size_t)G + 1) * G);
  for (int p = 0; p < G; p++) { mine[3 * p] = tc[p]; mine[3 * p + 1] = kc[p]; mine[3 * p + 2] = uc[p]; }
  mine[3 * (size_t)G] = my_max;
  DCD_TRY(comm_allgather_host(c, mine.data(), mine.size() * 8, all.data()));
  std::vector<u64> rtc(G), rkc(G), ruc(G);
  u64 mx = 0;
  for (int s = 0; s < G; s++) {
    const u64* m = all.data() + (size_t)s * mine.size();
    rtc[s] = m[3 * c->rank]; rkc[s] = m[3 * c->rank + 1]; ruc[s] = m[3 * c->rank + 2];
    mx = std::max(mx, m[3 * (size_t)G]);
  }
  auto scaled = [](const std::vector<u64>& v, u64 k) { std::vector<u64> r(v); for (u64& x : r) x *= k; return r; };
  const std::vector<u64> sm = scaled(tc, META_BYTES), sk = scaled(kc, 2), su = scaled(uc, 2);
  const std::vector<u64> rm = scaled(rtc, META_BYTES), rk = scaled(rkc, 2), ru = scaled(ruc, 2);
  std::vector<u64> smo, sko, suo, rmo, rko, ruo;
  offsets_of(sm, smo); offsets_of(sk, sko); offsets_of(su, suo); offsets_of(rm, rmo); offsets_of(rk, rko); offsets_of(ru, ruo);
  const bool receives = single_dst < 0 || single_dst == c->rank;
  DevBuf<u8> r_meta, r_keys, r_units;
  DCD_TRY(r_meta.alloc(c->device, sum_of(rm))); DCD_TRY(r_keys.alloc(c->device, sum_of(rk))); DCD_TRY(r_units.alloc(c->device, sum_of(ru)));
  // meta and keys first: the first half of the merge needs nothing else and runs while the units travel
  DCD_TRY(comm_alltoallv(c, s_meta.p, smo.data(), sm.data(), r_meta.p, rmo.data(), rm.data()));
  DCD_TRY(comm_alltoallv(c, s_keys.p, sko.data(), sk.data(), r_keys.p, rko.data(), rk.data()));
  hipEvent_t ev = event_acquire(c->device);
  if (!ev) DCD_FAIL(DCD_E_HIP, "hipEventCreate failed");
  struct EvRel { int d; hipEvent_t e; ~EvRel() { event_release(d, e); } } evrel{c->device, ev};
  DCD_HIP(hipEventRecord(ev, c->st));
  DCD_TRY(comm_alltoallv(c, s_units.p, suo.data(), su.data(), r_units.p, ruo.data(), ru.data()));
  DCD_HIP(hipEventSynchronize(ev));                            // meta + keys have landed (the units are still on the fabric)
  int32_t st = DCD_OK;
  dcd_merge* m = nullptr;
  if (receives) st = dcd_merge_begin(c->device, (uint32_t)G, rtc.data(), rkc.data(), ruc.data(), sum_of(rm) ? r_meta.p : nullptr,
                                     sum_of(rk) ? reinterpret_cast<const u16*>(r_keys.p) : nullptr, mx, &m);
  const hipError_t he = hipStreamSynchronize(c->st);           // the unit exchange has completed (also on the error path: the buffers die here)
  if (st != DCD_OK) return st;
  if (he != hipSuccess) { if (m) dcd_merge_free(m); DCD_FAIL(DCD_E_HIP, "HIP error %d on the communication stream", (int)he); }
  if (!receives) { if (out) *out = nullptr; return DCD_OK; }
  return dcd_merge_finish(m, sum_of(ru) ? reinterpret_cast<const u16*>(r_units.p) : nullptr, out);
}

// every rank receives every rank's one-owner export and merges (every term has exactly one segment)
int32_t allgather_and_merge(dcd_index* owned, dcd_comm* c, dcd_index** out) {
  const int G = c->n;
  DCD_HIP(hipSetDevice(c->device));
  StreamScope ss_(c->st);
  u64 t1 = 0, k1 = 0, u1 = 0, my_max = 0;
  DCD_TRY(dcd_index_export_plan(owned, 1, &t1, &k1, &u1));
  DCD_TRY(dcd_index_max_coord(owned, &my_max));
  DevBuf<u8> s_meta, s_keys, s_units;
  DCD_TRY(s_meta.alloc(c->device, t1 * META_BYTES)); DCD_TRY(s_keys.alloc(c->device, k1 * 2)); DCD_TRY(s_units.alloc(c->device, u1 * 2));
  DCD_TRY(dcd_index_export_fill(owned, t1 ? s_meta.p : nullptr, k1 ? reinterpret_cast<u16*>(s_keys.p) : nullptr, u1 ? reinterpret_cast<u16*>(s_units.p) : nullptr));
  u64 mine[4] = {t1, k1, u1, my_max};
  std::vector<u64> all(4 * (size_t)G);
  DCD_TRY(comm_allgather_host(c, mine, sizeof mine, all.data()));
  std::vector<u64> rtc(G), rkc(G), ruc(G);
  u64 mx = 0;
  for (int s = 0; s < G; s++) { rtc[s] = all[4 * s]; rkc[s] = all[4 * s + 1]; ruc[s] = all[4 * s + 2]; mx = std::max(mx, all[4 * s + 3]); }
  // an all-gather-v = an all-to-all-v in which every rank sends the same bytes to everybody (no padding to the largest part)
  auto bcast = [&](const DevBuf<u8>& sbuf, u64 my_bytes, const std::vector<u64>& rcnt_el, u64 width, DevBuf<u8>& rbuf) -> int32_t {
    std::vector<u64> so(G, 0), sc(G, my_bytes), rc(G), ro;
    for (int s = 0; s < G; s++) rc[s] = rcnt_el[s] * width;
    offsets_of(rc, ro);
    DCD_TRY(rbuf.alloc(c->device, sum_of(rc)));
    return comm_alltoallv(c, sbuf.p, so.data(), sc.data(), rbuf.p, ro.data(), rc.data());
  };
  DevBuf<u8> r_meta, r_keys, r_units;
  DCD_TRY(bcast(s_meta, t1 * META_BYTES, rtc, META_BYTES, r_meta));
  DCD_TRY(bcast(s_keys, k1 * 2, rkc, 2, r_keys));
  DCD_TRY(bcast(s_units, u1 * 2, ruc, 2, r_units));
  DCD_HIP(hipStreamSynchronize(c->st));
  return dcd_merge_shards(c->device, (uint32_t)G, rtc.data(), rkc.data(), ruc.data(), sum_of(rtc) ? r_meta.p : nullptr,
                          sum_of(rkc) ? reinterpret_cast<const u16*>(r_keys.p) : nullptr, sum_of(ruc) ? reinterpret_cast<const u16*>(r_units.p) : nullptr, mx, out);
}

}  // namespace

extern "C" {

int32_t dcd_build_merge_distributed(dcd_build* b, dcd_comm* c, int32_t balanced, dcd_index** owned) {
  if (!b || !c || !owned) DCD_FAIL(DCD_E_BADARG, "dcd_build_merge_distributed: NULL argument");
  if (c->device < 0) DCD_FAIL(DCD_E_BADARG, "the communicator has no device");
  std::lock_guard<std::mutex> g(c->mu);
  ShardSrc s; s.b = b;
  return exchange_and_merge(s, c, balanced != 0, -1, owned);
}
int32_t dcd_index_merge_distributed(dcd_index* shard, dcd_comm* c, int32_t balanced, dcd_index** owned) {
  if (!shard || !c || !owned) DCD_FAIL(DCD_E_BADARG, "dcd_index_merge_distributed: NULL argument");
  if (c->device < 0) DCD_FAIL(DCD_E_BADARG, "the communicator has no device");
  std::lock_guard<std::mutex> g(c->mu);
  ShardSrc s; s.ix = shard;
  return exchange_and_merge(s, c, balanced != 0, -1, owned);
}
int32_t dcd_index_gather(dcd_index* owned, dcd_comm* c, int32_t dst, dcd_index** out) {
  if (!owned || !c || !out || dst < 0 || dst >= c->n) DCD_FAIL(DCD_E_BADARG, "dcd_index_gather: bad argument");
  if (c->device < 0) DCD_FAIL(DCD_E_BADARG, "the communicator has no device");
  std::lock_guard<std::mutex> g(c->mu);
  ShardSrc s; s.ix = owned;
  *out = nullptr;
  return exchange_and_merge(s, c, false, dst, out);
}
int32_t dcd_index_allgather(dcd_index* owned, dcd_comm* c, dcd_index** replica) {
  if (!owned || !c || !replica) DCD_FAIL(DCD_E_BADARG, "dcd_index_allgather: NULL argument");
  if (c->device < 0) DCD_FAIL(DCD_E_BADARG, "the communicator has no device");
  std::lock_guard<std::mutex> g(c->mu);
  return allgather_and_merge(owned, c, replica);
}

}  // extern "C"

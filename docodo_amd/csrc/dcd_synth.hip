// dcd_synth.hip -- synthetic Zipf corpus + query generator of SURVEY.md Appendix F (measurement
// support for bench.py and the parity tests; not part of the reference).  The host and the device
// generator run the same arithmetic on the same host-computed CDF table, so their bytes are equal.
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "dcd_common.h"

namespace dcd {

#define DCD_GOLDEN 0x9E3779B97F4A7C15ull

__host__ __device__ inline u64 splitmix64(u64& state) {
  state += DCD_GOLDEN;
  u64 z = state;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Stream seed of document / query i.  SURVEY Appendix F proposed `seed + i * GOLDEN`, but GOLDEN is
// also splitmix64's own increment, so stream i+1 would be stream i shifted by one draw (consecutive
// documents would be near copies).  Refinement (documented in DESIGN.md): scramble the index with a
// different odd constant and one splitmix64 finaliser round.
__host__ __device__ inline u64 stream_seed(u64 seed, u64 i) {
  u64 z = seed + i * 0xD1B54A32D192ED03ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// term of Zipf rank k (1-based): base-26 little-endian digits of k-1 as 'a'..'z', right-padded with
// 'a' to L(k) = 3 + (((k * 2654435761) mod 2^32) >> 29)
__host__ __device__ inline u32 synth_term(u32 k, char* out) {
  u32 L = 3u + ((u32)(k * 2654435761u) >> 29);
  u32 v = k - 1, n = 0;
  do { out[n++] = (char)('a' + v % 26u); v /= 26u; } while (v);
  while (n < L) out[n++] = 'a';
  return n;
}

// first k (1-based) with cdf[k-1] > u
__host__ __device__ inline u32 zipf_rank(const double* cdf, u32 vocab, double u) {
  u32 lo = 0, hi = vocab - 1;  // cdf[vocab-1] == 1.0 > u always
  while (lo < hi) {
    u32 mid = (lo + hi) >> 1;
    if (cdf[mid] > u) hi = mid; else lo = mid + 1;
  }
  return lo + 1;
}

__host__ __device__ inline void synth_doc(u8* out, u64 doc, u32 doc_bytes, const double* cdf, u32 vocab, u64 seed) {
  u64 state = stream_seed(seed, doc);
  u32 pos = 0;
  char term[16];
  for (;;) {
    double u = (double)(splitmix64(state) >> 11) * (1.0 / 9007199254740992.0);
    u32 k = zipf_rank(cdf, vocab, u);
    u32 n = synth_term(k, term);
    if (pos + n + 1 > doc_bytes) break;
    for (u32 i = 0; i < n; i++) out[pos + i] = (u8)term[i];
    out[pos + n] = ' ';
    pos += n + 1;
  }
  for (; pos < doc_bytes; pos++) out[pos] = ' ';
}

__global__ void k_synth(u8* out, u64 first_doc, u64 n_docs, u32 doc_bytes, const double* cdf, u32 vocab, u64 seed) {
  u64 d = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (d < n_docs) synth_doc(out + d * doc_bytes, first_doc + d, doc_bytes, cdf, vocab, seed);
}

static std::shared_ptr<std::vector<double>> zipf_cdf(u32 vocab, double s) {
  static std::mutex mu;
  static std::map<std::pair<u32, double>, std::shared_ptr<std::vector<double>>> cache;
  std::lock_guard<std::mutex> g(mu);
  auto key = std::make_pair(vocab, s);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  auto v = std::make_shared<std::vector<double>>(vocab);
  double acc = 0;
  for (u32 k = 1; k <= vocab; k++) { acc += std::pow((double)k, -s); (*v)[k - 1] = acc; }
  double H = acc;
  for (u32 k = 0; k < vocab; k++) (*v)[k] /= H;
  (*v)[vocab - 1] = 1.0;
  cache[key] = v;
  return v;
}

}  // namespace dcd

using namespace dcd;

extern "C" {

int32_t dcd_synth_corpus_host(uint8_t* out, uint64_t first_doc, uint64_t n_docs, uint32_t doc_bytes, uint32_t vocab,
                              double zipf_s, uint64_t seed) {
  if (!out || vocab == 0 || doc_bytes == 0) DCD_FAIL(DCD_E_BADARG, "dcd_synth_corpus_host: bad argument");
  auto cdf = zipf_cdf(vocab, zipf_s);
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 16) nt = 16;
  if (n_docs < 4096) nt = 1;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) {
    u64 a = n_docs * t / nt, b = n_docs * (t + 1) / nt;
    th.emplace_back([=]() {
      for (u64 d = a; d < b; d++) synth_doc(out + d * doc_bytes, first_doc + d, doc_bytes, cdf->data(), vocab, seed);
    });
  }
  for (auto& t : th) t.join();
  return DCD_OK;
}

int32_t dcd_synth_corpus_dev(void* out_dev, uint64_t first_doc, uint64_t n_docs, uint32_t doc_bytes, uint32_t vocab,
                             double zipf_s, uint64_t seed, int32_t device) {
  if (!out_dev || vocab == 0 || doc_bytes == 0) DCD_FAIL(DCD_E_BADARG, "dcd_synth_corpus_dev: bad argument");
  DCD_TRY(ensure_device(device));
  StreamScope ss_(nullptr);   // the generator runs on the null stream
  auto cdf = zipf_cdf(vocab, zipf_s);
  DevBuf<double> dc;
  DCD_TRY(dc.alloc(device, vocab));
  DCD_HIP(hipMemcpy(dc.p, cdf->data(), (size_t)vocab * 8, hipMemcpyHostToDevice));
  if (n_docs) {
    hipLaunchKernelGGL(k_synth, dim3((u32)div_up(n_docs, 64)), dim3(64), 0, 0, (u8*)out_dev, first_doc, n_docs, doc_bytes,
                       dc.p, vocab, seed);
    DCD_HIP(hipGetLastError());
  }
  DCD_HIP(hipDeviceSynchronize());
  return DCD_OK;
}

int32_t dcd_synth_term(uint32_t rank, uint32_t vocab, char* out, size_t cap, size_t* len) {
  if (!out || !len || rank == 0 || rank > vocab) DCD_FAIL(DCD_E_BADARG, "dcd_synth_term: bad argument");
  char t[16];
  u32 n = synth_term(rank, t);
  *len = n;
  if (n > cap) return DCD_E_OVERFLOW;
  memcpy(out, t, n);
  return DCD_OK;
}

int32_t dcd_synth_query_ranks(uint32_t* out, uint64_t n_queries, uint32_t terms_per_query, int32_t distinct,
                              uint32_t lo_rank, uint32_t vocab, double zipf_s, uint64_t seed) {
  if (!out || lo_rank == 0 || lo_rank > vocab || terms_per_query == 0 || (distinct && terms_per_query > vocab - lo_rank + 1))
    DCD_FAIL(DCD_E_BADARG, "dcd_synth_query_ranks: bad argument");
  auto cdf = zipf_cdf(vocab, zipf_s);
  double c0 = lo_rank > 1 ? (*cdf)[lo_rank - 2] : 0.0;
  for (u64 q = 0; q < n_queries; q++) {
    u64 state = stream_seed(seed, q);
    for (u32 t = 0; t < terms_per_query; t++) {
      for (;;) {
        double u = (double)(splitmix64(state) >> 11) * (1.0 / 9007199254740992.0);
        double uu = c0 + u * (1.0 - c0);
        u32 k = zipf_rank(cdf->data(), vocab, uu);
        if (k < lo_rank) k = lo_rank;
        bool dup = false;
        if (distinct) for (u32 j = 0; j < t; j++) if (out[q * terms_per_query + j] == k) dup = true;
        if (!dup) { out[q * terms_per_query + t] = k; break; }
      }
    }
  }
  return DCD_OK;
}

}  // extern "C"

// dcd_query.hip -- query evaluation (placeholder entry points; kernels follow).
#include "dcd_index_int.h"

using namespace dcd;

extern "C" {

void dcd_query_default_opts(dcd_query_opts* o) {
  if (!o) return;
  o->max_found_pages = 30000;  // Index.cs:101
  o->max_found_docs = 500;     // Index.cs:102
  o->doc_rank_multiply = 10.f; // Index.cs:115
  o->want_sequences = 0;
}

int32_t dcd_query_batch(dcd_index*, const int32_t*, const uint64_t*, size_t, const dcd_query_opts*, dcd_results**) {
  DCD_FAIL(DCD_E_UNSUPPORTED, "dcd_query_batch: not built yet");
}
int32_t dcd_results_get(dcd_results*, dcd_results_view*) { DCD_FAIL(DCD_E_UNSUPPORTED, "dcd_results_get: not built yet"); }
void dcd_results_free(dcd_results*) {}
int32_t dcd_seq_and(const uint16_t*, size_t, int32_t, const uint16_t*, size_t, int32_t, uint16_t*, size_t, size_t*, int32_t*) {
  DCD_FAIL(DCD_E_UNSUPPORTED, "dcd_seq_and: not built yet");
}
int32_t dcd_seq_or(const uint16_t*, size_t, int32_t, const uint16_t*, size_t, int32_t, uint16_t*, size_t, size_t*, int32_t*) {
  DCD_FAIL(DCD_E_UNSUPPORTED, "dcd_seq_or: not built yet");
}
}

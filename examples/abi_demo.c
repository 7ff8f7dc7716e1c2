/* abi_demo.c -- the drop-in boundary used from plain C (no Python, no torch): what a C# / C++ host does through
 * include/docodo_hip.h.  Builds a two-document index from UTF-16 pages (one header page each), saves ".index" +
 * ".index.list", reopens them (Index.Load) and evaluates a batch of two queries: `pickwick` and the exact phrase
 * "mr tupman" (R = -(len + 4), Search.cs:308).
 *   gcc -std=c11 -I include examples/abi_demo.c -L docodo_amd -ldocodo_hip -Wl,-rpath,$PWD/docodo_amd -o abi_demo && ./abi_demo /tmp/ix
 * Exit code 0 = results as expected; prints the library error and returns 1 otherwise (e.g. no GPU: DCD_E_HIP). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "docodo_hip.h"

static size_t u16(const char* s, uint16_t* out) { size_t n = 0; while (s[n]) { out[n] = (uint16_t)(unsigned char)s[n]; n++; } return n; }
static int fail(const char* what, int32_t st) {
  char msg[512] = "";
  dcd_last_error(msg, sizeof msg);
  fprintf(stderr, "%s failed: status %d: %s\n", what, (int)st, msg);
  return 1;
}
#define CK(call) do { int32_t st_ = (call); if (st_ != DCD_OK) return fail(#call, st_); } while (0)

int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : ".";
  const char* docs[2][2] = {
      {"Name=First letter\nAuthor=Dickens\n", "Mr. Pickwick and Mr. Tupman walked on; Mr. Tupman smiled at Mr. Pickwick."},
      {"Name=Second letter\n", "Nobody but mr tupman ever doubted that pickwick was right."}};
  dcd_build_opts bo;
  dcd_build_default_opts(&bo);
  bo.encoding = DCD_ENC_UTF16LE;
  dcd_build* b = NULL;
  CK(dcd_build_create(&bo, &b));
  uint16_t src[8], name[16], text[512], ids[8];
  const size_t nsrc = u16("demo", src);
  for (int d = 0; d < 2; d++) {
    char nm[16];
    snprintf(nm, sizeof nm, "doc%d", d + 1);
    CK(dcd_build_add_doc(b, src, nsrc, name, u16(nm, name)));
    uint64_t page_off[3] = {0, 0, 0}, ids_off[3] = {0, 1, 2};
    size_t n = u16(docs[d][0], text);
    page_off[1] = n;
    n += u16(docs[d][1], text + n);
    page_off[2] = n;
    u16("01", ids);                                  /* page ids "0" (header page) and "1" */
    CK(dcd_build_add_pages(b, text, n, page_off, 2, ids, ids_off, 0, 0, NULL, 0));
  }
  CK(dcd_build_finish(b));
  dcd_build_stats st;
  CK(dcd_build_get_stats(b, &st));
  printf("built: %llu postings, %llu codes, %llu pages, .index = %llu bytes\n", (unsigned long long)st.n_postings,
         (unsigned long long)st.n_terms, (unsigned long long)st.n_pages, (unsigned long long)st.index_bytes);
  CK(dcd_build_save(b, dir));
  dcd_build_free(b);

  dcd_index* ix = NULL;
  CK(dcd_index_open(dir, 0, &ix));
  uint16_t w[32];
  uint32_t t_pick = 0, t_mr = 0, t_tup = 0;
  CK(dcd_index_find(ix, w, u16("pickwick", w), &t_pick));
  CK(dcd_index_find(ix, w, u16("mr", w), &t_mr) == DCD_E_NOTFOUND ? DCD_OK : DCD_E_STATE);   /* "mr" is shorter than MIN_WORD_LENGTH */
  CK(dcd_index_find(ix, w, u16("tupman", w), &t_tup));
  (void)t_mr;
  /* query 0: pickwick (Dist = 255 -> R = 255 + 8); query 1: "tupman smiled"-style exact pair is not available ("mr" is
   * not indexed), so: pickwick * tupman with Dist = 255, both documents match */
  int32_t prog[] = {DCD_OP_TERM, (int32_t)t_pick, 255 + 8,
                    DCD_OP_TERM, (int32_t)t_pick, 255 + 8, DCD_OP_TERM, (int32_t)t_tup, 255 + 6, DCD_OP_AND};
  uint64_t prog_off[] = {0, 3, 10};
  dcd_results* r = NULL;
  CK(dcd_query_batch(ix, prog, prog_off, 2, NULL, &r));
  dcd_results_view v;
  CK(dcd_results_get(r, &v));
  int ok = v.n_queries == 2 && v.status[0] == DCD_OK && v.status[1] == DCD_OK;
  for (uint64_t q = 0; q < v.n_queries; q++) {
    printf("query %llu: %llu pages, %llu documents\n", (unsigned long long)q, (unsigned long long)(v.page_off[q + 1] - v.page_off[q]),
           (unsigned long long)(v.doc_off[q + 1] - v.doc_off[q]));
    for (uint64_t p = v.page_off[q]; p < v.page_off[q + 1]; p++)
      printf("  page %u rank %.4f positions %llu\n", v.page_index[p], v.page_rank[p], (unsigned long long)(v.pos_off[p + 1] - v.pos_off[p]));
  }
  ok = ok && (v.doc_off[1] - v.doc_off[0]) == 2 && (v.doc_off[2] - v.doc_off[1]) == 2;
  dcd_results_free(r);
  dcd_index_free(ix);
  printf(ok ? "ok\n" : "UNEXPECTED RESULT\n");
  return ok ? 0 : 1;
}

// Host-side unit test of the hot / cold sort key (dcd_sort.h rs_hot_key, dcd_codec.h EncSrcPairs::hot_key_id): the key
// packs {pass-A digit, cold id} into 32 bits; round trip and order at the widths the build may choose, and the guard
// (rs_hot_key_fits) that makes the build give up the split where the packing would wrap (255 hot terms and a
// cold id range beyond 2^24).  Built with hipcc (host code only runs) by tests/test_abi.py.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "dcd_common.h"
#include "dcd_sort.h"
#include "dcd_codec.h"
using namespace dcd;

int main() {
  int bad = 0;
  for (u32 hot_n : {1u, 5u, 63u, 64u, 127u, 255u}) {
    for (u32 hot_s = 1; hot_s <= 31; hot_s++) {
      const bool fits = rs_hot_key_fits(hot_n, hot_s);
      const u64 range = (u64)hot_n + (1ull << hot_s);               // ids 0 .. range - 1
      // exhaustive at the edges, strided in between
      std::vector<u64> probes = {0, 1, (u64)hot_n - 1, hot_n, (u64)hot_n + 1, range / 2, range - 2, range - 1};
      std::sort(probes.begin(), probes.end());
      probes.erase(std::unique(probes.begin(), probes.end()), probes.end());
      bool ok = true;
      u32 prev_key = 0; bool have_prev = false;
      for (u64 d64 : probes) {
        if (d64 >= range) continue;
        const u32 d = (u32)d64;
        const u64 wide = d < hot_n ? (u64)d << hot_s : ((u64)hot_n << hot_s) + (d - hot_n);
        const u32 key = rs_hot_key(d, hot_n, hot_s);
        if (fits) {
          if ((u64)key != wide) ok = false;                         // no wrap
          if (EncSrcPairs::hot_key_id(key, hot_n, hot_s) != d) ok = false;
          if ((key >> hot_s) != (d < hot_n ? d : hot_n)) ok = false;   // pass-A digit: own bucket per hot term, one for the cold ones
          if (have_prev && key <= prev_key) ok = false;             // strictly order-preserving (probes ascend)
          prev_key = key; have_prev = true;
        }
      }
      // where the guard says no, the widest cold id really does not fit
      if (!fits) {
        const u64 top = ((u64)hot_n << hot_s) + ((1ull << hot_s) - 1);
        if (top <= 0xFFFFFFFFull) ok = false;
      }
      if (!ok) { printf("FAIL hot_n=%u hot_s=%u fits=%d\n", hot_n, hot_s, (int)fits); bad++; }
    }
  }
  // the case of the finding: 255 hot terms, cold range 2^25 .. 2^29
  for (u32 hot_s = 25; hot_s <= 29; hot_s++)
    if (rs_hot_key_fits(255, hot_s)) { printf("FAIL guard lets 255 << %u through\n", hot_s); bad++; }
  if (!rs_hot_key_fits(255, 24) || !rs_hot_key_fits(127, 25)) { printf("FAIL guard too strict\n"); bad++; }
  printf(bad ? "hot key: %d failures\n" : "hot key ok\n", bad);
  return bad ? 1 : 0;
}

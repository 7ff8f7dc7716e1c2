"""Kernel timeline of the LAST build of a `rocprofv3 --kernel-trace --output-format csv` run of scripts/quick_build.py:
start offset, duration and the idle gap before every dispatch (host round trips show up as gaps).
usage: python scripts/timeline.py DIR/…_kernel_trace.csv"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last build starts at the last k_uniform_page_end (first kernel of build_phase1)
starts = [i for i, r in enumerate(rows) if "k_uniform_page_end" in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void dcd::", "").replace("dcd::", "")[:60]
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.1f us, busy %.1f us, idle %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3, (prev_end - t0 - busy) / 1e3))

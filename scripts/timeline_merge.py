"""Kernel timeline of the LAST export + merge of a `rocprofv3 --kernel-trace --output-format csv` run of scripts/quick_merge.py.
usage: python scripts/timeline_merge.py DIR/…_kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last build ends with its last k_img_units; what follows is export + cat + merge
last = max(i for i, r in enumerate(rows) if "k_img_units" in r["Kernel_Name"])
rows = rows[last:]
t0 = int(rows[0]["End_Timestamp"]); prev = t0; busy = 0
for r in rows[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void dcd::", "").replace("dcd::", "")[:70]
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name))
    busy += e - s; prev = max(prev, e)
print("span %.1f us, busy %.1f us" % ((prev - t0) / 1e3, busy / 1e3))

"""Lists the last N dispatches of a rocprofv3 kernel trace (name, duration in microseconds).  usage: python scripts/trace_tail.py DIR [N] [min_us]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
for r in rows[-n:]:
    if (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 < min_us:
        continue
    print("%10.1f us  +%8.1f  %s  grid %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t_end) / 1e3,
                                         r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))

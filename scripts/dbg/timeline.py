#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace CSV: per build step (or whole file) the kernels in start order with queue, start offset,
duration and the idle gap in front of each (no kernel of any queue running).  usage: timeline.py trace.csv [first_kernel_substr] [step]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else None
want = int(sys.argv[3]) if len(sys.argv) > 3 else -1
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], re.sub(r"\(.*", "", r["Kernel_Name"]).replace("dcd::", "").replace("void ", "")[:60],
       r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"]) for r in rows]
ev.sort()
steps = []
if mark:
    cur = None
    for e in ev:
        if mark in e[3]:
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append(e)
else:
    steps = [ev]
sel = steps[want]
t0 = sel[0][0]
busy_end = t0
idle = 0
print("%-62s %3s %9s %9s %8s %5s %6s %8s" % ("kernel", "q", "start_us", "dur_us", "gap_us", "vgpr", "lds", "grid"))
for s, e, q, n, v, l, g in sel:
    gap = max(0, s - busy_end)
    idle += gap
    print("%-62s %3s %9.1f %9.1f %8.1f %5s %6s %8s" % (n, q, (s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, v, l, g))
    busy_end = max(busy_end, e)
print("span %.1f us, idle (no kernel running) %.1f us, %d kernels" % ((busy_end - t0) / 1e3, idle / 1e3, len(sel)))

"""Prints the first rows of the kernel_stats.csv rocprofv3 left under a directory.  usage: python scripts/top_kernels.py DIR [rows]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for i, r in enumerate(csv.reader(open(f))):
    if i <= n:
        print(",".join(x[:60] for x in r[:7]))

#!/bin/bash
out=$1; lib=$2
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
export DOCODO_HIP_LIB=$lib
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/$c -o p -- python3 $R/scripts/diag_tok.py 1000000 2 > $R/$out/$c.log 2>&1
done
cd $R
python3 - <<P
import csv, collections, re
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    mx = collections.defaultdict(float)
    for r in csv.DictReader(open("$out/%s/p_counter_collection.csv" % c)):
        if r["Counter_Name"] != c: continue
        m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", r["Kernel_Name"])
        if m: mx[m.group(1)] = max(mx[m.group(1)], float(r["Counter_Value"]))
    for k in sorted(mx):
        if mx[k] * 1024 > 5e7: print("%-12s %-34s %8.1f MB" % (c, k, mx[k] * 1024 / 1e6))
P
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE

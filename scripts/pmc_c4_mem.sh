#!/bin/bash
# usage (GPU box): scripts/pmc_c4_mem.sh OUTDIR  -- memory-side counters of k_pair_parts on one configs[3] sub-batch (c4_profile.py, 20 k queries):
# one rocprofv3 --pmc pass per counter, the largest dispatch (round 0 of the bigger sub-batch) reported
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp
R=$PWD
cd /tmp
CTRS=${CTRS:-"FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum"}
for c in $CTRS; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$out/$c -o p -- python3 $R/scripts/c4_profile.py 10000000 20000 1 > $R/$out/$c.log 2>&1
done
cd $R
python3 - <<P
import csv, collections, os
for c in "$CTRS".split():
    f = "$out/%s/p_counter_collection.csv" % c
    if not os.path.exists(f):
        print(c, "no output"); continue
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "k_pair_parts" in r["Kernel_Name"]]
    print("%-24s k_pair_parts: %d dispatches, max %.4g, sum of the last 4 (one batch of 20 k queries) %.4g" % (c, len(vals), max(vals) if vals else 0, sum(vals[-4:])))
P

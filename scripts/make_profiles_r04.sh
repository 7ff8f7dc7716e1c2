#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/make_profiles_r04.sh [TAG]
# -> gpurun_out/profiles/TAG_bench.json          the default bench.py run (every leg)
#    TAG_kernel_stats.csv                        rocprofv3 --kernel-trace --stats of the build steps (same command line, CPU legs off)
#    TAG_traffic.json                            FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes) -> bench.py roofline.traffic
#    TAG_c4_kernel_stats.csv, TAG_c5_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the configs[2..4] legs (100 k AND3; 100 k phrases)
#    TAG_pmc_c4.txt, TAG_pmc_c5.txt              SQ counters of k_pair_stream / k_pair_parts (instructions, issue / wait) on 20 k AND3 / 20 k phrases
# (copy what is to be judged into profiles/ afterwards)
set -e
TAG=${1:-r04_final}
R=$PWD
O=$R/gpurun_out/profiles
RAW=/tmp/dcd_prof_raw          # raw rocprofv3 output stays on the box: gpurun_out/ may carry 64 MiB back
rm -rf $RAW
mkdir -p $O $RAW
export TMPDIR=/tmp
echo "[profiles] bench"; timeout -k 10 600 python3 bench.py > $O/${TAG}_bench.json 2> $RAW/bench.err
cd /tmp
BUILD_ARGS="--steps 5 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 --total-docs 0"
echo "[profiles] build kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o stats -- python3 $R/bench.py $BUILD_ARGS > $RAW/stats.log 2>&1
cp $RAW/stats_kernel_stats.csv $O/${TAG}_kernel_stats.csv
echo "[profiles] traffic"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-docs 0 --host-legs 0 --queries 0 --inflight 0 --total-docs 0 > $RAW/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW -o write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-docs 0 --host-legs 0 --queries 0 --inflight 0 --total-docs 0 > $RAW/write.log 2>&1
cd $R
python3 scripts/traffic_json.py $RAW/fetch_counter_collection.csv $RAW/write_counter_collection.csv $O/${TAG}_bench.json $O/${TAG}_traffic.json
cd /tmp
echo "[profiles] query kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o q4 -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 --c4-queries 100000 --c5-queries 0 > $RAW/q4.log 2>&1
cp $RAW/q4_kernel_stats.csv $O/${TAG}_c4_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o q5 -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 --c4-queries 0 --c5-queries 100000 > $RAW/q5.log 2>&1
cp $RAW/q5_kernel_stats.csv $O/${TAG}_c5_kernel_stats.csv
cd $R
for leg in c4 c5; do
  echo "[profiles] $leg PMC"
  if [ $leg = c4 ]; then QA="--c4-queries 20000 --c5-queries 0"; else QA="--c4-queries 0 --c5-queries 20000"; fi
  i=0
  for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $RAW/pmc$leg/g$i -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 $QA > $RAW/pmc${leg}_g$i.log 2>&1
  done
  python3 scripts/pmc_summary.py $RAW/pmc$leg k_pair_stream k_pair_parts k_pair_gather k_result k_adopt_apply > $O/${TAG}_pmc_$leg.txt
done
echo "[profiles] c4 traffic (FETCH_SIZE of the operator kernels on one 20 k-query call)"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW -o c4fetch -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 --c4-queries 20000 --c5-queries 0 > $RAW/c4fetch.log 2>&1
cd $R
python3 - <<P
import csv, json, collections
rows = list(csv.DictReader(open("$RAW/c4fetch_counter_collection.csv")))
per = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] == "FETCH_SIZE":
        for k in ("k_pair_stream", "k_pair_parts"):
            if k in r["Kernel_Name"]:
                per[k].append(float(r["Counter_Value"]))
# two calls of 20 k queries ran (warm-up + timed): the second half of the dispatches is the timed call
out = {"total_docs": 10000000, "queries": 20000, "kernels": {}}
tot = 0.0
for k, v in per.items():
    half = v[len(v) // 2:]
    kb = sum(half)
    out["kernels"][k] = {"dispatches": len(half), "fetch_kb": kb}
    tot += kb
out["fetch_bytes"] = tot * 1024.0          # FETCH_SIZE is in KB; random 64-byte requests: taken as counted (MI355X_MICROARCH.md: the x2 applies to wide streamed reads)
json.dump(out, open("$O/${TAG}_c4_traffic.json", "w"), indent=1)
print(out)
P
cp $RAW/stats_kernel_trace.csv $O/${TAG}_build_trace.csv 2>/dev/null || true
cp $RAW/q4_kernel_trace.csv $O/${TAG}_c4_trace.csv 2>/dev/null || true
cp $RAW/bench.err $O/${TAG}_bench.err 2>/dev/null || true
for f in stats q4 q5 fetch; do tail -3 $RAW/$f.log; done
du -sh $O
ls -la $O

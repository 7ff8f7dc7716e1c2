#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/make_profiles.sh TAG
# -> gpurun_out/profiles/TAG_bench.json, TAG_kernel_stats.csv, TAG_traffic.json, TAG_c4_scale.json  (copy into profiles/ afterwards)
set -e
TAG=${1:-r02_final}
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O/raw
export TMPDIR=/tmp
timeout -k 10 600 python3 bench.py > $O/${TAG}_bench.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o stats -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-docs 0 --inflight 0 > $O/raw/stats.log 2>&1
cp $O/raw/stats_kernel_stats.csv $O/${TAG}_kernel_stats.csv
timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-docs 0 --queries 0 --inflight 0 > $O/raw/fetch.log 2>&1
timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw -o write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-docs 0 --queries 0 --inflight 0 > $O/raw/write.log 2>&1
cd $R
python3 scripts/traffic_json.py $O/raw/fetch_counter_collection.csv $O/raw/write_counter_collection.csv $O/${TAG}_bench.json $O/${TAG}_traffic.json
timeout -k 10 900 python3 scripts/c4_scale.py 10000000 100000 2000 20000 | tail -1 > $O/${TAG}_c4_scale.json
rm -rf $O/raw/*trace.csv
ls -la $O

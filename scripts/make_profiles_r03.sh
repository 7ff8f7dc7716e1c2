#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/make_profiles_r03.sh [TAG]
# -> gpurun_out/profiles/TAG_bench.json          the default bench.py run (every leg)
#    TAG_kernel_stats.csv                        rocprofv3 --kernel-trace --stats of the build steps (same command line, CPU legs off)
#    TAG_traffic.json                            FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes) -> bench.py roofline.traffic
#    TAG_c4_kernel_stats.csv                     rocprofv3 --kernel-trace --stats of the configs[2..4] legs (100 k AND3 + 4 k phrases)
#    TAG_pmc_c4.txt                              SQ counters of k_pair_parts on configs[3] (issue / wait / LDS)
#    TAG_c4_mem.txt                              L2 / HBM-side counters of k_pair_parts on one 20 k configs[3] batch (scripts/pmc_c4_mem.sh)
#    TAG_c4_phases.txt, TAG_c5_phases.txt        in-kernel phase profile of k_pair_parts (needs variants/libdocodo_hip_prof.so:
#                                                scripts/build_variant.sh prof "-DDCD_PROF")
# (copy what is to be judged into profiles/ afterwards)
set -e
TAG=${1:-r03_final}
R=$PWD
O=$R/gpurun_out/profiles
mkdir -p $O/raw
export TMPDIR=/tmp
echo "[profiles] bench"; timeout -k 10 500 python3 bench.py > $O/${TAG}_bench.json 2> $O/raw/bench.err
cd /tmp
BUILD_ARGS="--steps 5 --warmup 1 --cpu-docs 0 --inflight 0 --queries 0 --total-docs 0"
echo "[profiles] build kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o stats -- python3 $R/bench.py $BUILD_ARGS > $O/raw/stats.log 2>&1
cp $O/raw/stats_kernel_stats.csv $O/${TAG}_kernel_stats.csv
echo "[profiles] traffic"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw -o fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-docs 0 --queries 0 --inflight 0 --total-docs 0 > $O/raw/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw -o write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-docs 0 --queries 0 --inflight 0 --total-docs 0 > $O/raw/write.log 2>&1
cd $R
python3 scripts/traffic_json.py $O/raw/fetch_counter_collection.csv $O/raw/write_counter_collection.csv $O/${TAG}_bench.json $O/${TAG}_traffic.json
cd /tmp
Q_ARGS="--steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --queries 0 --c4-queries 100000 --c5-queries 4000"
echo "[profiles] query kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -o qstats -- python3 $R/bench.py $Q_ARGS > $O/raw/qstats.log 2>&1
cp $O/raw/qstats_kernel_stats.csv $O/${TAG}_c4_kernel_stats.csv
cd $R
echo "[profiles] k_pair_parts PMC"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $O/raw/pmcq/g$i -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --queries 0 --c4-queries 20000 --c5-queries 0 > $O/raw/pmcq_g$i.log 2>&1
done
python3 scripts/pmc_summary.py $O/raw/pmcq k_pair_parts k_pair_gather k_result > $O/${TAG}_pmc_c4.txt
echo "[profiles] k_pair_parts memory counters"
CTRS="FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum" bash scripts/pmc_c4_mem.sh gpurun_out/profiles/raw/c4mem > $O/${TAG}_c4_mem.txt 2>&1 || true
if [ -f variants/libdocodo_hip_prof.so ]; then
  echo "[profiles] k_pair_parts phase profiles"
  DOCODO_HIP_LIB=variants/libdocodo_hip_prof.so timeout -k 10 300 python3 scripts/c4_phase_profile.py 10000000 20000 2>&1 | grep -v amdgpu.ids > $O/${TAG}_c4_phases.txt || true
  DOCODO_HIP_LIB=variants/libdocodo_hip_prof.so timeout -k 10 300 python3 scripts/c4_phase_profile.py 10000000 2500 phrase 2>&1 | grep -v amdgpu.ids > $O/${TAG}_c5_phases.txt || true
fi
rm -rf $O/raw/*trace.csv $O/raw/c4mem
ls -la $O

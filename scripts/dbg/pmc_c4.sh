#!/bin/bash
# usage (GPU box, repo root): scripts/dbg/pmc_c4.sh TAG [ENV=VALUE ...]  -> gpurun_out/mf/pmc_TAG.txt (k_pair_stream / k_pair_parts, 20 k AND3)
TAG=$1; shift
for e in "$@"; do export "$e"; done
RAW=/tmp/pmc_$TAG; rm -rf $RAW; mkdir -p $RAW gpurun_out/mf
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d $RAW/g$i -o p -- python3 bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 --c4-queries 20000 --c5-queries 0 > $RAW/g$i.log 2>&1 || { tail -5 $RAW/g$i.log; exit 1; }
done
python3 scripts/pmc_summary.py $RAW k_pair_stream k_pair_parts > gpurun_out/mf/pmc_$TAG.txt
grep -A20 "== k_pair_stream" gpurun_out/mf/pmc_$TAG.txt | grep -E "==|INSTS_VALU|INSTS_SALU|WAVE_CYCLES|BUSY|WAIT_INST_ANY|WAIT_ANY|ACTIVE_INST_VALU"

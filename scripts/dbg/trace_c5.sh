#!/bin/bash
# usage: trace_c5.sh TAG [ENV=VAL ...] -> gpurun_out/mf/trace_c5_TAG.csv (kernel trace of the operator kernels of a configs[4] run)
TAG=$1; shift
for e in "$@"; do export "$e"; done
RAW=/tmp/tr5_$TAG; rm -rf $RAW; mkdir -p $RAW gpurun_out/mf; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $RAW -o t -- python3 bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 --c4-queries 0 --c5-queries ${C5Q:-100000} > $RAW/log 2>&1 || { tail -5 $RAW/log; exit 1; }
grep -E "k_pair_stream|k_pair_parts|k_pair_gather|k_result|Kernel_Name" $RAW/t_kernel_trace.csv > gpurun_out/mf/trace_c5_$TAG.csv
grep "c5_phrase:" $RAW/log

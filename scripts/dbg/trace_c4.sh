#!/bin/bash
# usage: trace_c4.sh TAG [ENV=VAL ...] -> gpurun_out/mf/trace_TAG.csv (kernel trace of a 20 k-query configs[3] run)
TAG=$1; shift
for e in "$@"; do export "$e"; done
RAW=/tmp/tr_$TAG; rm -rf $RAW; mkdir -p $RAW gpurun_out/mf; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $RAW -o t -- python3 bench.py --steps 1 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 --c4-queries ${C4Q:-20000} --c5-queries 0 > $RAW/log 2>&1 || { tail -5 $RAW/log; exit 1; }
grep -v "k_synth\|k_tok_emit\|k_rs_\|k_pack\|k_img\|k_merge\|k_export\|k_adopt\|k_vocab\|k_term\|k_char\|k_hot\|k_dir_\|k_hist" $RAW/t_kernel_trace.csv > gpurun_out/mf/trace_$TAG.csv
grep "c4_and3:" $RAW/log

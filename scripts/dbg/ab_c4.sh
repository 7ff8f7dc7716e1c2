#!/bin/bash
# usage: ab_c4.sh "ENV=VAL ENV=VAL" ...   one bench (c4 + c5 legs) per argument
mkdir -p gpurun_out/mf
i=0
for cfg in "$@"; do
  i=$((i+1))
  ( for e in $cfg; do export "$e"; done; timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 > gpurun_out/mf/ab_$i.json 2> gpurun_out/mf/ab_$i.err; echo "[$cfg] rc=$? $(grep -E 'c4_and3:|c5_phrase:' gpurun_out/mf/ab_$i.err | sed 's/.*queries, //' | tr '\n' ' ')" )
done

#!/bin/bash
# usage: ab_c4.sh "ENV=VAL ENV=VAL" ...   one bench (c4 + c5 legs) per argument: device / wall ms of both legs
mkdir -p gpurun_out/mf
i=0
for cfg in "$@"; do
  i=$((i+1))
  ( for e in $cfg; do export "$e"; done; timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-docs 0 --inflight 0 --host-legs 0 --queries 0 > gpurun_out/mf/ab_$i.json 2> gpurun_out/mf/ab_$i.err; echo "[$cfg] rc=$? $(python3 -c "
import json; c=json.load(open('gpurun_out/mf/ab_$i.json'))['configs234']
print('c4 %.1f / %.1f ms  c5 %.0f / %.0f ms' % (c['c4_and3']['device_ms'], c['c4_and3']['call_wall_ms'], c['c5_phrase']['device_ms'], c['c5_phrase']['call_wall_ms']))")" )
done

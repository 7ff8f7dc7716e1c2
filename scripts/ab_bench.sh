#!/bin/bash
# usage (GPU box): scripts/ab_bench.sh NAME...   per-kernel ms of the C2 build step for library variants (base = in-tree)
for v in "$@"; do
  unset DOCODO_HIP_LIB
  [ "$v" != base ] && export DOCODO_HIP_LIB=$PWD/variants/libdocodo_hip_$v.so
  timeout -k 10 300 python bench.py --cpu-docs 0 --queries 0 --steps 8 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('%-8s total %.2f |' % ('$v', d['device_ms_per_step']['total']), ' '.join('%s %.3f' % (a.replace('k_',''), b) for a, b in k.items()))
" || exit 1
done

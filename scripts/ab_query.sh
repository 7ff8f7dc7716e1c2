#!/bin/bash
# usage (on the GPU box): scripts/ab_query.sh NAME...   runs the 1M-doc AND3 bench and the C4-scale bench for variant libs
for v in "$@"; do
  if [ "$v" = base ]; then unset DOCODO_HIP_LIB; else export DOCODO_HIP_LIB=$PWD/variants/libdocodo_hip_$v.so; fi
  echo "== $v"
  timeout -k 10 300 python scripts/quick_query.py 1000000 100 20000 2>&1 | tail -1 || exit 1
  timeout -k 10 600 python scripts/c4_scale.py 10000000 20000 0 20000 2>&1 | grep -E "^and3" || exit 1
done

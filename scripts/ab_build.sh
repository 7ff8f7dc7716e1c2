#!/bin/bash
# usage (GPU box): scripts/ab_build.sh NAME...   (base = in-tree library; others = variants/libdocodo_hip_NAME.so)
for v in "$@"; do
  unset DOCODO_HIP_LIB
  [ "$v" != base ] && export DOCODO_HIP_LIB=$PWD/variants/libdocodo_hip_$v.so
  echo "== $v"
  timeout -k 10 300 python scripts/quick_build.py 1000000 5 2>&1 | grep "^rep" | tail -3 | sed 's/| N=.*->/->/' || exit 1
done

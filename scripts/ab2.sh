#!/bin/bash
# usage (GPU box): scripts/ab2.sh NAME[+pPART]...   e.g. base gp1024 gp1024+p4096
for v in "$@"; do
  unset DOCODO_HIP_LIB DOCODO_QUERY_PART
  lib=${v%%+*}; part=${v#*+p}
  [ "$lib" != base ] && export DOCODO_HIP_LIB=$PWD/variants/libdocodo_hip_$lib.so
  [ "$part" != "$v" ] && export DOCODO_QUERY_PART=$part
  echo "== $v"
  timeout -k 10 300 python scripts/quick_query.py 1000000 100 20000 2>&1 | tail -1 | sed 's/.*dev \([0-9.]*\) ms.*/1M: \1 ms/' || exit 1
  timeout -k 10 600 python scripts/c4_scale.py 10000000 20000 0 20000 2>&1 | grep -E "^and3" | sed "s/.*'device_ms': \([0-9.]*\).*/C4: \1 ms/" || exit 1
done

#!/bin/bash
# usage: tl.sh TAG [ENV=VAL ...]
tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl/$tag -o kt -- python3 $GRAFT_REPO_ROOT/scripts/quick_build.py 1000000 4 > $GRAFT_REPO_ROOT/gpurun_out/tl/$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python3 scripts/timeline.py gpurun_out/tl/$tag/kt_kernel_trace.csv > gpurun_out/tl/$tag.txt
rm -rf gpurun_out/tl/$tag

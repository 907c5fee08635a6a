#!/bin/bash
# GPU box: the clock each BASELINE kernel sustains on THIS device (one --pmc GRBM_GUI_ACTIVE pass per workload).
#   bash tools/clock_probe.sh <tag>
TAG=${1:-box}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/clock_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in c3 c2 c3_s16; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/$wl -- python3 $REPO/bench.py --workload $wl --steps 100 --warmup 3 --no-cpu-baseline --no-extra > $OUT/$wl.log 2>&1 || { tail -5 $OUT/$wl.log; exit 1; }
  python3 $REPO/tools/clock_probe.py $OUT/$wl "$TAG/$wl"
done

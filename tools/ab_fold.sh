#!/bin/bash
# GPU box: the default (plain) unpack against the folded option, interleaved, bench.py --steps 10 per workload
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2 3; do for fold in 0 1; do for wl in target c3; do
  r=$(DEMOD_FOLD_UNPACK=$fold python bench.py --steps 10 --warmup 2 --workload $wl --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
  echo "$round fold=$fold $wl $r"
done; done; done

#!/bin/bash
# On the GPU box: interleaved A/B of library variants, each with its own segment size.
#   SPECS="name:segrows name:segrows ..." WL=c3 ROUNDS=2 tools/ab_list.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-2}); do
for spec in $SPECS; do
  v=${spec%%:*}; seg=${spec#*:}
  r=$(DEMOD_HIP_LIBRARY=$PWD/demodulator_amd/variants/libdemod_hip_$v.so python bench.py --steps 10 --warmup 2 --workload ${WL:-c3} --segment-rows $seg --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
  echo "$round $v seg$seg $r"
done; done

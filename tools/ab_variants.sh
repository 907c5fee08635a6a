#!/bin/bash
# On the GPU box: interleaved A/B of library variants.  VARIANTS="a b" WLS="c3 target" SEGS="32" ROUNDS=2
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in $(seq 1 ${ROUNDS:-2}); do
for v in ${VARIANTS}; do
  for wl in ${WLS:-c3 c2 target}; do
    for seg in ${SEGS:-32}; do
    r=$(DEMOD_HIP_LIBRARY=$PWD/demodulator_amd/variants/libdemod_hip_$v.so python bench.py --steps 10 --warmup 2 --workload $wl --segment-rows $seg --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
    echo "$round $v $wl seg$seg $r"
    done
  done
done
done

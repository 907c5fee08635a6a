#!/bin/bash
# Laboratory builds of libdemod_hip.so (-DDEMOD_LAB: only the BASELINE configurations' kernels, 15 s each) into
# demodulator_amd/variants/ for A/B runs on the GPU box (tools/ab_variants.sh).
#   tools/build_lab_variants.sh name1:"flags" name2:"flags" ...
set -e
cd "$(dirname "$0")/.."
SRC=demodulator_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -Wno-pass-failed -Iinclude -I$SRC -shared -DDEMOD_LAB"
mkdir -p demodulator_amd/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  [ "$flags" = "$spec" ] && flags=""
  ( /opt/rocm/bin/hipcc $BASE $flags -o demodulator_amd/variants/libdemod_hip_$name.so $SRC/kernels.hip $SRC/api.cpp $SRC/host.cpp $SRC/multi.cpp $SRC/plan.cpp -lpthread -ldl 2>&1 | grep -E "error" || true ) &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls -la demodulator_amd/variants/

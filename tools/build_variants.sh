#!/bin/bash
# Build tuning variants of libdemod_hip.so into demodulator_amd/variants/ (A/B experiments).
#   tools/build_variants.sh name1:"flags" name2:"flags" ...
set -e
cd "$(dirname "$0")/.."
SRC=demodulator_amd/csrc
BASE="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -fPIC -Iinclude -I$SRC -shared"
mkdir -p demodulator_amd/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc $BASE $flags -o demodulator_amd/variants/libdemod_hip_$name.so $SRC/kernels.hip $SRC/api.cpp $SRC/host.cpp $SRC/multi.cpp $SRC/plan.cpp -lpthread -ldl 2>&1 | grep -E "error" || true ) &
done
wait
ls demodulator_amd/variants/

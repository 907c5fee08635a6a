#!/bin/bash
# Builds demodulator_amd/variants/{tsan,asan}/{libdemod_hip.so,libdemod_backend.so,demod} with the given sanitizer (thread by default, or address) on the host side
# (device code untouched) for tools/tsan_cli.sh.  ROCm's clang for the C files: gcc's libtsan lacks the interface clang-built
# objects call (__tsan_memcpy).
set -e
cd "$(dirname "$0")/.."
SAN=${1:-thread}; V=demodulator_amd/variants/${SAN:0:1}san; S=demodulator_amd/csrc; CL=/opt/rocm/lib/llvm/bin/clang
mkdir -p $V
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
    -fPIC -Wno-pass-failed -Iinclude -I$S -Xarch_host -fsanitize=$SAN -shared -o $V/libdemod_hip.so \
    $S/kernels.hip $S/api.cpp $S/host.cpp $S/multi.cpp $S/plan.cpp -lpthread -ldl
$CL -O1 -g -std=gnu99 -fsanitize=$SAN -Iinclude -fPIC -shared -o $V/libdemod_backend.so $S/backend.c -L$V -ldemod_hip -Wl,-rpath,'$ORIGIN' -lpthread
$CL -O1 -g -std=gnu99 -fsanitize=$SAN -Iinclude -o $V/demod $S/main.c -L$V -ldemod_backend -ldemod_hip -Wl,-rpath,'$ORIGIN' -lpthread -lm
ls -la $V

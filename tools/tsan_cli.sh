#!/bin/bash
# GPU box: the CLI and its backend built with a sanitizer on the host side (demodulator_amd/variants/tsan or asan, built
# here by tools/build_sanitized.sh [thread|address]), file -> file, pipe -> pipe, tiny chunks, -q1.
#   bash tools/tsan_cli.sh [tsan|asan]
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=demodulator_amd/variants/${1:-tsan}
D=/dev/shm/tsan.$$; mkdir -p $D
head -c $((96 << 20)) /dev/urandom > $D/in.u8
export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4"
export ASAN_OPTIONS="protect_shadow_gap=0 detect_leaks=0 halt_on_error=0"
N=0
run() { N=$((N + 1)); echo "== $*"; "$@" 2> gpurun_out/tsan_full_$N.txt; rc=$?; echo "$(grep -c -E "WARNING: ThreadSanitizer|ERROR: AddressSanitizer" gpurun_out/tsan_full_$N.txt) reports, rc=$rc (full text: gpurun_out/tsan_full_$N.txt; tools/tsan_filter.py keeps those that do not run through the HIP/HSA runtime)"; }
run $V/demod -i $D/in.u8 -o $D/out.f32 -L 12500 -l 6500
run $V/demod -i $D/in.u8 -o $D/out2.f32 -b -4
run bash -c "cat $D/in.u8 | $V/demod -i - -o - -q 1 > $D/out3.f32"
run bash -c "head -c $((4 << 20)) $D/in.u8 | $V/demod -i - -o $D/out4.f32 -b -10"
cmp $D/out.f32 <(demodulator_amd/demod -i $D/in.u8 -o - -L 12500 -l 6500) && echo "tsan build == product build"
rm -rf $D

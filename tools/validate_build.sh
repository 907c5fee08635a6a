#!/bin/bash
# GPU box: the checks a kernel change has to pass -- bit-identity of 126 cases against the
# round-1 build (if demodulator_amd/variants/libdemod_hip_r1.so is there), the GPU suite,
# the smoke entry, a profile and a bench line.
cd ${GRAFT_REPO_ROOT:-/root/repo}
V=$PWD/demodulator_amd/variants
if [ -f $V/libdemod_hip_r1.so ]; then
python tools/dump_outputs.py $V/libdemod_hip_r1.so /tmp/dump_r1.npz && python tools/dump_outputs.py demodulator_amd/libdemod_hip.so /tmp/dump_new.npz && python tools/dump_outputs.py --compare /tmp/dump_r1.npz /tmp/dump_new.npz
echo "compare rc=$?"
fi
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
bash tools/profile.sh r2_target > /dev/null 2>&1; grep -E "demod_rows|demod_heads" gpurun_out/prof_r2_target/summary.md | head -24
python bench.py > gpurun_out/r2_bench_line.json 2> gpurun_out/r2_bench_err.log; cat gpurun_out/r2_bench_line.json

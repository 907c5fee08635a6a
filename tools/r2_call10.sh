cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/demodulator_amd/variants
python tools/dump_outputs.py $V/libdemod_hip_r1.so /tmp/dump_r1.npz && python tools/dump_outputs.py demodulator_amd/libdemod_hip.so /tmp/dump_new.npz && python tools/dump_outputs.py --compare /tmp/dump_r1.npz /tmp/dump_new.npz
echo "compare rc=$?"
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4
bash tools/profile.sh r2_target
bash tools/profile.sh r2_c2 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --workload c2
python bench.py > gpurun_out/r2_bench_line.json 2> gpurun_out/r2_bench_err.log; tail -c 3000 gpurun_out/r2_bench_line.json

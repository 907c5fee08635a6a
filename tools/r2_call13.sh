cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/demodulator_amd/variants
python tools/dump_outputs.py $V/libdemod_hip_r1.so /tmp/dump_r1.npz && python tools/dump_outputs.py demodulator_amd/libdemod_hip.so /tmp/dump_new.npz && python tools/dump_outputs.py --compare /tmp/dump_r1.npz /tmp/dump_new.npz
echo "compare rc=$?"
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
bash tools/profile.sh r2_target > /dev/null 2>&1; cat gpurun_out/prof_r2_target/summary.md | grep -E "demod_rows|demod_heads" | head -30
python bench.py > gpurun_out/r2_bench_line.json 2> gpurun_out/r2_bench_err.log; cat gpurun_out/r2_bench_line.json
python tools/chunk_sweep.py

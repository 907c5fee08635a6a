set -x
cd $GRAFT_REPO_ROOT
python tools/dump_outputs.py demodulator_amd/variants/libdemod_hip_r1.so /tmp/dump_r1.npz &&
python tools/dump_outputs.py demodulator_amd/libdemod_hip.so /tmp/dump_new.npz &&
python tools/dump_outputs.py --compare /tmp/dump_r1.npz /tmp/dump_new.npz
echo "compare rc=$?"
python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5
echo "== chunk sweep new"; python tools/chunk_sweep.py
echo "== chunk sweep r1"; DEMOD_HIP_LIBRARY=$GRAFT_REPO_ROOT/demodulator_amd/variants/libdemod_hip_r1.so python tools/chunk_sweep.py
echo "== bench new"; python bench.py --no-cpu-baseline
echo "== bench new again"; python bench.py --no-cpu-baseline --no-extra

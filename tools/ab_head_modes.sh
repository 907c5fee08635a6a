set -x
cd $GRAFT_REPO_ROOT
python tools/dump_outputs.py demodulator_amd/variants/libdemod_hip_r1.so /tmp/dump_r1.npz &&
python tools/dump_outputs.py demodulator_amd/libdemod_hip.so /tmp/dump_new.npz &&
python tools/dump_outputs.py --compare /tmp/dump_r1.npz /tmp/dump_new.npz
echo "compare rc=$?"
for rep in 1 2; do
echo "== chunk sweep new, heads in the streaming kernel"; python tools/chunk_sweep.py 1
echo "== chunk sweep new, demod_heads kernel"; python tools/chunk_sweep.py 0
echo "== chunk sweep r1"; DEMOD_HIP_LIBRARY=$GRAFT_REPO_ROOT/demodulator_amd/variants/libdemod_hip_r1.so python tools/chunk_sweep.py
done
python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5

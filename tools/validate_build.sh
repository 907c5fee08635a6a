cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/demodulator_amd/variants
python tools/dump_outputs.py $V/libdemod_hip_r1.so /tmp/dump_r1.npz && python tools/dump_outputs.py demodulator_amd/libdemod_hip.so /tmp/dump_new.npz && python tools/dump_outputs.py --compare /tmp/dump_r1.npz /tmp/dump_new.npz
echo "compare rc=$?"
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python tools/chunk_sweep.py
python tools/chunk_sweep.py 1 | grep -E " (1024|2048|4096) B"

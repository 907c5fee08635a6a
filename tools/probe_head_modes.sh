cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
python tools/head_mode_probe.py 0 1
DEMOD_HIP_LIBRARY=$GRAFT_REPO_ROOT/demodulator_amd/variants/libdemod_hip_r1.so python tools/head_mode_probe.py
done

cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/demodulator_amd/variants
python tools/dump_outputs.py $V/libdemod_hip_cur.so /tmp/dump_cur.npz && python tools/dump_outputs.py $V/libdemod_hip_pk.so /tmp/dump_pk.npz && python tools/dump_outputs.py --compare /tmp/dump_cur.npz /tmp/dump_pk.npz
echo "compare rc=$?"
VARIANTS="cur pk" WLS="target c3 c2" SEGS="0" ROUNDS=3 bash tools/ab_variants.sh

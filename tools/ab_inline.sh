cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for v in cur r16 r4; do for wl in c3 target; do
r=$(DEMOD_HIP_LIBRARY=$PWD/demodulator_amd/variants/libdemod_hip_$v.so python bench.py --steps 10 --warmup 2 --workload $wl --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
echo "$round $v $wl $r"; done; done
for wl in c3 target; do
r=$(DEMOD_HEADS_INLINE=1 DEMOD_HIP_LIBRARY=$PWD/demodulator_amd/variants/libdemod_hip_cur.so python bench.py --steps 10 --warmup 2 --workload $wl --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])")
echo "$round cur-headsinline $wl $r"; done
done

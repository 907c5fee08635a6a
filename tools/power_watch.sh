#!/bin/bash
# GPU box: what rocm-smi says (socket power, shader/memory clocks, temperature) while a workload runs back to back for a few seconds.
#   bash tools/power_watch.sh [workload=c3] [steps=4000]
WL=${1:-c3}; STEPS=${2:-4000}
cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py --workload $WL --steps $STEPS --warmup 3 --no-extra --no-cpu-baseline > /tmp/pw_$WL.json 2>/dev/null &
PID=$!
sleep 4   # imports, set-up, spin-up
for i in $(seq 1 8); do
  kill -0 $PID 2>/dev/null || break
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" | tr -s ' ' | sed "s/^/$WL t$i: /"
  sleep 0.4
done
wait $PID
python -c "import json; d=json.load(open('/tmp/pw_$WL.json')); print('$WL', 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'])"

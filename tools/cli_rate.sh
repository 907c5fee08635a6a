#!/bin/bash
# GPU box: the CLI file -> file / file -> /dev/null with DEMOD_RING_STATS=1 (who waits for whom), 4 GiB of input in /dev/shm
cd ${GRAFT_REPO_ROOT:-/root/repo}
D=/dev/shm/cli_rate.$$; mkdir -p $D
python3 - <<PY
import numpy as np
d = np.frombuffer(np.random.default_rng(1).bytes(1 << 30), dtype=np.uint8)
with open("$D/in.u8", "wb") as f:
    for _ in range(4): f.write(d)
PY
for out in $D/out.f32 /dev/null; do
  for rep in 1 2; do
    rm -f $D/out.f32
    /usr/bin/env time -f "%e s wall" true 2>/dev/null
    s=$(date +%s.%N)
    DEMOD_RING_STATS=1 demodulator_amd/demod -i $D/in.u8 -o $out -L 12500 -l 6500 "$@"
    e=$(date +%s.%N)
    echo "-> $out: $(python3 -c "print(f'{($e - $s):.3f} s, {(1 << 31) / ($e - $s) / 1e6:.0f} Msamples/s including start-up')")"
  done
done
rm -rf $D

#!/usr/bin/env python3
"""GPU box: a few -q1 calls for `rocprofv3 --kernel-trace --stats -- python3 tools/q1_profile.py [chunk_bytes]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
n = 2 << 29
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
d = dm.Demodulator(dm.Options(mode=0x14, buf_size=B, lowpass_in=12500.0, lowpass_out=6500.0))
d.synth_fill(iq, 0, 7)
for _ in range(4):
    d.process(iq, out)
torch.cuda.synchronize()
d.close()

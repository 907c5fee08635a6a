import sys, os
sys.path.insert(0, os.getcwd())
import torch
import demodulator_amd as dm
n = 2 << 29
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
for rnd in range(2):
  for flags, name in ((dict(lowpass_in=12500.0, lowpass_out=6500.0), "-L"), (dict(), "default")):
    for B in (1024, 2048, 4096, 8192, 16384, 65536, 262144):
        for mode in (0, 1):
            d = dm.Demodulator(dm.Options(buf_size=B, **flags)); d.set_head_mode(mode)
            d.synth_fill(iq, 0, 7)
            for _ in range(3): d.process(iq, out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): d.process(iq, out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            print(f"{name:8s} chunk {B:5d} head_mode {mode:2d}: {ms:.3f} ms per 2^29 samples  frac {4 * (1 << 29) / (ms * 1e-3) / 8e12:.3f}", flush=True)
            d.close()

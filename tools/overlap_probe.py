#!/usr/bin/env python3
"""GPU box: does running the dispatch pieces of one call on TWO streams (no barrier between consecutive pieces, so
that a piece's first workgroups start while the previous piece's last wavefronts drain) beat one stream?
Target workload (2^32 samples, -L 12500 -l 6500) cut into pieces of 2^PIECE rows, issued round-robin on 1 or 2 streams;
wall time per pass from events on a third stream that forks and joins them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import demodulator_amd as dm

opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
n = 2 << 32
d = dm.Demodulator(opt)
iq = torch.empty(n, dtype=torch.uint8, device="cuda"); out = torch.empty(n // 4, dtype=torch.float32, device="cuda")
d.synth_fill(iq, 0, 7); torch.cuda.synchronize()
main = torch.cuda.current_stream()
streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]


def one_pass(piece_bytes, n_streams):
    fork = torch.cuda.Event(); fork.record(main)
    for s in streams[:n_streams]:
        s.wait_event(fork)
    k = 0
    for a in range(0, n, piece_bytes):
        s = streams[k % n_streams]
        d.process_ptr(iq.data_ptr() + a, piece_bytes, out.data_ptr() + a, s.cuda_stream)   # whole chunks: every piece ends a "stream"
        k += 1
    for s in streams[:n_streams]:
        e = torch.cuda.Event(); e.record(s); main.wait_event(e)


for _ in range(6):
    one_pass(2 << 30, 1)
torch.cuda.synchronize()
for rnd in range(3):
    for log2 in (31, 30, 29, 28):
        for ns in (1, 2, 3):
            for _ in range(2):
                one_pass(1 << log2, ns)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(6):
                one_pass(1 << log2, ns)
            e1.record(main); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 6
            print(f"round {rnd} pieces of 2^{log2} bytes on {ns} stream(s): {ms:.3f} ms per pass, frac {4 * (1 << 32) / (ms * 1e-3) / 8e12:.4f}", flush=True)

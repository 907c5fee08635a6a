#!/usr/bin/env python3
"""How far the reference is from ITSELF on the BASELINE streams: its generic build against its IS_NATIVE (FMA)
build -- both compiled from /root/reference by oracle/Makefile (oracle/_ref/libdemod_ref.so, libdemod_ref_fma.so) --
under tests/parity.py's criterion, over EVERY chunk.  These counts are the yardstick the full-size GPU tests bound
their own counts with (tests/test_gpu_parity.py: REF_PAIR).  CPU only; about a minute on 8 cores.

    python tools/ref_pair_counts.py [c3] [c5] [c2]
"""
import os, sys
from multiprocessing import Pool
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

C3 = dict(lowpass_in=12500.0, lowpass_out=6500.0)
STREAMS = {   # name: (flags, [(seed, n_bytes)] one entry per independent stream)
    "c2": (dict(), [(0xD3300D02, 2 << 30)]),
    "c3": (C3, [(0xD3300D03, 2 << 30)]),
    "c5": (C3, [(0xD3300D05 + ch, 2 << 24) for ch in range(32)]),
}
SLAB = 64   # chunks per job


def job(args):
    name, seed, first, n_chunks = args
    import demodulator_amd as dm, oracle, parity
    flags = STREAMS[name][0]
    opt = dm.Options(**flags)
    fl = oracle.Flags(**opt.__dict__)
    B = opt.buf_size
    data = oracle.restatement().synth_bytes(seed, first * B, n_chunks * B)
    a = oracle.reference().process(fl, data)
    b = oracle.reference_fma().process(fl, data)
    rep = parity.compare_stream(dm, opt, b, a, data)
    plain = parity.compare(b, a, dm.design(opt)["taps"], parity.inexact_path(opt.lowpass_in, opt.mode), None)
    return name, rep, plain.n_beyond, int((a.view(np.uint32) != b.view(np.uint32)).sum())


def main():
    import oracle
    assert oracle.reference() is not None and oracle.reference_fma() is not None, "needs oracle/_ref (make -C oracle)"
    names = [a for a in sys.argv[1:] if a in STREAMS] or ["c3", "c5"]
    jobs = []
    for name in names:
        flags, streams = STREAMS[name]
        B = 262144
        for seed, n_bytes in streams:
            for first in range(0, n_bytes // B, SLAB):
                jobs.append((name, seed, first, min(SLAB, n_bytes // B - first)))
    tot, beyond_plain, differ = {}, {}, {}
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        for name, rep, nb, nd in pool.imap_unordered(job, jobs):
            tot[name] = rep if name not in tot else tot[name] + rep
            beyond_plain[name] = beyond_plain.get(name, 0) + nb
            differ[name] = differ.get(name, 0) + nd
    for name in names:
        print(f"{name}: reference generic vs reference FMA build, every chunk: {tot[name]}; beyond the stated loose bound without the "
              f"clause or the wrap model: {beyond_plain[name]}; outputs that differ at all: {differ[name]}")


if __name__ == "__main__":
    main()

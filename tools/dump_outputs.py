#!/usr/bin/env python3
"""Run a fixed list of cases through one build of libdemod_hip.so and save every output bit.

    python tools/dump_outputs.py <libdemod_hip.so> <out.npz>
    python tools/dump_outputs.py --compare a.npz b.npz

Used to show that a kernel change leaves results bit-identical (e.g. the chunk heads moved
into the streaming kernel, round 2): the same cases through the old and the new build, then
--compare.  Cases: every golden stream of tests/golden/manifest.json, the BASELINE flag sets
at chunk sizes 1 KiB .. 256 KiB with ragged lengths and segment sizes, degrees 1..8 on both
filters, Chebyshev, -q2, -q3, shard cuts with halos, int16 output."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Params(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("lowpass_in", C.c_double), ("lowpass_out", C.c_double),
                ("epsilon", C.c_double), ("in_degree", C.c_uint32), ("out_degree", C.c_uint32),
                ("mode", C.c_uint32), ("reserved", C.c_uint32), ("buf_size", C.c_uint64)]


class Span(C.Structure):
    _fields_ = [("chunk_offset", C.c_uint64), ("halo", C.c_void_p), ("end_of_stream", C.c_uint32), ("reserved", C.c_uint32)]


def synth(seed, n):
    import oracle
    return oracle.synth_bytes(seed, 0, n)


def cases():
    m = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    for name, c in m["streams"].items():
        f = c["flags"]
        yield name, f, synth(c["seed"], c["n_chunks"] * f["buf_size"]), {}
    base = dict(sample_rate=125000.0, lowpass_in=0.0, lowpass_out=12500.0, epsilon=0.3, in_degree=0, out_degree=3, mode=16)
    c3 = dict(base, lowpass_in=12500.0, lowpass_out=6500.0)
    rng = np.random.default_rng(20261004)
    for fl, tag in ((base, "c2"), (c3, "c3")):
        for b in (10, 11, 12, 14, 16, 18):
            B = 1 << b
            for seg in (0, 1, 2):
                n = 5 * B + int(rng.integers(0, B)) // 4 * 4
                yield f"{tag}_b{B}_seg{seg}", dict(fl, buf_size=B), rng.integers(0, 256, n, dtype=np.uint8), dict(seg=seg)
        yield f"{tag}_s16", dict(fl, buf_size=4096), rng.integers(0, 256, 3 * 4096 + 1000, dtype=np.uint8), dict(s16=0.05)
        yield f"{tag}_shard", dict(fl, buf_size=16384), rng.integers(0, 256, 6 * 16384, dtype=np.uint8), dict(cut=16384 + 3072)
    for d in range(1, 9):
        for D in (0, 1, 4, 6, 8):
            fl = dict(base, out_degree=d, in_degree=D, lowpass_in=12500.0 if D else 0.0, buf_size=2048)
            yield f"d{d}_D{D}", fl, rng.integers(0, 256, 4 * 2048 + 512, dtype=np.uint8), {}
    for mode in (0x11, 0x12, 0x13, 0x18, 0x1C):
        for L in (0.0, 12500.0):
            fl = dict(base, mode=mode, lowpass_in=L, buf_size=32768, out_degree=5)
            yield f"m{mode:x}_L{int(L)}", fl, rng.integers(0, 256, 3 * 32768, dtype=np.uint8), {}


def dump(lib_path, out_path):
    try:
        import torch  # noqa: F401  (same HIP runtime as the tests use)
    except ImportError:
        pass
    lib = C.CDLL(lib_path, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.demod_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(vp)]
    lib.demod_process_host.argtypes = [vp, vp, C.c_uint64, vp, C.POINTER(Span)]
    lib.demod_destroy.argtypes = [vp]
    lib.demod_set_segment_rows.argtypes = [vp, C.c_uint32]
    lib.demod_set_output_format.argtypes = [vp, C.c_int, C.c_float]
    lib.demod_last_error.argtypes = [vp]
    lib.demod_last_error.restype = C.c_char_p
    res = {}
    for name, f, data, extra in cases():
        p = Params(f["sample_rate"], f["lowpass_in"], f["lowpass_out"], f["epsilon"], f["in_degree"], f["out_degree"],
                   f["mode"], 0, f["buf_size"])
        ctx = vp()
        rc = lib.demod_create(C.byref(p), 0, C.byref(ctx))
        if rc != 0:
            res[name] = np.zeros(0, np.float32)
            continue
        if extra.get("seg"):
            lib.demod_set_segment_rows(ctx, extra["seg"])
        dt = np.float32
        if "s16" in extra:
            lib.demod_set_output_format(ctx, 1, extra["s16"])
            dt = np.int16
        data = np.ascontiguousarray(data)
        n = data.size & ~3
        out = np.zeros(n // 4, dtype=dt)
        if "cut" in extra:
            cut = extra["cut"]
            halo = np.ascontiguousarray(data[cut:cut + 64])
            sp = Span(0, halo.ctypes.data, 0, 0)
            rc = lib.demod_process_host(ctx, data.ctypes.data, cut, out.ctypes.data, C.byref(sp))
            sp2 = Span(cut % f["buf_size"], None, 1, 0)
            rc |= lib.demod_process_host(ctx, data[cut:].ctypes.data, n - cut, out[cut // 4:].ctypes.data, C.byref(sp2))
        else:
            sp = Span(0, None, 1, 0)
            rc = lib.demod_process_host(ctx, data.ctypes.data, n, out.ctypes.data, C.byref(sp))
        assert rc == 0, (name, lib.demod_last_error(ctx))
        lib.demod_destroy(ctx)
        res[name] = out
    np.savez(out_path, **res)
    print(f"{lib_path}: {len(res)} cases, {sum(v.size for v in res.values())} outputs -> {out_path}")


def compare(a_path, b_path):
    a, b = np.load(a_path), np.load(b_path)
    assert sorted(a.files) == sorted(b.files)
    bad = 0
    skipped = 0
    for k in a.files:
        x, y = a[k], b[k]
        if x.size == 0 or y.size == 0:      # an option set one of the two builds refuses (e.g. -q1 before round 2)
            skipped += x.size != y.size
            continue
        same = x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))
        if not same:
            bad += 1
            d = np.nonzero(x != y)[0] if x.shape == y.shape else []
            print(f"DIFFERS {k}: {len(d)} outputs, first at {d[:8]}")
    print(f"{len(a.files)} cases compared, {bad} differ, {skipped} skipped (refused by one build), "
          f"{sum(a[k].size for k in a.files)} outputs")
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "--compare":
        sys.exit(compare(sys.argv[2], sys.argv[3]))
    dump(sys.argv[1], sys.argv[2])

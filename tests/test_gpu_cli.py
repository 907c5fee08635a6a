"""End-to-end through real pipes (BASELINE config 1, plumbing): the `demod` CLI
host keeps the reference's -i/-o/-S/-l/-L contract, and the reference's own
host runs unchanged on this repo's backend adapter (processMatrix /
allocateBuffer)."""
import os
import subprocess

import numpy as np
import pytest

import demodulator_amd as dm
import oracle
from parity import check, compare

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMOD = os.path.join(ROOT, "demodulator_amd", "demod")
REFHOST = os.path.join(ROOT, "oracle", "_ref", "demodulator_refhost")


def _run(cmd, data: bytes) -> bytes:
    r = subprocess.run(cmd, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    return r.stdout


def test_cli_readme_dsd_pipe(manifest, golden):
    """cat iq.u8 | demod -i - -o - -S96000 -l12500 (README.md:19-24), 16 KiB chunks."""
    data = golden(manifest["fm_like"]["input"])
    c = manifest["fm_like"]["dsd"]
    out = _run([DEMOD, "-i", "-", "-o", "-", "-S96000", "-l12500", "-b", "-4"], data.tobytes())
    got = np.frombuffer(out, dtype=np.float32)
    want = golden(c["file"])
    assert got.size == want.size == data.size // 4      # byte-exact output length
    opt = dm.Options(**c["flags"])
    check(dm, opt, got, want, data)


def test_cli_readme_multimon_pipe_files(tmp_path, manifest, golden):
    """File in, file out, both lowpasses (README.md:26-31) + a ragged tail."""
    data = golden(manifest["fm_like"]["input"])
    c = manifest["fm_like"]["multimon"]
    src = tmp_path / "iq.u8"
    dst = tmp_path / "out.f32"
    raw = data.tobytes() + bytes(range(200, 241))        # 41 extra bytes: tail of 40 bytes -> 10 outputs
    src.write_bytes(raw)
    _run([DEMOD, "-i", str(src), "-o", str(dst), "-S", "96000", "-l", "6500", "-L", "12500", "-b", "-4"], b"")
    got = np.fromfile(dst, dtype=np.float32)
    opt = dm.Options(**c["flags"])
    want = oracle.restatement().process(oracle.Flags(**opt.__dict__), np.frombuffer(raw, np.uint8))
    assert got.size == want.size == len(raw) // 4
    assert np.array_equal(want[: data.size // 4].view(np.uint32), golden(c["file"]).view(np.uint32))
    check(dm, opt, got, want, np.frombuffer(raw, np.uint8))


def test_cli_default_chunk_many_chunks():
    opt = dm.Options()
    data = oracle.synth_bytes(0xD3300D02, 0, 5 * opt.buf_size + 1000)
    out = _run([DEMOD, "-i", "-", "-o", "-"], data.tobytes())
    got = np.frombuffer(out, dtype=np.float32)
    want = oracle.restatement().process(oracle.Flags(**opt.__dict__), data)
    assert got.size == want.size
    check(dm, opt, got, want, data)


def test_cli_highpass_dc_conditioning(manifest, golden):
    """-q 2 (highpassDc, src/matrix.c:142-157) through the CLI, against the golden
    vector of the compiled reference."""
    c = manifest["streams"]["q2_b32768"]
    opt = dm.Options(**c["flags"])
    data = oracle.synth_bytes(c["seed"], 0, c["n_chunks"] * opt.buf_size)
    out = _run([DEMOD, "-i", "-", "-o", "-", "-q", "2", "-b", "-3"], data.tobytes())
    got = np.frombuffer(out, dtype=np.float32)
    want = golden(c["file"])
    assert got.size == want.size
    check(dm, opt, got, want, data)
    # ... and in front of -L
    c = manifest["streams"]["q2_L_b32768"]
    opt = dm.Options(**c["flags"])
    data = oracle.synth_bytes(c["seed"], 0, c["n_chunks"] * opt.buf_size)
    out = _run([DEMOD, "-i", "-", "-o", "-", "-q", "2", "-L", "12500", "-l", "6500", "-b", "-3"], data.tobytes())
    got = np.frombuffer(out, dtype=np.float32)
    check(dm, opt, got, golden(c["file"]), data)


def test_cli_int16_output_env():
    """DEMOD_OUT_S16=<scale> makes the CLI (through libdemod_backend.so) write signed
    16-bit samples: the stated conversion of what it writes without it."""
    import os
    opt = dm.Options(buf_size=16384)
    data = oracle.synth_bytes(0x517, 0, 4 * opt.buf_size + 400)
    f32 = np.frombuffer(_run([DEMOD, "-i", "-", "-o", "-", "-b", "-4"], data.tobytes()), dtype=np.float32)
    r = subprocess.run([DEMOD, "-i", "-", "-o", "-", "-b", "-4"], input=data.tobytes(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120, env={**os.environ, "DEMOD_OUT_S16": "0.125"})
    assert r.returncode == 0, r.stderr.decode()
    s16 = np.frombuffer(r.stdout, dtype=np.int16)
    assert s16.size == f32.size == data.size // 4
    want = np.rint(np.clip(f32 * np.float32(0.125), -1, 1).astype(np.float64) * 32767.0).astype(np.int16)
    diff = np.abs(s16.astype(np.int32) - want.astype(np.int32))
    assert diff.max() <= 1 and np.count_nonzero(diff) <= 1e-3 * diff.size + 1


def test_cli_rejects_undefined_options():
    r = subprocess.run([DEMOD, "-i", "-", "-o", "-", "-D", "3"], input=b"", stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=60)
    assert r.returncode != 0 and b"-D without -L" in r.stderr


def test_reference_host_on_this_backend(manifest, golden):
    """The reference's own main.c, linked against libdemod_backend.so: same
    flags, same pipe, results within the stated tolerance of its CPU path."""
    if not os.path.exists(REFHOST):
        pytest.skip("oracle/_ref/demodulator_refhost not built (needs /root/reference at build time)")
    data = golden(manifest["fm_like"]["input"])
    for key, flags in (("dsd", ["-S96000", "-l12500"]), ("multimon", ["-S96000", "-l6500", "-L12500"])):
        c = manifest["fm_like"][key]
        out = _run([REFHOST, "-i", "-", "-o", "-", "-b", "-4"] + flags, data.tobytes())
        got = np.frombuffer(out, dtype=np.float32)
        want = golden(c["file"])
        assert got.size == want.size
        opt = dm.Options(**c["flags"])
        check(dm, opt, got, want, data, name=key)


def test_cli_seamless_env(manifest, golden):
    """DEMOD_SEAMLESS=1: the backend adapter holds every hand-off back by one so
    that it can pass the look-ahead; result = the reference with one long chunk."""
    data = golden(manifest["fm_like"]["input"])
    opt = dm.Options(sample_rate=96000.0, lowpass_out=6500.0, lowpass_in=12500.0, buf_size=16384)
    env = dict(os.environ, DEMOD_SEAMLESS="1")
    r = subprocess.run([DEMOD, "-i", "-", "-o", "-", "-S96000", "-l6500", "-L12500", "-b", "-4"],
                       input=data.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
    assert r.returncode == 0, r.stderr.decode()
    got = np.frombuffer(r.stdout, dtype=np.float32)
    want = oracle.restatement().process(oracle.Flags(**{**opt.__dict__, "buf_size": data.size}), data)
    assert got.size == want.size
    assert compare(got, want, dm.design(opt)["taps"], True).ok


def test_cli_input_shorter_than_one_chunk():
    """The whole input is one short read: the producer overwrites args->bufSize with the
    byte count before the backend thread has looked at it (src/main.c:88).  The backend
    reads the field under the mutex and rounds it up to a power of two; a short only-chunk
    is filtered as a chunk of its own length either way."""
    for n, flags, kw in ((1000, [], {}), (4, [], {}), (70000, ["-L", "12500", "-l", "6500"], dict(lowpass_in=12500.0, lowpass_out=6500.0)),
                         (262140, [], {})):
        opt = dm.Options(**kw)
        data = oracle.synth_bytes(0x518 + n, 0, n)
        for _ in range(3):                                   # the race has two outcomes; both must be right
            got = np.frombuffer(_run([DEMOD, "-i", "-", "-o", "-"] + flags, data.tobytes()), dtype=np.float32)
            want = oracle.restatement().process(oracle.Flags(**opt.__dict__), data)
            assert got.size == want.size == n // 4
            check(dm, opt, got, want, data, name=f"short input {n}")


def test_cli_tiny_chunks_many_handoffs():
    """-b -12 (64-byte chunks): every hand-off is one chunk through the literal kernel, two in
    flight on different streams sharing one scratch buffer."""
    for flags, kw, b, B in (([], {}, "-12", 64), (["-L", "12500", "-l", "6500"], dict(lowpass_in=12500.0, lowpass_out=6500.0), "-13", 32)):
        opt = dm.Options(buf_size=B, **kw)
        data = oracle.synth_bytes(0x519, 0, 6000 * B + 20)
        got = np.frombuffer(_run([DEMOD, "-i", "-", "-o", "-", "-b", b] + flags, data.tobytes()), dtype=np.float32)
        want = oracle.restatement().process(oracle.Flags(**opt.__dict__), data)
        assert got.size == want.size
        check(dm, opt, got, want, data, name=f"cli -b {b}")


def test_cli_large_file_to_file_in_batches(tmp_path):
    """40 MiB + a ragged tail, file in, file out, default chunk: the hand-offs travel through
    the pinned ring in batches while a second thread writes the results; the bytes are those
    of one demod_process_host call."""
    opt = dm.Options(lowpass_in=12500.0, lowpass_out=6500.0)
    data = oracle.synth_bytes(0x51A, 0, (40 << 20) + 70000)
    src, dst = tmp_path / "iq.u8", tmp_path / "out.f32"
    src.write_bytes(data.tobytes())
    r = subprocess.run([DEMOD, "-i", str(src), "-o", str(dst), "-L", "12500", "-l", "6500"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120, env={**os.environ, "DEMOD_RING_STATS": "1"})
    assert r.returncode == 0, r.stderr.decode()
    assert b"demod ring:" in r.stderr and b"161 hand-offs" in r.stderr, r.stderr
    got = np.fromfile(dst, dtype=np.float32)
    with dm.Demodulator(opt) as d:
        want = d.process_host(data)
    assert got.size == want.size == data.size // 4
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # ... and through a pipe (fwrite) the same
    out = _run([DEMOD, "-i", str(src), "-o", "-", "-L", "12500", "-l", "6500"], b"")
    assert np.array_equal(np.frombuffer(out, dtype=np.uint32), want.view(np.uint32))


def test_cli_correct_iq_q1(manifest, golden):
    """-q 1 (correctIq) through the CLI against the golden vector of the compiled reference: every
    hand-off is a call of its own, the running offsets carry over on the device."""
    for name, extra in (("q1_b32768", ["-b", "-3"]), ("q1_L_b16384", ["-b", "-4", "-L", "12500", "-l", "6500"]), ("q1_b1024", ["-b", "-8"])):
        c = manifest["streams"][name]
        opt = dm.Options(**c["flags"])
        data = oracle.synth_bytes(c["seed"], 0, c["n_chunks"] * opt.buf_size)
        out = _run([DEMOD, "-i", "-", "-o", "-", "-q", "1"] + extra, data.tobytes())
        got = np.frombuffer(out, dtype=np.float32)
        assert got.size == data.size // 4
        check(dm, opt, got, golden(c["file"]), data, name=name)

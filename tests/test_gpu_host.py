"""The host-buffer paths of the C ABI (demodulator_amd/csrc/host.cpp): the three-stream
pipeline behind demod_process_host (pinned memory in place, pageable memory staged), and the
streaming ring the backend adapter drives (hand-offs laid in a pinned ring, batched).
What the reference does around its per-chunk pipeline: /root/reference/src/main.c:58-98,
src/matrix.c:236-273."""
import threading
import time

import numpy as np
import pytest

import demodulator_amd as dm
import oracle
from parity import check

pytestmark = pytest.mark.gpu

C3 = dict(lowpass_in=12500.0, lowpass_out=6500.0)


def _drive_ring(ring, data, chunk, *, delay=0.0):
    """Producer in this thread, consumer in another (as processMatrix and its writer thread)."""
    parts, errors = [], []

    def consume():
        try:
            while True:
                got = ring.take()
                if got is None:
                    return
                parts.append(got)
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    t = threading.Thread(target=consume)
    t.start()
    n = data.size
    pos = 0
    while pos < n:
        step = min(chunk, n - pos)
        last = pos + step == n
        ring.put(data[pos:pos + step], final=last and step < chunk)
        pos += step
        if delay:
            time.sleep(delay)
    if n % chunk == 0:
        ring.put(np.zeros(0, np.uint8), final=True)       # EOF exactly on a chunk boundary
    t.join(timeout=120)
    assert not t.is_alive() and not errors, errors
    return (np.concatenate(parts) if parts else np.zeros(0, np.float32)), [p.size for p in parts]


@pytest.mark.parametrize("flags,B,n_chunks,extra,slots", [
    (C3, 262144, 70, 1000, None),        # default chunk: three 8 MiB batches and more, short last chunk
    ({}, 16384, 1500, 0, None),          # EOF on a chunk boundary
    (C3, 4096, 300, 2052, 5),            # a 5-slot arena of the caller's: wrap-around and back-pressure all the time
    ({}, 1024, 5000, 4, None),           # smallest ring chunk
], ids=["c3_default", "c2_16k", "c3_4k_5slots", "c2_1k"])
def test_ring_equals_one_call_and_the_oracle(flags, B, n_chunks, extra, slots):
    opt = dm.Options(buf_size=B, **flags)
    data = oracle.synth_bytes(0x7A00 + B, 0, n_chunks * B + extra)
    want = oracle.restatement().process(oracle.Flags(**opt.__dict__), data, threads=8)
    with dm.Demodulator(opt) as d:
        whole = d.process_host(data)
        with d.ring(slots) as ring:
            got, sizes = _drive_ring(ring, data, B)
        assert got.size == want.size == data.size // 4
        # batching must not change a bit: chunks are independent, every batch is whole chunks
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
        check(dm, opt, got, want, data, name=f"ring B={B}")
        if slots is None and B <= 16384:
            assert max(sizes) > B // 4                            # a fast producer's small hand-offs leave in batches
        assert max(sizes) <= (8 << 20) // 4                       # ... of at most 8 MiB


def test_ring_slow_producer_and_int16():
    """Hand-offs that trickle in leave one by one (the GPU is idle when each arrives); the
    int16 output format goes through the ring like float32."""
    opt = dm.Options(buf_size=65536, **C3)
    data = oracle.synth_bytes(0x7B00, 0, 12 * 65536 + 600)
    with dm.Demodulator(opt) as d:
        whole = d.process_host(data)
        with d.ring() as ring:
            got, sizes = _drive_ring(ring, data, 65536, delay=0.02)
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
        assert sizes.count(65536 // 4) >= 8                         # mostly single chunks
        d.set_output_format("s16", 0.05)
        whole16 = d.process_host(data)
        with d.ring() as ring:
            got16, _ = _drive_ring(ring, data, 65536)
        assert got16.dtype == np.int16 and np.array_equal(got16, whole16)


def test_ring_misuse_is_reported():
    with dm.Demodulator(dm.Options(buf_size=4096)) as d:
        with d.ring() as ring:
            with pytest.raises(dm.DemodError, match="whole chunk"):
                ring.put(np.zeros(1000, np.uint8))                   # short, but not final
            ring.put(np.zeros(4096, np.uint8))
            ring.put(np.zeros(100, np.uint8), final=True)
            sizes = []
            while (got := ring.take()) is not None:
                sizes.append(got.size)
            assert sum(sizes) == 1024 + 25 and ring.take() is None
            with pytest.raises(dm.DemodError):
                ring.put(np.zeros(4096, np.uint8))                   # after the end
            with pytest.raises(dm.DemodError, match="ring is open"):
                d.set_output_format("s16", 1.0)
            # ... and so is everything else that would change the results of a stream under way
            with pytest.raises(dm.DemodError, match="ring is open"):
                d.set_seamless(True)
            d.set_seamless(False)                                    # (no change: accepted)
    with dm.Demodulator(dm.Options(buf_size=4096, lowpass_in=12500.0, lowpass_out=6500.0)) as d:
        with d.ring():
            with pytest.raises(dm.DemodError, match="ring is open"):
                d.set_stencil_form("plain")
            d.set_stencil_form("auto")
        d.set_stencil_form("plain")                                  # the ring is closed: accepted
        assert d.stencil_form == "plain"
    with dm.Demodulator(dm.Options(buf_size=64)) as d:
        with pytest.raises(dm.DemodError, match="1 KiB"):
            d.ring()


@pytest.mark.parametrize("flags,B", [(C3, 262144), ({}, 16384), (C3, 64)], ids=["c3", "c2_16k", "c3_b64"])
def test_process_host_pinned_in_place_equals_staged(flags, B):
    """demod_process_host DMA's pinned caller memory in place and stages pageable memory
    through pinned buffers with copy threads: same bits either way, over more pieces than
    the pipeline is deep, with a chunk offset and a halo, and against the oracle."""
    import torch
    opt = dm.Options(buf_size=B, **flags)
    n = (100 << 20) + 5 * 1024 + 8
    data = oracle.synth_bytes(0x7C00 + B, 0, n)
    with dm.Demodulator(opt) as d:
        staged = d.process_host(data)
        pin_in = torch.from_numpy(data.copy()).pin_memory()
        pin_out = torch.empty(n // 4, dtype=torch.float32).pin_memory()
        span = dm._native.Span(0, None, 1, 0)
        import ctypes as C
        d._check(d._lib.demod_process_host(d._ctx, pin_in.data_ptr(), n, pin_out.data_ptr(), C.byref(span)), "process_host")
        assert np.array_equal(pin_out.numpy().view(np.uint32), staged.view(np.uint32))
        want = oracle.restatement().process(oracle.Flags(**opt.__dict__), data, threads=8)
        check(dm, opt, staged, want, data, name=f"process_host B={B}")
        if B >= 1024:
            cut = 40 * (1 << 20) + 3 * 1024                          # a continuing stream cut inside a chunk
            a = d.process_host(data[:cut], halo=data[cut:cut + 64], end_of_stream=False)
            b = d.process_host(data[cut:], chunk_offset=cut % B)
            assert np.array_equal(np.concatenate([a, b]).view(np.uint32), staged.view(np.uint32))

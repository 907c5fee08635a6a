"""demod_multi_*: the hot path over several devices from one process (multi.cpp).  Only one
MI355X is visible here: n = 1, and n = 2 / 3 contexts on the SAME device, where RCCL refuses
the device list and the halos travel by peer copy (the stated fallback) -- the planner, the
per-device streams, the halo hand-over and the channel batching are exercised; RCCL
send/recv between distinct devices is not (unmeasured, said so in DESIGN.md)."""
import numpy as np
import pytest

import demodulator_amd as dm
import oracle
from parity import check

pytestmark = pytest.mark.gpu

C3 = dict(lowpass_in=12500.0, lowpass_out=6500.0)


def _run(m, data, shards):
    import torch
    iq = torch.from_numpy(data).cuda()
    ins = [iq[s.start:s.start + s.length].clone() if s.length else torch.zeros(16, dtype=torch.uint8, device="cuda") for s in shards]
    outs = [torch.full((max(4, s.length // 4),), float("nan"), dtype=torch.float32, device="cuda") for s in shards]
    for _ in range(2):          # a second step reuses communicator / streams
        m.process(ins, outs, shards)
    m.synchronize()
    return np.concatenate([o[: s.length // 4].cpu().numpy() for o, s in zip(outs, shards)])


@pytest.mark.parametrize("n,phase", [(1, 0), (2, 0), (2, 3072), (3, 5120)])
def test_shards_on_n_contexts_equal_the_whole_stream(n, phase, monkeypatch):
    opt = dm.Options(buf_size=16384, **C3)
    total = 11 * 16384 + 5 * 1024 + 100
    # the stream starts `phase` bytes into a chunk: the oracle sees those bytes as well
    data = oracle.synth_bytes(0x9000 + n, 0, phase + total)
    want = oracle.restatement().process(oracle.Flags(**opt.__dict__), data)[phase // 4:]
    with dm.Demodulator(opt) as d:
        import torch
        whole = d.process(torch.from_numpy(data[phase:].copy()).cuda(), chunk_offset=phase).cpu().numpy()
    with dm.MultiDemodulator(opt, [0] * n) as m:
        shards = m.plan(total & ~3, phase)
        assert sum(s.length for s in shards) == total & ~3
        got = _run(m, data[phase:phase + (total & ~3)], shards)
        cut_inside = any(s.needs_halo for s in shards)
        assert m.halo_transport == ("peer-copy" if cut_inside else "none")   # RCCL refuses one device listed twice
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
    # interior outputs look forward only: what lies before the stream's first byte does not matter
    H = dm.design(opt)["head_outputs"]
    n0 = (16384 - phase) // 4 if phase else 0
    check(dm, opt, got[n0:], want[n0:], None, name=f"multi n={n} phase={phase}")
    assert H <= 8


def test_forced_peer_copy_and_channels(monkeypatch):
    """BASELINE config 5 in small through demod_plan_channels: 7 channels over 2 contexts,
    whole channels each, no exchange; every channel equals the channel alone."""
    import torch
    monkeypatch.setenv("DEMOD_MULTI_TRANSPORT", "peer")
    opt = dm.Options(buf_size=16384, **C3)
    per = 4 * 16384
    chans = [oracle.synth_bytes(0xD3300D05 + c, 0, per) for c in range(7)]
    data = np.concatenate(chans)
    with dm.MultiDemodulator(opt, [0, 0]) as m:
        shards = m.plan_channels(7, per)
        assert [s.length // per for s in shards] == [4, 3] and not any(s.needs_halo or s.chunk_offset for s in shards)
        got = _run(m, data, shards).reshape(7, -1)
        assert m.halo_transport == "none"
        with dm.Demodulator(opt) as d:
            for c in range(7):
                alone = d.process(torch.from_numpy(chans[c]).cuda()).cpu().numpy()
                assert np.array_equal(alone.view(np.uint32), got[c].view(np.uint32)), c
        # and a sharded stream with the copy forced
        total = 6 * 16384 + 2048
        stream = oracle.synth_bytes(0x9100, 0, total)
        sh = m.plan(total, 1024)
        got = _run(m, stream, sh)
        assert m.halo_transport == "peer-copy"
        with dm.Demodulator(opt) as d:
            whole = d.process(torch.from_numpy(stream).cuda(), chunk_offset=1024).cpu().numpy()
        assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))


def test_q1_is_refused_over_several_devices():
    """-q1's running offsets belong to ONE stream walked in order (src/matrix.c:125: static state): every shard but the
    first would start them from zero, and a device's batched channels would hand them from one channel to the next.
    demod_multi_create refuses the option set instead of computing something else silently."""
    with pytest.raises(dm.DemodError, match="cannot be sharded"):
        dm.MultiDemodulator(dm.Options(mode=0x14), [0])
    with pytest.raises(dm.DemodError, match="cannot be sharded"):
        dm.MultiDemodulator(dm.Options(mode=0x14, **C3), [0, 0])

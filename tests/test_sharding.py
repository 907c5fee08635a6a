"""Multi-GPU path on CPU: the shard plan and the halo exchange (world_size 2 and
3 over gloo).  The arithmetic of a shard with a halo is covered on the GPU in
test_gpu_parity.py::test_shards_with_halo_equal_whole_stream; here the plumbing
around it is checked with the real torch.distributed point-to-point calls."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from demodulator_amd import HALO_BYTES
from demodulator_amd.sharding import plan_shards, exchange_halo, run_probe_child


def test_plan_chunk_aligned_needs_no_exchange():
    B = 262144
    shards = plan_shards(8 * (1 << 20) * 8, 8, B)
    assert [s.length for s in shards] == [8 << 20] * 8
    assert all(s.chunk_offset == 0 and not s.needs_halo and not s.sends_halo for s in shards)
    assert shards[-1].end_of_stream and not shards[0].end_of_stream


def test_plan_unaligned_cuts():
    B = 16384
    total = 6 * B + 5000
    shards = plan_shards(total, 4, B, align=1024)
    assert shards[0].start == 0 and shards[-1].stop == total
    for a, b in zip(shards[:-1], shards[1:]):
        assert a.stop == b.start and a.stop % 1024 == 0
        assert b.chunk_offset == b.start % B
        assert a.needs_halo == (a.stop % B != 0) == b.sends_halo
    # phase: the stream itself starts 3 KiB into a chunk
    sh = plan_shards(4 * B, 2, B, phase_bytes=3072)
    assert sh[0].chunk_offset == 3072 and sh[1].chunk_offset == (3072 + sh[1].start) % B
    assert sh[0].needs_halo and sh[1].sends_halo


def test_plan_tiny_chunks_and_degenerate():
    sh = plan_shards(64 * 100, 3, 64)
    assert all(s.start % 64 == 0 and not s.needs_halo for s in sh) and sh[-1].stop == 6400
    sh = plan_shards(1024, 4, 262144)          # fewer rows than ranks: empty shards are legal
    assert sum(s.length for s in sh) == 1024 and not any(s.needs_halo and s.length == 0 for s in sh)
    assert not any(s.sends_halo or s.needs_halo for s in sh)      # one row: nothing follows it
    assert [s.end_of_stream for s in sh] == [s.stop == 1024 for s in sh]


def test_plan_empty_shards_take_no_part_in_the_exchange():
    """Every send has exactly one matching receive, also when shards are empty: the
    look-ahead of a shard comes from the next NON-EMPTY shard."""
    B = 262144
    for total, world in ((3 * 1024, 4), (2 * 1024 + 40, 5), (5 * 1024, 8), (7 * 1024 + 4, 3)):
        sh = plan_shards(total, world, B, align=1024)
        assert sum(s.length for s in sh) == total
        for s in sh:
            if s.length == 0:
                assert not s.needs_halo and not s.sends_halo
            if s.needs_halo:
                src = sh[s.halo_src]
                assert src.length > 0 and src.start == s.stop and src.halo_dst == s.rank
            if s.sends_halo:
                dst = sh[s.halo_dst]
                assert dst.needs_halo and dst.halo_src == s.rank
        last = max(s.rank for s in sh if s.length)
        assert sh[last].end_of_stream and not sh[last].needs_halo
        assert sum(s.needs_halo for s in sh) == sum(s.sends_halo for s in sh)
    with pytest.raises(ValueError):
        plan_shards(1 << 20, 2, 16384, phase_bytes=100)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, B, phase, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shards = plan_shards(total, world, B, phase_bytes=phase, align=1024)
        me = shards[rank]
        right = shards[me.halo_src] if me.needs_halo else None
        iq = torch.from_numpy(oracle.synth_bytes(seed, me.start, me.length))
        halo = torch.zeros(HALO_BYTES, dtype=torch.uint8)
        for _ in range(3):                      # a few steps, as bench.py does
            exchange_halo(me, right, iq, halo)
        dist.barrier()
        q.put((rank, me, halo.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,phase", [(2, 0), (2, 3072), (3, 1024), (8, 5120)])
def test_halo_exchange_over_gloo(world, phase):
    B, seed = 16384, 0xD3300D04
    total = 5 * B + 7 * 1024
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, B, phase, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    stream = oracle.synth_bytes(seed, 0, total)
    for rank, shard, halo in got:
        if shard.needs_halo:
            want = stream[shard.stop: shard.stop + HALO_BYTES]
            assert np.array_equal(halo[: want.size], want), rank
        else:
            assert not halo.any()                # untouched: no message was sent to this rank


def _probe_worker(rank, world, port, q, force_fail):
    q.put((rank, run_probe_child("gloo", rank, world, rank, port, 60.0, force_fail=force_fail)))


@pytest.mark.parametrize("force_fail", [False, True])
def test_exchange_probe_in_child_processes(force_fail):
    """The health check bench.py runs before trusting the point-to-point path: one
    throw-away process per rank with a rendezvous of its own; a failing probe is
    reported (False), never raised into the caller."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_probe_worker, args=(r, world, port, q, force_fail)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert got == {0: not force_fail, 1: not force_fail}


def test_exchange_probe_that_hangs_is_killed():
    """A probe whose peer never shows up is killed after its timeout (by PID) and reads as failed."""
    import time
    t0 = time.time()
    assert run_probe_child("gloo", 0, 2, 0, _free_port(), 8.0) is False
    assert time.time() - t0 < 60


def test_c_planner_equals_python_planner():
    """demod_plan_shards (multi.cpp, what demod_multi_* uses) and sharding.plan_shards (what
    bench.py's one-process-per-GPU flow uses) are the same function."""
    import random
    import demodulator_amd as dm
    rnd = random.Random(7)
    for _ in range(600):
        B = 1 << rnd.randint(2, 18)
        world = rnd.randint(1, 9)
        phase = rnd.randrange(0, max(1, B // 1024)) * 1024 if B >= 1024 else 0
        total = rnd.randrange(0, 40 * B + 5000)
        align = rnd.choice([0, 0, 1024, B]) if B >= 1024 else 0
        py = plan_shards(total, world, B, phase_bytes=phase, align=(align or None))
        c = dm.plan_shards_c(total, world, B, phase, align)
        for a, b in zip(py, c):
            assert (a.start, a.length, a.chunk_offset, int(a.needs_halo), int(a.end_of_stream), a.halo_src, a.halo_dst) == \
                   (b.start, b.length, b.chunk_offset, b.needs_halo, b.end_of_stream, b.halo_src, b.halo_dst), (B, world, phase, total)
    with pytest.raises(dm.DemodError):
        dm.plan_shards_c(1 << 20, 2, 16384, 100)          # phase not a multiple of 1024


def test_probe_under_torch_distributed_run():
    """bench.py's probe sequence started the way the driver starts it: under
    torch.distributed.run the environment tells env:// rendezvous to use the elastic agent's
    store, which does not exist on the probe's own port -- the probe child must not inherit
    that (it would wait for a store nobody serves and the run would end as "halo_exchange":
    "failed")."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(here, "_probe_under_launcher.py")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0 and b"PROBE_OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])

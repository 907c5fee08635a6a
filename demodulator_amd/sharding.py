"""Sharding of one IQ stream over the GPUs of a node.

The reference filters every chunk on its own (zero padding, no carried state,
/root/reference/src/matrix.c:272), so a stream splits into contiguous shards
with NO data exchange when the cuts fall on chunk boundaries.  When a cut
falls inside a chunk, the look-ahead stencils of the rank on its left need the
first HALO_BYTES bytes that follow it in the stream: one point-to-point message
per cut (RCCL send/recv over xGMI with backend "nccl"; gloo in the CPU tests).
Nothing is ever reduced or gathered -- this path has no collective.

`probe_exchange()` / `python -m demodulator_amd.sharding --probe` is the health
check bench.py runs in FRESH child processes before it trusts the RCCL
point-to-point path: a hung probe is killed from outside and the benchmark
process itself never holds a stuck communicator.
"""
from __future__ import annotations

from dataclasses import dataclass

from ._native import HALO_BYTES, ROW_BYTES


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    start: int           # first byte of the shard in the global stream
    length: int          # bytes
    chunk_offset: int    # bytes between the enclosing chunk's start and `start`
    needs_halo: bool     # the shard ends inside a chunk and bytes follow it in the stream
    end_of_stream: bool  # nothing follows this shard (the last non-empty shard and every empty one after it)
    halo_src: int = -1   # rank that holds the bytes following this shard (needs_halo only)
    halo_dst: int = -1   # rank whose look-ahead this shard's first bytes are, or -1

    @property
    def stop(self) -> int:
        return self.start + self.length

    @property
    def sends_halo(self) -> bool:
        """This rank's first HALO_BYTES bytes are somebody's look-ahead."""
        return self.halo_dst >= 0


def plan_shards(total_bytes: int, world: int, chunk_bytes: int, *, phase_bytes: int = 0,
                align: int | None = None) -> list[Shard]:
    """Cut [0, total_bytes) into `world` contiguous shards of (nearly) equal
    size.  Cuts are multiples of `align` bytes (default: the chunk size when it
    divides the even share, else 1 KiB rows, which is what the kernel needs).
    `phase_bytes` is how far into a chunk the stream itself starts.

    Shards may come out empty (fewer rows than ranks).  An empty shard takes no
    part in any exchange: the look-ahead of a shard comes from the next
    NON-EMPTY shard (`halo_src`), and that shard alone sends (`halo_dst`), so
    every send has exactly one matching receive."""
    if world < 1 or total_bytes < 0:
        raise ValueError("world >= 1 and total_bytes >= 0 required")
    if phase_bytes % ROW_BYTES or not 0 <= phase_bytes < max(chunk_bytes, 1):
        raise ValueError("phase_bytes must be a multiple of 1024 below the chunk size")
    share = total_bytes // world
    if align is None:
        align = chunk_bytes if (chunk_bytes >= ROW_BYTES and phase_bytes == 0
                                and share >= chunk_bytes) else ROW_BYTES
    if chunk_bytes < ROW_BYTES:
        align = chunk_bytes  # tiny chunks: whole chunks only
    cuts = [0]
    for r in range(1, world):
        cuts.append(min(total_bytes, (share * r) // align * align))
    cuts.append(total_bytes)
    nonempty = [r for r in range(world) if cuts[r + 1] > cuts[r]]
    nxt = {r: (nonempty[i + 1] if i + 1 < len(nonempty) else -1) for i, r in enumerate(nonempty)}
    needs = {}
    for r in nonempty:
        stop = cuts[r + 1]
        needs[r] = (chunk_bytes >= ROW_BYTES and (phase_bytes + stop) % chunk_bytes != 0
                    and stop < total_bytes and nxt[r] >= 0)
    dst = {nxt[r]: r for r in nonempty if needs[r]}
    shards = []
    for r in range(world):
        start, stop = cuts[r], cuts[r + 1]
        shards.append(Shard(rank=r, world=world, start=start, length=stop - start,
                            chunk_offset=(phase_bytes + start) % chunk_bytes,
                            needs_halo=needs.get(r, False), end_of_stream=stop == total_bytes,
                            halo_src=nxt[r] if needs.get(r, False) else -1, halo_dst=dst.get(r, -1)))
    return shards


def exchange_halo(shard: Shard, right: Shard | None, iq, halo, group=None) -> None:
    """Blocking halo exchange for one step.  `iq` is this rank's shard (1-D uint8
    tensor, CPU for gloo / CUDA for nccl), `halo` a uint8 tensor of HALO_BYTES
    that receives the first bytes of the next non-empty shard when this shard
    needs them (`right` is that shard, shards[shard.halo_src]; a source shorter
    than HALO_BYTES can only be the stream's last bytes, and what lies behind the
    stream end is zeros: the rest of `halo` is cleared).
    Uses one batched isend/irecv pair per rank (RCCL ncclSend/ncclRecv grouped)."""
    import torch.distributed as dist

    ops = []
    if shard.sends_halo:
        n = min(HALO_BYTES, shard.length)
        ops.append(dist.P2POp(dist.isend, iq[:n], shard.halo_dst, group))
    if shard.needs_halo:
        n = HALO_BYTES if right is None else min(HALO_BYTES, right.length)
        if n < HALO_BYTES:
            halo[n:].zero_()
        ops.append(dist.P2POp(dist.irecv, halo[:n], shard.halo_src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


# ---------------------------------------------------------------------------
# health check of the point-to-point path, run in throw-away processes
# ---------------------------------------------------------------------------

def probe_exchange(backend: str, rank: int, world: int, device: int | None = None) -> None:
    """One process of the probe: joins a process group of its own (MASTER_ADDR /
    MASTER_PORT from the environment), sends HALO_BYTES to the rank on its left and
    receives from the rank on its right exactly as exchange_halo does every step,
    and checks the bytes.  Raises (or hangs, for the caller to kill) on failure."""
    import torch
    import torch.distributed as dist

    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        if backend == "nccl":
            dev = torch.device("cuda", device if device is not None else 0)
            torch.cuda.set_device(dev)
        else:
            dev = torch.device("cpu")
        total = world * 4 * ROW_BYTES
        shards = plan_shards(total, world, 8 * ROW_BYTES, phase_bytes=ROW_BYTES, align=ROW_BYTES)
        me = shards[rank]
        right = shards[me.halo_src] if me.needs_halo else None
        iq = (torch.arange(me.length, dtype=torch.int64) + 7 * me.start).to(torch.uint8).to(dev)
        halo = torch.zeros(HALO_BYTES, dtype=torch.uint8, device=dev)
        if backend == "nccl":
            # every rank takes part in one collective first, so that the communicator exists
            # before the point-to-point messages (which not every rank sends or receives)
            dist.all_reduce(torch.zeros(1, device=dev))
        for _ in range(2):
            exchange_halo(me, right, iq, halo)
        if backend == "nccl":
            torch.cuda.synchronize()
        if me.needs_halo:
            want = (torch.arange(HALO_BYTES, dtype=torch.int64) + 7 * right.start).to(torch.uint8)
            if not torch.equal(halo.cpu(), want):
                raise RuntimeError(f"probe rank {rank}: halo bytes differ")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def run_probe_child(backend: str, rank: int, world: int, local_rank: int, port: int, timeout_s: float,
                    force_fail: bool = False) -> bool:
    """Start this rank's probe process (fresh interpreter, rendezvous on `port`),
    wait for it at most timeout_s, kill it by PID if it has not finished.  True
    only if it exited 0.  The calling process makes no RCCL call here."""
    import os
    import subprocess
    import sys

    # a rendezvous of its own: under torch.distributed.run the parent's environment says "use the
    # elastic agent's store" (TORCHELASTIC_USE_AGENT_STORE), which only exists on the parent's port
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
               LOCAL_RANK=str(local_rank))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "demodulator_amd.sharding", "--probe", backend]
    if force_fail:
        cmd.append("--fail")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=root)
    try:
        _, err = p.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        p.kill()
        _, err = p.communicate()
        sys.stderr.write(f"[halo probe rank {rank}] no answer within {timeout_s:.0f} s: killed\n")
        return False
    if p.returncode != 0:
        tail = err.decode(errors="replace").strip().splitlines()[-3:]
        sys.stderr.write(f"[halo probe rank {rank}] exit {p.returncode}: " + " | ".join(tail) + "\n")
    return p.returncode == 0


def _probe_main(argv) -> int:
    import os

    backend = argv[argv.index("--probe") + 1]
    if "--fail" in argv:      # rehearsal of the failure path (bench.py DEMOD_BENCH_FORCE_PROBE_FAIL=1)
        return 3
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = None
    if backend == "nccl":
        import torch

        device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    probe_exchange(backend, rank, world, device)
    return 0


if __name__ == "__main__":
    import sys

    sys.exit(_probe_main(sys.argv))

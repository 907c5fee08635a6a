"""Sharding of one IQ stream over the GPUs of a node.

The reference filters every chunk on its own (zero padding, no carried state,
/root/reference/src/matrix.c:272), so a stream splits into contiguous shards
with NO data exchange when the cuts fall on chunk boundaries.  When a cut
falls inside a chunk, the look-ahead stencils of the rank on its left need the
first HALO_BYTES bytes of the rank on its right: one point-to-point message per
cut (RCCL send/recv over xGMI with backend "nccl"; gloo in the CPU tests).
Nothing is ever reduced or gathered -- this path has no collective.
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
    needs_halo: bool     # the shard ends inside a chunk of a continuing stream
    end_of_stream: bool  # last shard

    @property
    def stop(self) -> int:
        return self.start + self.length

    @property
    def sends_halo(self) -> bool:
        """This rank's first HALO_BYTES bytes are somebody's look-ahead."""
        return self.rank > 0 and self.chunk_offset != 0


def plan_shards(total_bytes: int, world: int, chunk_bytes: int, *, phase_bytes: int = 0,
                align: int | None = None) -> list[Shard]:
    """Cut [0, total_bytes) into `world` contiguous shards of (nearly) equal
    size.  Cuts are multiples of `align` bytes (default: the chunk size when it
    divides the even share, else 1 KiB rows, which is what the kernel needs).
    `phase_bytes` is how far into a chunk the stream itself starts."""
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
    shards = []
    for r in range(world):
        start, stop = cuts[r], cuts[r + 1]
        last = r == world - 1
        off = (phase_bytes + start) % chunk_bytes
        ends_inside = (phase_bytes + stop) % chunk_bytes != 0
        shards.append(Shard(rank=r, world=world, start=start, length=stop - start, chunk_offset=off,
                            needs_halo=ends_inside and not last and stop > start, end_of_stream=last))
    return shards


def exchange_halo(shard: Shard, right: Shard | None, iq, halo, group=None) -> None:
    """Blocking halo exchange for one step.  `iq` is this rank's shard (1-D uint8
    tensor, CPU for gloo / CUDA for nccl), `halo` a uint8 tensor of HALO_BYTES
    that receives the first bytes of rank+1's shard when this shard needs them.
    Uses one batched isend/irecv pair per rank (RCCL ncclSend/ncclRecv grouped)."""
    import torch.distributed as dist

    ops = []
    if shard.sends_halo:
        n = min(HALO_BYTES, shard.length)
        ops.append(dist.P2POp(dist.isend, iq[:n], shard.rank - 1, group))
    if shard.needs_halo:
        n = HALO_BYTES if right is None else min(HALO_BYTES, right.length)
        ops.append(dist.P2POp(dist.irecv, halo[:n], shard.rank + 1, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()

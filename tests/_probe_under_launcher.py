"""Helper of tests/test_sharding.py::test_probe_under_torch_distributed_run: what bench.py does
around its halo probe, on CPU -- started by `python -m torch.distributed.run`, a gloo control
group, rank 0 picks a port, every rank runs the probe in a child process, all vote."""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demodulator_amd.sharding import run_probe_child  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
port = [0]
if rank == 0:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port[0] = s.getsockname()[1]
dist.broadcast_object_list(port, src=0)
ok = run_probe_child("gloo", rank, world, int(os.environ.get("LOCAL_RANK", "0")), port[0], 90.0)
flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("PROBE_OK" if flag.item() else "PROBE_FAILED", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if flag.item() else 3)

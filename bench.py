#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native FM-demodulation hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload target|c2|c3|c4|c5]

A "step" is one pass of the hot path (one call of demod_process_device through
the C ABI: the fused streaming kernel, chunk heads included) over one shard of
synthetic uint8 IQ that is already resident in HBM.  Metric: IQ Msamples/s of
input (1 sample = 2 bytes in, 0.5 float32 out), whole job.

Workloads (BASELINE.json / SURVEY.md section 8d):
    target  2^32 samples per GPU, -L 12500 -l 6500   (north_star target; default)
    c3      2^30 samples per GPU, -L 12500 -l 6500   (configs[2])
    c2      2^30 samples per GPU, no -L, default -l  (configs[1])
    c4      2^31 samples per GPU of one sharded stream (configs[3]: 2^34 at 8 GPUs)
    c5      32 channels x 2^24 samples per GPU, batched (configs[4]: 256 channels at 8 GPUs)
  (beside the default line, under "also": c3, c2, c3 with int16 output, the other forms of the complex stencil, and c3 on a
   smooth FM-like signal -- the kernel's clock is power-managed and the prescribed random bytes are its most expensive input)

For N > 1 there is one process per GPU: either the driver launches them with
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment), or -- invoked plainly as `python bench.py --gpus N` -- this
script starts N child processes itself, BEFORE anything touches the GPU, waits
for them and returns the worst exit code (rank 0 prints the line).  The global
stream is cut into N contiguous shards.  By default the cut points are NOT
chunk boundaries (every shard starts 96 KiB into a 256 KiB chunk), so each rank
needs the 64 look-ahead bytes that follow its shard: they are exchanged every
step with RCCL send/recv (torch.distributed P2P, backend "nccl" = RCCL over
xGMI) inside the timed region.  --aligned cuts at chunk boundaries instead (no
exchange is needed then, and none is made).  Barriers, the max over ranks and
the votes run over a gloo control group.

The RCCL point-to-point path is health-checked first in throw-away child
processes (demodulator_amd/sharding.py::run_probe_child): a probe that hangs is
killed from outside and this process never holds a stuck communicator.  The
ranks make their first GPU call only after the probe's processes have gone: never
more than one process of this benchmark per GPU at a time.  If the
probe does not pass on every rank, the run still measures and prints the line
for chunk-aligned cuts (no exchange), but says `"halo_exchange": "failed"` at
top level and EXITS NON-ZERO: a line produced that way has not exercised the
halo mechanism and must not read as a pass.

Before the W warm-up steps every rank spins the device up for --prewarm-ms
(default 300 ms) of the same work, as part of set-up: a fresh process runs its
first 10-20 ms of kernels at lower clocks than it then holds, and a step of the
2^30-sample workloads is under a millisecond.  The timed region is exactly the
K steps.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E (MI355X_MICROARCH.md)
ALGO_BYTES_PER_SAMPLE = 4  # 2 B read + 2 B written

WORKLOADS = {
    "target": dict(log2_samples=32, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D03,
                   desc="2^32-sample synthetic uint8 IQ per GPU, -L 12500 -l 6500 (both lowpasses)"),
    "c3": dict(log2_samples=30, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D03,
               desc="2^30-sample synthetic uint8 IQ per GPU, -L 12500 -l 6500"),
    "c2": dict(log2_samples=30, flags=dict(), seed=0xD3300D02,
               desc="2^30-sample synthetic uint8 IQ per GPU, no -L, default -l"),
    # configs[3]: 2^34 samples over 8 GPUs = 2^31 per GPU (weak scaling keeps that share at every N)
    "c4": dict(log2_samples=31, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D04,
               desc="2^31-sample shard per GPU of one stream (2^34 samples at 8 GPUs), -L 12500 -l 6500"),
    # configs[4]: 256 channels x 2^24 samples over 8 GPUs = 32 channels per GPU.  Channels are
    # independent streams of whole chunks, so a GPU's channels are laid back to back and go
    # through ONE launch (chunks never interact: tests/test_gpu_parity.py::test_batched_channels)
    "c5": dict(log2_samples=29, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D05,
               desc="32 channels x 2^24 samples per GPU (256 channels at 8 GPUs), batched in one launch, -L 12500 -l 6500"),
    # not reference behaviour: the float32 -> int16 hop that follows the reference in its README pipes,
    # fused into the kernel's store (demod_set_output_format): 2 B read + 1 B written per sample
    "c3_s16": dict(log2_samples=30, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D03, out="s16",
                   desc="C3 with the fused int16 output: 2^30 samples, -L 12500 -l 6500, 3 B of HBM traffic per sample"),
    # the other forms of the complex stencil (include/demod_hip.h, demod_set_stencil_form).  The default is the
    # stencil divided by its first tap where build_plan allows it (the BASELINE designs), the plain form otherwise.
    "c3_plain": dict(log2_samples=30, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D03, form="plain",
                     desc="C3 with the plain form of the complex stencil (round 2's default): 2^30 samples, -L 12500 -l 6500"),
    # not the default arithmetic: the unpack's -128 folded into the complex stencil's constant
    # (faster than plain, but outside the parity criterion at samples that filter to next to nothing: DESIGN.md section 5)
    # not the prescribed synthetic input: the same C3 run on a smooth FM-like signal (what an SDR delivers).  The -L kernel's
    # clock is power-managed and random bytes toggle the most bits: this line shows what the prescribed input costs
    "c3_fm": dict(log2_samples=30, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D03, signal="fm",
                  desc="C3 on a smooth FM-like signal instead of uniformly random bytes (same kernel, same sizes): 2^30 samples, -L 12500 -l 6500"),
    "c3_fold": dict(log2_samples=30, flags=dict(lowpass_in=12500.0, lowpass_out=6500.0), seed=0xD3300D03, form="folded",
                    desc="C3 with the optional folded unpack (not the default arithmetic): 2^30 samples, -L 12500 -l 6500"),
}
PHASE_BYTES = 96 * 1024  # unaligned cut: shards start this far into a chunk


def cpu_baseline(sample_log2: int, flags: dict, seed: int, fetch=None) -> dict:
    """The reference CPU path timed on this host: the reference's own functions
    compiled from its sources (oracle/_ref, prebuilt) when present, else the
    restatement.  One compute thread, as the reference runs (src/main.c:67-70).
    `fetch(n_bytes)` supplies the first bytes of the same synthetic stream (the
    device generator's output, bit-identical to oracle.synth_bytes and much
    faster than regenerating 1 GiB in numpy)."""
    import oracle

    n_bytes = 2 << sample_log2
    data = fetch(n_bytes) if fetch is not None else oracle.synth_bytes(seed, 0, n_bytes)
    fl = oracle.Flags(**flags)
    ref = oracle.reference()
    impl, kind = (ref, "reference") if ref is not None else (oracle.restatement(), "port")
    impl.process(fl, data[: 4 * fl.buf_size])  # warm up (page in, design)
    t0 = time.perf_counter()
    out = impl.process(fl, data)
    dt = time.perf_counter() - t0
    assert out.size == n_bytes // 4
    res = {"value": round((n_bytes / 2) / dt / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": kind,
           "sample": f"first 2^{sample_log2} samples of the same synthetic stream, same flags, "
                     f"{n_bytes // fl.buf_size} chunks, {dt:.2f} s of CPU time"}
    # chunk-parallel upper bound on the host's cores (chunks are independent, src/matrix.c:272)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 64)
    orc = oracle.restatement()
    t0 = time.perf_counter()
    orc.process(fl, data, threads=cores)
    dt = time.perf_counter() - t0
    res["all_cores"] = {"value": round((n_bytes / 2) / dt / 1e6, 2), "cores": cores, "kind": "port"}
    return res


def run_workload(dm, torch, dist, name: str, steps: int, warmup: int, rank: int, world: int,
                 aligned: bool, seg_rows: int, prewarm_ms: float = 300.0, data_group=None):
    wl = WORKLOADS[name]
    n_bytes = 2 << wl["log2_samples"]  # per GPU
    opt = dm.Options(**wl["flags"])
    B = opt.buf_size
    dev = torch.device("cuda", torch.cuda.current_device())
    d = dm.Demodulator(opt, device=dev.index)
    if seg_rows:
        d.set_segment_rows(seg_rows)
    if wl.get("form"):
        if hasattr(d, "set_stencil_form"):
            d.set_stencil_form(wl["form"])
        elif wl["form"] == "folded":
            d.set_unpack_fold(True)
    d.enable_timing(True)

    from demodulator_amd.sharding import exchange_halo, plan_shards

    phase = 0 if (aligned or world == 1) else PHASE_BYTES
    shards = plan_shards(n_bytes * world, world, B, phase_bytes=phase)
    me = shards[rank]
    right = shards[me.halo_src] if me.needs_halo else None
    assert me.length == n_bytes
    iq = torch.empty(n_bytes, dtype=torch.uint8, device=dev)
    s16 = wl.get("out") == "s16"
    if s16:
        d.set_output_format("s16", 1.0 / (3.14159265 * float(abs(d.taps("out")[0]).sum())))
    # The output buffer sits where the allocator puts it -- what a user of the library gets.
    # (How the input and output buffers lie relative to each other in the HBM channel/bank
    # interleave moves the kernel time by up to 7 % on some devices of the pool and not at
    # all on others: tools/skew_probe.py, profiles/r1_buffer_placement.txt.)
    out = torch.empty(n_bytes // 4, dtype=torch.int16 if s16 else torch.float32, device=dev)
    if wl.get("signal") == "fm":
        # 127.5 + 100 cos/sin(phase), phase = a carrier plus a slow tone's integral (SURVEY.md section 8d's secondary input), one
        # 2 MiB period repeated
        import numpy as np
        m = 1 << 20
        t = np.arange(m, dtype=np.float64)
        ph = 2 * np.pi * (0.05 * t + 40.0 * np.sin(2 * np.pi * t / 4096.0))
        period = np.empty(2 * m, dtype=np.uint8)
        period[0::2] = np.clip(np.rint(127.5 + 100.0 * np.cos(ph)), 0, 255)
        period[1::2] = np.clip(np.rint(127.5 + 100.0 * np.sin(ph)), 0, 255)
        iq.copy_(torch.from_numpy(period).to(dev).repeat(n_bytes // (2 * m)))
    else:
        d.synth_fill(iq, me.start, wl["seed"])  # this rank's bytes of the global stream
    halo = torch.zeros(dm.HALO_BYTES, dtype=torch.uint8, device=dev)

    def exchange():
        if me.needs_halo or me.sends_halo:
            # RCCL send/recv: my first 64 bytes go to the rank on my left, the
            # first 64 bytes of the rank on my right come to me
            if data_group is not None:
                exchange_halo(me, right, iq, halo, group=data_group)
            else:  # rehearsal: through host memory over the control group
                mine = iq[: dm.HALO_BYTES].cpu()
                theirs = torch.zeros(dm.HALO_BYTES, dtype=torch.uint8)
                exchange_halo(me, right, mine, theirs)
                halo.copy_(theirs)

    def step():
        exchange()
        d.process(iq, out, chunk_offset=me.chunk_offset, halo=halo if me.needs_halo else None,
                  end_of_stream=me.end_of_stream)

    # Device spin-up, part of set-up (not a step, never timed): a freshly started
    # process runs its first ~10-20 ms of kernels well below the clocks it then
    # holds (tools/alloc_probe.py: the first 13 launches 0.61 of peak, every later
    # one 0.70-0.71, same buffers), and one step of the short workloads is < 1 ms.
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < prewarm_ms:
        d.process(iq, out, chunk_offset=me.chunk_offset, halo=halo if me.needs_halo else None,
                  end_of_stream=me.end_of_stream)
        torch.cuda.synchronize()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    d.kernel_ms()  # drop warm-up timings
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    kms, launches = d.kernel_ms()
    dispatches = max(1, d.last_call_launches)
    # what the look-ahead exchange costs a step, measured on its own AFTER the timed region (the K steps above are
    # not instrumented): K exchanges, each waited for -- issue, transport and completion, an upper bound on what a
    # step pays, since there the exchange overlaps the previous step's kernel
    halo_us = None
    if me.needs_halo or me.sends_halo:
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(steps):
            exchange()
            torch.cuda.synchronize()
        halo_us = (time.perf_counter() - th) / steps * 1e6
    checksum = float(out[1000003:: max(1, out.numel() // 4099)].double().abs().sum().item())
    d.close()
    del iq, out
    torch.cuda.empty_cache()
    return dict(dt=dt, kernel_ms=kms, launches=launches, n_bytes=n_bytes, opt=opt, checksum=checksum,
                phase=phase, chunk=B, exchanged=bool(me.needs_halo or me.sends_halo), dispatches=dispatches, halo_us=halo_us)


def _free_port() -> int:
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks of this script as
    child processes (the parent makes no GPU call, before or after), wait for all of
    them and return the worst exit code.  Rank 0 prints the JSON line on the
    inherited stdout.  When a rank dies the others get a grace period, then are
    killed by PID (they would wait in a barrier for ever)."""
    import subprocess

    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    worst = 0
    deadline = None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        codes = [p.poll() for p in procs]
        if deadline is None and any(c not in (None, 0) for c in codes):
            deadline = time.time() + 60.0
        if deadline is not None and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
    for p in procs:
        rc = p.wait()
        worst = max(worst, abs(rc) if rc else 0)
    return worst


def single_process(args) -> int:
    """`--single-process`: the same workload and the same line, with ONE process driving every device through
    demod_multi_* (demodulator_amd/csrc/multi.cpp) -- the C-level multi-GPU entry and its native grouped
    ncclSend/ncclRecv get a number of their own beside torch.distributed's.  A step = exchange + one enqueue per
    device; the K steps are bracketed by a synchronize of every device's stream (there is one process: no barrier)."""
    import torch

    import demodulator_amd as dm

    n = args.gpus
    have = torch.cuda.device_count()
    if have == 0:
        raise SystemExit("bench.py needs an MI355X: the demodulation path has no CPU fallback")
    distinct = have >= n
    devices = list(range(n)) if distinct else [i % have for i in range(n)]   # fewer devices than shards: a rehearsal, said so in the line
    wl = WORKLOADS[args.workload]
    n_bytes = 2 << wl["log2_samples"]
    opt = dm.Options(**wl["flags"])
    phase = 0 if (args.aligned or n == 1) else PHASE_BYTES
    m = dm.MultiDemodulator(opt, devices)
    shards = m.plan(n_bytes * n, phase)
    ins, outs = [], []
    for i, dev in enumerate(devices):
        with torch.cuda.device(dev):
            iq = torch.empty(n_bytes, dtype=torch.uint8, device=f"cuda:{dev}")
            with dm.Demodulator(opt, device=dev) as d:
                d.synth_fill(iq, shards[i].start, wl["seed"])
                torch.cuda.synchronize()
            ins.append(iq)
            outs.append(torch.empty(n_bytes // 4, dtype=torch.float32, device=f"cuda:{dev}"))
    m.enable_timing(True)
    t_spin = time.perf_counter()
    while (time.perf_counter() - t_spin) * 1e3 < args.prewarm_ms:
        m.process(ins, outs, shards)
        m.synchronize()
    for _ in range(args.warmup):
        m.process(ins, outs, shards)
    m.synchronize()
    for i in range(n):
        m.kernel_ms(i)          # drop warm-up timings
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m.process(ins, outs, shards)
    m.synchronize()
    dt = time.perf_counter() - t0
    per_dev = [m.kernel_ms(i) for i in range(n)]
    transport = m.halo_transport
    checksum = float(outs[0][1000003:: max(1, outs[0].numel() // 4099)].double().abs().sum().item())
    m.close()
    samples_per_gpu = n_bytes // 2
    kern = [ms / max(1, disp) for ms, _, disp in per_dev]
    algo = ALGO_BYTES_PER_SAMPLE * samples_per_gpu // max(1, per_dev[0][2])
    achieved = algo / (kern[0] * 1e-3) if kern[0] > 0 else None
    line = {
        "metric": "IQ Msamples/sec (input) end-to-end demod+filter",
        "value": round(samples_per_gpu * n * args.steps / dt / 1e6, 1), "unit": "Msamples/s", "n_gpus": n,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (counter-based splitmix64 uint8 IQ, generated on device)",
        "halo_exchange": {"rccl": "rccl send/recv (grouped, one thread)", "peer-copy": "hipMemcpyPeerAsync (RCCL unavailable or refused the device list)",
                          "none": "none (chunk-aligned cuts)" if n > 1 else "none (1 GPU)"}.get(transport, transport),
        "config": {"workload": f"{args.workload}: {wl['desc']}", "samples_per_gpu": samples_per_gpu, "chunk_bytes": opt.buf_size,
                   "sharding": "1 GPU" if n == 1 else f"{n} contiguous shards, " + (f"cuts {phase} B into a chunk, 64-byte look-ahead per step" if phase else "chunk-aligned cuts, no exchange")},
        "multi": {"mode": "one process, demod_multi_* (C ABI, multi.cpp)", "halo_transport": transport,
                  "devices": devices, "devices_distinct": distinct,
                  "kernel_ms_per_dispatch_min": round(min(kern), 4), "kernel_ms_per_dispatch_max": round(max(kern), 4)},
        "roofline": {"bound": "hbm", "kernel": "demod_rows (fused unpack + complex stencil + discriminator/atan2 + real stencil), device 0",
                     "achieved": round(achieved / 1e9, 1) if achieved else None, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK, 4) if achieved else None, "algorithmic_bytes_per_launch": algo,
                     "kernel_ms": round(kern[0], 4), "launches_per_step": per_dev[0][2], "traffic": None},
        "checksum": checksum,
    }
    if not distinct:
        line["multi"]["note"] = f"REHEARSAL: {have} device(s) for {n} shards -- contexts share devices, not a scaling number"
    print(json.dumps(line))
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="target")
    ap.add_argument("--aligned", action="store_true", help="cut shards at chunk boundaries (no halo exchange)")
    ap.add_argument("--segment-rows", type=int, default=0)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed device spin-up before the warm-up steps (set-up, see run_workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--probe-timeout", type=float, default=240.0,
                    help="seconds the point-to-point health check may take before it is killed (on a fresh box the "
                         "probe's processes page in torch and the 570 MB librccl before they start)")
    ap.add_argument("--single-process", action="store_true",
                    help="drive all --gpus devices from THIS process through the C ABI's demod_multi_* (multi.cpp: one context and "
                         "stream per device, the look-ahead of every cut by grouped ncclSend/ncclRecv from one thread) instead of one "
                         "process per GPU")
    args = ap.parse_args()

    if args.single_process:
        return single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)   # before torch is imported: the parent never touches the GPU

    # stdout carries ONE line, the JSON: whatever libraries print there on the way (gloo announces its
    # connections on std::cout, "[Gloo] Rank 0 is connected to ...") goes to stderr instead -- file
    # descriptor 1 points at stderr from here on, the line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import demodulator_amd as dm

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() == 0:   # (counting devices does not initialise the GPU)
        raise SystemExit("bench.py needs an MI355X: the demodulation path has no CPU fallback")
    # DEMOD_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks
    # (ranks share devices, the 64-byte halos travel through host memory); never for reported numbers
    backend = os.environ.get("DEMOD_BENCH_BACKEND", "nccl")
    data_group = None
    halo_exchange = "none (1 GPU)" if world == 1 else "none (chunk-aligned cuts)"
    probe_ok = False
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # control plane (barriers, the max over ranks, the votes) on gloo;
        # the data path -- the 64-byte look-ahead of every cut -- on RCCL
        dist.init_process_group("gloo", rank=rank, world_size=world)

        def all_agree(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        if not args.aligned:
            # 1. health check in throw-away processes with a rendezvous of their own: a hang
            #    is killed from outside, this process makes no RCCL call before it has passed --
            #    and no GPU call at all: while the probe runs its processes are the only ones of
            #    ours on the GPUs (one per device), and they are gone before the ranks arrive
            from demodulator_amd.sharding import run_probe_child

            port = [_free_port() if rank == 0 else 0]
            dist.broadcast_object_list(port, src=0)
            ok = run_probe_child(backend, rank, world, local_rank, port[0], args.probe_timeout,
                                 force_fail=os.environ.get("DEMOD_BENCH_FORCE_PROBE_FAIL") == "1")
            probe_ok = all_agree(ok)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the demodulation path has no CPU fallback")
    torch.cuda.set_device(local_rank % torch.cuda.device_count())

    if world > 1 and not args.aligned:
        # 2. the communicator of this process; every rank votes on the outcome, so that all
        #    of them cut the stream the same way
        ok = probe_ok
        if ok and backend == "nccl":
            try:
                data_group = dist.new_group(backend="nccl")
                dist.all_reduce(torch.zeros(1, device="cuda"), group=data_group)  # communicator exists
                torch.cuda.synchronize()                                          # before any send/recv
                mine = True
            except Exception as e:  # noqa: BLE001
                sys.stderr.write(f"[rank {rank}] RCCL group: {e!r}\n")
                mine = False
            ok = all_agree(mine)
        if ok:
            halo_exchange = "rccl send/recv" if backend == "nccl" else f"{backend} send/recv (rehearsal)"
        else:
            halo_exchange = "failed"
            data_group = None
            args.aligned = True   # still measure: chunk-aligned cuts need no exchange

    r = run_workload(dm, torch, dist, args.workload, args.steps, args.warmup, rank, world, args.aligned,
                     args.segment_rows, args.prewarm_ms, data_group)
    dt = r["dt"]
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # every rank's own numbers, for the line: step time, kernel time per dispatch, what an exchange costs
        mine = torch.tensor([r["dt"] / args.steps * 1e3, r["kernel_ms"] / max(1, r["dispatches"]),
                             -1.0 if r["halo_us"] is None else r["halo_us"]], dtype=torch.float64)
        every = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[float(v) for v in e] for e in every]
        dt = float(t.item())

    samples_per_gpu = r["n_bytes"] // 2
    total_samples = samples_per_gpu * world * args.steps
    value = total_samples / dt / 1e6
    line = {
        "metric": "IQ Msamples/sec (input) end-to-end demod+filter",
        "value": round(value, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (counter-based splitmix64 uint8 IQ, generated on device)",
        "halo_exchange": halo_exchange,
        "config": {
            "workload": f"{args.workload}: {WORKLOADS[args.workload]['desc']}",
            "samples_per_gpu": samples_per_gpu,
            "chunk_bytes": r["chunk"],
            "sharding": ("1 GPU" if world == 1 else
                         f"{world} contiguous shards, " +
                         ("chunk-aligned cuts, no exchange" +
                          (" (the point-to-point health check FAILED on this node: the halo path was not "
                           "exercised, exit code 3)" if halo_exchange == "failed" else "")
                          if not r["phase"] else
                          f"cuts {r['phase']} B into a chunk, 64-byte look-ahead via "
                          f"{'RCCL' if backend == 'nccl' else backend + ' (rehearsal)'} send/recv per step")),
        },
    }
    if world > 1:
        # what the first real multi-GPU run has to tell: how many ranks the RCCL group really has (0: none was made),
        # how far the ranks' step and kernel times spread, what the 64-byte exchange costs
        steps_ms = [p[0] for p in per_rank]
        kern_ms = [p[1] for p in per_rank]
        halos = [p[2] for p in per_rank if p[2] >= 0]
        line["multi"] = {
            "mode": "one process per GPU (torch.distributed)",
            "rccl_ranks": int(dist.get_world_size(group=data_group)) if data_group is not None else 0,
            "control_ranks": int(dist.get_world_size()),
            "devices_visible_to_rank0": int(torch.cuda.device_count()),
            "ms_per_step_min": round(min(steps_ms), 4), "ms_per_step_max": round(max(steps_ms), 4),
            "kernel_ms_per_dispatch_min": round(min(kern_ms), 4), "kernel_ms_per_dispatch_max": round(max(kern_ms), 4),
            "halo_us_per_step": round(max(halos), 1) if halos else None,
            "halo_us_per_step_note": "exchange alone, waited for, after the timed region (upper bound; max over ranks)" if halos else "no exchange",
        }
    if rank == 0:
        # one step dispatches demod_rows once per 2 GiB piece of input, back to back;
        # the library's HIP events bracket all of them: per-dispatch figures below
        nd = r["dispatches"]
        kms = r["kernel_ms"] / nd
        algo = (3 if WORKLOADS[args.workload].get("out") == "s16" else ALGO_BYTES_PER_SAMPLE) * samples_per_gpu // nd
        achieved = algo / (kms * 1e-3) if kms > 0 else None
        line["roofline"] = {
            "bound": "hbm",
            "kernel": "demod_rows (fused unpack + complex stencil + discriminator/atan2 + real stencil)",
            "achieved": round(achieved / 1e9, 1) if achieved else None,
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK, 4) if achieved else None,
            "read_only_frac": round(achieved / 2 / HBM_PEAK, 4) if achieved else None,
            # MI355X_MICROARCH.md: "HBM3E peak BW 8.0 TB/s spec; 6.29 TB/s measured (float4 copy, 79%)"
            "frac_of_measured_copy_rate": round(achieved / 6.29e12, 4) if achieved else None,
            "algorithmic_bytes_per_launch": algo,
            "kernel_ms": round(kms, 4),
            "launches_per_step": nd,
            "launches_timed": r["launches"] * nd,
            "traffic": None,
        }
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tr = json.load(open(tpath)).get(args.workload)
                if tr:
                    line["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
                    line["roofline"]["traffic_source"] = tr.get("source")
            except Exception:
                pass
        line["checksum"] = r["checksum"]

        if world == 1 and not args.no_extra:
            also = {}
            for name in ("c3", "c2", "c3_s16", "c3_plain", "c3_fold", "c3_fm"):
                if name == args.workload:
                    continue
                bps = 3 if WORKLOADS[name].get("out") == "s16" else ALGO_BYTES_PER_SAMPLE
                x = run_workload(dm, torch, dist, name, max(5, args.steps), 2, 0, 1, True, args.segment_rows,
                                 args.prewarm_ms, None)
                sp = x["n_bytes"] // 2
                steps = max(5, args.steps)
                a = bps * sp / (x["kernel_ms"] * 1e-3)  # per step (1 dispatch at 2^30 samples)
                also[name] = {"workload": WORKLOADS[name]["desc"],
                              "value": round(sp * steps / x["dt"] / 1e6, 1), "unit": "Msamples/s",
                              "kernel_ms": round(x["kernel_ms"], 4), "algorithmic_bytes_per_sample": bps,
                              "roofline_frac": round(a / HBM_PEAK, 4)}
            line["also"] = also

        if world == 1 and not args.no_cpu_baseline:
            wl = WORKLOADS[args.workload]

            def fetch(n_bytes):
                d = dm.Demodulator(dm.Options(**wl["flags"]), device=torch.cuda.current_device())
                buf = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
                d.synth_fill(buf, 0, wl["seed"])
                torch.cuda.synchronize()
                host = buf.cpu().numpy()
                d.close()
                return host

            line["cpu_baseline"] = cpu_baseline(min(29, wl["log2_samples"]), {**wl["flags"]}, wl["seed"], fetch)
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # a line measured without the halo mechanism it was asked to exercise is not a pass
    return 3 if halo_exchange == "failed" else 0


if __name__ == "__main__":
    sys.exit(main())

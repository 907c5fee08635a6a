"""bench.py's contract on a one-GPU box: the single-GPU line, and the N > 1 flow rehearsed
with two ranks sharing the device over gloo (RCCL refuses two ranks on one device; the real
RCCL path only exists on the driver's multi-GPU node)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench(args, **env):
    e = dict(os.environ, **env)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    r = subprocess.run([sys.executable, BENCH] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=e)
    lines = r.stdout.decode().splitlines()
    # the contract is ONE line on stdout (gloo's "[Gloo] Rank 0 is connected to ..." chatter once landed there)
    assert len(lines) <= 1, lines
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr.decode()[-2000:]


def test_single_gpu_line_has_roofline_and_cpu_baseline():
    rc, line, err = _bench(["--workload", "c3", "--steps", "5", "--warmup", "2", "--no-extra"])
    assert rc == 0 and line is not None, err
    assert line["n_gpus"] == 1 and line["unit"] == "Msamples/s" and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["halo_exchange"] == "none (1 GPU)"
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0.2 < rf["frac"] <= 1.0
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3
    assert rf["kernel_ms"] * rf["launches_per_step"] <= line["ms_per_step"] * 1.02
    assert cb["kind"] in ("reference", "port") and cb["cores"] == 1 and cb["value"] > 1


def test_two_ranks_self_launched_over_the_gloo_rehearsal_backend():
    """`python bench.py --gpus 2` with no launcher: the script spawns its ranks itself (before
    any GPU call), the probe passes in child processes, the halo travels every step, rc 0."""
    rc, line, err = _bench(["--gpus", "2", "--workload", "c3", "--steps", "3", "--warmup", "1"],
                           DEMOD_BENCH_BACKEND="gloo")
    assert rc == 0 and line is not None, err
    assert line["n_gpus"] == 2 and line["halo_exchange"].startswith("gloo send/recv")
    assert "cuts 98304 B into a chunk" in line["config"]["sharding"]
    assert line["value"] > 0 and "cpu_baseline" not in line


def test_a_failed_halo_probe_is_not_a_pass():
    """The point-to-point health check fails (forced): the aligned-cut line is still printed,
    carries "halo_exchange": "failed", and the exit code is non-zero."""
    rc, line, err = _bench(["--gpus", "2", "--workload", "c3", "--steps", "2", "--warmup", "1"],
                           DEMOD_BENCH_BACKEND="gloo", DEMOD_BENCH_FORCE_PROBE_FAIL="1")
    assert rc != 0 and line is not None, err
    assert line["halo_exchange"] == "failed" and line["n_gpus"] == 2
    assert "chunk-aligned cuts, no exchange" in line["config"]["sharding"] and "FAILED" in line["config"]["sharding"]


def test_two_ranks_under_torch_distributed_run_as_the_driver_launches_them():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    --master-port P bench.py --gpus 2 ...`: the ranks find RANK / WORLD_SIZE in the environment,
    do not self-launch, and the probe's child processes keep a rendezvous of their own although
    the launcher's environment points at the elastic agent's store."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ, DEMOD_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--workload", "c3", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=e)
    lines = r.stdout.decode().splitlines()
    assert r.returncode == 0 and len(lines) == 1, (lines, r.stderr.decode()[-2000:])   # one line on stdout, nothing else
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["halo_exchange"].startswith("gloo send/recv") and line["value"] > 0

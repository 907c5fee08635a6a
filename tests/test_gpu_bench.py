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
    # what the first real multi-GPU run has to tell: the size of the RCCL group as created (0 in this
    # rehearsal: the halos travel over gloo), every rank's own step and kernel time, the cost of an exchange
    m = line["multi"]
    assert m["rccl_ranks"] == 0 and m["control_ranks"] == 2
    assert 0 < m["ms_per_step_min"] <= m["ms_per_step_max"] and 0 < m["kernel_ms_per_dispatch_min"] <= m["kernel_ms_per_dispatch_max"]
    assert m["halo_us_per_step"] is not None and m["halo_us_per_step"] > 0


def test_single_process_over_the_c_level_multi_gpu_entry():
    """`python bench.py --gpus 2 --single-process`: the same line from ONE process driving both shards through
    demod_multi_* (multi.cpp).  With one device here the two contexts share it (the line says REHEARSAL, RCCL refuses
    the device list and the halos travel by peer copy); on an 8-GPU node the same command times the native grouped
    ncclSend/ncclRecv path."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--single-process", "--workload", "c3", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout.decode()[-500:], r.stderr.decode()[-2000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["roofline"]["frac"] > 0
    assert line["multi"]["mode"].startswith("one process") and line["multi"]["halo_transport"] == "peer-copy"
    assert line["multi"]["devices_distinct"] is False and "REHEARSAL" in line["multi"]["note"]
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--single-process", "--workload", "c2", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    line = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert r.returncode == 0 and line["n_gpus"] == 1 and line["multi"]["devices_distinct"] and line["multi"]["halo_transport"] == "none"

"""The driver's entry points stay runnable: bench.py prints one JSON line with the contract's keys, smoke() passes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _bench(extra_env=None, *args):
    env = dict(os.environ)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2", *args],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_line_has_the_contract_keys():
    d = _bench(None, "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["value"] > 0
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3


def test_bench_sharded_driver_runs_on_one_rank():
    d = _bench({"DBA_BENCH_FORCE_SHARDED": "1"}, "--no-cpu-baseline")
    assert d["value"] > 0


def test_smoke_entry_point():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def _bench_ranks(world, *args):
    """bench.py as the driver launches it for N > 1 (one process per rank, torch.distributed.run), with the exchange step on
    gloo so that the ranks can share this box's GPU(s)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", str(world), "--steps", "4", "--warmup", "2", "--backend", "gloo", "--no-cpu-baseline",
                        *args], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # rank 0 prints ONE line
    return json.loads(lines[-1])


def test_bench_weak_scaling_mode_on_one_gpu():
    """--scaling weak at N = 1: the 64-KF / 64-edge member of the weak-scaling family, edge-normalised value"""
    d = _bench(None, "--no-cpu-baseline", "--no-extras", "--scaling", "weak")
    assert d["scaling"] == "weak" and d["config"]["edges"] == 64 and d["config"]["keyframes"] == 64
    assert abs(d["value"] - d["extra"]["dba_update_per_s"] * 64 / 96) < 1e-2 * d["value"]
    assert abs(d["extra"]["edges_per_s"] - 64 * d["extra"]["dba_update_per_s"]) < 1e-3 * d["extra"]["edges_per_s"]


def test_bench_two_ranks_whole_path_weak_and_strong():
    """two processes (sharing the GPU if the box has one), whole bench path: partition by source frame, front stage,
    exchange (gloo, host-staged), redundant solves, depth all-gather; default mode for N > 1 is weak scaling"""
    d = _bench_ranks(2)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["edges"] == 128 and d["value"] > 0
    assert d["extra"]["edge_throughput_vs_1gpu"] > 0 and d["config"]["exchange"].startswith("gloo")
    d = _bench_ranks(2, "--scaling", "strong")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["edges"] == 96 and d["value"] > 0


def test_bench_reports_the_sharded_driver_overhead():
    d = _bench(None, "--no-cpu-baseline")
    assert "sharded_x1_overhead_us" in d["extra"]

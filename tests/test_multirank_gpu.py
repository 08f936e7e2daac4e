"""Two ranks on ONE MI355X (torchrun, exchange over gloo so both processes may share the device): the process-per-GPU path of
bench.py -- MSMs sharded by key stripes, cross terms and ProtoGalaxy leaves by row stripes, witness uploads of 1 / world, partial
commitments and polynomials all-gathered -- folds the same chain as one rank (`state_digest`)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


@pytest.mark.parametrize("config,k,log_key", [("cyclefold", 14, 18), ("cyclefold", 17, 21)])
def test_bench_two_ranks_one_gpu_same_chain(srs, config, k, log_key):
    common = ["--config", config, "--k", str(k), "--log-key", str(log_key), "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True,
                        timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    one = json.loads([l for l in r1.stdout.strip().splitlines() if l.startswith("{")][-1])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                         "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo"] + common,
                        capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-4000:]
    two = json.loads([l for l in r2.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["config"]["parallelism"] == "msm+leaf-shard2"
    assert one["state_digest"] == two["state_digest"]


def test_bench_threads_same_chain(srs):
    """`bench.py --gpus N --threads` (r06): ONE process, one host thread per device (srs_init_thread), every thread with its own sharded key and
    row-sharded structure, partial commitments / polynomials exchanged in memory -- folds the chain `--gpus 1` folds.  On a one-GPU box the
    threads share device 0 (N = 2 logical ranks); on a multi-GPU node every thread drives its own device (N = min(devices, 8))."""
    import torch
    n = max(2, min(torch.cuda.device_count(), 8))
    while (1 << 7) % n:
        n -= 1
    common = ["--k", "17", "--log-key", "21", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = [sys.executable, os.path.join(ROOT, "bench.py")]
    r1 = subprocess.run(bench + ["--gpus", "1"] + common, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    rn = subprocess.run(bench + ["--gpus", str(n), "--threads"] + common, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert rn.returncode == 0, rn.stdout[-2000:] + rn.stderr[-4000:]
    last = lambda so: json.loads([l for l in so.strip().splitlines() if l.startswith("{")][-1])
    one, many = last(r1.stdout), last(rn.stdout)
    assert many["n_gpus"] == n and many["config"]["parallelism"].startswith(f"msm+leaf-shard{n}-threads")
    assert one["state_digest"] == many["state_digest"]


def test_bench_rccl_all_devices_same_chain(srs):
    """When the box has >= 2 devices (the driver's 8-GPU node): `bench.py --gpus N` with the REAL backend (nccl = RCCL over xGMI), one rank
    per GPU, must fold the chain `--gpus 1` folds -- partial commitments and partial polynomials through RCCL all_gather with N > 1 ranks.
    Skips on a one-GPU box (there the exchange runs over gloo above and through RCCL with one rank in tests/test_chain_gpu.py)."""
    import torch
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("one visible device: RCCL with N > 1 ranks needs a multi-GPU node")
    while (1 << 7) % n:                      # 2^17 rows = 2^7 row stripes: the leaves shard when n divides them
        n -= 1
    common = ["--k", "17", "--log-key", "21", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    env = {k: v for k, v in dict(os.environ, MASTER_ADDR="127.0.0.1").items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    bench = [sys.executable, os.path.join(ROOT, "bench.py")]
    r1 = subprocess.run(bench + ["--gpus", "1"] + common, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    rn = subprocess.run(bench + ["--gpus", str(n), "--dist-backend", "nccl"] + common, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert rn.returncode == 0, rn.stdout[-2000:] + rn.stderr[-4000:]
    last = lambda so: json.loads([l for l in so.strip().splitlines() if l.startswith("{")][-1])
    one, many = last(r1.stdout), last(rn.stdout)
    assert many["n_gpus"] == n and many["config"]["parallelism"] == f"msm+leaf-shard{n}"
    assert one["state_digest"] == many["state_digest"]


@pytest.mark.parametrize("shards", [2, 5])
def test_bench_single_process_multi_device_key_same_chain(srs, shards):
    """`bench.py --gpus N --single-process`: the in-library multi-device path (srs_ck_create_multi -- what a single-process Rust driver
    calls; logical shards beyond the physical devices share them) folds the chain of --gpus 1."""
    common = ["--k", "14", "--log-key", "18", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = [sys.executable, os.path.join(ROOT, "bench.py")]
    r1 = subprocess.run(bench + ["--gpus", "1"] + common, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    rn = subprocess.run(bench + ["--gpus", str(shards), "--single-process"] + common, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert rn.returncode == 0, rn.stdout[-2000:] + rn.stderr[-4000:]
    last = lambda so: json.loads([l for l in so.strip().splitlines() if l.startswith("{")][-1])
    one, many = last(r1.stdout), last(rn.stdout)
    assert many["n_gpus"] == shards and many["config"]["parallelism"].startswith(f"msm-multi{shards}-single-process")
    assert one["state_digest"] == many["state_digest"]

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

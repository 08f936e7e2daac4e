"""CPU, world_size 2 over gloo: the multi-GPU MSM path (sharded keys -> per-rank partial -> all-gather ->
host sum).  The kernels run through the logic emulator (tests/emu) because there is no GPU here;
sharding arithmetic, the exchange and the combination are the product's own code."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SRS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SRS_ROOT"], "tests"))
import torch, torch.distributed as dist
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import sirius_amd._lib as L
L.load(os.path.join(os.environ["SRS_ROOT"], "tests", "emu", "libsirius_emu.so"))
import sirius_amd as S
import oracle as O
from conftest import seeded_scalars
from sirius_amd.distributed import all_gather_commitments
ok = True
for cid, N, n in ((0, 2600, 2600), (1, 3000, 2047)):
    full = S.CommitmentKey.setup_synthetic(cid, N, seed=3).bases()
    ck = S.CommitmentKey.setup_synthetic(cid, N, seed=3, rank=rank, world=world)
    sc = seeded_scalars(O, cid, n, 9, "trace" if cid else "uniform")
    got = all_gather_commitments(cid, ck.commit(sc))
    ok &= bool(np.array_equal(got, O.msm(cid, sc, full[:n])))
    batch = all_gather_commitments(cid, ck.commit_batch([sc[:1500], sc[:7]]))
    ok &= bool(np.array_equal(batch[0], O.msm(cid, sc[:1500], full[:1500])) and np.array_equal(batch[1], O.msm(cid, sc[:7], full[:7])))
t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
dist.destroy_process_group()
sys.exit(0 if int(t.item()) == 1 else 1)
'''


def test_sharded_commit_world2_gloo(tmp_path, oracle):
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-C", emu_dir, "-j4"], stdout=subprocess.DEVNULL)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, SRS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29591", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]

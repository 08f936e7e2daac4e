"""GPU parity of the chain bench.py's headline times: `CyclefoldIVC::next` hot path with the reference's leaf rows
(src/plonk/mod.rs:714) and Poseidon-derived challenges (src/nifs/protogalaxy/mod.rs:80-133,400-481;
src/nifs/sangria/mod.rs:162-179,253-277), several steps, against the same chain recomputed on the CPU oracle
(tests/chain_cases.py).  Bit-exact: the digest covers e, the folded instance commitments of both circuits and the last
witness commitments, and every step's challenges depend on all of the previous step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,log_key,ks,steps", [(10, 14, 8, 2), (12, 16, 9, 3)])
def test_cyclefold_chain_digest_vs_oracle(srs, oracle, k, log_key, ks, steps):
    import chain_cases as CC
    want = CC.oracle_chain(oracle, srs, k, log_key, ks, steps)
    assert CC.product_chain(srs, k, log_key, ks, steps) == want                           # support trace: srs_sangria_prove_incoming
    assert CC.product_chain(srs, k, log_key, ks, steps, split_support=True) == want       # commit, then srs_sangria_prove


@pytest.mark.parametrize("k,log_key,steps", [(10, 14, 2), (12, 16, 3)])
def test_sangria_chain_digest_vs_oracle(srs, oracle, k, log_key, steps):
    """BASELINE configs[1]'s two-curve chain (bench.py's secondary.sangria_k17 leg: both orders of the step -- resident traces, and
    witnesses from the host) against the same chain on the oracle: commitments of both curves, cross-term commitments, the Poseidon-derived
    challenge of every step, the folded instances."""
    import chain_cases as CC
    want = CC.oracle_chain_sangria(oracle, srs, k, log_key, steps)
    assert CC.product_chain_sangria(srs, k, log_key, steps, from_host=False) == want
    assert CC.product_chain_sangria(srs, k, log_key, steps, from_host=True) == want


def test_cyclefold_chain_digest_k20_vs_oracle(srs, oracle):
    """The headline chain AT ITS OWN SIZE (BASELINE configs[2]: k = 20, key 2^24, support circuit k = 15,
    src/ivc/cyclefold/support_circuit/mod.rs:68): one warm-up step + one step -- the 12 * 2^20 streamed commits in slot mode, the
    3 * 2^15 + 2 x 2^15 grumpkin batch of srs_sangria_prove_incoming, F / G / K / e on 2^21 leaves, every Poseidon-derived challenge --
    against the same chain on the oracle (oracle/chain.py, the C legs of compute_F / compute_G: ~15 s of CPU).  bench.py folds exactly
    this chain, so its `state_digest` is tied to an oracle value through this test (and `bench.py --verify` prints both)."""
    import os
    import chain_cases as CC
    threads = min(32, 2 * (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8))
    want = CC.oracle_chain(oracle, srs, 20, 24, 15, 2, fast=True, threads=threads)
    assert CC.product_chain(srs, 20, 24, 15, 2) == want


def test_cyclefold_chain_true_rows_differs(srs, oracle):
    """the intended leaf rows (`--leaf-rows true`) fold a DIFFERENT chain: the switch is not a no-op"""
    import chain_cases as CC
    assert CC.product_chain(srs, 10, 14, 8, 1, compat=False) != CC.product_chain(srs, 10, 14, 8, 1, compat=True)


NCCL_WORLD1 = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SRS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SRS_ROOT"], "tests"))
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
import sirius_amd as S
import oracle as O
from conftest import seeded_scalars
from sirius_amd.distributed import all_gather_commitments, all_gather_field_sum
dev = torch.device("cuda", 0)
for cid in (0, 1):
    ck = S.CommitmentKey.setup_synthetic(cid, 3000, seed=3, rank=0, world=1)
    sc = seeded_scalars(O, cid, 2500, 9, "trace")
    part = ck.commit(sc)
    got = all_gather_commitments(cid, part, device=dev)              # RCCL all_gather of the 64 raw bytes, then the host sum
    assert np.array_equal(got, O.msm(cid, sc, ck.bases()[:2500])), cid
    batch = all_gather_commitments(cid, np.stack([part, np.zeros(8, np.uint64)]), device=dev)
    assert np.array_equal(batch[0], part) and not batch[1].any()
poly = seeded_scalars(O, 0, 33, 5)
assert np.array_equal(all_gather_field_sum(0, poly, device=dev), poly)    # RCCL all_gather of 33 field elements, summed through the library
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 1.5
dist.barrier()
dist.destroy_process_group()
print("nccl-ok")
'''


def test_rccl_world1_collectives(srs, oracle, tmp_path):
    """The process-per-GPU exchange on the REAL backend: torch.distributed `nccl` (= RCCL on ROCm) initialised with one rank on
    the MI355X, partial commitments and partial polynomials sent through the actual all_gather (sirius_amd/distributed.py no
    longer short-circuits a one-rank group), the max-over-ranks all_reduce and the barrier bench.py uses.  Proves the RCCL
    load, the int64-view dtype path and the device placement on hardware; world > 1 is covered by the gloo tests."""
    import os
    import socket
    import subprocess
    import sys
    from conftest import ROOT
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = str(so.getsockname()[1])
    script = tmp_path / "nccl_world1.py"
    script.write_text(NCCL_WORLD1)
    env = dict(os.environ, SRS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "nccl-ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_bench_gpus_flag_fails_loudly_without_devices():
    """`python bench.py --gpus 8` on a one-GPU box must refuse (one rank per GPU), not silently run one process."""
    import os
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    want = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


def test_device_sponge_equals_host_sponge(srs):
    """srs_poseidon_squeeze_device (one wavefront on the MI355X) == srs_poseidon_squeeze (host), which tests/test_poseidon.py pins
    to the reference's known answer through the oracle"""
    from test_emu_logic import POSEIDON_DEVICE_CODE
    exec(compile(POSEIDON_DEVICE_CODE.replace("import sirius_amd as S\n", ""), "<device sponge>", "exec"), {"S": srs})


def test_device_sponge_vs_oracle(srs, oracle):
    """srs_poseidon_squeeze_device against oracle/poseidon.py DIRECTLY (not through the host sponge): both fields, the reference's
    parameter set (T = 5, RATE = 4, R_F = R_P = 10) and two others, buffers around the rate, 128- and 253-bit squeezes."""
    import random
    from oracle import poseidon as OP
    from oracle import pyref as P
    from sirius_amd.field import MODULUS, ints_to_mont
    O = oracle
    for field in (0, 1):
        for (t, rf, rp) in ((5, 10, 10), (3, 4, 3)):
            for n in (0, 1, 4, 5, 29):
                vals = [random.Random(n * 11 + i).randrange(MODULUS[field]) for i in range(n)]
                h = srs.PoseidonHash(field, t, t - 1, rf, rp)
                if n:
                    h.absorb_field(ints_to_mont(field, vals))
                for bits, of in ((128, field), (253, 1 - field)):
                    oh = OP.PoseidonHash(P.MODULI[field], t, t - 1, rf, rp)
                    oh.absorb_field_iter(vals)
                    got = O.mont_to_ints(of, np.asarray(h.squeeze_device(bits, of)[0]).reshape(1, 4))
                    assert got == [oh.squeeze(bits) % P.MODULI[of]], (field, t, n, bits)

"""Randomised parity soak on the GPU (not part of the test suite): random lengths / batches / value mixtures for the MSM,
random sizes for the NTT, random gate sets for the cross terms -- everything against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
import sirius_amd as S
from oracle import expr as OE
from conftest import seeded_scalars
from workloads import gates_for, rand_fe

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
t0 = time.time()
counts = dict(msm=0, ntt=0, cross=0)
keys = {}
while time.time() - t0 < budget:
    what = rng.integers(0, 3)
    if what == 0:
        cid = int(rng.integers(0, 2))
        if cid not in keys:
            bases = O.make_bases(cid, 11 + cid, 150000)
            bases[rng.integers(0, 150000, size=20)] = 0                      # a few identity bases
            keys[cid] = (bases, S.CommitmentKey(cid, bases))
        bases, ck = keys[cid]
        batch = int(rng.integers(1, 5))
        vs = []
        for _ in range(batch):
            n = int(rng.choice([1, 2, 17, 1000, 1023, 1025, 4097, 65537, int(rng.integers(1, 150000))]))
            kind = ["uniform", "trace"][int(rng.integers(0, 2))]
            v = seeded_scalars(O, cid, n, int(rng.integers(0, 1 << 30)), kind)
            if rng.random() < 0.2:
                v[:] = v[0]                                                  # all-equal scalars
            vs.append(v)
        for g, v in zip(ck.commit_batch(vs), vs):
            assert np.array_equal(g, O.msm(cid, v, bases[: len(v)])), ("msm", cid, [len(x) for x in vs])
        counts["msm"] += batch
    elif what == 1:
        k = int(rng.integers(0, 19))
        a = rand_fe(rng, 1 << k)
        fn = ["fft", "ifft", "coset_fft", "coset_ifft"][int(rng.integers(0, 4))]
        assert np.array_equal(getattr(S.fft, fn)(a.copy()), getattr(O, fn)(a)), ("ntt", k, fn)
        counts["ntt"] += 1
    else:
        gate_T = [int(x) for x in rng.choice([2, 3, 5], size=int(rng.integers(1, 4)))]
        field, k = int(rng.integers(0, 2)), int(rng.integers(1, 12))
        rows = 1 << k
        gates, nfix, nadv = gates_for(gate_T)
        og, fo, ao = [], 0, 0
        for T in gate_T:
            og.append(OE.main_gate_expression(T, 0, fo, ao, nfix)); fo += 2 * T + 5; ao += T + 2
        fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
        W1, W2 = rand_fe(rng, nadv * rows, 0.3), rand_fe(rng, nadv * rows)
        St = S.PlonkStructure(field, k, [], fixed, nadv, gates)
        nch = St.num_challenges
        u1c, u1u, u2c = rand_fe(rng, nch), rand_fe(rng, 1)[0], rand_fe(rng, nch)
        terms, _ = S.VanillaFS.commit_cross_terms(None, St, u1c, u1u, W1, u2c, W2)
        ch = S.VanillaFS.cross_term_challenges(u1c, u1u, u2c, field)
        _, exp = OE.cross_terms_oracle(O, field, og, 0, nfix, nadv, [], fixed, W1, W2, ch)
        assert all(np.array_equal(a, b) for a, b in zip(terms, exp)), ("cross", gate_T, field, k)
        St.close()
        counts["cross"] += 1
print("soak OK", counts, "seed", seed)

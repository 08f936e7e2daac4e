"""Run-time compiled row program vs the LDS interpreter on a gate set without an ahead-of-time kernel (k = 17)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S
from workloads import gates_for, rand_fe

def run(gate_T, field, k=17):
    rows = 1 << k
    gates, nfix, nadv = gates_for(gate_T)
    rng = np.random.default_rng(1)
    fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    W1, W2 = dev(rand_fe(rng, nadv * rows)), dev(rand_fe(rng, nadv * rows))
    out = {}
    for tag, env in (("jit", None), ("interpreter", "1")):
        if env: os.environ["SRS_NO_JIT"] = env
        t0 = time.perf_counter()
        St = S.PlonkStructure(field, k, [], fixed, nadv, gates)
        t_create = time.perf_counter() - t0
        os.environ.pop("SRS_NO_JIT", None)
        nch = St.num_challenges
        u1c, u1u, u2c = rand_fe(rng, nch), rand_fe(rng, 1)[0], rand_fe(rng, nch)
        f = lambda: S.VanillaFS.commit_cross_terms(None, St, u1c, u1u, W1, u2c, W2)
        f(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(5): f()
        torch.cuda.synchronize()
        out[tag] = dict(create_s=round(t_create, 2), cross_terms_ms=round((time.perf_counter() - t) / 5 * 1e3, 3), d=St.num_cross_terms)
        St.close()
    print(gate_T, "field", field, out, flush=True)

run([3, 2], 1)
run([2, 5, 2], 0)

"""BASELINE config 3 shapes (cyclefold_poseidon, k = 20, benches/cyclefold_poseidon.rs:27-34,116-126): wall time of the
hot-path pieces of one `CyclefoldIVC::next` on device-resident synthetic data:
  ProtoGalaxy::prove  : compute_F (n = 2^21 leaves, 32 points), compute_G (8 points, folded witness fused),
                        compute_K_from_G (256 points), fold_witness (12 * 2^20)
  witness commit      : MSM of 12 * 2^20 scalars on bn256 (run_sps_protocol_1)
Both leaf modes are timed: reference_compat (every leaf at row 0, the reference's behaviour) and true rows."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import sirius_amd as S  # noqa: E402
from sirius_amd import protogalaxy as PG  # noqa: E402
from sirius_amd.field import FR, ints_to_mont  # noqa: E402
from workloads import make_structure_inputs  # noqa: E402


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=20)
    args = ap.parse_args()
    k = args.k
    w = make_structure_inputs("primary", k, seed=7)
    St = S.PlonkStructure(0, k, [], w["fixed"], w["num_advice"], w["gates"])
    ctx = PG.PolyContext(St, 1)
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    W0, W1 = dev(w["W1"]), dev(w["W2"])
    rnd = random.Random(3)
    m = lambda v: ints_to_mont(0, list(v))
    betas = m([rnd.randrange(FR) for _ in range(ctx.betas_count)])
    delta, fa = m([rnd.randrange(FR)])[0], m([rnd.randrange(FR)])[0]
    out = {"k": k, "n_leaves": ctx.count_of_evaluation_with_padding, "points_F": ctx.fft_points_count_F,
           "points_G": ctx.fft_points_count_G, "K_domain": 1 << ctx.fft_log_domain_size_K}
    for compat in (True, False):
        tag = "compat" if compat else "true_rows"
        out[f"compute_F_ms_{tag}"] = round(timeit(lambda: PG.compute_F(ctx, betas, delta, W0, reference_compat=compat)), 3)
        out[f"compute_G_ms_{tag}"] = round(timeit(lambda: PG.compute_G(ctx, betas, [W0, W1], reference_compat=compat)), 3)
        out[f"evaluate_e_ms_{tag}"] = round(timeit(lambda: PG.evaluate_e_from_trace(ctx, betas, W0, reference_compat=compat)), 3)
    pG = PG.compute_G(ctx, betas, [W0, W1])
    out["compute_K_from_G_ms"] = round(timeit(lambda: PG.compute_K_from_G(ctx, pG, fa)), 3)
    L = PG.eval_lagrange_poly_for_cyclic_group(delta, ctx.lagrange_domain)
    out["fold_witness_ms"] = round(timeit(lambda: PG.fold_witness(0, [W0, W1], L)), 3)
    ck = S.CommitmentKey.setup_synthetic(S.CURVE_BN256, w["num_advice"] << k, seed=3)
    out["witness_commit_ms"] = round(timeit(lambda: ck.commit(W1)), 3)
    out["witness_commit_scalars"] = w["num_advice"] << k
    out["sum_prove_plus_commit_ms_compat"] = round(out["compute_F_ms_compat"] + out["compute_G_ms_compat"] + out["compute_K_from_G_ms"] +
                                                  out["fold_witness_ms"] + out["witness_commit_ms"], 3)
    out["sum_prove_plus_commit_ms_true_rows"] = round(out["compute_F_ms_true_rows"] + out["compute_G_ms_true_rows"] +
                                                     out["compute_K_from_G_ms"] + out["fold_witness_ms"] + out["witness_commit_ms"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

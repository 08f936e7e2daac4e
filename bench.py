#!/usr/bin/env python
"""bench.py -- IVC fold-steps/s of the Sangria prover hot path on synthetic Poseidon-shaped traces.

One "step" = the hot-path work of one `SangriaIVC::fold_step`
(reference src/ivc/sangria/incrementally_verifiable_computation.rs:429-635) at the
`sangria_poseidon` bench shapes (benches/sangria_poseidon.rs:28-36: k = 17, key 2^21 per curve):
  A. VanillaFS::prove on the secondary (grumpkin) circuit : 5 cross terms x 2^17 rows evaluated,
     committed (batched MSM), instance fold (host scalar-muls), witness + error-vector fold
  B. primary witness commit   : MSM of 12 * 2^17 scalars on bn256
  C. VanillaFS::prove on the primary (bn256) circuit      : 6 cross terms x 2^17, commit, folds
  D. secondary witness commit : MSM of 7 * 2^17 scalars on grumpkin
Everything the reference keeps on the CPU by construction (halo2 witness synthesis, Poseidon
random oracle, circuit bookkeeping) is outside the path and outside the step; the Fiat-Shamir
challenge r is a seeded constant.  Inputs (witnesses, fixed columns, expanded keys) are resident
in HBM before the timed region; the accumulator produced by step i is the input of step i+1.

Multi-GPU (torchrun, one rank per GPU): every MSM is sharded block-cyclically over the ranks
(rank-local window tables), partial commitments (64 B) are all-gathered over RCCL and summed on
the host; cross-term evaluation and the folds are replicated.  One IVC chain is inherently
sequential, so this is STRONG scaling of a single fold step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (MSM bucket accumulation):
achieved = 96 B x scalars per launch / measured launch time (HIP events on the launch stream).
`cpu_baseline` times the CPU oracle (oracle/, a port of the reference algorithms -- the Rust
reference itself cannot be built here) on the same workload, rank 0 / N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


MADD_PEAK_G = 10.6
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
MSM_BYTES_PER_SCALAR = 96.0    # 64 B base + 32 B scalar, each read once


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--k", type=int, default=17, help="log2 rows (BASELINE configs[1]: 17)")
    ap.add_argument("--log-key", type=int, default=21, help="log2 commitment-key length per curve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ro-challenge", action="store_true",
                    help="derive the folding challenge r of every prove from the off-circuit Poseidon oracle over the commitments "
                         "(T=5, RATE=4, R_F=R_P=10 as benches/sangria_poseidon.rs:80-116) instead of a seeded constant")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (CPU exchange; lets 2 ranks share one GPU in tests)")
    return ap.parse_args()


class Side:
    """One circuit of the cycle: structure + key + accumulator + incoming instance, all in HBM."""

    def __init__(self, which, k, log_key, rank, world, dev):
        import torch
        import sirius_amd as S
        from workloads import make_structure_inputs
        w = make_structure_inputs(which, k, seed=0x5349524955530000 + (2 if which == "primary" else 3))
        self.w = w
        self.field, self.curve, self.rows = w["field"], w["curve"], w["rows"]
        self.S = S.PlonkStructure(self.field, k, [], w["fixed"], w["num_advice"], w["gates"])
        if world > 1:      # cross terms only on the rows of this rank's key stripes (the rows its partial MSMs read)
            self.S.set_shard(rank, world)
        assert (w["num_advice"] * self.rows) <= (1 << log_key)
        self.ck = S.CommitmentKey.setup_synthetic(self.curve, 1 << log_key, seed=42 + self.curve, rank=rank, world=world)
        up = lambda a: torch.from_numpy(a.view(np.int64)).to(dev)
        self.accW, self.accE = up(w["W1"]), up(w["E"])          # running accumulator (RelaxedPlonkWitness)
        self.inW = up(w["W2"])                                   # incoming witness (PlonkWitness.W[0])
        self.u1c, self.u1u, self.u2c, self.r = w["u1_challenges"], w["u1_u"], w["u2_challenges"], w["r"]
        self.d = self.S.num_cross_terms
        # r^1..r^d for the E-commitment fold (accumulator.rs:240-244); host big-int, once
        from sirius_amd.field import MODULUS, from_mont, ints_to_mont
        sf = 0 if self.curve == 0 else 1
        rv = from_mont(sf, self.r)
        self.rpows = ints_to_mont(sf, [pow(rv, i + 1, MODULUS[sf]) for i in range(self.d)])
        self.accCW = np.zeros(8, dtype=np.uint64)
        self.accCE = np.zeros(8, dtype=np.uint64)
        self.pending = None      # instance fold of the previous step, still running on the host workers
        # random oracle over the base field of this side's curve (its points are what gets absorbed)
        self.ro = S.PoseidonHash(1 if self.curve == 0 else 0, 5, 4, 10, 10)


COUNT_NONZERO = False
RO_CHALLENGE = False


def combine(S, side, partial, dist, world, dev):
    """all-gather the per-rank partial commitments (RCCL) and sum them on the host."""
    if world == 1:
        return partial
    from sirius_amd.distributed import all_gather_commitments
    return all_gather_commitments(side.curve, partial, device=dev if dist.get_backend() == "nccl" else None)


def settle(side):
    """Joins the instance fold of the previous step (its result is this step's accumulator instance)."""
    if side.pending is not None:
        side.accCE, side.accCW = side.pending[0].wait(), side.pending[1].wait()
        side.pending = None


def prove(S, side, dist, world, dev):
    """VanillaFS::prove hot path (src/nifs/sangria/mod.rs:253-277)."""
    settle(side)
    terms, commits = S.VanillaFS.commit_cross_terms(side.ck, side.S, side.u1c, side.u1u, side.accW, side.u2c, side.inW)
    if COUNT_NONZERO:     # untimed warm-up only: mixed additions issued = 16 windows x non-zero scalars (ALU roofline)
        side.nz_terms = sum(int((t != 0).any(dim=1).sum().item()) for t in terms)
    commits = combine(S, side, commits, dist, world, dev)
    if RO_CHALLENGE:      # VanillaFS::generate_challenge (src/nifs/sangria/mod.rs:162-179): absorb U1, U2, the cross-term commits
        from sirius_amd.field import MODULUS, from_mont, ints_to_mont
        ro = side.ro.reset()
        for pt in (side.accCW, side.accCE, side.inC):
            ro.absorb_point(side.curve, pt)
        for pt in commits:
            ro.absorb_point(side.curve, pt)
        sf = 0 if side.curve == 0 else 1
        side.r = ro.squeeze(128, sf)
        rv = from_mont(sf, side.r)
        side.rpows = ints_to_mont(sf, [pow(rv, i + 1, MODULUS[sf]) for i in range(side.d)])
    # generate_challenge: Poseidon RO on the CPU in the reference -> seeded constant r here.
    # Both folds depend only on r: the instance fold (host scalar-muls, accumulator.rs:201-264:
    # W' = W1 + r*W2 ; E' = E + sum r^k T_k) runs on the host while the GPU folds the witness.
    # The witness fold is stream-ordered (device-resident operands): both kernels are enqueued first; the instance fold (d + 1
    # host scalar multiplications) is handed to the library's host workers and joined when this side's accumulator instance is
    # next needed (settle) -- nothing on the device waits for it, so the next commitment is enqueued meanwhile.
    acc = S.RelaxedPlonkWitness(side.field, [side.accW], side.accE).fold([side.inW], terms, side.r)
    side.accW, side.accE = acc.W[0], acc.E
    side.pending = (S.point_lincomb_async(side.curve, side.accCE, commits, side.rpows),
                    S.point_lincomb_async(side.curve, side.accCW, side.inC.reshape(1, 8), side.r.reshape(1, 4)))
    return commits


def witness_commit(S, side, dist, world, dev):
    """run_sps_protocol_0: ck.commit(W1)  (src/plonk/mod.rs:441-447)."""
    side.inC = combine(S, side, side.ck.commit(side.inW), dist, world, dev)
    return side.inC


def fold_step(S, pri, sec, dist, world, dev):
    prove(S, sec, dist, world, dev)          # A
    witness_commit(S, pri, dist, world, dev)  # B
    prove(S, pri, dist, world, dev)          # C
    witness_commit(S, sec, dist, world, dev)  # D


def host_cpus():
    """CPUs this process may actually use: min(affinity, cgroup v2 quota).  On the GPU box os.cpu_count() is 256 but the
    container's cpu.max is 16 CPUs -- 256 OpenMP threads there are throttled to a crawl (tools/cpu_probe.py)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except Exception:
        pass
    return n


def cpu_baseline(args, pri, sec):
    """The CPU oracle (port of the reference algorithms) on ONE fold step of the same workload."""
    import oracle as O
    from oracle import expr as OE
    quota = host_cpus()
    threads = args.cpu_threads or min(2 * quota, os.cpu_count() or 1)     # 2 threads per granted CPU measured fastest
    t_all = 0.0
    for side in (sec, pri):
        w = side.w
        gate_T = [5, 3] if side is pri else [5]
        nfix = w["num_fixed"]
        og, fo, ao = [], 0, 0
        for T in gate_T:
            og.append(OE.main_gate_expression(T, 0, fo, ao, nfix)); fo += 2 * T + 5; ao += T + 2
        bases = side.ck.bases() if side.ck.world == 1 else None
        if bases is None:
            return None
        ch = np.concatenate([w["u1_challenges"].reshape(-1, 4), w["u1_u"].reshape(1, 4), w["u2_challenges"].reshape(-1, 4),
                             O.ints_to_mont(side.field, [1])])
        t0 = time.perf_counter()
        cg, T = OE.cross_terms_oracle(O, side.field, og, 0, nfix, w["num_advice"], [], w["fixed"], w["W1"], w["W2"], ch, threads)
        for t in T:                                               # commit_cross_terms: sequential commits
            O.msm(side.curve, t, bases, threads)
        O.fold_w(side.field, w["W1"], w["W2"], w["r"], threads)
        O.fold_e(side.field, w["E"], T, w["r"], threads)
        O.msm(side.curve, w["W2"], bases, threads)               # witness commit
        t_all += time.perf_counter() - t0
    return dict(value=1.0 / t_all, unit="fold-steps/s", cores=threads, kind="port",
                sample=f"1 fold step, k={args.k} (same synthetic workload; oracle/ = C port of best_multiexp + "
                       f"GroupedPoly/GraphEvaluator interpreter, OpenMP {threads} threads; host grants {quota} CPUs "
                       f"of {os.cpu_count()} via cgroup cpu.max)")


def main():
    args = parse()
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    import sirius_amd as S

    pri = Side("primary", args.k, args.log_key, rank, world, dev)
    sec = Side("secondary", args.k, args.log_key, rank, world, dev)
    for side in (pri, sec):
        side.inC = np.zeros(8, dtype=np.uint64)
    witness_commit(S, pri, dist, world, dev)
    witness_commit(S, sec, dist, world, dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    global COUNT_NONZERO, RO_CHALLENGE
    RO_CHALLENGE = bool(args.ro_challenge)
    COUNT_NONZERO = True
    fold_step(S, pri, sec, dist, world, dev)          # extra untimed step that also counts non-zero scalars
    COUNT_NONZERO = False
    for _ in range(args.warmup):
        fold_step(S, pri, sec, dist, world, dev)
    S.profile_enable(True)
    S.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fold_step(S, pri, sec, dist, world, dev)
    settle(pri); settle(sec)     # the last step's instance folds belong to the timed region
    barrier()
    dt = time.perf_counter() - t0
    S.profile_enable(False)
    red_dev = dev if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- N > 1 only: the same step as N independent replicas (one IVC chain per GPU, unsharded keys, no
    # collective).  A single chain is sequential by construction, so this is the number that
    # scales; it is reported as an extra field, `value` stays the sharded single-chain rate.
    replicas = None
    if world > 1:
        rp = Side("primary", args.k, args.log_key, 0, 1, dev)
        rs = Side("secondary", args.k, args.log_key, 0, 1, dev)
        for side in (rp, rs):
            side.inC = np.zeros(8, dtype=np.uint64)
        witness_commit(S, rp, None, 1, dev)
        witness_commit(S, rs, None, 1, dev)
        fold_step(S, rp, rs, None, 1, dev)
        rsteps = max(3, args.steps // 2)
        barrier()
        t1 = time.perf_counter()
        for _ in range(rsteps):
            fold_step(S, rp, rs, None, 1, dev)
        settle(rp); settle(rs)
        barrier()
        rdt = time.perf_counter() - t1
        tt = torch.tensor([rdt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        replicas = world * rsteps / float(tt.item())

    if rank == 0:
        acc0 = S.profile_get("msm_accum0") or dict(total_ms=0.0, launches=0, units=0)
        roof = None
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_accum0.json")))
            if acc0["launches"] and world == 1:
                traffic = round(pmc["hbm_bytes_per_scalar"] * acc0["units"] / acc0["launches"])
        except Exception:
            traffic = None
        if acc0["launches"]:
            achieved = MSM_BYTES_PER_SCALAR * acc0["units"] / (acc0["total_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "msm::k_accum0 (bucket accumulation)", "achieved": round(achieved, 3),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                    "traffic_source": "profiles/r01_pmc_accum0.json (PMC bytes/scalar x scalars per launch)" if traffic else None,
                    "avg_launch_ms": round(acc0["total_ms"] / acc0["launches"], 4), "launches": acc0["launches"],
                    "note": "integer-ALU bound (256-bit modmul), not HBM bound: see DESIGN.md and the `alu` object"}
            # the bound that actually applies: mixed additions (8M+2S in XYZZ) per second against the rate the same
            # instruction mix reaches in isolation on this chip (tools/ubench.hip, profiles/r01_ubench_gfx950_v2_fips_mul.txt)
            nz = sum(int((sd.inW != 0).any(dim=1).sum().item()) + getattr(sd, "nz_terms", 0) for sd in (pri, sec))
            madds = 16.0 * nz * args.steps / max(world, 1)
            roof["alu"] = {"unit": "G mixed-add/s", "achieved": round(madds / (acc0["total_ms"] * 1e-3) / 1e9, 3),
                           "peak": MADD_PEAK_G, "frac": round(madds / (acc0["total_ms"] * 1e-3) / 1e9 / MADD_PEAK_G, 4),
                           "peak_source": "measured: profiles/r01_ubench_gfx950_v2_fips_mul.txt (madd 10.6 G/s)"}
        ct = S.profile_get("rowprog_cross_terms")
        scalars_per_step = (12 + 7 + 6 + 5) * (1 << args.k)
        out = {
            "metric": "IVC fold-steps/s (Sangria, Poseidon-shaped synthetic trace, 2^k rows)",
            "value": round(args.steps / dt, 4), "unit": "fold-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u256 (8 x u32 Montgomery limbs, modular)",
            "data": "synthetic",
            "config": {"workload": f"sangria_poseidon fold_step hot path, k={args.k}, bn256/grumpkin, key 2^{args.log_key}",
                       "primary": "12 advice / 26 fixed / 2 gates / 6 cross terms", "secondary": "7 advice / 15 fixed / 1 gate / 5 cross terms",
                       "parallelism": f"msm-shard{world}" if world > 1 else "single-gpu"},
            "msm_scalars_per_s": round(scalars_per_step * args.steps / dt, 1),
            "roofline": roof,
            "replicas_fold_steps_per_s": None if replicas is None else round(replicas, 3),
            "cross_terms_ms_per_launch": round(ct["total_ms"] / ct["launches"], 4) if ct and ct["launches"] else None,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, pri, sec)
            except Exception as e:   # the baseline is a reported number, never the product path
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

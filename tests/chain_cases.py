"""The CycleFold chain bench.py times, run through bench.py's own classes, next to the oracle's restatement (oracle/chain.py).

bench.py's headline folds a chain with the reference's leaf rows (`index & 2^k`, src/plonk/mod.rs:714) and every challenge
squeezed from the off-circuit Poseidon oracle over the transcript (protogalaxy/mod.rs:80-133, :400-481; sangria/mod.rs:162-179,
:253-277).  `oracle_chain` restates the same steps with the oracle's literal ProtoGalaxy / Sangria / best_multiexp / Poseidon
restatements on the same synthetic inputs and returns the same `state_digest`; `product_chain` runs bench.py's own classes.
Sizes: k <= 12 or so (the literal ProtoGalaxy restatement is pure Python over the leaves)."""
import argparse
import hashlib
import random

import numpy as np


def product_chain(S, k, log_key, ks, steps, emu=False, compat=True, ro=True, split_support=False):
    """bench.py's chain objects, `steps` CycleFold steps -> state_digest (hex).  split_support: the support trace committed and
    then folded by two calls (the reference's call order) instead of srs_sangria_prove_incoming's single batched MSM."""
    import bench
    bench.SPLIT_SUPPORT = bool(split_support)
    D = bench.Dist(argparse.Namespace(emu=emu, gpus=1, dist_backend="nccl"))
    pri, sup, _ = bench.build_cyclefold(S, D, k, log_key, compat, ks)
    for _ in range(steps):
        bench.cyclefold_step(S, D, pri, sup, ro)
    return bench.chain_digest(pri, sup)


from oracle.chain import oracle_chain, oracle_chain_sangria  # noqa: E402,F401  (moved: bench.py --verify uses them too)


def product_chain_sangria(S, k, log_key, steps, from_host=False, emu=False):
    """bench.py's Sangria sides (BASELINE configs[1] shapes), `steps` fold steps with Poseidon-derived r -> bench.sangria_chain_digest"""
    import bench
    from workloads import make_structure_inputs
    bench.SPLIT_SUPPORT = False
    D = bench.Dist(argparse.Namespace(emu=emu, gpus=1, dist_backend="nccl"))
    pri = bench.SangriaSide(S, D, make_structure_inputs("primary", k, seed=0x5349524955530000 + 2), log_key, "primary")
    sec = bench.SangriaSide(S, D, make_structure_inputs("secondary", k, seed=0x5349524955530000 + 3), log_key, "secondary")
    pri.witness_commit(S, D, False)
    sec.witness_commit(S, D, False)
    for _ in range(steps):
        bench.sangria_step(S, D, pri, sec, from_host, True)
    return bench.sangria_chain_digest(pri, sec)

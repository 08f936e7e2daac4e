"""CPU: the run-time compilation path for ARBITRARY circuits (structures without an ahead-of-time kernel), end to end on the logic
emulator: the library plans and emits the sweep-form translation unit exactly as it does for hiprtc (plan_sweep, emit_sweep_source:
clusters, levels, affine clusters, hoisted factors), tests/emu/jit_emu.cpp compiles that text with g++ and loads it, the cross terms
come out of the loaded kernel -- and must equal the oracle's GroupedPoly + GraphEvaluator restatement and the interpreter, bit for
bit.  Includes randomly generated gate expressions (sums, products, negations, scalings, constants, challenges, rotations)."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libsirius_emu.so")


@pytest.fixture(scope="module")
def emu_jit():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    import sirius_amd as S
    from sirius_amd import _lib
    _lib.load(EMU_LIB)
    old = os.environ.get("SRS_EMU_JIT")
    os.environ["SRS_EMU_JIT"] = "1"          # tests/emu/jit_emu.cpp: compile the emitted source with g++ (the emulator's stand-in for hiprtc)
    with S.tuning(jit_always=1):              # run-time compilation below k = 14 as well
        yield S
    if old is None:
        os.environ.pop("SRS_EMU_JIT", None)
    else:
        os.environ["SRS_EMU_JIT"] = old
    _lib._lib = None


def _kind(St):
    from sirius_amd import _lib
    sid, fp = C.c_int(), C.c_uint64()
    _lib.lib().srs_structure_program_source(St._h, 0, None, 0, C.byref(fp), C.byref(sid))
    return sid.value


def _check(S, O, field, k, gates, nfix, nadv, seed, nsel=0):
    """cross terms of `gates` (tuples shared by sirius_amd.expression and oracle.expr) through the compiled kernel, the
    interpreter and the oracle"""
    from oracle import expr as OE
    from workloads import rand_fe
    rows = 1 << k
    rng = np.random.default_rng(seed)
    fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
    sels = [(rng.random(rows) < 0.7).astype(np.uint8) for _ in range(nsel)]
    W1, W2 = rand_fe(rng, nadv * rows, 0.3), rand_fe(rng, nadv * rows)
    St = S.PlonkStructure(field, k, sels, fixed, nadv, gates)
    assert _kind(St) == -2, "expected the run-time compiled kernel"
    with S.tuning(no_jit=1):
        Si = S.PlonkStructure(field, k, sels, fixed, nadv, gates)
    assert _kind(Si) == -1
    nch = St.num_challenges
    u1c, u1u, u2c = rand_fe(rng, nch), rand_fe(rng, 1)[0], rand_fe(rng, nch)
    tj, _ = S.VanillaFS.commit_cross_terms(None, St, u1c, u1u, W1, u2c, W2)
    ti, _ = S.VanillaFS.commit_cross_terms(None, Si, u1c, u1u, W1, u2c, W2)
    ch = S.VanillaFS.cross_term_challenges(u1c, u1u, u2c, field)
    cg, exp = OE.cross_terms_oracle(O, field, list(gates), nsel, nfix, nadv, sels, fixed, W1, W2, ch)
    assert len(tj) == len(ti) == len(exp)
    for a, b, c in zip(tj, ti, exp):
        assert np.array_equal(a, c), "compiled kernel vs oracle"
        assert np.array_equal(b, c), "interpreter vs oracle"
    St.close()
    Si.close()


@pytest.mark.parametrize("field,gate_T", [(1, [3, 2]), (0, [2, 5, 2])])
def test_emu_jit_main_gates(emu_jit, oracle, field, gate_T):
    from workloads import gates_for
    gates, nfix, nadv = gates_for(gate_T)
    _check(emu_jit, oracle, field, 5, gates, nfix, nadv, seed=len(gate_T) + field)


def _random_expr(rnd, nsel, nfix, nadv, depth):
    """a random Sirius Expression over selectors / fixed / advice queries, constants and one challenge"""
    if depth == 0 or rnd.random() < 0.25:
        r = rnd.random()
        if r < 0.45:
            return ("poly", nsel + nfix + rnd.randrange(nadv), rnd.choice([0, 0, 0, 1]))
        if r < 0.75:
            return ("poly", rnd.randrange(nsel + nfix), 0)
        if r < 0.9:
            return ("const", rnd.choice([0, 1, 2, 3, 7, (1 << 64) + 5, rnd.randrange(1 << 250)]))
        return ("chal", 0)
    op = rnd.random()
    a = _random_expr(rnd, nsel, nfix, nadv, depth - 1)
    if op < 0.4:
        return ("prod", a, _random_expr(rnd, nsel, nfix, nadv, depth - 1))
    if op < 0.8:
        return ("sum", a, _random_expr(rnd, nsel, nfix, nadv, depth - 1))
    if op < 0.9:
        return ("neg", a)
    return ("scaled", a, rnd.choice([1, 2, 5, rnd.randrange(1 << 200)]))


def test_emu_jit_random_circuits(emu_jit, oracle):
    from oracle import expr as OE
    rnd = random.Random(2029)
    done = 0
    attempts = 0
    while done < 8 and attempts < 400:
        attempts += 1
        nsel, nfix, nadv = rnd.choice([0, 1]), rnd.randrange(1, 4), rnd.randrange(1, 4)
        gates = [_random_expr(rnd, nsel, nfix, nadv, rnd.randrange(2, 5)) for _ in range(rnd.choice([1, 1, 2]))]
        ctx = OE.QueryIndexContext(nsel, nfix, nadv, 0, 0)
        try:
            degs = [OE.homogeneous(g, ctx)[1] for g in gates]
        except Exception:
            continue
        if not all(2 <= d <= 6 for d in degs):              # folding degrees the single-pass kernel covers, and real cross terms
            continue
        _check(emu_jit, oracle, rnd.choice([0, 1]), 3, gates, nfix, nadv, seed=attempts, nsel=nsel)
        done += 1
    assert done == 8


def test_emu_jit_lookup_structure(emu_jit, oracle):
    """a structure with lookup arguments: its lookup-extended cross-term program (multi-round witness columns) through the compiled kernel"""
    from lookup_cases import run_lookup_case
    run_lookup_case(emu_jit, oracle, "vector", 5)
    run_lookup_case(emu_jit, oracle, "two", 4)


def test_emu_jit_row_sharded(emu_jit, oracle):
    """the compiled kernel under srs_structure_set_shard: only this rank's row stripes are evaluated"""
    from test_sangria_gpu import _row_shard_case
    _row_shard_case(emu_jit, oracle, 1, 1, 12, (2,), 2, with_commit=False)


def test_emu_protogalaxy_random_circuits(emu_jit, oracle):
    """ProtoGalaxy F / G / e of randomly generated gate sets (interpreter leaves: no ahead-of-time kernel exists for them) against the
    oracle's literal restatement -- sizes, F, betas', G (one incoming trace), K, e."""
    import random as _r
    from oracle import expr as OE, protogalaxy as OPG, pyref as P
    from sirius_amd import protogalaxy as PG
    from workloads import rand_fe
    O, S = oracle, emu_jit
    rnd = _r.Random(77)
    done = attempts = 0
    while done < 8 and attempts < 600:
        attempts += 1
        nfix, nadv = rnd.randrange(1, 4), rnd.randrange(1, 4)
        gates = [_random_expr(rnd, 0, nfix, nadv, rnd.randrange(2, 4)) for _ in range(rnd.choice([1, 2]))]
        if any(g[0] == "const" for g in gates):
            continue
        ctx0 = OE.QueryIndexContext(0, nfix, nadv, 0, 0)
        try:
            degs = [OE.homogeneous(g, ctx0)[1] for g in gates]
        except Exception:
            continue
        if not all(1 <= d <= 5 for d in degs):
            continue
        nch = 1 if any(_has_challenge(g) for g in gates) else 0
        k = rnd.choice([2, 3])
        rows = 1 << k
        rng = np.random.default_rng(attempts)
        fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
        Ws = [rand_fe(rng, nadv * rows) for _ in range(2)]
        St = S.PlonkStructure(0, k, [], fixed, nadv, gates)
        ctx = PG.PolyContext(St, 1)
        oS = OPG.Structure(O, list(gates), k, [], fixed, nadv, nch)
        octx = oS.context(1)
        betas = OPG.new_accumulator_betas(rnd.randrange(P.FR), ctx.betas_count)
        delta, alpha = rnd.randrange(P.FR), rnd.randrange(P.FR)
        m = lambda v: O.ints_to_mont(O.FR, list(v))
        chs_i = [[rnd.randrange(P.FR) for _ in range(nch)] for _ in range(2)]          # the traces' challenges (folded with L_j(X) in G)
        chs = [m(c) if nch else np.zeros((0, 4), np.uint64) for c in chs_i]
        for compat in (True, False):
            pF = PG.compute_F(ctx, m(betas), m([delta])[0], Ws[0], challenges=chs[0], reference_compat=compat)
            assert O.mont_to_ints(O.FR, pF) == OPG.compute_F(oS, octx, betas, delta, Ws[0], chs_i[0], compat), ("F", gates)
            bs = OPG.beta_stroke(betas, alpha, delta)
            pG = PG.compute_G(ctx, m(bs), Ws, challenges_list=chs, reference_compat=compat)
            assert O.mont_to_ints(O.FR, pG) == OPG.compute_G(oS, octx, bs, Ws, chs_i, compat), ("G", gates)
            pe = PG.evaluate_e_from_trace(ctx, m(betas), Ws[1], challenges=chs[1], reference_compat=compat)
            assert O.mont_to_ints(O.FR, pe) == [OPG.evaluate_e_from_trace(oS, octx, betas, Ws[1], chs_i[1], compat)], ("e", gates)
        St.close()
        done += 1
    assert done == 8


def _has_challenge(e):
    return e[0] == "chal" or any(isinstance(x, tuple) and _has_challenge(x) for x in e[1:])

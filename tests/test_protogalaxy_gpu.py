"""GPU parity: ProtoGalaxy prover polynomials through the C-ABI vs the oracle's literal restatement
(src/nifs/protogalaxy/poly/mod.rs, folded_witness.rs, mod.rs).  Bit-exact, with the reference's
quirks Q1 (row index), Q2 (K domain) and Q3 (betas) reproduced."""
import numpy as np
import pytest

from pg_cases import run_pg_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,gate_T,L,compat", [(4, [2], 1, True), (4, [2], 1, False), (5, [3, 2], 1, True),
                                              (3, [5, 3, 2], 1, False), (8, [5, 3], 1, True), (10, [5, 3], 1, False),
                                              (4, [2], 3, False), (7, [5], 3, True)])
def test_protogalaxy_vs_oracle(srs, oracle, k, gate_T, L, compat):
    # the reference's own protogalaxy tests fold L = 3 traces at k = 10 (src/nifs/protogalaxy/tests.rs:187-309)
    run_pg_case(srs, oracle, k, gate_T, L, compat)


def test_lagrange_kats(srs, oracle):
    """basic_lagrange_test + correctness_for_cyclic_element (src/polynomial/lagrange.rs:97-127) through the library."""
    from conftest import golden
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    O = oracle
    kat = golden("kat.json")["basic_lagrange_test"]
    got = O.mont_to_ints(O.FR, PG.eval_lagrange_poly_for_cyclic_group(O.ints_to_mont(O.FR, [kat["X"]])[0], kat["log_n"]))
    assert [str(v) for v in got] == kat["output"]
    log_n = 5
    for j, wj in enumerate(P.iter_cyclic_subgroup(log_n)):
        vals = O.mont_to_ints(O.FR, PG.eval_lagrange_poly_for_cyclic_group(O.ints_to_mont(O.FR, [wj])[0], log_n))
        assert vals == [1 if i == j else 0 for i in range(1 << log_n)]


def test_cyclefold_shape_k17_properties(srs, oracle):
    """CycleFold-shaped primary structure at k = 17 (2 gates -> n = 2^18 leaves): size-independent checks.
    (a) F(0) = sum_i pow_i(beta) f_i = evaluate_e(betas);  (b) G(1) = sum of G's coefficients: at X = w^0 = 1
    the folded witness is the accumulator alone (L_0(1) = 1, L_1(1) = 0), so G(1) = evaluate_e(betas_stroke);
    both in reference-compat mode and with true per-row leaves (which must differ from each other)."""
    import torch
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    from workloads import make_structure_inputs
    O = oracle
    w = make_structure_inputs("primary", 17, seed=99)
    S = srs.PlonkStructure(0, 17, [], w["fixed"], w["num_advice"], w["gates"])
    ctx = PG.PolyContext(S, 1)
    assert (ctx.count_of_evaluation_with_padding, ctx.betas_count, ctx.fft_points_count_F, ctx.fft_points_count_G,
            ctx.fft_log_domain_size_K) == (1 << 18, 18, 32, 8, 8)
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    W0, W1 = dev(w["W1"]), dev(w["W2"])
    import random
    rnd = random.Random(1)
    betas = [rnd.randrange(P.FR) for _ in range(18)]
    delta, alpha = rnd.randrange(P.FR), rnd.randrange(P.FR)
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    pF = O.mont_to_ints(O.FR, PG.compute_F(ctx, m(betas), m([delta])[0], W0))
    # evaluate_e with betas_stroke == G at X = w^0 = 1 (L_0(1) = 1, L_1(1) = 0)
    bs = [(b + alpha * pow(delta, 1 << i, P.FR)) % P.FR for i, b in enumerate(betas)]
    pG = O.mont_to_ints(O.FR, PG.compute_G(ctx, m(bs), [W0, W1]))
    e_acc = O.mont_to_ints(O.FR, PG.evaluate_e_from_trace(ctx, m(bs), W0))[0]
    assert sum(pG) % P.FR == e_acc                      # G(1) = sum of coefficients
    # F(0) = sum_i pow_i(beta) f_i = evaluate_e(betas)
    assert pF[0] == O.mont_to_ints(O.FR, PG.evaluate_e_from_trace(ctx, m(betas), W0))[0]
    # non-compat (true per-row) mode: same identities must hold as well
    pG2 = O.mont_to_ints(O.FR, PG.compute_G(ctx, m(bs), [W0, W1], reference_compat=False))
    e2 = O.mont_to_ints(O.FR, PG.evaluate_e_from_trace(ctx, m(bs), W0, reference_compat=False))[0]
    assert sum(pG2) % P.FR == e2 and pG2 != pG

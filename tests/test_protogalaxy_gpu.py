"""GPU parity: ProtoGalaxy prover polynomials through the C-ABI vs the oracle's literal restatement
(src/nifs/protogalaxy/poly/mod.rs, folded_witness.rs, mod.rs).  Bit-exact, with the reference's
quirks Q1 (row index), Q2 (K domain) and Q3 (betas) reproduced."""
import numpy as np
import pytest

from pg_cases import direct_eval_case, high_degree_gates, run_pg_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,gate_T,L,compat", [(4, [2], 1, True), (4, [2], 1, False), (5, [3, 2], 1, True),
                                              (3, [5, 3, 2], 1, False), (8, [5, 3], 1, True), (10, [5, 3], 1, False),
                                              (4, [2], 3, False), (7, [5], 3, True), (4, [2], 7, True), (10, [5, 3], 7, False), (5, [3, 2], 15, True),   # L up to 15 (r05)
                                              (10, [5, 3], 1, True), (11, [5, 3], 3, False)])   # k >= 10 with [5, 3]: specialised leaf kernel
def test_protogalaxy_vs_oracle(srs, oracle, k, gate_T, L, compat):
    # every case compares F, G, K, e, the Lagrange values, fold_witness and the one-call prove; L = 3 cases have a 2^16-point K domain
    # (k_pg_K_points + a device coset_ifft), L >= 7 cases a K "log" above F::S: refused with rc 3 where the reference panics
    run_pg_case(srs, oracle, k, gate_T, L, compat)


def test_protogalaxy_reference_test_shape(srs, oracle):
    """The shape the reference's own ProtoGalaxy tests fold (src/nifs/protogalaxy/tests.rs:187-309: three incoming traces at k = 10;
    with a degree-5 gate: 16 points of G and, by quirk Q2, a 2^16-point K domain), end to end: prove -> K -> gamma -> calculate_e with
    alpha and gamma squeezed from the Poseidon transcript over F's 16 and K's 65536 coefficients."""
    ctx = run_pg_case(srs, oracle, 10, [5], 3, True, ro_check=True)
    assert (ctx.fft_points_count_G, ctx.fft_log_domain_size_K, ctx.instances_to_fold) == (16, 16, 4)


@pytest.mark.parametrize("k,d,L,compat,ro", [(5, 9, 1, True, True), (10, 15, 1, False, False), (10, 9, 1, True, False), (4, 8, 1, False, False),
                                            (6, 6, 1, True, None), (4, 7, 3, False, False)])
def test_protogalaxy_high_degree_gate_vs_oracle(srs, oracle, k, d, L, compat, ro):
    """BASELINE configs[3] names a "high-degree gate": one incoming trace and a gate of degree 8..15 give 16 points of G
    (src/nifs/protogalaxy/poly/mod.rs:535-545) -- the non-8-point instantiation of the leaf kernels -- and a 2^16-point K domain
    (:263-268).  Degree 6 / 7 stay at 8 points; L = 3 at degree 7 reaches 32 points and a refused K."""
    ctx = run_pg_case(srs, oracle, k, None, L, compat, gates=high_degree_gates(d), ro_check=ro)
    want_G = 1 << (L * d).bit_length()
    assert ctx.fft_points_count_G == want_G


def test_protogalaxy_K_domain_above_S_is_refused(srs, oracle):
    """points_G >= 32 -> fft_log_domain_size_K = 32 > F::S = 28: `get_omega_or_inv` panics "k should no larger than F::S" in the
    reference (src/fft.rs:13); rc 3 here from compute_K_from_G and from prove (F, G, e and the fold still compare with the oracle)."""
    ctx = run_pg_case(srs, oracle, 3, None, 1, False, gates=high_degree_gates(16))
    assert (ctx.fft_points_count_G, ctx.fft_log_domain_size_K) == (32, 32)
    ctx = run_pg_case(srs, oracle, 4, [2], 7, True)            # 8 instances at degree 5: 64 points, "log" 64
    assert ctx.fft_log_domain_size_K == 64


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


def _pg_fold_identity(S, O, k, gate_T, compat):
    """ProtoGalaxy correctness identity the reference's decider checks (is_sat_accumulation, src/nifs/protogalaxy/mod.rs:
    `acc.e == evaluate_e_from_trace(acc)`): with a SATISFYING incoming trace, after folding with gamma
        calculate_e(F, K, gamma, alpha) == evaluate_e_from_trace(fold_witness(acc, incoming; L(gamma)), betas_stroke)."""
    import random
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    from test_sangria_gpu import _satisfying_witness
    from workloads import gates_for, rand_fe
    gate_T = list(gate_T)
    rows = 1 << k
    gates, nfix, nadv = gates_for(gate_T)
    rng = np.random.default_rng(k + 17)
    fixed = [rand_fe(rng, rows, 0.2) for _ in range(nfix)]
    one = O.ints_to_mont(O.FR, [1])[0]
    fo = 0
    for T in gate_T:
        fixed[fo + 2 * T + 3][:] = one
        fo += 2 * T + 5
    St = S.PlonkStructure(0, k, [], fixed, nadv, gates)
    ctx = PG.PolyContext(St, 1)
    Wacc = rand_fe(rng, nadv * rows)                       # arbitrary accumulator
    Win = _satisfying_witness(O, 0, k, gate_T, fixed, rng)  # satisfying incoming trace: f_i(w_in) = 0
    rnd = random.Random(k)
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    betas = [rnd.randrange(P.FR) for _ in range(ctx.betas_count)]
    delta, alpha, gamma = (rnd.randrange(P.FR) for _ in range(3))
    pF = PG.compute_F(ctx, m(betas), m([delta])[0], Wacc, reference_compat=compat)
    bs, d = [], delta
    for b in betas:
        bs.append((b + alpha * d) % P.FR); d = d * d % P.FR
    pG = PG.compute_G(ctx, m(bs), [Wacc, Win], reference_compat=compat)
    Fa = PG.poly_eval(pF, m([alpha])[0])
    pK = PG.compute_K_from_G(ctx, pG, Fa)
    e_new = PG.calculate_e(pF, pK, m([gamma])[0], m([alpha])[0], ctx.lagrange_domain)
    Lg = PG.eval_lagrange_poly_for_cyclic_group(m([gamma])[0], ctx.lagrange_domain)
    Wf = PG.fold_witness(0, [Wacc, Win], Lg)
    assert np.array_equal(e_new, PG.evaluate_e_from_trace(ctx, m(bs), Wf, reference_compat=compat))
    # and G(gamma) itself equals that value: K was an exact quotient
    assert np.array_equal(PG.poly_eval(pG, m([gamma])[0]), e_new)
    St.close()


def test_protogalaxy_fold_identity(srs, oracle):
    _pg_fold_identity(srs, oracle, 6, (5, 3), True)
    _pg_fold_identity(srs, oracle, 10, (5, 3), False)
    _pg_fold_identity(srs, oracle, 5, (2,), False)


def test_cmp_with_direct_eval(srs, oracle):
    """src/nifs/protogalaxy/poly/mod.rs:639-760 restated on the product."""
    direct_eval_case(srs, oracle, 5, [5, 3], True)
    direct_eval_case(srs, oracle, 6, [3, 2], False)
    direct_eval_case(srs, oracle, 10, [5, 3], False)      # specialised leaf kernel


def test_fast_paths_equal_general_paths(srs, oracle):
    """compute_F polynomial tree == evaluate-and-interpolate, compute_G integer points == roots of unity + ifft (k = 13,
    both leaf modes): the fast paths are algebraic rewrites, the coefficient vectors must be identical."""
    import random
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    from workloads import gates_for, rand_fe
    O = oracle
    k, gate_T = 13, [5, 3]
    rows = 1 << k
    gates, nfix, nadv = gates_for(gate_T)
    rng = np.random.default_rng(3)
    fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
    Ws = [rand_fe(rng, nadv * rows) for _ in range(2)]
    St = srs.PlonkStructure(0, k, [], fixed, nadv, gates)
    ctx = PG.PolyContext(St, 1)
    rnd = random.Random(1)
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    betas = m([rnd.randrange(P.FR) for _ in range(ctx.betas_count)])
    delta = m([rnd.randrange(P.FR)])[0]
    for compat in (True, False):
        fast_F = PG.compute_F(ctx, betas, delta, Ws[0], reference_compat=compat)
        fast_G = PG.compute_G(ctx, betas, Ws, reference_compat=compat)
        with srs.tuning(pg_f_eval=1, pg_g_fft=1):      # the routes of small tables / of L >= 2 incoming traces, on this shape
            gen_F = PG.compute_F(ctx, betas, delta, Ws[0], reference_compat=compat)
            gen_G = PG.compute_G(ctx, betas, Ws, reference_compat=compat)
        assert np.array_equal(fast_F, gen_F) and np.array_equal(fast_G, gen_G)
        assert fast_F[ctx.betas_count + 1:].any() == False      # degree t polynomial: higher coefficients are exactly zero
    St.close()


def _config_size_case(srs, oracle, k, compare_oracle, high_degree=None, modes=(True, False)):
    """BASELINE configs[2] / [3] shapes: the primary CycleFold structure (MainGate<5> + MainGate<3>, 12 advice / 26 fixed,
    2 gates -> n = 2^(k+1) leaves).  Size-independent identities in both leaf modes:
      F(0) = evaluate_e(betas);  G(1) = evaluate_e(betas');  deg F <= t;  K and e against the oracle's (cheap) restatement;
      fold_witness is linear: fold(acc, in; L) - L0 acc - L1 in = 0 on a sample of rows.
    With compare_oracle the coefficient vectors of F and G and the value e are compared with the CPU oracle (oracle/
    protogalaxy.py *_fast: the reference's leaf function, folded witnesses and reduction trees, in C) at FULL size.
    high_degree = d: the first gate is `workloads.high_degree_gate(5, d)` (same columns): 16 points of G, K on 2^16 points."""
    import random
    import torch
    from oracle import expr as OE
    from oracle import protogalaxy as OPG
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    from workloads import make_structure_inputs
    O = oracle
    w = make_structure_inputs("primary", k, seed=1000 + k)
    if high_degree:      # configs[3]'s "high-degree gate": MainGate<5> + a monomial of degree d over the same 12 advice / 26 fixed columns
        from workloads import high_degree_gate
        w["gates"] = [high_degree_gate(5, high_degree, 0, 0, 0, w["num_fixed"]), w["gates"][1]]
    S = srs.PlonkStructure(0, k, [], w["fixed"], w["num_advice"], w["gates"])
    ctx = PG.PolyContext(S, 1)
    t = k + 1
    assert (ctx.count_of_evaluation_with_padding, ctx.betas_count, ctx.fft_points_count_F, ctx.fft_points_count_G,
            ctx.fft_log_domain_size_K) == ((1 << t, t, 32, 16, 16) if high_degree else (1 << t, t, 32, 8, 8))
    dev = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    W0, W1 = dev(w["W1"]), dev(w["W2"])
    rnd = random.Random(k)
    betas = [rnd.randrange(P.FR) for _ in range(t)]
    delta, alpha, gamma = (rnd.randrange(P.FR) for _ in range(3))
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    bs = OPG.beta_stroke(betas, alpha, delta)
    if compare_oracle:
        gate_T = [5, 3]
        og, fo, ao = [], 0, 0
        for T in gate_T:
            og.append(OE.main_gate_expression(T, 0, fo, ao, w["num_fixed"])); fo += 2 * T + 5; ao += T + 2
        if high_degree:
            og[0] = w["gates"][0]                                  # the node tuples are the oracle's too
        oS = OPG.Structure(O, og, k, [], w["fixed"], w["num_advice"], 0)
        octx = oS.context(1)
    seen = {}
    for compat in modes:
        pF = PG.compute_F(ctx, m(betas), m([delta])[0], W0, reference_compat=compat)
        pG = PG.compute_G(ctx, m(bs), [W0, W1], reference_compat=compat)
        iF, iG = O.mont_to_ints(O.FR, pF), O.mont_to_ints(O.FR, pG)
        e_b = O.mont_to_ints(O.FR, PG.evaluate_e_from_trace(ctx, m(betas), W0, reference_compat=compat))[0]
        e_bs = O.mont_to_ints(O.FR, PG.evaluate_e_from_trace(ctx, m(bs), W0, reference_compat=compat))[0]
        assert iF[0] == e_b, "F(0) = evaluate_e(betas)"
        assert sum(iG) % P.FR == e_bs, "G(1) = evaluate_e(betas')"
        assert all(c == 0 for c in iF[t + 1:]), "deg F <= t"
        Fa = PG.poly_eval(pF, m([alpha])[0])
        pK = PG.compute_K_from_G(ctx, pG, Fa)
        # K (256 coefficients, quirk Q2; 2^16 with a high-degree gate) and e against the oracle's literal compute_K_from_G / calculate_e on the product's F, G:
        # cheap at any k (the incoming witness is random, not satisfying, so K is an interpolant, not an exact quotient)
        Fa_i = OPG.poly_eval(iF, alpha)
        assert O.mont_to_ints(O.FR, Fa) == [Fa_i]
        from oracle.protogalaxy import PolyContext as _OC      # sizes only
        class _Ctx:                                            # the three sizes compute_K_from_G needs
            def fft_log_domain_size_K(self): return ctx.fft_log_domain_size_K
            def lagrange_domain(self): return ctx.lagrange_domain
            instances_to_fold = ctx.instances_to_fold
        eK = OPG.compute_K_from_G(_Ctx(), iG, Fa_i)
        assert O.mont_to_ints(O.FR, pK) == eK, "compute_K_from_G"
        e_new = PG.calculate_e(pF, pK, m([gamma])[0], m([alpha])[0], ctx.lagrange_domain)
        assert O.mont_to_ints(O.FR, e_new) == [OPG.calculate_e(iF, eK, gamma, alpha, ctx.lagrange_domain)], "calculate_e"
        seen[compat] = (iF, iG)
        if compare_oracle:
            assert iF == OPG.compute_F_fast(oS, octx, betas, delta, w["W1"], [], compat), f"compute_F vs oracle (compat={compat})"
            assert iG == OPG.compute_G_fast(oS, octx, bs, [w["W1"], w["W2"]], [[], []], compat), f"compute_G vs oracle (compat={compat})"
            assert e_b == OPG.evaluate_e_fast(oS, octx, betas, w["W1"], [], compat), "evaluate_e vs oracle"
    assert len(modes) < 2 or seen[True] != seen[False]
    Lg = PG.eval_lagrange_poly_for_cyclic_group(m([gamma])[0], ctx.lagrange_domain)
    Wf = PG.fold_witness(0, [W0, W1], Lg)
    idx = torch.from_numpy(np.random.default_rng(k).integers(0, w["W1"].shape[0], size=4096)).cuda()
    got = Wf[idx].cpu().numpy().view(np.uint64)
    want = O.lincomb(O.FR, [np.ascontiguousarray(w["W1"][idx.cpu().numpy()]), np.ascontiguousarray(w["W2"][idx.cpu().numpy()])],
                     O.ints_to_mont(O.FR, O.mont_to_ints(O.FR, Lg)[:2]))
    assert np.array_equal(got, want), "fold_witness rows"
    S.close()


def test_cyclefold_config_k20_vs_oracle(srs, oracle):
    """BASELINE configs[2] (cyclefold_poseidon, k = 20): ProtoGalaxy F / G / e at full size against the CPU oracle."""
    _config_size_case(srs, oracle, 20, compare_oracle=True)


def test_protogalaxy_config_k22_vs_oracle(srs, oracle):
    """BASELINE configs[3] (ProtoGalaxy at k = 22, n = 2^23 leaves) on one GPU: F / G / e at full size against the CPU oracle
    (both leaf modes) and the size-independent identities."""
    _config_size_case(srs, oracle, 22, compare_oracle=True)


def test_protogalaxy_high_degree_config_k20_vs_oracle(srs, oracle):
    """configs[3]'s "high-degree gate" at a bench size: a degree-9 gate -> 16 points of G (the leaf kernels' 16-point
    instantiation), K on 2^16 points; F / G / e at full size against the CPU oracle, K and calculate_e against the literal restatement.
    Intended leaf rows here; the k = 22 case below runs the reference's rows."""
    _config_size_case(srs, oracle, 20, compare_oracle=True, high_degree=9, modes=(False,))


def test_protogalaxy_high_degree_config_k22_vs_oracle(srs, oracle):
    """BASELINE configs[3] as named: ProtoGalaxy high-degree-gate fold at k = 22 (n = 2^23 leaves, degree 15) on one GPU, the
    reference's leaf rows, full size against the CPU oracle."""
    _config_size_case(srs, oracle, 22, compare_oracle=True, high_degree=15, modes=(True,))

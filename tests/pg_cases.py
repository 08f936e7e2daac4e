"""Shared ProtoGalaxy parity case (product vs the oracle's literal restatement)."""
import random

import numpy as np


def high_degree_gates(d, T=2):
    """([gate], nfix, nadv) of `workloads.high_degree_gate`: MainGate<T> plus one monomial of degree d (d = 8..15 -> 16 points of G,
    a 2^16-point K domain by quirk Q2; d >= 16 -> 32 points -> K's domain "log" 32 > F::S: the reference panics in fft.rs:13)."""
    from workloads import high_degree_gate
    return [high_degree_gate(T, d)], 2 * T + 5, T + 2


def run_pg_case(S, O, k, gate_T, L_traces, compat, seed=4, gates=None, ro_check=None):
    """Every ProtoGalaxy prover polynomial of one (structure, L) case, product vs the oracle's literal restatement -- F, beta',
    G, F(alpha), K, calculate_e, evaluate_e, the Lagrange values, fold_witness, and `prove` as one call (given challenges, and
    with alpha / gamma squeezed from a Poseidon transcript) -- whatever K's domain is: 2^8 points (L = 1, degree <= 7), 2^16 points
    (L = 3 at degree 5 = the reference's own test shape, src/nifs/protogalaxy/tests.rs:187-309; L = 1 at degree 8..15), or a
    "log" above F::S = 28 (L >= 7; degree >= 16), where the reference panics in fft.rs:13 and the product returns rc 3.
    gates = (product gates, nfix, nadv) replaces the MainGate<T> list (the node tuples are the oracle's too).  ro_check: compare the
    transcript variant of prove as well (default: whenever K has 256 coefficients; the python sponge over 2^16 takes ~10 s)."""
    from oracle import expr as OE
    from oracle import protogalaxy as OPG
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    from workloads import gates_for, rand_fe
    rnd = random.Random(seed)
    rows = 1 << k
    if gates is None:
        gates, nfix, nadv = gates_for(gate_T)
        og, fo, ao = [], 0, 0
        for T in gate_T:
            og.append(OE.main_gate_expression(T, 0, fo, ao, nfix)); fo += 2 * T + 5; ao += T + 2
    else:
        gates, nfix, nadv = gates
        og = list(gates)
    rng = np.random.default_rng(k * 7 + len(gates) + seed)
    fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
    Ws = [rand_fe(rng, nadv * rows) for _ in range(L_traces + 1)]
    St = S.PlonkStructure(0, k, [], fixed, nadv, gates)
    ctx = PG.PolyContext(St, L_traces)
    oS = OPG.Structure(O, og, k, [], fixed, nadv, 0)
    octx = oS.context(L_traces)
    # PolyContext sizes incl. Q2 (K "log" domain)  -- poly/mod.rs:205-269
    assert (ctx.count_of_evaluation_with_padding, ctx.betas_count, ctx.fft_points_count_F, ctx.fft_points_count_G,
            ctx.fft_log_domain_size_K, ctx.lagrange_domain) == \
        (octx.count_with_padding, octx.betas_count(), octx.fft_points_count_F(), octx.fft_points_count_G,
         octx.fft_log_domain_size_K(), octx.lagrange_domain())
    logK = octx.fft_log_domain_size_K()
    t = ctx.betas_count
    beta = rnd.randrange(P.FR)
    betas = OPG.new_accumulator_betas(beta, t)                 # Q3
    delta, alpha, gamma = (rnd.randrange(P.FR) for _ in range(3))
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    pF = PG.compute_F(ctx, m(betas), m([delta])[0], Ws[0], reference_compat=compat)
    eF = OPG.compute_F(oS, octx, betas, delta, Ws[0], [], compat)
    assert O.mont_to_ints(O.FR, pF) == eF, "compute_F"
    bs = OPG.beta_stroke(betas, alpha, delta)
    assert O.mont_to_ints(O.FR, PG.beta_stroke(m(betas), m([alpha])[0], m([delta])[0])) == bs, "beta_stroke"
    pG = PG.compute_G(ctx, m(bs), Ws, reference_compat=compat)
    eG = OPG.compute_G(oS, octx, bs, Ws, [[] for _ in Ws], compat)
    assert O.mont_to_ints(O.FR, pG) == eG, "compute_G"
    Fa = OPG.poly_eval(eF, alpha)
    assert O.mont_to_ints(O.FR, PG.poly_eval(pF, m([alpha])[0])) == [Fa]
    pe = PG.evaluate_e_from_trace(ctx, m(betas), Ws[0], reference_compat=compat)
    assert O.mont_to_ints(O.FR, pe) == [OPG.evaluate_e_from_trace(oS, octx, betas, Ws[0], [], compat)], "evaluate_e"
    Lg = P.eval_lagrange_poly_for_cyclic_group(gamma, octx.lagrange_domain())
    assert O.mont_to_ints(O.FR, PG.eval_lagrange_poly_for_cyclic_group(m([gamma])[0], ctx.lagrange_domain)) == Lg
    assert np.array_equal(PG.fold_witness(0, Ws, m(Lg)), OPG.fold_witness(O, Ws, Lg)), "fold_witness"
    import torch
    dWs = [torch.from_numpy(w.view(np.int64)).cuda() for w in Ws] if torch.cuda.is_available() else Ws    # device-resident (emulator: host)
    if logK > P.FR_S:
        # the reference cannot get past K here: coset_ifft -> get_omega_or_inv asserts "k should no larger than F::S" (src/fft.rs:13).
        # The product reports the same condition as rc 3 from K itself and from the one-call prove, outputs untouched.
        for call in (lambda: PG.compute_K_from_G(ctx, pG, m([Fa])[0]),
                     lambda: PG.prove(ctx, m(betas), m([delta])[0], dWs, alpha=m([alpha])[0], gamma=m([gamma])[0], reference_compat=compat)):
            try:
                call()
            except S.SiriusAmdError as e:
                assert e.rc == 3 and "should no larger than F::S" in str(e), e
            else:
                raise AssertionError("a K domain above F::S must be refused")
        St.close()
        return ctx
    pK = PG.compute_K_from_G(ctx, pG, m([Fa])[0])
    eK = OPG.compute_K_from_G(octx, eG, Fa)
    assert pK.shape[0] == len(eK) == 1 << logK
    assert O.mont_to_ints(O.FR, pK) == eK, "compute_K_from_G"
    got = PG.calculate_e(pF, pK, m([gamma])[0], m([alpha])[0], ctx.lagrange_domain)
    assert O.mont_to_ints(O.FR, got) == [OPG.calculate_e(eF, eK, gamma, alpha, octx.lagrange_domain())], "calculate_e"
    # ProtoGalaxy::prove as one call (srs_pg_prove) == the step-by-step results; with an oracle: alpha / gamma squeezed inside
    pr = PG.prove(ctx, m(betas), m([delta])[0], dWs, alpha=m([alpha])[0], gamma=m([gamma])[0], reference_compat=compat)
    assert np.array_equal(pr["poly_F"], pF) and np.array_equal(pr["poly_K"], pK) and O.mont_to_ints(O.FR, pr["betas_stroke"]) == bs
    assert np.array_equal(pr["e"], got) and np.array_equal(pr["lagrange"], m(Lg)[: len(Ws)])
    Wf = pr["W"]
    assert np.array_equal(Wf.cpu().numpy().view(np.uint64).reshape(-1, 4) if hasattr(Wf, "cpu") else Wf, OPG.fold_witness(O, Ws, Lg))
    if ro_check if ro_check is not None else logK <= 8:
        from oracle import poseidon as OP
        ro, oro = S.PoseidonHash(0, 5, 4, 10, 10), OP.PoseidonHash(P.FR, 5, 4, 10, 10)
        ro.absorb_field(m([delta])); oro.absorb_field_iter([delta])
        pr2 = PG.prove(ctx, m(betas), m([delta])[0], dWs, ro=ro, reference_compat=compat)
        a2 = oro.absorb_field_iter(O.mont_to_ints(O.FR, pr2["poly_F"])).squeeze(255)
        assert O.mont_to_ints(O.FR, pr2["alpha"]) == [a2] and np.array_equal(pr2["poly_F"], pF)
        g2 = oro.absorb_field_iter(O.mont_to_ints(O.FR, pr2["poly_K"])).squeeze(255)
        assert O.mont_to_ints(O.FR, pr2["gamma"]) == [g2]
        eG2 = OPG.compute_G(oS, octx, OPG.beta_stroke(betas, a2, delta), Ws, [[] for _ in Ws], compat)
        eK2 = OPG.compute_K_from_G(octx, eG2, OPG.poly_eval(eF, a2))
        assert O.mont_to_ints(O.FR, pr2["poly_K"]) == eK2
        assert O.mont_to_ints(O.FR, pr2["e"]) == [OPG.calculate_e(eF, eK2, g2, a2, octx.lagrange_domain())]
    St.close()
    return ctx


def direct_eval_case(S, O, k, gate_T, compat, seed=9):
    """The reference's own ProtoGalaxy tests (src/nifs/protogalaxy/poly/mod.rs:639-760: cmp_with_direct_eval_of_F / _G):
    the polynomial returned by compute_F / compute_G, evaluated at points of the FFT domain and at random points, equals
    the direct sum  sum_i pow_i(challenge vector at X) * f_i(witness at X)."""
    from oracle import expr as OE
    from oracle import protogalaxy as OPG
    from oracle import pyref as P
    from sirius_amd import protogalaxy as PG
    from workloads import gates_for, rand_fe
    FR = P.FR
    rnd = random.Random(seed)
    rows = 1 << k
    gates, nfix, nadv = gates_for(gate_T)
    og, fo, ao = [], 0, 0
    for T in gate_T:
        og.append(OE.main_gate_expression(T, 0, fo, ao, nfix)); fo += 2 * T + 5; ao += T + 2
    rng = np.random.default_rng(seed + k)
    fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
    Ws = [rand_fe(rng, nadv * rows) for _ in range(2)]
    St = S.PlonkStructure(0, k, [], fixed, nadv, gates)
    ctx = PG.PolyContext(St, 1)
    oS = OPG.Structure(O, og, k, [], fixed, nadv, 0)
    n, t = ctx.count_of_evaluation_with_padding, ctx.betas_count
    m = lambda v: O.ints_to_mont(O.FR, list(v))
    betas = [rnd.randrange(FR) for _ in range(t)]
    delta = rnd.randrange(FR)

    def pow_i(i, c):
        out = 1
        for b in range(len(c)):
            if (i >> b) & 1:
                out = out * c[b] % FR
        return out
    # ---- F
    pF = O.mont_to_ints(O.FR, PG.compute_F(ctx, m(betas), m([delta])[0], Ws[0], reference_compat=compat))
    f = oS.evaluate_witness_fn(Ws[0], [], compat)
    leaves = [f(i) for i in range(n)]
    deltas = [delta]
    for _ in range(t - 1):
        deltas.append(deltas[-1] * deltas[-1] % FR)
    w = P.get_omega_or_inv(ctx.fft_points_count_F.bit_length() - 1, False)
    pts = [pow(w, j, FR) for j in range(ctx.fft_points_count_F)] + [rnd.randrange(FR) for _ in range(3)]
    for X in pts:
        c = [(b + X * d) % FR for b, d in zip(betas, deltas)]
        direct = sum(pow_i(i, c) * leaves[i] for i in range(n)) % FR
        assert OPG.poly_eval(pF, X) == direct, "compute_F vs direct evaluation"
    # ---- G: leaves on the Lagrange-folded witness at X, weights beta' (constant in X)
    bs = [rnd.randrange(FR) for _ in range(t)]
    pG = O.mont_to_ints(O.FR, PG.compute_G(ctx, m(bs), Ws, reference_compat=compat))
    wG = P.get_omega_or_inv(ctx.fft_points_count_G.bit_length() - 1, False)
    ptsG = [pow(wG, j, FR) for j in range(ctx.fft_points_count_G)][:4] + [rnd.randrange(FR)]
    W_int = [O.mont_to_ints(O.FR, w_) for w_ in Ws]
    for X in ptsG:
        Lx = P.eval_lagrange_poly_for_cyclic_group(X, ctx.lagrange_domain)
        folded = [(Lx[0] * a + Lx[1] * b) % FR for a, b in zip(*W_int)]
        fX = oS.evaluate_witness_fn(O.ints_to_mont(O.FR, folded), [], compat)
        direct = sum(pow_i(i, bs) * fX(i) for i in range(n)) % FR
        assert OPG.poly_eval(pG, X) == direct, "compute_G vs direct evaluation"
    St.close()

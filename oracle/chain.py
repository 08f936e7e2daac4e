"""The chains bench.py times, recomputed on the CPU oracle.  TEST INFRASTRUCTURE (allowed importers: tests/, bench.py --verify).

bench.py's headline folds a CycleFold chain with the reference's leaf rows (`index & 2^k`, src/plonk/mod.rs:714) and every challenge
squeezed from the off-circuit Poseidon oracle over the transcript (protogalaxy/mod.rs:80-133, :400-481; sangria/mod.rs:162-179,
:253-277); `oracle_chain` restates the same steps with the oracle's ProtoGalaxy / Sangria / best_multiexp / Poseidon restatements
on the same synthetic inputs and returns the same `state_digest`.  `oracle_chain_sangria` does the same for the two-curve Sangria
chain of BASELINE configs[1].  The product package is used ONLY to read the synthetic commitment keys (the bases are inputs,
SURVEY.md 8c)."""
import hashlib
import random

import numpy as np


def oracle_chain(O, S_for_keys, k, log_key, ks, steps, compat=True, fast=False, threads=0):
    """The same chain on the oracle.  `S_for_keys`: the product package, used ONLY to read the synthetic commitment keys
    (P_i = [h(seed, i)] G is generated on the device; the bases are inputs) -- every commitment, polynomial,
    challenge and fold below is computed by oracle/."""
    from oracle import expr as OE
    from oracle import poseidon as OP
    from oracle import protogalaxy as OPG
    from oracle import pyref as P
    from sirius_amd.workloads import make_structure_inputs, make_support_inputs, trace_like
    FR, FQ = P.MODULI[0], P.MODULI[1]
    ints = lambda f, a: O.mont_to_ints(f, np.asarray(a).reshape(-1, 4))
    mont = lambda f, v: O.ints_to_mont(f, [int(x) for x in v])

    # ---- primary (bn256 circuit over Fr): bench.PgPrimary
    w = make_structure_inputs("primary", k, seed=0x5349524955530000 + 3)
    n = w["num_advice"] * w["rows"]
    ck = S_for_keys.CommitmentKey.setup_synthetic(0, 1 << log_key, seed=42)
    bases = ck.bases()
    ck.close()
    oS = OPG.Structure(O, w["gates"], k, [], w["fixed"], w["num_advice"], 0)
    octx = oS.context(1)
    rnd = random.Random(3)
    betas = [rnd.randrange(FR) for _ in range(octx.betas_count())]
    [rnd.randrange(FR) for _ in range(3)]                       # the seeded delta / alpha / gamma bench.py draws and does not use here
    accW, inW = w["W1"], w["W2"]
    host_W = [w["W2"], trace_like(np.random.default_rng(77), n)]
    ident = np.zeros(8, np.uint64)
    accC, inC = ident.copy(), O.msm(0, inW, bases[:n], threads)
    # ---- support circuit (grumpkin circuit over Fq): bench.SangriaSide
    sw = make_support_inputs(ks, seed=0x5349524955530000 + 4)
    sck = S_for_keys.CommitmentKey.setup_synthetic(1, 1 << (ks + 2), seed=43)
    sbases = sck.bases()
    sck.close()
    s_accW, s_accE, s_inW = sw["W1"], sw["E"], sw["W2"]
    s_n = sw["num_advice"] * sw["rows"]
    s_accCW, s_accCE = ident.copy(), ident.copy()
    s_inC = O.msm(1, s_inW, sbases[:s_n], threads)
    s_ch = np.concatenate([sw["u1_u"].reshape(1, 4), O.ints_to_mont(1, [1])])      # U1.ch (none) || U1.u || U2.ch (none) || 1

    def padd(curve, a, b):
        return O.point_add(curve, a, b)

    def pmul(curve, s_mont, p):
        return O.point_mul(curve, s_mont, p)

    step_no, e = 0, None
    for _ in range(steps):
        # A. ProtoGalaxy::prove on the primary  (protogalaxy/mod.rs:400-481)
        ro = OP.PoseidonHash(FR, 5, 4, 10, 10)
        ro.absorb_field_iter(ints(0, np.concatenate([accC.reshape(2, 4), inC.reshape(2, 4)])))     # bench: coordinates as Fr bit patterns
        ro.absorb_field_iter(betas)
        delta = ro.squeeze(255)                                                                    # MAX_BITS, protogalaxy/mod.rs:96-101
        pF = (OPG.compute_F_fast(oS, octx, betas, delta, accW, [], compat, threads) if fast else
              OPG.compute_F(oS, octx, betas, delta, accW, [], compat))
        alpha = ro.absorb_field_iter(pF).squeeze(255)
        bs = OPG.beta_stroke(betas, alpha, delta)
        pG = (OPG.compute_G_fast(oS, octx, bs, [accW, inW], [[], []], compat, threads) if fast else
              OPG.compute_G(oS, octx, bs, [accW, inW], [[], []], compat))
        pK = OPG.compute_K_from_G(octx, pG, OPG.poly_eval(pF, alpha))
        gamma = ro.absorb_field_iter(pK).squeeze(255)
        e = OPG.calculate_e(pF, pK, gamma, alpha, octx.lagrange_domain())
        Lg = P.eval_lagrange_poly_for_cyclic_group(gamma, octx.lagrange_domain())
        accW = (O.lincomb(O.FR, [accW, inW], mont(0, Lg[:2]), threads) if fast else OPG.fold_witness(O, [accW, inW], Lg))
        Lm = mont(0, Lg[:2])
        accC = padd(0, pmul(0, Lm[0], accC), pmul(0, Lm[1], inC))             # fold_instance (:212-271)
        betas = bs
        # B. the support circuit: witness commit, then VanillaFS::prove  (sangria/mod.rs:253-277)
        s_inC = O.msm(1, s_inW, sbases[:s_n], threads)
        sro = OP.PoseidonHash(FR, 5, 4, 10, 10)                                # RO over grumpkin's base field = Fr
        for pt in (s_accCW, s_accCE, s_inC):
            sro.absorb_point(tuple(ints(0, pt.reshape(2, 4))))
        _, T = OE.cross_terms_oracle(O, 1, sw["gates"], 1, sw["num_fixed"], sw["num_advice"], sw["selectors"], sw["fixed"], s_accW, s_inW, s_ch, threads)
        Tc = [O.msm(1, t, sbases[: t.shape[0]], threads) for t in T]
        for c in Tc:
            sro.absorb_point(tuple(ints(0, c.reshape(2, 4))))
        r = mont(1, [sro.squeeze(128)])[0]                                     # the challenge lives in the scalar field Fq
        s_accW = O.fold_w(1, s_accW, s_inW, r, threads)
        s_accE = O.fold_e(1, s_accE, T, r, threads)
        rp = r.copy()
        for c in Tc:                                                           # E_c' = E_c + sum r^k T_c_k  (accumulator.rs:240-244)
            s_accCE = padd(1, s_accCE, pmul(1, rp, c))
            rp = O.fe_mul(1, rp.reshape(1, 4), r.reshape(1, 4))[0]
        s_accCW = padd(1, s_accCW, pmul(1, r, s_inC))
        # C. the new primary witness is committed  (run_sps_protocol_1, plonk/mod.rs:441-447)
        inW = host_W[step_no & 1]
        step_no += 1
        inC = O.msm(0, inW, bases[:n], threads)
    e_m = mont(0, [e])[0]
    return hashlib.sha256(b"".join(np.ascontiguousarray(x, dtype=np.uint64).tobytes() for x in
                                   (e_m, accC, inC, s_accCW, s_accCE, s_inC))).hexdigest()



def oracle_chain_sangria(O, S_for_keys, k, log_key, steps):
    """The same two-curve Sangria chain (SangriaIVC::fold_step's hot path, src/ivc/sangria/incrementally_verifiable_computation.rs:429-635;
    VanillaFS::prove, src/nifs/sangria/mod.rs:253-277) on the oracle: literal GroupedPoly / GraphEvaluator cross terms, best_multiexp,
    the Poseidon transcript of generate_challenge (:162-179), the witness / error / instance folds (accumulator.rs:201-264, 364-404).
    Every step folds the same incoming trace into each circuit's accumulator (the bench's synthetic traces are fixed), so the order of the
    two circuits inside a step does not matter."""
    from oracle import expr as OE
    from oracle import poseidon as OP
    from oracle import pyref as P
    from sirius_amd.workloads import make_structure_inputs, sangria_shape
    ints = lambda f, a: O.mont_to_ints(f, np.asarray(a).reshape(-1, 4))
    ident = np.zeros(8, np.uint64)
    sides = []
    for which, seed in (("primary", 2), ("secondary", 3)):
        w = make_structure_inputs(which, k, seed=0x5349524955530000 + seed)
        gate_T = sangria_shape(which)["gate_T"]
        nfix = sum(2 * T + 5 for T in gate_T)
        og, fo, ao = [], 0, 0
        for T in gate_T:
            og.append(OE.main_gate_expression(T, 0, fo, ao, nfix))
            fo += 2 * T + 5
            ao += T + 2
        field, curve = w["field"], w["curve"]
        ck = S_for_keys.CommitmentKey.setup_synthetic(curve, 1 << log_key, seed=42 + curve)
        bases = ck.bases()
        ck.close()
        n = w["num_advice"] * w["rows"]
        one = O.ints_to_mont(field, [1])
        ch = np.concatenate([w["u1_challenges"].reshape(-1, 4), w["u1_u"].reshape(1, 4), w["u2_challenges"].reshape(-1, 4), one])
        sides.append(dict(w=w, og=og, field=field, curve=curve, bases=bases, n=n, ch=ch, accW=w["W1"], accE=w["E"], inW=w["W2"],
                          accCW=ident.copy(), accCE=ident.copy(), inC=None, r=w["r"], sf=0 if curve == 0 else 1, bf=1 if curve == 0 else 0))
    for _ in range(steps):
        for sd in sides:
            w, curve, field, sf, bf = sd["w"], sd["curve"], sd["field"], sd["sf"], sd["bf"]
            sd["inC"] = O.msm(curve, sd["inW"], sd["bases"][: sd["n"]])
            ro = OP.PoseidonHash(P.MODULI[bf], 5, 4, 10, 10)
            for pt in (sd["accCW"], sd["accCE"], sd["inC"]):
                ro.absorb_point(tuple(ints(bf, pt.reshape(2, 4))))
            _, T = OE.cross_terms_oracle(O, field, sd["og"], 0, w["num_fixed"], w["num_advice"], [], w["fixed"], sd["accW"], sd["inW"], sd["ch"])
            Tc = [O.msm(curve, t, sd["bases"][: t.shape[0]]) for t in T]
            for c in Tc:
                ro.absorb_point(tuple(ints(bf, c.reshape(2, 4))))
            r = O.ints_to_mont(sf, [ro.squeeze(128)])[0]
            sd["accW"] = O.fold_w(field, sd["accW"], sd["inW"], r)
            sd["accE"] = O.fold_e(field, sd["accE"], T, r)
            rp = r.copy()
            for c in Tc:
                sd["accCE"] = O.point_add(curve, sd["accCE"], O.point_mul(curve, rp, c))
                rp = O.fe_mul(sf, rp.reshape(1, 4), r.reshape(1, 4))[0]
            sd["accCW"] = O.point_add(curve, sd["accCW"], O.point_mul(curve, r, sd["inC"]))
            sd["r"] = r
    return hashlib.sha256(b"".join(np.ascontiguousarray(x, dtype=np.uint64).tobytes() for sd in sides for x in
                                   (sd["accCW"], sd["accCE"], sd["inC"], sd["r"]))).hexdigest()

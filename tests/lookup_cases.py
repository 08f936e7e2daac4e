"""Lookup-argument cases shared by the GPU parity tests and the CPU emulator tests.

The reference's only exercise of this path is nifs::sangria::tests::three_rounds_test
(src/nifs/sangria/tests.rs:313-346): FiboCircuitWithLookup (src/nifs/tests.rs:232-420) = 3 advice columns, selectors
s_add / s_xor, a 3-column xor table and ONE vector lookup (s_xor*a, s_xor*b, s_xor*c) in (t0, t1, t2); two satisfying
traces are folded into a fresh accumulator and the decider must accept.  halo2 synthesis is not available here, so the
same constraint-system shape is stated directly as Sirius expressions and the traces are written by hand.

  "vector": that circuit (3 prover rounds, challenges r1 r2 r3)
  "scalar": a single-column range lookup s_rc*a in t0 (2 prover rounds, r1 r2)
  "two"   : two single-column lookups -- exercises the literal index_map of src/plonk/eval.rs:169-201, whose
            (l,t,m)/(h,g) addressing is per-lookup interleaved while run_sps_protocol_* concatenates grouped (quirk Q5);
            here only product == oracle is asserted, not satisfiability.
"""
import numpy as np

from oracle import expr as OE
from oracle import lookup as OL
from oracle import pyref as P


def _shape(variant):
    """-> num_selectors, num_fixed, num_advice, custom gates, lookups (oracle expression tuples)."""
    if variant == "vector":
        ns, nf, na = 2, 3, 3
        s_add, s_xor = OE.Poly(0), OE.Poly(1)
        t = [OE.Poly(ns + i) for i in range(3)]
        a, b, c = [OE.Poly(ns + nf + i) for i in range(3)]
        gates = [OE.Prod(s_add, OE.Sum(OE.Sum(a, b), OE.Neg(c)))]
        lookups = [([OE.Prod(s_xor, a), OE.Prod(s_xor, b), OE.Prod(s_xor, c)], t)]
    elif variant == "scalar":
        ns, nf, na = 2, 1, 3
        s_add, s_rc = OE.Poly(0), OE.Poly(1)
        a, b, c = [OE.Poly(ns + nf + i) for i in range(3)]
        gates = [OE.Prod(s_add, OE.Sum(OE.Sum(a, b), OE.Neg(c)))]
        lookups = [([OE.Prod(s_rc, a)], [OE.Poly(ns)])]
    else:
        ns, nf, na = 2, 2, 3
        s_add, s_rc = OE.Poly(0), OE.Poly(1)
        a, b, c = [OE.Poly(ns + nf + i) for i in range(3)]
        gates = []          # 8 lookup expressions already reach the supported folding degree (y^7 * degree 1)
        lookups = [([OE.Prod(s_rc, a)], [OE.Poly(ns)]), ([OE.Prod(s_rc, b)], [OE.Poly(ns + 1)])]
    return ns, nf, na, gates, lookups


def _circuit(O, field, variant, k, rng):
    """selectors, fixed (table) columns and a satisfying advice assignment generator."""
    p = P.MODULI[field]
    rows = 1 << k
    q = rows // 4
    s_add = np.zeros(rows, np.uint8); s_add[:q] = 1
    s_lk = np.zeros(rows, np.uint8); s_lk[q:2 * q] = 1
    if variant == "vector":
        tab = [(x, y, x ^ y) for x in range(4) for y in range(4)][:rows]
        cols = [[e[i] for e in tab] + [0] * (rows - len(tab)) for i in range(3)]     # padding rows repeat (0,0,0)
    elif variant == "scalar":
        cols = [[v % 8 for v in range(rows)]]                                          # every value repeated rows/8 times
    else:
        cols = [[v % 8 for v in range(rows)], list(range(min(rows, 16))) + [0] * max(0, rows - 16)]
    fixed = [O.ints_to_mont(field, c) for c in cols]

    def advice():
        a = [int(rng.integers(0, 1 << 62)) for _ in range(rows)]
        b = [int(rng.integers(0, 1 << 62)) for _ in range(rows)]
        c = [int(rng.integers(0, 1 << 62)) for _ in range(rows)]
        for r in range(q):
            c[r] = (a[r] + b[r]) % p
        for r in range(q, 2 * q):
            if variant == "vector":
                a[r], b[r] = int(rng.integers(0, 4)), int(rng.integers(0, 4))
                c[r] = a[r] ^ b[r]
            else:
                a[r], b[r] = int(rng.integers(0, 8)), int(rng.integers(0, 16))
        return [O.ints_to_mont(field, col) for col in (a, b, c)]
    return [s_add, s_lk], fixed, advice


def _to_product_expr(X, e):
    k = e[0]
    if k == 'const': return X.Constant(e[1])
    if k == 'poly': return X.Polynomial(e[1], e[2])
    if k == 'chal': return X.Challenge(e[1])
    if k == 'neg': return X.Negated(_to_product_expr(X, e[1]))
    if k == 'sum': return X.Sum(_to_product_expr(X, e[1]), _to_product_expr(X, e[2]))
    if k == 'prod': return X.Product(_to_product_expr(X, e[1]), _to_product_expr(X, e[2]))
    return X.Scaled(_to_product_expr(X, e[1]), e[2])


def _rand_fe(O, field, rng, n):
    return O.ints_to_mont(field, [int(rng.integers(0, 1 << 62)) * int(rng.integers(1, 1 << 62)) + 1 for _ in range(n)])


def _sps_product(S, St, advice_cols, ch):
    """run_sps_protocol_2 / _3 (src/plonk/mod.rs:503-672) with the product's kernels; challenges supplied."""
    cat = lambda cols: np.ascontiguousarray(np.concatenate(cols))
    adv = cat(advice_cols)
    zero = np.zeros(4, np.uint64)
    if St.has_vector_lookup:
        ls, ts, ms = St.lookup_coeff_1(adv, ch[0])
        hs, gs = St.lookup_coeff_2(ls, ts, ms, ch[1])
        return [adv, cat(ls + ts + ms), cat(hs + gs)]
    ls, ts, ms = St.lookup_coeff_1(adv, zero)
    hs, gs = St.lookup_coeff_2(ls, ts, ms, ch[0])
    return [cat(list(advice_cols) + ls + ts + ms), cat(hs + gs)]


def run_lookup_case(S, O, variant, k, field=0):
    X = S.expression
    p = P.MODULI[field]
    rows = 1 << k
    rng = np.random.default_rng(1000 + k + len(variant))
    ns, nf, na, ogates, olookups = _shape(variant)
    selectors, fixed, advice = _circuit(O, field, variant, k, rng)
    meta = OL.build_metainfo(k, ns, nf, na, ogates, olookups)
    St = S.PlonkStructure(field, k, selectors, fixed, na, [_to_product_expr(X, g) for g in ogates],
                          lookups=[([_to_product_expr(X, e) for e in i], [_to_product_expr(X, e) for e in t]) for i, t in olookups])
    # ---- structure metadata (constraint_system_metainfo.rs:27-104)
    assert St.num_lookups == meta.num_lookups and St.has_vector_lookup == meta.has_vector_lookup
    assert St.round_sizes == meta.round_sizes and St.num_challenges == meta.num_challenges
    assert St.num_cross_terms == len(meta.compressed.grouped()) - 1
    assert St.num_witness_columns == sum(meta.round_sizes) // rows
    nch = St.num_challenges
    traces = []
    for _ in range(2):
        cols = advice()
        ch = _rand_fe(O, field, rng, nch)
        ch_int = O.mont_to_ints(field, ch)
        W = _sps_product(S, St, cols, ch)
        We = OL.run_sps_witness(O, field, meta, selectors, fixed, cols, ch_int, p)
        assert len(W) == len(We)
        for a, b in zip(W, We):
            assert np.array_equal(a, b)
        traces.append((W, ch))
    # m is not trivial: some table rows are hit several times, repeats of a table value carry 0
    m_ints = O.mont_to_ints(field, traces[0][0][1 if meta.has_vector_lookup else 0].reshape(-1, 4)
                            [(2 * meta.num_lookups if meta.has_vector_lookup else na + 2 * meta.num_lookups) * rows:][:rows])
    assert max(m_ints) > 1 and sum(m_ints) == rows
    # ---- decider on the fresh traces: gates + log-derivative (PlonkStructure::is_sat, plonk/mod.rs:304-361)
    satisfiable = meta.num_lookups == 1
    prog_c = OE.GraphEvaluator(meta.compressed.compressed, p).export(field, O)
    for W, ch in traces:
        vals = O.eval_program(field, prog_c, selectors, fixed, W, None, ch, num_advice=na, num_lookup=meta.num_lookups)
        assert np.array_equal(St.eval_gates(W, ch), vals)
        bad = int(np.count_nonzero(np.any(vals != 0, axis=1)))
        assert St.is_sat_gates(W, ch) == bad
        assert (St.is_sat_log_derivative(W) == 0) == OL.is_sat_log_derivative(O, field, meta, W, rows, p)
        if satisfiable:
            assert bad == 0 and St.is_sat_log_derivative(W) == 0
    if satisfiable:      # a wrong multiplicity breaks the log-derivative sum (and the g-gate on that row)
        W, ch = traces[0]
        Wb = [w.copy() for w in W]
        Wb[-1][3] = O.ints_to_mont(field, [12345])[0]
        assert St.is_sat_log_derivative(Wb) == 1 and St.is_sat_gates(Wb, ch) >= 1
    # ---- fold both traces into a fresh accumulator (sangria/tests.rs:151-252) and decide after each fold
    one = O.ints_to_mont(field, [1])[0]
    acc_W = [np.zeros((sz, 4), np.uint64) for sz in meta.round_sizes]
    acc_E = np.zeros((rows, 4), np.uint64)
    acc_ch, acc_u = np.zeros((nch, 4), np.uint64), np.zeros(4, np.uint64)
    prog_h = OE.GraphEvaluator(meta.compressed.homogeneous, p).export(field, O)
    for W, ch in traces:
        terms, _ = S.VanillaFS.commit_cross_terms(None, St, acc_ch, acc_u, acc_W, ch, W)
        chx = S.VanillaFS.cross_term_challenges(acc_ch, acc_u, ch, field)
        exp = OL.cross_terms_oracle(O, field, meta, ns, nf, na, selectors, fixed, acc_W, W, chx)
        assert len(terms) == len(exp) == St.num_cross_terms
        for a, b in zip(terms, exp):
            assert np.array_equal(a, b)
        r = _rand_fe(O, field, rng, 1)[0]
        acc = S.RelaxedPlonkWitness(field, acc_W, acc_E).fold(W, terms, r)
        rb = lambda x: np.broadcast_to(r, x.shape).copy()
        acc_ch = O.fe_add(field, acc_ch, O.fe_mul(field, rb(ch), ch))
        acc_u = O.fe_add(field, acc_u.reshape(1, 4), O.fe_mul(field, r.reshape(1, 4), one.reshape(1, 4)))[0]
        acc_W, acc_E = acc.W, acc.E
        chh = np.concatenate([acc_ch, acc_u.reshape(1, 4)])
        hom = O.eval_program(field, prog_h, selectors, fixed, acc_W, None, chh, num_advice=na, num_lookup=meta.num_lookups)
        assert np.array_equal(St.eval_gates(acc_W, chh, homogeneous=True), hom)
        mism = int(np.count_nonzero(np.any(hom != acc_E, axis=1)))
        assert St.is_sat_gates(acc_W, chh, acc_E) == mism
        if satisfiable:   # is_sat_accumulation + is_sat_log_derivative hold on the folded trace (three_rounds_test)
            assert mism == 0 and St.is_sat_log_derivative(acc_W) == 0
    St.close()

"""CPU: pins the oracle's restatement of Expression / GroupedPoly / GraphEvaluator to the reference's
string KATs (src/main_gate.rs:893-927) and to the GraphEvaluator property test
(src/polynomial/graph_evaluator.rs:447-634: random values vs closed form)."""
import numpy as np

from oracle import expr as E
from oracle import pyref as P


def test_main_gate_expr_string():          # src/main_gate.rs:893-908
    g = E.main_gate_expression(2)
    assert E.visualize(g) == (
        "Z_4 * Z_9 * Z_10 + Z_6 * Z_11 + Z_8 + Z_7 * Z_12 + Z_0 * Z_9 + Z_2 * Z_9 * Z_9 * Z_9 * Z_9 * Z_9 + "
        "Z_1 * Z_10 + Z_3 * Z_10 * Z_10 * Z_10 * Z_10 * Z_10")


def test_main_gate_cross_term_strings():   # src/main_gate.rs:910-927
    ctx = E.QueryIndexContext(0, 9, 4, 0, 0)
    cg = E.CompressedGates.new([E.main_gate_expression(2)], ctx)
    gp = cg.grouped()
    assert E.visualize(gp.get(0)) == (
        "r_0 * r_0 * r_0 * (Z_10 * Z_9 * Z_4 + r_0 * Z_11 * Z_6 + r_0 * r_0 * Z_8 + r_0 * Z_12 * Z_7) + "
        "r_0 * r_0 * r_0 * r_0 * Z_9 * Z_0 + Z_9 * Z_9 * Z_9 * Z_9 * Z_9 * Z_2 + r_0 * r_0 * r_0 * r_0 * Z_10 * Z_1 + "
        "Z_10 * Z_10 * Z_10 * Z_10 * Z_10 * Z_3")
    assert E.visualize(gp.get(5)) == (
        "r_1 * r_1 * r_1 * (Z_14 * Z_13 * Z_4 + r_1 * Z_15 * Z_6 + r_1 * r_1 * Z_8 + r_1 * Z_16 * Z_7) + "
        "r_1 * r_1 * r_1 * r_1 * Z_13 * Z_0 + Z_13 * Z_13 * Z_13 * Z_13 * Z_13 * Z_2 + r_1 * r_1 * r_1 * r_1 * Z_14 * Z_1 + "
        "Z_14 * Z_14 * Z_14 * Z_14 * Z_14 * Z_3")
    assert len(gp) == 6 and cg.degree == 5


def test_grouped_poly_doc_example():       # src/polynomial/grouped_poly.rs:76-87: (a+b+c)(d+e) has 3 terms
    ctx = E.QueryIndexContext(0, 0, 5, 0, 0)
    a, b, c, d, e = (E.Poly(i) for i in range(5))
    gp = E.GroupedPoly.new(E.Prod(E.Sum(E.Sum(a, b), c), E.Sum(d, e)), ctx)
    assert len(gp) == 3


def _closed_form(e, p, getp, getc):
    k = e[0]
    if k == 'const': return e[1] % p
    if k == 'poly': return getp(e[1], e[2])
    if k == 'chal': return getc(e[1])
    if k == 'neg': return (-_closed_form(e[1], p, getp, getc)) % p
    if k == 'sum': return (_closed_form(e[1], p, getp, getc) + _closed_form(e[2], p, getp, getc)) % p
    if k == 'prod': return _closed_form(e[1], p, getp, getc) * _closed_form(e[2], p, getp, getc) % p
    return _closed_form(e[1], p, getp, getc) * e[2] % p


def test_graph_evaluator_vs_closed_form(oracle):
    """Compiled calculation list interpreted by oracle.c == direct recursive evaluation, incl. rotations,
    negated constants, scaling by 0/1/2, x*x, x+(-y) (graph_evaluator.rs:261-351 branches)."""
    O = oracle
    import random
    rnd = random.Random(11)
    for field in (0, 1):
        p = P.MODULI[field]
        rows, nsel, nfix, nadv = 8, 1, 2, 3
        X = [E.Poly(i, r) for i in range(nsel + nfix + 2 * nadv) for r in (0, 1, -1)]
        ex = E.Sum(E.Prod(X[0], E.Sum(X[4], E.Neg(X[9]))), E.Scaled(E.Prod(X[10], X[10]), 7))
        ex = E.Sum(ex, E.Neg(E.Const(5)))
        ex = E.Sum(E.Prod(ex, E.Chal(1)), E.Prod(E.Const(2), E.Sum(X[13], E.Prod(E.Const(1), X[20]))))
        ex = E.Sum(ex, E.Prod(E.Scaled(X[26], 0), X[3]))
        ex = E.Sum(E.Neg(E.Prod(X[17], E.Chal(0))), E.Prod(ex, X[22]))
        sel = [np.array([rnd.randrange(2) for _ in range(rows)], dtype=np.uint8)]
        fixv = [[rnd.randrange(p) for _ in range(rows)] for _ in range(nfix)]
        w1 = [rnd.randrange(p) for _ in range(nadv * rows)]
        w2 = [rnd.randrange(p) for _ in range(nadv * rows)]
        ch = [rnd.randrange(p) for _ in range(2)]
        prog = E.GraphEvaluator(ex, p).export(field, O)
        got = O.mont_to_ints(field, O.eval_program(field, prog, sel, [O.ints_to_mont(field, f) for f in fixv],
                                                    O.ints_to_mont(field, w1), O.ints_to_mont(field, w2), O.ints_to_mont(field, ch)))
        for row in range(rows):
            def getp(i, rot):
                r = (row + rot) % rows
                if i < nsel: return int(sel[i][r])
                i -= nsel
                if i < nfix: return fixv[i][r]
                i -= nfix
                return w1[i * rows + r] if i < nadv else w2[(i - nadv) * rows + r]
            assert got[row] == _closed_form(ex, p, getp, lambda i: ch[i]), (field, row)


def test_cross_terms_are_coefficients(oracle):
    """Definition check of the oracle's cross terms: sum_k T_k x^k + T_0 == P_hom(W1 + x W2, ch1 + x ch2)
    for a random x (pure big-int evaluation of the homogeneous expression)."""
    O = oracle
    import random
    rnd = random.Random(5)
    field, k = 0, 2
    p = P.MODULI[field]
    rows = 1 << k
    gates = [E.main_gate_expression(2)]
    nfix, nadv = 9, 4
    fixv = [[rnd.randrange(p) for _ in range(rows)] for _ in range(nfix)]
    w1 = [rnd.randrange(p) for _ in range(nadv * rows)]
    w2 = [rnd.randrange(p) for _ in range(nadv * rows)]
    ch = [rnd.randrange(p), 1]           # U1.u, DEFAULT_u  (no structure challenges: single gate)
    cg, T = E.cross_terms_oracle(O, field, gates, 0, nfix, nadv, [], [O.ints_to_mont(field, f) for f in fixv],
                                 O.ints_to_mont(field, w1), O.ints_to_mont(field, w2), O.ints_to_mont(field, ch))
    Ti = [O.mont_to_ints(field, t) for t in T]
    x = rnd.randrange(p)
    for row in range(rows):
        def at(xv):
            def getp(i, rot):
                r = (row + rot) % rows
                if i < nfix: return fixv[i][r]
                i -= nfix
                return (w1[i * rows + r] + xv * w2[i * rows + r]) % p
            return _closed_form(cg.homogeneous, p, getp, lambda i: (ch[0] + xv * ch[1]) % p)
        lhs = at(x)
        rhs = (at(0) + sum(Ti[kk][row] * pow(x, kk + 1, p) for kk in range(cg.degree))) % p
        assert lhs == rhs

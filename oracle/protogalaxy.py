"""Literal restatement of the reference's ProtoGalaxy prover polynomials.  TEST INFRASTRUCTURE ONLY.

  PolyContext, compute_F, compute_G, compute_K_from_G, BetaStrokeIter   src/nifs/protogalaxy/poly/mod.rs:68-545
  FoldedWitness (fold_witnesses, fold_plonk_challenges)                src/nifs/protogalaxy/poly/folded_witness.rs:20-180
  get_evaluate_witness_fn (incl. the `index & total_row` quirk Q1)      src/plonk/mod.rs:683-718
  evaluate_e_from_trace, calculate_e, fold_witness, new_accumulator betas (Q3)   src/nifs/protogalaxy/mod.rs:144-210,571-640,748-764
Values are python ints (canonical, mod Fr); leaves are evaluated with the GraphEvaluator restatement
(oracle/expr.py) interpreted by oracle.c.  Sizes here are test-sized (n <= 2^14 or so).
"""
import numpy as np

from . import expr as E
from . import pyref as P

FR = P.FR


def next_pow2(v):
    p = 1
    while p < v:
        p <<= 1
    return p


class PolyContext:                                   # poly/mod.rs:205-269
    def __init__(self, gates, k, num_selectors, num_fixed, num_advice, num_challenges, traces_len):
        count = (1 << k) * len(gates)                # get_count_of_valuation :511-516
        self.count_with_padding = next_pow2(count)   # :518-533
        self.instances_to_fold = traces_len + 1
        assert self.instances_to_fold & (self.instances_to_fold - 1) == 0
        ctx = E.QueryIndexContext(num_selectors, num_fixed, num_advice, num_challenges, 0)
        max_degree = max([degree(g, ctx) for g in gates] or [0])     # get_points_count :535-545
        self.fft_points_count_G = next_pow2(traces_len * max_degree + 1)

    def betas_count(self): return self.count_with_padding.bit_length() - 1
    def fft_points_count_F(self): return next_pow2(self.betas_count() + 1)
    def lagrange_domain(self): return self.instances_to_fold.bit_length() - 1
    def fft_log_domain_size_G(self): return self.fft_points_count_G.bit_length() - 1

    def fft_log_domain_size_K(self):                 # :263-268 (Q2: a count used as a log)
        c = self.fft_points_count_G + 1
        c = c - self.instances_to_fold if c > self.instances_to_fold else 0
        return next_pow2(c)


def degree(e, ctx):                                  # Expression::degree, expression.rs:431-447
    k = e[0]
    if k == 'const': return 0
    if k == 'poly': return 1 if ctx.subtype(e[1]) in ('advice', 'lookup') else 0
    if k == 'chal': return 1
    if k in ('neg', 'scaled'): return degree(e[1], ctx)
    if k == 'sum': return max(degree(e[1], ctx), degree(e[2], ctx))
    return degree(e[1], ctx) + degree(e[2], ctx)


class Structure:
    """The slice of PlonkStructure the polynomials need."""

    def __init__(self, O, gates, k, selectors, fixed, num_advice, num_challenges=0):
        self.O, self.gates, self.k, self.rows = O, gates, k, 1 << k
        self.selectors, self.fixed, self.num_advice, self.num_challenges = selectors, fixed, num_advice, num_challenges
        self.progs = [E.GraphEvaluator(g, FR).export(O.FR, O) for g in gates]      # plonk/mod.rs:697-701

    def context(self, traces_len):
        return PolyContext(self.gates, self.k, len(self.selectors), len(self.fixed), self.num_advice, self.num_challenges, traces_len)

    def evaluate_witness_fn(self, W_mont, challenges_ints, compat=True):
        """get_evaluate_witness_fn: index -> gates[index / 2^k] at row (index & 2^k) [Q1] (or index % 2^k)."""
        O = self.O
        ch = O.ints_to_mont(O.FR, list(challenges_ints)) if len(challenges_ints) else np.zeros((0, 4), np.uint64)
        tables = [O.mont_to_ints(O.FR, O.eval_program(O.FR, pr, self.selectors, self.fixed, W_mont, W_mont, ch)) for pr in self.progs]
        total_row, limit = self.rows, len(self.gates) * self.rows

        def f(index):
            if index >= limit:
                return 0
            gate_index = index // total_row
            row_index = (index & total_row) if compat else (index % total_row)
            return tables[gate_index][row_index % total_row]     # row 2^k wraps to 0 (+rot) : graph_evaluator.rs:51-53
        return f


def tree_reduce(leaves, c):
    """itertools tree_reduce over a power-of-two list with the reducers of the reference:
    node = left + right * c[height]  (poly/mod.rs:119-162, :332-360; protogalaxy/mod.rs:585-606)."""
    vals, h = list(leaves), 0
    assert len(vals) & (len(vals) - 1) == 0 and len(vals) >= 1
    while len(vals) > 1:
        vals = [(vals[2 * i] + vals[2 * i + 1] * c[h]) % FR for i in range(len(vals) // 2)]
        h += 1
    return vals[0]


def compute_F(S, ctx, betas, delta, W_mont, challenges, compat=True):          # poly/mod.rs:68-203
    n, t = ctx.count_with_padding, ctx.betas_count()
    betas = list(betas)[:t]
    assert len(betas) == t
    deltas = [delta]
    for _ in range(t - 1):
        deltas.append(deltas[-1] * deltas[-1] % FR)
    f = S.evaluate_witness_fn(W_mont, challenges, compat)
    leaves = [f(i) for i in range(n)]
    points = []
    for X in P.iter_cyclic_subgroup(ctx.fft_points_count_F().bit_length() - 1):
        c = [(b + X * d) % FR for b, d in zip(betas, deltas)]
        points.append(tree_reduce(leaves, c))
    P.ifft(points)
    return points


def folded_witness(S, ctx, points_for_fft, Ws_mont, challenges_list):        # folded_witness.rs:20-180
    O = S.O
    out = []
    Wi = [np.array(O.mont_to_ints(O.FR, w), dtype=object) for w in Ws_mont]
    for X in points_for_fft:
        L = P.eval_lagrange_poly_for_cyclic_group(X, ctx.lagrange_domain())
        w = sum(int(L[j]) * Wi[j] for j in range(len(Wi))) % FR
        ch = [sum(L[j] * challenges_list[j][c] for j in range(len(Wi))) % FR for c in range(len(challenges_list[0]))]
        out.append((O.ints_to_mont(O.FR, [int(v) for v in w]), ch))
    return out


def compute_G(S, ctx, betas_stroke, Ws_mont, challenges_list, compat=True):  # poly/mod.rs:308-425
    assert len(Ws_mont) >= 2, "You can't fold 0 traces"
    n, t = ctx.count_with_padding, ctx.betas_count()
    bs = list(betas_stroke)[:t]
    assert len(bs) == t
    pts = list(P.iter_cyclic_subgroup(ctx.fft_log_domain_size_G()))[:ctx.fft_points_count_G]
    points = []
    for w, ch in folded_witness(S, ctx, pts, Ws_mont, challenges_list):
        f = S.evaluate_witness_fn(w, ch, compat)
        points.append(tree_reduce([f(i) for i in range(n)], bs))
    P.ifft(points)
    return points


def beta_stroke(betas, alpha, delta):                                          # BetaStrokeIter :432-462
    out, d = [], delta
    for b in betas:
        out.append((b + alpha * d) % FR)
        d = d * d % FR
    return out


def poly_eval(coeffs, x):                                                      # univariate.rs:67-75
    acc, xp = 0, 1
    for c in coeffs:
        acc = (acc + xp * c) % FR
        xp = xp * x % FR
    return acc


def compute_K_from_G(ctx, poly_G, poly_F_in_alpha):                            # poly/mod.rs:475-509
    logK = ctx.fft_log_domain_size_K()
    vals = []
    for w in P.iter_cyclic_subgroup(logK):
        X = P.FR_ZETA * w % FR
        g = poly_eval(poly_G, X)
        l0 = P.eval_lagrange_poly_for_cyclic_group(X, ctx.lagrange_domain())[0]
        z = P.eval_vanish_polynomial(ctx.instances_to_fold, X)
        assert z != 0, "Z(X) must be not equal to 0"
        kx = (g - poly_F_in_alpha * l0) * P.inv(z, FR) % FR
        assert (poly_F_in_alpha * l0 + z * kx) % FR == g
        vals.append(kx)
    P.coset_ifft(vals)
    return vals


def evaluate_e_from_trace(S, ctx, betas, W_mont, challenges, compat=True):     # protogalaxy/mod.rs:571-640
    f = S.evaluate_witness_fn(W_mont, challenges, compat)
    return tree_reduce([f(i) for i in range(ctx.count_with_padding)], list(betas))


def calculate_e(poly_F, poly_K, gamma, alpha, log_n):                          # protogalaxy/mod.rs:748-764
    l0 = P.eval_lagrange_poly_for_cyclic_group(gamma, log_n)[0]
    return (poly_eval(poly_F, alpha) * l0 + P.eval_vanish_polynomial(1 << log_n, gamma) * poly_eval(poly_K, gamma)) % FR


def fold_witness(O, Ws_mont, lagrange_for_gamma):                              # protogalaxy/mod.rs:176-210
    acc = None
    for w, l in zip(Ws_mont, lagrange_for_gamma):
        term = O.fe_mul(O.FR, np.broadcast_to(O.ints_to_mont(O.FR, [l])[0], w.shape).copy(), w)
        acc = term if acc is None else O.fe_add(O.FR, acc, term)
    return acc


def new_accumulator_betas(beta, count):                                        # Q3: beta * 2^i (protogalaxy/mod.rs:164-168)
    out, b = [], beta
    for _ in range(count):
        out.append(b)
        b = b * 2 % FR
    return out


# ---- the same polynomials with the leaf tables, witness folds and reduction trees in C (oracle.c: o_eval_program, o_lincomb,
# o_pg_tree) -- identical values (tests/test_oracle_pins.py::test_pg_fast_equals_literal), usable at the bench sizes
# (bench.py's cpu_baseline leg, mid-size GPU parity tests).  `threads` = OpenMP threads.
def _leaf_table(S, W_mont, challenges_ints, compat, threads):
    """(count, 4) Montgomery leaves: leaf i = gates[i / 2^k] at row i % 2^k (compat: row 0, Q1)."""
    O = S.O
    ch = O.ints_to_mont(O.FR, list(challenges_ints)) if len(challenges_ints) else np.zeros((0, 4), np.uint64)
    tabs = [O.eval_program(O.FR, pr, S.selectors, S.fixed, W_mont, W_mont, ch, threads) for pr in S.progs]
    if compat:
        tabs = [np.ascontiguousarray(np.broadcast_to(t[0], t.shape)) for t in tabs]
    return np.ascontiguousarray(np.concatenate(tabs, axis=0))


def compute_F_fast(S, ctx, betas, delta, W_mont, challenges, compat=True, threads=0):
    O = S.O
    n, t = ctx.count_with_padding, ctx.betas_count()
    betas = list(betas)[:t]
    deltas = [delta]
    for _ in range(t - 1):
        deltas.append(deltas[-1] * deltas[-1] % FR)
    leaves = _leaf_table(S, W_mont, challenges, compat, threads)
    Xs = list(P.iter_cyclic_subgroup(ctx.fft_points_count_F().bit_length() - 1))
    weights = np.stack([O.ints_to_mont(O.FR, [(b + X * d) % FR for b, d in zip(betas, deltas)]) for X in Xs])
    points = O.mont_to_ints(O.FR, O.pg_tree(O.FR, leaves, leaves.shape[0], weights, threads))
    P.ifft(points)
    return points


def compute_G_fast(S, ctx, betas_stroke, Ws_mont, challenges_list, compat=True, threads=0):
    O = S.O
    assert len(Ws_mont) >= 2, "You can't fold 0 traces"
    t = ctx.betas_count()
    bs = O.ints_to_mont(O.FR, list(betas_stroke)[:t]).reshape(1, t, 4)
    pts = list(P.iter_cyclic_subgroup(ctx.fft_log_domain_size_G()))[:ctx.fft_points_count_G]
    points = []
    for X in pts:
        L = P.eval_lagrange_poly_for_cyclic_group(X, ctx.lagrange_domain())
        w = O.lincomb(O.FR, Ws_mont, O.ints_to_mont(O.FR, [int(l) for l in L[:len(Ws_mont)]]), threads)
        ch = [sum(L[j] * challenges_list[j][c] for j in range(len(Ws_mont))) % FR for c in range(len(challenges_list[0]))]
        leaves = _leaf_table(S, w, ch, compat, threads)
        points.append(O.mont_to_ints(O.FR, O.pg_tree(O.FR, leaves, leaves.shape[0], bs, threads))[0])
    P.ifft(points)
    return points


def evaluate_e_fast(S, ctx, betas, W_mont, challenges, compat=True, threads=0):
    O = S.O
    t = ctx.betas_count()
    leaves = _leaf_table(S, W_mont, challenges, compat, threads)
    w = O.ints_to_mont(O.FR, list(betas)[:t]).reshape(1, t, 4)
    return O.mont_to_ints(O.FR, O.pg_tree(O.FR, leaves, leaves.shape[0], w, threads))[0]

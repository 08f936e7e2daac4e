"""Literal Python restatement of the reference's lookup-argument host pipeline.  TEST INFRASTRUCTURE ONLY.

  Arguments (compress_from, to_expressions, vanishing_lookup_polys, log_derivative_expr)   src/plonk/lookup.rs:72-206
  evaluate_ls / evaluate_ts / evaluate_m / evaluate_h_g, evaluate_coefficient_1 / _2       src/plonk/lookup.rs:209-365
  ConstraintSystemMetainfo::build (gates ++ lookup expressions, round_sizes, r3 index)     src/table/constraint_system_metainfo.rs:20-108
  run_sps_protocol_2 / _3 witness assembly (challenges supplied by the caller: the random
  oracle is host-side protocol code outside the hot path)                                  src/plonk/mod.rs:503-672
  is_sat_log_derivative                                                                    src/plonk/mod.rs:366-398

Parity: UNPINNED by vectors -- the reference holds no golden values for lookups; its only test of this
path is nifs::sangria::tests::three_rounds_test (fold two satisfying traces of FiboCircuitWithLookup,
then is_sat).  tests/test_lookup_* restate that test's structure and assert the same property.

Expressions are the tuples of oracle/expr.py; field values python ints (canonical).
"""
from dataclasses import dataclass, field as dc_field

import numpy as np

from . import expr as E


@dataclass
class Arguments:                     # lookup.rs:72-82
    lookup_polys: list
    table_polys: list
    has_vector_lookup: bool

    @staticmethod
    def compress_from(lookups):
        """lookups: list of (input_expressions, table_expressions), each a list of Sirius Expressions
        (= halo2 expressions after Expression::from_halo2_expr).  lookup.rs:86-126"""
        lens = [len(inp) for inp, _ in lookups]
        if not lens or max(lens) == 0:
            return None
        has_vec = max(lens) > 1
        # compress_halo2_expression(.., challenge_index = 0)   (src/plonk/util.rs:12-32)
        lp = [E.compress_expression(list(inp), 0) for inp, _ in lookups]
        tp = [E.compress_expression(list(tab), 0) for _, tab in lookups]
        return Arguments(lp, tp, has_vec)

    def num_lookups(self):
        return len(self.lookup_polys)

    def vanishing_lookup_polys(self, ctx):          # lookup.rs:137-166
        off = ctx.num_selectors + ctx.num_fixed + ctx.num_advice
        ls = [E.Sum(L, E.Neg(E.Poly(off + i * 5))) for i, L in enumerate(self.lookup_polys)]
        ts = [E.Sum(T, E.Neg(E.Poly(off + i * 5 + 1))) for i, T in enumerate(self.table_polys)]
        return ls + ts

    def log_derivative_expr(self, ctx, lookup_index, challenge_index):   # lookup.rs:174-195
        r = E.Chal(challenge_index)
        off = ctx.num_selectors + ctx.num_fixed + ctx.num_advice
        l, t, m, h, g = [E.Poly(off + lookup_index * 5 + i) for i in range(5)]
        lhs = E.Sum(E.Prod(h, E.Sum(l, r)), E.Neg(E.Const(1)))
        rhs = E.Sum(E.Prod(g, E.Sum(t, r)), E.Neg(m))
        return lhs, rhs

    def log_derivative_lhs_and_rhs(self, ctx):      # lookup.rs:198-206
        ci = 1 if self.has_vector_lookup else 0
        out = []
        for i in range(self.num_lookups()):
            out.extend(self.log_derivative_expr(ctx, i, ci))
        return out

    def to_expressions(self, ctx):                  # lookup.rs:129-133
        return self.vanishing_lookup_polys(ctx) + self.log_derivative_lhs_and_rhs(ctx)


@dataclass
class Metainfo:                      # ConstraintSystemMetainfo
    num_challenges: int
    round_sizes: list
    gates: list
    compressed: E.CompressedGates
    arguments: object
    num_lookups: int
    has_vector_lookup: bool


def build_metainfo(k, num_selectors, num_fixed, num_advice, custom_gates, lookups):
    """constraint_system_metainfo.rs:20-108"""
    ctx0 = E.QueryIndexContext(num_selectors, num_fixed, num_advice, 0, 0)
    args = Arguments.compress_from(lookups)
    nl = args.num_lookups() if args else 0
    hv = args.has_vector_lookup if args else False
    gates = list(custom_gates) + (args.to_expressions(ctx0) if args else [])
    rows = 1 << k
    if hv:
        round_sizes = [num_advice * rows, 3 * nl * rows, 2 * nl * rows]
    elif nl > 0:
        round_sizes = [(num_advice + 3 * nl) * rows, 2 * nl * rows]
    else:
        round_sizes = [num_advice * rows]
    ctx = E.QueryIndexContext(num_selectors, num_fixed, num_advice, 2 if hv else (1 if nl > 0 else 0), nl)
    cg = E.CompressedGates.new(gates, ctx)
    return Metainfo(cg.num_challenges_compressed, round_sizes, gates, cg, args, nl, hv)


# ---------------------------------------------------------------------------- coefficient evaluation
def _eval_exprs(O, field, exprs, selectors, fixed, advice_cols, r_int, p):
    """evaluate_ls / evaluate_ts (lookup.rs:209-272): LookupEvalDomain = advice COLUMNS only, challenges = [r]."""
    rows = fixed[0].shape[0] if len(fixed) else selectors[0].shape[0]
    W = np.concatenate([np.asarray(c, dtype=np.uint64).reshape(rows, 4) for c in advice_cols]) if advice_cols else np.zeros((0, 4), np.uint64)
    ch = O.ints_to_mont(field, [r_int])
    out = []
    for ex in exprs:
        prog = E.GraphEvaluator(ex, p).export(field, O)
        out.append(O.eval_program(field, prog, selectors, fixed, W, None, ch, num_advice=len(advice_cols)))
    return out


def evaluate_m(l_ints, t_ints):      # lookup.rs:275-303
    counts = {}
    for v in l_ints:
        counts[v] = counts.get(v, 0) + 1
    seen = set()
    out = []
    for v in t_ints:
        if v in seen:
            out.append(0)
        else:
            seen.add(v)
            out.append(counts.get(v, 0))
    return out


def evaluate_h_g(l, t, r, m, p):     # lookup.rs:305-317
    inv = lambda x: pow(x, p - 2, p) if x % p else 0
    h = [inv((li + r) % p) for li in l]
    g = [mi * inv((ti + r) % p) % p for ti, mi in zip(t, m)]
    return h, g


@dataclass
class Coeff1:
    ls: list
    ts: list
    ms: list                         # each: list of (rows,4) Montgomery arrays


def evaluate_coefficient_1(O, field, args, selectors, fixed, advice_cols, r_int, p):   # lookup.rs:319-341
    ls = _eval_exprs(O, field, args.lookup_polys, selectors, fixed, advice_cols, r_int, p)
    ts = _eval_exprs(O, field, args.table_polys, selectors, fixed, advice_cols, r_int, p)
    ms = [O.ints_to_mont(field, evaluate_m(O.mont_to_ints(field, l), O.mont_to_ints(field, t))) for l, t in zip(ls, ts)]
    return Coeff1(ls, ts, ms)


def evaluate_coefficient_2(O, field, c1, r_int, p):                                    # lookup.rs:350-365
    hs, gs = [], []
    for l, t, m in zip(c1.ls, c1.ts, c1.ms):
        h, g = evaluate_h_g(O.mont_to_ints(field, l), O.mont_to_ints(field, t), r_int, O.mont_to_ints(field, m), p)
        hs.append(O.ints_to_mont(field, h))
        gs.append(O.ints_to_mont(field, g))
    return hs, gs


def _concat(cols):
    return np.concatenate([np.asarray(c, dtype=np.uint64).reshape(-1, 4) for c in cols])


def run_sps_witness(O, field, meta, selectors, fixed, advice_cols, challenges_int, p):
    """Witness rounds of run_sps_protocol_2 / _3 (plonk/mod.rs:503-672) for given challenges
    (r1, r2[, r3]); commitments and the random oracle are the caller's business."""
    if meta.num_lookups == 0:
        return [_concat(advice_cols)]
    if not meta.has_vector_lookup:                 # _2: r1 = challenges[0], lookup polys evaluated with r = 0
        c1 = evaluate_coefficient_1(O, field, meta.arguments, selectors, fixed, advice_cols, 0, p)
        W1 = _concat(list(advice_cols) + c1.ls + c1.ts + c1.ms)
        hs, gs = evaluate_coefficient_2(O, field, c1, challenges_int[0], p)
        return [W1, _concat(hs + gs)]
    W1 = _concat(advice_cols)                       # _3
    c1 = evaluate_coefficient_1(O, field, meta.arguments, selectors, fixed, advice_cols, challenges_int[0], p)
    W2 = _concat(c1.ls + c1.ts + c1.ms)
    hs, gs = evaluate_coefficient_2(O, field, c1, challenges_int[1], p)
    return [W1, W2, _concat(hs + gs)]


def is_sat_log_derivative(O, field, meta, W, rows, p):   # plonk/mod.rs:366-398
    if meta.num_lookups == 0:
        return True
    Wl = W[2] if meta.has_vector_lookup else W[1]
    vals = O.mont_to_ints(field, Wl)

    def gather(start):
        idxs = [start + 2 * i for i in range(meta.num_lookups)]
        return [vals[i * rows:(i + 1) * rows] for i in idxs]
    hs, gs = gather(0), gather(1)
    return all(sum((a - b) for a, b in zip(h, g)) % p == 0 for h, g in zip(hs, gs))


def cross_terms_oracle(O, field, meta, num_selectors, num_fixed, num_advice, selectors, fixed, W1s, W2s, challenges, threads=0):
    """commit_cross_terms evaluation half with lookups (src/nifs/sangria/mod.rs:102-148)."""
    from . import pyref as P
    p = P.MODULI[field]
    rows = fixed[0].shape[0] if len(fixed) else selectors[0].shape[0]
    out = []
    for term in meta.compressed.grouped().iter_from_first():
        if term is None:
            out.append(np.zeros((rows, 4), dtype=np.uint64))
            continue
        prog = E.GraphEvaluator(term, p).export(field, O)
        out.append(O.eval_program(field, prog, selectors, fixed, W1s, W2s, challenges, threads,
                                  num_advice=num_advice, num_lookup=meta.num_lookups))
    return out

"""`Expression` -- host-side mirror of reference src/polynomial/expression.rs:112-120.

Only construction and serialisation live here: compression, homogenisation and compilation to the
device row program happen inside the library (csrc/rowprog.hip), as they do inside
`PlonkStructure` in the reference.  Nodes are tuples:
  ('const', int) ('poly', index, rotation) ('chal', index) ('neg', a) ('sum', a, b) ('prod', a, b) ('scaled', a, int)
"""
import numpy as np

from .field import to_mont

EX_CONST, EX_POLY, EX_CHALLENGE, EX_NEG, EX_SUM, EX_PRODUCT, EX_SCALED, EX_END = range(8)


def Constant(v): return ('const', v)
def Polynomial(index, rotation=0): return ('poly', index, rotation)
def Challenge(i): return ('chal', i)
def Negated(a): return ('neg', a)
def Sum(a, b): return ('sum', a, b)
def Product(a, b): return ('prod', a, b)
def Scaled(a, k): return ('scaled', a, k)


def serialize_gates(exprs, field):
    """Vec<Expression<F>> -> postfix u64 stream (SRS_EX_*, include/sirius_amd.h)."""
    words = []

    def walk(e):
        k = e[0]
        if k == 'const':
            words.append(EX_CONST); words.extend(int(x) for x in to_mont(field, e[1]))
        elif k == 'poly':
            words.extend([EX_POLY, e[1], e[2] & 0xFFFFFFFFFFFFFFFF])
        elif k == 'chal':
            words.extend([EX_CHALLENGE, e[1]])
        elif k == 'neg':
            walk(e[1]); words.append(EX_NEG)
        elif k == 'sum':
            walk(e[1]); walk(e[2]); words.append(EX_SUM)
        elif k == 'prod':
            walk(e[1]); walk(e[2]); words.append(EX_PRODUCT)
        elif k == 'scaled':
            walk(e[1]); words.append(EX_SCALED); words.extend(int(x) for x in to_mont(field, e[2]))
        else:
            raise ValueError(f"unknown expression node {k}")
    for ex in exprs:
        walk(ex)
        words.append(EX_END)
    return np.array(words, dtype=np.uint64)


def compress_expression(exprs, challenge_index):
    """src/plonk/util.rs:34-56: fold(0, |acc, e| e + acc * y), y = Challenge(challenge_index); a single expression is kept."""
    y = Challenge(challenge_index)
    if len(exprs) > 1:
        acc = Constant(0)
        for ex in exprs:
            acc = Sum(ex, Product(acc, y))
        return acc
    return exprs[0] if exprs else Constant(0)


class LookupArguments:
    """`plonk::lookup::Arguments` (src/plonk/lookup.rs:72-206): the expression-building half of the log-derivative
    lookup argument.  `lookups` = [(input_expressions, table_expressions), ...] as Sirius expressions
    (halo2 expressions after Expression::from_halo2_expr)."""

    def __init__(self, lookups):
        lens = [len(inp) for inp, _ in lookups]
        if not lens or max(lens) == 0:
            raise ValueError("no lookup arguments")            # compress_from returns None (:87-93)
        self.has_vector_lookup = max(lens) > 1
        # vector lookups are compressed with r1 = Challenge(0)  (:97-121, compress_halo2_expression util.rs:12-32)
        self.lookup_polys = [compress_expression(list(inp), 0) for inp, _ in lookups]
        self.table_polys = [compress_expression(list(tab), 0) for _, tab in lookups]

    @property
    def num_lookups(self):
        return len(self.lookup_polys)

    def to_expressions(self, num_selectors, num_fixed, num_advice):
        """vanishing_lookup_polys ++ log_derivative_lhs_and_rhs (:129-206); fold variables (l,t,m,h,g) of lookup i
        are query indices offset + 5 i + {0..4}."""
        off = num_selectors + num_fixed + num_advice
        var = lambda i, j: Polynomial(off + 5 * i + j)
        out = [Sum(L, Negated(var(i, 0))) for i, L in enumerate(self.lookup_polys)]
        out += [Sum(T, Negated(var(i, 1))) for i, T in enumerate(self.table_polys)]
        r = Challenge(1 if self.has_vector_lookup else 0)
        for i in range(self.num_lookups):
            l, t, m, h, g = [var(i, j) for j in range(5)]
            out.append(Sum(Product(h, Sum(l, r)), Negated(Constant(1))))       # h (l + r) - 1
            out.append(Sum(Product(g, Sum(t, r)), Negated(m)))                 # g (t + r) - m
        return out


def main_gate(T, num_selectors=0, fixed_offset=0, advice_offset=0, num_fixed_total=None):
    """The polynomial of `MainGate<T>::configure` (reference src/main_gate.rs:535-583):
      q_m[0]*s[0]*s[1] + q_m[1]*s[2]*s[3] + sum_i q_1[i]*s[i] + sum_i q_5[i]*s[i]^5 + rc + q_i*input + q_o*out
    with the column -> query index map of Expression::from_halo2_expr (src/polynomial/expression.rs:301-336).
    T+2 advice columns (state, input, out), 2T+5 fixed columns (q_1, q_5, q_m[2], q_i, q_o, rc)."""
    nf = 2 * T + 5
    if num_fixed_total is None:
        num_fixed_total = nf
    F = lambda i: Polynomial(num_selectors + fixed_offset + i)
    A = lambda i: Polynomial(num_selectors + num_fixed_total + advice_offset + i)
    state = [A(i) for i in range(T)]
    inp, out = A(T), A(T + 1)
    q_1 = [F(i) for i in range(T)]
    q_5 = [F(T + i) for i in range(T)]
    q_m = [F(2 * T), F(2 * T + 1)]
    q_i, q_o, rc = F(2 * T + 2), F(2 * T + 3), F(2 * T + 4)

    def pow_5(v):
        v2 = Product(v, v)
        return Product(Product(v2, v2), v)
    acc = Sum(Sum(Sum(Product(Product(q_m[0], state[0]), state[1]), Product(q_i, inp)), rc), Product(q_o, out))
    if T >= 4:
        acc = Sum(Product(Product(q_m[1], state[2]), state[3]), acc)
    for s, q1, q5 in zip(state, q_1, q_5):
        acc = Sum(acc, Sum(Product(q1, s), Product(q5, pow_5(s))))
    return acc

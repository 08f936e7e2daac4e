"""Literal Python restatement of the reference's host-side expression pipeline.  TEST INFRASTRUCTURE ONLY.

  Expression / homogeneous / degree / visualize   src/polynomial/expression.rs:112-120,244-285,356-447,501-513
  compress_expression                             src/plonk/util.rs:34-56
  CompressedGates::new                            src/plonk/mod.rs:84-107
  GroupedPoly (new, Add, Mul, Neg, Mul<&F>)       src/polynomial/grouped_poly.rs:88-282
  GraphEvaluator (compile)                        src/polynomial/graph_evaluator.rs:166-351
  MainGate<T> gate                                src/main_gate.rs:535-583

Pinned by the reference's own string KATs (src/main_gate.rs:893-927): see tests/test_oracle_expr.py.
The per-row interpretation of the compiled calculation list is done by oracle/oracle.c
(o_eval_program).  Field values are canonical python ints mod p.
"""
from dataclasses import dataclass

# ---------------------------------------------------------------------------- Expression
# nodes are tuples: ('const', v) ('poly', index, rot) ('chal', i) ('neg', a) ('sum', a, b) ('prod', a, b) ('scaled', a, k)


def Const(v): return ('const', v)
def Poly(index, rot=0): return ('poly', index, rot)
def Chal(i): return ('chal', i)
def Neg(a): return ('neg', a)
def Sum(a, b): return ('sum', a, b)
def Prod(a, b): return ('prod', a, b)
def Scaled(a, k): return ('scaled', a, k)


@dataclass
class QueryIndexContext:           # expression.rs:39-46
    num_selectors: int
    num_fixed: int
    num_advice: int
    num_challenges: int
    num_lookups: int = 0

    def num_fold_vars(self):
        return self.num_advice + self.num_lookups * 5

    def subtype(self, index):      # Query::subtype, expression.rs:86-100
        if index < self.num_selectors:
            return 'selector'
        if index < self.num_selectors + self.num_fixed:
            return 'fixed'
        if index < self.num_selectors + self.num_fixed + self.num_advice:
            return 'advice'
        if index < self.num_selectors + self.num_fixed + self.num_advice + 5 * self.num_lookups:
            return 'lookup'
        raise AssertionError(f"unknown index {index}")


def visualize(e):                  # expression.rs:244-285
    k = e[0]
    if k == 'const':
        return "0x" + format(e[1], 'x').lstrip('0') if e[1] else "0x"
    if k == 'poly':
        rot = "" if e[2] == 0 else (f"[{e[2]}]" if e[2] < 0 else f"[+{e[2]}]")
        return f"Z_{e[1]}{rot}"
    if k == 'chal':
        return f"r_{e[1]}"
    if k == 'neg':
        return "-" + visualize(e[1])
    if k == 'sum':
        if e[2][0] == 'neg':
            return f"{visualize(e[1])} - {visualize(e[2][1])}"
        return f"{visualize(e[1])} + {visualize(e[2])}"
    if k == 'prod':
        l = f"({visualize(e[1])})" if e[1][0] == 'sum' else visualize(e[1])
        r = f"({visualize(e[2])})" if e[2][0] == 'sum' else visualize(e[2])
        return f"{l} * {r}"
    if k == 'scaled':                  # format!("{:?} * {}", trim_leading_zeros(..), a): the constant is a quoted String
        c = "0x" + format(e[2], 'x').lstrip('0') if e[2] else "0x"
        return f'"{c}" * {visualize(e[1])}'
    raise NotImplementedError(k)


def collect_challenges(e, s):
    k = e[0]
    if k == 'chal':
        s.add(e[1])
    elif k in ('neg', 'scaled'):
        collect_challenges(e[1], s)
    elif k in ('sum', 'prod'):
        collect_challenges(e[1], s)
        collect_challenges(e[2], s)


def num_challenges(e):             # expression.rs:163-167
    s = set()
    collect_challenges(e, s)
    return len(s)


def challenge_in_degree(idx, degree):   # expression.rs:501-513
    r = Chal(idx)
    for _ in range(2, degree + 1):
        r = Prod(r, Chal(idx))
    return r


def homogeneous(e, ctx):           # expression.rs:356-429 -> (expr, degree)
    new_idx = ctx.num_challenges
    k = e[0]
    if k == 'const':
        return e, 0
    if k == 'poly':
        return e, (1 if ctx.subtype(e[1]) in ('advice', 'lookup') else 0)
    if k == 'chal':
        return e, 1
    if k == 'neg':
        a, d = homogeneous(e[1], ctx)
        return Neg(a), d
    if k == 'sum':
        (l, dl), (r, dr) = homogeneous(e[1], ctx), homogeneous(e[2], ctx)
        if dl > dr:
            return Sum(l, Prod(r, challenge_in_degree(new_idx, dl - dr))), dl
        if dl < dr:
            return Sum(Prod(l, challenge_in_degree(new_idx, dr - dl)), r), dr
        return Sum(l, r), dl
    if k == 'prod':
        (l, dl), (r, dr) = homogeneous(e[1], ctx), homogeneous(e[2], ctx)
        return Prod(l, r), dl + dr
    if k == 'scaled':
        a, d = homogeneous(e[1], ctx)
        return Scaled(a, e[2]), d
    raise NotImplementedError(k)


def compress_expression(exprs, challenge_index):   # src/plonk/util.rs:34-56
    y = Chal(challenge_index)
    if len(exprs) > 1:
        acc = Const(0)
        for ex in exprs:
            acc = Sum(ex, Prod(acc, y))
        return acc
    return exprs[0] if exprs else Const(0)


# ---------------------------------------------------------------------------- GroupedPoly
class GroupedPoly:
    def __init__(self, terms):
        self.terms = list(terms)           # list of Optional[expr]

    @staticmethod
    def new(e, ctx):                       # grouped_poly.rs:88-138
        k = e[0]
        if k == 'const':
            return GroupedPoly([e])
        if k == 'poly':
            terms = [e]
            st = ctx.subtype(e[1])
            if st in ('advice', 'lookup'):
                terms.append(Poly(e[1] + ctx.num_fold_vars(), e[2]))
            return GroupedPoly(terms)
        if k == 'chal':
            return GroupedPoly([e, Chal(e[1] + ctx.num_challenges)])
        if k == 'neg':
            return GroupedPoly.new(e[1], ctx).neg()
        if k == 'sum':
            return GroupedPoly.new(e[1], ctx).add(GroupedPoly.new(e[2], ctx))
        if k == 'prod':
            return GroupedPoly.new(e[1], ctx).mul(GroupedPoly.new(e[2], ctx))
        if k == 'scaled':
            return GroupedPoly.new(e[1], ctx).mul_scalar(e[2])
        raise NotImplementedError(k)

    def neg(self):                         # :270-282
        return GroupedPoly([None if t is None else Neg(t) for t in self.terms])

    def add(self, rhs):                    # impl_poly_ops!(Add...), :166-196
        out = []
        for i in range(max(len(self.terms), len(rhs.terms))):
            l = self.terms[i] if i < len(self.terms) else None
            r = rhs.terms[i] if i < len(rhs.terms) else None
            if l is not None and r is not None:
                out.append(Sum(l, r))
            elif r is not None:
                out.append(r)
            elif l is not None:
                out.append(l)
            else:
                out.append(None)
        return GroupedPoly(out)

    def sub(self, rhs):                    # impl_poly_ops!(Sub, sub, Sum, Neg::neg), :166-197
        out = []
        for i in range(max(len(self.terms), len(rhs.terms))):
            l = self.terms[i] if i < len(self.terms) else None
            r = rhs.terms[i] if i < len(rhs.terms) else None
            if l is not None and r is not None:
                out.append(Sum(l, Neg(r)))
            elif r is not None:
                out.append(Neg(r))
            elif l is not None:
                out.append(l)
            else:
                out.append(None)
        return GroupedPoly(out)

    @staticmethod
    def from_map(m):                       # From<HashMap<usize, Expression>> (tests, :285-300)
        terms = [None] * (max(m) + 1)
        for d, e in m.items():
            terms[d] = e
        return GroupedPoly(terms)

    def iter_with_degree(self):            # :140-145
        return [(d, t) for d, t in enumerate(self.terms) if t is not None]

    def mul_scalar(self, k):               # Mul<&F>, :198-214
        return GroupedPoly([None if t is None else Prod(Const(k), t) for t in self.terms])

    def mul(self, other):                  # :216-268
        if len(self.terms) <= len(other.terms):
            lhs, rhs = other, self
        else:
            lhs, rhs = self, other
        res = []
        rhs_terms = [(d, t) for d, t in enumerate(rhs.terms) if t is not None][::-1]
        for ld, lt in [(d, t) for d, t in enumerate(lhs.terms) if t is not None][::-1]:
            for rd, rt in rhs_terms:
                deg = ld + rd
                ex = Prod(lt, rt)
                if deg >= len(res):
                    res.extend([None] * (deg + 1 - len(res)))
                res[deg] = ex if res[deg] is None else Sum(res[deg], ex)
        return GroupedPoly(res)

    def get(self, i):
        return self.terms[i] if i < len(self.terms) else None

    def iter_from_first(self):             # :149-151
        return self.terms[1:]

    def __len__(self):
        return len(self.terms)


@dataclass
class CompressedGates:                     # src/plonk/mod.rs:68-121
    compressed: tuple
    homogeneous: tuple
    degree: int
    ctx: QueryIndexContext
    num_challenges_compressed: int

    @staticmethod
    def new(exprs, ctx):
        compressed = compress_expression(exprs, ctx.num_challenges)
        ctx.num_challenges = num_challenges(compressed)
        nc = ctx.num_challenges
        hom, deg = homogeneous(compressed, ctx)
        ctx.num_challenges = num_challenges(hom)
        return CompressedGates(compressed, hom, deg, QueryIndexContext(**vars(ctx)), nc)

    def grouped(self):
        return GroupedPoly.new(self.homogeneous, self.ctx)


# ---------------------------------------------------------------------------- GraphEvaluator (compile)
K_CONST, K_INTER, K_FIXED, K_POLY, K_CHAL = range(5)
OP_ADD, OP_SUB, OP_MUL, OP_SQUARE, OP_DOUBLE, OP_NEGATE, OP_STORE = range(7)


class GraphEvaluator:                      # graph_evaluator.rs:166-351
    def __init__(self, expr, p):
        self.p = p
        self.constants = [0, 1, 2]         # :186
        self.rotations = []
        self.calcs = []                    # (calc tuple, target)
        self._index = {}                   # calc tuple -> target (same "first equal entry" as the reference's linear scan)
        self.num_intermediates = 0
        vs = self.add_expression(expr)
        self.add_calculation((OP_STORE, vs, None))

    def add_rotation(self, rot):
        if rot in self.rotations:
            return self.rotations.index(rot)
        self.rotations.append(rot)
        return len(self.rotations) - 1

    def add_constant(self, c):
        c %= self.p
        if c in self.constants:
            return (K_CONST, self.constants.index(c))
        self.constants.append(c)
        return (K_CONST, len(self.constants) - 1)

    def add_calculation(self, calc):
        hit = self._index.get(calc)
        if hit is not None:
            return (K_INTER, hit)
        target = self.num_intermediates
        self.calcs.append((calc, target))
        self._index[calc] = target
        self.num_intermediates += 1
        return (K_INTER, target)

    def add_expression(self, e):           # :261-351
        k = e[0]
        Z, ONE, TWO = (K_CONST, 0), (K_CONST, 1), (K_CONST, 2)
        if k == 'const':
            return self.add_constant(e[1])
        if k == 'poly':
            r = self.add_rotation(e[2])
            return self.add_calculation((OP_STORE, (K_POLY, e[1], r), None))
        if k == 'chal':
            return self.add_calculation((OP_STORE, (K_CHAL, e[1]), None))
        if k == 'neg':
            if e[1][0] == 'const':
                return self.add_constant(-e[1][1])
            a = self.add_expression(e[1])
            return a if a == Z else self.add_calculation((OP_NEGATE, a, None))
        if k == 'sum':
            if e[2][0] == 'neg':
                a = self.add_expression(e[1])
                b = self.add_expression(e[2][1])
                if a == Z:
                    return self.add_calculation((OP_NEGATE, b, None))
                if b == Z:
                    return a
                return self.add_calculation((OP_SUB, a, b))
            a = self.add_expression(e[1])
            b = self.add_expression(e[2])
            return self.add_calculation((OP_ADD, a, b) if a <= b else (OP_ADD, b, a))
        if k == 'prod':
            a = self.add_expression(e[1])
            b = self.add_expression(e[2])
            if a == Z or b == Z:
                return Z
            if a == ONE:
                return b
            if b == ONE:
                return a
            if a == TWO:
                return self.add_calculation((OP_DOUBLE, b, None))
            if b == TWO:
                return self.add_calculation((OP_DOUBLE, a, None))
            if a == b:
                return self.add_calculation((OP_SQUARE, a, None))
            return self.add_calculation((OP_MUL, a, b) if a <= b else (OP_MUL, b, a))
        if k == 'scaled':
            f = e[2] % self.p
            if f == 0:
                return Z
            if f == 1:
                return self.add_expression(e[1])
            cst = self.add_constant(f)
            a = self.add_expression(e[1])
            return self.add_calculation((OP_MUL, a, cst))
        raise NotImplementedError(k)

    def export(self, field, O):
        """-> dict for oracle.eval_program (calcs flattened, constants in Montgomery form)."""
        import numpy as np

        def src(v):
            if v is None:
                return [0, 0, 0]
            if v[0] in (K_CONST, K_INTER, K_CHAL):
                return [v[0], v[1], 0]
            return [v[0], v[1], v[2]]
        rows = []
        for (op, a, b), target in self.calcs:
            rows.append([op] + src(a) + src(b) + [target])
        return {"calcs": np.array(rows, dtype=np.int64).reshape(-1, 8),
                "constants": O.ints_to_mont(field, self.constants),
                "rotations": np.array(self.rotations if self.rotations else [0], dtype=np.int32),
                "n_intermediates": self.num_intermediates}


# ---------------------------------------------------------------------------- MainGate<T> (src/main_gate.rs:535-583)
def main_gate_expression(T, num_selectors=0, fixed_offset=0, advice_offset=0, num_fixed_total=None):
    """The single polynomial of MainGate<T>::configure as a Sirius Expression.

    Column creation order (main_gate.rs:537-546): advice state[0..T], input, out; fixed q_1[0..T],
    q_5[0..T], q_m[0..2], q_i, q_o, rc.  Index map = Expression::from_halo2_expr (expression.rs:301-336):
    fixed -> num_selectors + column, advice -> num_selectors + num_fixed + column.
    `*_offset`: column index of this gate's first fixed/advice column when several gates share a
    constraint system; num_fixed_total = total fixed columns of the system."""
    nf = 2 * T + 5
    if num_fixed_total is None:
        num_fixed_total = nf
    F = lambda i: Poly(num_selectors + fixed_offset + i)
    A = lambda i: Poly(num_selectors + num_fixed_total + advice_offset + i)
    state = [A(i) for i in range(T)]
    inp, out = A(T), A(T + 1)
    q_1 = [F(i) for i in range(T)]
    q_5 = [F(T + i) for i in range(T)]
    q_m = [F(2 * T), F(2 * T + 1)]
    q_i, q_o, rc = F(2 * T + 2), F(2 * T + 3), F(2 * T + 4)

    def pow_5(v):
        v2 = Prod(v, v)
        return Prod(Prod(v2, v2), v)
    init = Sum(Sum(Sum(Prod(Prod(q_m[0], state[0]), state[1]), Prod(q_i, inp)), rc), Prod(q_o, out))
    if T >= 4:
        init = Sum(Prod(Prod(q_m[1], state[2]), state[3]), init)
    acc = init
    for s, q1, q5 in zip(state, q_1, q_5):
        acc = Sum(acc, Sum(Prod(q1, s), Prod(q5, pow_5(s))))
    return acc


def serialize_gates(exprs, field, O):
    """Expression list -> the SRS_EX_* postfix word stream of include/sirius_amd.h (numpy uint64)."""
    import numpy as np
    words = []

    def fe(v):
        return [int(x) for x in O.ints_to_mont(field, [v])[0]]

    def walk(e):
        k = e[0]
        if k == 'const':
            words.extend([0] + fe(e[1]))
        elif k == 'poly':
            words.extend([1, e[1], e[2] & 0xFFFFFFFFFFFFFFFF])
        elif k == 'chal':
            words.extend([2, e[1]])
        elif k == 'neg':
            walk(e[1]); words.append(3)
        elif k == 'sum':
            walk(e[1]); walk(e[2]); words.append(4)
        elif k == 'prod':
            walk(e[1]); walk(e[2]); words.append(5)
        elif k == 'scaled':
            walk(e[1]); words.extend([6] + fe(e[2]))
        else:
            raise NotImplementedError(k)
    for ex in exprs:
        walk(ex)
        words.append(7)
    return np.array(words, dtype=np.uint64)


def cross_terms_oracle(O, field, gates, num_selectors, num_fixed, num_advice, selectors, fixed, W1, W2, challenges, threads=0):
    """VanillaFS::commit_cross_terms evaluation half (src/nifs/sangria/mod.rs:102-148), the reference's
    way: GroupedPoly terms 1..d, one GraphEvaluator + one pass over the rows per term."""
    import numpy as np
    from . import pyref as P
    p = P.MODULI[field]
    ctx = QueryIndexContext(num_selectors, num_fixed, num_advice, 0, 0)
    cg = CompressedGates.new(gates, ctx)
    out = []
    rows = fixed[0].shape[0] if len(fixed) else selectors[0].shape[0]
    for term in cg.grouped().iter_from_first():
        if term is None:
            out.append(np.zeros((rows, 4), dtype=np.uint64))     # sangria/mod.rs:145
            continue
        prog = GraphEvaluator(term, p).export(field, O)
        out.append(O.eval_program(field, prog, selectors, fixed, W1, W2, challenges, threads))
    return cg, out

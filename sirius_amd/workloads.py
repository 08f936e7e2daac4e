"""Synthetic Poseidon-shaped Sangria / CycleFold workloads: what bench.py folds, shared with the tests (tests/workloads.py
re-exports this module).

A real halo2 trace cannot be produced without the Rust front-end, so shapes follow the reference
configs: the primary structure is MainGate<5> (step-folding circuit) + MainGate<3> (Poseidon step
circuit) = 12 advice / 26 fixed columns, 2 gates (benches/sangria_poseidon.rs:28-36,80-81); the
secondary structure is MainGate<5> alone = 7 advice / 15 fixed, 1 gate.  Values are seeded.
Product-side only (no oracle import): usable from bench.py's timed leg."""
import numpy as np

from . import expression as X
from .field import MODULUS


def rand_fe(rng, n, zero_frac=0.0):
    """(n,4) uint64 values < 2^253 (canonical residues of either field; used as Montgomery bit patterns)."""
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 61) - 1)
    if zero_frac:
        raw[rng.random(n) < zero_frac] = 0
    return raw


def trace_like(rng, n):
    """bench.py's witness: 55 % zero scalars, 45 % uniform 253-bit patterns (used as Montgomery bit patterns).  NOT SURVEY.md 8d(ii)'s
    mixture: every non-zero scalar has 16 non-zero 16-bit digits (7.2 bucket additions per scalar against ~2.3 for the mixture), and
    no bucket is hot -- conservative in additions, but the overflow path of slot mode never runs.  `survey_mixture` is the mixture."""
    return rand_fe(rng, n, zero_frac=0.55)


def survey_mixture(rng, n, field=0):
    """SURVEY.md 8d(ii)'s ASSUMED trace mixture as CANONICAL scalar values, returned in the ABI's Montgomery form: 55 % zero, 20 % bits
    (0 / 1), 15 % uniform < 2^64, 10 % uniform < 2^253.  10 % of all scalars are the value 1: bucket 0 of window 0 is hot in every
    chunk of a streamed commit (slot mode's overflow kernels run on every set)."""
    p = MODULUS[field]
    R = (1 << 256) % p
    u = rng.random(n)
    out = rand_fe(rng, n)                                   # the uniform tenth: a uniform Montgomery pattern is a uniform value
    out[u < 0.90] = 0
    bits = (u >= 0.55) & (u < 0.75)
    ones = bits & (rng.random(n) < 0.5)
    out[ones] = np.array([(R >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    small = np.nonzero((u >= 0.75) & (u < 0.90))[0]
    vals = rng.integers(0, 1 << 63, size=small.size, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=small.size, dtype=np.uint64)
    m64 = (1 << 64) - 1
    conv = np.empty((small.size, 4), dtype=np.uint64)
    for j, v in enumerate(vals.tolist()):                   # v * R mod p: python integers (a few seconds for the 1.9 M of a k = 20 witness)
        m = v * R % p
        conv[j] = (m & m64, (m >> 64) & m64, (m >> 128) & m64, m >> 192)
    out[small] = conv
    return out


def sangria_shape(which):
    if which == "primary":      # bn256 circuit over Fr
        return dict(field=0, curve=0, gate_T=[5, 3])
    return dict(field=1, curve=1, gate_T=[5])   # grumpkin circuit over Fq


def support_gate(num_selectors=1, num_fixed=4):
    """The one gate of the CycleFold support circuit (reference src/ivc/cyclefold/support_circuit/tiny_gate.rs:56-82):
      s * (state0 * state1 * mul + state0 * sum0 + state1 * sum1 + rc - output)
    1 selector, fixed = (mul, sum0, sum1, rc), advice = (state0, state1, output); k = 15 (support_circuit/mod.rs:68)."""
    s = X.Polynomial(0)
    F = lambda i: X.Polynomial(num_selectors + i)
    A = lambda i: X.Polynomial(num_selectors + num_fixed + i)
    mul, sum0, sum1, rc = F(0), F(1), F(2), F(3)
    s0, s1, out = A(0), A(1), A(2)
    inner = X.Sum(X.Sum(X.Sum(X.Sum(X.Product(X.Product(s0, s1), mul), X.Product(s0, sum0)), X.Product(s1, sum1)), rc), X.Negated(out))
    return X.Product(s, inner)


def make_support_inputs(k, seed):
    """Synthetic support-circuit structure (grumpkin circuit over Fq): -> dict like make_structure_inputs, plus `selectors`."""
    rng = np.random.default_rng(seed)
    rows = 1 << k
    sel = (rng.random(rows) < 0.8).astype(np.uint8)
    fixed = [rand_fe(rng, rows, zero_frac=0.5) for _ in range(4)]
    return dict(field=1, curve=1, k=k, rows=rows, gates=[support_gate()], selectors=[sel], fixed=fixed, num_fixed=4,
                num_advice=3, W1=trace_like(rng, 3 * rows), W2=trace_like(rng, 3 * rows), E=rand_fe(rng, rows),
                u1_challenges=rand_fe(rng, 0), u1_u=rand_fe(rng, 1)[0], u2_challenges=rand_fe(rng, 0), r=rand_fe(rng, 1)[0],
                modulus=MODULUS[1])


def high_degree_gate(T, d, num_selectors=0, fixed_offset=0, advice_offset=0, num_fixed_total=None):
    """A "high-degree gate" for BASELINE configs[3]: MainGate<T>'s polynomial plus the monomial  q_1[0] * s[0]^ceil(d/2) * s[1]^floor(d/2)
    of degree d (Expression::degree counts advice queries only, src/polynomial/expression.rs:431-447) over the SAME columns (T + 2 advice,
    2T + 5 fixed).  ProtoGalaxy sizes that follow (src/nifs/protogalaxy/poly/mod.rs:535-545, :263-268): one incoming trace and
    d = 8..15 -> 16 points of G and, by quirk Q2, a 2^16-point K domain; d >= 16 -> 32 points, whose K "log" exceeds F::S."""
    nf = 2 * T + 5
    if num_fixed_total is None:
        num_fixed_total = nf
    q = X.Polynomial(num_selectors + fixed_offset)
    s0 = X.Polynomial(num_selectors + num_fixed_total + advice_offset)
    s1 = X.Polynomial(num_selectors + num_fixed_total + advice_offset + 1)

    def power(v, e):
        out = v
        for _ in range(e - 1):
            out = X.Product(out, v)
        return out
    a, b = (d + 1) // 2, d // 2
    mono = X.Product(q, X.Product(power(s0, a), power(s1, b)) if b else power(s0, a))
    return X.Sum(X.main_gate(T, num_selectors, fixed_offset, advice_offset, num_fixed_total), mono)


def gates_for(gate_T):
    nfix = sum(2 * T + 5 for T in gate_T)
    nadv = sum(T + 2 for T in gate_T)
    gates, fo, ao = [], 0, 0
    for T in gate_T:
        gates.append(X.main_gate(T, 0, fo, ao, nfix))
        fo += 2 * T + 5
        ao += T + 2
    return gates, nfix, nadv


def make_structure_inputs(which, k, seed):
    """-> dict(field, curve, k, gates, fixed (list of (rows,4)), num_advice, W1, W2, E, challenges...)."""
    sh = sangria_shape(which)
    rng = np.random.default_rng(seed)
    rows = 1 << k
    gates, nfix, nadv = gates_for(sh["gate_T"])
    fixed = [rand_fe(rng, rows, zero_frac=0.9 if i % 2 else 0.3) for i in range(nfix)]
    nch = 1 if len(gates) > 1 else 0
    return dict(field=sh["field"], curve=sh["curve"], k=k, rows=rows, gates=gates, fixed=fixed, num_fixed=nfix,
                num_advice=nadv, W1=trace_like(rng, nadv * rows), W2=trace_like(rng, nadv * rows),
                E=rand_fe(rng, rows), u1_challenges=rand_fe(rng, nch), u1_u=rand_fe(rng, 1)[0],
                u2_challenges=rand_fe(rng, nch), r=rand_fe(rng, 1)[0], modulus=MODULUS[sh["field"]])

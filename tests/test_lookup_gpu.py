"""GPU parity: lookup arguments (log-derivative) -- prover coefficients l/t/m/h/g, the lookup-extended gate
programs (cross terms, deciders) and the log-derivative check, against oracle/lookup.py.
Mirrors nifs::sangria::tests::three_rounds_test (src/nifs/sangria/tests.rs:313-346); see tests/lookup_cases.py."""
import numpy as np
import pytest

from lookup_cases import run_lookup_case

pytestmark = pytest.mark.gpu


def test_three_rounds_vector_lookup(srs, oracle):
    run_lookup_case(srs, oracle, "vector", 5)          # the reference's K = 5
    run_lookup_case(srs, oracle, "vector", 9)


def test_two_rounds_scalar_lookup(srs, oracle):
    run_lookup_case(srs, oracle, "scalar", 6)
    run_lookup_case(srs, oracle, "scalar", 10)


def test_two_lookups_index_map(srs, oracle):
    run_lookup_case(srs, oracle, "two", 6)


def test_lookup_coefficients_device_resident_k16(srs, oracle):
    """2^16 rows on HBM-resident tensors: multiplicities through size-independent properties (sum m == rows when every
    looked-up value is in the table; repeats of a table value carry 0; h (l + r) == 1, g (t + r) == m row by row)."""
    import torch
    O = oracle
    X = srs.expression
    k, field = 16, 0
    rows = 1 << k
    rng = np.random.default_rng(77)
    tbl = (np.arange(rows) % 4096).astype(np.int64)                 # every table value 16 times
    fixed = [O.ints_to_mont(field, [int(v) for v in tbl])]
    sel = [np.ones(rows, np.uint8)]
    a = rng.integers(0, 4096, size=rows)
    a[: rows // 2] = 7                                              # one hot value: contended slot
    adv = O.ints_to_mont(field, [int(v) for v in a])
    St = srs.PlonkStructure(field, k, sel, fixed, 1, [], lookups=[([X.Product(X.Polynomial(0), X.Polynomial(2))], [X.Polynomial(1)])])
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).cuda()
    ls, ts, ms = St.lookup_coeff_1(dev(adv), np.zeros(4, np.uint64))
    m = np.array(O.mont_to_ints(field, ms[0].cpu().numpy().view(np.uint64)), dtype=object)
    cnt = np.bincount(a, minlength=4096)
    assert int(m.sum()) == rows
    assert all(int(m[i]) == int(cnt[i]) for i in range(4096)) and not any(m[4096:])
    r = O.ints_to_mont(field, [0x1234567])[0]
    hs, gs = St.lookup_coeff_2(ls, ts, ms, r)
    rb = np.broadcast_to(r, (rows, 4)).copy()
    h, g = hs[0].cpu().numpy().view(np.uint64), gs[0].cpu().numpy().view(np.uint64)
    l, t = ls[0].cpu().numpy().view(np.uint64), ts[0].cpu().numpy().view(np.uint64)
    one = O.ints_to_mont(field, [1])[0]
    assert np.array_equal(O.fe_mul(field, h, O.fe_add(field, l, rb)), np.broadcast_to(one, (rows, 4)))
    assert np.array_equal(O.fe_mul(field, g, O.fe_add(field, t, rb)), ms[0].cpu().numpy().view(np.uint64))
    W = [torch.cat([dev(adv)] + ls + ts + ms), torch.cat(hs + gs)]
    assert St.is_sat_log_derivative(W) == 0
    St.close()


def _assigned_case(S, O, field, n, seed=1):
    """batch_invert_assigned (src/util/mod.rs:119-153): Zero / Trivial / Rational cells incl. zero denominators."""
    from oracle import pyref as P
    p = P.MODULI[field]
    rng = np.random.default_rng(seed + n)
    num = [int(rng.integers(0, 1 << 62)) * int(rng.integers(1, 1 << 62)) % p for _ in range(n)]
    den = [int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % p for _ in range(n)]
    has = rng.integers(0, 2, size=n).astype(np.uint8)
    for i in range(0, n, 7):
        den[i] = 0                                              # Rational(n, 0) -> 0
    for i in range(3, n, 11):
        num[i] = 0
        has[i] = 0                                              # Assigned::Zero
    exp = [(a * pow(d, p - 2, p) % p if d else 0) if h else a for a, d, h in zip(num, den, has)]
    got = S.batch_invert_assigned(field, O.ints_to_mont(field, num), O.ints_to_mont(field, den), has)
    assert O.mont_to_ints(field, got) == exp
    exp_all = [a * pow(d, p - 2, p) % p if d else 0 for a, d in zip(num, den)]
    assert O.mont_to_ints(field, S.batch_invert_assigned(field, O.ints_to_mont(field, num), O.ints_to_mont(field, den))) == exp_all


def test_batch_invert_assigned(srs, oracle):
    _assigned_case(srs, oracle, 0, 5000)
    _assigned_case(srs, oracle, 1, 1)
    _assigned_case(srs, oracle, 1, 1025)

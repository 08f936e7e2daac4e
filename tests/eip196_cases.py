"""EIP-196 (alt_bn128 ecAdd / ecMul) known answers -> checks shared by the CPU and GPU tests (tests/golden/alt_bn128_eip196.json)."""
import numpy as np

from conftest import golden, h2i


def vectors():
    g = golden("alt_bn128_eip196.json")
    assert h2i(g["prime"]) == 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
    return g


def pt(v):
    return (h2i(v[0]), h2i(v[1]))


def mont_point(O, v):
    """-> (8,) uint64 affine point in the ABI layout (Montgomery x || y, identity all-zero)"""
    return O.ints_to_mont(O.BASE_FIELD[0], [h2i(v[0]), h2i(v[1])]).reshape(8)


def check_adder(O, add_fn, mul_fn, msm_fn=None):
    """add_fn(a, b) / mul_fn(k_mont, p) / msm_fn(scalars (n,4), bases (n,8)) on ABI-layout arrays -> (8,) points"""
    from oracle import pyref as P
    r = P.CURVES[0].q
    one = O.ints_to_mont(0, [1])
    g = vectors()
    for rec in g["add"]:
        a, b, want = mont_point(O, rec["a"]), mont_point(O, rec["b"]), mont_point(O, rec["out"])
        assert np.array_equal(add_fn(a, b), want), ("ecAdd", rec["name"])
        assert np.array_equal(add_fn(b, a), want), ("ecAdd commuted", rec["name"])
        if msm_fn is not None:       # P + Q as the 2-term MSM 1*P + 1*Q
            assert np.array_equal(msm_fn(np.concatenate([one, one]), np.stack([a, b])), want), ("ecAdd as MSM", rec["name"])
    for rec in g["mul"]:
        p, want = mont_point(O, rec["p"]), mont_point(O, rec["out"])
        k = h2i(rec["k"]) % r        # EIP-196 takes any 256-bit scalar; the group has order r, the ABI takes field elements
        km = O.ints_to_mont(0, [k])
        assert np.array_equal(mul_fn(km[0], p), want), ("ecMul", rec["name"])
        if msm_fn is not None:
            assert np.array_equal(msm_fn(km, p.reshape(1, 8)), want), ("ecMul as MSM", rec["name"])
            # [k]P = [k - 1]P + [1]P as a 2-term MSM over the repeated base
            k1 = O.ints_to_mont(0, [(k - 1) % r])
            assert np.array_equal(msm_fn(np.concatenate([k1, one]), np.stack([p, p])), want), ("ecMul split", rec["name"])

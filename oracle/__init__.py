"""ctypes binding of the CPU oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package ``sirius_amd`` never imports this module.

Arrays are numpy uint64 with a trailing limb axis: field elements ``(..., 4)``, affine points
``(..., 8)`` (x limbs then y limbs), Montgomery form, little-endian limbs -- byte-identical to the
product C-ABI (include/sirius_amd.h) so results can be compared with ``np.array_equal``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

FR, FQ = 0, 1
BN256, GRUMPKIN = 0, 1
SCALAR_FIELD = {BN256: FR, GRUMPKIN: FQ}
BASE_FIELD = {BN256: FQ, GRUMPKIN: FR}


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _fe(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4, a.shape
    return a


def _aff(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 8, a.shape
    return a


def _unary(name, field, a):
    a = _fe(a)
    out = np.empty_like(a)
    getattr(lib(), name)(C.c_int(field), _p(a), _p(out), C.c_size_t(a.size // 4))
    return out


def _binary(name, field, a, b):
    a, b = _fe(a), _fe(b)
    assert a.shape == b.shape
    out = np.empty_like(a)
    getattr(lib(), name)(C.c_int(field), _p(a), _p(b), _p(out), C.c_size_t(a.size // 4))
    return out


def to_mont(field, a): return _unary("o_fe_to_mont", field, a)
def from_mont(field, a): return _unary("o_fe_from_mont", field, a)
def fe_inv(field, a): return _unary("o_fe_inv", field, a)
def fe_mul(field, a, b): return _binary("o_fe_mul", field, a, b)
def fe_add(field, a, b): return _binary("o_fe_add", field, a, b)
def fe_sub(field, a, b): return _binary("o_fe_sub", field, a, b)


def ints_to_limbs(vals):
    """python ints (canonical) -> (n,4) uint64 limbs (NOT Montgomery)."""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(a[i, j]) << (64 * j) for j in range(4)) for i in range(a.shape[0])]


def ints_to_mont(field, vals):
    return to_mont(field, ints_to_limbs(vals))


def mont_to_ints(field, a):
    return limbs_to_ints(from_mont(field, np.asarray(a).reshape(-1, 4)))


def point_add(curve, a, b):
    a, b = _aff(a), _aff(b)
    out = np.empty(8, dtype=np.uint64)
    lib().o_point_add(C.c_int(curve), _p(a), _p(b), _p(out))
    return out


def point_mul(curve, scalar_mont, p):
    s, p = _fe(scalar_mont), _aff(p)
    out = np.empty(8, dtype=np.uint64)
    lib().o_point_mul(C.c_int(curve), _p(s), _p(p), _p(out))
    return out


def is_on_curve(curve, p):
    p = _aff(p)
    return bool(lib().o_point_is_on_curve(C.c_int(curve), _p(p)))


def make_bases(curve, seed, n, threads=0):
    out = np.empty((n, 8), dtype=np.uint64)
    lib().o_make_bases(C.c_int(curve), C.c_uint64(seed), _p(out), C.c_size_t(n), C.c_int(threads))
    return out


def msm(curve, scalars, bases, threads=0):
    s, b = _fe(scalars), _aff(bases)
    n = s.shape[0]
    assert b.shape[0] >= n
    out = np.empty(8, dtype=np.uint64)
    lib().o_msm(C.c_int(curve), _p(s), _p(b), C.c_size_t(n), C.c_int(threads), _p(out))
    return out


def msm_naive(curve, scalars, bases):
    s, b = _fe(scalars), _aff(bases)
    out = np.empty(8, dtype=np.uint64)
    lib().o_msm_naive(C.c_int(curve), _p(s), _p(b), C.c_size_t(s.shape[0]), _p(out))
    return out


def _ntt(name, a, threads):
    a = _fe(a).copy()
    rc = getattr(lib(), name)(_p(a), C.c_size_t(a.shape[0]), C.c_int(threads))
    if rc:
        raise ValueError(f"{name}: rc={rc}")
    return a


def fft(a, threads=0): return _ntt("o_fft", a, threads)
def ifft(a, threads=0): return _ntt("o_ifft", a, threads)
def coset_fft(a, threads=0): return _ntt("o_coset_fft", a, threads)
def coset_ifft(a, threads=0): return _ntt("o_coset_ifft", a, threads)


class _EvalDomain(C.Structure):
    _fields_ = [("field", C.c_int), ("rows", C.c_size_t), ("n_sel", C.c_size_t),
                ("n_fixed", C.c_size_t), ("num_advice", C.c_size_t), ("num_lookup", C.c_size_t),
                ("selectors", C.POINTER(C.c_void_p)), ("fixed", C.POINTER(C.c_void_p)),
                ("n_w1", C.c_size_t), ("n_w2", C.c_size_t),
                ("W1s", C.c_void_p * 3), ("W2s", C.c_void_p * 3),
                ("w1_len", C.c_size_t * 3), ("w2_len", C.c_size_t * 3),
                ("challenges", C.c_void_p), ("n_challenges", C.c_size_t)]


def _rounds(W):
    """a single (n,4) vector or a list of round vectors -> list of contiguous round arrays"""
    if W is None:
        return []
    if isinstance(W, (list, tuple)):
        return [_fe(np.asarray(w).reshape(-1, 4)) for w in W]
    return [_fe(np.asarray(W).reshape(-1, 4))]


def eval_program(field, prog, selectors, fixed, W1, W2, challenges, threads=0, num_advice=None, num_lookup=0):
    """prog = dict(calcs (n,8) int64, constants (m,4) mont, rotations int32, n_intermediates).
    W1 / W2: one round-0 vector, or the list of witness rounds (PlonkWitness::W) when the structure has lookups
    (then num_advice must be given)."""
    rows = fixed[0].shape[0] if len(fixed) else selectors[0].shape[0]
    sel = [np.ascontiguousarray(s, dtype=np.uint8) for s in selectors]
    fx = [_fe(f) for f in fixed]
    r1, r2 = _rounds(W1), _rounds(W2)
    ch = _fe(np.asarray(challenges, dtype=np.uint64).reshape(-1, 4))
    if num_advice is None:
        assert num_lookup == 0 and len(r1) == 1
        num_advice = r1[0].shape[0] // rows
    selp = (C.c_void_p * max(len(sel), 1))(*[s.ctypes.data for s in sel])
    fxp = (C.c_void_p * max(len(fx), 1))(*[f.ctypes.data for f in fx])
    d = _EvalDomain()
    d.field, d.rows, d.n_sel, d.n_fixed, d.num_advice, d.num_lookup = field, rows, len(sel), len(fx), num_advice, num_lookup
    d.selectors, d.fixed = selp, fxp
    d.n_w1, d.n_w2 = len(r1), len(r2)
    for i, w in enumerate(r1):
        d.W1s[i], d.w1_len[i] = w.ctypes.data, w.shape[0]
    for i, w in enumerate(r2):
        d.W2s[i], d.w2_len[i] = w.ctypes.data, w.shape[0]
    d.challenges, d.n_challenges = ch.ctypes.data, ch.shape[0]
    calcs = np.ascontiguousarray(prog["calcs"], dtype=np.int64).reshape(-1, 8)
    consts = _fe(prog["constants"])
    rots = np.ascontiguousarray(prog["rotations"], dtype=np.int32)
    out = np.empty((rows, 4), dtype=np.uint64)
    rc = lib().o_eval_program(C.byref(d), _p(calcs), C.c_size_t(calcs.shape[0]), _p(consts),
                              C.c_size_t(consts.shape[0]), _p(rots), C.c_size_t(rots.shape[0]),
                              C.c_size_t(int(prog["n_intermediates"])), _p(out), C.c_int(threads))
    if rc:
        raise ValueError("o_eval_program: index out of range")
    return out


def fold_w(field, w1, w2, r, threads=0):
    w1, w2, r = _fe(w1), _fe(w2), _fe(r)
    out = np.empty_like(w1)
    lib().o_fold_w(C.c_int(field), _p(w1), _p(w2), _p(r), _p(out), C.c_size_t(w1.size // 4), C.c_int(threads))
    return out


def fold_e(field, e, terms, r, threads=0):
    e, r = _fe(e), _fe(r)
    ts = [_fe(t) for t in terms]
    tp = (C.c_void_p * max(len(ts), 1))(*[t.ctypes.data for t in ts])
    out = np.empty_like(e)
    lib().o_fold_e(C.c_int(field), _p(e), tp, C.c_size_t(len(ts)), _p(r), _p(out),
                   C.c_size_t(e.shape[0]), C.c_int(threads))
    return out


def lincomb(field, ws, coefs, threads=0):
    """sum_j coefs[j] * ws[j]  (ProtoGalaxy::fold_witness / FoldedWitness::new)."""
    ws = [_fe(w) for w in ws]
    cf = _fe(np.asarray(coefs, dtype=np.uint64).reshape(-1, 4))
    assert cf.shape[0] >= len(ws)
    wp = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
    out = np.empty_like(ws[0])
    lib().o_lincomb(C.c_int(field), wp, _p(cf), C.c_size_t(len(ws)), _p(out), C.c_size_t(ws[0].size // 4), C.c_int(threads))
    return out


def pg_tree(field, leaves, count, weights, threads=0):
    """The reduction tree of compute_F / compute_G.  leaves: (n, 4) shared by all points or (P, n, 4); weights (P, t, 4)
    with n = 2^t; leaves beyond `count` are zero.  -> (P, 4)."""
    lv = _fe(leaves)
    w = _fe(weights)
    P, t = w.shape[0], w.shape[1]
    n = 1 << t
    stride = 0 if lv.ndim == 2 else lv.shape[1]
    assert (lv.shape[0] if lv.ndim == 2 else lv.shape[1]) >= min(count, n)
    out = np.empty((P, 4), dtype=np.uint64)
    lib().o_pg_tree(C.c_int(field), _p(lv), C.c_size_t(stride), C.c_size_t(n), C.c_size_t(count), _p(w), C.c_size_t(P),
                    C.c_size_t(t), _p(out), C.c_int(threads))
    return out

"""Host-side scalar helpers (python big ints <-> the 4 x u64 Montgomery layout of the C-ABI)."""
import numpy as np

FR = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001   # bn256::Fr = grumpkin::Fq(base)
FQ = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47   # bn256::Fq = grumpkin scalar
MODULUS = {0: FR, 1: FQ}
_R = 1 << 256
_MASK = (1 << 64) - 1


def to_mont(field, v):
    """python int (any residue) -> (4,) uint64 Montgomery limbs."""
    m = (v % MODULUS[field]) * _R % MODULUS[field]
    return np.array([(m >> (64 * i)) & _MASK for i in range(4)], dtype=np.uint64)


def from_mont(field, limbs):
    p = MODULUS[field]
    m = sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(4)))
    return m * pow(_R % p, p - 2, p) % p


def ints_to_mont(field, vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = to_mont(field, v)
    return out


def powers(field, r, n):
    """r^1 .. r^n as (n, 4) Montgomery limbs (library host code, srs_fe_powers): the scalars of the E-commitment fold."""
    from . import _lib as L
    rr = np.ascontiguousarray(r, dtype=np.uint64).reshape(4)
    out = np.zeros((n, 4), dtype=np.uint64)
    L.check(L.lib().srs_fe_powers(field, rr.ctypes.data, n, out.ctypes.data))
    return out

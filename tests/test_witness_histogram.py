"""tools/witness_histogram.py recodes scalars exactly as csrc/msm.hip:k_digits does: its count of non-zero signed 16-bit digits per scalar is
the `density` the library reports after a commit (srs_ck_msm_stats) -- checked here against a direct big-integer recoding."""
import importlib.util
import os

import numpy as np

from conftest import ROOT


def _tool():
    spec = importlib.util.spec_from_file_location("witness_histogram", os.path.join(ROOT, "tools", "witness_histogram.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _nonzero_digits(v):
    """k_digits on a python integer: v = word + carry; v > 0x8000 is a negative digit with carry 1; a digit is dropped when it is zero"""
    carry, nz = 0, 0
    for w in range(16):
        x = ((v >> (16 * w)) & 0xFFFF) + carry
        carry = 1 if x > 0x8000 else 0
        nz += 1 if x not in (0, 0x10000) else 0
    return nz


def test_digit_recoding_matches_k_digits():
    T = _tool()
    rng = np.random.default_rng(7)
    vals = [0, 1, 0x8000, 0x8001, 0xFFFF, 0x10000, (1 << 64) - 1, (1 << 128) - 1, 0x7FFF8000, 0xFFFFFFFF, (1 << 253) + 5]
    vals += [int.from_bytes(rng.bytes(32), "little") >> 3 for _ in range(200)]
    raw = b"".join(v.to_bytes(32, "little") for v in vals)
    words = np.frombuffer(raw, dtype="<u2").reshape(-1, 16).astype(np.uint32)
    nz, d0 = T.digits_nonzero(words)
    assert [int(x) for x in nz] == [_nonzero_digits(v) for v in vals]
    for v, d in zip(vals, d0):
        lo = v & 0xFFFF
        assert int(d) == (lo - 0x10000 if lo > 0x8000 else lo)


def test_report_on_the_bench_mixture(capsys):
    T = _tool()
    rng = np.random.default_rng(1)
    x = rng.integers(0, 1 << 63, size=(50000, 4), dtype=np.uint64)
    x[:, 3] &= np.uint64((1 << 60) - 1)
    x[rng.random(50000) < 0.55] = 0
    dens = T.report(x.tobytes())
    assert abs(dens - 7.2) < 0.1
    assert "nearer to 'bench'" in capsys.readouterr().out

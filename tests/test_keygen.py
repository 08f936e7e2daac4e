"""CommitmentKey::setup (reference src/commitment.rs:55-79): the SHAKE256(label) chunk stream through the C-ABI against
Python's hashlib (an independent FIPS 202 implementation) and the FIPS 202 known answer for the empty message; the
hash_to_curve half is explicitly unsupported.  Host code: runs without a GPU."""
import hashlib

import numpy as np
import pytest


def test_shake256_known_answer(srs):
    # FIPS 202 / NIST CAVP: SHAKE256("") starts 46b9dd2b 0ba88d13 233b3feb 743eeb24 3fcd52ea 62b81b82 b50c2764 6ed5762f
    got = srs.CommitmentKey.setup_uniform_bytes(b"", 0, 1)
    assert got.tobytes().hex() == "46b9dd2b0ba88d13233b3feb743eeb243fcd52ea62b81b82b50c27646ed5762f"


@pytest.mark.parametrize("label", [b"bn256", b"grumpkin", b"", b"x" * 135, b"y" * 136, b"z" * 137, bytes(range(256)) * 3])
def test_setup_stream_vs_hashlib(srs, label):
    """labels around the 136-byte rate (padding in the same / a fresh block), chunk windows across squeeze-block boundaries
    (136 is not a multiple of 32), a late window reached by skipping"""
    ref = hashlib.shake_256(label).digest(32 * 5000)
    got = srs.CommitmentKey.setup_uniform_bytes(label, 0, 300)
    assert got.tobytes() == ref[: 32 * 300]
    for first, count in ((1, 1), (4, 9), (17, 1), (4990, 10)):
        assert srs.CommitmentKey.setup_uniform_bytes(label, first, count).tobytes() == ref[32 * first: 32 * (first + count)]
    assert srs.CommitmentKey.setup_uniform_bytes(label, 7, 0).size == 0


def test_setup_is_explicitly_unsupported(srs):
    from sirius_amd import _lib as L
    with pytest.raises(L.SiriusAmdError) as e:
        srs.CommitmentKey.setup(0, 10, b"bn256")
    assert e.value.rc == L.ERR_UNSUPPORTED and "hash_to_curve" in str(e.value) and "srs_ck_load_file" in str(e.value)

"""sirius_amd -- MI355X-native folding-prover hot path for snarkify/sirius.

Host-side mirror of the reference interfaces this library replaces (names follow the reference):
  CommitmentKey.commit           <- src/commitment.rs:81-90
Arrays are numpy uint64 `(n, 4)` field elements / `(n, 8)` affine points (Montgomery, LE limbs)
or torch uint64/int64 CUDA tensors of the same shape for data already resident in HBM.
"""
from ._lib import (CURVE_BN256, CURVE_GRUMPKIN, FIELD_FQ, FIELD_FR, SiriusAmdError)  # noqa: F401
from .commitment import CommitmentKey, TooLongInput, point_mul, point_sum  # noqa: F401

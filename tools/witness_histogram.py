"""The value histogram of a REAL witness, and what it costs the MSM -- the figure SURVEY.md 8d(ii) could only assume (VERDICT r05 "missing" 7:
no trace can be produced in this image; this is the tool for a maintainer who has one).

Input: a raw dump of the witness as CANONICAL little-endian 32-byte values (`PrimeField::to_repr()` of every scalar, columns concatenated as
`concatenate_with_padding` orders them), e.g. from the reference:

    let bytes: Vec<u8> = primary_witness.iter().flatten().flat_map(|f| f.to_repr().as_ref().to_vec()).collect();
    std::fs::write("witness_k20.bin", bytes)?;                    // in CyclefoldIVC::next, after try_collect_witness (mod.rs:301-311)

Prints: the shares of zeros / bits / values below 2^16, 2^64, 2^128 / wide values; the non-zero signed 16-bit digits per scalar exactly as
csrc/msm.hip:k_digits recodes them (= bucket additions per scalar, the `density` the library's chunk schedule follows and
srs_ck_msm_stats reports after a commit); the heaviest buckets of window 0 (hot buckets: values repeated many times); and which of
bench.py's two witness mixtures (`--witness bench`: 7.2 additions per scalar, `--witness survey`: ~2.3) is nearer.
usage: python tools/witness_histogram.py witness.bin [--self-test]"""
import sys
import numpy as np


def digits_nonzero(words):
    """words: (n, 16) uint32 array of the 16-bit words of canonical values, least significant first -> non-zero signed digits per scalar,
    and the signed digit of window 0 (k_digits: v = word + carry; v > 0x8000 -> negative digit, carry 1)"""
    n = words.shape[0]
    carry = np.zeros(n, dtype=np.uint32)
    nz = np.zeros(n, dtype=np.uint32)
    d0 = None
    for w in range(16):
        v = words[:, w] + carry
        neg = v > 0x8000
        carry = neg.astype(np.uint32)
        nz += ((v != 0) & (v != 0x10000)).astype(np.uint32)
        if w == 0:
            d0 = np.where(neg, v.astype(np.int64) - 0x10000, v.astype(np.int64))
    return nz, d0


def report(raw):
    a = np.frombuffer(raw, dtype="<u2").reshape(-1, 16).astype(np.uint32)
    n = a.shape[0]
    limbs = np.frombuffer(raw, dtype="<u8").reshape(-1, 4)
    hi3 = (limbs[:, 1] | limbs[:, 2] | limbs[:, 3]) == 0
    zero = hi3 & (limbs[:, 0] == 0)
    bit = hi3 & (limbs[:, 0] == 1)
    lt16 = hi3 & (limbs[:, 0] < (1 << 16)) & ~zero & ~bit
    lt64 = hi3 & ~zero & ~bit & ~lt16
    lt128 = ((limbs[:, 2] | limbs[:, 3]) == 0) & ~hi3
    wide = ~(((limbs[:, 2] | limbs[:, 3]) == 0))
    nz, d0 = digits_nonzero(a)
    dens = float(nz.mean())
    print(f"{n} scalars")
    for name, m in (("zero", zero), ("one", bit), ("2 .. 2^16 - 1", lt16), ("2^16 .. 2^64 - 1", lt64), ("2^64 .. 2^128 - 1", lt128), (">= 2^128", wide)):
        print(f"  {name:20s} {100.0 * m.mean():6.2f} %")
    print(f"non-zero 16-bit digits (bucket additions) per scalar: {dens:.3f}   (bench.py --witness bench: 7.2, --witness survey: ~2.3 -> "
          f"nearer to '{'bench' if abs(dens - 7.2) < abs(dens - 2.3) else 'survey'}')")
    vals, cnt = np.unique(d0[d0 != 0], return_counts=True)
    top = np.argsort(-cnt)[:5]
    mean_load = (d0 != 0).sum() / 32768.0
    print("window 0, heaviest buckets (digit: entries; mean bucket load %.1f): " % mean_load + ", ".join(f"{int(vals[i])}: {int(cnt[i])}" for i in top))
    return dens


if __name__ == "__main__":
    if "--self-test" in sys.argv:
        rng = np.random.default_rng(1)
        x = rng.integers(0, 1 << 63, size=(100000, 4), dtype=np.uint64)
        x[:, 3] &= np.uint64((1 << 60) - 1)
        x[rng.random(100000) < 0.55] = 0
        d = report(x.tobytes())
        assert abs(d - 7.2) < 0.05, d
        print("self-test ok")
    else:
        report(open(sys.argv[1], "rb").read())

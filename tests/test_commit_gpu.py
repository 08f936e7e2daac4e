"""GPU parity: CommitmentKey::commit (src/commitment.rs:81-90) through the C-ABI vs the oracle.
Bit-exact (integer / group arithmetic)."""
import numpy as np
import pytest

from conftest import golden, h2i, seeded_scalars

pytestmark = pytest.mark.gpu


def _pts(O, cid, pts):
    bf = O.BASE_FIELD[cid]
    return O.ints_to_mont(bf, [h2i(c) for p in pts for c in p]).reshape(-1, 8)


@pytest.mark.parametrize("name,cid", [("bn256", 0), ("grumpkin", 1)])
def test_commit_golden(srs, oracle, name, cid):
    O = oracle
    g = golden("curve_msm.json")[name]
    bases = _pts(O, cid, g["bases"])
    ck = srs.CommitmentKey(cid, bases)
    sf = O.SCALAR_FIELD[cid]
    for rec in g["msm"]:
        n = rec["n"]
        sc = O.ints_to_mont(sf, [h2i(s) for s in rec["scalars"]]) if n else np.zeros((0, 4), np.uint64)
        assert np.array_equal(ck.commit(sc), _pts(O, cid, [rec["out"]])[0]), (n, rec["kind"])


@pytest.mark.parametrize("cid", [0, 1])
@pytest.mark.parametrize("n,kind", [(1, "uniform"), (63, "uniform"), (1000, "uniform"), (1000, "trace"),
                                    (4097, "uniform"), (1 << 14, "trace"), (1 << 15, "uniform")])
def test_commit_vs_oracle(srs, oracle, cid, n, kind):
    O = oracle
    bases = O.make_bases(cid, 1234 + cid, 1 << 15)
    ck = srs.CommitmentKey(cid, bases)
    sc = seeded_scalars(O, cid, n, 77 + n, kind)
    assert np.array_equal(ck.commit(sc), O.msm(cid, sc, bases[:n]))


@pytest.mark.parametrize("cid", [0, 1])
def test_commit_edge_scalars(srs, oracle, cid):
    """0, 1, q-1, 2^128-1, all-equal scalars, identity bases (SURVEY.md 8c golden list)."""
    from oracle import pyref as P
    O = oracle
    q = P.CURVES[cid].q
    sf = O.SCALAR_FIELD[cid]
    n = 2048
    bases = O.make_bases(cid, 5, n)
    bases[7] = 0                       # identity point in the key
    ck = srs.CommitmentKey(cid, bases)
    for vals in ([0] * n, [1] * n, [q - 1] * n, [(1 << 128) - 1] * n, [0x8000] * n, [0x8001] * n,
                 [(i % 3) * (q - 1) // 2 for i in range(n)]):
        sc = O.ints_to_mont(sf, vals)
        assert np.array_equal(ck.commit(sc), O.msm(cid, sc, bases))
    # canonical-representation entry
    vals = [(i * 0x9E3779B97F4A7C15) % q for i in range(n)]
    got = ck.commit(O.ints_to_limbs(vals), repr=1)
    assert np.array_equal(got, O.msm(cid, O.ints_to_mont(sf, vals), bases))


@pytest.mark.parametrize("cid", [0, 1])
def test_commit_batch_and_errors(srs, oracle, cid):
    O = oracle
    bases = O.make_bases(cid, 9, 3000)
    ck = srs.CommitmentKey(cid, bases)
    vs = [seeded_scalars(O, cid, n, 10 + i, "trace" if i % 2 else "uniform") for i, n in enumerate((3000, 17, 1024, 2999, 1, 512))]
    got = ck.commit_batch(vs)
    for g, v in zip(got, vs):
        assert np.array_equal(g, O.msm(cid, v, bases[: len(v)]))
    many = [seeded_scalars(O, cid, 100 + 7 * i, 40 + i, "uniform") for i in range(21)]       # more than one batch descriptor (16)
    for g, v in zip(ck.commit_batch(many), many):
        assert np.array_equal(g, O.msm(cid, v, bases[: len(v)]))
    assert np.array_equal(ck.commit(np.zeros((0, 4), np.uint64)), np.zeros(8, np.uint64))   # n == 0 -> identity
    with pytest.raises(srs.TooLongInput):                                                    # src/commitment.rs:82-88
        ck.commit(np.zeros((3001, 4), np.uint64))


def test_commit_device_resident_and_homomorphism(srs, oracle):
    """Witness-sized MSM on device-resident scalars; size-independent check
    commit(W1 + r W2) = commit(W1) + [r] commit(W2)  (src/nifs/sangria/mod.rs:455-474)."""
    import torch
    O = oracle
    cid = 0
    n = 12 << 14
    bases = O.make_bases(cid, 2024, n)
    ck = srs.CommitmentKey(cid, bases)
    w1, w2 = seeded_scalars(O, cid, n, 1, "trace"), seeded_scalars(O, cid, n, 2)
    r = seeded_scalars(O, cid, 1, 3)[0]
    folded = O.fold_w(O.FR, w1, w2, r)
    d = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    c1, c2, cf = ck.commit(d(w1)), ck.commit(d(w2)), ck.commit(d(folded))
    assert np.array_equal(cf, srs.point_sum(cid, np.stack([c1, srs.point_mul(cid, r, c2)])))
    assert np.array_equal(c2, O.msm(cid, w2, bases))


def test_commit_sharded_partials(srs, oracle):
    """Multi-GPU path on one device: each rank's partial over its block-cyclic stripes, summed."""
    O = oracle
    cid, n, world = 1, 5000, 4
    bases = O.make_bases(cid, 11, n)
    sc = seeded_scalars(O, cid, 4321, 5)
    parts = []
    for r in range(world):
        ck = srs.CommitmentKey(cid, bases, rank=r, world=world)
        parts.append(ck.commit(sc))
    assert np.array_equal(srs.point_sum(cid, np.stack(parts)), O.msm(cid, sc, bases[:4321]))


def _key_file_roundtrip(S, O, tmp_path):
    """CommitmentKey cache file (src/commitment.rs:99-170; reference test file_tests::consistency :199-213)."""
    cid, k = 1, 9
    key = S.CommitmentKey.setup_synthetic(cid, 1 << k, seed=77)
    path = tmp_path / "grumpkin" / f"{k}.bin"
    path.parent.mkdir(parents=True)
    key.save_to_file(path)
    raw = np.fromfile(path, dtype=np.uint64).reshape(-1, 8)
    assert np.array_equal(raw, key.bases()) and raw.shape[0] == 1 << k       # raw memory dump of [C; 2^k]
    loaded = S.CommitmentKey.load_from_file(cid, path, k)
    sc = seeded_scalars(O, cid, 300, 4)
    assert np.array_equal(loaded.commit(sc), key.commit(sc)) and loaded.count_off_curve() == 0
    again = S.CommitmentKey.load_or_setup_cache(cid, tmp_path, "grumpkin", k)
    assert np.array_equal(again.bases(), raw)
    made = S.CommitmentKey.load_or_setup_cache(cid, tmp_path, "fresh", 6, setup=lambda kk: S.CommitmentKey.setup_synthetic(cid, 1 << kk, seed=1))
    assert (tmp_path / "fresh" / "6.bin").exists() and len(made) == 64
    # a point off the curve -> InvalidData ("Wrong file in cache, some ptr out of curve", :152-158)
    bad = raw.copy()
    bad[5, 0] ^= np.uint64(1)
    bad.tofile(path)
    with pytest.raises(ValueError, match="out of curve"):
        S.CommitmentKey.load_from_file(cid, path, k)
    # short file -> read_exact error
    raw[: (1 << k) - 1].tofile(path)
    with pytest.raises(IOError):
        S.CommitmentKey.load_from_file(cid, path, k)
    with pytest.raises(IOError):
        S.CommitmentKey.load_from_file(cid, tmp_path / "missing.bin", k)


def test_key_cache_file(srs, oracle, tmp_path):
    _key_file_roundtrip(srs, oracle, tmp_path)


def test_two_host_threads_distinct_handles(srs, oracle):
    """SURVEY 8b threading contract: calls on DISTINCT handles may run concurrently from different host threads
    (HIP's current device is per thread; the library rebinds it).  Two threads hammer their own key / structure / NTTs;
    every result must equal the single-threaded one."""
    import threading
    from workloads import make_structure_inputs
    O = oracle
    jobs, errs = [], []
    for t, (which, cid) in enumerate((("primary", 0), ("secondary", 1))):
        w = make_structure_inputs(which, 10, seed=90 + t)
        ck = srs.CommitmentKey.setup_synthetic(cid, w["num_advice"] * w["rows"], seed=7 + t)
        St = srs.PlonkStructure(w["field"], 10, [], w["fixed"], w["num_advice"], w["gates"])
        a = O.to_mont(O.FR, np.random.default_rng(t).integers(0, 1 << 62, size=(1 << 12, 4), dtype=np.uint64))
        exp_c = ck.commit(w["W1"])
        exp_t, exp_tc = srs.VanillaFS.commit_cross_terms(ck, St, w["u1_challenges"], w["u1_u"], w["W1"], w["u2_challenges"], w["W2"])
        exp_f = srs.fft.fft(a.copy())
        jobs.append((w, ck, St, a, exp_c, exp_t, exp_tc, exp_f))

    def run(job):
        w, ck, St, a, exp_c, exp_t, exp_tc, exp_f = job
        try:
            for _ in range(25):
                assert np.array_equal(ck.commit(w["W1"]), exp_c)
                tt, tc = srs.VanillaFS.commit_cross_terms(ck, St, w["u1_challenges"], w["u1_u"], w["W1"], w["u2_challenges"], w["W2"])
                assert np.array_equal(tc, exp_tc) and all(np.array_equal(x, y) for x, y in zip(tt, exp_t))
                assert np.array_equal(srs.fft.fft(a.copy()), exp_f)
        except Exception as e:      # surfaced in the main thread
            errs.append(repr(e))

    th = [threading.Thread(target=run, args=(j,)) for j in jobs]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def _degenerate_bases_case(S, O, cid, n):
    """Keys with REPEATED and NEGATED points: equal points meet in one bucket (mixed add -> doubling), equal partial sums
    meet in the tail (XYZZ add / quad add -> doubling), P + (-P) cancels to the identity at every level.  The reference's
    best_multiexp is complete for all of these; so must every addition here be."""
    from oracle import pyref as P
    q = P.CURVES[cid].q
    sf = O.SCALAR_FIELD[cid]
    pool = O.make_bases(cid, 3, 4)
    neg = pool.copy()
    fld = O.BASE_FIELD[cid]
    neg[:, 4:] = O.fe_sub(fld, np.zeros_like(pool[:, 4:]), pool[:, 4:])          # (x, -y)
    rng = np.random.default_rng(n + cid)
    # (a) one point repeated n times, one scalar: every bucket entry is the same point
    bases = np.repeat(pool[:1], n, axis=0)
    ck = S.CommitmentKey(cid, bases)
    for v in (1, 5, 0x10001, q - 1, (1 << 200) + 12345):
        sc = O.ints_to_mont(sf, [v] * n)
        assert np.array_equal(ck.commit(sc), O.msm(cid, sc, bases)), ("repeat", v)
    ck.close()
    # (b) P, -P interleaved with equal scalars: total is the identity, all-zero affine encoding
    bases = np.empty((n, 8), np.uint64)
    bases[0::2], bases[1::2] = pool[1], neg[1]
    ck = S.CommitmentKey(cid, bases)
    sc = O.ints_to_mont(sf, [0x123456789ABCDEF] * n)
    got = ck.commit(sc)
    assert not got.any() and np.array_equal(got, O.msm(cid, sc, bases))
    ck.close()
    # (c) random mixture of 4 points and their negatives, few distinct scalars (collisions everywhere), batch of 3
    idx = rng.integers(0, 8, size=n)
    bases = np.where((idx < 4)[:, None], pool[idx % 4], neg[idx % 4])
    ck = S.CommitmentKey(cid, bases)
    vs = []
    for j in range(3):
        vals = [int(x) for x in rng.choice([1, 2, 3, 0xFFFF, 0x10000, q - 2, q - 1, (1 << 128) + 7], size=n)]
        vs.append(O.ints_to_mont(sf, vals))
    for g, v in zip(ck.commit_batch(vs), vs):
        assert np.array_equal(g, O.msm(cid, v, bases))
    # (d) r06: the same collisions through the STREAMED commit in slot mode (three chunks): a slot's running sum that cancelled to the identity in
    # one chunk is the start value of the next (accumulate_part's flagged chain, redone with the complete formulas), equal points meet the
    # running sum of an earlier chunk, and an identity base sits in the key
    bases[n // 3] = 0
    ck2 = S.CommitmentKey(cid, bases)
    with S.tuning(commit_chunks=3, msm_slots=2):
        for v in vs[:2]:
            assert np.array_equal(ck2.commit_upload(v), O.msm(cid, v, bases)), "streamed, degenerate key"
        bases2 = np.empty((n, 8), np.uint64)
        bases2[0::2], bases2[1::2] = pool[1], neg[1]
        ck3 = S.CommitmentKey(cid, bases2)
        sc = O.ints_to_mont(sf, [0x123456789ABCDEF] * n)
        got = ck3.commit_upload(sc)
        assert not got.any() and np.array_equal(got, O.msm(cid, sc, bases2))
        ck3.close()
    ck2.close()
    ck.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_commit_degenerate_bases(srs, oracle, cid):
    _degenerate_bases_case(srs, oracle, cid, 6000)
    _degenerate_bases_case(srs, oracle, cid, 70000)       # enough entries for several accumulation levels


@pytest.mark.parametrize("cid,n", [(0, 1 << 24), (1, 12 << 20), (0, 12 << 20)])      # the last: configs[2]'s witness commit on ITS curve (bn256)
def test_commit_full_size_properties(srs, oracle, cid, n):
    """BASELINE configs 5 (2^24-point MSM) and 3 (12 * 2^20 witness commit) at FULL size, through size-independent
    properties: additivity over a split of the vector, commit(2 v) = 2 commit(v) (the scalar doubling done by the fold
    kernel), and unit vectors hitting single bases."""
    import torch
    O = oracle
    sf = O.SCALAR_FIELD[cid]
    ck = srs.CommitmentKey.setup_synthetic(cid, n, seed=11 + cid)
    g = torch.Generator(device="cuda").manual_seed(5 + cid)
    v = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    v[:, 3] &= (1 << 60) - 1                                       # < 2^252: a valid residue of either field
    v[torch.rand(n, device="cuda", generator=g) < 0.5] = 0         # trace-like: half the scalars are zero
    C = ck.commit(v)
    assert C.any()
    lo, hi = v.clone(), v.clone()
    lo[n // 2:] = 0
    hi[: n // 2] = 0
    assert np.array_equal(srs.point_sum(cid, np.stack([ck.commit(lo), ck.commit(hi)])), C)
    one = O.ints_to_mont(sf, [1])[0]
    v2 = srs.RelaxedPlonkWitness(sf, [v], v[:1]).fold([v], [], one).W[0]         # v + 1 * v
    assert np.array_equal(ck.commit(v2), srs.point_sum(cid, np.stack([C, C])))
    bases = ck.bases()
    for j in (0, 1, n // 3, n - 1):
        e = torch.zeros_like(v)
        e[j] = torch.from_numpy(one.view(np.int64))
        assert np.array_equal(ck.commit(e), bases[j])
    ck.close()


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def test_commit_config_sizes_vs_oracle(srs, oracle):
    """BASELINE configs[4] (2^24-point MSM) and configs[2] (12 * 2^20 witness commit, bn256) at FULL size, compared DIRECTLY
    with the oracle's best_multiexp restatement (src/commitment.rs:81-90) on the key's own bases: the production pipelines at the
    sizes BASELINE names, not only forced onto small inputs --
      * the 20-bit wide windows (r04: the default again for whole device-resident MSMs of >= 2^23 scalars on keys of >= 2^23 bases):
        the two 2^24 commits and the resident 12 * 2^20 commit;
      * the streamed nine-chunk upload on the 16-bit windows in slot mode (persistent per-bucket partial sums, one reduction per
        commit; the "trace" mixture has hot buckets, so the overflow kernels and the once-per-key redo run too: msm_stats);
      * a 3-shard multi-device key (16-bit windows per shard)."""
    O = oracle
    cid = 0
    n24 = 1 << 24
    ck = srs.CommitmentKey.setup_synthetic(cid, n24, seed=11)
    bases = ck.bases()
    for kind, seed in (("uniform", 1), ("trace", 2)):                       # configs[4]
        v = seeded_scalars(O, cid, n24, seed, kind)
        assert np.array_equal(ck.commit(_dev(v)), O.msm(cid, v, bases)), ("2^24", kind)
    n = 12 << 20                                                            # configs[2]: the primary witness commit
    v = seeded_scalars(O, cid, n, 3, "trace")
    v[1::7] = seeded_scalars(O, cid, (n + 5) // 7, 4, "uniform")[: v[1::7].shape[0]]      # denser than the mixture: every chunk has full-width scalars
    want = O.msm(cid, v, bases[:n])
    d = _dev(v)
    assert np.array_equal(ck.commit(d), want), "12*2^20 commit (device resident)"
    hb = srs.HostBuffer(n)
    hb.array[:] = v
    d.zero_()
    assert np.array_equal(ck.commit_upload(hb.array, dev_copy=d), want), "12*2^20 commit_upload (page-locked, default chunks)"
    import torch
    assert torch.equal(d, _dev(v))
    assert np.array_equal(ck.commit_upload(v), want), "12*2^20 commit_upload (pageable)"
    st = ck.msm_stats()
    assert st["slot_sets"] >= 18 and st["hot_sets"] >= 1 and st["redo"] <= 1 and st["other_sets"] >= 3, st     # 2 streamed commits in slot mode, 3 wide MSMs
    hb.close()
    ck.close()
    del d
    torch.cuda.empty_cache()
    mk = srs.CommitmentKey.create_multi(cid, bases[:n], 3)                  # scalars partitioned over 3 shards
    assert np.array_equal(mk.commit(v), want), "12*2^20 commit on a 3-shard key"
    assert np.array_equal(mk.commit_upload(v), want), "12*2^20 commit_upload on a 3-shard key"
    mk.close()


def test_commit_config_k22_vs_oracle(srs, oracle):
    """BASELINE configs[3]: the 12 * 2^22 = 50 M-scalar witness commit on a 2^26 key (64 GiB of window tables), device
    resident and streamed from host memory, against the oracle."""
    import torch
    O = oracle
    cid, n = 0, 12 << 22
    ck = srs.CommitmentKey.setup_synthetic(cid, 1 << 26, seed=12)
    bases = ck.bases()[:n]
    v = seeded_scalars(O, cid, n, 5, "trace")
    want = O.msm(cid, v, bases)
    del bases
    d = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    assert np.array_equal(ck.commit_upload(v, dev_copy=d), want), "12*2^22 commit_upload"
    assert np.array_equal(ck.commit(d), want), "12*2^22 commit (device resident)"
    ck.close()


def test_two_pass_scatter_matches_single_pass(srs, oracle):
    """The MSD two-pass scatter (k_group + k_scatter2, used from 2^23 digit slots on) against the single-pass one and the
    oracle on sizes the oracle can do: forced through the tunable msm_sort (a subprocess per mode)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O, sirius_amd as S\n"
        "from conftest import seeded_scalars\n"
        "for cid, n, kind in ((0, 70000, 'uniform'), (1, 33333, 'trace'), (0, 5, 'uniform')):\n"
        "    bases = O.make_bases(cid, 4, n); ck = S.CommitmentKey(cid, bases)\n"
        "    vs = [seeded_scalars(O, cid, n, 9 + j, kind) for j in range(3)]\n"
        "    for g, v in zip(ck.commit_batch(vs), vs): assert np.array_equal(g, O.msm(cid, v, bases))\n"
        "# heavy buckets (every digit in a few hundred buckets: a tile of the grouped array holds 1-2 buckets) next to uniform scalars\n"
        "import random; rnd = random.Random(5)\n"
        "dense = lambda lo, hi: sum(rnd.randrange(lo, hi) << (16 * w) for w in range(15))\n"
        "vals = [dense(1, 200) for _ in range(30000)] + [dense(0xFF00, 0xFFFF) for _ in range(9000)] + [3] * 7000\n"
        "vd = np.concatenate([O.ints_to_mont(O.SCALAR_FIELD[0], vals), seeded_scalars(O, 0, 20000, 4, 'uniform')])\n"
        "bases = O.make_bases(0, 4, len(vd)); ck = S.CommitmentKey(0, bases)\n"
        "assert np.array_equal(ck.commit(vd), O.msm(0, vd, bases))\n"
        "print('ok')\n")
    from conftest import ROOT, tune_env
    for mode in (1, 2):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=tune_env(msm_sort=mode), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (mode, r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("cid", [0, 1])
def test_commit_upload_chunked(srs, oracle, cid):
    """srs_commit_upload (witness straight from host memory, chunked upload overlapped with the MSM of the chunks already in
    HBM, device copy left behind) == the oracle's commitment, for every chunk count incl. ragged last chunks, pageable and
    page-locked sources; the device copy equals the source."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, tune_env
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O, sirius_amd as S\n"
        "from conftest import seeded_scalars\n"
        f"cid = {cid}\n"
        "for n, kind in ((70001, 'uniform'), (4096, 'trace'), (1, 'uniform'), (3000, 'uniform')):\n"
        "    bases = O.make_bases(cid, 4, n + 5); ck = S.CommitmentKey(cid, bases)\n"
        "    v = seeded_scalars(O, cid, n, 19, kind); want = O.msm(cid, v, bases[:n])\n"
        "    hb = S.HostBuffer(n); hb.array[:] = v\n"
        "    d = torch.zeros((n, 4), dtype=torch.int64, device='cuda')\n"
        "    assert np.array_equal(ck.commit_upload(v), want)\n"
        "    assert np.array_equal(ck.commit_upload(hb.array, dev_copy=d), want)\n"
        "    assert np.array_equal(d.cpu().numpy().view(np.uint64), v)\n"
        "    assert np.array_equal(ck.commit(d), want)\n"
        "    try:\n"
        "        ck.commit_upload(np.zeros((n + 6, 4), np.uint64)); raise SystemExit('no TooLongInput')\n"
        "    except S.TooLongInput: pass\n"
        "    ck.close(); hb.close()\n"
        "print('ok')\n")
    for chunks in ("1", "2", "3", "7", "16"):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=tune_env(commit_chunks=chunks), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (chunks, r.stdout[-500:], r.stderr[-1500:])


def test_commit_upload_default_chunking_large(srs, oracle):
    """Default chunking (4 chunks from 2^22 scalars on) at a size with real overlap: additivity against plain commits."""
    import torch
    n = (5 << 20) + 12345
    ck = srs.CommitmentKey.setup_synthetic(0, n, seed=3)
    g = torch.Generator(device="cuda").manual_seed(8)
    v = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    v[:, 3] &= (1 << 60) - 1
    v[torch.rand(n, device="cuda", generator=g) < 0.5] = 0
    want = ck.commit(v)
    hb = srs.HostBuffer(n)
    hb.array[:] = v.cpu().numpy().view(np.uint64)
    d = torch.empty_like(v)
    assert np.array_equal(ck.commit_upload(hb.array, dev_copy=d), want)
    assert torch.equal(d, v)
    assert np.array_equal(ck.commit_upload(hb.array), want)          # library staging, no copy kept
    ck.close()
    hb.close()


@pytest.mark.parametrize("cid", [0, 1])
def test_multi_device_key_single_process(srs, oracle, cid):
    """srs_ck_create_multi: ONE process, the library spreads the key over several shards (2, 3, 5 logical shards folded onto
    the visible device(s)) and partitions the scalars of every commit; the caller sees the FULL commitment.  Host and
    device-resident scalars, batches, ragged lengths (partial stripes, fewer stripes than shards), key read-back."""
    import torch
    O = oracle
    n = 5 * 1024 + 777
    bases = O.make_bases(cid, 21, n)
    vs = [seeded_scalars(O, cid, m, 30 + j, kind) for j, (m, kind) in enumerate(((n, "uniform"), (2049, "trace"), (1, "uniform"), (1024, "uniform")))]
    want = [O.msm(cid, v, bases[: v.shape[0]]) for v in vs]
    for shards in (2, 3, 5):
        ck = srs.CommitmentKey.create_multi(cid, bases, shards)
        assert ck.num_shards == shards and len(ck) == n
        for v, w in zip(vs, want):
            assert np.array_equal(ck.commit(v), w)
            assert np.array_equal(ck.commit(torch.from_numpy(v.view(np.int64)).cuda()), w)
            assert np.array_equal(ck.commit_upload(v), w)
        assert np.array_equal(ck.commit_batch(vs), np.stack(want))
        assert np.array_equal(ck.commit_batch([torch.from_numpy(v.view(np.int64)).cuda() for v in vs]), np.stack(want))
        assert np.array_equal(ck.commit(vs[0][:0]), np.zeros(8, np.uint64))
        assert np.array_equal(ck.bases(), bases) and ck.count_off_curve() == 0
        with pytest.raises(srs.TooLongInput):
            ck.commit(np.zeros((n + 1, 4), np.uint64))
        ck.close()
    # r05: a streamed commit on a multi-device key -- every shard uploads ITS stripes over its own link in chunks that overlap its MSM,
    # the device copy is assembled on the process's device by peer copies; bytes per link asserted through srs_ck_shard_stats
    n2 = 37 * 1024 + 555
    bases2 = O.make_bases(cid, 22, n2)
    for kind in ("uniform", "trace"):
        v = seeded_scalars(O, cid, n2, 41, kind)
        want2 = O.msm(cid, v, bases2)
        for shards in (2, 3, 8):
            ck = srs.CommitmentKey.create_multi(cid, bases2, shards)
            d = torch.zeros((n2, 4), dtype=torch.int64, device="cuda")
            for rep in range(2):
                d.zero_()
                assert np.array_equal(ck.commit_upload(v, dev_copy=d), want2), (kind, shards, rep)
                torch.cuda.synchronize()
                assert np.array_equal(d.cpu().numpy().view(np.uint64), v), (kind, shards, rep)
            st = [ck.shard_stats(j) for j in range(shards)]
            stripes = lambda j: sum(min(1024, n2 - s * 1024) for s in range(j, (n2 + 1023) // 1024, shards))
            for j in range(shards):
                assert st[j]["streamed_commits"] == 2 and st[j]["h2d_bytes"] == 2 * stripes(j) * 32, (j, st[j])
                assert st[j]["peer_bytes"] == (0 if j == 0 else 2 * stripes(j) * 32), (j, st[j])
            assert sum(x["h2d_bytes"] for x in st) == 2 * n2 * 32
            ck.close()
    # synthetic multi key == synthetic single key (same seeded bases, whatever the sharding)
    a = srs.CommitmentKey.setup_synthetic(cid, 3000, seed=5)
    b = srs.CommitmentKey.setup_synthetic_multi(cid, 3000, seed=5, n_devices=3)
    assert np.array_equal(a.bases(), b.bases())
    assert np.array_equal(a.commit(vs[1]), b.commit(vs[1]))
    a.close(); b.close()


def test_multi_device_key_in_a_prove(srs, oracle):
    """A multi-device key behind commit_cross_terms: the cross terms are produced on the process's device and every shard
    fetches its stripes with a (peer) copy -- same commitments as the single-device key."""
    from workloads import make_structure_inputs
    O = oracle
    k = 11
    w = make_structure_inputs("secondary", k, seed=77)
    St = srs.PlonkStructure(w["field"], k, [], w["fixed"], w["num_advice"], w["gates"])
    bases = O.make_bases(1, 5, 1 << k)
    ck1, ck3 = srs.CommitmentKey(1, bases), srs.CommitmentKey.create_multi(1, bases, 3)
    args = (St, w["u1_challenges"], w["u1_u"], w["W1"], w["u2_challenges"], w["W2"])
    t1, c1 = srs.VanillaFS.commit_cross_terms(ck1, *args)
    t3, c3 = srs.VanillaFS.commit_cross_terms(ck3, *args)
    assert np.array_equal(c1, c3) and all(np.array_equal(a, b) for a, b in zip(t1, t3))
    import torch
    dv = lambda a: torch.from_numpy(a.view(np.int64)).cuda()
    t3d, c3d = srs.VanillaFS.commit_cross_terms(ck3, St, w["u1_challenges"], w["u1_u"], dv(w["W1"]), w["u2_challenges"], dv(w["W2"]))
    assert np.array_equal(c1, c3d)
    ck1.close(); ck3.close(); St.close()


def test_long_level0_parts_match_oracle(srs, oracle):
    """Large MSMs give every level-0 thread up to 128 gathered additions (msm.hip l0_log_for); forced here through
    the tunable msm_l0 on sizes the oracle can do (a subprocess per value), skewed and uniform scalars."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O, sirius_amd as S\n"
        "from conftest import seeded_scalars\n"
        "for cid, n, kind in ((0, 70000, 'uniform'), (1, 33333, 'trace'), (0, 5, 'uniform')):\n"
        "    bases = O.make_bases(cid, 4, n); ck = S.CommitmentKey(cid, bases)\n"
        "    vs = [seeded_scalars(O, cid, n, 9 + j, kind) for j in range(2)]\n"
        "    for g, v in zip(ck.commit_batch(vs), vs): assert np.array_equal(g, O.msm(cid, v, bases))\n"
        "print('ok')\n")
    from conftest import ROOT, tune_env
    for l0 in (5, 7):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=tune_env(msm_l0=l0), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (l0, r.stdout[-500:], r.stderr[-1500:])


WIDE_CODE = (
    "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
    "import oracle as O, sirius_amd as S\n"
    "from oracle import pyref as P\n"
    "from conftest import seeded_scalars\n"
    "for cid, n, kind in ((0, 70000, 'uniform'), (1, 33333, 'trace'), (0, 5, 'uniform'), (1, 1, 'uniform')):\n"
    "    bases = O.make_bases(cid, 4, n); ck = S.CommitmentKey(cid, bases)\n"
    "    for j in range(2):\n"
    "        v = seeded_scalars(O, cid, n, 9 + j, kind); assert np.array_equal(ck.commit(v), O.msm(cid, v, bases)), (cid, n, kind)\n"
    "    if n > 1000: assert np.array_equal(ck.commit_upload(v), O.msm(cid, v, bases))     # commit_chunks = 3: chunks slide the base offset\n"
    "for cid in (0, 1):\n"
    "    q = P.CURVES[cid].q\n"
    "    vals = [0, 1, 2, q - 1, q - 2, 1 << 19, (1 << 19) + 1, (1 << 19) - 1, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 1 << 253, (1 << 240) - 1,\n"
    "            (1 << 40) - 1, ((1 << 20) - 1) << 20, (q - 1) // 2, (q + 1) // 2, 0x80000 << 20, 0x80001 << 40, 0x7ffff << 60] * 3\n"
    "    v = O.ints_to_mont(O.SCALAR_FIELD[cid], vals)\n"
    "    bases = O.make_bases(cid, 6, len(vals)); ck = S.CommitmentKey(cid, bases)\n"
    "    assert np.array_equal(ck.commit(v), O.msm(cid, v, bases))\n"
    "    for x in (0, 1, q - 1, 0x80000, 0x80001):           # every entry in ONE bucket of one segment\n"
    "        v = O.ints_to_mont(O.SCALAR_FIELD[cid], [x] * 3000); b2 = O.make_bases(cid, 7, 3000); ck = S.CommitmentKey(cid, b2)\n"
    "        assert np.array_equal(ck.commit(v), O.msm(cid, v, b2)), x\n"
    "print('ok')\n")


def test_wide_windows_match_oracle(srs, oracle):
    """The 13 x 20-bit window pipeline of large MSMs (msm.hip, `wide`), forced on sizes the oracle can do (tunables msm_wide /
    msm_wide_min, in a subprocess): uniform and skewed scalars, both curves, digit-boundary values,
    every entry in one bucket, chunked uploads (base offsets)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, tune_env
    r = subprocess.run([sys.executable, "-c", WIDE_CODE], cwd=ROOT, env=tune_env(msm_wide=1, msm_wide_min=0, commit_chunks=3),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_wide_windows_equal_narrow_at_scale(srs, oracle):
    """3 * 2^20 + 77 trace-like scalars: the wide pipeline (msm_wide = 1) against the 16-bit-window pipeline
    (msm_wide = 0), one subprocess each -- same seeded inputs, same affine point."""
    import os
    import subprocess
    import sys
    import torch
    from conftest import ROOT, tune_env
    code = (
        "import sys, numpy as np, torch; sys.path.insert(0, '.')\n"
        "import sirius_amd as S\n"
        "n = (3 << 20) + 77\n"
        "ck = S.CommitmentKey.setup_synthetic(0, n, seed=21)\n"
        "g = torch.Generator(device='cuda').manual_seed(8)\n"
        "v = torch.randint(0, 1 << 62, (n, 4), dtype=torch.int64, device='cuda', generator=g); v[:, 3] &= (1 << 60) - 1\n"
        "v[torch.rand(n, device='cuda', generator=g) < 0.5] = 0\n"
        "v[torch.rand(n, device='cuda', generator=g) < 0.2, 1:] = 0\n"
        "print('C', ck.commit(v).tobytes().hex())\n")
    outs = []
    for wide in (0, 1):
        env = tune_env(msm_wide=wide, msm_wide_min=20)
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("C ")][-1])
    assert outs[0] == outs[1] and len(outs[0]) > 100


def _concat_ref(cols, pad):
    """util::concatenate_with_padding (src/util/mod.rs:214-218): every vector followed by zeros up to pad_size."""
    out = []
    for c in cols:
        out.append(np.asarray(c, dtype=np.uint64).reshape(-1, 4))
        if out[-1].shape[0] < pad:
            out.append(np.zeros((pad - out[-1].shape[0], 4), np.uint64))
    return np.ascontiguousarray(np.concatenate(out)) if out else np.zeros((0, 4), np.uint64)


def _concat_cases(S, O, cid, make_dev, k=11):
    """The reference's own unit tests of concatenate_with_padding (src/util/mod.rs:233-290: empty input, padding of one vector,
    perfect fit, several ragged vectors, pad_size = 1 with a longer vector) on the device, then the column-wise streamed
    witness commit `ck.commit(&concatenate_with_padding(advice, 2^k))` (src/plonk/mod.rs:441-447) against the oracle, with columns
    SHORTER than 2^k (and one empty), for several chunk layouts."""
    sf = O.SCALAR_FIELD[cid]
    fp = lambda *v: O.ints_to_mont(sf, list(v))
    for cols, pad in (([], 4), ([fp(1, 2)], 4), ([fp(1, 2, 3, 4)], 4), ([fp(1, 2), fp(3), fp(4, 5, 6)], 4), ([fp(1), fp(2, 3)], 1)):
        want = _concat_ref(cols, pad)
        d = make_dev(want.shape[0])
        S.concatenate_with_padding(cols, pad, d)
        assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(-1, 4), want), (len(cols), pad)
    rows = 1 << k
    bases = O.make_bases(cid, 31, 5 * rows)
    ck = S.CommitmentKey(cid, bases)
    v = seeded_scalars(O, cid, 5 * rows, 41, "trace")
    cols = [v[:rows], v[rows:rows + (3 * rows) // 4], v[3:3], v[2 * rows:3 * rows], v[4 * rows:4 * rows + 17]]      # full, short, empty, full, very short
    W = _concat_ref(cols, rows)
    want = O.msm(cid, W, bases[: W.shape[0]])
    d = make_dev(W.shape[0])
    assert np.array_equal(ck.commit_upload_columns(cols, rows, dev_copy=d), want)
    assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(-1, 4), W)
    assert np.array_equal(ck.commit_upload_columns(cols, rows), want)
    assert np.array_equal(ck.commit(d), want)
    with pytest.raises(S.TooLongInput):
        ck.commit_upload_columns(cols + [v[:rows]], rows)
    mk = S.CommitmentKey.create_multi(cid, bases, 3)
    assert np.array_equal(mk.commit_upload_columns(cols, rows), want)
    mk.close()
    ck.close()
    # r05: the column form STREAMS on sharded keys too -- every shard / rank uploads its own stripes of the columns and memsets its stripes of the
    # padding (upload_stripes): (a) a multi-device key with a device copy assembled by peer copies, bytes per link = the data elements of the
    # shard's stripes; (b) a key sharded over "processes": rank r's call returns its partial commitment and fills ITS stripes of its buffer
    n, SL = W.shape[0], 1024
    data = np.zeros(n, dtype=bool)
    at = 0
    for c in cols:
        data[at:at + c.shape[0]] = True
        at += max(c.shape[0], rows)
    for shards in (2, 3):
        mk = S.CommitmentKey.create_multi(cid, bases, shards)
        d = make_dev(n)
        assert np.array_equal(mk.commit_upload_columns(cols, rows, dev_copy=d), want), shards
        assert np.array_equal(d.cpu().numpy().view(np.uint64).reshape(-1, 4), W), shards
        if n >= shards << 11:                # the streamed path (short vectors take the one-copy path and report no traffic)
            for j in range(shards):
                mine = sum(int(data[s * SL:(s + 1) * SL].sum()) for s in range(j, (n + SL - 1) // SL, shards))
                assert mk.shard_stats(j)["h2d_bytes"] == mine * 32, (shards, j, mk.shard_stats(j), mine)
        mk.close()
    world = 2
    parts = []
    for r in range(world):
        rk = S.CommitmentKey(cid, bases, rank=r, world=world)
        d = make_dev(n)
        parts.append(rk.commit_upload_columns(cols, rows, dev_copy=d))
        got = d.cpu().numpy().view(np.uint64).reshape(-1, 4)
        for s_ in range(r, (n + SL - 1) // SL, world):
            assert np.array_equal(got[s_ * SL:(s_ + 1) * SL], W[s_ * SL:(s_ + 1) * SL]), (r, s_)
        rk.close()
    assert np.array_equal(S.point_sum(cid, np.stack(parts)), want)


@pytest.mark.parametrize("cid", [0, 1])
def test_concatenate_with_padding_and_column_commit(srs, oracle, cid):
    import torch
    _concat_cases(srs, oracle, cid, lambda n: torch.full((n, 4), 7, dtype=torch.int64, device="cuda"))


def test_commit_vs_eip196_known_answers(srs, oracle):
    """CommitmentKey::commit on the device (n = 1, 2: ecMul / ecAdd as MSMs over keys made of the vectors' points) and the
    host group entries against the EIP-196 known answers -- an authority outside this repository for the bn256 arithmetic
    every MSM kernel is built from (9 x 29-bit mixed / full additions, doublings through equal points, the identity)."""
    import eip196_cases as E

    def msm(scalars, bases):
        ck = srs.CommitmentKey(0, bases)
        try:
            return ck.commit(np.ascontiguousarray(scalars))
        finally:
            ck.close()
    E.check_adder(oracle, lambda a, b: srs.point_sum(0, np.stack([a, b])), lambda k, p: srs.point_mul(0, k, p), msm)


def test_sharded_commit_upload_chunked(srs, oracle):
    """srs_commit_upload on a process-sharded key streams the rank's stripes in chunks (upload of chunk j + 1 under the MSM of chunk j);
    the ranks' partial commitments sum to the oracle's, foreign stripes of the device copy stay untouched (tests/test_emu_logic.py)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, tune_env
    from test_emu_logic import SHARDED_UPLOAD_CODE
    code = "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\nimport sirius_amd as S\n" + SHARDED_UPLOAD_CODE
    for chunks in ("1", "3"):
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=tune_env(commit_chunks=chunks), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (chunks, r.stdout[-500:], r.stderr[-1500:])

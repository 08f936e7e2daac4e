"""CPU: runs the product kernels' *logic* (sort, segmented accumulation, bucket reduction, NTT passes,
row programs) on the host through tests/emu/hipemu.h and checks them against the oracle.
This is a development aid for a GPU-less container -- the emulated build is a separate test
library (tests/emu/libsirius_emu.so), never the product."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, seeded_scalars, tune_env

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libsirius_emu.so")


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    import sirius_amd as S
    from sirius_amd import _lib
    _lib.load(EMU_LIB)
    yield S
    _lib._lib = None          # the real library is (re)loaded lazily by later tests


def _run_all(jobs, timeout=900):
    """Independent subprocesses side by side (the emulator runs one kernel at a time per process): [(tag, argv, env)] -> {tag: CompletedProcess}"""
    procs = [(tag, subprocess.Popen(argv, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for tag, argv, env in jobs]
    out = {}
    for tag, p in procs:
        so, se = p.communicate(timeout=timeout)
        out[tag] = subprocess.CompletedProcess(p.args, p.returncode, so, se)
    return out


def test_emu_commit(emu, oracle):
    O = oracle
    for cid, n, kind in ((0, 300, "uniform"), (1, 300, "trace"), (0, 1, "uniform")):
        bases = O.make_bases(cid, 7 + cid, 300)
        ck = emu.CommitmentKey(cid, bases)
        sc = seeded_scalars(O, cid, n, 5, kind)
        assert np.array_equal(ck.commit(sc), O.msm(cid, sc, bases[:n]))
        ck.close()


def test_emu_ntt(emu, oracle):
    O = oracle
    rng = np.random.default_rng(1)
    for k in (0, 3, 7, 11, 13):
        raw = rng.integers(0, 1 << 63, size=(1 << k, 4), dtype=np.uint64)
        raw[:, 3] &= np.uint64((1 << 60) - 1)
        a = O.to_mont(O.FR, raw)
        for fn in ("fft", "ifft", "coset_fft", "coset_ifft"):
            assert np.array_equal(getattr(emu.fft, fn)(a.copy()), getattr(O, fn)(a)), (k, fn)


def test_emu_ntt_lazy_tile_extremes(emu, oracle):
    """The multi-pass kernels on the lazy 9 x 29-bit tile (the default since r04): every digit split the plans produce up to 2^16 (widths
    4..8, odd widths take the single trailing stage), with inputs that drive the lazy bounds -- every element the largest canonical bit
    pattern p - 1, and alternating -- against the oracle, all four transforms."""
    O = oracle
    from oracle import pyref as P
    rng = np.random.default_rng(2)
    pm1 = np.array([[((P.MODULI[0] - 1) >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]], dtype=np.uint64)
    for k in (11, 12, 14, 15, 16):
        raw = rng.integers(0, 1 << 63, size=(1 << k, 4), dtype=np.uint64)
        raw[:, 3] &= np.uint64((1 << 60) - 1)
        a = O.to_mont(O.FR, raw)
        cases = [a]
        if k in (12, 16):
            cases.append(np.repeat(pm1, 1 << k, axis=0))
            z = a.copy()
            z[::2] = pm1
            cases.append(z)
        for ci, x in enumerate(cases):
            for fn in ("fft", "ifft", "coset_fft", "coset_ifft"):
                assert np.array_equal(getattr(emu.fft, fn)(x.copy()), getattr(O, fn)(x)), (k, ci, fn)


def test_emu_cross_terms(emu, oracle):
    O = oracle
    from oracle import expr as OE
    from workloads import gates_for, rand_fe
    for field, k, gate_T in ((0, 4, [5, 3]), (1, 3, [5])):
        rows = 1 << k
        gates, nfix, nadv = gates_for(gate_T)
        og, fo, ao = [], 0, 0
        for T in gate_T:
            og.append(OE.main_gate_expression(T, 0, fo, ao, nfix)); fo += 2 * T + 5; ao += T + 2
        rng = np.random.default_rng(k)
        fixed = [rand_fe(rng, rows, 0.3) for _ in range(nfix)]
        W1, W2 = rand_fe(rng, nadv * rows), rand_fe(rng, nadv * rows)
        S = emu.PlonkStructure(field, k, [], fixed, nadv, gates)
        nch = S.num_challenges
        u1c, u1u, u2c = rand_fe(rng, nch), rand_fe(rng, 1)[0], rand_fe(rng, nch)
        terms, _ = emu.VanillaFS.commit_cross_terms(None, S, u1c, u1u, W1, u2c, W2)
        ch = emu.VanillaFS.cross_term_challenges(u1c, u1u, u2c, field)
        cg, exp = OE.cross_terms_oracle(O, field, og, 0, nfix, nadv, [], fixed, W1, W2, ch)
        assert len(terms) == cg.degree
        for a, b in zip(terms, exp):
            assert np.array_equal(a, b)
        r, E = rand_fe(rng, 1)[0], rand_fe(rng, rows)
        acc = emu.RelaxedPlonkWitness(field, [W1], E).fold([W2], terms, r)
        assert np.array_equal(acc.W[0], O.fold_w(field, W1, W2, r)) and np.array_equal(acc.E, O.fold_e(field, E, exp, r))
        S.close()


def test_emu_protogalaxy(emu, oracle):
    from pg_cases import run_pg_case
    run_pg_case(emu, oracle, 4, [2], 1, True)
    run_pg_case(emu, oracle, 3, [5, 3, 2], 1, False)
    run_pg_case(emu, oracle, 8, [5, 3], 1, True)
    run_pg_case(emu, oracle, 4, [2], 3, False)
    run_pg_case(emu, oracle, 4, [2], 7, False)             # r05: L + 1 up to 16 (JMAX)
    run_pg_case(emu, oracle, 5, [3, 2], 15, True)
    with pytest.raises(emu.SiriusAmdError, match="unsupported number of traces"):      # 32 instances: refused, not truncated
        run_pg_case(emu, oracle, 4, [2], 31, False)


def test_emu_protogalaxy_K_domain_2p16_and_high_degree(emu, oracle):
    """r06: K domains above 256 points (k_pg_K_points + the 2^16-point coset_ifft: three incoming traces = the reference's own test
    shape, src/nifs/protogalaxy/tests.rs:187-309; a gate of degree 9 / 15 with one incoming trace), the transcript variant of the
    one-call prove over K's 65536 coefficients, and the refusal of a K "log" above F::S (degree 16: 32 points of G)."""
    from pg_cases import high_degree_gates, run_pg_case
    ctx = run_pg_case(emu, oracle, 4, [2], 3, False, ro_check=True)
    assert (ctx.fft_points_count_G, ctx.fft_log_domain_size_K) == (16, 16)
    ctx = run_pg_case(emu, oracle, 4, None, 1, True, gates=high_degree_gates(9))
    assert (ctx.fft_points_count_G, ctx.fft_log_domain_size_K) == (16, 16)
    run_pg_case(emu, oracle, 3, None, 1, False, gates=high_degree_gates(15))
    ctx = run_pg_case(emu, oracle, 3, None, 1, False, gates=high_degree_gates(16))
    assert ctx.fft_log_domain_size_K == 32


def test_emu_protogalaxy_polynomial_tree_F(emu, oracle):
    """k >= 10: compute_F runs the polynomial tree (k_pg_F_leaves / k_pg_F_level), interpreter and specialised leaves."""
    from pg_cases import run_pg_case
    run_pg_case(emu, oracle, 10, [2], 1, False)
    run_pg_case(emu, oracle, 10, [5, 3], 1, True)


def test_emu_protogalaxy_direct_eval(emu, oracle):
    from pg_cases import direct_eval_case
    direct_eval_case(emu, oracle, 4, [3, 2], True)
    direct_eval_case(emu, oracle, 4, [2], False)


def test_emu_key_file_and_deciders(emu, oracle, tmp_path):
    from test_commit_gpu import _key_file_roundtrip
    from test_sangria_gpu import _is_sat_case
    _key_file_roundtrip(emu, oracle, tmp_path)
    _is_sat_case(emu, oracle, 0, 4, (5, 3))
    _is_sat_case(emu, oracle, 1, 5, (2,))


def test_emu_fold_then_decider(emu, oracle):
    from test_sangria_gpu import _fold_then_decide
    _fold_then_decide(emu, oracle, 0, 0, 4, (5, 3))
    _fold_then_decide(emu, oracle, 1, 1, 3, (2,))


def test_emu_protogalaxy_fold_identity(emu, oracle):
    from test_protogalaxy_gpu import _pg_fold_identity
    _pg_fold_identity(emu, oracle, 4, (5, 3), True)
    _pg_fold_identity(emu, oracle, 3, (2,), False)


def test_emu_lookup_arguments(emu, oracle):
    from lookup_cases import run_lookup_case
    run_lookup_case(emu, oracle, "vector", 5)
    run_lookup_case(emu, oracle, "scalar", 4)
    run_lookup_case(emu, oracle, "two", 4)


def test_emu_high_folding_degree(emu, oracle):
    from test_sangria_gpu import _high_degree_case
    _high_degree_case(emu, oracle, 0, 3, 10)
    _high_degree_case(emu, oracle, 1, 3, 12)


def test_emu_deciders(emu, oracle):
    from test_deciders_gpu import _general_sparse_case, _permutation_case, _witness_commit_case
    _permutation_case(emu, oracle, 0, 5, 3, 2)
    _general_sparse_case(emu, oracle, 1, 40, 200, 3)
    _witness_commit_case(emu, oracle, 1, 200, 96)


def test_emu_commit_degenerate_bases(emu, oracle):
    from test_commit_gpu import _degenerate_bases_case
    _degenerate_bases_case(emu, oracle, 1, 200)


def test_emu_row_sharded_cross_terms(emu, oracle):
    from test_sangria_gpu import _row_shard_case
    _row_shard_case(emu, oracle, 1, 1, 12, (2,), 2, with_commit=False)
    _row_shard_case(emu, oracle, 0, 0, 12, (5, 3), 3, with_commit=False)


def test_emu_batch_invert_assigned(emu, oracle):
    from test_lookup_gpu import _assigned_case
    _assigned_case(emu, oracle, 0, 300)
    _assigned_case(emu, oracle, 1, 1)


def test_emu_two_pass_scatter():
    """tuning msm_sort = 2 (the two-pass scatter of large MSMs incl. the XCD-aware tile mapping) on a small MSM, on the emulator."""
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O\n"
        "from sirius_amd import _lib; _lib.load('tests/emu/libsirius_emu.so')\n"
        "import sirius_amd as S\n"
        "from conftest import seeded_scalars\n"
        "bases = O.make_bases(1, 4, 9000); ck = S.CommitmentKey(1, bases)\n"
        "for n in (9000, 4097, 3):\n"
        "    v = seeded_scalars(O, 1, n, n, 'uniform')\n"
        "    assert np.array_equal(ck.commit(v), O.msm(1, v, bases[:n]))\n"
        "# dense buckets: every 16-bit digit in a few hundred buckets (tiles of the grouped array inside one or two segments: the LDS\n"
        "# path of k_scatter2), alone and mixed with uniform scalars (tiles over many segments: entry by entry), negative digits too\n"
        "import random; rnd = random.Random(5)\n"
        "dense = lambda lo, hi: sum(rnd.randrange(lo, hi) << (16 * w) for w in range(15))\n"
        "vals = [dense(1, 200) for _ in range(3000)] + [dense(250, 300) for _ in range(1500)] + [dense(0xFF00, 0xFFFF) for _ in range(700)]\n"
        "vd = O.ints_to_mont(O.SCALAR_FIELD[1], vals)\n"
        "assert np.array_equal(ck.commit(vd), O.msm(1, vd, bases[:len(vals)]))\n"
        "vm = np.concatenate([vd[:2500], seeded_scalars(O, 1, 1500, 7, 'uniform'), seeded_scalars(O, 1, 600, 8, 'trace')])\n"
        "assert np.array_equal(ck.commit(vm), O.msm(1, vm, bases[:len(vm)]))\n"
        "# extremes: one bucket holds everything / nothing to sort / one entry\n"
        "q = __import__('oracle.pyref', fromlist=['x']).CURVES[1].q\n"
        "for vals in ([1] * 9000, [0] * 5000, [0] * 4999 + [q - 1], [(1 << 16) - 1] * 700 + [1 << 255 >> 2] * 9):\n"
        "    ve = O.ints_to_mont(O.SCALAR_FIELD[1], [x % q for x in vals])\n"
        "    assert np.array_equal(ck.commit(ve), O.msm(1, ve, bases[:len(vals)])), vals[-1]\n"
        "print('ok')\n")
    # msm_sort = 2: the two-pass flow (k_hist / k_scan_seg / k_group / k_scatter2)
    jobs = [("two_pass", [sys.executable, "-c", code], tune_env(msm_sort=2))]
    for tag, r in _run_all(jobs, timeout=1800).items():
        assert r.returncode == 0 and "ok" in r.stdout, (tag, r.stdout[-300:], r.stderr[-1500:])


def test_emu_commit_upload_and_multi_device_key(emu, oracle):
    """Host logic of srs_commit_upload (chunks with sliding base offsets, partial sums added on the host) and of the
    multi-device key (stripe partition of bases AND scalars, strided copies, partials per shard), on the emulator."""
    O = oracle
    with emu.tuning(commit_chunks=3):
        _commit_upload_and_multi_device_key(emu, O)


def _commit_upload_and_multi_device_key(emu, O):
    for cid, shard_counts in ((1, (3,)),):
        bases = O.make_bases(cid, 7 + cid, 1100)
        sc = seeded_scalars(O, cid, 1097, 5, "trace")          # 1 whole stripe + a tail
        want = O.msm(cid, sc, bases[:1097])
        ck = emu.CommitmentKey(cid, bases)
        assert np.array_equal(ck.commit_upload(sc), want)
        dev = np.zeros_like(sc)
        import torch
        d = torch.from_numpy(dev.view(np.int64))
        assert np.array_equal(ck.commit_upload(sc, dev_copy=d), want) and np.array_equal(dev, sc)
        ck.close()
        for shards in shard_counts:
            mk = emu.CommitmentKey.create_multi(cid, bases, shards)
            assert mk.num_shards == shards
            assert np.array_equal(mk.commit(sc), want)
            assert np.array_equal(mk.commit_upload(sc), want)
            assert np.array_equal(mk.commit_batch([sc, sc[:1000], sc[:0]])[1], O.msm(cid, sc[:1000], bases[:1000]))
            assert np.array_equal(mk.bases(), bases) and mk.count_off_curve() == 0
            mk.close()
    # r05: srs_commit_upload on a multi-device key streams every shard's OWN stripes over its own link (chunks overlapped with the shard's
    # MSM) and assembles the device copy on the process's device with peer copies: result, device copy and the bytes per link
    cid, n, shards = 0, 6500, 3
    bases = O.make_bases(cid, 9, n)
    sc = seeded_scalars(O, cid, n, 6, "trace")
    want = O.msm(cid, sc, bases)
    mk = emu.CommitmentKey.create_multi(cid, bases, shards)
    import torch
    for rep in range(2):
        dev = np.zeros_like(sc)
        assert np.array_equal(mk.commit_upload(sc, dev_copy=torch.from_numpy(dev.view(np.int64))), want), rep
        assert np.array_equal(dev, sc), rep
    assert np.array_equal(mk.commit_upload(sc), want)                          # no device copy asked for: nothing is forwarded
    st = [mk.shard_stats(d) for d in range(shards)]
    stripes = lambda d: sum(min(1024, n - s * 1024) for s in range(d, (n + 1023) // 1024, shards))
    for d in range(shards):
        assert st[d]["streamed_commits"] == 3 and st[d]["h2d_bytes"] == 3 * stripes(d) * 32, (d, st[d])
        assert st[d]["peer_bytes"] == (0 if d == 0 else 2 * stripes(d) * 32), (d, st[d])
    assert sum(x["h2d_bytes"] for x in st) == 3 * n * 32                       # the witness crosses PCIe exactly once per commit, split over the links
    mk.close()


def test_emu_bench_harness():
    """bench.py end to end on the emulator (tiny sizes): the harness logic -- CycleFold step order, host witness buffers,
    commit_upload with a device copy, the CPU-baseline leg -- produces the JSON line with the contract's fields."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emu", "--k", "3", "--log-key", "7", "--steps", "1", "--warmup", "0",
                        "--cpu-threads", "2", "--verify"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    # the secondary lines run too (tiny sizes); device_resident asserts inside bench.py that every resident commit == the streamed commit of the same witness
    assert line["secondary"]["device_resident"]["ms_per_step"] > 0 and line["secondary"]["true_leaf_rows"]["ms_per_step"] > 0
    assert line["verify"]["match"] and line["verify"]["first_step_digest"] == line["verify"]["oracle_first_step_digest"]     # --verify: first step == oracle chain
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["config"]["workload"].startswith("cyclefold_poseidon") and line["cpu_baseline"]["kind"] == "port"


def test_emu_bench_resident_chain():
    """bench.py --resident (the step with its inputs already in HBM: secondary.device_resident's form) folds the SAME chain as the headline
    form, which uploads the witness inside the step: equal state_digest after the same number of steps (resident runs one extra from-host
    step first, so that both witnesses have been committed by the streamed path -- every resident commitment is asserted equal inside)."""
    import json
    import sys
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--emu", "--k", "3", "--log-key", "7", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]
    res = _run_all([("host", common + ["--steps", "3"], dict(os.environ)), ("resident", common + ["--steps", "2", "--resident"], dict(os.environ))])
    lines = {}
    for tag, r in res.items():
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        lines[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert lines["host"]["state_digest"] == lines["resident"]["state_digest"]
    assert "inputs" in lines["resident"]["config"] and "inputs" not in lines["host"]["config"]


def test_emu_commit_vs_eip196():
    """the MSM kernel logic (emulator) on the EIP-196 known answers: ecAdd / ecMul as 1- and 2-term commitments"""
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from sirius_amd import _lib\n"
        f"_lib.load({EMU_LIB!r})\n"
        "import sirius_amd as S, oracle as O, eip196_cases as E\n"
        "def msm(s, b):\n"
        "    ck = S.CommitmentKey(0, b); out = ck.commit(np.ascontiguousarray(s)); ck.close(); return out\n"
        "E.check_adder(O, lambda a, b: S.point_sum(0, np.stack([a, b])), lambda k, p: S.point_mul(0, k, p), msm)\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_emu_non_contiguous_and_int64_inputs():
    """strided views / int64 arrays through RelaxedPlonkWitness.fold, lookup_coeff_2, batch_invert_assigned (tests/input_forms_cases.py)"""
    import sys
    code = (
        "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from sirius_amd import _lib\n"
        f"_lib.load({EMU_LIB!r})\n"
        "import sirius_amd as S, oracle as O\n"
        "from input_forms_cases import run_input_forms_case\n"
        "run_input_forms_case(S, O, 0, 700, 4); run_input_forms_case(S, O, 1, 300, 2)\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


POSEIDON_DEVICE_CODE = (
    "import random, numpy as np\n"
    "import sirius_amd as S\n"
    "from sirius_amd.field import MODULUS, ints_to_mont\n"
    "for field in (0, 1):\n"
    "    for (t, rf, rp) in ((5, 10, 10), (3, 4, 3), (7, 8, 5)):\n"
    "        for n in (0, 1, 3, 4, 5, 8, 29):\n"
    "            h = S.PoseidonHash(field, t, t - 1, rf, rp)\n"
    "            vals = [random.Random(n * 7 + i).randrange(MODULUS[field]) for i in range(n)]\n"
    "            if n: h.absorb_field(ints_to_mont(field, vals))\n"
    "            for bits, of in ((128, field), (253, 1 - field)):\n"
    "                assert np.array_equal(h.squeeze(bits, of), h.squeeze_device(bits, of)[0]), (field, t, n, bits)\n"
    "print('ok')\n")


def test_emu_device_sponge_equals_host_sponge():
    """k_poseidon_sponge (the device version of the random oracle, kept for the measured host-vs-device comparison) returns the host
    sponge's value: both fields, T = 3 / 5 / 7, buffers around the rate (padding in the same / an extra chunk)"""
    import sys
    code = ("import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\nfrom sirius_amd import _lib\n"
            f"_lib.load({EMU_LIB!r})\n" + POSEIDON_DEVICE_CODE)
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


SHARDED_UPLOAD_CODE = (
    "import numpy as np, torch\n"
    "import oracle as O\n"
    "from conftest import seeded_scalars\n"
    "dev = 'cuda' if torch.cuda.is_available() else 'cpu'\n"
    "for cid, n, world in ((0, 9000, 2), (1, 7 * 1024 + 5, 3), (0, 1024, 4), (1, 3000, 2)):\n"
    "    bases = O.make_bases(cid, 4, n + 3)\n"
    "    v = seeded_scalars(O, cid, n, 19, 'trace'); want = O.msm(cid, v, bases[:n]); parts = []\n"
    "    for r in range(world):\n"
    "        ck = S.CommitmentKey(cid, bases, rank=r, world=world)\n"
    "        d = torch.full((n, 4), 7, dtype=torch.int64, device=dev)\n"
    "        parts.append(ck.commit_upload(v, dev_copy=d))\n"
    "        got = d.cpu().numpy().view(np.uint64)\n"
    "        for s0 in range(0, n, 1024):                     # the rank's stripes are up, every other stripe is untouched\n"
    "            blk = got[s0:s0 + 1024]\n"
    "            assert np.array_equal(blk, v[s0:s0 + 1024]) if (s0 // 1024) % world == r else (blk == 7).all(), (cid, n, world, r, s0)\n"
    "        ck.close()\n"
    "    assert np.array_equal(S.point_sum(cid, np.stack(parts)), want), (cid, n, world)\n"
    "print('ok')\n")


def test_emu_sharded_commit_upload_chunked():
    """srs_commit_upload on a key sharded over processes: the rank's stripes go up in chunks overlapped with their MSM
    (commit_streamed with world > 1) -- partial commitments of all ranks sum to the oracle's, for 1 / 3 chunks, ragged ends,
    ranks without a stripe in the last chunk; foreign stripes of the device copy stay untouched"""
    import sys
    code = ("import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\nfrom sirius_amd import _lib\n"
            f"_lib.load({EMU_LIB!r})\nimport sirius_amd as S\n" + SHARDED_UPLOAD_CODE)
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    res = _run_all([(chunks, [sys.executable, "-c", code], tune_env(commit_chunks=chunks)) for chunks in ("1", "3")])
    for chunks, r in res.items():
        assert r.returncode == 0 and "ok" in r.stdout, (chunks, r.stdout[-500:], r.stderr[-1500:])


def test_emu_chain_digest_vs_oracle():
    """bench.py's headline chain (reference leaf rows + Poseidon-derived challenges, 2 CycleFold steps) through the emulator's
    kernel logic == the same chain recomputed on the oracle (tests/chain_cases.py); the GPU version is tests/test_chain_gpu.py."""
    import sys
    code = (
        "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from sirius_amd import _lib\n"
        f"_lib.load({EMU_LIB!r})\n"
        "import sirius_amd as S, oracle as O, chain_cases as CC\n"
        "a = CC.product_chain(S, 4, 8, 3, 2, emu=True); b = CC.oracle_chain(O, S, 4, 8, 3, 2)\n"
        "assert a == b, (a, b)\n"
        "assert CC.oracle_chain(O, S, 4, 8, 3, 2, fast=True, threads=2) == b      # the C legs the k = 20 GPU test uses\n"
        "c = CC.product_chain(S, 4, 8, 3, 2, emu=True, split_support=True)\n"
        "assert c == b, (c, b)\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_emu_sangria_chain_digest_vs_oracle():
    """bench.py's two-curve Sangria chain (secondary.sangria_k17's shapes, 2 steps, both step orders) through the emulator == the same
    chain on the oracle (tests/chain_cases.py::oracle_chain_sangria); the GPU version is tests/test_chain_gpu.py."""
    import sys
    code = (
        "import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from sirius_amd import _lib\n"
        f"_lib.load({EMU_LIB!r})\n"
        "import sirius_amd as S, oracle as O, chain_cases as CC\n"
        "b = CC.oracle_chain_sangria(O, S, 4, 8, 2)\n"
        "a = CC.product_chain_sangria(S, 4, 8, 2, from_host=False, emu=True)\n"
        "assert a == b, (a, b)\n"
        "c = CC.product_chain_sangria(S, 4, 8, 2, from_host=True, emu=True)\n"
        "assert c == b, (c, b)\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_emu_sangria_step_merged_commits():
    """bench.py's k=17-shaped Sangria step with resident traces: a trace's commitment in the same batched MSM as the cross terms of the
    prove that folds it (srs_sangria_prove_incoming, W2 resident) == commit, then srs_sangria_prove; Poseidon-derived challenges."""
    import sys
    code = (
        "import sys, argparse, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "from sirius_amd import _lib\n"
        f"_lib.load({EMU_LIB!r})\n"
        "import sirius_amd as S, bench\n"
        "from workloads import make_structure_inputs\n"
        "D = bench.Dist(argparse.Namespace(emu=True, gpus=1, dist_backend='nccl'))\n"
        "def chain(split):\n"
        "    bench.SPLIT_SUPPORT = split\n"
        "    pri = bench.SangriaSide(S, D, make_structure_inputs('primary', 4, seed=11), 8, 'primary')\n"
        "    sec = bench.SangriaSide(S, D, make_structure_inputs('secondary', 4, seed=12), 8, 'secondary')\n"
        "    pri.witness_commit(S, D, False); sec.witness_commit(S, D, False)\n"
        "    bench.sangria_step(S, D, pri, sec, False, True)\n"
        "    pri.settle(); sec.settle()\n"
        "    return [np.array(x).copy() for sd in (pri, sec) for x in (sd.accCW, sd.accCE, sd.inC, sd.r)]\n"
        "a, b = chain(False), chain(True)\n"
        "assert all(np.array_equal(x, y) for x, y in zip(a, b)) and any(x.any() for x in a)\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


SLOT_MODE_CODE = (
    "import os, sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
    "from sirius_amd import _lib\n"
    f"_lib.load({EMU_LIB!r})\n"
    "import sirius_amd as S, oracle as O\n"
    "from conftest import seeded_scalars\n"
    "n = int(os.environ['N'])\n"
    "for cid in [int(c) for c in os.environ['CURVES']]:\n"
    "    bases = O.make_bases(cid, 7 + cid, n); ck = S.CommitmentKey(cid, bases)\n"
    "    # regimes in turn: no hot buckets -> hot (unexpected: redo) -> hot (expected: overflow kernels) -> none (kernels launched for nothing) -> none\n"
    "    for rep, kind in enumerate(('uniform', 'trace', 'trace', 'uniform', 'uniform')):\n"
    "        sc = seeded_scalars(O, cid, n, 20 + rep, kind)\n"
    "        assert np.array_equal(ck.commit_upload(sc), O.msm(cid, sc, bases[:n])), (cid, rep, kind)\n"
    "    st = ck.msm_stats(); assert st['slot_sets'] >= 5 and st['hot_sets'] >= 2 and st['redo'] == int(os.environ['REDO']), st\n"
    "    sc = seeded_scalars(O, cid, n, 31, 'trace'); assert np.array_equal(ck.commit(sc), O.msm(cid, sc, bases[:n]))      # a whole MSM (msm_slots = 2: slot mode too)\n"
    "    vs = [seeded_scalars(O, cid, m, 40 + i, k) for i, (m, k) in enumerate(((n, 'trace'), (n // 2, 'uniform'), (7, 'trace')))]\n"
    "    for g, v in zip(ck.commit_batch(vs), vs): assert np.array_equal(g, O.msm(cid, v, bases[:len(v)]))\n"
    "    for x in (1, 5):      # every scalar the same small value: ONE bucket holds everything\n"
    "        v = O.ints_to_mont(O.SCALAR_FIELD[cid], [x] * n); assert np.array_equal(ck.commit_upload(v), O.msm(cid, v, bases[:n])), x\n"
    "    ck.close()\n"
    "print('ok')\n")


def test_emu_slot_mode_commits():
    """msm.hip slot mode (r04): persistent per-bucket partial sums across the chunks of a streamed commit, the part length chosen on the
    device, parts beyond the slots through the level kernels into the last slot, the per-key prediction with its redo -- forced onto
    small inputs: 4, 8 and 32 slots per bucket (so that hot buckets overflow), three chunks, both sort paths, whole MSMs and batches in slot mode
    (tuning msm_slots = 2), against the oracle; msm_stats must show the hot sets and exactly one redo per key that starts cold, none for one that
    starts with the default prediction."""
    import sys
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    jobs = []
    # a new key EXPECTS hot buckets (r05): no commit runs twice; msm_expect_ovf = 0 starts cold, so that the redo path stays covered
    for tag, slot_log, sort, n, curves, cold in (("s2", "2", "1", "3000", "0", "0"), ("s2g", "2", "2", "2100", "1", "0"), ("s3", "3", "2", "2500", "1", "0"),
                                                 ("s5", "5", "2", "2700", "0", "0"), ("s3warm", "3", "2", "2200", "0", "1")):
        env = tune_env(msm_slots=2, msm_slot_log=slot_log, msm_sort=sort, commit_chunks=3, msm_expect_ovf=cold, N=n, CURVES=curves,
                       REDO="1" if cold == "0" else "0")
        jobs.append((tag, [sys.executable, "-c", SLOT_MODE_CODE], env))
    for tag, r in _run_all(jobs).items():
        assert r.returncode == 0 and "ok" in r.stdout, (tag, r.stdout[-500:], r.stderr[-1500:])


def test_emu_long_level0_parts():
    """msm.hip l0_log_for: 64 gathered additions per level-0 thread (the setting of large MSMs), forced on a small one."""
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O, sirius_amd as S\n"
        "from sirius_amd import _lib\n"
        "from conftest import seeded_scalars\n"
        f"_lib.load({EMU_LIB!r})\n"
        "bases = O.make_bases(1, 4, 400); ck = S.CommitmentKey(1, bases)\n"
        "for kind in ('uniform', 'trace'):\n"
        "    v = seeded_scalars(O, 1, 400, 9, kind); assert np.array_equal(ck.commit(v), O.msm(1, v, bases))\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=tune_env(msm_l0=6), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_emu_accum1_lane_mode_deep_levels():
    """msm.hip k_accum1 with one lane per output (the mode of levels too large for quads) forced on a small MSM whose heavy buckets have
    hundreds of 2-entry level-0 parts (repeated small / negative scalars), ragged part counts included: several levels deep."""
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O, sirius_amd as S\n"
        "from oracle import pyref as P\n"
        "from sirius_amd import _lib\n"
        "from conftest import seeded_scalars\n"
        f"_lib.load({EMU_LIB!r})\n"
        "q = P.CURVES[1].q\n"
        "vals = [3] * 301 + [q - 5] * 187 + [7] * 33 + [1 << 16] * 18 + [(1 << 32) + 9] * 5 + [0] * 11\n"
        "v = np.concatenate([O.ints_to_mont(O.SCALAR_FIELD[1], vals), seeded_scalars(O, 1, 60, 9, 'trace'), seeded_scalars(O, 1, 25, 3, 'uniform')])\n"
        "bases = O.make_bases(1, 4, len(v)); ck = S.CommitmentKey(1, bases)\n"
        "assert np.array_equal(ck.commit(v), O.msm(1, v, bases))\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=tune_env(msm_l0=1, msm_quad_max=1),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_emu_wide_windows():
    """msm.hip wide pipeline (13 x 20-bit windows, 16 segments of 2^15 buckets) forced on a small MSM: digit-boundary values,
    zeros / bits / small values (every entry of a segment in a few buckets) and full-width scalars in one commit."""
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')\n"
        "import oracle as O, sirius_amd as S\n"
        "from oracle import pyref as P\n"
        "from sirius_amd import _lib\n"
        "from conftest import seeded_scalars\n"
        f"_lib.load({EMU_LIB!r})\n"
        "q = P.CURVES[1].q\n"
        "vals = [0, 1, 2, q - 1, q - 2, 1 << 19, (1 << 19) + 1, (1 << 19) - 1, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, 1 << 253,\n"
        "        (1 << 240) - 1, ((1 << 20) - 1) << 20, (q - 1) // 2, (q + 1) // 2, 0x80000 << 20, 0x80001 << 40, 0x7ffff << 60, 1, 1, 1]\n"
        "v = np.concatenate([O.ints_to_mont(O.SCALAR_FIELD[1], vals), seeded_scalars(O, 1, 150, 9, 'trace'), seeded_scalars(O, 1, 40, 3, 'uniform')])\n"
        "# the segments' counting sort in two passes through LDS (k_group_g / k_scatter2_g): dense buckets (a tile of a segment's pairs\n"
        "# inside one or two sub-segments), several segments, negative digits, one heavy bucket, mixed with uniform scalars (tiles\n"
        "# entry by entry) -- everything in ONE commit (a wide MSM costs the emulator ~8 s whatever its size), then an almost empty one\n"
        "import random; rnd = random.Random(6)\n"
        "dense = lambda lo, hi: sum(rnd.randrange(lo, hi) << (20 * w) for w in range(12))\n"
        "dv = [dense(1, 200) for _ in range(1500)] + [dense(0x3FF00, 0x40000) for _ in range(900)] + [dense(0xFFF00, 0xFFFFF) for _ in range(500)] + [1] * 1500\n"
        "v = np.concatenate([v, O.ints_to_mont(O.SCALAR_FIELD[1], dv), seeded_scalars(O, 1, 600, 5, 'uniform')])\n"
        "bases = O.make_bases(1, 4, len(v)); ck = S.CommitmentKey(1, bases)\n"
        "assert np.array_equal(ck.commit(v), O.msm(1, v, bases))\n"
        "ve = O.ints_to_mont(O.SCALAR_FIELD[1], [0] * 1000 + [5])\n"
        "assert np.array_equal(ck.commit(ve), O.msm(1, ve, bases[:1001]))\n"
        "print('ok')\n")
    subprocess.check_call(["make", "-C", EMU_DIR, "-j4"], stdout=subprocess.DEVNULL)
    res = _run_all([("wide", [sys.executable, "-c", code], tune_env(msm_wide=1, msm_wide_min=0))], timeout=1800)
    for sort, r in res.items():
        assert r.returncode == 0 and "ok" in r.stdout, (sort, r.stdout[-500:], r.stderr[-1500:])


def test_emu_concatenate_with_padding(emu, oracle):
    """SURVEY A17 on the emulator: the reference's concatenate_with_padding unit tests + the column-wise witness commit."""
    import torch
    from test_commit_gpu import _concat_cases
    _concat_cases(emu, oracle, 1, lambda n: torch.full((n, 4), 7, dtype=torch.int64), k=8)
    _concat_cases(emu, oracle, 0, lambda n: torch.full((n, 4), 7, dtype=torch.int64), k=10)     # 5 * 2^10 scalars: the sharded forms stream (>= 2 stripes per shard)


def test_emu_sharded_fold_with_rotations(emu, oracle):
    """a row-sharded chain over a gate WITH rotated queries.  Two 'ranks' (two structure handles, set_shard(r, 2)) fold two
    steps; after every fold everything a rank does not own and does not read (non-halo rows) is overwritten with garbage -- the rank's
    rows of the cross terms must still equal the unsharded chain's.  srs_structure_fold_sharded folds the halo rows that
    srs_fold_lincomb_sharded left stale; srs_structure_upload_shard_halo brings the incoming trace's."""
    O = oracle
    from sirius_amd import expression as X
    from sirius_amd import protogalaxy as PG
    from workloads import rand_fe
    field, k, world = 1, 11, 2
    rows, SL = 1 << k, 1 << 10
    rng = np.random.default_rng(41)
    nfix, nadv = 2, 3
    # fixed: 0, 1; advice: 2, 3, 4     f0 * (a(+1) * b(-2) - c) + f1 * a(-1)
    gate = X.Sum(X.Product(X.Polynomial(0), X.Sum(X.Product(X.Polynomial(2, 1), X.Polynomial(3, -2)), X.Negated(X.Polynomial(4)))),
                 X.Product(X.Polynomial(1), X.Polynomial(2, -1)))
    fixed = [rand_fe(rng, rows) for _ in range(nfix)]
    acc0 = rand_fe(rng, nadv * rows)
    incoming = [rand_fe(rng, nadv * rows) for _ in range(2)]
    rs = [rand_fe(rng, 1)[0] for _ in range(2)]
    one = O.ints_to_mont(field, [1])[0]
    none = rand_fe(rng, 0)
    u = rand_fe(rng, 1)[0]

    def cross(St, W1, W2):
        terms, _ = emu.VanillaFS.commit_cross_terms(None, St, none, u, W1, none, W2)
        return terms

    St = emu.PlonkStructure(field, k, [], fixed, nadv, [gate])
    ref, acc = [], acc0.copy()
    for t in range(2):
        ref.append(cross(St, acc, incoming[t]))
        acc = PG.fold_witness(field, [acc, incoming[t]], np.stack([one, rs[t]]))
    St.close()
    row_of = np.tile(np.arange(rows), nadv)
    for rank in range(world):
        Sr = emu.PlonkStructure(field, k, [], fixed, nadv, [gate])
        Sr.set_shard(rank, world)
        own = ((row_of >> 10) % world) == rank
        halo = np.zeros(rows, bool)                         # rotations -2 .. +1 around the rank's stripes
        for s in range(rank, rows // SL, world):
            for r in (-2, -1):
                halo[(s * SL + r) % rows] = True
            halo[((s + 1) * SL) % rows] = True
        keep = own | halo[row_of]
        assert (keep & ~own).any()
        accr = acc0.copy()
        for t in range(2):
            inr = rand_fe(rng, nadv * rows)                  # garbage ...
            inr[own] = incoming[t][own]                      # ... except the rank's stripes (a sharded commit_upload) ...
            Sr.upload_shard_halo(incoming[t], inr)           # ... and the halo rows
            got = cross(Sr, accr, inr)
            for a, b in zip(got, ref[t]):
                assert np.array_equal(a[own[:rows]], b[own[:rows]]), (rank, t)
            PG.fold_witness(field, [accr, inr], np.stack([one, rs[t]]), out=accr, structure=Sr, reference_compat=False)
            accr[~keep] = rand_fe(rng, int((~keep).sum()))   # what the rank neither owns nor reads may be anything
        Sr.close()

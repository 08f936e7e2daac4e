"""Randomised soak of the STREAMED commit in slot mode on the GPU (not part of the test suite): random lengths around the chunking
thresholds, random value mixtures (no hot buckets / bits and small values / every scalar equal / one scalar hot), pinned and pageable
sources, two keys on two host threads, plus (r05) a 3-shard multi-device key on a third thread and the column form on a fourth -- every commitment against the oracle; prints the
slot-mode counters at the end.
usage: python tools/soak_stream.py [seed] [seconds]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
import sirius_amd as S
from conftest import seeded_scalars

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
NMAX = 3 << 20
errs, stats = [], {}


def worker(cid, wseed):
    rng = np.random.default_rng(wseed)
    ck = S.CommitmentKey.setup_synthetic(cid, NMAX, seed=100 + cid)
    bases = ck.bases()
    sf = O.SCALAR_FIELD[cid]
    t0, done = time.time(), 0
    try:
        while time.time() - t0 < budget:
            n = int(rng.choice([(1 << 19) - 1, (1 << 19) + 1, 1 << 20, (1 << 20) + 7, (1 << 21) + 12345, NMAX, int(rng.integers(1 << 19, NMAX))]))
            kind = int(rng.integers(0, 5))
            if kind == 0:
                v = seeded_scalars(O, cid, n, int(rng.integers(0, 1 << 30)), "uniform")
            elif kind == 1:
                v = seeded_scalars(O, cid, n, int(rng.integers(0, 1 << 30)), "trace")
            elif kind == 2:
                v = np.repeat(seeded_scalars(O, cid, 1, int(rng.integers(0, 1 << 30)), "uniform"), n, axis=0)          # one bucket per window holds everything
            elif kind == 3:
                v = seeded_scalars(O, cid, n, int(rng.integers(0, 1 << 30)), "uniform")
                v[:: int(rng.integers(2, 9))] = O.ints_to_mont(sf, [int(rng.integers(1, 4))])[0]                          # a hot bucket among uniform ones
            else:
                v = np.zeros((n, 4), np.uint64)
                v[int(rng.integers(0, n))] = O.ints_to_mont(sf, [5])[0]                                                   # (almost) empty
            src = v
            hb = None
            if rng.random() < 0.5:
                hb = S.HostBuffer(n)
                hb.array[:] = v
                src = hb.array
            got = ck.commit_upload(src)
            assert np.array_equal(got, O.msm(cid, v, bases[:n])), ("commit_upload", cid, n, kind)
            if hb is not None:
                hb.close()
            done += 1
        stats[cid] = (done, ck.msm_stats())
    except Exception as e:      # surfaced in the main thread
        errs.append(repr(e))
    ck.close()


def worker_multi(cid, wseed, shards):
    """r05: the streamed commit on a MULTI-DEVICE key (logical shards on the visible devices): every shard streams its own stripes, the device copy
    is assembled by peer copies -- commitment against the oracle, device copy against the source, bytes per link against the stripe count"""
    import torch
    rng = np.random.default_rng(wseed)
    n_key = 1 << 20
    ck = S.CommitmentKey.setup_synthetic_multi(cid, n_key, seed=200 + cid, n_devices=shards)
    bases = ck.bases()
    t0, done, h2d = time.time(), 0, 0
    try:
        while time.time() - t0 < budget:
            n = int(rng.choice([n_key, n_key - 1, (1 << 19) + 3, int(rng.integers(shards << 11, n_key))]))
            v = seeded_scalars(O, cid, n, int(rng.integers(0, 1 << 30)), ("uniform", "trace")[int(rng.integers(0, 2))])
            d = torch.zeros((n, 4), dtype=torch.int64, device="cuda") if rng.random() < 0.7 else None
            got = ck.commit_upload(v, dev_copy=d)
            assert np.array_equal(got, O.msm(cid, v, bases[:n])), ("multi commit_upload", cid, n)
            if d is not None:
                torch.cuda.synchronize()
                assert np.array_equal(d.cpu().numpy().view(np.uint64), v), ("multi device copy", cid, n)
            h2d += n * 32
            done += 1
        st = [ck.shard_stats(j) for j in range(shards)]
        assert sum(x["h2d_bytes"] for x in st) == h2d, (st, h2d)          # every byte crossed exactly one link, once
        stats[f"multi{shards}_{cid}"] = (done, ck.msm_stats(), [x["h2d_bytes"] for x in st])
    except Exception as e:
        errs.append(repr(e))
    ck.close()


def worker_columns(cid, wseed, shards):
    """r05: the COLUMN form (concatenate_with_padding assembled in HBM) streamed per shard on a multi-device key: random column lengths and pads"""
    import torch
    rng = np.random.default_rng(wseed)
    n_key = 1 << 20
    ck = S.CommitmentKey.setup_synthetic_multi(cid, n_key, seed=300 + cid, n_devices=shards)
    bases = ck.bases()
    t0, done = time.time(), 0
    try:
        while time.time() - t0 < budget:
            pad = int(rng.choice([0, 1, 1 << 15, (1 << 16) + 7, 1 << 17]))
            cols, tot = [], 0
            while True:
                ln = int(rng.choice([0, 1, 1000, 1 << 15, (1 << 16) - 3, (1 << 17) + 11, int(rng.integers(1, 1 << 17))]))
                if tot + max(ln, pad) > n_key or (tot >= n_key // 2 and rng.random() < 0.3):
                    break
                cols.append(seeded_scalars(O, cid, ln, int(rng.integers(0, 1 << 30)), ("uniform", "trace")[int(rng.integers(0, 2))]) if ln
                            else np.zeros((0, 4), np.uint64))
                tot += max(ln, pad)
            if tot == 0:
                continue
            W = np.zeros((tot, 4), np.uint64)
            at = 0
            for c in cols:
                W[at:at + c.shape[0]] = c
                at += max(c.shape[0], pad)
            d = torch.full((tot, 4), 7, dtype=torch.int64, device="cuda")
            got = ck.commit_upload_columns(cols, pad, dev_copy=d)
            assert np.array_equal(got, O.msm(cid, W, bases[:tot])), ("columns commit", cid, tot, pad, [c.shape[0] for c in cols])
            torch.cuda.synchronize()
            assert np.array_equal(d.cpu().numpy().view(np.uint64), W), ("columns device copy", cid, tot, pad)
            done += 1
        stats[f"columns{shards}_{cid}"] = (done, ck.msm_stats())
    except Exception as e:
        errs.append(repr(e))
    ck.close()


th = [threading.Thread(target=worker, args=(c, seed * 7 + c)) for c in (0, 1)] + [threading.Thread(target=worker_multi, args=(0, seed * 7 + 5, 3)),
                                                                                 threading.Thread(target=worker_columns, args=(1, seed * 7 + 6, 3))]
[t.start() for t in th]
[t.join() for t in th]
assert not errs, errs
print("soak_stream OK", stats, "seed", seed)

"""Randomised emulator check of srs_commit_upload_columns on sharded keys (not part of the test suite): random column lengths / pads, multi-device keys of
2-3 shards and the process-sharded form -- commitment vs the oracle, device copy vs the concatenation.   usage: python tools/soak_columns_emu.py [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import sirius_amd as S
from sirius_amd import _lib
_lib.load(os.path.join(ROOT, "tests", "emu", "libsirius_emu.so"))
import oracle as O
from conftest import seeded_scalars
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cid = 0
NK = 9000
bases = O.make_bases(cid, 11, NK)
for it in range(8):
    shards = int(rng.integers(2, 4))
    pad = int(rng.choice([0, 1, 700, 1024, 1500, 2048, 3000]))
    cols, tot = [], 0
    while True:
        ln = int(rng.choice([0, 1, 5, 900, 1024, 1025, 2047, 2500]))
        step = max(ln, pad)
        if tot + step > NK: break
        cols.append(seeded_scalars(O, cid, ln, int(rng.integers(1 << 30)), "uniform") if ln else np.zeros((0, 4), np.uint64)); tot += step
        if tot >= shards << 11 and rng.random() < 0.3: break
    W = np.zeros((tot, 4), np.uint64); at = 0
    for c in cols:
        W[at:at + c.shape[0]] = c; at += max(c.shape[0], pad)
    if tot == 0: continue
    want = O.msm(cid, W, bases[:tot])
    mk = S.CommitmentKey.create_multi(cid, bases, shards)
    d = torch.full((tot, 4), 7, dtype=torch.int64)
    got = mk.commit_upload_columns(cols, pad, dev_copy=d)
    ok = np.array_equal(got, want) and np.array_equal(d.numpy().view(np.uint64), W)
    mk.close()
    parts = []
    ok2 = True
    for r in range(shards):
        rk = S.CommitmentKey(cid, bases, rank=r, world=shards)
        d = torch.full((tot, 4), 7, dtype=torch.int64)
        parts.append(rk.commit_upload_columns(cols, pad, dev_copy=d))
        g = d.numpy().view(np.uint64)
        for s_ in range(r, (tot + 1023) // 1024, shards):
            ok2 = ok2 and np.array_equal(g[s_ * 1024:(s_ + 1) * 1024], W[s_ * 1024:(s_ + 1) * 1024])
        rk.close()
    ok2 = ok2 and np.array_equal(S.point_sum(cid, np.stack(parts)), want)
    print(it, "shards", shards, "pad", pad, "lens", [c.shape[0] for c in cols], "n", tot, "streamed", tot >= shards << 11, "OK" if ok and ok2 else "FAIL", ok, ok2, flush=True)

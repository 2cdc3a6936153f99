"""One-off wide sweep of the explicit-segment_ids bookkeeping (glx_seg_scan_kernel / glx_seg_fixup_kernel) against the
oracle's cursor (aggregating_request.cc:86-105): random lengths, segment counts, run structures (uniform, ragged, long
empty runs, everything in one segment), with and without a violation at a random place (negative, too large, or
decreasing), host and device pointers.  python scripts/r05/seg_sweep.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import glx  # noqa: E402
from oracle_bindings import Oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
orc = Oracle()
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(os.environ.get("GLX_SWEEP_SEED", "20260924")))
tables = {}
for D in (4, 64, 256):
    X = rng.standard_normal((2000, D)).astype(np.float32)
    tables[D] = (X, glx.Features(X))
kinds = {}
bad = 0
for case in range(cases):
    n = int(rng.choice([0, 1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4097, 20000, 50001]))
    sg = int(rng.choice([1, 2, 3, 7, 64, 65, 66, 130, 1000, 5000, 20000]))
    kind = int(rng.integers(0, 6))
    if kind == 0 and n % sg == 0 and n > 0:
        seg = (np.arange(n) // (n // sg)).astype(np.int32)
        name = "uniform"
    elif kind == 1:
        seg = np.sort(rng.integers(0, sg, n)).astype(np.int32)
        name = "ragged"
    elif kind == 2:
        pick = np.sort(rng.choice(sg, min(sg, int(rng.integers(1, 5))), replace=False))
        seg = np.sort(pick[rng.integers(0, pick.shape[0], n)]).astype(np.int32)
        name = "few segments, long empty runs"
    elif kind == 3:
        seg = np.full(n, int(rng.integers(0, sg)), np.int32)
        name = "one segment"
    elif kind == 4:
        seg = np.sort(rng.integers(max(0, sg - 3), sg, n)).astype(np.int32)
        name = "all at the end"
    else:
        seg = np.repeat(np.arange(sg, dtype=np.int32), rng.integers(0, 4, sg))[:n]
        seg = np.concatenate([seg, np.full(n - seg.shape[0], sg - 1, np.int32)]) if seg.shape[0] < n else seg
        name = "short segments"
    if n > 1 and rng.random() < 0.4:
        at = int(rng.integers(0, n))
        seg = seg.copy()
        seg[at] = [-1, -1000000, sg, sg + 5, max(int(seg[at - 1]) - 1 - int(rng.integers(0, 3)), -3) if at else -2][int(rng.integers(0, 5))]
        name += " + violation"
    D = int(rng.choice([4, 64, 256]))
    X, f = tables[D]
    ids = rng.integers(-1, 2001, n).astype(np.int64)
    op = ["SumAggregator", "MaxAggregator", "MeanAggregator"][case % 3]
    we, wc = orc.aggregate(X, op, ids, seg, sg, default_attr=0.5)
    if case % 2:
        e, c = f.aggregate(op, ids, seg, sg, default_attr=0.5)
    else:
        te, tc = f.aggregate(op, torch.from_numpy(ids).to(dev), torch.from_numpy(seg).to(dev), sg, default_attr=0.5)
        e, c = te.cpu().numpy(), tc.cpu().numpy()
    ok = np.array_equal(c, wc) and np.array_equal(e.view(np.uint32), we.view(np.uint32))
    kinds[name] = kinds.get(name, 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH case %d: n=%d sg=%d D=%d %s %s" % (case, n, sg, D, op, name))
print("%d cases, %d mismatches" % (cases, bad))
for k in sorted(kinds):
    print("  %-45s %d" % (k, kinds[k]))
sys.exit(1 if bad else 0)

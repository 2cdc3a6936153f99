import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch, glx, synth
dev = torch.device("cuda", 0)
for V, D in ((1_000_000, 256), (5_000_000, 256), (10_000_000, 256), (10_000_000, 64)):
    X = synth.features_torch(V, D, 5, dev)
    dense = glx.Features(X)
    ids = torch.arange(0, V, dtype=torch.int64, device=dev)
    hashed = glx.Features(X, ids=ids)
    q = torch.randint(0, V, (2_000_000,), device=dev)
    a = dense.lookup(q); b = hashed.lookup(q)
    bad = (a != b).any(dim=1)
    print(V, D, "lookup mismatching rows:", int(bad.sum()), "of", q.shape[0])
    if bad.any():
        r = q[bad][:5]; print("  ids", r.tolist(), "max bad id", int(q[bad].max()), "min bad id", int(q[bad].min()))
    n = 4_000_000
    nid = torch.randint(0, V, (n,), device=dev); seg = (torch.arange(n, device=dev) // 10).to(torch.int32)
    e1, c1 = dense.aggregate("MaxAggregator", nid, seg, n // 10); e2, c2 = hashed.aggregate("MaxAggregator", nid, seg, n // 10)
    print("   aggregate equal:", bool(torch.equal(e1, e2)), bool(torch.equal(c1, c2)))
    del dense, hashed, X
    torch.cuda.empty_cache()

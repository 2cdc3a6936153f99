"""Cache-free rate of the segmented reduce against the ROW size: ids uniform over a 10 GB table of V = 10 GB / (4 D)
rows, segments of 10, the same number of gathered bytes per launch (16.8 GB) at every D."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx
dev = torch.device("cuda", 0)
k = 10
gen = torch.Generator(device=dev); gen.manual_seed(5)
for D in (32, 64, 128, 256, 512, 1024, 2048):
    V = int(10.24e9 // (4 * D))
    Sg = int(65536 * 25 * 256 // D)
    X = torch.empty((V, D), dtype=torch.float32, device=dev)
    step = max(1, V // 8)
    for a in range(0, V, step):
        X[a:a + step].uniform_(-1, 1)
    f = glx.Features(X); del X
    emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
    ids = torch.randint(0, V, (Sg * k,), generator=gen, device=dev)
    alg = Sg * k * (D * 4 + 12) + Sg * (D * 4 + 4)
    for agg in ("MaxAggregator", "MeanAggregator"):
        r = []
        for _ in range(6):
            torch.cuda.synchronize(); glx.profile_enable(True)
            f.aggregate(agg, ids, None, Sg, out=(emb, cnt))
            torch.cuda.synchronize(); glx.profile_enable(False)
            r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE).sum()))
        t = float(np.median(r[1:]))
        print("D %5d  row %5d B  %-15s %.3f ms  %.0f GB/s algorithmic" % (D, D * 4, agg, t, alg / t / 1e6), flush=True)
    f.close(); del emb, cnt, ids
    torch.cuda.empty_cache()

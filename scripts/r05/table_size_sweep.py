"""Does the cache-free rate of the hop-2 reduce depend on how much memory the random rows are spread over?
Same request shape as the headline's hop-2 launch (1.64 M segments of 10, D = 256, MaxAggregator), ids uniform over the
first R rows of ONE 40 M-row table (41 GB), R from 0.5 M (0.5 GB: twice the Infinity Cache) to 40 M.  If address
translation (TLB reach) were what separates the row gather (5.2 TB/s) from a streaming read (6.2 TB/s), the rate would
fall as R grows; DRAM page / bank behaviour alone does not care about R once R is far beyond the caches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx
dev = torch.device("cuda", 0)
V, D, Sg, k = 40_000_000, 256, 65536 * 25, 10
X = torch.empty((V, D), dtype=torch.float32, device=dev)
for a in range(0, V, 4_000_000):
    X[a:a + 4_000_000].uniform_(-1, 1)
f = glx.Features(X); del X
gen = torch.Generator(device=dev); gen.manual_seed(5)
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
alg = Sg * k * (D * 4 + 12) + Sg * (D * 4 + 4)
for R in (500_000, 1_000_000, 2_500_000, 5_000_000, 10_000_000, 20_000_000, 40_000_000):
    ids = torch.randint(0, R, (Sg * k,), generator=gen, device=dev)
    r = []
    for _ in range(6):
        torch.cuda.synchronize(); glx.profile_enable(True)
        f.aggregate("MaxAggregator", ids, None, Sg, out=(emb, cnt))
        torch.cuda.synchronize(); glx.profile_enable(False)
        r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE).sum()))
    t = float(np.median(r[1:]))
    print("rows %9d  span %6.1f GB  %.3f ms  %.0f GB/s algorithmic" % (R, R * D * 4 / 1e9, t, alg / t / 1e6), flush=True)

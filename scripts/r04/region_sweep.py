"""What bounds the aggregation kernel on rows that miss the L2s: the C3 hop-2 request shape (16.4 M ids -> 1.64 M segments of
10, D = 256, MaxAggregator) with ids uniform over R rows of the 10 M-row table, R from L2-resident to the whole table, laid
out two ways:
  dense  : rows 0 .. R-1 (R KiB of consecutive addresses: few pages)
  spread : rows j * (V // R) (the same number of lines, hence the same cache footprint, but strewn over all 10 GB: every
           row in a page region of its own once R <= V / 2048)
If address translation (TLB reach) mattered, `spread` would be slower than `dense` at equal R; if only the caches and the
DRAM matter, the two agree.  Column slices off (x1) and the default (x2) for each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, D, B0, k1, k2 = 10_000_000, 256, 65536, 25, 10
f = glx.Features(synth.features_torch(V, D, 5, dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
Sg = B0 * k1
N = Sg * k2
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
alg = N * (4 * D + 12) + Sg * (4 * D + 4)

def t(ids, reps=5):
    r = []
    for _ in range(reps):
        torch.cuda.synchronize(); glx.profile_enable(True)
        f.aggregate("MaxAggregator", ids, None, Sg, out=(emb, cnt))
        torch.cuda.synchronize(); glx.profile_enable(False)
        r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE).sum()))
    return float(np.median(r))

print("# ids uniform over R rows; ms per launch (median of 5) and algorithmic TB/s (%.2f GB per launch)" % (alg / 1e9))
print("%10s %9s | %8s %8s | %8s %8s | %s" % ("R rows", "MB", "dense x1", "dense x2", "sprd x1", "sprd x2", "TB/s (dense x2, spread x2)"))
for R in (2048, 16384, 100_000, 250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000):
    base = torch.randint(0, R, (N,), generator=gen, device=dev)
    row = []
    for layout in ("dense", "spread"):
        ids = base if layout == "dense" else base * (V // R)
        for x in (1, 2):
            glx.tune("agg_xcd_slices", x)
            row.append(t(ids))
    glx.tune("agg_xcd_slices", 0)
    print("%10d %9.0f | %8.3f %8.3f | %8.3f %8.3f | %.2f %.2f" % (R, R * D * 4 / 1e6, row[0], row[1], row[2], row[3],
                                                          alg / row[1] / 1e9, alg / row[3] / 1e9), flush=True)

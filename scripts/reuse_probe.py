"""How much row reuse is there in a hop-2 aggregation request, and what do L2 /
Infinity Cache deliver when the working set fits?  (design probe, not a test)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)  # seeds = vertices that have out-edges (as in bench.py)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
f = glx.Features(synth.features_torch(V, D, 5, dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
Sg = B0 * k1
seg = (torch.arange(Sg * k2, device=dev) // k2).to(torch.int32)
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=1, call_counter=1)
ids = n2.view(-1)
c = torch.bincount(ids, minlength=V)
cs = torch.sort(c, descending=True).values.double()
tot = cs.sum().item()
cum = torch.cumsum(cs, 0)
for n in (1024, 4096, 32768, 131072, 262144, 1 << 20):
    print("top %7d rows (%6.1f MB) cover %.1f%% of the %d accesses" % (n, n * D * 4 / 1e6, 100 * cum[n - 1].item() / tot, int(tot)))
def t(ids, reps=5):
    out = []
    for r in range(reps):
        torch.cuda.synchronize(); glx.profile_enable(True)
        f.aggregate("MaxAggregator", ids, seg, Sg, out=(emb, cnt))
        torch.cuda.synchronize(); glx.profile_enable(False)
        out.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE)[0]))
    return out
print("real ids           :", ["%.3f" % x for x in t(ids)])
for rows in (2048, 16384, 100_000, 200_000, 1_000_000, V):
    fake = torch.randint(0, rows, (Sg * k2,), generator=gen, device=dev)
    ms = t(fake)
    print("uniform over %8d rows (%7.1f MB): %s ms -> %.1f TB/s algorithmic" % (rows, rows * 1024 / 1e6, ["%.3f" % x for x in ms], 18.658 / np.mean(ms[1:])))

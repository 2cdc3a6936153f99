"""Would the hop-2 reduce run faster if segments that stem from the SAME hop-1 vertex were processed next to each other?
(Their ten draws come from the same adjacency row, so their feature rows overlap; today segments run in request
order, where such twins are far apart.)  The C3 hop-2 request as sampled, then with its segments permuted by hop-1
vertex id (stable), by hop-1 vertex frequency (hubs first), and at random -- same kernel, same bytes, different order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
pool = torch.unique(src)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
f = glx.Features(synth.features_torch(V, D, 5, dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
Sg = B0 * k1
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=1, call_counter=1)
h1 = n1.view(-1)
n2 = n2.view(Sg, k2)
def t(ids, reps=7):
    r = []
    for _ in range(reps):
        torch.cuda.synchronize(); glx.profile_enable(True)
        f.aggregate("MaxAggregator", ids, None, Sg, out=(emb, cnt))
        torch.cuda.synchronize(); glx.profile_enable(False)
        r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE).sum()))
    return float(np.median(r))
uniq, inv, counts = torch.unique(h1, return_inverse=True, return_counts=True)
print("# hop-1 frontier: %d samples, %d distinct vertices; hop-2 ids: %d, distinct %d" % (h1.numel(), uniq.numel(), n2.numel(), torch.unique(n2).numel()))
orders = {
    "as sampled": torch.arange(Sg, device=dev),
    "by hop-1 vertex id": torch.argsort(h1, stable=True),
    "by hop-1 vertex frequency (hubs first)": torch.argsort(-counts[inv] * (1 << 24) + inv, stable=True),
    "random": torch.randperm(Sg, device=dev, generator=gen),
}
for x in (1, 2):
    glx.tune("agg_xcd_slices", x)
    for name, o in orders.items():
        ids = n2[o].reshape(-1).contiguous()
        print("x%d  %-42s %.3f ms" % (x, name, t(ids)), flush=True)
glx.tune("agg_xcd_slices", 0)

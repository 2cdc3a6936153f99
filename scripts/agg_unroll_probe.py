"""A/B of rows in flight per lane in glx_aggregate_kernel (GLX_AGG_UNROLL), same process, same C3 hop-2 request."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
f = glx.Features(synth.features_torch(V, D, 5, dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
Sg = B0 * k1
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
emb1 = torch.empty((B0, D), dtype=torch.float32, device=dev); cnt1 = torch.empty(B0, dtype=torch.int32, device=dev)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=1, call_counter=1)
ids2, ids1 = n2.view(-1).contiguous(), n1.view(-1).contiguous()
def t(ids, sg, out, reps=8):
    r = []
    for _ in range(reps):
        torch.cuda.synchronize(); glx.profile_enable(True)
        f.aggregate("MaxAggregator", ids, None, sg, out=out)
        torch.cuda.synchronize(); glx.profile_enable(False)
        r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE)[0]))
    return np.median(r)
for rnd in range(2):
    for u in (8, 3, 4, 5, 6, 8):
        os.environ["GLX_AGG_UNROLL"] = str(u)
        print("round %d unroll %2d: hop-2 (k=10) %.3f ms   hop-1 (k=25) %.3f ms" % (rnd, u, t(ids2, Sg, (emb, cnt)), t(ids1, B0, (emb1, cnt1))), flush=True)

"""A/B of column slices in glx_aggregate_kernel (GLX_AGG_SLICES), same process, same C3 hop-2 request."""
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
ref = torch.empty_like(emb)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=1, call_counter=1)
ids2 = n2.view(-1).contiguous()
fake = torch.randint(0, V, (ids2.shape[0],), generator=gen, device=dev)
def t(ids, out, reps=6):
    r = []
    for _ in range(reps):
        torch.cuda.synchronize(); glx.profile_enable(True)
        f.aggregate("MaxAggregator", ids, None, Sg, out=out)
        torch.cuda.synchronize(); glx.profile_enable(False)
        r.append(float(np.sum(glx.profile_collect(glx.KERNEL_AGGREGATE))))
    return np.median(r)
os.environ["GLX_AGG_SLICES"] = "1"
f.aggregate("MaxAggregator", ids2, None, Sg, out=(ref, cnt))
for rnd in range(2):
    for sl in (1, 2, 4, 8):
        os.environ["GLX_AGG_SLICES"] = str(sl)
        a = t(ids2, (emb, cnt))
        same = bool(torch.equal(emb.view(torch.int32), ref.view(torch.int32)))
        b = t(fake, (emb, cnt))
        print("round %d slices %d: hop-2 request %.3f ms (bit-identical %s)   uniform random rows %.3f ms" % (rnd, sl, a, same, b), flush=True)

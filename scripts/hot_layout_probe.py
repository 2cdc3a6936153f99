"""Design probe: does storing the feature rows of high in-degree vertices next to each other (hot-first
layout) make the hop-2 aggregation kernel faster, as the 3-source kernel of the distributed store suggests
(1.93 ms with 87 % of the rows coming from a dense 1 GB replica vs 2.27 ms over the 10 GB table)?
Times glx_aggregate on the same C3 hop-2 request (a) as is, (b) over a table whose rows are permuted by
descending in-degree with pre-translated ids, and the cost of translating ids on the fly."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)
indeg = torch.bincount(dst, minlength=V)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
X = synth.features_torch(V, D, 5, dev)
f = glx.Features(X)
gen = torch.Generator(device=dev); gen.manual_seed(3)
Sg = B0 * k1
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=1, call_counter=1)
ids = n2.view(-1).contiguous()
def t(feats, ids, reps=6):
    out = []
    for r in range(reps):
        torch.cuda.synchronize(); glx.profile_enable(True)
        feats.aggregate("MaxAggregator", ids, None, Sg, out=(emb, cnt))
        torch.cuda.synchronize(); glx.profile_enable(False)
        out.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE)[0]))
    return out
base = t(f, ids)
ref = emb.clone()
print("as is (swizzled id order)       :", ["%.3f" % x for x in base])
order = torch.sort(indeg, descending=True, stable=True).indices  # slot -> id
slot_of = torch.empty(V, dtype=torch.int64, device=dev); slot_of[order] = torch.arange(V, device=dev)
del f
X2 = X[order].contiguous(); del X
f2 = glx.Features(X2); del X2
ids2 = slot_of[ids]
hot = t(f2, ids2)
assert torch.equal(emb, ref)
print("hot-first layout, translated ids:", ["%.3f" % x for x in hot])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
slot32 = slot_of.to(torch.int32)
for _ in range(2): y = slot32[ids]
e0.record()
for _ in range(5): y = slot32[ids]
e1.record(); torch.cuda.synchronize()
print("translate 16.4M ids through an int32 map (torch gather): %.3f ms" % (e0.elapsed_time(e1) / 5))
for u in (10, 12):
    os.environ["GLX_AGG_UNROLL"] = str(u)
    print("hot-first, unroll %d           :" % u, ["%.3f" % x for x in t(f2, ids2)])
os.environ["GLX_AGG_UNROLL"] = "8"
print("hot-first, unroll 8 again       :", ["%.3f" % x for x in t(f2, ids2)])

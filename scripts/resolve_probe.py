"""The collective half of a hop-2 aggregation (resolve + dedup + exchange; world size 1, generic path) in isolation:
rank-select membership of the feature replica vs the packed hash map (GLX_DIST_NO_BITMAP), same process."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
os.environ["GLX_DIST_NO_SHORTCUT"] = "1"
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)
indeg = torch.bincount(dst, minlength=V)
hot = torch.topk(indeg, V // 4).indices.to(torch.int64)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
f = glx.Features(synth.features_torch(V, D, 5, dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=1, call_counter=1)
ids2 = n2.view(-1).contiguous()
comm = glx.Comm.local(991, 0, 0, 1)
for mode in ("bitmap", "hash", "bitmap", "hash"):
    if mode.startswith("hash"):
        glx.tune("dist_no_bitmap", 1)
    else:
        glx.tune("dist_no_bitmap", -1)
    st = glx.DistStore(comm, features=f)
    st.set_cache(hot)
    for _ in range(3):
        st.aggregate_begin(0, ids2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        st.aggregate_begin(0, ids2)
    torch.cuda.synchronize()
    print("%s: aggregate_begin of %d ids: %.3f ms per call; %s" % (mode, ids2.shape[0], (time.perf_counter() - t0) / 20 * 1e3, st.stats()), flush=True)
    st.close()

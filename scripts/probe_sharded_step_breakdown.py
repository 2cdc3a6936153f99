"""Where does the edge-cut sampling stage spend its time at world-size 1 (RCCL self-exchange)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch, torch.distributed as dist, glx, synth
import dist as gdist
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
V, E, B0, k1, k2 = 10_000_000, 100_000_000, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
pool = torch.unique(src)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
store = gdist.ShardedStore(gdist.DeviceOps(), g)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), device=dev)]
a, _ = store.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
ids = a.view(-1)


def T(label, fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    print("%-44s %7.3f ms" % (label, (time.perf_counter() - t0) / n * 1e3))
    return out


T("direct sample hop-2 (no sharding)", lambda: g.sample("EdgeWeightSampler", ids, k2, seed=1, call_counter=1))
T("store.sample hop-2 (edge-cut path)", lambda: store.sample("EdgeWeightSampler", ids, k2, seed=1, call_counter=1))
bucketed, order, send, recv, most = store._route(ids)
T("  _route: partition + count all_gather + sync", lambda: store._route(ids))
T("  a2a ids (13 MB)", lambda: gdist._a2a(bucketed, send, recv, None, most))
nbr, eid = g.sample("EdgeWeightSampler", bucketed, k2, seed=1, call_counter=1, rng_rows=order)
T("  a2a nbr (131 MB)", lambda: gdist._a2a(nbr, recv, send, None, most))
T("  stitch nbr (131 MB)", lambda: glx.stitch(nbr, order))
T("  torch.empty + copy 131 MB (reference point)", lambda: nbr.clone())
dist.destroy_process_group()

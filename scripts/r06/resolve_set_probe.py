"""The resolve + dedup passes of a partitioned aggregation at the headline's P = 8 shape, in isolation, for A/B inside
one process: only rank 0 asks (the 18 M ids of one step: hop-2 + hop-1 neighbours), ranks 1..7 are threads that take
part in the collectives with empty requests and serve rank 0's halo rows.  Run under rocprofv3 --kernel-trace and feed
the trace to scripts/r06/resolve_set_parse.py: phases are REPS consecutive aggregate_begin calls each, in the order of
PHASES below (knobs: the halo id set's minimum size in 1/1024 of the request, the plain load before the CAS, own ids before the replica)."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth

P, REPS = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 6
# (name, set share / 1024, peek, own_first, workgroups (0 = default 1024), ids per thread per pass (0 = default 2))
PHASES = [("share=16,peek=1", 16, 1, 1, 0, 0), ("share=64,peek=1", 64, 1, 1, 0, 0), ("share=256,peek=1", 256, 1, 1, 0, 0),
          ("share=16,peek=0", 16, 0, 1, 0, 0), ("share=64,peek=0", 64, 0, 1, 0, 0), ("share=256,peek=0", 256, 0, 1, 0, 0),
          ("own_first=0", 16, 1, 0, 0, 0), ("own_first=1", 16, 1, 1, 0, 0), ("own_first=0", 16, 1, 0, 0, 0), ("own_first=1", 16, 1, 1, 0, 0),
          ("blocks=2048", 16, 1, 0, 2048, 0), ("blocks=4096", 16, 1, 0, 4096, 0), ("blocks=8192", 16, 1, 0, 8192, 0),
          ("blocks=768,ids=4", 16, 1, 0, 768, 4), ("blocks=1024,ids=4", 16, 1, 0, 0, 4), ("blocks=1280,ids=4", 16, 1, 0, 1280, 4),
          ("blocks=1536,ids=4", 16, 1, 0, 1536, 4), ("blocks=1792,ids=4", 16, 1, 0, 1792, 4), ("blocks=1024,ids=8", 16, 1, 0, 0, 8),
          ("blocks=1536,ids=8", 16, 1, 0, 1536, 8), ("blocks=1280", 16, 1, 0, 1280, 0), ("blocks=1536", 16, 1, 0, 1536, 0),
          ("blocks=1024", 16, 1, 0, 0, 0)]
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)
hot = torch.topk(torch.bincount(dst, minlength=V), V // 4).indices.to(torch.int64)
g = glx.Graph.from_edges(src, dst, w)
del src, dst, w
gen = torch.Generator(device=dev)
gen.manual_seed(1000)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=42, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=42, call_counter=1)
both = torch.cat([n2.view(-1), n1.view(-1)]).contiguous()
g.close()
X = synth.features_torch(V, D, 5, dev)
fshards = [glx.Features(X[r::P].contiguous(), ids=torch.arange(r, V, P, dtype=torch.int64, device=dev)) for r in range(P)]
del X
torch.cuda.empty_cache()
bar = threading.Barrier(P)
empty = torch.empty(0, dtype=torch.int64, device=dev)
stats = {}


def rank_main(r):
    comm = glx.Comm.local(777, 0, r, P)
    with torch.cuda.stream(torch.cuda.Stream(device=0)):
        st = glx.DistStore(comm, features=fshards[r])
        st.set_cache(hot)
        for pi, (name, share, peek, own_first, blocks, per) in enumerate(PHASES):
            bar.wait()
            if r == 0:
                glx.tune("resolve_set_share", share)
                glx.tune("resolve_peek", peek)
                glx.tune("resolve_own_first", own_first)
                glx.tune("resolve_blocks", blocks if blocks else -1)
                glx.tune("resolve_ids", per if per else -1)
            bar.wait()
            for _ in range(REPS):
                st.aggregate_begin(0, both if r == 0 else empty)
            torch.cuda.current_stream().synchronize()
            if r == 0:
                stats[pi] = st.stats()
        st.close()
    comm.close()


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in ts: t.start()
for t in ts: t.join(600)
print("phases:", [p[0] for p in PHASES], "reps", REPS)
print("last request:", stats.get(len(PHASES) - 1))
print("own_first=0 request:", stats.get(len(PHASES) - 2))

"""Kernel breakdown of the filtered alias samplers (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch, glx, synth
dev = torch.device("cuda", 0)
V, E, B0, K = 1 << 20, 32 << 20, 65536, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
tsgen = torch.Generator(device=dev); tsgen.manual_seed(5)
ts = torch.randperm(E, device=dev, generator=tsgen)
pool = torch.unique(src)
g = glx.Graph.from_edges(src, dst, w, timestamp=ts)
g.enable_id_index()
del src, dst, w
gen = torch.Generator(device=dev); gen.manual_seed(1)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
hop1, _ = g.sample("RandomSampler", seeds, K, seed=1)
rows = hop1.reshape(-1).contiguous()
back = seeds.repeat_interleave(K).contiguous()
median_ts = torch.full_like(back, E // 2)
which = sys.argv[1] if len(sys.argv) > 1 else "ts"
for _ in range(3):
    if which == "ts":
        g.sample_filtered("EdgeWeightSampler", rows, K, glx.FILTER_LARGER_THAN, glx.FILTER_FIELD_TIMESTAMP, median_ts, seed=1, call_counter=3)
    else:
        g.sample_filtered(os.environ.get("PROBE_SAMPLER", "EdgeWeightSampler"), rows, K, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, back, seed=1, call_counter=3)
torch.cuda.synchronize()

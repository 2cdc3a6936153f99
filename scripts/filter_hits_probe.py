"""id == value filters in front of Topk / RandomWithoutReplacement: how many rows have how many hits (parallel edges
back to the filtered id), and the kernel breakdown (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch, glx, synth
dev = torch.device("cuda", 0)
V, E, B0, K = 1 << 20, 32 << 20, 65536, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
pool = torch.unique(src)
key, cnt = torch.unique(src * V + dst, return_counts=True)
g = glx.Graph.from_edges(src, dst, w)
g.enable_id_index()
del src, dst, w
gen = torch.Generator(device=dev); gen.manual_seed(1)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
hop1, _ = g.sample("RandomSampler", seeds, K, seed=1)
rows = hop1.reshape(-1).contiguous()
back = seeds.repeat_interleave(K).contiguous()
q = rows * V + back
at = torch.searchsorted(key, q).clamp(max=key.shape[0] - 1)
H = torch.where(key[at] == q, cnt[at], torch.zeros_like(q))
deg = g.degrees(rows)
print("rows", rows.shape[0], "with hits", int((H > 0).sum()), "H>8", int((H > 8).sum()), "H>32", int((H > 32).sum()),
      "H>256", int((H > 256).sum()), "H>4096", int((H > 4096).sum()), "max H", int(H.max()))
big = H > 8
print("rows with H>8: sum of degrees", int(deg[big].sum()), "max degree", int(deg[big].max()) if big.any() else 0,
      "distinct rows", int(torch.unique(rows[big]).shape[0]), "sum of H", int(H[big].sum()))
which = os.environ.get("PROBE_SAMPLER", "TopkSampler")
for _ in range(5):
    g.sample_filtered(which, rows, K, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, back, seed=1, call_counter=3)
torch.cuda.synchronize()

"""How much of C3's topology the graph replica holds and how many request rows it serves, by hot fraction
(hot set = top vertices by global in-degree, as glx_dist_hot_ids picks them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch, glx, synth
dev = torch.device("cuda", 0)
V, E, B0, k1, k2 = 10_000_000, 100_000_000, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)
indeg = torch.bincount(dst, minlength=V)
order = torch.argsort(indeg, descending=True, stable=True)
g = glx.Graph.from_edges(src, dst, w)
gen = torch.Generator(device=dev); gen.manual_seed(3)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample("EdgeWeightSampler", n1.view(-1), k2, seed=1, call_counter=1)
hop2_rows, hop2_ids = n1.view(-1), n2.view(-1)
print("| hot fraction of the vertices | edges in the graph replica | hop-1 request rows served locally | hop-2 request rows served locally | hop-2 feature ids in the feature replica |")
print("|---|---|---|---|---|")
for frac in (0.001, 0.01, 0.05, 0.10, 0.25, 0.50):
    hot = torch.zeros(V, dtype=torch.bool, device=dev)
    hot[order[: int(V * frac)]] = True
    print("| %.1f %% | %.1f %% | %.1f %% | %.1f %% | %.1f %% |" % (
        100 * frac, 100 * hot[src].float().mean().item(), 100 * hot[seeds].float().mean().item(),
        100 * hot[hop2_rows.clamp(0, V - 1)].float().mean().item(), 100 * hot[hop2_ids.clamp(0, V - 1)].float().mean().item()))

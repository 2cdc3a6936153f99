"""PCIe-inclusive rate of the host-pointer boundary (GLX_PTR_HOST: what the C++
operators' Process() uses): numpy in / numpy out on the C3 store."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)  # seeds = vertices that have out-edges (as in bench.py)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
f = glx.Features(synth.features_torch(V, D, 5, dev))
rng = np.random.default_rng(0)
seeds = pool.cpu().numpy()[rng.integers(0, pool.shape[0], B0)]
n1 = np.empty((B0, k1), np.int64); e1 = np.empty_like(n1)
n2 = np.empty((B0 * k1, k2), np.int64); e2 = np.empty_like(n2)
emb = np.empty((B0 * k1, D), np.float32); cnt = np.empty(B0 * k1, np.int32)
seg = (np.arange(B0 * k1 * k2) // k2).astype(np.int32)
def step(i):
    g.sample("EdgeWeightSampler", seeds, k1, seed=1, call_counter=2 * i, out=(n1, e1))
    g.sample("EdgeWeightSampler", n1.reshape(-1), k2, seed=1, call_counter=2 * i + 1, out=(n2, e2))
    t = time.perf_counter()
    f.aggregate("MaxAggregator", n2.reshape(-1), seg, B0 * k1, out=(emb, cnt))
    return time.perf_counter() - t
step(0)
t0 = time.perf_counter(); ta = 0.0
for i in range(1, 4): ta += step(i)
dt = (time.perf_counter() - t0) / 3
slots = B0 * k1 + B0 * k1 * k2
print(json.dumps({"host_pointer_step_ms": dt * 1e3, "sampling_ms": (dt - ta / 3) * 1e3, "aggregate_ms": ta / 3 * 1e3,
                  "edges_per_s_pcie_inclusive": slots / dt,
                  "bytes_over_pcie_per_step": int(seeds.nbytes + n1.nbytes * 3 + e1.nbytes + n2.nbytes * 3 + e2.nbytes + seg.nbytes + emb.nbytes + cnt.nbytes)}))

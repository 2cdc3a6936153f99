"""Device timings of filtered vs unfiltered neighbour sampling (hop 2 of a [10, 10] request, B0 = 65536) on an
RMAT graph: the cost of op::Filter on the device (rows + scan + reserve + per-strategy draw).  One JSON line per case."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch  # noqa: E402
import glx  # noqa: E402
import synth  # noqa: E402

dev = torch.device("cuda", 0)
V, E, B0, K = 1 << 20, 32 << 20, 65536, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
tsgen = torch.Generator(device=dev)
tsgen.manual_seed(5)
ts = torch.randperm(E, device=dev, generator=tsgen)
pool = torch.unique(src)
g = glx.Graph.from_edges(src, dst, w, timestamp=ts)
g.enable_in_degree()
del src, dst, w
gen = torch.Generator(device=dev)
gen.manual_seed(1)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
hop1, _ = g.sample("RandomSampler", seeds, K, seed=1)
rows = hop1.reshape(-1).contiguous()
# GSL's .filter(seed): never walk back to the vertex the path came from
back = seeds.repeat_interleave(K).contiguous()
median_ts = torch.full_like(back, E // 2)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


slots = rows.shape[0] * K
print(json.dumps({"graph": "rmat V=2^20 E=2^25", "rows": int(rows.shape[0]), "k": K,
                  "sum_of_row_degrees": int(g.degrees(rows).sum()), "max_row_degree": int(g.degrees(rows).max())}))
if "--index" in sys.argv:
    g.enable_id_index()  # per-row id-sorted index: id == value filters find their hits by binary search
print(json.dumps({"id_sorted_row_index": "--index" in sys.argv}))
for name in list(glx.SAMPLER_IDS) + ["InDegreeSampler"]:
    base = timed(lambda: g.sample(name, rows, K, seed=1, call_counter=3))
    by_id = timed(lambda: g.sample_filtered(name, rows, K, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, back, seed=1, call_counter=3))
    by_ts = timed(lambda: g.sample_filtered(name, rows, K, glx.FILTER_LARGER_THAN, glx.FILTER_FIELD_TIMESTAMP, median_ts,
                                            seed=1, call_counter=3))
    print(json.dumps({"op": name, "rows": int(rows.shape[0]), "k": K, "unfiltered_ms": base, "id_equal_ms": by_id,
                      "timestamp_larger_ms": by_ts, "filtered_edges_per_s": slots / (by_id * 1e-3)}))

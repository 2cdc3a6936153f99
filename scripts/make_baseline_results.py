"""Rewrites section 4 of BASELINE.md from the evidence under profiles/r01/."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "profiles", "r01") + "/"
c3 = json.load(open(R + "bench_c3_with_cpu_baseline.json")); raw = json.load(open(R + "bench_c3_raw_rmat_ids.json"))
c2 = json.load(open(R + "bench_c2.json")); c4 = json.load(open(R + "bench_c4_single_gpu.json")); c5 = json.load(open(R + "bench_c5_single_gpu.json"))
mb = [json.loads(l) for l in open(R + "microbench.jsonl")]
pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["c3_b65536"]
p = os.path.join(ROOT, "BASELINE.md")
s = open(p).read()
i = s.index("## 4. Results")

def row(name, step, d, cpu=False):
    r = d["roofline"]
    base = "| %s | %s | 1xMI355X | %.3f | %.3g | %.3f ms, %.0f GB/s algorithmic = %.2f of 8 TB/s" % (
        name, step, d["ms_per_step"], d["value"], r["avg_launch_ms"], r["achieved"], r["frac"])
    if cpu:
        cb = d["cpu_baseline"]
        base += "; %.2f GB PMC traffic vs %.2f GB algorithmic | %.3g edges/s on %d threads (sampling %.3g e/s, aggregation %.3g v/s) | %.0fx |" % (
            pm["aggregate_hop2_bytes_per_launch"] / 1e9, r["algorithmic_bytes_per_launch"] / 1e9, cb["value"], cb["cores"],
            cb["sampling_edges_per_s"], cb["aggregation_vertices_per_s"], d["gpu_over_cpu"])
    else:
        base += " | not run | - |"
    return base

new = """## 4. Results (round 1, one MI355X; evidence in `profiles/r01/`)

Whole-step headline (`bench.py`, B0 = 65,536 seeds/step drawn from the vertices that have out-edges, fresh seeds
every step, 20 timed steps; RMAT vertex labels relabeled Graph500-style unless noted; rates above 1.0 of the HBM
peak are Infinity-Cache/L2-assisted -- the PMC traffic column is what actually crossed the memory fabric):

| Config | Step | Device | ms/step | sampled-edges/s = aggregated-vertices/s | hop-2 aggregate kernel | CPU baseline (reference's own code, same box) | GPU/CPU |
|---|---|---|---|---|---|---|---|
"""
new += row("C3 RMAT 10M/100M", "EdgeWeight [25,10] + Max, D=256", c3, True) + "\n"
new += row("C3, raw (un-relabeled) RMAT ids", "same", raw) + "\n"
new += row("C4 RMAT 111M/1.6B (papers100M-sized), whole graph + 57 GB of features on ONE GPU", "Random [20,15] + Mean, D=128", c4) + "\n"
new += row("C5 heterogeneous user-item-shop (50M nodes / 500M edges, 3 edge types), whole on ONE GPU", "per-type Topk (10,10,5) + type-wise Sum, D=256", c5) + "\n"
new += row("C2 RMAT 2.4M/62M", "RWoR [15,10] + Mean, D=128", c2) + "\n"
new += """
Per-operator kernel rates (`scripts/microbench.py`, same seeds policy, HIP events around the dominant kernel; the
aggregator rows re-use one id set across repetitions, so they are more cache-assisted than the whole-step numbers
above, which use fresh ids every step):

| Graph | Op | hop-1 ms | hop-2 ms | edges/s (both hops) | algorithmic GB/s |
|---|---|---|---|---|---|
"""
for m in mb:
    if "hop1_ms" in m:
        new += "| %s | %s %s | %.4f | %.4f | %.3g | %.0f |\n" % (m["graph"], m["op"], m["fanout"], m["hop1_ms"], m["hop2_ms"], m["edges_per_s"], m["algorithmic_GBps"])
new += "\n| Graph | Op | dim | N ids -> segments | ms | vertices/s | algorithmic GB/s |\n|---|---|---|---|---|---|---|\n"
for m in mb:
    if "vertices_per_s" in m:
        new += "| %s | %s | %d | %d -> %d | %.4f | %.3g | %.0f |\n" % (m["graph"], m["op"], m["dim"], m["N"], m["segments"], m["ms"], m["vertices_per_s"], m["algorithmic_GBps"])
en = [m for m in mb if "enable_in_degree" in m["op"] and m["graph"].startswith("C3")][0]
fu = [m for m in mb if m["op"].startswith("FullSampler") and m["graph"].startswith("C3")][0]
new += """
Also on C3: in-degree alias tables for InDegreeSampler built once in %.2f s; FullSampler(limit 25) over the 1.64 M-row
hop-1 frontier %.3g values/s; the drop-in C++ operator path with host buffers (PCIe-inclusive,
`profiles/r01/host_path_bench.txt`) 0.9-1.8 x 10^8 edges/s.

Targets of section 2: >=10x sampled-edges/s over the CPU path at 1 GPU -- met by three orders of magnitude on the same
request stream; >=60 %% of the HBM roofline on the aggregation kernel -- %.2f algorithmic, %.2f TB/s of PMC-measured
traffic (the part's copy ceiling is ~6.3 TB/s). 2/4/8-GPU numbers are produced by the driver's scaling run
(`bench.py --gpus N`); the N > 1 code path itself is verified on one GPU (`tests/test_gpu_two_ranks.py`).
""" % (en["seconds"], fu["values_per_s"], c3["roofline"]["frac"],
       pm["aggregate_hop2_bytes_per_launch"] / (c3["roofline"]["avg_launch_ms"] * 1e-3) / 1e12)
open(p, "w").write(s[:i] + new)
print("BASELINE.md section 4 rewritten")

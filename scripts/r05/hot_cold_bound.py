"""VERDICT r04 'Next round' item 3 -- the hot/cold two-phase reduce on C3's hop-2 aggregate, measured with the kernels
that exist before anything new is written (one gpurun budget; a documented negative result is an outcome).

The proposal: feature rows ranked by in-degree; phase A reduces only the segment members among the top-H rows with
8 XCD-affine column slices (each XCD's L2 then holds one 128-byte slice of every hot row: 8 x 4 MB = 32 K rows of
1 KB instead of the 4 K rows one L2 holds when every XCD reads whole rows), phase B reduces the cold members with
whole-row loads and folds A's partial result in.  Both phases can be timed TODAY:
  A  = the grouped kernel with agg_xcd_slices = 8 on the request with every cold id replaced by -1 (an unknown id:
       no row is gathered for it beyond the L2-resident row 0 the kernel reads in its place); it writes the
       [segments, D] partial result exactly as phase A would;
  B' = the default kernel on the request with every hot id replaced by -1: phase B without its read of A's partial
       result, i.e. a LOWER bound of phase B (the missing 1.68 GB read is timed separately as a streaming read);
so  A + B' (+ partial read)  bounds the two-phase design from below, against the single launch it would replace.
Also printed: the share of the request's accesses that go to the top-H rows, and the ideal case in which hot rows
cost nothing at all (every hot id replaced by ONE row: always an L2 hit) -- the most any hot-row scheme could gain.
Max is associative and exact, so max(A, B') must equal the single launch bit for bit: checked.  (An unknown id
contributes the default attribute -- Aggregator::Aggregate reduces the type's default row for it -- so the masked
requests run with default_attr = -3e38, below MaxAggregator's initial -37: a no-op under max.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import glx  # noqa: E402
import synth  # noqa: E402

dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2, gseed = 10_000_000, 100_000_000, 256, 65536, 25, 10, 4
smp, agg = "EdgeWeightSampler", "MaxAggregator"
src, dst, w = synth.rmat_edges_torch(V, E, gseed, dev, weighted=True)
indeg = torch.bincount(dst, minlength=V)
pool = torch.unique(src)
g = glx.Graph.from_edges(src, dst, w)
del src, dst, w
f = glx.Features(synth.features_torch(V, D, gseed + 1, dev))
gen = torch.Generator(device=dev)
gen.manual_seed(3)
Sg = B0 * k1
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev)
cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample(smp, seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample(smp, n1.view(-1), k2, seed=1, call_counter=1)
ids = n2.view(-1).contiguous()
rank = torch.argsort(indeg, descending=True)


def timed(i, reps=7, **knobs):
    for k in ("agg_xcd_slices", "agg_unroll", "agg_segs"):
        glx.tune(k, knobs.get(k, 0))
    r = []
    for _ in range(reps):
        torch.cuda.synchronize()
        glx.profile_enable(True)
        f.aggregate(agg, i, None, Sg, default_attr=-3.0e38, out=(emb, cnt))
        torch.cuda.synchronize()
        glx.profile_enable(False)
        r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE).sum()))
    for k in ("agg_xcd_slices", "agg_unroll", "agg_segs"):
        glx.tune(k, 0)
    return float(np.median(r)), emb.clone()


base_ms, base_out = timed(ids)
read = glx.probe_bandwidth("stream_read", (Sg * D * 4 // 4096) * 4096, reps=10, device=0)
print("single launch (today)            : %.3f ms   [%d ids -> %d segments, D = %d]" % (base_ms, ids.numel(), Sg, D))
print("streaming read of one [segments, D] partial result (%.2f GB): %.3f ms at %.0f GB/s" % (Sg * D * 4 / 1e9, read["ms"], read["gbps"]))
print("%8s %9s | %8s %8s %8s | %9s %9s | %s" % ("H", "hot share", "A x8", "B' (x2)", "A+B'", "+partial", "vs today", "ideal: hot rows free"))
for H in (4096, 32768, 262144, 1048576):
    hot = torch.zeros(V, dtype=torch.bool, device=dev)
    hot[rank[:H]] = True
    is_hot = hot[ids.clamp(0, V - 1)]
    share = float(is_hot.double().mean().item())
    ids_hot = torch.where(is_hot, ids, torch.full_like(ids, -1))
    ids_cold = torch.where(is_hot, torch.full_like(ids, -1), ids)
    a_ms, a_out = timed(ids_hot, agg_xcd_slices=8)
    b_ms, b_out = timed(ids_cold)
    ok = bool(torch.equal(torch.maximum(a_out, b_out).view(torch.int32), base_out.view(torch.int32)))
    ideal_ms, _ = timed(torch.where(is_hot, torch.full_like(ids, int(rank[0])), ids))
    total = a_ms + b_ms + read["ms"]
    print("%8d %8.1f%% | %8.3f %8.3f %8.3f | %9.3f %8.2fx | %.3f ms (%.2fx)  max(A,B') == single launch: %s"
          % (H, 100 * share, a_ms, b_ms, a_ms + b_ms, total, total / base_ms, ideal_ms, ideal_ms / base_ms, ok))
    del hot, is_hot, ids_hot, ids_cold, a_out, b_out

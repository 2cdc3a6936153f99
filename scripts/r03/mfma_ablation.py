"""Row g ablation (north_star's "dense feature tile staged in LDS, MFMA on the tile"): the fixed-fanout Sum / Mean
reduce as (a) the VALU kernel of the product path and (b) the LDS-staged v_mfma_f32_16x16x4_f32 formulation
(GLX_AGG_MFMA=1), on the C2 hop-2 request (ogbn-products shape: RMAT 2.4 M / 62 M, RWoR [15, 10], dim 128, 65,536 seeds)
and on a dim-256 variant.  Prints kernel times from HIP events (glx_profile) and checks bit equality of the outputs.
Counters (SQ_INSTS_VALU_MFMA_MOPS_F32, FETCH_SIZE) come from the rocprofv3 --pmc passes of this same script
(scripts/r03/run1.sh)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import glx  # noqa: E402
import synth  # noqa: E402

dev = torch.device("cuda", 0)
V, E, B0, k1, k2 = 2_400_000, 62_000_000, 65536, 15, 10
src, dst, _ = synth.rmat_edges_torch(V, E, 2, dev, weighted=False)
g = glx.Graph.from_edges(src, dst, None, device=0)
pool = torch.unique(src)
del src, dst
gen = torch.Generator(device=dev)
gen.manual_seed(1000)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample("RandomWithoutReplacementSampler", seeds, k1, seed=42, call_counter=0)
n2, _ = g.sample("RandomWithoutReplacementSampler", n1.view(-1), k2, seed=42, call_counter=1)
ids = n2.view(-1)
Sg = n1.numel()
reps = int(os.environ.get("ABLATION_REPS", "10"))
for D in (128, 256):
    f = glx.Features(synth.features_torch(V, D, 3, dev), device=0)
    bytes_alg = ids.numel() * (4 * D + 12) + Sg * (4 * D + 4)
    for op in ("SumAggregator", "MeanAggregator"):
        outs = {}
        for knob in ("0", "1"):
            glx.tune("agg_mfma", int(knob))
            emb = torch.empty((Sg, D), dtype=torch.float32, device=dev)
            cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
            for _ in range(3):
                f.aggregate(op, ids, None, Sg, out=(emb, cnt))
            torch.cuda.synchronize()
            glx.profile_enable(True)
            for _ in range(reps):
                f.aggregate(op, ids, None, Sg, out=(emb, cnt))
            torch.cuda.synchronize()
            glx.profile_enable(False)
            ms = float(np.mean(glx.profile_collect(glx.KERNEL_AGGREGATE)))
            outs[knob] = (emb.clone(), cnt.clone())
            print("dim %3d %-15s %-28s %.3f ms  %.0f GB/s algorithmic (%.2f of 8 TB/s)"
                  % (D, op, "LDS + MFMA 16x16x4 f32" if knob == "1" else "VALU (product path)", ms,
                     bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 8000), flush=True)
        same = torch.equal(outs["0"][0].view(torch.int32), outs["1"][0].view(torch.int32)) and torch.equal(outs["0"][1], outs["1"][1])
        print("dim %3d %-15s outputs bit-identical: %s" % (D, op, same), flush=True)
    del f
    torch.cuda.empty_cache()
glx.tune("agg_mfma", 0)

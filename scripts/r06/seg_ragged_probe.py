"""Explicit segment_ids at the headline's size: what the one-pass bookkeeping (glx_seg_scan_kernel + fix-up) costs for
  dense    : the tensor spells out the dense sampler response (level 0: arithmetic bounds)
  ragged   : sorted random segment ids, n NOT a multiple of the segment count (the host knows it cannot be dense)
  ragged_div : sorted random segment ids, n a multiple of the segment count (only the scan can tell: level 1)
  swapped  : the dense layout with one id moved to its neighbour segment
against segment_ids = None.  Whole aggregate call, HIP events on the current stream, median of 7."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
V, D, Sg, f = 10_000_000, 256, 1_638_400, 10
n = Sg * f
feats = glx.Features(synth.features_torch(V, D, 5, dev))
gen = torch.Generator(device=dev); gen.manual_seed(1)
ids = torch.randint(0, V, (n,), generator=gen, device=dev)
dense = (torch.arange(n, device=dev) // f).to(torch.int32)
ragged_div = torch.sort(torch.randint(0, Sg, (n,), generator=gen, device=dev)).values.to(torch.int32)
ragged = ragged_div[: n - 3].contiguous()
swapped = dense.clone(); swapped[f - 1] = 1
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
def t(seg, m):
    r = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); feats.aggregate("SumAggregator", ids[:m], seg, Sg, out=(emb, cnt)); b.record(); torch.cuda.synchronize()
        r.append(a.elapsed_time(b))
    return float(np.median(r[1:]))
print("segment_ids = None       %.3f ms" % t(None, n))
for name, seg, m in (("dense", dense, n), ("swapped", swapped, n), ("ragged_div", ragged_div, n), ("ragged", ragged, n - 3)):
    print("%-24s %.3f ms" % (name, t(seg, m)), flush=True)

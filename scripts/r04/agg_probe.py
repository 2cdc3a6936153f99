"""A/B of the aggregation kernels inside one process (glx_tune): the round-3 kernel (agg_legacy) against the grouped
kernel at several rows-in-flight (agg_unroll) / segments-per-group (agg_segs) / XCD column slice settings, on
  real   : the C3 hop-2 request (16.4 M ids -> 1.64 M segments of 10, D = 256) and its hop-1 request (f = 25)
  l2     : the same shape with ids uniform over 2048 rows (2 MB: L2 resident) -- the non-memory floor
  mall   : ids uniform over 100,000 rows (102 MB at D = 256: Infinity-Cache resident, far beyond the L2s) -- the fabric ceiling
  uniform: ids uniform over all 10 M rows (cache-free: algorithmic bytes == HBM traffic)
Every variant's output is compared bit for bit with the legacy kernel's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
if wl == "c3":
    V, E, D, B0, k1, k2, gseed, smp, agg = 10_000_000, 100_000_000, 256, 65536, 25, 10, 4, "EdgeWeightSampler", "MaxAggregator"
elif wl == "c2":
    V, E, D, B0, k1, k2, gseed, smp, agg = 2_400_000, 62_000_000, 128, 65536, 15, 10, 2, "RandomWithoutReplacementSampler", "MeanAggregator"
else:  # c4-like fanout on a smaller graph
    V, E, D, B0, k1, k2, gseed, smp, agg = 10_000_000, 100_000_000, 128, 65536, 20, 15, 6, "RandomSampler", "MeanAggregator"
src, dst, w = synth.rmat_edges_torch(V, E, gseed, dev, weighted=(smp == "EdgeWeightSampler"))
pool = torch.unique(src)
g = glx.Graph.from_edges(src, dst, w); del src, dst, w
f = glx.Features(synth.features_torch(V, D, gseed + 1, dev))
gen = torch.Generator(device=dev); gen.manual_seed(3)
Sg = B0 * k1
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
emb1 = torch.empty((B0, D), dtype=torch.float32, device=dev); cnt1 = torch.empty(B0, dtype=torch.int32, device=dev)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
n1, _ = g.sample(smp, seeds, k1, seed=1, call_counter=0)
n2, _ = g.sample(smp, n1.view(-1), k2, seed=1, call_counter=1)
ids = {"real": n2.view(-1).contiguous(), "hop1": n1.view(-1).contiguous(),
       "l2": torch.randint(0, 2048, (Sg * k2,), generator=gen, device=dev),
       "mall": torch.randint(0, 100_000, (Sg * k2,), generator=gen, device=dev),
       "uniform": torch.randint(0, V, (Sg * k2,), generator=gen, device=dev)}
def t(name, reps=7):
    i, sg, out = (ids[name], B0, (emb1, cnt1)) if name == "hop1" else (ids[name], Sg, (emb, cnt))
    r = []
    for _ in range(reps):
        torch.cuda.synchronize(); glx.profile_enable(True)
        f.aggregate(agg, i, None, sg, out=out)
        torch.cuda.synchronize(); glx.profile_enable(False)
        r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE).sum()))
    return float(np.median(r)), out[0].clone()
KNOBS = ("agg_legacy", "agg_unroll", "agg_segs", "agg_xcd_slices", "agg_occupancy", "agg_store")
def setk(**kw):
    for k in KNOBS:
        glx.tune(k, kw.get(k, 0))
def parse(spec):  # "x4,s1,u15" -> knobs
    kw = {}
    for tok in spec.split(","):
        if tok == "legacy": kw["agg_legacy"] = 1
        elif tok[0] == "x": kw["agg_xcd_slices"] = int(tok[1:])
        elif tok[0] == "s": kw["agg_segs"] = int(tok[1:])
        elif tok[0] == "u": kw["agg_unroll"] = int(tok[1:])
        elif tok[0] == "o": kw["agg_occupancy"] = int(tok[1:])
        elif tok[0] == "w": kw["agg_store"] = int(tok[1:])
    return kw
specs = sys.argv[2].split(":") if len(sys.argv) > 2 else ["legacy", "default", "u6", "u8", "u10", "u12", "u15", "s1", "s2", "s3", "s6", "s12",
                                                         "x8", "x4", "x2", "legacy"]
variants = [(sp, parse(sp)) for sp in specs]
ref = {}
print("# %s: D=%d, hop-2 fanout %d (%d segments), hop-1 fanout %d; median of 7 launches, ms" % (wl, D, k2, Sg, k1))
print("%-22s %9s %9s %9s %9s %9s  bit-identical" % ("variant", "real", "hop1", "l2", "mall", "uniform"))
for label, kw in variants:
    setk(**kw)
    row, same = [], True
    for name in ("real", "hop1", "l2", "mall", "uniform"):
        ms, out = t(name)
        if name not in ref:
            ref[name] = out
        same = same and bool(torch.equal(out.view(torch.int32), ref[name].view(torch.int32)))
        row.append(ms)
    print("%-22s %9.3f %9.3f %9.3f %9.3f %9.3f  %s" % (label, *row, same), flush=True)
setk()

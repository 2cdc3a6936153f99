"""What bounds C5's dominant launch: the i-s hop's SumAggregator over the 1 M-row shop table
(6.55 M ids -> 655,360 segments of 10, D = 256) -- glx_aggregate_grp_kernel<0, 32, 4, 10, 1, 1>.

  python scripts/r06/c5_is_probe.py sweep      variants of the launch (glx_tune, one process) on five id streams of the
                                               SAME shape, every output compared bit for bit with the default's:
      real    : the request the workload issues (Topk answers of the u-i hop's items: deterministic duplicates)
      l2      : ids uniform over 2,048 rows (2 MB: resident in every XCD's L2)      -- the non-memory floor
      mall    : ids uniform over 100,000 rows (102 MB: Infinity-Cache resident)      -- the fabric ceiling
      uniform : ids uniform over all 1 M rows (1 GB: beyond the 256 MB Infinity Cache) -- the table's HBM case
      sorted  : the real request's SEGMENTS ordered by their first id (neighbouring groups share rows)
  `default` is what the library picks for this shape (round 6: three segments per lane group for tables <= 1 GiB);
  s1 = one segment per group (rounds 4-5), xN = XCD-affine column slices, uN = rows in flight, oN = occupancy cap.
  python scripts/r06/c5_is_probe.py pmc [N]    N launches of `real` at the default knobs and N of `l2`, nothing else in
                                               the timed part (run under rocprofv3 --pmc ...: counters per launch)
Distinct rows / lines per request are printed so the traffic figures can be read against a byte count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth

mode = sys.argv[1] if len(sys.argv) > 1 else "sweep"
dev = torch.device("cuda", 0)
D, B0, k1, k2 = 256, 65536, 10, 10
n_user, n_item, n_shop = 40_000_000, 9_000_000, 1_000_000
graphs = {}
for i, (t, (ns, nd, ne)) in enumerate({"u-i": (n_user, n_item, 300_000_000), "i-s": (n_item, n_shop, 100_000_000)}.items()):
    src, dst, w = synth.rmat_edges_torch(1 << 26, ne, 20 + i, dev, weighted=True)  # bench.py's c5 streams
    src %= ns
    dst %= nd
    if t == "u-i":
        pool = torch.unique(src)
    graphs[t] = glx.Graph.from_edges(src, dst, w)
    del src, dst, w
x_shop = glx.Features(synth.features_torch(n_shop, D, 32, dev))
torch.cuda.empty_cache()
gen = torch.Generator(device=dev)
gen.manual_seed(7)
seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
a1, _ = graphs["u-i"].sample("TopkSampler", seeds, k1)
a2, _ = graphs["i-s"].sample("TopkSampler", a1.view(-1), k2)
del graphs
torch.cuda.empty_cache()
Sg, N = B0 * k1, B0 * k1 * k2
real = a2.view(-1).contiguous()
order = torch.argsort(a2[:, 0], stable=True)
ids = {"real": real, "l2": torch.randint(0, 2048, (N,), generator=gen, device=dev),
       "mall": torch.randint(0, 100_000, (N,), generator=gen, device=dev),
       "uniform": torch.randint(0, n_shop, (N,), generator=gen, device=dev),
       "sorted": a2[order].view(-1).contiguous()}
emb = torch.empty((Sg, D), dtype=torch.float32, device=dev)
cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
uniq_items = int(torch.unique(a1).numel())
for name in ("real", "uniform"):
    u = int(torch.unique(ids[name]).numel())
    print("# %-8s %d ids, %d distinct shop rows (%.1f MB of rows); %d distinct items among the %d request rows"
          % (name, N, u, u * D * 4 / 1e6, uniq_items, Sg), flush=True)
alg = N * (4 * D + 12) + Sg * (4 * D + 4)
print("# algorithmic bytes per launch: %.3f GB (%.3f GB of it written)" % (alg / 1e9, Sg * (4 * D + 4) / 1e9))


def launch(name, reps):
    r = []
    for _ in range(reps):
        torch.cuda.synchronize()
        glx.profile_enable(True)
        x_shop.aggregate("SumAggregator", ids[name], None, Sg, out=(emb, cnt))
        torch.cuda.synchronize()
        glx.profile_enable(False)
        r.append(float(glx.profile_collect(glx.KERNEL_AGGREGATE).sum()))
    return r


if mode == "pmc":
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    for name in ("real", "l2", "uniform"):
        r = launch(name, n)
        print("%s: %s ms" % (name, ["%.3f" % x for x in r]), flush=True)
    sys.exit(0)

KNOBS = ("agg_legacy", "agg_unroll", "agg_segs", "agg_xcd_slices", "agg_occupancy", "agg_store")


def setk(**kw):
    for k in KNOBS:
        glx.tune(k, kw.get(k, 0))


def parse(spec):
    kw = {}
    for tok in spec.split(","):
        if tok == "legacy": kw["agg_legacy"] = 1
        elif tok[0] == "x": kw["agg_xcd_slices"] = int(tok[1:])
        elif tok[0] == "s": kw["agg_segs"] = int(tok[1:])
        elif tok[0] == "u": kw["agg_unroll"] = int(tok[1:])
        elif tok[0] == "o": kw["agg_occupancy"] = int(tok[1:])
        elif tok[0] == "w": kw["agg_store"] = int(tok[1:])
    return kw


specs = sys.argv[2].split(":") if len(sys.argv) > 2 else ["default", "s1", "s1,x1", "s1,x4", "s2", "s3", "s4", "s3,x1", "s6,x1",
                                                         "s1,u6", "s3,u6", "s1,o6", "s1,w1", "legacy", "default"]
names = ("real", "sorted", "l2", "mall", "uniform")
print("# median of 7 launches, ms; TB/s = algorithmic bytes / time on `real`")
print("%-12s " % "variant" + " ".join("%9s" % n for n in names) + "   TB/s(real)  bit-identical")
ref = {}
for sp in specs:
    setk(**parse(sp))
    row, same = [], True
    for name in names:
        ms = float(np.median(launch(name, 7)))
        if name not in ref:
            ref[name] = emb.clone()
        same = same and bool(torch.equal(emb.view(torch.int32), ref[name].view(torch.int32)))
        row.append(ms)
    print("%-12s " % sp + " ".join("%9.3f" % x for x in row) + "   %9.2f  %s" % (alg / row[0] / 1e9, same), flush=True)
setk()

"""Per-operator device timings on the C3 / C2 graphs (kernel-only, HIP events
around the dominant kernel via glx_profile_*).  Output: one JSON line per case."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth

dev = torch.device("cuda", 0)
def run(tag, V, E, D, fan, B0=65536, reps=10):
    src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
    pool = torch.unique(src)  # seeds = vertices that have out-edges (as in bench.py)
    g = glx.Graph.from_edges(src, dst, w)
    import time as _t
    t0 = _t.time(); g.enable_in_degree(); torch.cuda.synchronize()
    print(json.dumps({"graph": tag, "op": "glx_graph_enable_in_degree (one-time)", "seconds": _t.time() - t0}))
    del src, dst, w
    X = synth.features_torch(V, D, 5, dev)
    f = glx.Features(X); del X
    k1, k2 = fan
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
    n1 = torch.empty((B0, k1), dtype=torch.int64, device=dev); e1 = torch.empty_like(n1)
    n2 = torch.empty((B0 * k1, k2), dtype=torch.int64, device=dev); e2 = torch.empty_like(n2)
    for name in list(glx.SAMPLER_IDS) + ["InDegreeSampler"]:
        for pad in (1,):
            g.sample(name, seeds, k1, seed=1, call_counter=0, out=(n1, e1))
            g.sample(name, n1.view(-1), k2, seed=1, call_counter=1, out=(n2, e2))
            torch.cuda.synchronize(); glx.profile_enable(True)
            for r in range(reps):
                g.sample(name, seeds, k1, seed=1, call_counter=2 * r, out=(n1, e1))
                g.sample(name, n1.view(-1), k2, seed=1, call_counter=2 * r + 1, out=(n2, e2))
            torch.cuda.synchronize(); glx.profile_enable(False)
            t = glx.profile_collect(glx.KERNEL_SAMPLE)
            h1, h2 = float(np.mean(t[0::2])), float(np.mean(t[1::2]))
            slots = B0 * k1 + B0 * k1 * k2
            alg = slots * 32 + (B0 + B0 * k1) * 24 + (slots * 8 if name in ("EdgeWeightSampler", "InDegreeSampler") else 0)
            print(json.dumps({"graph": tag, "op": name, "fanout": fan, "B0": B0, "hop1_ms": h1, "hop2_ms": h2,
                              "edges_per_s": slots / ((h1 + h2) * 1e-3), "algorithmic_GBps": alg / ((h1 + h2) * 1e-3) / 1e9}))
    # negative samplers: k = 10 candidates for each of the B0 * k1 hop-1 vertices
    t0 = _t.time(); neg_u = glx.Negative.from_graph(g); neg_d = glx.Negative.from_graph(g, by_in_degree=True)
    g.enable_negative(); torch.cuda.synchronize()
    print(json.dumps({"graph": tag, "op": "negative tables + sorted adjacency (one-time)", "seconds": _t.time() - t0,
                      "candidates": neg_u.num_ids}))
    fr = n1.view(-1)
    for label, tab, ex in (("RandomNegativeSampler", neg_u, glx.NEG_EXCLUDE_NONE),
                           ("SoftInDegreeNegativeSampler", neg_d, glx.NEG_EXCLUDE_NONE),
                           ("InDegreeNegativeSampler", neg_d, glx.NEG_EXCLUDE_NEIGHBORS)):
        tab.sample(fr, 10, exclude=ex, graph=g, seed=1, call_counter=0)
        torch.cuda.synchronize(); glx.profile_enable(True)
        for r in range(reps):
            tab.sample(fr, 10, exclude=ex, graph=g, seed=1, call_counter=r)
        torch.cuda.synchronize(); glx.profile_enable(False)
        ms = float(np.mean(glx.profile_collect(glx.KERNEL_SAMPLE)))
        slots = fr.shape[0] * 10
        # algorithmic bytes per candidate: 8 (read id) + 8 (write) [+ 8 alias entry]; per row 8 (src id)
        alg = slots * (16 + (8 if tab.weighted else 0)) + fr.shape[0] * 8
        print(json.dumps({"graph": tag, "op": label, "rows": int(fr.shape[0]), "count": 10, "ms": ms,
                          "negatives_per_s": slots / (ms * 1e-3), "algorithmic_GBps": alg / (ms * 1e-3) / 1e9}))
    del neg_u, neg_d
    # FullSampler on the hop-1 frontier, limit 25 (sparse response)
    fr = n1.view(-1)
    g.sample_full(fr, 25); torch.cuda.synchronize(); t0 = _t.time()
    for r in range(reps):
        d, fn, fe = g.sample_full(fr, 25)
    torch.cuda.synchronize(); dt = (_t.time() - t0) / reps
    print(json.dumps({"graph": tag, "op": "FullSampler(limit 25)", "rows": int(fr.shape[0]), "values": int(fn.shape[0]),
                      "ms_incl_size_readback": dt * 1e3, "values_per_s": fn.shape[0] / dt}))
    ids = n2.view(-1); Sg = B0 * k1
    seg = (torch.arange(ids.shape[0], device=dev) // k2).to(torch.int32)
    emb = torch.empty((Sg, D), dtype=torch.float32, device=dev); cnt = torch.empty(Sg, dtype=torch.int32, device=dev)
    for name in glx.AGGREGATOR_IDS:
        f.aggregate(name, ids, seg, Sg, out=(emb, cnt))
        torch.cuda.synchronize(); glx.profile_enable(True)
        for r in range(reps):
            f.aggregate(name, ids, seg, Sg, out=(emb, cnt))
        torch.cuda.synchronize(); glx.profile_enable(False)
        t = glx.profile_collect(glx.KERNEL_AGGREGATE)
        ms = float(np.mean(t)); N = ids.shape[0]
        alg = N * (4 * D + 12) + Sg * (4 * D + 4)
        print(json.dumps({"graph": tag, "op": name, "dim": D, "N": N, "segments": Sg, "ms": ms,
                          "vertices_per_s": N / (ms * 1e-3), "algorithmic_GBps": alg / (ms * 1e-3) / 1e9,
                          "frac_of_8TBps": alg / (ms * 1e-3) / 8e12}))
    out = torch.empty((1 << 22, D), dtype=torch.float32, device=dev)
    lid = ids[: 1 << 22].contiguous()
    f.lookup(lid); torch.cuda.synchronize(); glx.profile_enable(True)
    for r in range(reps): f.lookup(lid)
    torch.cuda.synchronize(); glx.profile_enable(False)
    t = glx.profile_collect(glx.KERNEL_LOOKUP); ms = float(np.mean(t))
    print(json.dumps({"graph": tag, "op": "LookupNodes(float attrs)", "dim": D, "rows": 1 << 22, "ms": ms,
                      "algorithmic_GBps": (1 << 22) * (8 * D + 8) / (ms * 1e-3) / 1e9}))
    del g, f
    torch.cuda.empty_cache()

run("C3 RMAT 10M/100M", 10_000_000, 100_000_000, 256, (25, 10))
run("C2 RMAT 2.4M/62M", 2_400_000, 62_000_000, 128, (15, 10))

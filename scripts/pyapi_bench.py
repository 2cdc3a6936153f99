"""Measure the rows either side of the hot path on one GPU:
TSV loader throughput, device build time behind Graph.init(), and the per-step rate of
a 2-hop EdgeWeight [25, 10] sample through the three call paths:
  numpy   : NeighborSampler.get()  -- pywrap requests, host buffers, one request per hop (the reference's API)
  device  : NeighborSampler.get_device() -- torch CUDA tensors, all hops in one C-ABI call
Usage: python scripts/pyapi_bench.py [nodes] [edges] [batch]"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))

import torch  # noqa: E402
import graphlearn as gl  # noqa: E402
import synth  # noqa: E402


def main():
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
    D = 64
    dev = torch.device("cuda", 0)
    src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
    d = tempfile.mkdtemp(prefix="glx_pyapi_")
    import pandas as pd
    t0 = time.time()
    epath, npath = os.path.join(d, "edges"), os.path.join(d, "nodes")
    pd.DataFrame({"src_id:int64": src.cpu().numpy(), "dst_id:int64": dst.cpu().numpy(),
                  "weight:float": w.cpu().numpy()}).to_csv(epath, sep="\t", index=False, float_format="%.6f")
    X = np.random.default_rng(1).standard_normal((V, D)).astype(np.float32)
    with open(npath, "w") as f:
        f.write("id:int64\tfeature:string\n")
        for lo in range(0, V, 100000):
            hi = min(V, lo + 100000)
            rows = pd.DataFrame(X[lo:hi]).to_csv(sep=":", header=False, index=False, float_format="%.5f").split("\n")
            f.write("".join("%d\t%s\n" % (lo + i, r) for i, r in enumerate(rows[:hi - lo])))
    print("wrote %s (%.0f MB) and %s (%.0f MB) in %.1fs" % (epath, os.path.getsize(epath) / 1e6, npath,
                                                        os.path.getsize(npath) / 1e6, time.time() - t0))
    del src, dst, w
    gl.set_padding_mode(gl.CIRCULAR)
    g = gl.Graph().node(npath, "v", gl.Decoder(attr_types=["float"] * D)) \
        .edge(epath, ("v", "v", "e"), gl.Decoder(weighted=True))
    t0 = time.time()
    g.init()
    t_init = time.time() - t0
    mb = (os.path.getsize(epath) + os.path.getsize(npath)) / 1e6
    print("Graph.init(): %.2fs for %d edges + %d nodes x %d floats (%.0f MB of TSV -> %.0f MB/s incl. device build)"
          % (t_init, E, V, D, mb, mb / t_init))
    s = g.neighbor_sampler(["e", "e"], expand_factor=[25, 10], strategy="edge_weight")
    rng = np.random.default_rng(2)
    seeds = rng.integers(0, V, (12, B)).astype(np.int64)
    edges_per_step = B * 25 + B * 250
    for _ in range(2):
        s.get(seeds[0])
    t0 = time.time()
    for i in range(2, 12):
        layers = s.get(seeds[i])
        emb = layers.layer_nodes(2).embedding_agg("max")
    t_np = (time.time() - t0) / 10
    print("numpy path   : %.2f ms/step  %.3g sampled edges/s (2 sampling requests + 1 Max aggregation, host buffers)"
          % (t_np * 1e3, edges_per_step / t_np))
    feats = g.device_features("v")
    dseeds = torch.from_numpy(seeds).to(dev)
    seg = (torch.arange(B * 250, device=dev) // 10).to(torch.int32)
    for _ in range(3):
        s.get_device(dseeds[0])
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(2, 12):
        hops = s.get_device(dseeds[i], call_counter=2 * i)
        emb, cnt = feats.aggregate("MaxAggregator", hops[1][0].view(-1), seg, B * 25)
    torch.cuda.synchronize()
    t_dev = (time.time() - t0) / 10
    print("device path  : %.2f ms/step  %.3g sampled edges/s (1 multi-hop call + 1 Max aggregation, torch CUDA tensors)"
          % (t_dev * 1e3, edges_per_step / t_dev))
    # the loader: seeds -> 2 hops -> float attributes of all three frontiers, per batch, on the GPU
    loader = gl.NeighborLoader(g, "v", ["e", "e"], [25, 10], batch_size=B, strategy="edge_weight", shuffle=True)
    it = iter(loader)
    for _ in range(3):
        next(it)
    torch.cuda.synchronize()
    t0 = time.time()
    nb = 0
    for batch in it:
        nb += 1
        if nb == 40:
            break
    torch.cuda.synchronize()
    t_ld = (time.time() - t0) / nb
    rows = B * (1 + 25 + 250)
    print("NeighborLoader: %.2f ms/batch  %.3g sampled edges/s + %.3g feature rows/s gathered (%d floats each), device resident"
          % (t_ld * 1e3, edges_per_step / t_ld, rows / t_ld, D))
    # the same work as a GSL query read through graphlearn.nn.pytorch.Dataset (the reference's PyTorch entry:
    # python/nn/pytorch/data/dataset.py): host values converted to tensors, and device="cuda" (fused chain in HBM)
    import graphlearn.nn.pytorch as thg
    for device, label in ((None, "host values -> torch.from_numpy"), ("cuda", "device='cuda': fused chain, ids + floats stay in HBM")):
        q = g.V("v").batch(B).shuffle(traverse=True).alias("seed") \
             .outV("e").sample(25).by("edge_weight").alias("hop1") \
             .outV("e").sample(10).by("edge_weight").alias("hop2").values()
        it = iter(thg.Dataset(q, device=device))
        for _ in range(2):
            data = next(it)
        torch.cuda.synchronize()
        t0 = time.time()
        nb = 0
        for data in it:
            rows_seen = data["hop2"].float_attrs.shape[0] + data["hop1"].float_attrs.shape[0]
            nb += 1
            if nb == (40 if device else 5):
                break
        torch.cuda.synchronize()
        t_q = (time.time() - t0) / nb
        print("nn.pytorch.Dataset (%s): %.2f ms/batch  %.3g sampled edges/s + %.3g feature rows/s"
              % (label, t_q * 1e3, edges_per_step / t_q, rows_seen / t_q))
    # prefetching (gsl.Dataset(prefetch=True), the reference's `window`): the same loop with a stand-in model step on
    # the GPU (a [rows, 64] x [64, 256] product + relu, a few hundred microseconds) -- sampled ahead vs on demand
    W = torch.randn(D, 256, device=dev)
    W2 = torch.randn(256, D, device=dev)
    for prefetch in (False, True):
        q = g.V("v").batch(B).shuffle(traverse=True).alias("seed") \
             .outV("e").sample(25).by("edge_weight").alias("hop1") \
             .outV("e").sample(10).by("edge_weight").alias("hop2").values()
        ds = thg.Dataset(q, window=4, device="cuda", prefetch=prefetch)
        it = iter(ds)
        for _ in range(3):
            next(it)
        torch.cuda.synchronize()
        t0 = time.time()
        nb = 0
        for data in it:
            h = torch.relu(data["hop2"].float_attrs @ W)
            for _ in range(4):
                h = torch.relu((h @ W2) @ W)
            nb += 1
            if nb == 40:
                break
        torch.cuda.synchronize()
        print("training-loop stand-in, nn.pytorch.Dataset(device='cuda', prefetch=%s): %.2f ms/iteration"
              % (prefetch, (time.time() - t0) / nb * 1e3))
        ds.close()
    g.close()


if __name__ == "__main__":
    main()

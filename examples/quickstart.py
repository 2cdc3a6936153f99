"""Quick start: the reference's Python workflow on the MI355X engine.

    python examples/quickstart.py            (needs one GPU)

Writes a small user-item graph in graph-learn's TSV format, loads it with
gl.Graph().node().edge().init() (host parse, device build), then runs the sampling /
aggregation path three ways: the reference-style numpy API, the device-tensor API, and the
device-resident mini-batch loader.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))

import graphlearn as gl  # noqa: E402


def write_sources(directory, users=1000, items=200, dim=8):
    rng = np.random.default_rng(0)
    upath, ipath, epath = (os.path.join(directory, n) for n in ("user", "item", "buy"))
    with open(upath, "w") as f:
        f.write("id:int64\tweight:float\n")
        f.writelines("%d\t%f\n" % (u, 1.0 + u % 3) for u in range(users))
    with open(ipath, "w") as f:
        f.write("id:int64\tlabel:int64\tfeature:string\n")
        for i in range(items):
            f.write("%d\t%d\t%s\n" % (10000 + i, i % 5, ":".join("%.4f" % x for x in rng.standard_normal(dim))))
    with open(epath, "w") as f:
        f.write("src_id:int64\tdst_id:int64\tweight:float\n")
        for u in range(users):
            for i in rng.choice(items, size=int(rng.integers(1, 12)), replace=False):
                f.write("%d\t%d\t%f\n" % (u, 10000 + i, rng.random() + 0.01))
    return upath, ipath, epath, dim


def main():
    d = tempfile.mkdtemp(prefix="glx_quickstart_")
    upath, ipath, epath, dim = write_sources(d)
    gl.set_padding_mode(gl.CIRCULAR)
    gl.set_sampling_seed(2024)
    g = gl.Graph() \
        .node(upath, "user", gl.Decoder(weighted=True)) \
        .node(ipath, "item", gl.Decoder(labeled=True, attr_types=["float"] * dim)) \
        .edge(epath, ("user", "item", "buy"), gl.Decoder(weighted=True), directed=False) \
        .init()

    # 1. the reference's API: numpy in, Layers of Nodes / Edges out
    seeds = g.node_sampler("user", batch_size=8, strategy="random").get().ids
    layers = g.neighbor_sampler(["buy", "buy_reverse"], expand_factor=[5, 3], strategy="edge_weight").get(seeds)
    items, co_buyers = layers.layer_nodes(1), layers.layer_nodes(2)
    print("seeds", seeds.tolist())
    print("5 weighted item draws per user", items.ids.shape, "labels", items.labels[0].tolist())
    print("3 co-buyers per item", co_buyers.ids.shape, "weights", np.round(co_buyers.weights[0], 1).tolist())
    print("max-aggregated item features per user", items.embedding_agg("max").shape)
    negatives = g.negative_sampler("buy", 4, strategy="in_degree").get(seeds)
    print("4 popularity-weighted negatives per user", negatives.ids.shape)

    # 1b. the same walk as a GSL query: every step is one request of the sampler object it names
    query = g.V("user").batch(8).shuffle(traverse=True).alias("u") \
             .outV("buy").sample(5).by("edge_weight").alias("items") \
             .inV("buy").sample(3).by("random").alias("co_buyers") \
             .values(lambda r: (r["u"].ids, r["items"].labels, r["co_buyers"].ids))
    users, labels, co = gl.Dataset(query).next()
    print("GSL batch: users", users.shape, "item labels", labels.shape, "co-buyers", co.shape)

    # 2. device tensors in / out: all hops in one call, nothing leaves the GPU
    import torch
    hops = g.neighbor_sampler(["buy", "buy_reverse"], [5, 3], "edge_weight").get_device(torch.from_numpy(seeds).cuda())
    print("device hop shapes", [tuple(h[0].shape) for h in hops], hops[0][0].device)

    # 3. a training-loop style loader: shuffled seed batches -> hops -> features of every frontier
    loader = gl.NeighborLoader(g, "user", ["buy"], [5], batch_size=256, strategy="random")
    for batch in loader:
        pass
    print("one epoch:", len(loader), "batches; last batch item features", tuple(batch.x[1].shape))
    g.close()


if __name__ == "__main__":
    main()

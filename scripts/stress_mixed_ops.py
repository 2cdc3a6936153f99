"""Stress: many host threads hammer every operator of one store for a fixed time (numpy path through
pywrap with the GIL released, plus device-tensor calls from the main thread).  Looks for GPU faults,
crashes and wrong shapes, not for speed."""
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import graphlearn as gl  # noqa: E402
import pyapi_fixture as fx  # noqa: E402

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
d = tempfile.mkdtemp()
n1 = fx.write_nodes(d, "n1", (0, 5000), [fx.ATTRIBUTED])
n2 = fx.write_nodes(d, "n2", (5000, 10000), [fx.WEIGHTED, fx.LABELED])
e1 = fx.write_edges(d, "e1", (0, 5000), (5000, 10000), [fx.ATTRIBUTED, fx.WEIGHTED])
e2 = fx.write_edges(d, "e2", (5000, 10000), (0, 5000), [fx.WEIGHTED])
ent = fx.write_entity_nodes(d, "ent", 5000)
gl.set_padding_mode(gl.CIRCULAR)
g = gl.Graph().node(n1, "a", gl.Decoder(attr_types=fx.ATTR_TYPES)).node(n2, "b", gl.Decoder(weighted=True, labeled=True)) \
    .node(ent, "ent", gl.Decoder(attr_types=["float"] * 4, labeled=True)) \
    .edge(e1, ("a", "b", "ab"), gl.Decoder(attr_types=fx.ATTR_TYPES, weighted=True), directed=False) \
    .edge(e2, ("b", "a", "ba"), gl.Decoder(weighted=True)).init()
stop = time.time() + SECONDS
errors, counts = [], [0] * 16


def worker(tid):
    rng = np.random.default_rng(tid)
    strategies = ["random", "random_without_replacement", "edge_weight", "topk", "in_degree", "full"]
    try:
        while time.time() < stop:
            ids = rng.integers(-5, 5200, int(rng.integers(1, 600)))
            k = int(rng.integers(1, 40))
            s = strategies[int(rng.integers(0, len(strategies)))]
            layers = g.neighbor_sampler(["ab", "ba"], [k, 3], strategy=s).get(ids)
            nodes = layers.layer_nodes(2)
            if s != "full":
                assert nodes.ids.shape == (ids.size * k, 3)
                _ = nodes.int_attrs, layers.layer_edges(1).weights, layers.layer_nodes(1).labels
            g.negative_sampler("ab", 5, strategy=["random", "in_degree", "soft_in_degree"][tid % 3]).get(ids)
            g.negative_sampler("b", 4, strategy="node_weight").get(rng.integers(5000, 10000, 50))
            e = g.get_nodes("ent", rng.integers(0, 5100, (int(rng.integers(1, 300)), 6)))
            assert e.embedding_agg(["sum", "mean", "max", "min", "prod"][tid % 5]).shape[1] == 4
            g.out_degrees(ids, "ab"), g.in_degrees(ids + 5000, "ab")
            counts[tid] += 1
    except Exception as ex:  # noqa: BLE001
        errors.append(repr(ex))


threads = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
for t in threads:
    t.start()
loader = gl.NeighborLoader(g, "a", ["ab", "ba"], [7, 5], batch_size=512, strategy="edge_weight")
dev_batches = 0
while time.time() < stop:
    for batch in loader:
        dev_batches += 1
        w = g.random_walk("ab_reverse", batch.seeds + 5000, 4)
        if time.time() >= stop:
            break
torch.cuda.synchronize()
for t in threads:
    t.join()
g.close()
print("stress: %d host-thread rounds, %d device batches in %.0fs, errors: %s" % (sum(counts), dev_batches, SECONDS, errors[:3]))
sys.exit(1 if errors else 0)

"""Throughput of a GSL query driven by the REFERENCE's own Python layer on this engine (run through tests/refpy.py's
environment: `python scripts/r05/refpy_gsl_bench.py` re-executes itself with it).  A 2-hop EdgeWeight query with the
attribute lookups the Python layer attaches to every traversal node, over a synthetic weighted graph written in the
reference's TSV format; batches per second as seen by `Dataset.next()`, with the C++ DAG's hop fusion on and -- for the
A/B -- with the second hop made unfusable by a different strategy."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("GLX_REFPY_CHILD") != "1":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refpy
    import subprocess
    env = refpy.env()
    env["GLX_REFPY_CHILD"] = "1"
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

import numpy as np  # noqa: E402

import graphlearn as gl  # noqa: E402

V, DEG, D, B, K1, K2 = 200_000, 20, 32, 4096, 25, 10
work = "/tmp/glx_refpy_gsl_bench"
os.makedirs(work, exist_ok=True)
node_path, edge_path = os.path.join(work, "nodes"), os.path.join(work, "edges")
rng = np.random.default_rng(0)
if not os.path.exists(edge_path):
    with open(node_path, "w") as f:
        f.write("id:int64\tfeature:string\n")
        x = rng.random((V, D)).astype(np.float32)
        for v in range(V):
            f.write("%d\t%s\n" % (v, ":".join("%.4f" % t for t in x[v])))
    with open(edge_path, "w") as f:
        f.write("src_id:int64\tdst_id:int64\tweight:float\n")
        dst = rng.integers(0, V, (V, DEG))
        w = rng.random((V, DEG)) + 0.01
        for v in range(V):
            for j in range(DEG):
                f.write("%d\t%d\t%.4f\n" % (v, dst[v, j], w[v, j]))
gl.set_padding_mode(gl.CIRCULAR)
t0 = time.time()
g = gl.Graph().node(node_path, node_type="i", decoder=gl.Decoder(attr_types=["float"] * D)) \
              .edge(edge_path, edge_type=("i", "i", "e"), decoder=gl.Decoder(weighted=True))
g.init()
print("loaded %d nodes x %d floats + %d edges (TSV -> HBM) in %.1f s" % (V, D, V * DEG, time.time() - t0))


def run(second_strategy, n=60):
    q = g.V("i").batch(B).alias("a").outV("e").sample(K1).by("edge_weight").alias("b") \
         .outV("e").sample(K2).by(second_strategy).alias("c").values()
    ds = gl.Dataset(q, 10)
    for _ in range(5):
        ds.next()
    t = time.time()
    done = 0
    while done < n:
        try:
            res = ds.next()
            done += 1
        except gl.OutOfRangeError:
            continue
    dt = time.time() - t
    assert res["c"].ids.shape == (B * K1, K2) and res["c"].float_attrs.shape == (B * K1, K2, D)
    return dt / n * 1e3


fused = run("edge_weight")
apart = run("topk")
edges = B * K1 + B * K1 * K2
print("query V.batch(%d).outV.sample(%d).by(edge_weight).outV.sample(%d): %.1f ms per batch as one fused glx_sample_hops step "
      "(%.1f M sampled edges/s incl. the float attributes of all %d vertices per batch through host tensors); %.1f ms with the "
      "second hop a topk node (two operator calls)" % (B, K1, K2, fused, edges / fused / 1e3, B + edges, apart))
g.close()

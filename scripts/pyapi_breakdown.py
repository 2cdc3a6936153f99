"""Where does a numpy-path step (NeighborSampler.get + embedding_agg) spend its time?"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch  # noqa: E402,F401
import graphlearn as gl  # noqa: E402
from graphlearn import pywrap_graphlearn as pywrap  # noqa: E402
import tempfile  # noqa: E402
import pandas as pd  # noqa: E402
import synth  # noqa: E402

V, E, B, D = 1_000_000, 10_000_000, 8192, 64
dev = torch.device("cuda", 0)
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
d = tempfile.mkdtemp()
ep, npth = os.path.join(d, "e"), os.path.join(d, "n")
pd.DataFrame({"src_id:int64": src.cpu().numpy(), "dst_id:int64": dst.cpu().numpy(),
              "weight:float": w.cpu().numpy()}).to_csv(ep, sep="\t", index=False, float_format="%.6f")
with open(npth, "w") as f:
    f.write("id:int64\tfeature:string\n")
    row = ":".join(["0.5"] * D)
    f.write("".join("%d\t%s\n" % (i, row) for i in range(V)))
gl.set_padding_mode(gl.CIRCULAR)
g = gl.Graph().node(npth, "v", gl.Decoder(attr_types=["float"] * D)).edge(ep, ("v", "v", "e"), gl.Decoder(weighted=True)).init()
client = g.get_client()
rng = np.random.default_rng(0)


def T(label, fn, n=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    dt = (time.perf_counter() - t0) / n
    print("%-46s %8.2f ms" % (label, dt * 1e3))
    return out


ids2 = rng.integers(0, V, B * 25).astype(np.int64)


def raw_request():
    req = pywrap.new_sampling_request("e", "EdgeWeightSampler", 10, pywrap.FilterType.OPERATOR_UNSPECIFIED,
                                      pywrap.FilterField.FIELD_UNSPECIFIED)
    pywrap.set_sampling_request(req, ids2)
    res = pywrap.new_sampling_response()
    client.sample_neighbor(req, res)
    return req, res


def raw_and_free():
    req, res = raw_request()
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)


T("hop-2 request: build + Process (2M slots)", raw_and_free)
req, res = raw_request()
nbr = T("get_sampling_node_ids (copy out 16 MB)", lambda: pywrap.get_sampling_node_ids(res))
T("np.repeat(src, 10)", lambda: np.repeat(ids2, 10))
T("get_nodes + get_edges objects", lambda: (g.get_nodes("v", nbr, shape=(B * 25, 10)),
                                            g.get_edges("e", np.repeat(ids2, 10), nbr, shape=(B * 25, 10))))
nodes = g.get_nodes("v", nbr, shape=(B * 25, 10))
T("embedding_agg('max') total", lambda: nodes.embedding_agg("max"))
seg = np.repeat(np.arange(B * 25, dtype=np.int32), 10)
T("  np.repeat segment ids", lambda: np.repeat(np.arange(B * 25, dtype=np.int32), 10))


def agg_raw():
    rq = pywrap.new_aggregating_request("v", "MaxAggregator")
    pywrap.set_aggregating_request(rq, nbr, seg, B * 25)
    rs = pywrap.new_aggregating_response()
    client.agg_nodes(rq, rs)
    return rq, rs


def agg_and_free():
    rq, rs = agg_raw()
    pywrap.del_op_response(rs)
    pywrap.del_op_request(rq)


T("  aggregating request: build + Process", agg_and_free)
rq, rs = agg_raw()
T("  get_aggregating_nodes (copy out 52 MB)", lambda: pywrap.get_aggregating_nodes(rs))
s = g.neighbor_sampler(["e", "e"], [25, 10], "edge_weight")
seeds = rng.integers(0, V, B).astype(np.int64)
T("NeighborSampler.get (2 hops)", lambda: s.get(seeds))
g.close()

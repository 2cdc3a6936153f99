"""Run with the STAGED reference Python layer on the path (tests/refpy.py env()).  A GSL query whose two sampling hops
form a fusable chain -- the host DAG runner lowers them to ONE glx_sample_hops call (host dag.h) -- must return what the
same two hops return as separate SamplingRequests through the reference's own NeighborSampler (sampler/
neighbor_sampler.py:93-127), value for value:
  topk    deterministic: compared in this process;
  random  the k-th call of an operator draws from stream (seed, k): a fresh process whose FIRST two RandomSampler calls
          are the query's fused hops must return what a fresh process returns whose first two calls are the direct hops.
          mode "dag" / "direct" print the values; the test compares the two outputs.
usage: refpy_dag_fusion.py topk | dag | direct"""
import sys

import numpy as np

import graphlearn as gl
import graphlearn.python.tests.utils as utils

mode = sys.argv[1]
gl.set_default_neighbor_id(-1)
gl.set_padding_mode(gl.CIRCULAR)
gl.set_tape_capacity(1)
gl.set_dataset_capacity(1)
if hasattr(gl.pywrap, "set_sampling_seed"):
    gl.pywrap.set_sampling_seed(1234)
utils.prepare_env()
n1 = utils.gen_node_data("node1", (0, 100), [utils.ATTRIBUTED])
n2 = utils.gen_node_data("node2", (100, 200), [utils.WEIGHTED, utils.LABELED])
e1 = utils.gen_edge_data("node1", "node2", (0, 100), (100, 200), schema=[utils.WEIGHTED])
e2 = utils.gen_edge_data("node2", "node1", (100, 200), (0, 100), schema=[utils.WEIGHTED])
g = gl.Graph() \
    .node(n1, node_type="node1", decoder=gl.Decoder(attr_types=utils.ATTR_TYPES)) \
    .node(n2, node_type="node2", decoder=gl.Decoder(weighted=True, labeled=True)) \
    .edge(e1, edge_type=("node1", "node2", "e1"), decoder=gl.Decoder(weighted=True)) \
    .edge(e2, edge_type=("node2", "node1", "e2"), decoder=gl.Decoder(weighted=True))
g.init()
strategy = "topk" if mode == "topk" else "random"
B, K1, K2 = 8, 3, 2


def via_dag():
    q = g.V("node1").batch(B).alias("a") \
         .outV("e1").sample(K1).by(strategy).alias("b") \
         .outV("e2").sample(K2).by(strategy).alias("c").values()
    ds = gl.Dataset(q, 1)
    res = ds.next()
    return res["a"].ids.copy(), res["b"].ids.copy(), res["c"].ids.copy()


def direct(seeds):
    layers = g.neighbor_sampler(["e1", "e2"], expand_factor=[K1, K2], strategy=strategy).get(seeds)
    return layers.layer_nodes(1).ids.copy(), layers.layer_nodes(2).ids.copy()


if mode == "topk":
    a, b, c = via_dag()
    db, dc = direct(a)
    assert a.shape == (B,) and b.shape == (B, K1) and c.shape == (B * K1, K2), (a.shape, b.shape, c.shape)
    assert np.array_equal(b, db) and np.array_equal(c.reshape(dc.shape), dc), (b, db)
    assert (b >= 100).all() or (b == -1).any()
    print("TOPK_OK")
elif mode == "dag":
    a, b, c = via_dag()
    print("VALUES", a.tolist(), b.reshape(-1).tolist(), c.reshape(-1).tolist())
else:
    a = np.arange(0, B, dtype=np.int64)  # what the by_order root hands out first
    b, c = direct(a)
    print("VALUES", a.tolist(), b.reshape(-1).tolist(), c.reshape(-1).tolist())
g.close()

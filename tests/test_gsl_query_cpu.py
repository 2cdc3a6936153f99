"""CPU tests of the GSL query layer (graph-learn_amd/python/graphlearn/gsl.py; the reference's
graphlearn/python/gsl/dag_node.py surface): chain construction, validation, evaluation order, branches, epochs and
drop_last -- against a stand-in graph that answers every sampler request with recognisable ids, so no device is
involved.  What the steps draw on a real graph is tests/test_gpu_pyapi_gsl.py's subject."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))

import graphlearn as gl  # noqa: E402
from graphlearn import gsl  # noqa: E402


class _Nodes(object):
    def __init__(self, ids, t, shape=None):
        self.ids = np.asarray(ids, dtype=np.int64).reshape(shape if shape is not None else (-1,))
        self.type = t
        self.shape = self.ids.shape


class _Edges(object):
    def __init__(self, src, dst, shape, edge_type="e"):
        self.src_ids = np.asarray(src, dtype=np.int64).reshape(shape)
        self.dst_ids = np.asarray(dst, dtype=np.int64).reshape(shape)
        self.shape = self.src_ids.shape
        self.edge_type = edge_type


class _Topology(object):
    def __init__(self, edges):
        self._e = edges

    def get_src_type(self, e):
        return self._e[e][0]

    def get_dst_type(self, e):
        return self._e[e][1]


class FakeGraph(object):
    """user 0..9 --buy--> item (100 + ...), item --sim--> item; buy is undirected (buy_reverse exists)."""

    def __init__(self):
        self.edges = {"buy": ("user", "item"), "buy_reverse": ("item", "user"), "sim": ("item", "item")}
        self.nodes = {"user": 10, "item": 10}
        self.requests = []
        self._cursor = {}

    def get_topology(self):
        return _Topology(self.edges)

    def get_edge_decoders(self):
        return self.edges

    def get_node_decoders(self):
        return self.nodes

    def get_nodes(self, t, ids, offsets=None, shape=None):
        return _Nodes(ids, t, shape)

    def node_sampler(self, t, batch_size=64, strategy="by_order", node_from=None, mask=None):
        if t not in self.nodes and t not in self.edges:
            raise ValueError("Graph has no node type of {}".format(t))
        graph, key = self, ("V", t)

        class S(object):
            _node_type = t if t in graph.nodes else graph.edges[t][0]

            def get(self):
                at = graph._cursor.get(key, 0)
                if at >= 10:
                    graph._cursor[key] = 0
                    raise gl.OutOfRangeError("epoch end")
                graph._cursor[key] = min(10, at + batch_size)
                graph.requests.append(("nodes", t, strategy))
                return _Nodes(np.arange(at, min(10, at + batch_size)), self._node_type)
        return S()

    def edge_sampler(self, edge_type, batch_size=64, strategy="by_order", mask=None):
        if edge_type not in self.edges:
            raise ValueError("Graph has no edge type of {}".format(edge_type))
        graph, key = self, ("E", edge_type)

        class S(object):
            _stored = edge_type

            def get(self):
                at = graph._cursor.get(key, 0)
                if at >= 7:
                    graph._cursor[key] = 0
                    raise gl.OutOfRangeError("epoch end")
                hi = min(7, at + batch_size)
                graph._cursor[key] = hi
                graph.requests.append(("edges", edge_type, strategy))
                return _Edges(np.arange(at, hi), 100 + np.arange(at, hi), (hi - at,), edge_type)
        return S()

    def neighbor_sampler(self, meta_path, expand_factor, strategy="random"):
        graph = self

        class S(object):
            flt = None

            def set_filter(self, a, b):
                self.flt = (a, b)
                return self

            def get(self, ids, filter_values=None):
                ids = np.asarray(ids).reshape(-1)
                graph.requests.append(("neighbors", meta_path, expand_factor, strategy, self.flt,
                                       None if filter_values is None else filter_values.tolist()))
                nbr = (ids[:, None] * 10 + np.arange(expand_factor)[None, :]) % 1000
                dst_t = graph.edges[meta_path][1]

                class L(object):
                    def layer_nodes(self, i):
                        return _Nodes(nbr, dst_t, (ids.size, expand_factor))

                    def layer_edges(self, i):
                        return _Edges(np.repeat(ids, expand_factor), nbr, (ids.size, expand_factor), meta_path)
                return L()
        return S()

    def negative_sampler(self, object_type, expand_factor, strategy="random", conditional=False, **kw):
        graph = self

        class S(object):
            def get(self, src, dst=None):
                src = np.asarray(src).reshape(-1)
                graph.requests.append(("negatives", object_type, expand_factor, strategy, conditional, sorted(kw),
                                       None if dst is None else np.asarray(dst).tolist()))
                return _Nodes(np.full((src.size, expand_factor), 777), "item", (src.size, expand_factor))
        return S()

    def random_walk(self, edge_type, ids, walk_len, p=1.0, q=1.0):
        self.requests.append(("walk", edge_type, walk_len, p, q))
        return np.tile(np.asarray(ids).reshape(-1, 1), (1, walk_len))


def test_chain_shapes_order_and_epochs():
    g = FakeGraph()
    q = gsl.VertexSource(gsl.Query(g), "user").batch(4).alias("a") \
        .outV("buy").sample(3).by("topk").alias("b") \
        .outV("sim").sample(2).by("edge_weight").alias("c") \
        .values()
    ds = gl.Dataset(q)
    sizes = []
    while True:
        try:
            res = ds.next()
        except gl.OutOfRangeError:
            break
        n = res["a"].shape[0]
        assert res["b"].shape == (n, 3) and res["c"].shape == (3 * n, 2)
        assert (res["a"].type, res["b"].type, res["c"].type) == ("user", "item", "item")
        np.testing.assert_equal(res["c"].ids[:, 0], (res["b"].ids.reshape(-1) * 10) % 1000)  # hop 2 fed by hop 1
        sizes.append(n)
    assert sizes == [4, 4, 2]  # the short tail is a batch; then the epoch ends
    kinds = [r[0] for r in g.requests]
    assert kinds == ["nodes", "neighbors", "neighbors"] * 3  # one request per step per batch, upstream first
    assert g.requests[1][1:4] == ("buy", 3, "topk") and g.requests[2][1:4] == ("sim", 2, "edge_weight")
    assert ds.next()["a"].ids.tolist() == [0, 1, 2, 3]  # the next epoch starts by itself


def test_drop_last_and_shuffle_strategies():
    g = FakeGraph()
    q = gsl.VertexSource(gsl.Query(g), "user").batch(4).shuffle(traverse=True).alias("a").values()
    ds = gl.Dataset(q, drop_last=True)
    assert ds.next()["a"].shape == (4,) and ds.next()["a"].shape == (4,)
    with pytest.raises(gl.OutOfRangeError):
        ds.next()  # the 2-vertex tail is skipped
    assert {r[2] for r in g.requests} == {"shuffle"}
    g2 = FakeGraph()
    gl.Dataset(gsl.VertexSource(gsl.Query(g2), "user").shuffle().alias("a").values()).next()
    assert g2.requests[0][2] == "random"


def test_edge_source_endpoints_each_filter_and_where():
    g = FakeGraph()
    q = gsl.EdgeSource(gsl.Query(g), "buy").batch(3).alias("e") \
        .each(lambda e: (
            e.inV().alias("dst"),
            e.outV().alias("src").outV("buy").sample(2).by("random").filter("dst").alias("nbr"),
            e.outV().outNeg("buy").sample(4).by("in_degree").where("dst", condition={"int_cols": [0], "int_props": [0.5]})
             .alias("neg"))) \
        .values(lambda r: (r["e"].src_ids, r["src"].ids, r["dst"].ids, r["nbr"].shape, r["neg"].shape))
    e_src, src, dst, nbr_shape, neg_shape = gl.Dataset(q).next()
    np.testing.assert_equal(src, e_src)
    np.testing.assert_equal(dst, 100 + e_src)
    assert nbr_shape == (3, 2) and neg_shape == (3, 4)
    nb = [r for r in g.requests if r[0] == "neighbors"][0]
    assert nb[4] == ("equal", "id") and nb[5] == dst.tolist()  # the filter values are the target step's ids, row for row
    ng = [r for r in g.requests if r[0] == "negatives"][0]
    assert ng[1:5] == ("buy", 4, "in_degree", True) and ng[5] == ["int_cols", "int_props"] and ng[6] == dst.tolist()


def test_in_traversals_walk_the_reversed_twin_and_edges_steps():
    g = FakeGraph()
    q = gsl.VertexSource(gsl.Query(g), "item").batch(2).alias("i") \
        .inV("buy").sample(3).by("random").alias("buyers").values()
    gl.Dataset(q).next()
    assert g.requests[1][1] == "buy_reverse"
    q = gsl.VertexSource(gsl.Query(g), "user").batch(2).alias("u").outE("buy").sample(2).by("random").alias("e") \
        .inV().alias("items").values()
    res = gl.Dataset(q).next()
    assert res["e"].shape == (2, 2) and res["items"].shape == (2, 2) and res["items"].type == "item"
    np.testing.assert_equal(res["items"].ids, res["e"].dst_ids)
    src = gsl.VertexSource(gsl.Query(g), "item").batch(2).alias("s")
    src.random_walk("sim", 5, 0.5, 2.0).alias("w")
    res = gl.Dataset(src.values()).next()
    assert res["w"].shape == (2, 5) and ("walk", "sim", 5, 0.5, 2.0) in g.requests


def test_validation():
    g = FakeGraph()
    V = lambda t: gsl.VertexSource(gsl.Query(g), t)  # noqa: E731
    with pytest.raises(ValueError):
        V("nothing")
    with pytest.raises(ValueError):
        V("user").outV("sim")  # sim starts at item
    with pytest.raises(ValueError):
        V("item").inV("sim")  # no sim_reverse: a directed type has no in-traversal
    with pytest.raises(ValueError):
        V("user").alias("x").outV("buy").alias("x")
    with pytest.raises(ValueError):
        V("user").alias("")
    with pytest.raises(ValueError):
        V("user").outV("buy").by("best")
    with pytest.raises(ValueError):
        V("user").outV("buy").sample(-1)
    with pytest.raises(ValueError):
        V("user").outV("buy").batch(2)  # batch() belongs to the source
    with pytest.raises(ValueError):
        V("user").sample(2)
    with pytest.raises(ValueError):
        V("user").outV("buy").filter("nobody")
    with pytest.raises(ValueError):
        V("user").outNeg("buy").where("x")  # unknown alias
    with pytest.raises(ValueError):
        V("user").alias("a").outNeg("buy").where("a", condition={"colour": 1})
    with pytest.raises(ValueError):
        V("user").batch(0)
    with pytest.raises(ValueError):
        gl.Dataset(V("user").alias("a"))  # not closed
    q = V("user").alias("a").values()
    with pytest.raises(ValueError):
        q.steps[0].values()  # closed twice
    with pytest.raises(ValueError):
        q.steps[0].outV("buy")  # closed: no more steps
    with pytest.raises(ValueError):
        gl.Dataset(V("user").batch(2).alias("a").outV("buy").alias("b").values()).next()  # sample() missing
    # a filter target of another size is reported when the batch runs
    bad = V("user").batch(2).alias("a").outV("buy").sample(3).by("random").alias("b")
    bad.outV("sim").sample(2).by("random").filter("a").alias("c")
    with pytest.raises(ValueError):
        gl.Dataset(bad.values()).next()


def test_prefetch_delivers_the_same_stream_ahead_of_time():
    """Dataset(prefetch=True): a background thread fills a queue of `window` batches (the reference's tapes,
    dag_dataset.py / core/dag/tape.h); what next() hands out -- batches, the OutOfRangeError that ends each epoch, the
    epoch after it -- is the stream a plain Dataset produces, and close() / garbage collection stop the thread."""
    import gc
    import threading
    import time

    def stream(ds, n):
        out = []
        for _ in range(n):
            try:
                out.append(ds.next()["a"].ids.tolist())
            except gl.OutOfRangeError:
                out.append("end")
        return out

    def query(g):
        return gsl.VertexSource(gsl.Query(g), "user").batch(4).alias("a") \
                  .outV("buy").sample(3).by("topk").alias("b").values()
    plain = stream(gl.Dataset(query(FakeGraph())), 9)
    assert plain == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9], "end"] * 2 + [[0, 1, 2, 3]]
    g = FakeGraph()
    ds = gl.Dataset(query(g), window=2, prefetch=True)
    assert ds.next()["a"].ids.tolist() == [0, 1, 2, 3]
    deadline = time.time() + 5
    while len([r for r in g.requests if r[0] == "nodes"]) < 3 and time.time() < deadline:
        time.sleep(0.01)  # the worker runs ahead: batch 1 handed out, batches 2 and 3 already produced ...
    time.sleep(0.05)
    assert len([r for r in g.requests if r[0] == "nodes"]) in (3, 4)  # ... but no further than the window (+ one in hand)
    assert [[0, 1, 2, 3]] + stream(ds, 8) == plain
    ds.close()
    assert not any(t.name == "gsl-prefetch" and t.is_alive() for t in threading.enumerate())
    # a closed dataset starts a fresh worker on demand; where the source's cursor stands depends on how far the stopped
    # worker had run ahead, so the first thing it delivers may be the end of that epoch
    assert stream(ds, 2)[-1] != "end" or stream(ds, 1)[0] != "end"
    ds.close()

    # an error inside a step reaches the caller and ends production
    class Broken(FakeGraph):
        def neighbor_sampler(self, *a, **k):
            raise RuntimeError("boom")
    bad = gl.Dataset(query(Broken()), window=2, prefetch=True)
    with pytest.raises(RuntimeError):
        bad.next()
    bad.close()

    # dropping the last reference stops the worker too
    ds2 = gl.Dataset(query(FakeGraph()), window=1, prefetch=True)
    ds2.next()
    del ds2
    gc.collect()
    deadline = time.time() + 5
    while any(t.name == "gsl-prefetch" and t.is_alive() for t in threading.enumerate()) and time.time() < deadline:
        time.sleep(0.02)
    assert not any(t.name == "gsl-prefetch" and t.is_alive() for t in threading.enumerate())

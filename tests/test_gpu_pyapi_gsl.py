"""The reference's GSL unit tests -- graphlearn/python/gsl/tests/test_gsl_{sampling,traverse,mask,random_walk}.py --
restated against the glx engine through `graphlearn.gsl` (g.V() / g.E() ... .values() + gl.Dataset).  Same fixture as
test_gpu_pyapi.py (the reference's test_sampling.py graph: node1 0..99 attributed, node2 100..199 weighted + labeled,
edge1 node1->node2, edge2 node2->node1, edge3 node2->node2, generator dst = src*i % 100 + lo).  Every step is one
operator request through the same client the sampler objects use, so what is under test is the query layer: shapes,
types, epochs, filters, conditions, branches."""
import numpy as np
import pytest

import pyapi_fixture as fx
from test_gpu_pyapi import (EDGE1, EDGE2, EDGE3, NODE1, NODE2, RANGE1, RANGE2, g, gl)  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


def _drain(gl, dataset, check, limit=10000):
    n = 0
    while True:
        try:
            res = dataset.next()
        except gl.OutOfRangeError:
            return n
        check(res)
        n += 1
        assert n < limit


def test_iterate_node_with_2hop(gl, g):
    q = g.V(NODE1).batch(2).alias('a') \
         .outV(EDGE1).sample(3).by('random').alias('b') \
         .outV(EDGE2).sample(4).by('random').alias('c') \
         .values()
    seen = []

    def check(res):
        assert res['a'].shape == (2,) and res['b'].shape == (2, 3) and res['c'].shape == (2 * 3, 4)
        assert (res['a'].type, res['b'].type, res['c'].type) == (NODE1, NODE2, NODE1)
        for s, row in zip(res['a'].ids, res['b'].ids):  # hop 1 follows the generator (or is the default id)
            assert set(row.tolist()) <= set(fx.fixed_dst_ids(int(s), RANGE2)) | {-1}
        for s, row in zip(res['b'].ids.reshape(-1), res['c'].ids):
            assert set(row.tolist()) <= (set(fx.fixed_dst_ids(int(s), RANGE1)) if s >= 0 else set()) | {-1}
        seen.extend(res['a'].ids.tolist())
    ds = gl.Dataset(q, 10)
    assert _drain(gl, ds, check) == 50
    assert sorted(seen) == list(range(*RANGE1))  # one epoch = every node1 once
    assert _drain(gl, ds, lambda r: None) == 50  # the next epoch starts by itself


def test_iterate_edge_with_1hop(gl, g):
    q = g.E(EDGE1).batch(4).alias("a") \
         .outV().alias("b") \
         .outV(EDGE1).sample(2).by("random").alias("c") \
         .values()

    def check(res):
        n = res['a'].shape[0]
        assert n <= 4 and res['b'].shape == (n,) and res['c'].shape == (n, 2)
        assert res['b'].int_attrs.shape == (n, 2)  # [batch_size, int_attr_num]
        np.testing.assert_equal(res['b'].ids, res['a'].src_ids)
        np.testing.assert_equal(res['b'].int_attrs[:, 0], res['b'].ids)
    total_edges = len(fx.fixed_dst_ids(range(*RANGE1), RANGE2))
    assert _drain(gl, gl.Dataset(q), check) == (total_edges + 3) // 4


def test_sample_edge(gl, g):
    q = g.V(NODE1).batch(8).alias('a') \
         .outE(EDGE1).sample(3).by("random").alias('b') \
         .inV().alias('c') \
         .values()
    res = gl.Dataset(q).next()
    assert res['a'].shape == (8,) and res['b'].shape == (8, 3) and res['c'].shape == (8, 3)
    np.testing.assert_equal(res['c'].ids, res['b'].dst_ids)
    np.testing.assert_equal(res['b'].src_ids, np.repeat(res['a'].ids, 3).reshape(8, 3))
    assert res['b'].edge_type == EDGE1 and res['c'].type == NODE2


def test_negative_sample(gl, g):
    q = g.V(NODE1).batch(2).alias('a') \
         .outNeg(EDGE1).sample(5).by("random").alias('b') \
         .values(lambda x: (x['a'].ids, x['b'].weights, x['b'].ids))
    ids, weights, negs = gl.Dataset(q).next()
    assert ids.shape == (2,) and weights.shape == (2, 5)
    assert set(negs.reshape(-1).tolist()) <= set(fx.fixed_dst_ids(range(*RANGE1), RANGE2))  # candidates = edge1's dst ids
    # (a "random" negative may still be a neighbour after SamplingRetryTimes redraws: random_negative_sampler.cc keeps
    # the last draw; the reference's test checks shapes only)
    np.testing.assert_almost_equal(weights, negs / 10.0, decimal=4)  # node2 weights = id / 10


def test_conditional_negative_sample(gl, g):
    q = g.E("cond_sim").batch(4).alias("e") \
         .each(lambda e: (
             e.inV().alias('dst'),
             e.outV().alias('src')
              .outNeg("cond_sim").sample(4).by('random').where(
                  "dst", condition={"int_cols": [0, 1], "int_props": [0.25, 0.25], "str_cols": [0], "str_props": [0.5]})
              .alias('neg'))) \
         .values()
    res = gl.Dataset(q).next()
    src_ids, dst_ids, neg_ids = res["src"].ids, res["dst"].ids, res["neg"].ids
    assert neg_ids.shape == (4, 4)
    for idx, sid in enumerate(src_ids):
        assert set(neg_ids[idx].tolist()).isdisjoint({sid + 2, sid + 3, sid + 5})
        pos, neg = dst_ids[idx], neg_ids[idx]
        assert neg[0] % 5 == pos % 5 and neg[1] % 4 == pos % 4 and neg[2] % 3 == pos % 3 and neg[3] % 3 == pos % 3


def test_sample_with_filter(gl, g):
    gl.set_sampler_retry_times(500)  # as the reference's test file does: a hit is redrawn until it misses
    q = g.E(EDGE1).batch(4).alias("a") \
         .each(lambda e: (
             e.inV().alias('dst'),
             e.outV().alias('src').outV(EDGE1).sample(2).by("random").filter('dst').alias("b"))) \
         .values()

    def check(res):
        n = res['a'].shape[0]
        assert res['b'].shape == (n, 2)
        for fid, rid in zip(res['dst'].ids, res['b'].ids):
            assert fid not in rid
    try:
        assert _drain(gl, gl.Dataset(q), check) > 0
    finally:
        gl.set_sampler_retry_times(5)  # the engine's default


def test_full_sample(gl, g):
    q = g.V(NODE2).batch(4).alias('a') \
         .outV(EDGE2).sample(3).by("full").alias('b') \
         .values(lambda x: (x['a'].ids, x['b'].ids, x['b'].offsets))

    def check(res):
        src, nbrs, offsets = res
        start = 0
        for idx, offset in enumerate(offsets):
            expected = fx.fixed_dst_ids(int(src[idx]), RANGE1)
            assert offset == min(len(expected), 3)
            assert set(nbrs[start: start + offset].tolist()) <= set(expected)
            start += offset
    assert _drain(gl, gl.Dataset(q), check) == 25


@pytest.mark.parametrize("drop_last", [False, True])
def test_iterate_edge_with_each(gl, g, drop_last):
    q = g.E(EDGE1).batch(7).alias('a') \
         .each(lambda x: (
             x.outV().alias('b').outV(EDGE1).sample(2).by('random').alias('d'),
             x.inV().alias('c').outV(EDGE2).sample(2).by('random').alias('e'))) \
         .values(lambda x: (x['a'].int_attrs, x['d'].weights, x['e'].ids))
    sizes = []
    _drain(gl, gl.Dataset(q, drop_last=drop_last), lambda res: sizes.append(res[0].size))
    total_edges = len(fx.fixed_dst_ids(range(*RANGE1), RANGE2))
    if drop_last:
        assert sizes == [14] * (total_edges // 7)  # 7 edges x 2 int attributes, short tail dropped
    else:
        assert sum(sizes) == 2 * total_edges and all(s <= 14 for s in sizes)


def test_traverse_with_mask(gl, g):
    q = g.V(NODE1, mask=gl.Mask.TRAIN).batch(8).alias('train').values()
    ds = gl.Dataset(q)
    for _ in range(2):
        got = []
        _drain(gl, ds, lambda res: got.extend(res['train'].ids.tolist()))
        assert sorted(got) == list(range(0, 50))  # the masked source of the fixture: node1 0..49 as TRAIN
    res = gl.Dataset(g.V(NODE1, mask=gl.Mask.TRAIN).batch(8).alias('t').values()).next()
    assert res['t'].type == NODE1  # the mask picks the source; the vertices are node1 vertices with node1's attributes
    np.testing.assert_equal(res['t'].int_attrs[:, 0], res['t'].ids)


def test_random_walk(gl, g):
    src = g.V(NODE2).batch(4).alias('src')
    src.random_walk(EDGE3, 10, 1.0, 1.0).alias('walks')
    ds = gl.Dataset(src.values())

    def check(res):
        n = res['src'].shape[0]
        walks = res['walks'].ids
        assert walks.shape == (n, 10) and res['walks'].type == NODE2
        for s, row in zip(res['src'].ids, walks):  # every step follows an edge3 edge (undirected here) or stops
            cur = int(s)
            for nxt in row.tolist():
                if nxt == -1:
                    break
                assert nxt in range(*RANGE2)
                cur = nxt
    assert _drain(gl, ds, check) == 25


def test_shuffle_and_node_from(gl, g):
    q = g.V(EDGE1, node_from=gl.EDGE_SRC).batch(16).shuffle(traverse=True).alias('s').values()
    got = []
    _drain(gl, gl.Dataset(q), lambda res: got.extend(res['s'].ids.tolist()))
    assert sorted(got) == sorted(set(s for s in range(*RANGE1) if s % 5))  # the sources of edge1, each once
    assert got != sorted(got)  # ... in a shuffled order
    res = gl.Dataset(g.V(NODE1).batch(5).shuffle().alias('r').values()).next()  # independent draws, no epochs
    assert res['r'].shape == (5,)
    # in-neighbours walk the reversed twin of an undirected type
    res = gl.Dataset(g.V(NODE2).batch(6).alias('a').inV(EDGE1).sample(2).by("random").alias('b').values()).next()
    assert res['b'].shape == (6, 2) and res['b'].type == NODE1


def test_subgraph_step(gl, g):
    """python/sampler/tests/test_subgraph_sampling.py through a query: batches of 8 entity nodes in order, the
    sub-graph the relation edges induce among them (the same expectations as the sampler-object test)."""
    q = g.V("entity").batch(8).alias('seed').SubGraph("relation").alias('sub').values()
    batches = []

    def check(res):
        sub, ids = res['sub'], res['sub'].nodes.ids
        np.testing.assert_equal(np.sort(ids), np.sort(res['seed'].ids))
        np.testing.assert_equal(sub.nodes.labels, ids)
        rows, cols = [], []
        for i in range(ids.size):
            for j in range(ids.size):
                if (ids[i] < 100 or ids[j] < 100) and abs(int(ids[i]) - int(ids[j])) in (2, 3, 5):
                    rows += [i, j]
                    cols += [j, i]
        np.testing.assert_equal(sub.edge_index[0], np.array(rows, dtype=np.int32))
        np.testing.assert_equal(sub.edge_index[1], np.array(cols, dtype=np.int32))
        batches.append(ids.size)
    assert _drain(gl, gl.Dataset(q), check) == 15
    # around the (src, dst) pairs of an edge batch
    res = gl.Dataset(g.E("relation").batch(1).alias('e').SubGraph("relation", num_nbrs=[2], need_dist=True).alias('s')
                     .values()).next()
    sub = res['s']
    assert {int(res['e'].src_ids[0]), int(res['e'].dst_ids[0])} <= set(sub.nodes.ids.tolist())
    assert sub.dist_to_src is not None and len(sub.dist_to_src) == sub.nodes.ids.size


def test_edge_and_node_iteration_reference_cases(gl, g):
    """python/tests/test_{node,edge}_{iterate,shuffle}_gsl.py restated on the shared fixture: by_order and
    shuffle(traverse=True) visit every element exactly once per epoch, shuffle() draws from the whole set for ever."""
    all_src = sorted(s for s in range(*RANGE2) for _ in fx.fixed_dst_ids(s, RANGE1))
    for shuffled in (False, True):
        src = g.E(EDGE2).batch(4)
        q = (src.shuffle(traverse=True) if shuffled else src).alias('seed').values()
        got_src, order = [], []

        def check(res):
            e = res['seed']
            np.testing.assert_almost_equal(e.weights, (e.src_ids + 0.1 * e.dst_ids) / 10.0, decimal=4)
            got_src.extend(e.src_ids.tolist())
            order.extend(e.edge_ids.tolist())
        _drain(gl, gl.Dataset(q, window=1), check)
        assert sorted(got_src) == all_src and sorted(order) == list(range(len(all_src)))
        assert (order != sorted(order)) == shuffled
    ds = gl.Dataset(g.V(NODE2).batch(4).shuffle().alias('n').values())
    for _ in range(60):  # more batches than an epoch holds: never out of range
        nodes = ds.next()['n']
        assert set(nodes.ids.tolist()) <= set(range(*RANGE2))
        np.testing.assert_equal(nodes.labels, nodes.ids)


def test_fused_hops_one_engine_call_per_chain(gl, g):
    """Dataset(fuse_hops=True): the two random hops of a chain run as one glx_sample_hops call (frontier in HBM);
    shapes, types, generator membership and epochs as hop by hop; a branch or a differing strategy ends the chain."""
    q = g.V(NODE1).batch(5).alias('a') \
         .outV(EDGE1).sample(3).by('random').alias('b') \
         .outV(EDGE2).sample(4).by('random').alias('c') \
         .outV(EDGE1).sample(2).by('topk').alias('d') \
         .values()
    ds = gl.Dataset(q, fuse_hops=True)
    assert [len(c) for c in ds._chains.values()] == [2]  # b + c fuse; d has another strategy

    def check(res):
        n = res['a'].shape[0]
        assert res['b'].shape == (n, 3) and res['c'].shape == (3 * n, 4) and res['d'].shape == (12 * n, 2)
        assert (res['b'].type, res['c'].type, res['d'].type) == (NODE2, NODE1, NODE2)
        for s, row in zip(res['a'].ids, res['b'].ids):
            assert set(row.tolist()) <= set(fx.fixed_dst_ids(int(s), RANGE2)) | {-1}
        for s, row in zip(res['b'].ids.reshape(-1), res['c'].ids):
            assert set(row.tolist()) <= (set(fx.fixed_dst_ids(int(s), RANGE1)) if s >= 0 else set()) | {-1}
        np.testing.assert_equal(res['b'].labels[res['b'].ids >= 0], res['b'].ids[res['b'].ids >= 0])  # attributes resolve
    assert 1 <= _drain(gl, ds, check) <= 20  # (the by_order cursor of node1 is shared with the tests before)
    assert _drain(gl, ds, check) == 20       # a whole epoch
    first = gl.Dataset(g.V(NODE1).batch(5).alias('a').outV(EDGE1).sample(3).by('random').alias('b')
                       .outV(EDGE2).sample(4).by('random').alias('c').values(), fuse_hops=True)
    r1, r2 = first.next(), first.next()
    assert not np.array_equal(r1['c'].ids, r2['c'].ids)  # a fresh stream per batch
    # a filtered hop or a branch point is not fused
    q2 = g.V(NODE1).batch(5).alias('a').outV(EDGE1).sample(3).by('random').alias('b')
    q2.outV(EDGE2).sample(4).by('random').alias('c')
    q2.outV(EDGE2).sample(2).by('random').alias('c2')
    assert gl.Dataset(q2.values(), fuse_hops=True)._chains == {}


def test_device_resident_values_of_a_fused_chain(gl, g):
    """Dataset(fuse_hops=True, device=True): the hops of a fused chain come back as DeviceNodes -- ids, float attributes
    and aggregates are CUDA tensors -- equal to what the host values of the same draws hold; a step outside the chain
    (another strategy) still works downstream, through the host view of its upstream."""
    import torch
    from graphlearn.values import DeviceNodes
    q = g.V(NODE1).batch(6).alias('a') \
         .outV(EDGE1).sample(3).by('random').alias('b') \
         .outV(EDGE2).sample(4).by('random').alias('c') \
         .outV(EDGE1).sample(2).by('topk').alias('d') \
         .values()
    ds = gl.Dataset(q, fuse_hops=True, device=True)
    with pytest.raises(ValueError):
        gl.Dataset(q, device=True)  # device values are what fused chains yield
    res = ds.next()
    b, c, d = res['b'], res['c'], res['d']
    assert isinstance(b, DeviceNodes) and isinstance(c, DeviceNodes) and not isinstance(d, DeviceNodes)
    n = res['a'].shape[0]
    assert b.ids.is_cuda and b.ids.dtype == torch.int64 and b.shape == (n, 3) and c.shape == (3 * n, 4)
    assert (b.type, c.type, d.type) == (NODE2, NODE1, NODE2) and d.shape == (12 * n, 2)
    hb, hc = b.to_host(), c.to_host()
    np.testing.assert_array_equal(hb.ids, b.ids.cpu().numpy())
    for s, row in zip(res['a'].ids, hb.ids):
        assert set(row.tolist()) <= set(fx.fixed_dst_ids(int(s), RANGE2)) | {-1}
    # float attributes and the aggregate, computed on the device, equal the host values' (one lookup / one aggregator call)
    fa = c.float_attrs
    assert fa.is_cuda and fa.shape[:2] == c.shape
    np.testing.assert_array_equal(fa.cpu().numpy(), hc.float_attrs)
    for func in ("sum", "mean", "max", "min"):
        np.testing.assert_array_equal(c.embedding_agg(func).cpu().numpy(), hc.embedding_agg(func))
    # the topk hop below the chain saw the chain's ids
    np.testing.assert_array_equal(d.ids.shape, (12 * n, 2))
    want_d = g.neighbor_sampler(EDGE1, 2, strategy='topk').get(hc.ids.reshape(-1)).layer_nodes(1)
    np.testing.assert_array_equal(d.ids, want_d.ids)


def test_query_errors(gl, g):
    with pytest.raises(ValueError):
        g.V("no_such_type")
    with pytest.raises(ValueError):
        g.V(NODE1).outV(EDGE2)  # edge2 starts at node2
    with pytest.raises(ValueError):
        g.V(NODE1).inV(EDGE2)  # edge2 is directed: no reversed twin
    with pytest.raises(ValueError):
        g.V(NODE1).alias('a').outV(EDGE1).alias('a')
    with pytest.raises(ValueError):
        g.V(NODE1).outV(EDGE1).sample(2).by("nonsense")
    with pytest.raises(ValueError):
        g.V(NODE1).outV(EDGE1).batch(3)
    with pytest.raises(ValueError):
        gl.Dataset(g.V(NODE1).alias('x'))  # not closed with values()
    q = g.V(NODE1).batch(2).alias('a').outV(EDGE1).alias('b').values()  # no sample(): reported when it runs
    with pytest.raises(ValueError):
        gl.Dataset(q).next()
    with pytest.raises(NotImplementedError):
        g.V(NODE1, feed=iter([]))


def test_graph_subgraph_entry_deploy_names_and_close_stops_datasets(gl, g):
    """Graph.SubGraph (graph.py:629-671): sub-graph sampling as a query entry, node and edge seeds, in order and shuffled;
    the deploy / vineyard / KNN entry points fail by name; a prefetching Dataset registers with its graph
    (dag_dataset.py:59) -- Graph.close() stops its thread (checked on a graph of its own below)."""
    import threading
    q = g.SubGraph("entity", "relation", batch_size=8, strategy="in_order_node").alias("sub").values()
    sub = gl.Dataset(q).next()["sub"]
    assert sub.nodes.ids.size == 8 and sub.edge_index.shape[0] == 2
    q = g.SubGraph("relation", "relation", batch_size=2, strategy="random_edge", num_nbrs=[2]).alias("sub").values()
    sub = gl.Dataset(q).next()["sub"]
    assert sub.nodes.ids.size >= 2
    with pytest.raises(ValueError):
        g.SubGraph("entity", "relation", strategy="by_magic")
    for call in (lambda: g.deploy_in_server_mode(), lambda: g.vineyard("x"), lambda: g.search("entity", [[0.0]], None)):
        with pytest.raises(NotImplementedError):
            call()
    ds = gl.Dataset(g.V("entity").batch(4).alias("a").values(), window=2, prefetch=True)
    ds.next()
    assert any(t.name == "gsl-prefetch" and t.is_alive() for t in threading.enumerate())
    ds.close()
    assert not any(t.name == "gsl-prefetch" and t.is_alive() for t in threading.enumerate())

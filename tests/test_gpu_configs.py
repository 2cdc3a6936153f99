"""BASELINE.json's configs as parity cases (scaled where the CPU reference would
take minutes), checked LIVE against the reference's own code (oracle/_ref =
the reference's sampler / aggregator / storage sources; the prebuilt .so travels
to the GPU box) and against the oracle.

  C1 Cora-shaped   : 2,708 nodes / 10,556 directed edges (5,278 mirrored pairs),
                     2-hop RandomSampler [10,5] + MeanAggregator, D=1433
  C2 products-shape: RMAT, RandomWithoutReplacement [15,10], Sum + Mean, D=128
  C3 headline shape: RMAT weighted, EdgeWeightSampler [25,10], Max, D=256
  C5 heterogeneous : 3 edge types (u-i, i-s, u-s), per-type TopkSampler +
                     type-wise SumAggregator, D=256
"""
import numpy as np
import pytest

import glx
import synth
from oracle_bindings import Oracle, RefLib, have_ref

pytestmark = pytest.mark.gpu


def beq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.fixture(scope="module")
def ref():
    if not have_ref():
        pytest.skip("oracle/_ref/libglref.so not present")
    r = RefLib(storage_mode=2, padding_mode=1, default_neighbor_id=0)
    yield r
    r.close()


def _check_rows_are_reference_rows(ref, etype, src, nbr, eid, k):
    """Every sampled (nbr, eid) is an out-edge of its source in the REFERENCE's storage."""
    nb = np.zeros(1 << 16, np.int64)
    ed = np.zeros(1 << 16, np.int64)
    import ctypes
    for i in range(0, src.shape[0], max(1, src.shape[0] // 200)):
        d = ref.L.glref_get_row(ref.h, etype.encode(), int(src[i]), nb.ctypes.data_as(ctypes.c_void_p),
                                ed.ctypes.data_as(ctypes.c_void_p), nb.shape[0])
        pairs = set(zip(nb[:d].tolist(), ed[:d].tolist()))
        for j in range(k):
            if d == 0:
                assert nbr[i, j] == 0 and eid[i, j] == -1
            else:
                assert (int(nbr[i, j]), int(eid[i, j])) in pairs


def test_c1_cora_shaped_two_hop_random_mean(ref, orc):
    rng = np.random.default_rng(1)
    V, pairs, D = 2708, 5278, 1433
    a = rng.integers(0, V, pairs)
    b = rng.integers(0, V, pairs)
    src = np.concatenate([a, b]).astype(np.int64)  # undirected homogeneous: both directions in
    dst = np.concatenate([b, a]).astype(np.int64)  # one edge type (graph.py:357-380)
    X = (rng.random((V, D)) < 0.01).astype(np.float32)  # bag-of-words-like
    ref.add_edges("cites", src, dst)
    ref.add_nodes("paper", np.arange(V, dtype=np.int64), X)
    g = glx.Graph.from_edges(src, dst)
    f = glx.Features(X)
    seeds = rng.permutation(V)[:140].astype(np.int64)
    n1, e1 = g.sample("RandomSampler", seeds, 10, seed=1, call_counter=1)
    n2, e2 = g.sample("RandomSampler", n1.reshape(-1), 5, seed=1, call_counter=2)
    assert n1.shape == (140, 10) and n2.shape == (1400, 5)  # test_gsl_sampling.py shapes
    _check_rows_are_reference_rows(ref, "cites", seeds, n1, e1, 10)
    _check_rows_are_reference_rows(ref, "cites", n1.reshape(-1), n2, e2, 5)
    # aggregation vs the reference's own MeanAggregator on the same ids: bit-exact
    for ids, fan, Sg in ((n2.reshape(-1), 5, 1400), (n1.reshape(-1), 10, 140)):
        seg = (np.arange(ids.shape[0]) // fan).astype(np.int32)
        emb, cnt = f.aggregate("MeanAggregator", ids, seg, Sg)
        remb, rcnt = ref.aggregate("paper", "MeanAggregator", ids, seg, Sg, D)
        assert np.array_equal(cnt, rcnt) and beq(emb, remb)


def test_c2_shape_rwor_sum_mean(ref, orc):
    V, E, D = 30000, 700000, 128
    rp, col, eid, _ = synth.small_graph(V, E, seed=2, weighted=False)
    src = np.repeat(np.arange(V, dtype=np.int64), np.diff(rp))
    order = np.argsort(eid)
    ref.add_edges("c2", src[order], col[order])
    X = (np.random.default_rng(3).random((V, D), dtype=np.float32) * 2 - 1)
    ref.add_nodes("c2n", np.arange(V, dtype=np.int64), X)
    g = glx.Graph(rp, col, eid)
    f = glx.Features(X)
    seeds = np.random.default_rng(4).integers(0, V, 512).astype(np.int64)
    n1, e1 = g.sample("RandomWithoutReplacementSampler", seeds, 15, seed=2, call_counter=1)
    n2, e2 = g.sample("RandomWithoutReplacementSampler", n1.reshape(-1), 10, seed=2, call_counter=2)
    og = dict(row_ptr=rp, col=col, eid=eid)
    on1, oe1 = orc.sample(og, "RandomWithoutReplacementSampler", seeds, 15, seed=2, call_counter=1)
    assert np.array_equal(n1, on1) and np.array_equal(e1, oe1)
    _check_rows_are_reference_rows(ref, "c2", n1.reshape(-1), n2, e2, 10)
    # without replacement: rows with deg >= k have k distinct edge ids
    deg = np.diff(rp)[n1.reshape(-1)]
    for i in np.nonzero(deg >= 10)[0][:2000]:
        assert len(set(e2[i].tolist())) == 10
    ids = n2.reshape(-1)
    seg = (np.arange(ids.shape[0]) // 10).astype(np.int32)
    for name in ("SumAggregator", "MeanAggregator"):
        emb, cnt = f.aggregate(name, ids, seg, n1.size)
        remb, rcnt = ref.aggregate("c2n", name, ids, seg, n1.size, D)
        assert np.array_equal(cnt, rcnt) and beq(emb, remb), name


def test_c3_shape_edge_weight_max(ref, orc):
    V, E, D = 20000, 400000, 256
    rp, col, eid, w = synth.small_graph(V, E, seed=5, weighted=True, hub_degree=20000)
    src = np.repeat(np.arange(V, dtype=np.int64), np.diff(rp))
    order = np.argsort(eid)
    ref.add_edges("c3", src[order], col[order], w[order])
    # the reference's post-Build adjacency == what glx_graph_build produces on the device
    g = glx.Graph.from_edges(src[order], col[order], w[order])
    rows = np.array([0, 1, 2, 77, 4096], np.int64)
    rrp, rcol, reid, rw = ref.export_csr("c3", rows, 1 << 16)
    n, e = g.sample("TopkSampler", rows, 64, padding_mode=0, default_neighbor_id=-1)
    for i in range(rows.shape[0]):
        d = int(rrp[i + 1] - rrp[i])
        m = min(d, 64)
        assert np.array_equal(n[i, :m], rcol[rrp[i]:rrp[i] + m]) and np.array_equal(e[i, :m], reid[rrp[i]:rrp[i] + m])
    X = (np.random.default_rng(6).random((V, D), dtype=np.float32) * 2 - 1)
    ref.add_nodes("c3n", np.arange(V, dtype=np.int64), X)
    f = glx.Features(X)
    seeds = np.random.default_rng(7).integers(0, V, 256).astype(np.int64)
    n1, e1 = g.sample("EdgeWeightSampler", seeds, 25, seed=3, call_counter=1)
    n2, e2 = g.sample("EdgeWeightSampler", n1.reshape(-1), 10, seed=3, call_counter=2)
    og = dict(row_ptr=rp, col=col, eid=eid, weight=w, alias=orc.alias_build(rp, w))
    on2, oe2 = orc.sample(og, "EdgeWeightSampler", n1.reshape(-1), 10, seed=3, call_counter=2)
    assert np.array_equal(n2, on2) and np.array_equal(e2, oe2)
    _check_rows_are_reference_rows(ref, "c3", n1.reshape(-1), n2, e2, 10)
    ids = n2.reshape(-1)
    seg = (np.arange(ids.shape[0]) // 10).astype(np.int32)
    emb, cnt = f.aggregate("MaxAggregator", ids, seg, n1.size)
    remb, rcnt = ref.aggregate("c3n", "MaxAggregator", ids, seg, n1.size, D)
    assert np.array_equal(cnt, rcnt) and beq(emb, remb)


def test_c5_heterogeneous_topk_typewise_sum(ref, orc):
    """user -> item -> shop with three weighted edge types: one device storage per type
    (the reference's HeterDispatcher, heter_dispatcher.h:44-56), per-type Topk."""
    rng = np.random.default_rng(8)
    n_user, n_item, n_shop, D = 4000, 900, 100, 256
    types = {"u-i": (n_user, n_item, 30000, 10), "i-s": (n_item, n_shop, 9000, 10), "u-s": (n_user, n_shop, 9000, 5)}
    graphs = {}
    for t, (ns, nd, ne, k) in types.items():
        s = rng.integers(0, ns, ne).astype(np.int64)
        d = (rng.integers(0, nd, ne) + 10_000_000).astype(np.int64)  # disjoint id ranges per node type
        w = (rng.random(ne) * 0.99 + 0.01).astype(np.float32)
        w = (w + np.arange(ne, dtype=np.float32) * np.float32(2.0 ** -22)).astype(np.float32)
        ref.add_edges(t, s, d, w)
        graphs[t] = (glx.Graph.from_edges(s, d, w), k, ns)
    Xi = rng.standard_normal((n_item, D)).astype(np.float32)
    Xs = rng.standard_normal((n_shop, D)).astype(np.float32)
    item_ids = np.arange(n_item, dtype=np.int64) + 10_000_000
    shop_ids = np.arange(n_shop, dtype=np.int64) + 10_000_000
    ref.add_nodes("item", item_ids, Xi)
    ref.add_nodes("shop", shop_ids, Xs)
    feats = {"u-i": glx.Features(Xi, ids=item_ids), "i-s": glx.Features(Xs, ids=shop_ids),
             "u-s": glx.Features(Xs, ids=shop_ids)}
    ntype = {"u-i": "item", "i-s": "shop", "u-s": "shop"}
    for t, (g, k, ns) in graphs.items():
        q = rng.integers(0, ns + 50, 700).astype(np.int64)  # some ids without out-edges
        for pad in (1, 0):
            ref.set_flags(pad, 0, 0.0)
            n, e = g.sample("TopkSampler", q, k, padding_mode=pad)
            rn, re = ref.sample(t, "TopkSampler", q, k)
            assert np.array_equal(n, rn) and np.array_equal(e, re), (t, pad)
        seg = (np.arange(n.size) // k).astype(np.int32)
        emb, cnt = feats[t].aggregate("SumAggregator", n.reshape(-1), seg, q.shape[0])
        remb, rcnt = ref.aggregate(ntype[t], "SumAggregator", n.reshape(-1), seg, q.shape[0], D)
        assert np.array_equal(cnt, rcnt) and beq(emb, remb), t
    ref.set_flags(1, 0, 0.0)

"""The reference's Python unit tests for neighbour sampling, restated against the glx
engine through the `graphlearn` Python API (graph-learn_amd/python/graphlearn):
graphlearn/python/sampler/tests/test_{random,random_worepl,edge_weight,topk,in_degree,
full}_neighbor_sampling.py over the fixture of test_sampling.py (TSV sources written by
python/tests/utils.py generators -> gl.Graph().node().edge().init()).  Every call goes
Python -> pywrap_graphlearn (pybind11) -> Operator::Process (C++ host mirror) -> HIP.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pyapi_fixture as fx  # noqa: E402

pytestmark = pytest.mark.gpu

NODE1, NODE2 = "node1", "node2"
EDGE1, EDGE2, EDGE3 = "edge1", "edge2", "edge3"
RANGE1, RANGE2 = (0, 100), (100, 200)
DEFAULT_ID, DEFAULT_INT, DEFAULT_FLOAT, DEFAULT_STR = -1, 1000, 999.9, "hehe"


@pytest.fixture(scope="module")
def gl():
    import graphlearn
    return graphlearn


@pytest.fixture(scope="module")
def g(gl, tmp_path_factory):
    d = str(tmp_path_factory.mktemp("gl_data"))
    gl.set_default_neighbor_id(DEFAULT_ID)
    gl.set_default_int_attribute(DEFAULT_INT)
    gl.set_default_float_attribute(DEFAULT_FLOAT)
    gl.set_default_string_attribute(DEFAULT_STR)
    gl.set_padding_mode(gl.REPLICATE)
    gl.set_sampling_seed(20240923)
    n1 = fx.write_nodes(d, "node1", RANGE1, [fx.ATTRIBUTED])
    n2 = fx.write_nodes(d, "node2", RANGE2, [fx.WEIGHTED, fx.LABELED])
    e1 = fx.write_edges(d, "edge1", RANGE1, RANGE2, [fx.ATTRIBUTED, fx.LABELED])
    e2 = fx.write_edges(d, "edge2", RANGE2, RANGE1, [fx.ATTRIBUTED, fx.WEIGHTED])
    e3 = fx.write_edges(d, "edge3", RANGE2, RANGE2, [fx.WEIGHTED])
    n3 = fx.write_entity_nodes(d, "entity")
    e4 = fx.write_relation_edges(d, "relation")
    cond = fx.write_cond_nodes(d, "cond_item")
    train = fx.write_nodes(d, "node1_train", (0, 50), [fx.WEIGHTED])
    graph = gl.Graph() \
        .node(n1, NODE1, gl.Decoder(attr_types=fx.ATTR_TYPES)) \
        .node(n2, NODE2, gl.Decoder(weighted=True, labeled=True)) \
        .node(n3, "entity", gl.Decoder(attr_types=["float"] * 4, labeled=True)) \
        .node(train, NODE1, gl.Decoder(weighted=True), mask=gl.Mask.TRAIN) \
        .edge(e1, (NODE1, NODE2, EDGE1), gl.Decoder(attr_types=fx.ATTR_TYPES, labeled=True), directed=False) \
        .edge(e2, (NODE2, NODE1, EDGE2), gl.Decoder(attr_types=fx.ATTR_TYPES, weighted=True)) \
        .edge(e3, (NODE2, NODE2, EDGE3), gl.Decoder(weighted=True), directed=False) \
        .edge(e4, ("entity", "entity", "relation"), gl.Decoder(weighted=True), directed=False) \
        .node(cond, "cond_item", gl.Decoder(attr_types=["int", "int", "float", "string"], weighted=True)) \
        .edge(e4, ("cond_item", "cond_item", "cond_sim"), gl.Decoder(weighted=True), directed=True)
    graph.init(tracker=d)
    yield graph
    graph.close()


NODE2_IDS = fx.fixed_dst_ids(range(*RANGE1), RANGE2)
SEEDS1 = np.array([2, 7, 8])
SEEDS2 = np.array([102, 107, 108])
SEEDS1_MISSING = np.array([5, 10, 110])  # no out-edges (multiples of 5) or not a node1 id at all


class padding(object):
    """The fixture's default is REPLICATE, like the reference's test_sampling.py.  Under REPLICATE the
    reference's padder ignores the sampled indices (replicate_padder.h:45-50): the alias samplers
    (edge_weight, in_degree) then return the FIRST min(k, degree) neighbours and, for degree < k, read
    out of bounds; glx default-fills instead.  Tests that want draws use CIRCULAR."""

    def __init__(self, gl, mode):
        self.gl, self.mode = gl, mode

    def __enter__(self):
        self.gl.set_padding_mode(self.mode)

    def __exit__(self, *exc):
        self.gl.set_padding_mode(self.gl.REPLICATE)


def one_hop_checks(gl, g, strategy, k=6):
    with padding(gl, gl.CIRCULAR if strategy == "in_degree" else gl.REPLICATE):
        nbrs = g.neighbor_sampler(EDGE1, expand_factor=k, strategy=strategy).get(SEEDS1)
    edges, nodes = nbrs.layer_edges(1), nbrs.layer_nodes(1)
    fx.expect_edges_follow_generator(edges, RANGE2, SEEDS1, DEFAULT_ID)
    assert (edges.src_type, edges.dst_type, edges.edge_type) == (NODE1, NODE2, EDGE1)
    assert edges.src_ids.size == SEEDS1.size * k and edges.shape == (SEEDS1.size, k)
    fx.expect_edge_columns(edges, labeled=True, attributed=True)
    np.testing.assert_equal(nodes.ids, edges.dst_ids)
    assert set(nodes.ids.reshape(-1).tolist()) <= set(NODE2_IDS) and nodes.type == NODE2
    fx.expect_node_columns(nodes, weighted=True, labeled=True)
    return nbrs


@pytest.mark.parametrize("strategy", ["random", "in_degree"])
def test_1hop(gl, g, strategy):
    """test_random_neighbor_sampling.py::test_1hop / test_in_degree_neighbor_sampling.py::test_1hop."""
    nbrs = one_hop_checks(gl, g, strategy)
    deg = nbrs.layer_edges(1).src_nodes.get_out_degrees(EDGE1)
    np.testing.assert_equal(deg.reshape(-1), np.repeat(SEEDS1 % 5, 6))


@pytest.mark.parametrize("strategy", ["random", "random_without_replacement", "in_degree"])
def test_1hop_with_neighbor_missing(gl, g, strategy):
    """Seeds without out-edges: default neighbour id and default columns everywhere."""
    k = 6
    nbrs = g.neighbor_sampler(EDGE1, expand_factor=k, strategy=strategy).get(SEEDS1_MISSING)
    edges, nodes = nbrs.layer_edges(1), nbrs.layer_nodes(1)
    np.testing.assert_equal(edges.dst_ids.reshape(-1), [DEFAULT_ID] * (3 * k))
    fx.expect_default_edge_columns(edges, labeled=True, attributed=True, default_int=DEFAULT_INT,
                                   default_float=DEFAULT_FLOAT, default_string=DEFAULT_STR)
    np.testing.assert_equal(nodes.ids, edges.dst_ids)
    np.testing.assert_equal(nodes.weights.reshape(-1), [0.0] * (3 * k))   # check_not_exist_node_weights
    np.testing.assert_equal(nodes.labels.reshape(-1), [-1] * (3 * k))     # check_not_exist_node_labels


@pytest.mark.parametrize("strategy", ["random", "random_without_replacement", "edge_weight"])
def test_2hop(gl, g, strategy):
    """node1 -edge1-> node2 -edge2-> node1 with fan-out [3, 2]."""
    first = EDGE1 if strategy != "edge_weight" else EDGE3  # edge1 carries no weights
    seeds = SEEDS1 if strategy != "edge_weight" else SEEDS2
    ks = [3, 2]
    # degree 2 < 3: replicate padding would default-fill the permutation / alias samplers
    with padding(gl, gl.REPLICATE if strategy == "random" else gl.CIRCULAR):
        nbrs = g.neighbor_sampler([first, EDGE2], expand_factor=ks, strategy=strategy).get(seeds)
    edges, nodes = nbrs.layer_edges(1), nbrs.layer_nodes(1)
    assert edges.shape == (3, 3)
    np.testing.assert_equal(nodes.ids, edges.dst_ids)
    if strategy != "edge_weight":
        fx.expect_edges_follow_generator(edges, RANGE2, seeds, DEFAULT_ID)
        fx.expect_edge_columns(edges, labeled=True, attributed=True)
    fx.expect_node_columns(nodes, weighted=True, labeled=True)
    hop1 = nodes.ids.reshape(-1)
    edges, nodes = nbrs.layer_edges(2), nbrs.layer_nodes(2)
    assert edges.shape == (9, 2) and (edges.src_type, edges.dst_type, edges.edge_type) == (NODE2, NODE1, EDGE2)
    np.testing.assert_equal(edges.src_ids.reshape(-1), np.repeat(hop1, 2))
    fx.expect_edges_follow_generator(edges, RANGE1, hop1, DEFAULT_ID)
    exist = edges.dst_ids != DEFAULT_ID
    np.testing.assert_almost_equal(edges.weights[exist], 0.1 * (edges.src_ids + 0.1 * edges.dst_ids)[exist], decimal=5)
    np.testing.assert_equal(edges.weights[~exist], 0.0)
    np.testing.assert_equal(nodes.ids, edges.dst_ids)
    real = nodes.ids != DEFAULT_ID
    np.testing.assert_equal(nodes.int_attrs[..., 0][real], nodes.ids[real])
    np.testing.assert_equal(nodes.int_attrs[..., 0][~real], DEFAULT_INT)
    # the 4th attribute ('string', 10) is stored as Hash64('hehe') % 10
    np.testing.assert_equal(nodes.int_attrs[..., 1][real], gl.pywrap.hash64(b"hehe") % 10)


@pytest.mark.parametrize("mode", ["replicate", "circular"])
def test_random_without_replacement_1hop_padding(gl, g, mode):
    """test_random_worepl_neighbor_sampling.py::test_1hop_{circular,replicate}_padding: k = 6 exceeds
    every degree, so a row is exactly its neighbour set (plus the default id under replicate padding)."""
    gl.set_padding_mode(gl.REPLICATE if mode == "replicate" else gl.CIRCULAR)
    try:
        nbrs = g.neighbor_sampler(EDGE1, expand_factor=6, strategy="random_without_replacement").get(SEEDS1)
        for s, row in zip(SEEDS1, nbrs.layer_nodes(1).ids):
            want = set(fx.fixed_dst_ids(int(s), RANGE2))
            if mode == "replicate":
                want.add(DEFAULT_ID)
            assert set(row.tolist()) == want
    finally:
        gl.set_padding_mode(gl.REPLICATE)


def test_random_without_replacement_returns_distinct_neighbours(gl, g):
    """test_random_worepl_neighbor_sampling.py: with k >= degree every neighbour shows up."""
    gl.set_padding_mode(gl.CIRCULAR)
    try:
        seeds = np.array([104, 109, 103])
        k = 4
        nbrs = g.neighbor_sampler(EDGE2, expand_factor=k, strategy="random_without_replacement").get(seeds)
        got = nbrs.layer_nodes(1).ids
        for row, s in zip(got, seeds):
            want = fx.fixed_dst_ids(int(s), RANGE1)
            assert set(row.tolist()) == set(want) and len(want) == s % 5
            assert sorted(row[:len(want)].tolist()) == sorted(want)  # a permutation first, then it repeats
    finally:
        gl.set_padding_mode(gl.REPLICATE)


@pytest.mark.parametrize("mode", ["replicate", "circular"])
def test_topk_padding_modes(gl, g, mode):
    """test_topk_neighbor_sampling.py: exact expectations for both padding modes."""
    gl.set_padding_mode(gl.REPLICATE if mode == "replicate" else gl.CIRCULAR)
    try:
        seeds = np.array([102, 107, 108, 105, 104])
        nbrs = g.neighbor_sampler(EDGE2, 6, strategy="topk").get(seeds)
        edges = nbrs.layer_edges(1)
        np.testing.assert_equal(edges.dst_ids.reshape(-1), fx.expected_topk(seeds, RANGE1, 6, DEFAULT_ID, mode))
        src, dst, w = edges.src_ids.reshape(-1), edges.dst_ids.reshape(-1), edges.weights.reshape(-1)
        for s, d_, w_ in zip(src, dst, w):  # check_half_exist_edge_weights
            np.testing.assert_almost_equal(w_, 0.0 if d_ == DEFAULT_ID else 0.1 * (s + 0.1 * d_), decimal=5)
    finally:
        gl.set_padding_mode(gl.REPLICATE)


def test_edge_weight_1hop(gl, g):
    """test_edge_weight_neighbor_sampling.py::test_1hop on the weighted homogeneous type."""
    k = 6
    with padding(gl, gl.CIRCULAR):
        nbrs = g.neighbor_sampler(EDGE3, expand_factor=k, strategy="edge_weight").get(SEEDS2)
    edges, nodes = nbrs.layer_edges(1), nbrs.layer_nodes(1)
    assert edges.shape == (3, k) and nodes.type == NODE2
    # edge3 was added with directed=False: both directions live in one adjacency
    src, dst = edges.src_ids.reshape(-1), edges.dst_ids.reshape(-1)
    for s, d_ in zip(src.tolist(), dst.tolist()):
        forward = d_ in fx.fixed_dst_ids(s, RANGE2)
        backward = s in fx.fixed_dst_ids(d_, RANGE2)
        assert forward or backward, (s, d_)
    fx.expect_node_columns(nodes, weighted=True, labeled=True)


def test_full_neighbor_sampler(gl, g):
    """test_full_neighbor_sampling.py: every neighbour, ragged rows."""
    nbrs = g.neighbor_sampler(EDGE1, 0, strategy="full").get(SEEDS1)
    nodes, edges = nbrs.layer_nodes(1), nbrs.layer_edges(1)
    assert nodes.offsets == [int(s % 5) for s in SEEDS1]
    for row, s in zip(nodes, SEEDS1):
        assert sorted(row.ids.tolist()) == sorted(fx.fixed_dst_ids(int(s), RANGE2))
    assert len(nodes.indices) == sum(nodes.offsets) and nodes.dense_shape == (3, 3)
    np.testing.assert_equal(edges.src_ids, np.repeat(SEEDS1, SEEDS1 % 5))
    capped = g.neighbor_sampler(EDGE1, 2, strategy="full").get(SEEDS1).layer_nodes(1)
    assert capped.offsets == [2, 2, 2]
    # vertices without neighbours: the next hop starts from an empty frontier
    lonely = g.neighbor_sampler([EDGE1, EDGE2], [0, 0], strategy="full").get(np.array([-3, 10 ** 9]))
    assert lonely.layer_nodes(1).offsets == [0, 0] and lonely.layer_nodes(1).ids.size == 0
    assert lonely.layer_nodes(2).offsets == [] and lonely.layer_nodes(2).dense_shape == (0, 0)


def test_reverse_edges_of_undirected_types(gl, g):
    """directed=False on a heterogeneous type registers `<type>_reverse` (graph.py:357-380)."""
    assert not g.is_directed(EDGE1) and g.is_directed(EDGE2)
    topo = g.get_topology()
    assert (topo.get_src_type(EDGE1 + "_reverse"), topo.get_dst_type(EDGE1 + "_reverse")) == (NODE2, NODE1)
    dst = 102
    want = sorted(s for s in range(*RANGE1) if dst in fx.fixed_dst_ids(s, RANGE2))
    got = g.neighbor_sampler(EDGE1 + "_reverse", 0, strategy="full").get(np.array([dst])).layer_nodes(1)
    assert sorted(got.ids.tolist()) == want


def test_masked_source_is_a_separate_type(gl, g):
    masked = gl.get_mask_type(NODE1, gl.Mask.TRAIN)
    assert masked == "MASK*node1"
    vals = g.lookup_nodes(masked, np.array([3, 49, 50]))
    np.testing.assert_almost_equal(vals.weights, [0.3, 4.9, 0.0], decimal=5)  # 50 is not in the TRAIN file


def test_embedding_agg_equals_numpy(gl, g):
    """Nodes.embedding_agg -> Sum/Mean/Max/Min/Prod aggregators over float attributes."""
    ids = np.array([[1, 2, 3], [10, 20, 119], [7, 7, 500]])  # 500 is unknown -> default row
    nodes = g.get_nodes("entity", ids)
    table = np.array([[v * 0.1, v * 0.2, v * 0.3, v * 0.4] for v in range(120)], np.float64)
    rows = np.stack([table[v] if v < 120 else np.full(4, DEFAULT_FLOAT) for v in ids.reshape(-1)]).reshape(3, 3, 4)
    np.testing.assert_allclose(nodes.embedding_agg("sum"), rows.sum(1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nodes.embedding_agg("mean"), rows.mean(1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nodes.embedding_agg("min"), rows.min(1), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nodes.embedding_agg("max"), np.maximum(rows.max(1), -37.0), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(nodes.embedding_agg("prod"), rows.prod(1), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(nodes.float_attrs, rows, rtol=1e-5, atol=1e-5)
    np.testing.assert_equal(nodes.labels, np.where(ids < 120, ids, -1))


def test_seeded_sampling_is_reproducible(gl, g):
    """New with this engine: a pinned call counter replays the same draws."""
    s = g.neighbor_sampler([EDGE3, EDGE3], expand_factor=[5, 4], strategy="edge_weight")
    with padding(gl, gl.CIRCULAR):
        s.set_call_counter(77)
        a = s.get(SEEDS2)
        b = s.get(SEEDS2)
        s.set_call_counter(78)
        c = s.get(SEEDS2)
    for hop in (1, 2):
        np.testing.assert_equal(a.layer_nodes(hop).ids, b.layer_nodes(hop).ids)
    assert not np.array_equal(a.layer_nodes(2).ids, c.layer_nodes(2).ids)


def test_error_surface(gl, g, tmp_path):
    with pytest.raises(ValueError):
        g.neighbor_sampler("no_such_edge", 3)
    with pytest.raises(ValueError):
        g.neighbor_sampler([EDGE1, EDGE2], expand_factor=[3]).get(SEEDS1)
    with pytest.raises(NotImplementedError):
        g.V(NODE1, feed=iter(()))  # generator-fed queries are the one GSL source not served (tests/test_gpu_pyapi_gsl.py)
    bad = fx.write_nodes(str(tmp_path), "bad_nodes", (0, 5), [fx.WEIGHTED])
    other = gl.Graph().node(bad, "x", gl.Decoder(labeled=True))  # file has weight:float, decoder says label
    with pytest.raises(gl.InvalidArgumentError):
        other.init()
    other.close()


def test_device_tensor_path_equals_numpy_path(gl, g):
    """NeighborSampler.get_device: all hops in one glx_sample_hops call on torch CUDA tensors, same
    draws as the request-per-hop numpy path; device_features().aggregate == Nodes.embedding_agg."""
    import torch
    s = g.neighbor_sampler([EDGE3, EDGE3], expand_factor=[5, 4], strategy="edge_weight")
    with padding(gl, gl.CIRCULAR):
        s.set_call_counter(500)
        host = s.get(SEEDS2)
        dev = s.get_device(torch.from_numpy(SEEDS2).cuda(), call_counter=500)
    for hop in (1, 2):
        nbr, eid = dev[hop - 1]
        assert nbr.is_cuda and tuple(nbr.shape) == host.layer_nodes(hop).ids.shape
        np.testing.assert_equal(nbr.cpu().numpy(), host.layer_nodes(hop).ids)
        np.testing.assert_equal(eid.cpu().numpy(), host.layer_edges(hop).edge_ids)
    feats = g.device_features("entity")
    ids = np.array([[1, 2, 3], [10, 20, 119]])
    seg = torch.arange(2, dtype=torch.int32).repeat_interleave(3).cuda()
    emb, cnt = feats.aggregate("MeanAggregator", torch.from_numpy(ids.reshape(-1)).cuda(), seg, 2,
                               default_attr=DEFAULT_FLOAT)
    want = g.get_nodes("entity", ids).embedding_agg("mean")
    np.testing.assert_equal(emb.cpu().numpy().view(np.uint32), want.astype(np.float32).view(np.uint32))
    assert cnt.tolist() == [3, 3]


@pytest.mark.parametrize("strategy", ["random", "in_degree", "soft_in_degree"])
def test_negative_sampler_on_edge_type(gl, g, strategy):
    """test_random_negative_sampling.py / test_in_degree_negavtive_sampling.py (the reference skips both
    as 'not always right': a neighbour can slip through once three retry blocks are exhausted; with 100
    candidates and at most 4 neighbours that does not happen here)."""
    k = 6
    nodes = g.negative_sampler(EDGE1, expand_factor=k, strategy=strategy).get(SEEDS1)
    assert nodes.type == NODE2 and nodes.ids.shape == (SEEDS1.size, k)
    assert set(nodes.ids.reshape(-1).tolist()) <= set(NODE2_IDS)
    if strategy == "in_degree":  # strict: no true neighbour among the negatives
        for s, row in zip(SEEDS1, nodes.ids):
            assert not set(row.tolist()) & set(fx.fixed_dst_ids(int(s), RANGE2))
    fx.expect_node_columns(nodes, weighted=True, labeled=True)


def test_negative_sampler_node_weight(gl, g):
    """test_node_weight_negavtive_sampling.py: negatives come from the node type's ids and never
    from the request's own ids."""
    k = 6
    nodes = g.negative_sampler(NODE2, expand_factor=k, strategy="node_weight").get(SEEDS2)
    assert nodes.type == NODE2 and nodes.ids.size == SEEDS2.size * k
    assert set(nodes.ids.reshape(-1).tolist()) <= set(range(*RANGE2)) - set(SEEDS2.tolist())
    with pytest.raises(ValueError):
        g.negative_sampler(EDGE1, 3, strategy="node_weight")  # needs a node type
    with pytest.raises(ValueError):
        g.negative_sampler(NODE2, 3, strategy="random")  # needs an edge type
    with pytest.raises(gl.InvalidArgumentError):
        g.negative_sampler(NODE1, 3, strategy="node_weight").get(SEEDS1)  # node1 has no weights


def test_node_and_edge_traversal_samplers(gl, g):
    """g.node_sampler / g.edge_sampler (python/sampler/{node,edge}_sampler.py): by_order covers every id
    once per epoch with a short last batch and an OutOfRangeError at the boundary; shuffle covers a
    permutation; random never ends; EDGE_SRC / EDGE_DST walk the distinct end points."""
    s = g.node_sampler(NODE2, batch_size=32, strategy="by_order")
    seen = []
    for _ in range(2):  # two epochs
        got = []
        while True:
            try:
                got.append(s.get().ids)
            except gl.OutOfRangeError:
                break
        assert [b.size for b in got] == [32, 32, 32, 4]
        seen.append(np.concatenate(got))
    assert seen[0].tolist() == list(range(*RANGE2)) == seen[1].tolist()
    nodes = g.node_sampler(NODE2, batch_size=5).get()
    fx.expect_node_columns(nodes, weighted=True, labeled=True)  # the batch is a normal Nodes object
    sh = g.node_sampler(NODE1, batch_size=64, strategy="shuffle")
    epoch = np.concatenate([sh.get().ids, sh.get().ids])
    assert sorted(epoch.tolist()) == list(range(*RANGE1)) and epoch.tolist() != list(range(*RANGE1))
    with pytest.raises(gl.OutOfRangeError):
        sh.get()
    rnd = g.node_sampler(NODE1, batch_size=300, strategy="random").get()
    assert rnd.ids.size == 300 and set(rnd.ids.tolist()) <= set(range(*RANGE1))
    srcs = g.node_sampler(EDGE1, batch_size=1000, node_from=gl.EDGE_SRC).get()
    assert srcs.type == NODE1 and srcs.ids.tolist() == [s_ for s_ in range(*RANGE1) if s_ % 5]
    dsts = g.node_sampler(EDGE1, batch_size=1000, node_from=gl.EDGE_DST).get()
    assert dsts.type == NODE2 and sorted(dsts.ids.tolist()) == sorted(set(NODE2_IDS))
    es = g.edge_sampler(EDGE2, batch_size=50, strategy="by_order")
    first = es.get()
    assert first.edge_ids.tolist() == list(range(50)) and first.src_type == NODE2 and first.dst_type == NODE1
    fx.expect_edges_follow_generator(first, RANGE1, range(*RANGE2), DEFAULT_ID)
    fx.expect_edge_columns(first, weighted=True, attributed=True)  # lookups by the sampled edge ids
    total = 50
    while True:
        try:
            total += es.get().src_ids.size
        except gl.OutOfRangeError:
            break
    assert total == len(fx.fixed_dst_ids(range(*RANGE2), RANGE1))
    # seed batches feed the device samplers
    seeds = g.node_sampler(NODE1, batch_size=16, strategy="random").get().ids
    nbrs = g.neighbor_sampler(EDGE1, 3, strategy="random").get(seeds)
    assert nbrs.layer_nodes(1).ids.shape == (16, 3)


def test_shuffle_traversal_walks_windows_of_the_shuffle_buffer_size(gl, g):
    """node_generator.h:168-216 / edge_generator.h: a "shuffle" epoch is NOT one permutation of the whole type -- the
    ids are taken ShuffleBufferSize (10240 by default) consecutive ones at a time and shuffled inside that window, and a
    window outlives the request that filled it.  With set_shuffle_buffer_size(16): ids 0..15 in some order, then
    16..31, ... for nodes, edges and edge end points alike."""
    gl.set_shuffle_buffer_size(16)
    try:
        for what in ("node", "edge"):
            if what == "node":
                s = g.node_sampler("entity", batch_size=10, strategy="shuffle")  # a type no other test shuffles
                get = lambda: s.get().ids  # noqa: E731
                all_ids = None
            else:
                s = g.edge_sampler(EDGE3, batch_size=10, strategy="shuffle")
                get = lambda: s.get().edge_ids  # noqa: E731
            for epoch in range(2):
                got = []
                while True:
                    try:
                        got.append(get())
                    except gl.OutOfRangeError:
                        break
                flat = np.concatenate(got)
                if what == "node":
                    if all_ids is None:
                        all_ids = np.sort(flat)  # the type's ids in storage order are ascending in this fixture
                    order = {int(v): i for i, v in enumerate(all_ids)}
                    pos = np.array([order[int(v)] for v in flat])
                else:
                    pos = flat
                n = pos.shape[0]
                assert sorted(pos.tolist()) == list(range(n))
                shuffled_somewhere = False
                for lo in range(0, n, 16):
                    window = pos[lo:lo + 16]
                    assert sorted(window.tolist()) == list(range(lo, min(n, lo + 16))), (what, epoch, lo)
                    shuffled_somewhere |= window.tolist() != sorted(window.tolist())
                assert shuffled_somewhere
    finally:
        gl.set_shuffle_buffer_size(10240)


def test_neighbor_loader_device_batches(gl, g):
    """gl.NeighborLoader: every batch is produced and kept on the GPU; one epoch covers every seed once;
    a batch equals what the request-per-hop numpy path returns for the same pinned call counter."""
    import torch
    with padding(gl, gl.CIRCULAR):
        loader = gl.NeighborLoader(g, "entity", ["relation", "relation"], [4, 3], batch_size=32,
                                   strategy="edge_weight", shuffle=True)
        assert len(loader) == 4  # 120 entity nodes
        seen = []
        batches = list(loader)
        for i, b in enumerate(batches):
            assert b.seeds.is_cuda and b.num_hops == 2
            bs = b.seeds.shape[0]
            assert tuple(b.nbr[0].shape) == (bs, 4) and tuple(b.nbr[1].shape) == (bs * 4, 3)
            assert tuple(b.x[0].shape) == (bs, 4) and tuple(b.x[2].shape) == (bs * 12, 4)
            seen.append(b.seeds.cpu().numpy())
            # features are the generator's closed form (0.1 id .. 0.4 id), default row for padding ids
            ids = b.frontier(2).cpu().numpy()
            want = np.where((ids >= 0) & (ids < 120), ids, np.nan)[:, None] * np.array([0.1, 0.2, 0.3, 0.4])
            got = b.x[2].cpu().numpy()
            known = ~np.isnan(want[:, 0])
            np.testing.assert_allclose(got[known], want[known], rtol=1e-5, atol=1e-5)
            src, dst = b.edge_index(1)
            assert torch.equal(b.frontier(1)[src], b.nbr[0].reshape(-1).repeat_interleave(3))
            # same draws as the numpy path with the same call counter
            s = g.neighbor_sampler(["relation", "relation"], [4, 3], strategy="edge_weight")
            s.set_call_counter(i * 2)
            host = s.get(seen[-1])
            np.testing.assert_equal(b.nbr[1].cpu().numpy(), host.layer_nodes(2).ids)
            np.testing.assert_equal(b.eid[0].cpu().numpy(), host.layer_edges(1).edge_ids)
        assert sorted(np.concatenate(seen).tolist()) == list(range(120))
        second = [b.seeds.cpu().numpy() for b in loader]  # next epoch: another permutation
        assert sorted(np.concatenate(second).tolist()) == list(range(120))
        assert not np.array_equal(np.concatenate(second), np.concatenate(seen))


def test_random_walk_deepwalk(gl, g):
    """GSL random_walk(edge_type, walk_len, 1.0, 1.0) (python/gsl/tests/test_gsl_random_walk.py): every
    step follows an edge of the type; a vertex without out-edges yields the default id."""
    import torch
    seeds = np.array([102, 107, 108, 111])
    walks = g.random_walk(EDGE3, seeds, 10, call_counter=0)
    assert walks.shape == (4, 10)
    prev = seeds
    for step in range(10):
        cur = walks[:, step]
        for a, b in zip(prev.tolist(), cur.tolist()):
            if a == DEFAULT_ID:
                assert b == DEFAULT_ID
                continue
            out = set(fx.fixed_dst_ids(a, RANGE2)) | {s for s in range(*RANGE2) if a in fx.fixed_dst_ids(s, RANGE2)}
            assert (b in out) if out else (b == DEFAULT_ID), (a, b)
        prev = cur
    dev = g.random_walk(EDGE3, torch.from_numpy(seeds).cuda(), 10)
    assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), walks)  # same pinned stream


def test_python_stack_equals_live_reference_on_the_same_records(gl, g):
    """TSV file -> glx loader -> device build -> Python API  ==  the reference's own storages and
    operators fed the same records (oracle/_ref): Topk rows, Full rows (post-Build row order) and
    out-degrees are identical for every source id of the weighted edge type."""
    from oracle_bindings import RefLib, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    src, dst, w = [], [], []
    for s in range(*RANGE2):
        for d_ in fx.fixed_dst_ids(s, RANGE1):
            src.append(s)
            dst.append(d_)
            w.append(np.float32(float("%f" % ((s + 0.1 * d_) / 10.0))))  # what the loader parses from the file
    ref = RefLib(padding_mode=1, default_neighbor_id=DEFAULT_ID)
    try:
        ref.add_edges("edge2", np.array(src, np.int64), np.array(dst, np.int64), np.array(w, np.float32))
        ids = np.arange(*RANGE2, dtype=np.int64)
        with padding(gl, gl.CIRCULAR):
            mine = g.neighbor_sampler(EDGE2, 4, strategy="topk").get(ids)
        rn, re_ = ref.sample("edge2", "TopkSampler", ids, 4)
        np.testing.assert_equal(mine.layer_nodes(1).ids, rn)
        np.testing.assert_equal(mine.layer_edges(1).edge_ids, re_)
        full = g.neighbor_sampler(EDGE2, 0, strategy="full").get(ids)
        deg, fn, fe = ref.sample_full("edge2", ids, 0)
        assert full.layer_nodes(1).offsets == deg.tolist()
        np.testing.assert_equal(full.layer_nodes(1).ids, fn)
        np.testing.assert_equal(full.layer_edges(1).edge_ids, fe)
        np.testing.assert_equal(g.out_degrees(ids, EDGE2), deg)
    finally:
        ref.close()


def test_get_stats(gl, g):
    stats = g.get_stats()
    assert stats[NODE1] == [100] and stats[NODE2] == [100] and stats["entity"] == [120]
    assert stats[EDGE2] == [len(fx.fixed_dst_ids(range(*RANGE2), RANGE1))]
    # directed=False on a homogeneous type: both directions live in ONE storage (2 E records) AND the type is declared by
    # two sources, which BuildLocalCount multiplies in once more (graph_store.cc:196-201, 305-311): 4 E, as the reference
    assert stats[EDGE3] == [4 * len(fx.fixed_dst_ids(range(*RANGE2), RANGE2))]
    assert stats[EDGE1] == stats[EDGE1 + "_reverse"] == [len(fx.fixed_dst_ids(range(*RANGE1), RANGE2))]
    assert stats["MASK*node1"] == [50]


def test_subgraph_sampler_through_the_python_api(gl, g):
    """Graph.subgraph_sampler -> "SubGraphSampler" (subgraph_sampler.{h,cc}) on the relation edges i -> i+2, i+3, i+5
    (both directions in one storage: directed=False): node set = seeds + sorted neighbour set, edges = every ordered
    pair of listed nodes joined by a relation edge, reported in both directions; distances as the SEAL labelling."""
    gl.set_default_full_nbr_num(100)
    seeds = np.array([10, 13])
    sub = g.subgraph_sampler("relation", num_nbrs=[6], need_dist=True).get(seeds)
    nbrs_of = lambda v: [x for d in (2, 3, 5) for x in (v + d, v - d) if 0 <= x < 120]  # noqa: E731
    want_nodes = list(seeds) + sorted(set(x for v in seeds for x in nbrs_of(v)))
    np.testing.assert_equal(sub.nodes.ids, want_nodes)
    idx = sub.edge_index
    assert idx.shape[0] == 2 and idx.shape[1] % 2 == 0
    pairs = set()
    for i, u in enumerate(want_nodes):
        for j, v in enumerate(want_nodes):
            if abs(int(u) - int(v)) in (2, 3, 5):
                pairs.add((i, j))
    got = set(zip(idx[0].tolist(), idx[1].tolist()))
    assert got == pairs
    # entries come in (i, j), (j, i) pairs sharing the edge id
    assert np.array_equal(idx[0][0::2], idx[1][1::2]) and np.array_equal(idx[1][0::2], idx[0][1::2])
    assert np.array_equal(sub.edges[0::2], sub.edges[1::2])
    assert sub.dist_to_src[0] == 0 and sub.dist_to_dst[1] == 0 and sub.dist_to_src[1] == 0 and sub.dist_to_dst[0] == 0
    # 10 and 13 are joined directly (|d| = 3); every listed neighbour of 13 is one step from it
    for j, v in enumerate(want_nodes[2:], start=2):
        if abs(int(v) - 13) in (2, 3, 5):
            assert sub.dist_to_dst[j] == 1
    # zero hops: the seeds and the edges among them
    sub0 = g.subgraph_sampler("relation").get(np.array([20, 22, 40]))
    np.testing.assert_equal(sub0.nodes.ids, [20, 22, 40])
    assert set(zip(sub0.edge_index[0].tolist(), sub0.edge_index[1].tolist())) == {(0, 1), (1, 0)}


def test_conditional_negative_sampler_through_the_python_api(gl, g):
    """Graph.negative_sampler(conditional=True) -> "ConditionalNegativeSampler" (conditional_negative_sampler.cc).  EDGE2
    runs node2 -> node1; node1 carries ints [v, Hash64("hehe") % 10], floats [float(v)], strings [str(v)]: int column 1
    is one group holding every candidate, int column 0 / the float / the string column put every node in a group of its
    own -- whose only member, the dst, is excluded, so those slots fall through to the default sampler."""
    src = np.array([102, 107, 108, 111])
    dst = np.array([fx.fixed_dst_ids(int(s_), RANGE1)[0] for s_ in src])
    cands = set(fx.fixed_dst_ids(range(*RANGE2), RANGE1))
    for strategy in ("random", "in_degree"):
        ns = g.negative_sampler(EDGE2, 6, strategy, conditional=True, unique=True, int_cols=[1, 0], int_props=[0.5, 0.25],
                                str_cols=[0], str_props=[0.25])
        ns.set_call_counter(5)
        nodes = ns.get(src, dst)
        assert nodes.ids.shape == (4, 6) and nodes.type == NODE1
        seen = set()
        for r in range(4):
            seen |= set(fx.fixed_dst_ids(int(src[r]), RANGE1)) | {int(dst[r])}
            row = nodes.ids[r].tolist()
            assert set(row) <= cands and not (set(row) & seen), (strategy, r, row)
            seen |= set(row)  # unique: accepted ids join the exclusion set
        assert len(set(nodes.ids.reshape(-1).tolist())) == 24
        again = g.negative_sampler(EDGE2, 6, strategy, conditional=True, unique=True, int_cols=[1, 0], int_props=[0.5, 0.25],
                                   str_cols=[0], str_props=[0.25])
        again.set_call_counter(5)
        np.testing.assert_equal(again.get(src, dst).ids, nodes.ids)  # the call counter pins the stream
    nw = g.negative_sampler(NODE2, 4, "node_weight", conditional=True)  # node2 has no attributes: default table only
    out = nw.get(np.array([1, 2]), np.array([150, 151]))
    assert out.ids.shape == (2, 4) and set(out.ids.reshape(-1).tolist()) <= set(range(*RANGE2)) - {150, 151}
    with pytest.raises(ValueError):
        g.negative_sampler(EDGE2, 6, "random", conditional=True, int_cols=[7], int_props=[0.5])
    with pytest.raises(ValueError):
        g.negative_sampler(EDGE2, 6, "random", conditional=True, int_cols=[0, 1], int_props=[0.7, 0.7])


@pytest.mark.parametrize("strategy,share", [("in_degree", False), ("random", False), ("node_weight", True)])
def test_conditional_negative_sampling_reference_test_case(gl, g, strategy, share):
    """python/sampler/tests/test_conditional_negative_sampling.py restated: cond_item nodes carry ints [id % 5, id % 4],
    a float and the string str(id % 3); cond_sim = i -> i + 2, i + 3, i + 5.  Of 4 negatives per pair, slot 0 shares the
    dst's first int attribute, slot 1 its second, slots 2-3 its string; none is a neighbour of the src (edge-type
    strategies) / a dst of the batch (batch_share)."""
    src_ids = np.array([1, 2, 3, 4, 5])
    dst_ids = np.array([12, 34, 2, 67, 128])
    object_type = "cond_item" if strategy == "node_weight" else "cond_sim"
    ns = g.negative_sampler(object_type, expand_factor=4, strategy=strategy, conditional=True, unique=False,
                            batch_share=share, int_cols=[0, 1], int_props=[0.25, 0.25], str_cols=[0], str_props=[0.5])
    for cc in range(20):  # the reference draws once; here 20 pinned streams
        ns.set_call_counter(cc)
        nodes = ns.get(src_ids, dst_ids)
        assert nodes.ids.shape == (5, 4)
        for idx, sid in enumerate(src_ids):
            neg, pos = nodes.ids[idx], dst_ids[idx]
            if share:
                assert set(neg.tolist()).isdisjoint(set(dst_ids.tolist()))
            else:
                assert set(neg.tolist()).isdisjoint({sid + 2, sid + 3, sid + 5})
            assert neg[0] % 5 == pos % 5 and neg[1] % 4 == pos % 4 and neg[2] % 3 == pos % 3 and neg[3] % 3 == pos % 3
        np.testing.assert_almost_equal(nodes.weights, nodes.ids * 0.1, decimal=4)  # the negatives come back as Nodes


def test_subgraph_sampling_reference_test_case(gl, g):
    """python/sampler/tests/test_subgraph_sampling.py restated: batches of 8 entity nodes in order, the sub-graph the
    relation edges induce among them; labels, float attributes and -- entry for entry, in order -- the edge index the
    reference's check_subgraph_edge_indices expects."""
    node_sampler = g.node_sampler("entity", batch_size=8)
    subgraph_sampler = g.subgraph_sampler(nbr_type="relation")
    batches = 0
    while True:
        try:
            nodes = node_sampler.get()
        except gl.OutOfRangeError:
            break
        sub = subgraph_sampler.get(nodes.ids)
        batches += 1
        ids = sub.nodes.ids
        np.testing.assert_equal(sub.nodes.labels, ids)
        for j in range(4):
            np.testing.assert_almost_equal(sub.nodes.float_attrs[:, j], ids * 0.1 * (j + 1), decimal=4)
        rows, cols = [], []
        for i in range(ids.size):
            for j in range(ids.size):
                if (ids[i] < 100 or ids[j] < 100) and abs(int(ids[i]) - int(ids[j])) in (2, 3, 5):
                    rows += [i, j]
                    cols += [j, i]
        np.testing.assert_equal(sub.edge_index[0], np.array(rows, dtype=np.int32))
        np.testing.assert_equal(sub.edge_index[1], np.array(cols, dtype=np.int32))
    assert batches == 15  # 120 entity nodes


def test_in_and_out_degree_lookups(gl, g):
    """Graph.out_degrees / in_degrees (GetDegree with NodeFrom EDGE_SRC / EDGE_DST) against the generator."""
    ids = np.array([102, 105, 107, 199, 5000])
    np.testing.assert_equal(g.out_degrees(ids, EDGE2), [2, 0, 2, 4, 0])
    dsts = np.arange(0, 100)
    want = np.zeros(100, np.int64)
    for s in range(*RANGE2):
        for d_ in fx.fixed_dst_ids(s, RANGE1):
            want[d_] += 1
    np.testing.assert_equal(g.in_degrees(dsts, EDGE2), want)
    np.testing.assert_equal(g.in_degrees(np.array([100, -7, 10 ** 9]), EDGE2), [0, 0, 0])
    nodes = g.get_nodes(NODE1, dsts)
    np.testing.assert_equal(nodes.get_in_degrees(EDGE2), want)


def test_empty_and_degenerate_sources(gl, g, tmp_path):
    """Header-only files, CRLF line endings, blank lines, a missing trailing newline, duplicate node ids
    and malformed records (skipped with ignore_invalid, fatal without)."""
    d = str(tmp_path)
    open(os.path.join(d, "e0"), "w").write("src_id:int64\tdst_id:int64\tweight:float\n")
    open(os.path.join(d, "n0"), "w").write("id:int64\tfeature:string\n")
    open(os.path.join(d, "e1"), "w").write("src_id:int64\tdst_id:int64\tweight:float\r\n1\t2\t0.5\r\n\r\n1\t3\t0.25\r\nbad\tline\there\r\n2\t3\t1.0")
    open(os.path.join(d, "n1"), "w").write("id:int64\tfeature:string\n1\t0.5:1.5\n2\t2.5:3.5\n1\t9.0:9.0\n3\tnot:a_number\n")
    g = gl.Graph().edge(os.path.join(d, "e0"), ("a", "a", "empty"), gl.Decoder(weighted=True)) \
        .node(os.path.join(d, "n0"), "a", gl.Decoder(attr_types=["float", "float"])) \
        .edge(os.path.join(d, "e1"), ("b", "b", "some"), gl.Decoder(weighted=True)) \
        .node(os.path.join(d, "n1"), "b", gl.Decoder(attr_types=["float", "float"]))
    g.init()
    try:
        gl.set_padding_mode(gl.CIRCULAR)
        ids = np.array([1, 2, 3])
        nb = g.neighbor_sampler("empty", 2, strategy="random").get(ids)
        assert (nb.layer_nodes(1).ids == -1).all()  # default neighbour id set by the module fixture
        top = g.neighbor_sampler("some", 2, strategy="topk").get(ids)
        np.testing.assert_equal(top.layer_nodes(1).ids, [[2, 3], [3, 3], [-1, -1]])
        np.testing.assert_equal(top.layer_edges(1).edge_ids, [[0, 1], [2, 2], [-1, -1]])  # the bad record took no id
        vals = g.lookup_nodes("b", np.array([1, 2, 3]))
        np.testing.assert_allclose(vals.float_attrs, [[0.5, 1.5], [2.5, 3.5], [999.9, 999.9]], rtol=1e-6)
        assert g.get_nodes("a", np.array([7])).float_attrs.tolist() == [[pytest.approx(999.9), pytest.approx(999.9)]]
    finally:
        gl.set_padding_mode(gl.REPLICATE)
        g.close()
    gl.set_ignore_invalid(False)
    try:
        strict = gl.Graph().edge(os.path.join(d, "e1"), ("b", "b", "some"), gl.Decoder(weighted=True))
        with pytest.raises(gl.InvalidArgumentError):
            strict.init()
        strict.close()
    finally:
        gl.set_ignore_invalid(True)


def test_module_graph_still_served_after_other_graphs_closed(gl, g):
    """Other Graph objects came and went in this process; the operators are bound to this one again."""
    nbrs = g.neighbor_sampler(EDGE1, 3, strategy="random").get(SEEDS1)
    fx.expect_edges_follow_generator(nbrs.layer_edges(1), RANGE2, SEEDS1, DEFAULT_ID)


def test_quickstart_example_runs():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "quickstart.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "one epoch: 4 batches" in r.stdout


def test_timestamped_source_through_the_loader(gl, g, tmp_path):
    """A timestamped TSV source (`timestamp:int64` column, Decoder(timestamped=True)): rows come out in
    timestamp order like the reference's Build(), and the edges' timestamps can be looked up."""
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "timestamped.npz")))
    path = os.path.join(str(tmp_path), "ts_edges")
    with open(path, "w") as f:
        f.write("src_id:int64\tdst_id:int64\tweight:float\ttimestamp:int64\n")
        for s, d_, w, t in zip(gold["src"], gold["dst"], gold["w"], gold["ts"]):
            f.write("%d\t%d\t%.9g\t%d\n" % (s, d_, w, t))
    tg = gl.Graph().edge(path, ("a", "a", "ts"), gl.Decoder(weighted=True, timestamped=True)).init()
    try:
        full = tg.neighbor_sampler("ts", 0, strategy="full").get(gold["rows"])
        np.testing.assert_equal(full.layer_nodes(1).ids, gold["col"])
        edges = full.layer_edges(1)
        np.testing.assert_equal(edges.edge_ids, gold["eid"])
        np.testing.assert_equal(edges.timestamps, gold["ts"][gold["eid"]])
        np.testing.assert_equal(edges.weights.view(np.uint32), gold["w_slot"].view(np.uint32))
    finally:
        tg.close()


def test_filtered_sampling_through_the_python_api(gl, g, tmp_path):
    """op::Filter end to end (TSV -> loader -> GraphStore -> SamplingRequest with filter values -> HIP): the
    reference's own Topk / Full answers of tests/golden/filtered.npz, plus the seed's value reaching hop 2."""
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "filtered.npz")))
    path = os.path.join(str(tmp_path), "flt_edges")
    with open(path, "w") as f:
        f.write("src_id:int64\tdst_id:int64\tweight:float\ttimestamp:int64\n")
        for s, d_, w, t in zip(gold["src"], gold["dst"], gold["w"], gold["ts"]):
            f.write("%d\t%d\t%.9g\t%d\n" % (s, d_, w, t))
    names = {"id_eq": ("equal", "id"), "id_gt": ("larger_than", "id"), "ts_eq": ("equal", "timestamp"),
             "ts_gt": ("larger_than", "timestamp")}
    gl.set_default_neighbor_id(-7)
    tg = gl.Graph().edge(path, ("a", "a", "flt"), gl.Decoder(weighted=True, timestamped=True)).init()
    try:
        for case in gold["cases"]:
            case = str(case)
            kind, strategy = case.rsplit("_", 3)[0], case.split("_")[2]
            k, pad = int(case.split("_k")[1][0]), int(case[-1])
            gl.set_padding_mode(gl.CIRCULAR if pad else gl.REPLICATE)
            sampler = tg.neighbor_sampler("flt", k, strategy="topk" if strategy == "TopkSampler" else "full")
            layer = sampler.set_filter(*names[kind]).get(gold[case + "_ids"], filter_values=gold[case + "_values"])
            np.testing.assert_equal(layer.layer_nodes(1).ids.reshape(-1), gold[case + "_nbr"].reshape(-1), err_msg=case)
            np.testing.assert_equal(layer.layer_edges(1).edge_ids.reshape(-1), gold[case + "_eid"].reshape(-1))
        gl.set_padding_mode(gl.CIRCULAR)
        # two hops: every descendant of a seed is filtered with the seed's value
        gl.set_sampler_retry_times(80)
        seeds = gold["rows"][5:25]
        values = np.full(seeds.shape[0], 905, np.int64)
        two = tg.neighbor_sampler(["flt", "flt"], [3, 4], strategy="random").set_filter("equal", "id")
        layers = two.get(seeds, filter_values=values)
        for hop in (1, 2):
            ids = layers.layer_nodes(hop).ids
            assert ids.shape == (seeds.size * (1 if hop == 1 else 3), 3 if hop == 1 else 4)
            assert not (ids == 905).any()
        with pytest.raises(ValueError):
            two.get(seeds)
        with pytest.raises(ValueError):
            tg.neighbor_sampler("flt", 2).get(seeds, filter_values=values)
    finally:
        gl.set_sampler_retry_times(5)
        gl.set_default_neighbor_id(DEFAULT_ID)
        gl.set_padding_mode(gl.REPLICATE)
        tg.close()


def test_random_walk_through_the_python_api(gl, g):
    """RandomWalk operator (random_walk.cc) behind Graph.random_walk: numpy ids go through the registered operator,
    CUDA ids straight to the device graph; the same pinned stream gives the same walks; node2vec biases are served."""
    import torch
    seeds = np.arange(100, 160, dtype=np.int64)
    for p, q in ((1.0, 1.0), (0.25, 4.0)):
        a = g.random_walk(EDGE3, seeds, 5, p=p, q=q, call_counter=7)
        b = g.random_walk(EDGE3, torch.from_numpy(seeds).cuda(), 5, p=p, q=q, call_counter=7).cpu().numpy()
        assert a.shape == (60, 5) and np.array_equal(a, b)
        for row, s in zip(a, seeds):  # every step follows an (undirected) edge3 edge, or is the default id at a dead end
            cur = int(s)
            for v in row:
                out = set(fx.fixed_dst_ids(cur, RANGE2)) | {x for x in range(*RANGE2) if cur in fx.fixed_dst_ids(x, RANGE2)}
                assert (int(v) in out) if (out and cur != DEFAULT_ID) else (v == DEFAULT_ID), (cur, v)
                cur = int(v)
    assert not np.array_equal(g.random_walk(EDGE3, seeds, 5), g.random_walk(EDGE3, seeds, 5))  # fresh streams per call

"""graphlearn/python/nn/pytorch/data/test/test_dataset.py restated against the glx engine: a GSL query with nested
each() branches (edge source, end points, two-hop neighbourhoods, negatives and their neighbourhoods) read through
`graphlearn.python.nn.pytorch.Dataset` -- {alias: Data of torch tensors} -- and through torch's DataLoader after
as_dict(); plus what the reference's nn.Dataset promises about masks, edge and full-neighbour (sparse) steps."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))

pytestmark = pytest.mark.gpu

BATCH, HOP0, HOP1, NEG = 20, 5, 2, 3


@pytest.fixture(scope="module")
def gl():
    import graphlearn
    return graphlearn


@pytest.fixture(scope="module")
def graph(gl, tmp_path_factory):
    d = tmp_path_factory.mktemp("gl_nn")
    rnd = random.Random(7)
    user, item, u2i, i2i = (str(d / n) for n in ("user", "item", "u2i", "i2i"))
    with open(user, "w") as f:  # id 0..99, attributes [float] * 4
        f.write("id:int64\tfeature:string\n")
        for i in range(100):
            f.write("%d\t%f:%f:%f:%f\n" % (i, i * 0.1, i * 0.2, i * 0.3, i * 0.4))
    with open(item, "w") as f:  # id 0..99, attributes [float, string, string]
        f.write("id:int64\tfeature:string\n")
        for i in range(100):
            f.write("%d\t%f:%s:%s\n" % (i, i * 0.1, "hello", str(i)))
    for path in (u2i, i2i):  # 3 out-edges per source
        with open(path, "w") as f:
            f.write("sid:int64\tdid:int64\n")
            for i in range(100):
                for _ in range(3):
                    f.write("%d\t%d\n" % (i, rnd.randint(0, 99)))
    g = gl.Graph() \
        .node(user, "u", decoder=gl.Decoder(attr_types=["float"] * 4, attr_dims=[None] * 4)) \
        .node(item, "i", decoder=gl.Decoder(attr_types=["float", ("string", 100), ("string", 50)],
                                            attr_dims=[None, 20, 10])) \
        .edge(u2i, ("u", "i", "u-i"), decoder=gl.Decoder()) \
        .edge(i2i, ("i", "i", "i-i"), decoder=gl.Decoder()) \
        .init()
    yield g
    g.close()


def _query(graph):
    return graph.E("u-i").batch(BATCH).alias("seed").each(lambda e: (
        e.inV().alias("i").outV("i-i").sample(HOP0).by("random").alias("dst_hop1")
         .outV("i-i").sample(HOP1).by("random").alias("dst_hop2"),
        e.outV().alias("u").each(lambda v: (
            v.outV("u-i").sample(HOP0).by("random").alias("src_hop1")
             .outV("i-i").sample(HOP1).by("random").alias("src_hop2"),
            v.outV("u-i").sample(HOP0).by("random").alias("extra_hop1")
             .outV("i-i").sample(HOP1).by("random").alias("extra_hop2"),
            v.outNeg("u-i").sample(NEG).by("random").alias("neg")
             .outV("i-i").sample(HOP0).by("random").alias("neg_hop1")
             .outV("i-i").sample(HOP1).by("random").alias("neg_hop2"))))) \
        .values()


def test_get_th_data(gl, graph):
    import torch
    import graphlearn.python.nn.pytorch as thg  # the reference's import path
    ds = thg.Dataset(_query(graph))
    batches = 0
    for data in ds:
        neg, src_hop1 = data.get("neg"), data.get("src_hop1")
        assert isinstance(neg.ids, torch.Tensor) and neg.ids.dtype == torch.int64
        assert list(neg.ids.shape) == [BATCH * NEG]
        assert list(src_hop1.ids.shape) == [BATCH * HOP0]
        assert list(src_hop1.float_attrs.shape) == [BATCH * HOP0, 1]
        assert list(data.get("dst_hop2").int_attrs.shape) == [BATCH * HOP0 * HOP1, 2]
        assert src_hop1.string_attrs is None and src_hop1.labels is None  # the item decoder has neither
        seed = data.get("seed")  # an edge step: ids = sources, dst_ids = destinations
        assert list(seed.ids.shape) == [BATCH] and list(seed.dst_ids.shape) == [BATCH]
        assert torch.equal(seed.ids, data.get("u").ids) and torch.equal(seed.dst_ids, data.get("i").ids)
        # float attribute 0 of item k is k * 0.1; the hashed strings are buckets
        ids = src_hop1.ids.numpy()
        np.testing.assert_allclose(src_hop1.float_attrs.numpy()[:, 0], np.where(ids >= 0, ids * 0.1, 0.0), rtol=1e-6)
        ints = data.get("dst_hop2").int_attrs.numpy()
        assert ints[:, 0].max() < 100 and ints[:, 1].max() < 50
        assert list(data.get("u").float_attrs.shape) == [BATCH, 4]
        batches += 1
    assert batches == 300 // BATCH  # one epoch over the 300 u-i edges

    loader = torch.utils.data.DataLoader(ds.as_dict())
    batches = 0
    for data in loader:
        assert list(data["neg"]["ids"].shape) == [1, BATCH * NEG]
        assert list(data["src_hop1"]["ids"].shape) == [1, BATCH * HOP0]
        assert list(data["src_hop1"]["float_attrs"].shape) == [1, BATCH * HOP0, 1]
        assert list(data["dst_hop2"]["int_attrs"].shape) == [1, BATCH * HOP0 * HOP1, 2]
        batches += 1
    assert batches == 300 // BATCH


def test_numpy_dataset_masks_edges_and_sparse_steps(gl, graph):
    from graphlearn.python.nn.dataset import Dataset
    q = graph.V("u").batch(8).alias("a") \
             .outE("u-i").sample(2).by("random").alias("e") \
             .inV().alias("b") \
             .outV("i-i").sample(0).by("full").alias("full") \
             .values()
    ds = Dataset(q)
    feat, ids, sparse = ds.masks["a"]
    assert feat == [False, True, False, False, False, False] and ids == [True, False] and sparse == [False] * 3
    assert ds.masks["e"][1] == [True, True]  # src_ids + dst_ids
    assert ds.masks["b"][0] == [True, True, False, False, False, False]
    assert ds.masks["full"][2] == [True, True, True]
    data = ds.get_data_dict()
    assert data["a"].ids.shape == (8,) and data["a"].float_attrs.shape == (8, 4) and data["a"].int_attrs is None
    assert data["e"].ids.shape == (16,) and data["e"].dst_ids.shape == (16,)
    np.testing.assert_equal(data["e"].ids, np.repeat(data["a"].ids, 2))
    np.testing.assert_equal(data["b"].ids, data["e"].dst_ids)
    full = data["full"]  # ragged: offsets[i] neighbours of b's i-th vertex
    assert len(full.offsets) == 16 and int(np.sum(full.offsets)) == full.ids.shape[0]
    assert full.indices.shape == (full.ids.shape[0], 2) and full.int_attrs.shape == (full.ids.shape[0], 2)
    assert all(int(o) == 3 or b < 0 for o, b in zip(full.offsets, data["b"].ids))  # every item has 3 out-edges
    n = 1
    for _ in ds:  # the iterator stops at the end of the epoch
        n += 1
    assert n == (100 + 7) // 8


def test_tensors_on_the_gpu(gl, graph):
    import torch
    import graphlearn.nn.pytorch as thg
    q = graph.V("i").batch(16).alias("seed").outV("i-i").sample(4).by("random").alias("hop").values()
    it = iter(thg.Dataset(q, device="cuda"))
    data = next(it)
    assert data["hop"].ids.is_cuda and data["hop"].float_attrs.is_cuda and list(data["hop"].ids.shape) == [64]
    # a chain of two dense hops is ONE engine call whose values never visit the host; every column still arrives
    q2 = graph.V("u").batch(16).alias("seed") \
              .outV("u-i").sample(4).by("random").alias("h1") \
              .outV("i-i").sample(3).by("random").alias("h2").values()
    data = next(iter(thg.Dataset(q2, device="cuda")))
    h1, h2 = data["h1"], data["h2"]
    assert h1.ids.is_cuda and list(h1.ids.shape) == [64] and list(h2.ids.shape) == [192]
    assert h2.float_attrs.is_cuda and list(h2.float_attrs.shape) == [192, 1]
    assert h2.int_attrs.is_cuda and list(h2.int_attrs.shape) == [192, 2]  # host-resident column, moved over
    torch.testing.assert_close(h2.float_attrs[:, 0], (h2.ids.clamp(min=0) * 0.1).float(), rtol=1e-6, atol=0)
    # hop 1 really holds neighbours of the seeds: compare with the ordinary (host) path's adjacency
    full = graph.neighbor_sampler("u-i", 0, strategy="full").get(data["seed"].ids.cpu().numpy()).layer_nodes(1)
    rows = np.split(full.ids, np.cumsum(full.offsets)[:-1])
    for r, got in zip(rows, h1.ids.cpu().numpy().reshape(16, 4)):
        assert set(got.tolist()) <= set(r.tolist())
    with pytest.raises(NotImplementedError):
        thg.Dataset(q, graph=graph)


def test_prefetching_dataset_yields_whole_epochs(gl, graph):
    """prefetch=True: batches are sampled ahead on a background thread (the reference's `window`); every epoch still
    visits every seed exactly once, tensors arrive complete on the GPU, and the iterator ends with the epoch."""
    import torch
    import graphlearn.nn.pytorch as thg
    q = graph.V("u").batch(16).alias("seed") \
             .outV("u-i").sample(4).by("random").alias("h1") \
             .outV("i-i").sample(3).by("random").alias("h2").values()
    ds = thg.Dataset(q, window=3, device="cuda", prefetch=True)
    for _ in ds:  # the by-order cursor of a node type is the operator's, shared with the tests above: finish its epoch
        pass
    for epoch in range(2):
        seen = []
        for data in ds:
            assert data["h2"].float_attrs.is_cuda and list(data["h2"].ids.shape) == [data["seed"].ids.shape[0] * 12]
            torch.testing.assert_close(data["h2"].float_attrs[:, 0], (data["h2"].ids.clamp(min=0) * 0.1).float(),
                                       rtol=1e-6, atol=0)
            seen.extend(data["seed"].ids.cpu().tolist())
        assert sorted(seen) == list(range(100)), epoch
    ds.close()


def test_training_example_learns_from_the_sampled_neighbourhoods():
    """examples/train_sage_pytorch.py in a process of its own: a vertex of that graph is hard to classify from its own
    16 noisy features and easy from its neighbourhood's, so two epochs well above what the features alone give (and far
    above the 0.2 of chance) mean the sampled two-hop neighbourhoods, their attributes and the labels line up."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_sage_pytorch.py"), "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("epoch ")]
    assert len(lines) == 2, r.stdout[-2000:]
    acc = [float(ln.split("accuracy ")[1].split(",")[0]) for ln in lines]
    loss = [float(ln.split("loss ")[1].split(",")[0]) for ln in lines]
    assert acc[1] > 0.7 and acc[1] > acc[0] and loss[1] < loss[0], lines

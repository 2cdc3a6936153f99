"""GPU parity tests (run with `-m gpu` on an MI355X through gpurun).

Everything goes through the C-ABI (include/glx.h) via the ctypes harness and is
compared BIT-EXACTLY with the oracle (oracle/glx_oracle.c, itself pinned against
the reference by test_oracle_golden.py) and with the golden fixtures generated
from the real reference.  Aggregation is bit-exact too (tolerance 0; the
north-star allowance is 1e-5 relative) because glx accumulates every output
element in the reference's left-to-right order.
"""
import os

import numpy as np
import pytest

import glx
import synth
from oracle_bindings import AGGREGATORS, SAMPLERS, Oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def beq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert glx.device_count() >= 1, "GPU tests need a HIP device; glx has no CPU fallback"


# ------------------------------------------------------------ golden fixtures ---
def test_golden_kat_topk():
    g = load("kat_sampler.npz")
    dev = glx.Graph(g["row_ptr"], g["col"], g["eid"], g["w_slot"], ids=g["rows"])
    nbr, _ = dev.sample("TopkSampler", np.array([0, 1], np.int64), 2)
    assert nbr.reshape(-1).tolist() == [20, 10, 21, 11]  # sampler_unittest.cpp:190-195
    for pad in (0, 1):
        for dflt in (0, -1):
            for k in (2, 4):
                nbr, eid = dev.sample("TopkSampler", g["query"], k, padding_mode=pad, default_neighbor_id=dflt)
                key = "topk_p%d_d%d_k%d" % (pad, dflt + 1, k)
                assert np.array_equal(nbr, g[key + "_nbr"]) and np.array_equal(eid, g[key + "_eid"]), key


def test_golden_python_fixture_topk():
    g = load("pyfixture_topk.npz")
    dev = glx.Graph(g["row_ptr"], g["col"], g["eid"], g["w_slot"], ids=g["rows"])
    for pad in (0, 1):
        nbr, eid = dev.sample("TopkSampler", g["query"], 6, padding_mode=pad, default_neighbor_id=-1)
        assert np.array_equal(nbr, g["topk_p%d_nbr" % pad]) and np.array_equal(eid, g["topk_p%d_eid" % pad])
    nbr, eid = dev.sample("RandomWithoutReplacementSampler", g["query"], 6, padding_mode=0,
                          default_neighbor_id=-1)
    assert np.array_equal(nbr, g["rwor_p0_nbr"]) and np.array_equal(eid, g["rwor_p0_eid"])


def test_golden_rand_graph_alias_and_topk():
    g = load("rand_graph.npz")
    dev = glx.Graph(g["row_ptr"], g["col"], g["eid"], g["w_slot"], ids=g["rows"])
    prob, alias = dev.export_alias()
    assert beq(prob, g["alias_prob"]) and np.array_equal(alias, g["alias_idx"])  # alias_method.cc:57-107
    for pad in (0, 1):
        for k in (1, 3, 10, 33, 70):
            nbr, eid = dev.sample("TopkSampler", g["query"], k, padding_mode=pad, default_neighbor_id=-7)
            assert np.array_equal(nbr, g["topk_p%d_k%d_nbr" % (pad, k)])
            assert np.array_equal(eid, g["topk_p%d_k%d_eid" % (pad, k)])
    for k in (3, 33):
        nbr, eid = dev.sample("RandomWithoutReplacementSampler", g["query"], k, padding_mode=0,
                              default_neighbor_id=-7)
        assert np.array_equal(nbr, g["rwor_p0_k%d_nbr" % k]) and np.array_equal(eid, g["rwor_p0_k%d_eid" % k])


def test_golden_aggregators():
    a = load("agg.npz")
    f = glx.Features(np.arange(100, dtype=np.float32).reshape(100, 1).copy())
    for name in AGGREGATORS:
        emb, cnt = f.aggregate(name, a["kat_ids"], a["kat_seg"], 5)
        assert beq(emb, a["kat_%s_emb" % name]) and np.array_equal(cnt, a["kat_%s_cnt" % name]), name
    for c in range(int(a["num_cases"])):
        f = glx.Features(a["c%d_X" % c], ids=a["c%d_raw" % c])
        for name in AGGREGATORS:
            emb, cnt = f.aggregate(name, a["c%d_ids" % c], a["c%d_seg" % c], int(a["c%d_num_segments" % c]),
                                   default_attr=float(a["c%d_default" % c]))
            assert np.array_equal(cnt, a["c%d_%s_cnt" % (c, name)]), (c, name)
            assert beq(emb, a["c%d_%s_emb" % (c, name)]), (c, name)


# ------------------------------------------------------- seeded vs the oracle ---
def _queries(rng, V, n, extra):
    q = rng.integers(0, V, n).astype(np.int64)
    return np.concatenate([q, np.asarray(extra, np.int64)])


@pytest.fixture(scope="module")
def graphs(orc):
    out = {}
    rp, col, eid, w = synth.small_graph(3000, 40000, seed=3, weighted=True, hub_degree=5000)
    out["dense"] = (dict(row_ptr=rp, col=col, eid=eid, weight=w, alias=orc.alias_build(rp, w)),
                    glx.Graph(rp, col, eid, w))
    raw = np.arange(3000, dtype=np.int64) * 11 - 7000
    rng = np.random.default_rng(8)
    raw = raw[rng.permutation(3000)]
    out["hashed"] = (dict(row_ptr=rp, col=raw[col], eid=eid, weight=w, alias=out["dense"][0]["alias"], ids=raw),
                     glx.Graph(rp, raw[col], eid, w, ids=raw))
    rp2, col2, eid2, _ = synth.small_graph(500, 3000, seed=4, weighted=False)
    out["unweighted"] = (dict(row_ptr=rp2, col=col2, eid=eid2), glx.Graph(rp2, col2, eid2))
    return out


KS = [1, 2, 3, 5, 8, 9, 16, 17, 25, 32, 33, 64, 65, 100, 300]


@pytest.mark.parametrize("which", ["dense", "hashed"])
@pytest.mark.parametrize("name", SAMPLERS)
def test_samplers_bit_exact(orc, graphs, which, name):
    og, dev = graphs[which]
    rng = np.random.default_rng(17)
    if which == "dense":
        q = _queries(rng, 3000, 700, [0, 0, -1, 3000, 10 ** 12])
    else:
        q = np.concatenate([og["ids"][rng.integers(0, 3000, 700)], og["ids"][:1], [5, -5, 1 << 40]]).astype(np.int64)
    cc = 0
    for pad in (1, 0):
        for k in KS:
            cc += 1
            nbr, eid = dev.sample(name, q, k, seed=0xabcdef12345, call_counter=cc, padding_mode=pad,
                                  default_neighbor_id=-3)
            on, oe = orc.sample(og, name, q, k, seed=0xabcdef12345, call_counter=cc, padding_mode=pad,
                                default_neighbor_id=-3)
            assert np.array_equal(nbr, on), (name, which, pad, k)
            assert np.array_equal(eid, oe), (name, which, pad, k)


def test_unweighted_graph(orc, graphs):
    og, dev = graphs["unweighted"]
    q = np.arange(500, dtype=np.int64)
    for name in ("RandomSampler", "RandomWithoutReplacementSampler", "TopkSampler"):
        nbr, eid = dev.sample(name, q, 7, seed=5, call_counter=9)
        on, oe = orc.sample(og, name, q, 7, seed=5, call_counter=9)
        assert np.array_equal(nbr, on) and np.array_equal(eid, oe)
    with pytest.raises(glx.GlxError) as e:
        dev.sample("EdgeWeightSampler", q, 7)
    assert e.value.code == 3


def test_seed_and_counter_select_the_stream(graphs):
    _, dev = graphs["dense"]
    q = np.arange(1000, dtype=np.int64)
    a = dev.sample("RandomSampler", q, 10, seed=1, call_counter=1)[1]
    b = dev.sample("RandomSampler", q, 10, seed=1, call_counter=1)[1]
    c = dev.sample("RandomSampler", q, 10, seed=1, call_counter=2)[1]
    d = dev.sample("RandomSampler", q, 10, seed=2, call_counter=1)[1]
    assert np.array_equal(a, b) and not np.array_equal(a, c) and not np.array_equal(a, d)


def test_device_pointers_match_host_pointers(graphs):
    import torch
    _, dev = graphs["dense"]
    q = np.random.default_rng(2).integers(0, 3000, 4096).astype(np.int64)
    tq = torch.from_numpy(q).cuda()
    for name in SAMPLERS:
        hn, he = dev.sample(name, q, 25, seed=3, call_counter=4)
        dn, de = dev.sample(name, tq, 25, seed=3, call_counter=4)
        torch.cuda.synchronize()
        assert np.array_equal(hn, dn.cpu().numpy()) and np.array_equal(he, de.cpu().numpy())
    deg_h = dev.degrees(q)
    deg_d = dev.degrees(tq)
    assert np.array_equal(deg_h, deg_d.cpu().numpy())


def test_empty_and_degenerate_requests(orc, graphs):
    og, dev = graphs["dense"]
    n, e = dev.sample("RandomSampler", np.zeros(0, np.int64), 5)
    assert n.shape == (0, 5)
    n, e = dev.sample("TopkSampler", np.arange(4, dtype=np.int64), 0)
    assert n.shape == (4, 0)
    with pytest.raises(glx.GlxError):
        dev.sample(17, np.arange(4, dtype=np.int64), 2)
    # a graph without any edge
    rp = np.zeros(11, np.int64)
    g0 = glx.Graph(rp, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32))
    for name in SAMPLERS:
        n, e = g0.sample(name, np.arange(12, dtype=np.int64), 3, default_neighbor_id=42)
        assert (n == 42).all() and (e == -1).all()
    with pytest.raises(glx.GlxError):
        glx.Graph(np.array([0, 2, 1], np.int64), np.zeros(1, np.int64), np.zeros(1, np.int64))


@pytest.mark.parametrize("D", [1, 2, 3, 4, 7, 8, 16, 32, 64, 100, 128, 256, 260, 512, 1024])
def test_aggregators_bit_exact(orc, D):
    rng = np.random.default_rng(D)
    V = 1500
    X = (rng.standard_normal((V, D)) * 4).astype(np.float32)
    X[rng.random((V, D)) < 0.02] = -50.0  # exercise Max's -37 initialiser
    raw = (np.arange(V, dtype=np.int64) * 3 + 17)
    Sg = 301
    sizes = rng.integers(0, 12, Sg)
    sizes[[0, 1, 100, Sg - 1]] = 0
    sizes[7] = 700
    seg = np.repeat(np.arange(Sg, dtype=np.int32), sizes)
    for ids_kind in ("dense", "hashed"):
        f = glx.Features(X, ids=(raw if ids_kind == "hashed" else None))
        pool = raw if ids_kind == "hashed" else np.arange(V, dtype=np.int64)
        nid = pool[rng.integers(0, V, seg.shape[0])].copy()
        nid[rng.random(seg.shape[0]) < 0.05] = -99  # unknown -> default row
        for name in AGGREGATORS:
            emb, cnt = f.aggregate(name, nid, seg, Sg, default_attr=2.5)
            oemb, ocnt = orc.aggregate(X, name, nid, seg, Sg, 2.5, ids=(raw if ids_kind == "hashed" else None))
            assert np.array_equal(cnt, ocnt), (name, D, ids_kind)
            assert beq(emb, oemb), (name, D, ids_kind)


def test_aggregator_cursor_semantics_and_edges(orc):
    """aggregating_request.cc:86-105: an out-of-order / out-of-range segment id stalls the cursor."""
    rng = np.random.default_rng(0)
    X = rng.standard_normal((50, 8)).astype(np.float32)
    f = glx.Features(X)
    cases = [
        (np.array([0, 0, 1, 3, 2, 3], np.int32), 5),      # 2 after 3: tail dropped
        (np.array([1, 1, 0], np.int32), 3),
        (np.array([0, 1, 9, 2], np.int32), 4),            # 9 >= num_segments
        (np.array([-1, 0, 1], np.int32), 3),
        (np.zeros(0, np.int32), 4),                       # no ids at all
        (np.array([2, 2, 2], np.int32), 3),
    ]
    for seg, Sg in cases:
        nid = rng.integers(0, 50, seg.shape[0]).astype(np.int64)
        for name in AGGREGATORS:
            emb, cnt = f.aggregate(name, nid, seg, Sg, default_attr=-1.5)
            oemb, ocnt = orc.aggregate(X, name, nid, seg, Sg, -1.5)
            assert np.array_equal(cnt, ocnt) and beq(emb, oemb), (seg, name)
    e, c = f.aggregate("SumAggregator", np.zeros(0, np.int64), np.zeros(0, np.int32), 0)
    assert e.shape == (0, 8)


def test_aggregate_device_pointers_and_lookup(orc):
    import torch
    rng = np.random.default_rng(1)
    V, D = 5000, 128
    X = rng.standard_normal((V, D)).astype(np.float32)
    f = glx.Features(torch.from_numpy(X).cuda())
    nid = rng.integers(-5, V + 5, 40000).astype(np.int64)
    seg = (np.arange(40000) // 10).astype(np.int32)
    for name in AGGREGATORS:
        emb, cnt = f.aggregate(name, torch.from_numpy(nid).cuda(), torch.from_numpy(seg).cuda(), 4000)
        torch.cuda.synchronize()
        oemb, ocnt = orc.aggregate(X, name, nid, seg, 4000)
        assert np.array_equal(cnt.cpu().numpy(), ocnt) and beq(emb.cpu().numpy(), oemb), name
    out = f.lookup(nid[:1000], default_attr=9.0)
    exp = np.where(((nid[:1000] >= 0) & (nid[:1000] < V))[:, None], X[np.clip(nid[:1000], 0, V - 1)], 9.0)
    assert beq(out, exp.astype(np.float32))


@pytest.mark.parametrize("P", [1, 2, 3, 8, 64])
def test_partition_and_stitch(orc, P):
    import torch
    rng = np.random.default_rng(P)
    # up to 32768 histogram cells (buckets x 2048-id tiles) the scatter kernel scans them itself; beyond that a scan
    # kernel runs in between (P = 64 at n = 2.2 M, P = 3 at n = 23 M)
    for n in (0, 1, 255, 2048, 2049, 100003) + ((2_200_000,) if P == 64 else (23_000_000,) if P == 3 else ()):
        ids = rng.integers(-10 ** 9, 10 ** 9, n).astype(np.int64)
        t = torch.from_numpy(ids).cuda()
        bucketed, order, counts = glx.partition(t, P)
        torch.cuda.synchronize()
        oorder, ocounts = orc.partition(ids, P)
        assert np.array_equal(counts.cpu().numpy(), ocounts), (P, n)
        if n == 0:
            continue
        assert np.array_equal(order.cpu().numpy(), oorder), (P, n)
        assert np.array_equal(bucketed.cpu().numpy(), ids[oorder])
        rows = torch.stack([bucketed * 2, bucketed * 2 + 1], 1).contiguous()
        back = glx.stitch(rows, order)
        assert np.array_equal(back.cpu().numpy(), np.stack([ids * 2, ids * 2 + 1], 1))
        fr = glx.stitch(rows.to(torch.float32), order)
        assert np.array_equal(fr.cpu().numpy(), np.stack([ids * 2, ids * 2 + 1], 1).astype(np.float32))


def test_size_limits_are_rejected_not_truncated(graphs):
    """Tensor sizes are int32 in the reference (tensor.h:47): batch*k and
    num_segments*dim beyond INT32_MAX are InvalidArgument, checked before any work."""
    _, dev = graphs["dense"]
    src = np.zeros(1 << 20, np.int64)
    with pytest.raises(glx.GlxError) as e:
        dev.sample("TopkSampler", src, 4096, out=(np.zeros((1, 1), np.int64), np.zeros((1, 1), np.int64)))
    assert e.value.code == 3 and "int32" in str(e.value)
    f = glx.Features(np.zeros((4, 1024), np.float32))
    with pytest.raises(glx.GlxError) as e:
        f.aggregate("SumAggregator", np.zeros(1, np.int64), np.zeros(1, np.int32), 1 << 22,
                    out=(np.zeros((1, 1), np.float32), np.zeros(1, np.int32)))
    assert e.value.code == 3
    with pytest.raises(glx.GlxError):
        dev.sample("TopkSampler", np.zeros(4, np.int64), 3, padding_mode=7)
    with pytest.raises(glx.GlxError):
        glx.partition(__import__("torch").zeros(4, dtype=__import__("torch").int64, device="cuda"), 65)


def test_id_map_with_extreme_and_colliding_ids(orc):
    """Hashed id map at its maximum load (V a power of two -> load 0.5), ids spread over
    the whole int64 range incl. INT64_MAX / negatives, probes for many absent ids."""
    rng = np.random.default_rng(99)
    V = 4096
    raw = np.unique(np.concatenate([
        rng.integers(-(1 << 62), 1 << 62, V * 2),
        np.array([np.iinfo(np.int64).max, np.iinfo(np.int64).min + 1, 0, -1, 1], np.int64)]))
    raw = raw[rng.permutation(raw.shape[0])][:V].astype(np.int64)
    # plus a run of ids that differ only in the high bits (same low-bit pattern)
    raw[:256] = (np.arange(256, dtype=np.int64) << 40) + 12345
    deg = rng.integers(0, 6, V)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    E = int(rp[-1])
    col = raw[rng.integers(0, V, E)]
    eid = rng.permutation(E).astype(np.int64)
    dev = glx.Graph(rp, col, eid, ids=raw)
    og = dict(row_ptr=rp, col=col, eid=eid, ids=raw)
    absent = rng.integers(-(1 << 62), 1 << 62, 3000).astype(np.int64)
    q = np.concatenate([raw, absent, (np.arange(300, dtype=np.int64) << 40) + 12346])
    for name in ("RandomSampler", "RandomWithoutReplacementSampler", "TopkSampler"):
        n, e = dev.sample(name, q, 4, seed=8, call_counter=2, default_neighbor_id=-9)
        on, oe = orc.sample(og, name, q, 4, seed=8, call_counter=2, default_neighbor_id=-9)
        assert np.array_equal(n, on) and np.array_equal(e, oe), name
    d = dev.degrees(q)
    known = {int(v): int(deg[i]) for i, v in enumerate(raw)}
    assert d.tolist() == [known.get(int(v), 0) for v in q]
    X = rng.standard_normal((V, 12)).astype(np.float32)
    f = glx.Features(X, ids=raw)
    seg = (np.arange(q.shape[0]) // 7).astype(np.int32)
    Sg = int(seg[-1]) + 1
    emb, cnt = f.aggregate("SumAggregator", q, seg, Sg, default_attr=0.25)
    oemb, ocnt = orc.aggregate(X, "SumAggregator", q, seg, Sg, 0.25, ids=raw)
    assert np.array_equal(cnt, ocnt) and beq(emb, oemb)
    out = f.lookup(q, default_attr=0.25)
    row = {int(v): i for i, v in enumerate(raw)}
    exp = np.stack([X[row[int(v)]] if int(v) in row else np.full(12, 0.25, np.float32) for v in q])
    assert beq(out, exp)


@pytest.mark.parametrize("weighted", [True, False])
def test_device_side_build_from_edge_list(orc, weighted):
    """glx_graph_build (radix sorts + RLE + scan on the GPU) must give the storage the
    reference builds on the host: rows by weight descending (ties by insertion),
    edge id = insertion index (memory_adj_matrix.cc:105-125, memory_edge_storage.cc:53-57)."""
    import torch
    rng = np.random.default_rng(31 + weighted)
    V, E = 4000, 120000
    raw = (np.arange(V, dtype=np.int64) * 13 - 20000)
    s_idx = np.minimum((rng.pareto(1.1, E) * 5).astype(np.int64), V - 1)
    src = raw[s_idx]
    dst = raw[rng.integers(0, V, E)]
    w = None
    if weighted:
        w = (rng.integers(1, 200, E) / 200.0).astype(np.float32)  # many exact ties
    # host-side construction the oracle way: rows in first-appearance order, stable sort
    _, first = np.unique(src, return_index=True)
    rows = src[np.sort(first)]
    row_of = {int(v): i for i, v in enumerate(rows)}
    r = np.array([row_of[int(x)] for x in src])
    order = np.argsort(r, kind="stable")
    rp = np.zeros(rows.shape[0] + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp)
    col, eid = dst[order].copy(), order.astype(np.int64)
    ws = None
    if weighted:
        col, eid, ws = orc.sort_rows(rp, col, eid, w[order].copy())
    og = dict(row_ptr=rp, col=col, eid=eid, weight=ws, ids=rows)
    if weighted:
        og["alias"] = orc.alias_build(rp, ws)
    for dev_graph in (glx.Graph.from_edges(src, dst, w),
                      glx.Graph.from_edges(torch.from_numpy(src).cuda(), torch.from_numpy(dst).cuda(),
                                           torch.from_numpy(w).cuda() if weighted else None)):
        assert dev_graph.num_rows == rows.shape[0] and dev_graph.num_edges == E
        q = np.concatenate([rows, [7, -7, 10 ** 15]]).astype(np.int64)
        names = SAMPLERS if weighted else ["RandomSampler", "RandomWithoutReplacementSampler", "TopkSampler"]
        for name in names:
            for k in (3, 40):
                n, e = dev_graph.sample(name, q, k, seed=5, call_counter=k, default_neighbor_id=-1)
                on, oe = orc.sample(og, name, q, k, seed=5, call_counter=k, default_neighbor_id=-1)
                assert np.array_equal(n, on) and np.array_equal(e, oe), (name, k, weighted)
    g0 = glx.Graph.from_edges(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32))
    n, e = g0.sample("TopkSampler", np.arange(3, dtype=np.int64), 2, default_neighbor_id=5)
    assert (n == 5).all() and (e == -1).all()


def test_multi_hop_driver_equals_chained_calls(orc, graphs):
    """glx_sample_hops == the hop loop of NeighborSampler.get (neighbor_sampler.py:93-127)."""
    import torch
    og, dev = graphs["dense"]
    _, dev2 = graphs["hashed"]
    seeds = np.random.default_rng(6).integers(0, 3000, 300).astype(np.int64)
    for name in SAMPLERS:
        outs = glx.sample_hops([dev, dev, dev], name, seeds, [5, 3, 2], seed=12, call_counter=40)
        n1, e1 = dev.sample(name, seeds, 5, seed=12, call_counter=40)
        n2, e2 = dev.sample(name, n1.reshape(-1), 3, seed=12, call_counter=41)
        n3, e3 = dev.sample(name, n2.reshape(-1), 2, seed=12, call_counter=42)
        for (a, b), (c, d) in zip(outs, [(n1, e1), (n2, e2), (n3, e3)]):
            assert np.array_equal(a, c) and np.array_equal(b, d), name
        t = glx.sample_hops([dev, dev, dev], name, torch.from_numpy(seeds).cuda(), [5, 3, 2], seed=12,
                            call_counter=40)
        torch.cuda.synchronize()
        assert np.array_equal(t[2][0].cpu().numpy(), n3) and np.array_equal(t[2][1].cpu().numpy(), e3)
    with pytest.raises(glx.GlxError):
        glx.sample_hops([dev], "TopkSampler", np.zeros(1 << 20, np.int64), [4096])


def test_partition_stitch_reference_unittest_layout():
    """Device HashPartitioner / Stitcher on partition_stitch_unittest.cpp's DenseReq_DenseRes case."""
    import torch
    ids = torch.tensor([1, 2, 3, 4], dtype=torch.int64, device="cuda")
    bucketed, order, counts = glx.partition(ids, 2)
    assert counts.tolist() == [2, 2] and order.tolist() == [1, 3, 0, 2] and bucketed.tolist() == [2, 4, 1, 3]
    rows = torch.stack([bucketed * 100 + j for j in range(6)], 1).contiguous()
    out = glx.stitch(rows, order)
    assert out.tolist() == [[i * 100 + j for j in range(6)] for i in (1, 2, 3, 4)]


# -------------------------------------------- next rows: Full / InDegree samplers ---
def test_full_sampler_golden_and_oracle(orc, graphs):
    """FullSampler (full_sampler.cc:28-97) vs the reference's own outputs and the oracle."""
    import torch
    g = load("rand_graph.npz")
    dev = glx.Graph(g["row_ptr"], g["col"], g["eid"], g["w_slot"], ids=g["rows"])
    for lim in (0, 3, 33):
        d, n, e = dev.sample_full(g["query"], lim)
        assert np.array_equal(d, g["full_l%d_deg" % lim])
        assert np.array_equal(n, g["full_l%d_nbr" % lim]) and np.array_equal(e, g["full_l%d_eid" % lim])
    og, dev2 = graphs["dense"]
    q = np.concatenate([np.random.default_rng(3).integers(0, 3000, 500), [0, -1, 3000]]).astype(np.int64)
    for lim in (0, 1, 7, 100000):
        d, n, e = dev2.sample_full(q, lim)
        od, on, oe = orc.sample_full(og, q, lim)
        assert np.array_equal(d, od) and np.array_equal(n, on) and np.array_equal(e, oe), lim
        td, tn, te = dev2.sample_full(torch.from_numpy(q).cuda(), lim)
        assert np.array_equal(td.cpu().numpy(), od) and np.array_equal(tn.cpu().numpy(), on)
    d, n, e = dev2.sample_full(np.zeros(0, np.int64), 5)
    assert d.shape == (0,) and n.shape == (0,)


def test_in_degree_sampler_tables_and_parity(orc, graphs):
    """InDegreeSampler: device in-degree alias tables == the reference's (golden), samples == oracle."""
    g = load("rand_graph.npz")
    dev = glx.Graph(g["row_ptr"], g["col"], g["eid"], g["w_slot"], ids=g["rows"])
    with pytest.raises(glx.GlxError):
        dev.sample("InDegreeSampler", g["query"], 3)  # tables not enabled yet
    dev.enable_in_degree()
    og = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], ids=g["rows"],
              indeg_alias=(g["indeg_alias_prob"], g["indeg_alias_idx"]))  # the REFERENCE's tables
    for pad in (1, 0):
        for k in (1, 4, 33):
            n, e = dev.sample("InDegreeSampler", g["query"], k, seed=21, call_counter=k, padding_mode=pad,
                              default_neighbor_id=-7)
            on, oe = orc.sample(og, "InDegreeSampler", g["query"], k, seed=21, call_counter=k, padding_mode=pad,
                                default_neighbor_id=-7)
            assert np.array_equal(n, on) and np.array_equal(e, oe), (pad, k)
    # larger graph with hubs, unweighted + device-built
    ogd, _ = graphs["dense"]
    big = glx.Graph(ogd["row_ptr"], ogd["col"], ogd["eid"]).enable_in_degree()
    o2 = dict(row_ptr=ogd["row_ptr"], col=ogd["col"], eid=ogd["eid"])
    o2["indeg_alias"], _ = orc.in_degree_alias(o2)
    q = np.arange(3000, dtype=np.int64)
    n, e = big.sample("InDegreeSampler", q, 10, seed=2, call_counter=5)
    on, oe = orc.sample(o2, "InDegreeSampler", q, 10, seed=2, call_counter=5)
    assert np.array_equal(n, on) and np.array_equal(e, oe)


def test_in_degree_sampler_matches_reference_distribution(orc):
    from scipy import stats
    g = load("dist_indegree.npz")
    dev = glx.Graph(g["row_ptr"], g["col"], g["eid"], ids=g["rows"]).enable_in_degree()
    T, degs, k = int(g["T"]), g["degs"], 4
    _, eid = dev.sample("InDegreeSampler", np.tile(g["rows"][:len(degs)], T), k, seed=77, call_counter=3)
    eid = eid.reshape(T, len(degs), k)
    rp = g["row_ptr"]
    for r, d in enumerate(degs):
        pos_of = {int(x): i for i, x in enumerate(g["eid"][rp[r]:rp[r + 1]])}
        pos = np.vectorize(pos_of.get)(eid[:, r, :])
        for j in range(k):
            a = np.bincount(pos[:, j], minlength=d).astype(float)
            b = g["hist"][r, j, :d].astype(float)
            keep = (a + b) > 0
            if keep.sum() >= 2:
                assert stats.chi2_contingency(np.stack([a[keep], b[keep]]))[1] > 1e-4, (r, j)


def test_concurrent_host_threads_on_private_streams(orc, graphs):
    """Handles are immutable and entry points re-entrant: several host threads, each on its own
    HIP stream (device pointers) or the per-thread stream (host pointers), get the same results
    as a serial run (the reference calls Process() from up to 32 pool threads)."""
    import threading
    import torch
    og, dev = graphs["dense"]
    X = np.random.default_rng(5).standard_normal((3000, 64)).astype(np.float32)
    f = glx.Features(X)
    q = np.random.default_rng(6).integers(0, 3000, 2000).astype(np.int64)
    seg = (np.arange(2000 * 8) // 8).astype(np.int32)
    exp = {}
    for t in range(8):
        n, e = orc.sample(og, SAMPLERS[t % 4], q, 8, seed=t, call_counter=t)
        emb, cnt = orc.aggregate(X, AGGREGATORS[t % 5], n.reshape(-1), seg, 2000)
        exp[t] = (n, e, emb)
    errors = []

    def work(t, device_mode):
        try:
            for rep in range(10):
                if device_mode:
                    st = torch.cuda.Stream()
                    with torch.cuda.stream(st):
                        n, e = dev.sample(SAMPLERS[t % 4], torch.from_numpy(q).cuda(), 8, seed=t, call_counter=t)
                        emb, _ = f.aggregate(AGGREGATORS[t % 5], n.view(-1), torch.from_numpy(seg).cuda(), 2000)
                        st.synchronize()
                    n, e, emb = n.cpu().numpy(), e.cpu().numpy(), emb.cpu().numpy()
                else:
                    n, e = dev.sample(SAMPLERS[t % 4], q, 8, seed=t, call_counter=t)
                    emb, _ = f.aggregate(AGGREGATORS[t % 5], n.reshape(-1), seg, 2000)
                if not (np.array_equal(n, exp[t][0]) and np.array_equal(e, exp[t][1]) and beq(emb, exp[t][2])):
                    errors.append((t, device_mode, rep))
        except Exception as ex:  # noqa: BLE001
            errors.append((t, device_mode, repr(ex)))

    threads = [threading.Thread(target=work, args=(t, t % 2 == 0)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


# ---------------------------------------------------------------- negative samplers ----
def _negative_world():
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "negative.npz")))
    graph = glx.Graph(g["row_ptr"], g["col"], g["eid"], g["w_slot"], ids=g["rows"])
    og = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
    return g, graph, og


def test_negative_tables_equal_reference_golden():
    """glx_negative_from_graph: candidate order, in-degrees and the global alias table are the
    reference's (tests/golden/negative.npz), bit for bit."""
    g, graph, _ = _negative_world()
    t = glx.Negative.from_graph(graph, by_in_degree=True)
    ids, prob, alias = t.export()
    assert np.array_equal(ids, g["dst_ids"])
    assert np.array_equal(prob.view(np.uint32), g["indeg_prob"].view(np.uint32)) and np.array_equal(alias, g["indeg_alias"])
    u = glx.Negative.from_graph(graph)
    assert not u.weighted and np.array_equal(u.export()[0], g["dst_ids"])
    n = glx.Negative(g["node_ids"], g["node_weights"])
    ids, prob, alias = n.export()
    assert np.array_equal(prob.view(np.uint32), g["node_prob"].view(np.uint32)) and np.array_equal(alias, g["node_alias"])


@pytest.mark.parametrize("count", [1, 6, 64, 150])
def test_negative_sampling_bit_exact_vs_oracle(count):
    from oracle_bindings import Oracle
    orc = Oracle()
    g, graph, og = _negative_world()
    graph.enable_negative()
    rng = np.random.default_rng(count)
    src = np.concatenate([g["rows"][rng.integers(0, g["rows"].shape[0], 300)], [10 ** 9, -7]]).astype(np.int64)
    ids = g["dst_ids"]
    uni = glx.Negative.from_graph(graph)
    deg = glx.Negative.from_graph(graph, by_in_degree=True)
    table = (g["indeg_prob"], g["indeg_alias"])
    for cc in (0, 9):
        a = uni.sample(src, count, seed=3, call_counter=cc)
        assert np.array_equal(a, orc.negative_sample(ids, None, 0, og, src, count, seed=3, call_counter=cc))
        a = deg.sample(src, count, seed=3, call_counter=cc)
        assert np.array_equal(a, orc.negative_sample(ids, table, 0, og, src, count, seed=3, call_counter=cc))
        a = deg.sample(src, count, exclude=glx.NEG_EXCLUDE_NEIGHBORS, graph=graph, seed=3, call_counter=cc)
        assert np.array_equal(a, orc.negative_sample(ids, table, 1, og, src, count, seed=3, call_counter=cc))
    nodes = glx.Negative(g["node_ids"], g["node_weights"])
    batch = g["node_ids"][rng.integers(0, 400, 200)]
    a = nodes.sample(batch, count, exclude=glx.NEG_EXCLUDE_BATCH, seed=11, call_counter=1)
    want = orc.negative_sample(g["node_ids"], (g["node_prob"], g["node_alias"]), 2, None, batch, count, seed=11, call_counter=1)
    assert np.array_equal(a, want)
    # torch CUDA tensors in -> out, same draws
    import torch
    d = deg.sample(torch.from_numpy(src).cuda(), count, exclude=glx.NEG_EXCLUDE_NEIGHBORS, graph=graph, seed=3, call_counter=9)
    assert d.is_cuda and np.array_equal(d.cpu().numpy(), orc.negative_sample(ids, table, 1, og, src, count, seed=3, call_counter=9))


def test_negative_sampling_exhausted_candidates_and_empty_list():
    from oracle_bindings import Oracle
    orc = Oracle()
    # every candidate is a neighbour of every source: only the 4th block can deliver
    rp = np.array([0, 3, 6], np.int64)
    col = np.array([5, 6, 7, 5, 6, 7], np.int64)
    eid = np.arange(6, dtype=np.int64)
    graph = glx.Graph(rp, col, eid, None)
    graph.enable_negative()
    og = dict(row_ptr=rp, col=col, eid=eid)
    t = glx.Negative.from_graph(graph, by_in_degree=True)
    ids, prob, alias = t.export()
    src = np.array([0, 1, 1, 0], np.int64)
    got = t.sample(src, 5, exclude=glx.NEG_EXCLUDE_NEIGHBORS, graph=graph, seed=2, call_counter=5)
    assert np.array_equal(got, orc.negative_sample(ids, (prob, alias), 1, og, src, 5, seed=2, call_counter=5))
    empty = glx.Negative(np.zeros(0, np.int64))
    assert (empty.sample(src, 3, default_neighbor_id=-4) == -4).all()


def test_node_weight_negative_sampler_drops_its_set_for_the_rest_of_the_request():
    """node_weight_negative_sampler.cc:68-80: the exclusion set (the request's own ids) is ONE object for all rows, so
    when a row exhausts its retries and clears it, every LATER row is sampled without any exclusion -- a behaviour the
    draw-for-draw comparison with the reference found (tests/test_oracle_refseq.py).  Six candidates, five of them in
    the batch, one negative per row: most blocks miss, some row soon runs out of retries, and from the next row on the
    batch's own ids come back.  Device == oracle, host and device pointers, also with larger counts."""
    import torch
    from oracle_bindings import Oracle
    orc = Oracle()
    ids = np.array([10, 11, 12, 13, 14, 15], np.int64)
    w = np.array([5, 5, 5, 5, 5, 0.2], np.float32)
    t = glx.Negative(ids, w)
    _, prob, alias = t.export()
    batch = np.tile(ids[:5], 40)
    dropped = False
    for count, seed in ((1, 1), (1, 2), (2, 3), (3, 4), (9, 5)):
        want = orc.negative_sample(ids, (prob, alias), 2, None, batch, count, seed=seed, call_counter=3)
        got = t.sample(batch, count, exclude=glx.NEG_EXCLUDE_BATCH, seed=seed, call_counter=3)
        assert np.array_equal(got, want), (count, seed)
        got_d = t.sample(torch.from_numpy(batch).cuda(), count, exclude=glx.NEG_EXCLUDE_BATCH, seed=seed, call_counter=3)
        assert np.array_equal(got_d.cpu().numpy(), want), (count, seed)
        inside = np.isin(want, ids[:5]).any(axis=1)
        if inside.any():
            first = int(np.argmax(inside))
            free = orc.negative_sample(ids, (prob, alias), 0, None, batch, count, seed=seed, call_counter=3)
            assert np.array_equal(want[first + 1:], free[first + 1:])  # behind the row that dropped the set: no exclusion
            dropped = True
    assert dropped


def test_in_degree_lookup_equals_bincount():
    rng = np.random.default_rng(5)
    V, E = 3000, 40000
    rp, col, eid, w = synth.small_graph(V, E, seed=31, weighted=True, hub_degree=900)
    g = glx.Graph(rp, col, eid, w).enable_in_degree()
    want = np.bincount(col, minlength=V + 10)
    q = np.concatenate([rng.integers(0, V + 10, 5000), [-1, 10 ** 12]]).astype(np.int64)
    got = g.in_degrees(q)
    ok = (q >= 0) & (q < V + 10)
    assert np.array_equal(got[ok], want[q[ok]]) and (got[~ok] == 0).all()
    import torch
    assert np.array_equal(g.in_degrees(torch.from_numpy(q).cuda()).cpu().numpy(), got)


def test_timestamped_edge_type_rows_in_timestamp_order():
    """glx_graph_build_ordered(GLX_ORDER_TIMESTAMP_ASC) == the reference's post-Build adjacency of a
    timestamped + weighted edge type (tests/golden/timestamped.npz), via the Full and Topk samplers."""
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "timestamped.npz")))
    dev = glx.Graph.from_edges(g["src"], g["dst"], g["w"], timestamp=g["ts"])
    deg, nbr, eid = dev.sample_full(g["rows"], 0)
    assert np.array_equal(deg, np.diff(g["row_ptr"])) and np.array_equal(nbr, g["col"]) and np.array_equal(eid, g["eid"])
    n, e = dev.sample("TopkSampler", g["rows"], 4)
    assert np.array_equal(n, g["topk_nbr"]) and np.array_equal(e, g["topk_eid"])
    import torch
    t = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    dev2 = glx.Graph.from_edges(t(g["src"]), t(g["dst"]), t(g["w"]), timestamp=t(g["ts"]))
    assert np.array_equal(dev2.sample("TopkSampler", g["rows"], 4)[0], g["topk_nbr"])


def test_request_plan_equals_separate_calls():
    """glx_plan (one hipGraph launch per batch: the small-batch path) == glx_sample_hops + glx_aggregate
    called one by one, bit for bit, for every sampler, with dense and hashed ids, over several runs with
    changing seeds and call counters."""
    import torch
    rng = np.random.default_rng(8)
    rp, col, eid, w = synth.small_graph(3000, 60000, seed=5, weighted=True, hub_degree=500)
    X = rng.standard_normal((3000, 32)).astype(np.float32)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    for hashed in (False, True):
        ids = (np.arange(3000, dtype=np.int64) * 3 - 500) if hashed else None
        mapped = (lambda a: a * 3 - 500) if hashed else (lambda a: a)
        g = glx.Graph(t(rp), t(mapped(col)), t(eid), t(w), ids=t(ids) if hashed else None)
        f = glx.Features(t(X), ids=t(ids) if hashed else None)
        for name in glx.SAMPLER_IDS:
            for agg in ("MaxAggregator", "MeanAggregator"):
                plan = glx.Plan([g, g], name, [7, 4], 512, features=[f, f], agg=agg, seed=31, default_neighbor_id=-4,
                                default_attr=0.5)
                for run in range(3):
                    seeds = t(mapped(rng.integers(0, 3005, 512)).astype(np.int64))
                    hops = plan.run(seeds, call_counter=10 * run)
                    ref = glx.sample_hops([g, g], name, seeds, [7, 4], seed=31, call_counter=10 * run,
                                          default_neighbor_id=-4)
                    torch.cuda.synchronize()
                    for h in range(2):
                        assert torch.equal(hops[h]["nbr"], ref[h][0]) and torch.equal(hops[h]["eid"], ref[h][1]), (name, h)
                        k = (7, 4)[h]
                        e, c = f.aggregate(agg, ref[h][0].view(-1), None, ref[h][0].shape[0], default_attr=0.5)
                        assert torch.equal(hops[h]["cnt"], c), (name, agg, h)
                        assert torch.equal(hops[h]["emb"].view(torch.int32), e.view(torch.int32)), (name, agg, h, k)
                plan.close()
        # sampling only
        plan = glx.Plan([g], "TopkSampler", [5], 64)
        seeds = t(mapped(rng.integers(0, 3000, 64)).astype(np.int64))
        out = plan.run(seeds)
        torch.cuda.synchronize()
        assert torch.equal(out[0]["nbr"], g.sample("TopkSampler", seeds, 5)[0])
        plan.close()


def test_alias_tables_of_long_and_odd_rows_bit_exact():
    """The wave-per-row alias build (rows longer than 96 slots): bit-identical to the oracle's restatement of
    AliasMethod::Build (alias_method.cc:57-107) for long rows of every kind -- weights in (0.01, 1] (parallel
    double sum is exact), integer weights (in-degrees), all-equal weights (no low / high entries at all), one
    dominating weight, zeros mixed in, and a dynamic range beyond 52 bits, denormals, a huge value (the
    sequential-sum fallback) -- at lengths around the window size of the pairing loop (256) and far above."""
    orc = Oracle()
    rng = np.random.default_rng(99)
    rows = []
    for n in (97, 255, 256, 257, 513, 4000, 70000):
        rows.append((rng.random(n) * 0.99 + 0.01).astype(np.float32))
        rows.append(rng.integers(1, 100000, n).astype(np.float32))
        rows.append(np.full(n, 0.25, np.float32))
        one = (rng.random(n) * 0.01).astype(np.float32)
        one[n // 3] = 1000.0
        rows.append(one)
        z = (rng.random(n)).astype(np.float32)
        z[::3] = 0.0
        rows.append(z)
        wide = (10.0 ** rng.uniform(-30, 30, n)).astype(np.float32)
        rows.append(wide)
        den = (rng.random(n) * 0.5).astype(np.float32)
        den[1] = np.float32(1e-42)  # denormal
        den[2] = np.float32(3e38)
        rows.append(den)
    # the pairing loop pops 64-entry register windows: stacks of exactly / just over / just under a window, steps whose
    # result is exactly 1 (0.5 + 1.5: nothing is carried), negative weights (negative probabilities are lows), and rows
    # with an infinite or NaN weight (the IEEE-comparison variant of the loop)
    for nl, nh in ((64, 64), (65, 63), (63, 65), (128, 64), (64, 128), (129, 127), (1, 200), (200, 1)):
        rows.append(np.concatenate([np.full(nl, 0.5, np.float32), np.full(nh, 1.5, np.float32)]))
        mixed = np.concatenate([rng.random(nl) * 0.9, 1.1 + rng.random(nh)]).astype(np.float32)
        rows.append(rng.permutation(mixed))
    for n in (127, 128, 129, 191, 192, 193, 1000):
        neg = (rng.random(n) + 0.05).astype(np.float32)
        neg[::7] = -0.01
        rows.append(neg)
        inf = (rng.random(n) + 0.05).astype(np.float32)
        inf[n // 2] = np.inf
        rows.append(inf)
        nan = (rng.random(n) + 0.05).astype(np.float32)
        nan[n // 5] = np.nan
        rows.append(nan)
        # weights that sum to exactly zero: probabilities +inf (highs) and -inf (lows), every step yields NaN
        pm = np.where(np.arange(n + (n & 1)) % 2 == 0, 1.0, -1.0).astype(np.float32)
        rows.append(pm)
        rows.append(np.zeros(n, np.float32))
    deg = np.array([r.shape[0] for r in rows], np.int64)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    w = np.concatenate(rows).astype(np.float32)
    E = int(rp[-1])
    col = (np.arange(E, dtype=np.int64) * 7) % 1000
    eid = np.arange(E, dtype=np.int64)
    dev = glx.Graph(rp, col, eid, w)
    prob, alias = dev.export_alias()
    oprob, oalias = orc.alias_build(rp, w)
    for r in range(len(rows)):
        a, b = rp[r], rp[r + 1]
        assert np.array_equal(alias[a:b], oalias[a:b]), ("alias", r, deg[r])
        assert np.array_equal(prob[a:b].view(np.uint32), oprob[a:b].view(np.uint32)), ("prob", r, deg[r])


def test_pinned_host_buffers_are_written_directly():
    """Host-pointer calls write their results straight into caller buffers pinned with glx_host_register (no
    staging copy); the answers equal those of pageable buffers, for sampling (plain and filtered) and aggregation,
    and registration can be undone.  The body (tests/scripts/pinned_host_check.py) runs in a process of its own: after
    hipHostUnregister the ROCm 7.0 runtime was seen to fault on later pageable copies of the same process -- once in
    ~10 runs of this suite the NEXT test's first host-to-device copy returned "an illegal memory access" (or the process
    aborted) and took every later GPU test with it.  GLX_TEST_PINNED_INPROC=1 runs the body here (scripts/r06/crash_hunt.sh)."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scripts", "pinned_host_check.py")
    if os.environ.get("GLX_TEST_PINNED_INPROC") == "1":
        import importlib.util
        spec = importlib.util.spec_from_file_location("pinned_host_check", script)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.check()
        return
    r = subprocess.run([sys.executable, script], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "PINNED_OK" in r.stdout, r.stdout[-3000:]


def test_alias_tables_fuzz_bit_exact():
    """300 random rows of random lengths above the one-lane limit and random weight families -- uniform, heavy-tailed,
    powers of two (sums and steps that land exactly on 1), small integers with many ties -- through the wave build's
    hand-scheduled pairing loop, against the oracle's AliasMethod::Build, bit for bit."""
    orc = Oracle()
    rng = np.random.default_rng(2024)
    rows = []
    for i in range(300):
        n = int(rng.integers(97, 3000))
        fam = i % 5
        if fam == 0:
            r = rng.random(n) + 1e-3
        elif fam == 1:
            r = rng.pareto(1.2, n) + 1e-3
        elif fam == 2:
            r = 2.0 ** rng.integers(-6, 7, n)
        elif fam == 3:
            r = rng.integers(1, 4, n).astype(np.float64)
        else:
            r = np.where(rng.random(n) < 0.5, 0.5, 1.5)  # half lows, half highs, every pair sums to exactly 1
            r[rng.integers(0, n)] += rng.random()
        rows.append(r.astype(np.float32))
    deg = np.array([r.shape[0] for r in rows], np.int64)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    w = np.concatenate(rows).astype(np.float32)
    E = int(rp[-1])
    dev = glx.Graph(rp, (np.arange(E, dtype=np.int64) * 3) % 500, np.arange(E, dtype=np.int64), w)
    prob, alias = dev.export_alias()
    oprob, oalias = orc.alias_build(rp, w)
    bad = np.nonzero((alias != oalias) | (prob.view(np.uint32) != oprob.view(np.uint32)))[0]
    assert bad.size == 0, ("first mismatch in row", int(np.searchsorted(rp, bad[0], side="right") - 1), int(bad.size))


def _fuzz_cases(n):
    first = int(os.environ.get("GLX_FUZZ_FIRST", "0"))
    return list(range(first, first + int(os.environ.get("GLX_FUZZ_CASES", str(n)))))


@pytest.mark.parametrize("case", _fuzz_cases(10))
def test_request_plan_fuzz(case):
    """Plans of random shape -- 1 to 3 hops, odd fanouts and batches (1 seed included), feature widths that are not a
    multiple of 4, either padding, with and without the aggregation -- replayed several times, against separate calls."""
    import torch
    rng = np.random.default_rng(600 + case)
    Vp = int(rng.choice([50, 800]))
    rp, col, eid, w = synth.small_graph(Vp, 12 * Vp, seed=case, weighted=True, hub_degree=int(rng.choice([0, 200])))
    Dp = int(rng.choice([1, 5, 8, 33]))
    X = rng.standard_normal((Vp, Dp)).astype(np.float32)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    g, f = glx.Graph(t(rp), t(col), t(eid), t(w)), glx.Features(t(X))
    hops = int(rng.integers(1, 4))
    fan = [int(x) for x in rng.choice([1, 2, 3, 5, 17], hops)]
    batch = int(rng.choice([1, 3, 100]))
    pad = int(rng.integers(0, 2))
    name = list(glx.SAMPLER_IDS)[case % len(glx.SAMPLER_IDS)]
    agg = [None, "SumAggregator", "MaxAggregator", "MeanAggregator"][case % 4]
    plan = glx.Plan([g] * hops, name, fan, batch, features=[f] * hops if agg else None, agg=agg, seed=77, padding_mode=pad,
                    default_neighbor_id=-6, default_attr=1.5)
    for run in range(3):
        seeds = t(rng.integers(-1, Vp + 2, batch).astype(np.int64))
        out = plan.run(seeds, call_counter=100 * run)
        ref = glx.sample_hops([g] * hops, name, seeds, fan, seed=77, call_counter=100 * run, padding_mode=pad,
                              default_neighbor_id=-6)
        torch.cuda.synchronize()
        for h in range(hops):
            assert torch.equal(out[h]["nbr"], ref[h][0]) and torch.equal(out[h]["eid"], ref[h][1]), (name, h, run)
            if agg:
                e, c = f.aggregate(agg, ref[h][0].reshape(-1), None, ref[h][0].shape[0], default_attr=1.5)
                assert torch.equal(out[h]["cnt"], c) and torch.equal(out[h]["emb"].view(torch.int32), e.view(torch.int32)), (agg, h)
    plan.close()


def test_alias_tables_of_degenerate_weight_rows_equal_the_oracle():
    """Rows whose weights are all zero (the reference's sum is 0 and every probability NaN), partly zero, all equal,
    spread over 60 orders of magnitude, or small integers: the device alias build (lane-serial and wave kernels) equals
    the oracle's -- which equals the reference's AliasMethod on exactly such rows (20,000 live cases on the CPU) -- bit for
    bit, NaNs included; and the EdgeWeight draws on them agree."""
    from oracle_bindings import Oracle
    orc = Oracle()
    rng = np.random.default_rng(77)
    degs, ws = [], []
    for r in range(600):
        n = int(rng.choice([1, 2, 3, 17, 64, 65, 300, 5000])) if r % 50 == 0 else int(rng.integers(1, 40))
        kind = r % 6
        if kind == 0:
            w = rng.random(n)
        elif kind == 1:
            w = np.zeros(n)
        elif kind == 2:
            w = rng.random(n) * (rng.random(n) < 0.5)
        elif kind == 3:
            w = np.full(n, rng.random())
        elif kind == 4:
            w = 10.0 ** rng.integers(-30, 30, n)
        else:
            w = rng.integers(1, 4, n)
        degs.append(n)
        ws.append(w.astype(np.float32))
    rp = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    w = np.concatenate(ws)
    E = int(rp[-1])
    col = (np.arange(E, dtype=np.int64) * 7) % 1000
    eid = np.arange(E, dtype=np.int64)
    dev = glx.Graph(rp, col, eid, w)
    prob, alias = dev.export_alias()
    oprob, oalias = orc.alias_build(rp, w)
    assert np.array_equal(prob.view(np.uint32), oprob.view(np.uint32)) and np.array_equal(alias, oalias)
    q = np.arange(600, dtype=np.int64)
    og = dict(row_ptr=rp, col=col, eid=eid, weight=w, alias=(oprob, oalias))
    n, e = dev.sample("EdgeWeightSampler", q, 9, seed=3, call_counter=1)
    on, oe = orc.sample(og, "EdgeWeightSampler", q, 9, seed=3, call_counter=1)
    assert np.array_equal(n, on) and np.array_equal(e, oe)

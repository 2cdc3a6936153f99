"""GPU tests of the round-3 additions: the LDS-staged MFMA formulation of Sum / Mean (ablation knob
GLX_AGG_MFMA=1, sum_aggregator.cc:25-33 / mean_aggregator.cc:26-61) and the memory-system probes."""
import os

import numpy as np
import pytest
import torch

import glx
from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture
def mfma_on():
    glx.tune("agg_mfma", 1)  # the knob is read from the environment once per process; glx_tune sets it at run time
    yield
    glx.tune("agg_mfma", 0)


@pytest.mark.parametrize("dim", [64, 128, 256])
@pytest.mark.parametrize("fanout", [1, 3, 10, 25, 33])
@pytest.mark.parametrize("op", ["SumAggregator", "MeanAggregator"])
def test_mfma_formulation_is_bit_identical_to_the_oracle(mfma_on, dim, fanout, op):
    rng = np.random.default_rng(dim + fanout)
    V, Sg = 3000, 1000 + fanout  # not a multiple of the 16-segment tile
    X = rng.standard_normal((V, dim)).astype(np.float32)
    ids = rng.integers(-5, V + 5, Sg * fanout).astype(np.int64)  # some unknown ids -> default rows
    f = glx.Features(torch.from_numpy(X).cuda(), device=0)
    emb, cnt = f.aggregate(op, torch.from_numpy(ids).cuda(), None, Sg, default_attr=0.25)
    seg = (np.arange(ids.shape[0]) // fanout).astype(np.int32)
    oemb, ocnt = Oracle().aggregate(X, op, ids, seg, Sg, default_attr=0.25)
    assert np.array_equal(cnt.cpu().numpy(), ocnt)
    assert np.array_equal(emb.cpu().numpy().view(np.uint32), oemb.view(np.uint32))


def test_mfma_knob_leaves_other_shapes_on_the_valu_kernel(mfma_on):
    rng = np.random.default_rng(1)
    X = rng.standard_normal((500, 96)).astype(np.float32)  # dim 96: not an MFMA shape
    ids = rng.integers(0, 500, 640).astype(np.int64)
    f = glx.Features(torch.from_numpy(X).cuda(), device=0)
    for op in ("SumAggregator", "MaxAggregator"):
        emb, cnt = f.aggregate(op, torch.from_numpy(ids).cuda(), None, 64)
        oemb, ocnt = Oracle().aggregate(X, op, ids, (np.arange(640) // 10).astype(np.int32), 64)
        assert np.array_equal(emb.cpu().numpy().view(np.uint32), oemb.view(np.uint32))


def test_probes_report_plausible_bandwidth():
    r = glx.probe_bandwidth("stream_read", 1 << 30, reps=5)
    assert 500 < r["gbps"] < 8000 and r["moved_bytes"] == float(1 << 30)
    c = glx.probe_bandwidth("copy", 1 << 30, reps=5)
    t = glx.probe_bandwidth("triad", 1 << 30, reps=5)
    assert 500 < c["gbps"] < 8000 and 500 < t["gbps"] < 8000
    g = glx.probe_bandwidth("gather32", 1 << 30, units=1 << 22, reps=5)
    assert g["moved_bytes"] == 48.0 * (1 << 22) and g["ms"] > 0
    w = glx.probe_bandwidth("gather_rows", 1 << 30, units=1 << 20, unit_bytes=1024, reps=5)
    assert 100 < w["gbps"] < 8000
    with pytest.raises(glx.GlxError):
        glx.probe_bandwidth("gather_rows", 1 << 30, units=1 << 20, unit_bytes=1000)


# ---- SubGraphSampler::InduceSubGraph on the device (glx_subgraph_induce, subgraph_sampler.cc:34-95) ------------------
import os as _os  # noqa: E402

SUB = dict(np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "subgraph.npz")))


def _gold_graph():
    return glx.Graph(SUB["row_ptr"], SUB["col"], SUB["eid"], SUB["w_slot"], ids=SUB["rows"])


@pytest.mark.parametrize("case", [str(c) for c in SUB["cases"]])
@pytest.mark.parametrize("kind", ["host", "device"])
def test_subgraph_pipeline_equals_the_reference_golden(case, kind):
    """FullSampler per hop + sorted set + FullSampler(DefaultFullNbrNum) + glx_subgraph_induce == the reference's own
    SubGraphSampler output (tests/golden/subgraph.npz), entry for entry."""
    g = _gold_graph()
    seeds = SUB[case + "_seeds"]
    frontier, found = seeds, set()
    for k in SUB[case + "_num_nbrs"]:
        if k > 0:
            _, nb, _ = g.sample_full(np.ascontiguousarray(frontier), int(k))
            frontier = nb
            found.update(int(x) for x in nb)
    nodes = np.concatenate([seeds, np.array(sorted(found), np.int64)]) if found else seeds
    assert np.array_equal(nodes, SUB[case + "_nodes"])
    deg, nb, ed = g.sample_full(nodes, int(SUB[case + "_full"]))
    off = np.zeros(nodes.shape[0] + 1, np.int64)
    off[1:] = np.cumsum(deg)
    if kind == "device":
        row, col, eid = glx.subgraph_induce(*(torch.from_numpy(x).cuda() for x in (nodes, off, nb, ed)))
        row, col, eid = row.cpu().numpy(), col.cpu().numpy(), eid.cpu().numpy()
    else:
        row, col, eid = glx.subgraph_induce(nodes, off, nb, ed)
    assert np.array_equal(row, SUB[case + "_row"]) and np.array_equal(col, SUB[case + "_col"])
    assert np.array_equal(eid, SUB[case + "_eid"])


def test_subgraph_induce_fuzz_equals_oracle():
    rng = np.random.default_rng(11 + int(_os.environ.get("GLX_FUZZ_FIRST", "0")))
    orc = Oracle()
    for trial in range(30 * max(1, int(_os.environ.get("GLX_FUZZ_CASES", "12")) // 12)):
        n = int(rng.integers(1, 400))
        ids_hi = int(rng.choice([n // 2 + 1, n * 4]))  # dense (many matches, duplicate nodes) or sparse
        nodes = rng.integers(-3, ids_hi, n).astype(np.int64)
        deg = rng.integers(0, 40, n)
        if trial % 5 == 0:
            deg[:] = 0
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum(deg)
        nbr = rng.integers(-3, ids_hi, int(off[-1])).astype(np.int64)  # multi-edges: repeated neighbour ids
        eid = rng.integers(0, 1 << 40, int(off[-1])).astype(np.int64)
        want = orc.subgraph_induce(nodes, off, nbr, eid)
        got = glx.subgraph_induce(nodes, off, nbr, eid)
        for a, b in zip(want, got):
            assert np.array_equal(a, b), trial
        gd = glx.subgraph_induce(*(torch.from_numpy(x).cuda() for x in (nodes, off, nbr, eid)))
        for a, b in zip(want, gd):
            assert np.array_equal(a, b.cpu().numpy()), trial
    r, c, e = glx.subgraph_induce(np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64))
    assert r.size == 0 and c.size == 0 and e.size == 0


# ---- ConditionalNegativeSampler on the device (glx_cond_*; conditional_negative_sampler.cc:37-161) ------------------
def _cond_case(rng, U, ncols, n_users, deg, batch, hashed_ids):
    items = (rng.permutation(U * 3)[:U] + 1000).astype(np.int64) if hashed_ids else np.arange(U, dtype=np.int64)
    weights = (rng.random(U) + 0.05).astype(np.float32)
    keys = np.stack([rng.integers(0, int(rng.integers(1, 6)), U) for _ in range(ncols)]).astype(np.int64) if ncols else \
        np.zeros((0, U), np.int64)
    src, dst = [], []
    for u in range(n_users):
        for d in rng.choice(items, deg, replace=True):
            src.append(u)
            dst.append(int(d))
    src, dst = np.array(src, np.int64), np.array(dst, np.int64)
    rp = np.zeros(n_users + 1, np.int64)
    np.add.at(rp, src + 1, 1)
    rp = np.cumsum(rp)
    req_src = rng.integers(-1, n_users + 2, batch).astype(np.int64)  # some unknown sources
    req_dst = rng.choice(items, batch).astype(np.int64)
    dk = np.zeros((batch, ncols), np.int64)
    pos = {int(v): i for i, v in enumerate(items)}
    for i, d in enumerate(req_dst):
        for c in range(ncols):
            dk[i, c] = keys[c, pos[int(d)]] if rng.random() > 0.1 else glx.NO_KEY  # some rows without a matching group
    return items, weights, keys, dict(row_ptr=rp, col=dst, eid=np.arange(dst.shape[0], dtype=np.int64)), req_src, req_dst, dk


def _fuzz_cases(n):
    first = int(_os.environ.get("GLX_FUZZ_FIRST", "0"))
    return list(range(first, first + int(_os.environ.get("GLX_FUZZ_CASES", str(n)))))


@pytest.mark.parametrize("trial", _fuzz_cases(12))
@pytest.mark.parametrize("rows", ["parallel", "sequential"])
def test_conditional_negative_sampler_is_bit_identical_to_the_oracle(trial, rows, monkeypatch):
    """rows: without `unique` the rows of a request are sampled one wave each against the first-insertion table
    (replayed row by row when a row drops the set); GLX_COND_SEQUENTIAL forces the one-wave walk `unique` always takes."""
    glx.tune("cond_sequential", 1 if rows == "sequential" else -1)  # read from the environment once; set at run time
    rng = np.random.default_rng(100 + trial)
    U = int(rng.choice([1, 2, 7, 60, 300]))
    ncols = int(rng.integers(0, 4))
    batch = int(rng.choice([1, 5, 40, 130]))
    count = int(rng.choice([1, 4, 9, 70]))
    items, w, keys, og, req_src, req_dst, dk = _cond_case(rng, U, ncols, 12, int(rng.integers(0, 9)), batch, trial % 2 == 1)
    props = (rng.dirichlet(np.ones(ncols + 1))[:ncols] if ncols else np.zeros(0)).astype(np.float32)
    weights = None if trial % 3 == 0 else w
    share, unique, retry = bool(trial & 1), bool(trial & 2), int(rng.choice([1, 2, 5]))
    orc = Oracle()
    want = orc.cond_negative_sample(items, weights, keys, props, og, req_src, req_dst, dk, count, batch_share=share,
                                    unique=unique, retry=retry, default_neighbor_id=-7, seed=5, call_counter=trial)
    g = glx.Graph(og["row_ptr"], og["col"], og["eid"])
    tab = glx.CondTable(items, weights, keys if ncols else None)
    got = tab.sample(g, req_src, req_dst, dk if ncols else None, props, count, batch_share=share, unique=unique, retry=retry,
                     default_neighbor_id=-7, seed=5, call_counter=trial)
    assert np.array_equal(got, want), trial
    # device pointers
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
    tab_d = glx.CondTable(T(items), None if weights is None else T(weights), T(keys) if ncols else None)
    got_d = tab_d.sample(g, T(req_src), T(req_dst), T(dk) if ncols else None, props, count, batch_share=share, unique=unique,
                         retry=retry, default_neighbor_id=-7, seed=5, call_counter=trial)
    assert np.array_equal(got_d.cpu().numpy(), want), trial
    # without a graph (node_weight strategy: no neighbour exclusion)
    want_n = orc.cond_negative_sample(items, weights, keys, props, None, req_src, req_dst, dk, count, batch_share=share,
                                      unique=unique, retry=retry, seed=6, call_counter=trial)
    got_n = tab.sample(None, req_src, req_dst, dk if ncols else None, props, count, batch_share=share, unique=unique,
                       retry=retry, seed=6, call_counter=trial)
    assert np.array_equal(got_n, want_n), trial


def test_conditional_negative_sampler_on_the_reference_fixture_matches_oracle():
    """The fixture the oracle is pinned to the reference on (tests/golden/cond_negative.npz): device == oracle there too."""
    G = dict(np.load(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "cond_negative.npz")))
    from test_oracle_cond_negative import setup
    for strategy, share, unique in (("random", False, False), ("in_degree", False, True), ("node_weight", True, True)):
        cand, w, keys, dk, og, _ = setup(strategy)
        props = np.concatenate([G["int_props"], G["float_props"], G["str_props"]])
        g = glx.Graph(og["row_ptr"], og["col"], og["eid"]) if og is not None else None
        tab = glx.CondTable(cand, w, keys)
        for cc in range(20):
            want = Oracle().cond_negative_sample(cand, w, keys, props, og, G["req_src"], G["req_dst"], dk, int(G["count"]),
                                                 batch_share=share, unique=unique, seed=9, call_counter=cc)
            got = tab.sample(g, G["req_src"], G["req_dst"], dk, props, int(G["count"]), batch_share=share, unique=unique,
                             seed=9, call_counter=cc)
            assert np.array_equal(got, want), (strategy, cc)

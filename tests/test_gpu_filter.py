"""GPU parity of sampling with a Filter (core/operator/sampler/filter.{h,cc}).

glx_sample_filtered / glx_sample_full_filtered through the C-ABI against
  * the reference's own answers (tests/golden/filtered.npz: Topk / Full, every filter, both paddings),
  * the oracle, bit for bit, for every sampler x filter x padding on random multigraphs
    (host pointers and device pointers, partitioned requests via rng_rows).
"""
import os

import numpy as np
import pytest

import glx
from oracle_bindings import ALL_SAMPLERS, Oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILTERS = {"id_eq": (1, 1), "id_gt": (2, 1), "ts_eq": (1, 2), "ts_gt": (2, 2)}  # (FilterType, FilterField)


@pytest.fixture(scope="module")
def orc():
    return Oracle()


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(GOLD, "filtered.npz")))


def test_filtered_topk_and_full_match_reference_golden(gold):
    g = gold
    dev = glx.Graph.from_edges(g["src"], g["dst"], g["w"], timestamp=g["ts"])
    for name in g["cases"]:
        name = str(name)
        kind, strategy = name.rsplit("_", 3)[0], name.split("_")[2]
        k, pad = int(name.split("_k")[1][0]), int(name[-1])
        ft, ff = FILTERS[kind]
        ids, vals = g[name + "_ids"], g[name + "_values"]
        if strategy == "FullSampler":
            deg, nbr, eid = dev.sample_full_filtered(ids, k, ft, ff, vals, padding_mode=pad, default_neighbor_id=-7)
            assert np.array_equal(deg, g[name + "_deg"]), name
        else:
            nbr, eid = dev.sample_filtered("TopkSampler", ids, k, ft, ff, vals, padding_mode=pad, default_neighbor_id=-7)
        assert np.array_equal(nbr, g[name + "_nbr"]), name
        assert np.array_equal(eid, g[name + "_eid"]), name
    dev.close()


def random_graph(rng, V=400, E=9000, n_dst=60):
    """Multigraph with repeated destinations, hubs, single-neighbour rows, sparse negative ids."""
    src = (rng.zipf(1.6, E) % V).astype(np.int64) * 7 - 300
    src[:40] = np.arange(40, dtype=np.int64) * 7 + 100000  # 40 rows with exactly one neighbour
    dst = rng.integers(0, n_dst, E).astype(np.int64) - 10
    ts = rng.permutation(E).astype(np.int64) * 5 + 77
    w = (rng.random(E) + 0.01).astype(np.float32)
    return src, dst, ts, w


def oracle_graph(orc, dev, src, ts, w):
    rows = np.unique(src)
    deg, col, eid = dev.sample_full(rows, 0)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    og = dict(row_ptr=rp, col=col, eid=eid, weight=w[eid], ids=rows, ts_slot=ts[eid])
    og["indeg_weight"] = orc.in_degree_alias(og)[1]
    return og, rows


def make_values(rng, og, rows, ids, kind):
    pos = {int(v): i for i, v in enumerate(rows)}
    vals = np.zeros(ids.shape[0], np.int64)
    for n, v in enumerate(ids):
        i = pos.get(int(v))
        if i is None or og["row_ptr"][i + 1] == og["row_ptr"][i]:
            vals[n] = rng.integers(-5, 50)
            continue
        a, b = og["row_ptr"][i], og["row_ptr"][i + 1]
        field = og["col"][a:b] if kind.startswith("id") else og["ts_slot"][a:b]
        pick = int(field[rng.integers(0, b - a)])
        vals[n] = pick + int(rng.integers(-1, 2)) if n % 4 else int(np.sort(field)[(b - a) // 2])
    return vals


@pytest.mark.parametrize("pad", [1, 0])
@pytest.mark.parametrize("kind", list(FILTERS))
def test_filtered_samplers_bit_exact_with_oracle(orc, kind, pad):
    rng = np.random.default_rng(list(FILTERS).index(kind) * 2 + pad + 50)
    src, dst, ts, w = random_graph(rng)
    dev = glx.Graph.from_edges(src, dst, w, timestamp=ts)
    dev.enable_in_degree()
    og, rows = oracle_graph(orc, dev, src, ts, w)
    ft, ff = FILTERS[kind]
    import torch
    for trial in range(3):
        ids = np.concatenate([rng.choice(rows, 700), [999999, -1]]).astype(np.int64)
        vals = make_values(rng, og, rows, ids, kind)
        rng_rows = rng.permutation(ids.shape[0]).astype(np.int64) if trial == 2 else None
        for k in (1, 5, 24):
            for name in ALL_SAMPLERS:
                flt = dict(type=ft, field=ff, values=vals, retry_times=trial)
                want = orc.sample_filtered(og, name, ids, k, flt, seed=17 + trial, call_counter=k, padding_mode=pad,
                                           default_neighbor_id=-9, rng_rows=rng_rows)
                got = dev.sample_filtered(name, ids, k, ft, ff, vals, seed=17 + trial, call_counter=k, padding_mode=pad,
                                          default_neighbor_id=-9, retry_times=trial, rng_rows=rng_rows)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, kind, pad, k, trial)
                if k == 5:  # the same request with device pointers
                    t = lambda a: None if a is None else torch.from_numpy(a).cuda()  # noqa: E731
                    gd = dev.sample_filtered(name, t(ids), k, ft, ff, t(vals), seed=17 + trial, call_counter=k,
                                             padding_mode=pad, default_neighbor_id=-9, retry_times=trial,
                                             rng_rows=t(rng_rows))
                    assert np.array_equal(gd[0].cpu().numpy(), want[0]) and np.array_equal(gd[1].cpu().numpy(), want[1])
        for limit in (0, 4):
            want = orc.sample_full_filtered(og, ids, limit, dict(type=ft, field=ff, values=vals), padding_mode=pad,
                                            default_neighbor_id=-9)
            got = dev.sample_full_filtered(ids, limit, ft, ff, vals, padding_mode=pad, default_neighbor_id=-9)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), (kind, pad, limit)
    dev.close()


def test_chunked_requests_match_unchunked(orc, monkeypatch):
    """Large requests are served in chunks of bounded total degree (GLX_FILTER_SPAN_CAP forces tiny ones)."""
    rng = np.random.default_rng(77)
    src, dst, ts, w = random_graph(rng)
    dev = glx.Graph.from_edges(src, dst, w, timestamp=ts)
    dev.enable_in_degree()
    og, rows = oracle_graph(orc, dev, src, ts, w)
    ids = rng.choice(rows, 500).astype(np.int64)
    for kind in ("id_eq", "id_gt", "ts_eq"):
        ft, ff = FILTERS[kind]
        vals = make_values(rng, og, rows, ids, kind)
        for name in ALL_SAMPLERS[1:]:
            glx.tune("filter_span_cap", -1)
            whole = dev.sample_filtered(name, ids, 7, ft, ff, vals, seed=3, call_counter=9)
            glx.tune("filter_span_cap", 90)
            cut = dev.sample_filtered(name, ids, 7, ft, ff, vals, seed=3, call_counter=9)
            want = orc.sample_filtered(og, name, ids, 7, dict(type=ft, field=ff, values=vals), seed=3, call_counter=9)
            assert np.array_equal(cut[0], whole[0]) and np.array_equal(cut[1], whole[1]), (kind, name)
            assert np.array_equal(cut[0], want[0]) and np.array_equal(cut[1], want[1]), (kind, name)
        full = dev.sample_full_filtered(ids, 0, ft, ff, vals)
        glx.tune("filter_span_cap", -1)
        assert all(np.array_equal(a, b) for a, b in zip(full, dev.sample_full_filtered(ids, 0, ft, ff, vals)))
    dev.close()


def test_filter_without_timestamps_uses_the_default_timestamp(orc):
    """GetEdgeTimestamp on a type without timestamps is GLOBAL_FLAG(DefaultTimestamp)
    (memory_edge_storage.cc:113-119): timestamp == default hits every neighbour."""
    rng = np.random.default_rng(3)
    src, dst, ts, w = random_graph(rng, V=50, E=600)
    dev = glx.Graph.from_edges(src, dst, w)
    rows = np.unique(src)[:20]
    vals = np.full(rows.shape[0], -1, np.int64)
    nbr, eid = dev.sample_filtered("TopkSampler", rows, 3, 1, 2, vals, default_neighbor_id=-4, default_timestamp=-1)
    assert (nbr == -4).all() and (eid == -1).all()
    nbr, _ = dev.sample_filtered("TopkSampler", rows, 3, 1, 2, vals + 1, default_neighbor_id=-4, default_timestamp=-1)
    assert np.array_equal(nbr, dev.sample("TopkSampler", rows, 3)[0])
    # set_timestamps on a CSR-built handle
    deg, col, eid = dev.sample_full(np.unique(src), 0)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    csr = glx.Graph(rp, col, eid, weight=w[eid], ids=np.unique(src))
    csr.set_timestamps(ts[eid])
    og = dict(row_ptr=rp, col=col, eid=eid, weight=w[eid], ids=np.unique(src), ts_slot=ts[eid])
    v = make_values(rng, og, np.unique(src), rows, "ts")
    want = orc.sample_filtered(og, "TopkSampler", rows, 4, dict(type=1, field=2, values=v))
    got = csr.sample_filtered("TopkSampler", rows, 4, 1, 2, v)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # no filter: the plain entry points
    a = dev.sample_filtered("RandomSampler", rows, 6, 0, 0, vals, seed=5)
    b = dev.sample("RandomSampler", rows, 6, seed=5)
    assert np.array_equal(a[0], b[0])
    dev.close()
    csr.close()


@pytest.mark.parametrize("indexed", [False, True])
@pytest.mark.parametrize("pad", [glx.PAD_CIRCULAR, glx.PAD_REPLICATE])
def test_id_equal_fast_path_bit_exact_with_oracle(orc, indexed, pad):
    """id == value filters take the closed-form path (hit positions from one look at the row -- a binary
    search in the id-sorted index when it is built, one ballot scan otherwise -- and an implicit reserved
    list); rows it does not serve fall back to the general path.  Hit patterns exercised: none, one, several,
    more than kMaxHits = 8 parallel edges (hub rows over 60 distinct destinations), hits at the first / last
    positions, single-neighbour rows (everything filtered), k beyond the register budget of the
    without-replacement path (k > 32)."""
    rng = np.random.default_rng(321 + pad)
    src, dst, ts, w = random_graph(rng, n_dst=60)
    dst[::3] = rng.integers(1000, 5000, dst[::3].shape[0])  # a third of the edges go to (mostly) unique ids
    dev = glx.Graph.from_edges(src, dst, w, timestamp=ts)
    dev.enable_in_degree()
    if indexed:
        dev.enable_id_index()
    og, rows = oracle_graph(orc, dev, src, ts, w)
    pos = {int(v): i for i, v in enumerate(rows)}
    ids = np.concatenate([rng.choice(rows, 900), [999999, -1]]).astype(np.int64)
    vals = rng.integers(-5, 5000, ids.shape[0]).astype(np.int64)
    for n, v in enumerate(ids):
        i = pos.get(int(v))
        if i is None:
            continue
        a, b = og["row_ptr"][i], og["row_ptr"][i + 1]
        if b > a and n % 5:
            vals[n] = og["col"][[a, b - 1, rng.integers(a, b), rng.integers(a, b)][n % 4]]
    flt = dict(type=glx.FILTER_EQUAL, field=glx.FILTER_FIELD_ID, values=vals)
    rev = np.arange(ids.shape[0], dtype=np.int64)[::-1].copy()  # a shard's slice: streams by original row
    for name in ("TopkSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "InDegreeSampler"):
        for k in (1, 5, 12, 40):
            for rr in (None, rev):
                want = orc.sample_filtered(og, name, ids, k, flt, seed=23, call_counter=k, padding_mode=pad,
                                           default_neighbor_id=-3, rng_rows=rr)
                got = dev.sample_filtered(name, ids, k, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, vals, seed=23,
                                          call_counter=k, padding_mode=pad, default_neighbor_id=-3, rng_rows=rr)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, k, pad, indexed)
    dev.close()


@pytest.mark.parametrize("indexed", [False, True])
def test_id_equal_rows_with_long_runs_of_parallel_edges(orc, indexed):
    """Hub rows of a multigraph: hundreds of parallel edges to the filtered id (the run of hits is read in place
    from the id-sorted index, in whatever order the segmented sort left it), every neighbour filtered, all hits
    in the last positions, all in the first ones."""
    rng = np.random.default_rng(99)
    rows = []
    for v, (same, others) in enumerate([(150, 50), (30, 0), (300, 3), (9, 500), (1, 1), (64, 64), (1000, 7)]):
        d = np.concatenate([np.full(same, 7, np.int64), rng.integers(100, 10**6, others)])
        rows.append((np.full(d.shape[0], v * 11 + 5, np.int64), d))
    src = np.concatenate([r[0] for r in rows])
    dst = np.concatenate([r[1] for r in rows])
    perm = rng.permutation(src.shape[0])
    src, dst = src[perm], dst[perm]
    # weights place the parallel edges first (heaviest), last (lightest) or anywhere, row by row
    w = rng.random(src.shape[0]).astype(np.float32) + 0.01
    hit = dst == 7
    w[hit & (src == 5)] += 10.0        # row 0: hits first
    w[hit & (src == 27)] *= 1e-3       # row 2: hits last
    w[hit & (src == 71)] += 10.0       # row 6: 1000 hits first
    ts = rng.permutation(src.shape[0]).astype(np.int64)
    dev = glx.Graph.from_edges(src, dst, w, timestamp=ts)
    dev.enable_in_degree()
    if indexed:
        dev.enable_id_index()
    og, ids = oracle_graph(orc, dev, src, ts, w)
    ids = np.concatenate([ids, ids]).astype(np.int64)
    vals = np.full(ids.shape[0], 7, np.int64)
    vals[ids.shape[0] // 2:] = og["col"][og["row_ptr"][:-1]]  # second half: filter each row's first neighbour
    flt = dict(type=glx.FILTER_EQUAL, field=glx.FILTER_FIELD_ID, values=vals)
    for name in ("TopkSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "InDegreeSampler"):
        for k in (1, 4, 10, 32, 40):
            for pad in (glx.PAD_CIRCULAR, glx.PAD_REPLICATE):
                want = orc.sample_filtered(og, name, ids, k, flt, seed=5, call_counter=k, padding_mode=pad,
                                           default_neighbor_id=-3)
                got = dev.sample_filtered(name, ids, k, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, vals, seed=5,
                                          call_counter=k, padding_mode=pad, default_neighbor_id=-3)
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, k, pad, indexed)
    dev.close()


@pytest.mark.parametrize("kind", list(FILTERS))
def test_alias_samplers_share_one_table_per_vertex_and_value(orc, kind, monkeypatch):
    """EdgeWeight / InDegree with a filter: request rows naming the same (vertex, filter value) pair draw from ONE
    table built for the pair (large requests only by default; GLX_FILTER_DEDUP_MIN_ROWS = 1 forces it here) --
    same answers as the per-row build, also when the pairs are served in several chunks and for a shard's slice."""
    rng = np.random.default_rng(400 + list(FILTERS).index(kind))
    src, dst, ts, w = random_graph(rng)
    dev = glx.Graph.from_edges(src, dst, w, timestamp=ts)
    dev.enable_in_degree()
    dev.enable_id_index()
    og, rows = oracle_graph(orc, dev, src, ts, w)
    ft, ff = FILTERS[kind]
    ids = np.concatenate([rng.choice(rows[:60], 1500), [999999, -1, 999999]]).astype(np.int64)
    vals = make_values(rng, og, rows, ids, kind)
    vals[::2] = vals[0]  # many rows share their value as well
    rev = rng.permutation(ids.shape[0]).astype(np.int64)
    for cap in (None, "500"):
        for name in ("EdgeWeightSampler", "InDegreeSampler"):
            for rr in (None, rev):
                flt = dict(type=ft, field=ff, values=vals)
                want = orc.sample_filtered(og, name, ids, 6, flt, seed=31, call_counter=2, default_neighbor_id=-4,
                                           rng_rows=rr)
                glx.tune("filter_dedup_min_rows", 0)
                per_row = dev.sample_filtered(name, ids, 6, ft, ff, vals, seed=31, call_counter=2,
                                              default_neighbor_id=-4, rng_rows=rr)
                glx.tune("filter_dedup_min_rows", 1)
                if cap:
                    glx.tune("filter_span_cap", int(cap))
                shared = dev.sample_filtered(name, ids, 6, ft, ff, vals, seed=31, call_counter=2,
                                             default_neighbor_id=-4, rng_rows=rr)
                glx.tune("filter_span_cap", -1)
                glx.tune("filter_dedup_min_rows", -1)
                assert np.array_equal(per_row[0], want[0]) and np.array_equal(per_row[1], want[1]), (name, kind)
                assert np.array_equal(shared[0], want[0]) and np.array_equal(shared[1], want[1]), (name, kind, cap)
    dev.close()


def test_full_sampler_with_an_empty_response_on_device_pointers():
    """Every requested vertex unknown or without out-edges: the response has no values and an empty tensor has no buffer
    to point at (its data pointer is NULL) -- both FullSampler entry points accept that for device pointers too (the
    offsets, which live on the device, say the response is empty)."""
    import torch
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    g = glx.Graph.from_edges(t(np.array([1, 1, 2], np.int64)), t(np.array([5, 6, 7], np.int64)),
                             t(np.array([0.5, 0.25, 1.0], np.float32)), timestamp=t(np.array([3, 4, 5], np.int64)))
    ids = t(np.array([5, 6, 99, -4], np.int64))  # destinations and strangers: no out-edges
    deg, nbr, eid = g.sample_full(ids, 0)
    assert deg.tolist() == [0, 0, 0, 0] and nbr.numel() == 0 and eid.numel() == 0
    vals = t(np.array([5, 5, 5, 5], np.int64))
    deg, nbr, eid = g.sample_full_filtered(ids, 3, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, vals)
    assert deg.tolist() == [0, 0, 0, 0] and nbr.numel() == 0 and eid.numel() == 0
    deg, nbr, eid = g.sample_full_filtered(ids.cpu().numpy(), 3, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, vals.cpu().numpy())
    assert deg.tolist() == [0, 0, 0, 0] and nbr.size == 0

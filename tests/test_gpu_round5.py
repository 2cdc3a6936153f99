"""Round 5 additions on the device path."""
import numpy as np
import pytest

import glx
from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("default_weight", [0.0, 0.5, 3.0])
def test_edge_weight_sampler_on_an_unweighted_graph_is_the_reference_with_default_weights(default_weight):
    """An unweighted edge type answers GetEdgeWeight with GLOBAL_FLAG(DefaultWeight) for every edge
    (memory_edge_storage.cc:97-103), so the reference's EdgeWeightSampler builds its alias row from a constant -- all
    NaN for the default 0.0 (alias_method.cc:73-83), where a draw is then idx = (int)(float)u[0, deg - 1).
    glx_graph_enable_default_weight gives the device graph exactly those tables and draws (the oracle restates
    AliasMethod; tests/test_oracle_golden.py pins it to the reference's on degenerate rows)."""
    orc = Oracle()
    rng = np.random.default_rng(5)
    degs = np.concatenate([rng.integers(0, 9, 400), [1, 2, 97, 300, 5000]]).astype(np.int64)
    rp = np.concatenate([[0], np.cumsum(degs)]).astype(np.int64)
    E = int(rp[-1])
    col = rng.integers(0, 10000, E).astype(np.int64)
    eid = rng.permutation(E).astype(np.int64)
    dev = glx.Graph(rp, col, eid, None)
    q = rng.integers(0, degs.size, 700).astype(np.int64)
    with pytest.raises(Exception):  # not before it is asked for: no silent guess at the weights
        dev.sample("EdgeWeightSampler", q, 4, seed=1, call_counter=1)
    dev.enable_default_weight(default_weight)
    w = np.full(E, default_weight, np.float32)
    oprob, oalias = orc.alias_build(rp, w)
    prob, alias = dev.export_alias()
    assert np.array_equal(prob.view(np.uint32), oprob.view(np.uint32)) and np.array_equal(alias, oalias)
    if default_weight == 0.0:
        assert np.isnan(prob[rp[:-1][degs > 0]]).all()
    og = dict(row_ptr=rp, col=col, eid=eid, weight=w, alias=(oprob, oalias))
    for k, pad in ((7, glx.PAD_CIRCULAR), (3, glx.PAD_REPLICATE)):
        n, e = dev.sample("EdgeWeightSampler", q, k, seed=11, call_counter=2, padding_mode=pad, default_neighbor_id=-1)
        on, oe = orc.sample(og, "EdgeWeightSampler", q, k, seed=11, call_counter=2, padding_mode=pad, default_neighbor_id=-1)
        assert np.array_equal(n, on) and np.array_equal(e, oe)
    # the other samplers of the same handle are untouched by the added weights
    n, e = dev.sample("RandomSampler", q, 5, seed=11, call_counter=3)
    og_unweighted = dict(row_ptr=rp, col=col, eid=eid, weight=None, alias=None)
    on, oe = orc.sample(og_unweighted, "RandomSampler", q, 5, seed=11, call_counter=3)
    assert np.array_equal(n, on) and np.array_equal(e, oe)
    assert dev.enable_default_weight(9.0) is dev  # a second call is a no-op
    prob2, _ = dev.export_alias()
    assert np.array_equal(prob2.view(np.uint32), prob.view(np.uint32))


def _segment_cases(rng):
    """(node count, segment count, segment_ids) shapes that exercise every branch of the one-pass bookkeeping."""
    cases = []
    for n, sg in ((0, 0), (0, 5), (1, 1), (7, 3), (4096, 512), (5000, 500), (100000, 10000), (1023, 1), (3, 4000)):
        if sg == 0:
            cases.append((n, sg, np.zeros(n, np.int32), "empty"))
            continue
        if n % sg == 0 and n > 0:
            f = n // sg
            cases.append((n, sg, (np.arange(n) // f).astype(np.int32), "uniform spelled out"))
            swapped = (np.arange(n) // f).astype(np.int32)
            if n > f + 1 and sg > 1:
                swapped[f - 1] = 1  # same count per segment on average, but not the dense layout
                cases.append((n, sg, swapped, "divisible but not uniform"))
        ragged = np.sort(rng.integers(0, sg, n)).astype(np.int32)
        cases.append((n, sg, ragged, "ragged"))
        if n > 2:
            gaps = np.sort(rng.choice(sg, min(sg, 3), replace=False))[rng.integers(0, min(sg, 3), n)]
            cases.append((n, sg, np.sort(gaps).astype(np.int32), "long runs of empty segments"))
            bad = ragged.copy()
            at = int(rng.integers(1, n))
            bad[at] = -1 if at % 3 == 0 else (sg if at % 3 == 1 else max(int(bad[at - 1]) - 1, -5))
            cases.append((n, sg, bad, "violation at %d" % at))
            neg = ragged.copy()
            neg[n // 2] = -2  # the ids after it are in range again: nothing of them may be counted (or written anywhere)
            cases.append((n, sg, neg, "a negative id in the middle"))
            late = np.full(n, sg - 1, np.int32)  # every id in the LAST segment: everything before it is empty
            cases.append((n, sg, late, "all in the last segment"))
            early = np.zeros(n, np.int32)
            cases.append((n, sg, early, "all in the first segment"))
    return cases


@pytest.mark.parametrize("dim", [256, 64, 7])
def test_explicit_segment_ids_one_pass_bookkeeping_equals_the_oracle(dim):
    """AggregatingRequest's cursor (aggregating_request.cc:86-105) over an explicit segment_ids tensor: the one-pass
    device bookkeeping (validity, segment starts, 'this is just the dense layout' detection, the fix-up for violations
    and for long runs of empty segments) gives the oracle's embeddings and counts bit for bit, for every aggregator, with
    device and host pointers -- and the same as segment_ids = None when the tensor spells out the dense layout."""
    import torch
    orc = Oracle()
    rng = np.random.default_rng(dim)
    V = 3000
    X = rng.standard_normal((V, dim)).astype(np.float32)
    feats = glx.Features(X)
    dev = torch.device("cuda", 0)
    for n, sg, seg, what in _segment_cases(rng):
        ids = rng.integers(-2, V + 2, n).astype(np.int64)
        for name in glx.AGGREGATOR_IDS:
            want_e, want_c = orc.aggregate(X, name, ids, seg, sg, default_attr=0.25)
            got_e, got_c = feats.aggregate(name, ids, seg, sg, default_attr=0.25)
            assert np.array_equal(got_c, want_c), (what, name, n, sg)
            assert np.array_equal(got_e.view(np.uint32), want_e.view(np.uint32)), (what, name, n, sg)
        if n > 0:
            t_e, t_c = feats.aggregate("MeanAggregator", torch.from_numpy(ids).to(dev), torch.from_numpy(seg).to(dev), sg,
                                       default_attr=0.25)
            want_e, want_c = orc.aggregate(X, "MeanAggregator", ids, seg, sg, default_attr=0.25)
            assert np.array_equal(t_c.cpu().numpy(), want_c) and np.array_equal(t_e.cpu().numpy().view(np.uint32), want_e.view(np.uint32)), what
        if what == "uniform spelled out":
            e0, c0 = feats.aggregate("SumAggregator", ids, None, sg, default_attr=0.25)
            e1, c1 = feats.aggregate("SumAggregator", ids, seg, sg, default_attr=0.25)
            assert np.array_equal(c0, c1) and np.array_equal(e0.view(np.uint32), e1.view(np.uint32))
    feats.close()


@pytest.mark.parametrize("base,step,n", [(3, 8, 5000), (0, 2, 4096), (-70, 7, 333), (5, 1, 100), (10 ** 15, 10 ** 12, 50), (9, 4, 2),
                                         (2 ** 62, 2 ** 60, 3)])
def test_feature_tables_with_arithmetic_ids_need_no_hash_table(base, step, n):
    """Shard r of a dense id space under the reference's ownership rule llabs(id) % P holds ids r, r + P, r + 2P, ...
    (hash_partitioner.h:88-92): such a table translates ids by arithmetic.  Same answers as the hash table
    (glx.tune("idmap_hash_only", 1): A/B in one process) and as the oracle, for ids inside, between, before and
    after the progression; ids that only look arithmetic at the start keep the table."""
    import torch
    orc = Oracle()
    rng = np.random.default_rng(n)
    D = 20
    X = rng.standard_normal((n, D)).astype(np.float32)
    ids = (base + step * np.arange(n, dtype=object)).astype(np.int64)
    probe = np.concatenate([ids[rng.integers(0, n, 400)], ids[:50] + 1, ids[:50] - 1, [base - step, int(ids[-1]) + step if int(ids[-1]) + step < 2 ** 63 else 0,
                                                                                       -1, 0, 2 ** 63 - 1, -2 ** 63]]).astype(np.int64)
    seg = np.sort(rng.integers(0, 40, probe.shape[0])).astype(np.int32)
    glx.tune("idmap_hash_only", 1)
    hashed = glx.Features(X, ids=ids)
    glx.tune("idmap_hash_only", -1)
    arith = glx.Features(X, ids=ids)
    try:
        for name in glx.AGGREGATOR_IDS:
            we, wc = orc.aggregate(X, name, probe, seg, 40, default_attr=1.5, ids=ids)
            for f in (hashed, arith):
                e, c = f.aggregate(name, probe, seg, 40, default_attr=1.5)
                assert np.array_equal(c, wc) and np.array_equal(e.view(np.uint32), we.view(np.uint32)), name
        a, b = hashed.lookup(probe, default_attr=-2.0), arith.lookup(probe, default_attr=-2.0)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        dev = torch.device("cuda", 0)
        ta = arith.lookup(torch.from_numpy(probe).to(dev), default_attr=-2.0)
        assert np.array_equal(ta.cpu().numpy().view(np.uint32), a.view(np.uint32))
    finally:
        hashed.close()
        arith.close()
    # almost arithmetic: one id off -- the table is kept and still right
    if n > 3:
        odd = ids.copy()
        odd[n // 2] += 1
        f = glx.Features(X, ids=odd)
        q = np.array([odd[n // 2], ids[n // 2], odd[0], odd[-1]], np.int64)
        got = f.lookup(q, default_attr=7.0)
        want = np.stack([X[n // 2], np.full(D, 7.0, np.float32), X[0], X[-1]])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        f.close()

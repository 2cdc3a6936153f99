"""GPU tests of round 4: the grouped aggregation kernel (ids fetched once per segment, coalesced, and handed to the
lanes with cross-lane reads; aggregator.cc:25-59's order kept) under every launch shape glx_tune can select -- XCD-affine
column slices, segments per group, rows in flight, 8-byte loads, the round-3 kernel -- bit for bit against the oracle."""
import numpy as np
import pytest
import torch

import glx
from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu
AGGREGATORS = ["SumAggregator", "MeanAggregator", "MaxAggregator", "MinAggregator", "ProdAggregator"]
KNOBS = ("agg_legacy", "agg_unroll", "agg_segs", "agg_xcd_slices", "agg_occupancy", "agg_store", "agg_slices", "agg_mfma")


def beq(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture
def knobs():
    def set_knobs(**kw):
        for k in KNOBS:
            glx.tune(k, kw.get(k, 0))
    yield set_knobs
    set_knobs()


SHAPES = [dict(), dict(agg_legacy=1), dict(agg_xcd_slices=1), dict(agg_xcd_slices=2), dict(agg_xcd_slices=4),
          dict(agg_xcd_slices=8), dict(agg_segs=3), dict(agg_segs=7, agg_xcd_slices=2), dict(agg_unroll=6),
          dict(agg_unroll=8, agg_xcd_slices=4), dict(agg_unroll=12), dict(agg_unroll=15), dict(agg_occupancy=4),
          dict(agg_slices=2), dict(agg_store=1), dict(agg_store=1, agg_xcd_slices=2)]


@pytest.mark.parametrize("D", [32, 64, 96, 128, 160, 256, 320, 512, 1024])
def test_every_launch_shape_is_bit_identical_on_ragged_segments(knobs, D):
    """Ragged segments (empty ones, one of 700 ids: several chunks), unknown ids (default rows), dense and hashed ids."""
    rng = np.random.default_rng(D)
    V = 1500
    X = (rng.standard_normal((V, D)) * 4).astype(np.float32)
    X[rng.random((V, D)) < 0.02] = -50.0  # Max's -37 initialiser
    raw = np.arange(V, dtype=np.int64) * 3 + 17
    Sg = 301
    sizes = rng.integers(0, 12, Sg)
    sizes[[0, 1, 100, Sg - 1]] = 0
    sizes[7] = 700
    sizes[8] = 64
    sizes[9] = 65
    seg = np.repeat(np.arange(Sg, dtype=np.int32), sizes)
    orc = Oracle()
    for ids_kind in ("dense", "hashed"):
        f = glx.Features(X, ids=(raw if ids_kind == "hashed" else None))
        pool = raw if ids_kind == "hashed" else np.arange(V, dtype=np.int64)
        nid = pool[rng.integers(0, V, seg.shape[0])].copy()
        nid[rng.random(seg.shape[0]) < 0.05] = -99
        want = {name: orc.aggregate(X, name, nid, seg, Sg, 2.5, ids=(raw if ids_kind == "hashed" else None)) for name in AGGREGATORS}
        for shape in SHAPES:
            knobs(**shape)
            for name in AGGREGATORS:
                emb, cnt = f.aggregate(name, nid, seg, Sg, default_attr=2.5)
                assert np.array_equal(cnt, want[name][1]), (name, D, ids_kind, shape)
                assert beq(emb, want[name][0]), (name, D, ids_kind, shape)


@pytest.mark.parametrize("fanout", [1, 5, 10, 15, 16, 25, 64, 70, 130])
@pytest.mark.parametrize("D", [64, 128, 256])
def test_dense_sampler_responses_every_fanout(knobs, D, fanout):
    """segment_ids = NULL: segment i = ids [i f, (i + 1) f) -- batches, tails and chunk reloads of every group width."""
    rng = np.random.default_rng(D * 1000 + fanout)
    V, Sg = 4000, 777
    X = rng.standard_normal((V, D)).astype(np.float32)
    ids = rng.integers(-3, V + 3, Sg * fanout).astype(np.int64)
    seg = (np.arange(ids.shape[0]) // fanout).astype(np.int32)
    f = glx.Features(torch.from_numpy(X).cuda(), device=0)
    d_ids = torch.from_numpy(ids).cuda()
    orc = Oracle()
    for name in ("SumAggregator", "MaxAggregator", "MeanAggregator"):
        oemb, ocnt = orc.aggregate(X, name, ids, seg, Sg, default_attr=-0.5)
        for shape in SHAPES:
            knobs(**shape)
            emb, cnt = f.aggregate(name, d_ids, None, Sg, default_attr=-0.5)
            assert np.array_equal(cnt.cpu().numpy(), ocnt), (name, shape)
            assert beq(emb.cpu().numpy(), oemb), (name, shape)


def test_default_launch_slices_a_big_request_and_stays_bit_identical(knobs):
    """Requests of >= 4 M ids take the XCD-affine two-slice launch by default (no knob): same bits as the unsliced one
    and as the round-3 kernel."""
    rng = np.random.default_rng(4)
    V, D, f = 200_000, 128, 10
    Sg = (4 << 20) // f + 3
    X = torch.from_numpy(rng.standard_normal((V, D)).astype(np.float32)).cuda()
    ids = torch.from_numpy(rng.integers(-2, V + 2, Sg * f).astype(np.int64)).cuda()
    feats = glx.Features(X, device=0)
    out = {}
    for label, shape in (("default", dict()), ("whole rows", dict(agg_xcd_slices=1)), ("round 3", dict(agg_legacy=1))):
        knobs(**shape)
        emb, cnt = feats.aggregate("SumAggregator", ids, None, Sg)
        torch.cuda.synchronize()
        out[label] = (emb.clone(), cnt.clone())
    for label in ("whole rows", "round 3"):
        assert torch.equal(out["default"][0].view(torch.int32), out[label][0].view(torch.int32)), label
        assert torch.equal(out["default"][1], out[label][1]), label
    # and a slice of it against the oracle
    n = 5000
    h_ids = ids[:n * f].cpu().numpy()
    oemb, ocnt = Oracle().aggregate(X.cpu().numpy(), "SumAggregator", h_ids, (np.arange(n * f) // f).astype(np.int32), n)
    assert beq(out["default"][0][:n].cpu().numpy(), oemb)


def test_unknown_knob_is_an_error():
    with pytest.raises(glx.GlxError):
        glx.tune("no_such_knob", 1)


def _fuzz_cases(n):
    import os
    first = int(os.environ.get("GLX_FUZZ_FIRST", "0"))
    return list(range(first, first + int(os.environ.get("GLX_FUZZ_CASES", str(n)))))


@pytest.mark.parametrize("case", _fuzz_cases(40))
def test_launch_shape_fuzz(knobs, case):
    """Random feature widths (1 .. 1100, aligned or not), segment layouts (dense fanouts and ragged, with empty and very
    long segments), table sizes, id kinds and random settings of every launch knob at once -- against the oracle, bit
    for bit, for the five aggregators."""
    rng = np.random.default_rng(8800 + case)
    D = int(rng.choice([1, 3, 4, 8, 20, 32, 60, 64, 100, 128, 132, 256, 260, 384, 512, 1000, 1024, 1100]))
    V = int(rng.choice([1, 7, 300, 5000]))
    X = (rng.standard_normal((V, D)) * 3).astype(np.float32)
    X[rng.random((V, D)) < 0.03] = -60.0
    hashed = bool(rng.integers(0, 2))
    raw = (rng.permutation(V * 4)[:V] - V).astype(np.int64)
    pool = raw if hashed else np.arange(V, dtype=np.int64)
    if rng.integers(0, 2):  # a dense sampler response: uniform fanout, no segment ids
        Sg = int(rng.choice([1, 5, 257, 3000]))
        fan = int(rng.choice([1, 2, 9, 10, 11, 25, 63, 64, 65, 200]))
        seg, n = None, Sg * fan
        seg_np = np.repeat(np.arange(Sg, dtype=np.int32), fan)
    else:
        Sg = int(rng.choice([1, 4, 300]))
        sizes = rng.integers(0, int(rng.choice([3, 15, 90])), Sg)
        if Sg > 2:
            sizes[int(rng.integers(0, Sg))] = int(rng.choice([0, 129, 1500]))
        seg_np = np.repeat(np.arange(Sg, dtype=np.int32), sizes)
        seg, n = seg_np, int(seg_np.shape[0])
    nid = pool[rng.integers(0, V, n)].copy() if n else np.zeros(0, np.int64)
    nid[rng.random(n) < 0.04] = 10 ** 9 + 7  # unknown ids: the default row
    shape = dict(agg_legacy=int(rng.random() < 0.15), agg_unroll=int(rng.choice([0, 6, 8, 10, 12, 15])),
                 agg_segs=int(rng.choice([0, 0, 1, 2, 5, 12])), agg_xcd_slices=int(rng.choice([0, 1, 2, 4, 8])),
                 agg_occupancy=int(rng.choice([0, 0, 3, 6])), agg_store=int(rng.integers(0, 2)),
                 agg_slices=int(rng.choice([0, 0, 2, 4])))
    knobs(**shape)
    f = glx.Features(X, ids=(raw if hashed else None))
    orc = Oracle()
    dflt = float(rng.choice([0.0, -1.5, 7.25]))
    for name in AGGREGATORS:
        want = orc.aggregate(X, name, nid, seg_np, Sg, dflt, ids=(raw if hashed else None))
        emb, cnt = f.aggregate(name, nid, seg, Sg, default_attr=dflt)
        assert np.array_equal(cnt, want[1]), (case, name, D, shape)
        assert beq(emb, want[0]), (case, name, D, shape)

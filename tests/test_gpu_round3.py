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
    os.environ["GLX_AGG_MFMA"] = "1"
    yield
    os.environ.pop("GLX_AGG_MFMA", None)


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

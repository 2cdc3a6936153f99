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

"""CPU test of the full-size checker's host logic (tests/headline_check.py): with the oracle itself playing
the device on a whole small graph, the row-subset check must pass -- and must notice one changed slot."""
import numpy as np
import pytest
import torch

import synth
from headline_check import check_step
from oracle_bindings import SAMPLERS, Oracle


@pytest.mark.parametrize("name", SAMPLERS)
def test_subset_check_agrees_with_whole_graph_oracle_and_detects_a_flip(name):
    orc = Oracle()
    V, E, D, k1, k2, B = 5000, 80000, 16, 5, 4, 512
    src, dst, w = synth.rmat_edges_torch(V, E, 4, torch.device("cpu"), weighted=True)
    rp, col, eid, ws = synth.csr_numpy(src.numpy(), dst.numpy(), w.numpy(), V)
    g = dict(row_ptr=rp, col=col, eid=eid, weight=ws, alias=orc.alias_build(rp, ws))
    X = np.random.default_rng(0).random((V, D), dtype=np.float32)
    seeds = torch.from_numpy(np.random.default_rng(1).integers(0, V, B))
    n1, e1 = orc.sample(g, name, seeds.numpy(), k1, seed=3, call_counter=1)
    n2, e2 = orc.sample(g, name, n1.reshape(-1), k2, seed=3, call_counter=2)
    emb2, cnt2 = orc.aggregate(X, "MaxAggregator", n2.reshape(-1), (np.arange(n2.size) // k2).astype(np.int32), n1.size)
    emb1, cnt1 = orc.aggregate(X, "MaxAggregator", n1.reshape(-1), (np.arange(n1.size) // k1).astype(np.int32), B)
    T = torch.from_numpy
    out = dict(n1=T(n1), e1=T(e1), n2=T(n2), e2=T(e2), emb2=T(emb2), cnt2=T(cnt2), emb1=T(emb1), cnt1=T(cnt1))
    hubs = np.argsort(-np.diff(rp))[:10]
    kw = dict(rows_hop1=B, rows_hop2=n1.size, segments=n1.size, hub_ids=hubs)
    r = check_step((src, dst, w), lambda ids: T(X)[ids], name, (k1, k2), "MaxAggregator", seeds, out, 3, (1, 2), **kw)
    assert r["ok"], r
    out["e2"] = out["e2"].clone()
    out["e2"][7, 1] += 1
    r = check_step((src, dst, w), lambda ids: T(X)[ids], name, (k1, k2), "MaxAggregator", seeds, out, 3, (1, 2), **kw)
    assert r["mismatches"] == ["hop-2 sample"], r

"""GPU tests: BIT-EXACT oracle parity at the size the headline metric is quoted on (BASELINE.json
configs[2]: RMAT 10 M nodes / 100 M edges, fanout [25, 10], dim 256, 65,536 seeds per step).

tests/test_gpu_fullsize.py checks size-independent PROPERTIES of the same requests; here the oracle
(oracle/glx_oracle.c, pinned to the reference by tests/test_oracle_golden.py) answers thousands of the
request rows of the full-size step itself, on rows cut from the raw edge list (tests/headline_check.py),
and the device store's hub rows and alias tables are compared entry by entry.
Reference: edge_weight_sampler.cc:31-92, alias_method.cc:57-124, random_sampler.cc:33-76,
random_without_replacement_sampler.cc:31-75, topk_sampler.cc:29-68, memory_adj_matrix.cc:105-125,
aggregator.cc:25-86, sum_/mean_/max_aggregator.cc."""
import numpy as np
import pytest
import torch

import glx
import synth
from headline_check import check_step, hub_rows_equal
from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu
V, E, D, B0, K1, K2 = 10_000_000, 100_000_000, 256, 65536, 25, 10


@pytest.fixture(scope="module")
def c3():
    dev = torch.device("cuda", 0)
    src, dst, w = synth.rmat_edges_torch(V, E, 4, dev, weighted=True)
    g = glx.Graph.from_edges(src, dst, w, device=0)  # the device build, as bench.py does it
    X = synth.features_torch(V, D, 5, dev)
    f = glx.Features(X, device=0)
    pool = torch.unique(src)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    seeds = pool[torch.randint(0, pool.shape[0], (B0,), generator=gen, device=dev)]
    hubs = torch.topk(torch.bincount(src, minlength=V), 100).indices.cpu().numpy()
    return dict(dev=dev, edges=(src, dst, w), g=g, X=X, f=f, seeds=seeds, hubs=hubs)


@pytest.mark.parametrize("name", list(glx.SAMPLER_IDS))
def test_full_size_step_samplers_equal_oracle(c3, name):
    c = c3
    n1, e1 = c["g"].sample(name, c["seeds"], K1, seed=42, call_counter=8)
    n2, e2 = c["g"].sample(name, n1.view(-1), K2, seed=42, call_counter=9)
    torch.cuda.synchronize()
    r = check_step(c["edges"], None, name, (K1, K2), None, c["seeds"], dict(n1=n1, e1=e1, n2=n2, e2=e2),
                   seed=42, call_counters=(8, 9), rows_hop1=4096, rows_hop2=8192, hub_ids=c["hubs"])
    assert r["ok"], r
    assert r["rows_hop1"] >= 4096 and r["rows_hop2"] >= 8192


@pytest.mark.parametrize("agg", ["SumAggregator", "MeanAggregator", "MaxAggregator"])
def test_full_size_step_aggregators_equal_oracle(c3, agg):
    c = c3
    n1, e1 = c["g"].sample("EdgeWeightSampler", c["seeds"], K1, seed=42, call_counter=8)
    n2, e2 = c["g"].sample("EdgeWeightSampler", n1.view(-1), K2, seed=42, call_counter=9)
    emb2, cnt2 = c["f"].aggregate(agg, n2.view(-1), None, B0 * K1)
    emb1, cnt1 = c["f"].aggregate(agg, n1.view(-1), None, B0)
    torch.cuda.synchronize()
    r = check_step(c["edges"], lambda ids: c["X"][ids], "EdgeWeightSampler", (K1, K2), agg, c["seeds"],
                   dict(n1=n1, e1=e1, n2=n2, e2=e2, emb2=emb2, cnt2=cnt2, emb1=emb1, cnt1=cnt1),
                   seed=42, call_counters=(8, 9), rows_hop1=256, rows_hop2=256, segments=65536)
    assert r["ok"], r
    assert r["segments_hop2"] == 65536 and r["segments_hop1"] == 8192


def test_full_size_hub_rows_and_alias_tables_equal_oracle(c3):
    """The 100 largest rows of the device store (out-degrees up to ~10^5: weight ties, the longest LIFO alias
    builds) against the oracle's sort + alias build of the same rows from the raw edge list."""
    rows_ok, alias_ok, edges = hub_rows_equal(Oracle(), c3["g"], c3["edges"], c3["hubs"])
    assert rows_ok, "device row order / edge ids of the hub rows differ from the oracle's Build()"
    assert alias_ok, "device alias tables of the hub rows differ from AliasMethod::Build"
    assert edges > 100_000

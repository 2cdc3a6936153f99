"""GPU tests: BIT-EXACT oracle parity at the FULL sizes of BASELINE.json configs[1], [3] and [4] (VERDICT r05 next-4;
tests/test_gpu_fullsize_oracle.py does the same for configs[2], the headline).

tests/test_gpu_fullsize_configs.py checks size-independent properties of these steps with torch; here the oracle
(oracle/glx_oracle.c, pinned to the reference by tests/test_oracle_golden.py / test_oracle_refseq.py) answers thousands
of the request rows and aggregated segments of the full-size step itself, on sub-graphs cut from the RAW edge list
(tests/headline_check.py: the device build -- row order, edge ids, weight sort -- is inside what is checked).  This is
what `bench.py --verify-oracle` runs after its timed region, as tests the driver counts.

  C2  RMAT 2.4 M / 62 M, RandomWithoutReplacementSampler [15, 10], Sum and Mean, dim 128
      (random_without_replacement_sampler.cc:31-75, circular_padder.h:46-63, sum_aggregator.cc:25-33,
      mean_aggregator.cc:26-61)
  C4  RMAT 111 M / 1.6 B on ONE GPU, RandomSampler [20, 15], Mean, dim 128 (random_sampler.cc:33-76)
  C5  user-item-shop, 3 weighted edge types (300 M / 100 M / 100 M edges), TopkSampler k = 10 / 10 / 5 per type,
      type-wise Sum, dim 256 (topk_sampler.cc:29-68, memory_adj_matrix.cc:105-125)
"""
import numpy as np
import pytest
import torch

import glx
import synth
from headline_check import _pick, check_aggregate, check_sample, check_step
from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu
B0 = 65536


def _two_hop(g, f, sampler, aggs, seeds, k1, k2, seed, ccs):
    n1, e1 = g.sample(sampler, seeds, k1, seed=seed, call_counter=ccs[0])
    n2, e2 = g.sample(sampler, n1.view(-1), k2, seed=seed, call_counter=ccs[1])
    outs = {}
    for agg in aggs:
        emb2, cnt2 = f.aggregate(agg, n2.view(-1), None, B0 * k1)
        emb1, cnt1 = f.aggregate(agg, n1.view(-1), None, B0)
        outs[agg] = dict(n1=n1, e1=e1, n2=n2, e2=e2, emb2=emb2, cnt2=cnt2, emb1=emb1, cnt1=cnt1)
    torch.cuda.synchronize()
    return outs


def test_c2_full_size_step_equals_oracle():
    V, E, D, K1, K2 = 2_400_000, 62_000_000, 128, 15, 10
    dev = torch.device("cuda", 0)
    src, dst, _ = synth.rmat_edges_torch(V, E, 2, dev, weighted=False)
    g = glx.Graph.from_edges(src, dst, None)
    X = synth.features_torch(V, D, 3, dev)
    f = glx.Features(X)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    seeds = torch.randint(0, V, (B0,), generator=gen, device=dev)  # incl. vertices without out-edges: default fill
    name = "RandomWithoutReplacementSampler"
    outs = _two_hop(g, f, name, ("SumAggregator", "MeanAggregator"), seeds, K1, K2, 11, (1, 2))
    for agg, out in outs.items():
        r = check_step((src, dst, None), lambda ids: X[ids], name, (K1, K2), agg, seeds, out, seed=11, call_counters=(1, 2),
                       rows_hop1=4096, rows_hop2=8192, segments=16384)
        assert r["ok"], (agg, r)
        assert r["rows_hop1"] >= 4096 and r["rows_hop2"] >= 8192 and r["segments_hop2"] == 16384, r
    f.close()
    g.close()


def test_c4_full_size_step_equals_oracle():
    V, E, D, K1, K2 = 111_000_000, 1_600_000_000, 128, 20, 15
    dev = torch.device("cuda", 0)
    src, dst, _ = synth.rmat_edges_torch(V, E, 6, dev, weighted=False)
    torch.cuda.synchronize()
    g = glx.Graph.from_edges(src, dst, None)
    torch.cuda.empty_cache()
    X = synth.features_torch(V, D, 9, dev)  # 57 GB
    f = glx.Features(X)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    seeds = torch.randint(0, V, (B0,), generator=gen, device=dev)
    out = _two_hop(g, f, "RandomSampler", ("MeanAggregator",), seeds, K1, K2, 11, (1, 2))["MeanAggregator"]
    assert int(out["e2"].max()) >= 2 ** 30  # edge ids of the far half of the edge list are among the answers
    r = check_step((src, dst, None), lambda ids: X[ids], "RandomSampler", (K1, K2), "MeanAggregator", seeds, out, seed=11,
                   call_counters=(1, 2), rows_hop1=4096, rows_hop2=8192, segments=16384)
    assert r["ok"], r
    assert r["rows_hop1"] >= 4096 and r["rows_hop2"] >= 8192 and r["segments_hop2"] == 16384, r
    f.close()
    g.close()
    del src, dst, X, out
    torch.cuda.empty_cache()


def test_c5_full_size_step_equals_oracle():
    D = 256
    dev = torch.device("cuda", 0)
    n_user, n_item, n_shop = 40_000_000, 9_000_000, 1_000_000
    spec = {"u-i": (n_user, n_item, 300_000_000, 10), "i-s": (n_item, n_shop, 100_000_000, 10),
            "u-s": (n_user, n_shop, 100_000_000, 5)}
    orc, pick = Oracle(), np.random.default_rng(5)
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    seeds = torch.randint(0, n_user, (B0,), generator=gen, device=dev)
    answers, requests = {}, {"u-i": seeds, "u-s": seeds}
    for i, (t, (ns, nd, ne, k)) in enumerate(spec.items()):  # one edge type at a time: build, sample, check, free
        src, dst, w = synth.rmat_edges_torch(1 << 26, ne, 20 + i, dev, weighted=True)
        src %= ns
        dst %= nd
        g = glx.Graph.from_edges(src, dst, w)
        req = requests[t]
        nbr, eid = g.sample("TopkSampler", req, k)
        torch.cuda.synchronize()
        ok, ne_sub = check_sample(orc, (src, dst, w), "TopkSampler", k, req, nbr, eid, 0, 0, _pick(req.shape[0], 4096, pick))
        assert ok, t + " sample"
        assert ne_sub > 0
        answers[t] = nbr
        if t == "u-i":
            requests["i-s"] = nbr.view(-1)
        g.close()
        del src, dst, w, g
        torch.cuda.empty_cache()
    for name, n_rows, fseed, want in (("i-s", n_shop, 32, 16384), ("u-i", n_item, 31, 2048), ("u-s", n_shop, 32, 2048)):
        X = synth.features_torch(n_rows, D, fseed, dev)
        f = glx.Features(X)
        ids2d = answers[name]
        emb, cnt = f.aggregate("SumAggregator", ids2d.view(-1), None, ids2d.shape[0])
        torch.cuda.synchronize()
        assert check_aggregate(orc, lambda ids: X[ids], "SumAggregator", ids2d, emb, cnt, _pick(ids2d.shape[0], want, pick)), \
            name + " aggregate"
        f.close()
        del X, f, emb, cnt
        torch.cuda.empty_cache()

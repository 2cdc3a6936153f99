"""GPU tests at BASELINE.json's full size (configs[2]: RMAT 10M nodes / 100M edges,
fanout [25,10], dim 256, 65,536 seeds) through size-independent properties:
every sampled (src, nbr, eid) is a real edge, without-replacement rows are
duplicate-free and circularly padded, Topk equals the row prefix, the stream is
reproducible, Max equals an order-free torch reduction exactly, Sum matches torch
within 1e-5 relative (the north-star tolerance; only the summation ORDER differs
from torch's), Mean == Sum / count bit-for-bit, counts == segment sizes."""
import numpy as np
import pytest
import torch

import glx
import synth

pytestmark = pytest.mark.gpu
V, E, D, B0, K1, K2 = 10_000_000, 100_000_000, 256, 65536, 25, 10


@pytest.fixture(scope="module")
def c3():
    dev = torch.device("cuda", 0)
    row_ptr, col, eid, w = synth.rmat_graph_torch(V, E, 4, dev, weighted=True)
    g = glx.Graph(row_ptr, col, eid, w)
    deg = row_ptr[1:] - row_ptr[:-1]
    slot_of_eid = torch.empty(E, dtype=torch.int64, device=dev)
    slot_of_eid[eid] = torch.arange(E, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    seeds = torch.randint(0, V, (B0,), generator=gen, device=dev)
    return dict(dev=dev, row_ptr=row_ptr, col=col, eid=eid, deg=deg, slot_of_eid=slot_of_eid, g=g, seeds=seeds)


def _check_membership(c, src, nbr, eid, default=0):
    k = nbr.shape[1]
    srcx = src.view(-1, 1).expand(-1, k)
    real = eid >= 0
    slot = c["slot_of_eid"][eid.clamp(min=0)]
    start = c["row_ptr"][srcx]
    end = c["row_ptr"][srcx + 1]
    assert bool(((slot >= start) & (slot < end))[real].all()), "sampled edge id is not an out-edge of its source"
    assert bool((c["col"][slot] == nbr)[real].all()), "neighbour id does not match the edge id"
    empty = (c["deg"][src] == 0).view(-1, 1).expand(-1, k)
    assert bool((real == ~empty).all()), "default fill must happen exactly for rows without neighbours"
    assert bool((nbr[~real] == default).all())
    return slot - start  # row-local positions


@pytest.mark.parametrize("name", list(glx.SAMPLER_IDS))
def test_two_hop_sampling_properties(c3, name):
    c = c3
    n1, e1 = c["g"].sample(name, c["seeds"], K1, seed=11, call_counter=1)
    n2, e2 = c["g"].sample(name, n1.view(-1), K2, seed=11, call_counter=2)
    torch.cuda.synchronize()
    assert n1.shape == (B0, K1) and n2.shape == (B0 * K1, K2)
    for src, nbr, eid, k in ((c["seeds"], n1, e1, K1), (n1.view(-1), n2, e2, K2)):
        pos = _check_membership(c, src, nbr, eid)
        deg = c["deg"][src].view(-1, 1)
        if name == "TopkSampler":
            j = torch.arange(k, device=c["dev"]).view(1, -1)
            ok = (pos == j % deg.clamp(min=1)) | (deg == 0)
            assert bool(ok.all())
        if name == "RandomWithoutReplacementSampler":
            m = deg.clamp(max=k)
            j = torch.arange(k, device=c["dev"]).view(1, -1)
            # circular padding: slot j repeats slot j % min(k, deg)
            rep = torch.gather(pos, 1, (j % m.clamp(min=1)).expand_as(pos))
            assert bool(((pos == rep) | (deg == 0)).all())
            # the first min(k, deg) picks are distinct: sort and compare neighbours
            big = torch.where(j < m, pos, torch.full_like(pos, -1) - j)  # unique fillers
            s = torch.sort(big, dim=1).values
            assert bool((s[:, 1:] != s[:, :-1]).all())
        if name == "RandomSampler":
            sel = (deg >= 100).expand_as(pos)
            frac = (pos[sel].double() / deg.expand_as(pos)[sel].double()).mean().item()
            assert abs(frac - 0.5) < 0.01, frac
    # reproducible stream; a new call counter gives new draws
    m1, f1 = c["g"].sample(name, c["seeds"], K1, seed=11, call_counter=1)
    assert torch.equal(m1, n1) and torch.equal(f1, e1)
    if name != "TopkSampler":
        m2, _ = c["g"].sample(name, c["seeds"], K1, seed=11, call_counter=3)
        assert not torch.equal(m2, n1)


def test_aggregation_properties_full_size(c3):
    c = c3
    dev = c["dev"]
    X = synth.features_torch(V, D, 5, dev)
    f = glx.Features(X)
    n1, _ = c["g"].sample("EdgeWeightSampler", c["seeds"], K1, seed=11, call_counter=1)
    n2, _ = c["g"].sample("EdgeWeightSampler", n1.view(-1), K2, seed=11, call_counter=2)
    ids = n2.view(-1)
    Sg = B0 * K1
    seg = (torch.arange(ids.shape[0], device=dev) // K2).to(torch.int32)
    emb_max, cnt = f.aggregate("MaxAggregator", ids, seg, Sg)
    emb_sum, cnt_s = f.aggregate("SumAggregator", ids, seg, Sg)
    emb_mean, _ = f.aggregate("MeanAggregator", ids, seg, Sg)
    torch.cuda.synchronize()
    assert bool((cnt == K2).all()) and torch.equal(cnt, cnt_s)
    # reference reductions in chunks (16.8 GB of gathered rows otherwise)
    chunk = 1 << 16
    for lo in range(0, Sg, chunk * 8):  # sample 1/8 of the segments, evenly spread
        hi = min(lo + chunk, Sg)
        rows = X[ids[lo * K2:hi * K2]].view(hi - lo, K2, D)
        ref_max = torch.maximum(rows.amax(1), torch.full((), -37.0, device=dev))  # max_aggregator.cc:28
        assert torch.equal(emb_max[lo:hi], ref_max)
        ref_sum = rows.sum(1)
        err = (emb_sum[lo:hi] - ref_sum).abs()
        tol = 1e-5 * rows.abs().sum(1) + 1e-30
        assert bool((err <= tol).all()), float((err / tol).max())
    assert torch.equal(emb_mean, emb_sum / cnt.view(-1, 1).to(torch.float32))
    # hop-1 level: segments of K1
    e1, c1 = f.aggregate("MaxAggregator", n1.view(-1), (torch.arange(B0 * K1, device=dev) // K1).to(torch.int32), B0)
    ref1 = torch.maximum(X[n1.view(-1)].view(B0, K1, D).amax(1), torch.full((), -37.0, device=dev))
    assert torch.equal(e1, ref1) and bool((c1 == K1).all())


def test_device_build_matches_independent_csr_at_full_size(c3):
    """glx_graph_build on the 100M-edge list == the torch-sorted CSR handed to
    glx_graph_create: same degrees and the same samples for every sampler."""
    c = c3
    src, dst, w = synth.rmat_edges_torch(V, E, 4, c["dev"], weighted=True)
    built = glx.Graph.from_edges(src, dst, w)
    del src, dst, w
    assert built.num_edges == E and built.num_rows == int((c["deg"] > 0).sum())
    q = c["seeds"][:8192].contiguous()
    assert torch.equal(built.degrees(q), c["g"].degrees(q))
    for name in glx.SAMPLER_IDS:
        a = built.sample(name, q, K1, seed=5, call_counter=9)
        b = c["g"].sample(name, q, K1, seed=5, call_counter=9)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), name


def test_negative_sampling_properties_full_size(c3):
    """Negative samplers on the C3 graph: the candidate list is exactly the distinct destinations
    with their in-degrees (torch recount), first-appearance ordered by edge id; strict in-degree
    negatives of 1.6M rows are never out-neighbours of their source except through the documented
    4th-block fallback (rare: bounded here); all outputs are candidates; the stream is reproducible."""
    c = c3
    g = c["g"]
    g.enable_negative()
    uni = glx.Negative.from_graph(g)
    deg = glx.Negative.from_graph(g, by_in_degree=True)
    ids, prob, alias = deg.export()
    dev = c["dev"]
    tid = torch.from_numpy(ids).to(dev)
    uniq, counts = torch.unique(c["col"], return_counts=True)
    assert tid.shape[0] == uniq.shape[0] and bool((torch.sort(tid).values == uniq).all())
    # first appearance = smallest edge id of each destination, ascending
    first = torch.full((V,), E, dtype=torch.int64, device=dev).scatter_reduce(0, c["col"], c["eid"], "amin")
    assert bool((first[tid][1:] > first[tid][:-1]).all())
    # alias table mass: sum over slots of (prob[i] at i + (1 - prob[j]) for j aliased to i) = in-degree share * U
    p = torch.from_numpy(prob).to(dev).double()
    a = torch.from_numpy(alias).to(dev).long()
    mass = p.clone()
    mass.scatter_add_(0, a, 1.0 - p)
    want = counts[torch.searchsorted(uniq, tid)].double() / E * tid.shape[0]
    # (float32 arithmetic of the serial build over 4.6 M entries, as in the reference: ~1e-3 relative drift)
    assert bool(((mass - want).abs() <= 1e-2 * want + 1e-3).all())
    src = c["seeds"].repeat_interleave(4)[: 200_000]
    out = deg.sample(src, 10, exclude=glx.NEG_EXCLUDE_NEIGHBORS, graph=g, seed=5, call_counter=7)
    again = deg.sample(src, 10, exclude=glx.NEG_EXCLUDE_NEIGHBORS, graph=g, seed=5, call_counter=7)
    assert torch.equal(out, again)
    cand = torch.zeros(V, dtype=torch.bool, device=dev)
    cand[tid] = True
    assert bool(cand[out].all()) and bool(cand[uni.sample(src, 10, seed=5, call_counter=8)].all())
    # is (src, negative) an existing edge?  encode pairs and test membership against the edge list
    row = torch.repeat_interleave(torch.arange(V, device=dev), c["deg"])
    edge_keys = torch.sort(row * V + c["col"]).values
    keys = (src.view(-1, 1) * V + out).view(-1)
    pos = torch.searchsorted(edge_keys, keys).clamp(max=E - 1)
    leaks = int((edge_keys[pos] == keys).sum())
    soft = deg.sample(src, 10, seed=5, call_counter=7)
    skeys = (src.view(-1, 1) * V + soft).view(-1)
    spos = torch.searchsorted(edge_keys, skeys).clamp(max=E - 1)
    soft_leaks = int((edge_keys[spos] == skeys).sum())
    assert leaks * 20 < max(soft_leaks, 1), (leaks, soft_leaks)  # strict removes (almost) all of them


def test_filtered_sampling_properties_full_size(c3):
    """op::Filter at full size (id == value with the value = the seed's first neighbour, i.e. GSL's .filter()):
    the filtered id never comes back (RandomSampler: only through exhausted retries, none with a budget of 60),
    every answer is still a real out-edge, rows whose neighbours all hit are default-filled, without-replacement
    rows stay duplicate-free, and Topk is the unfiltered row with the hits taken out in ActOn's order (checked on
    the rows that have no hit among their first K1 + 1 slots, where that order is the identity)."""
    c = c3
    seeds = c["seeds"]
    g = c["g"]
    first, _ = g.sample("TopkSampler", seeds, 1)
    values = first.view(-1).contiguous()
    deg = c["deg"][seeds]
    start = c["row_ptr"][seeds]
    for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "TopkSampler"):
        nbr, eid = g.sample_filtered(name, seeds, K1, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, values, seed=5, call_counter=2,
                                     retry_times=60)
        torch.cuda.synchronize()
        real = eid >= 0
        slot = c["slot_of_eid"][eid.clamp(min=0)]
        srcx = seeds.view(-1, 1).expand(-1, K1)
        assert bool(((slot >= c["row_ptr"][srcx]) & (slot < c["row_ptr"][srcx + 1]))[real].all()), name
        assert bool((c["col"][slot] == nbr)[real].all()), name
        assert bool((nbr[real] != values.view(-1, 1).expand(-1, K1)[real]).all()), name  # the filtered id is gone
        assert bool((nbr[~real] == 0).all())
        # a row is default-filled exactly when it has no neighbour other than the filtered id
        survivors = torch.zeros_like(deg)
        has = deg > 0
        # count neighbours != value over the rows (segment sum over each row's slots)
        rows_idx = torch.repeat_interleave(torch.arange(seeds.shape[0], device=c["dev"])[has], deg[has])
        offs = torch.arange(int(deg[has].sum()), device=c["dev"]) - torch.repeat_interleave(
            torch.cumsum(deg[has], 0) - deg[has], deg[has])
        col = c["col"][torch.repeat_interleave(start[has], deg[has]) + offs]
        survivors.index_add_(0, rows_idx, (col != values[rows_idx]).to(survivors.dtype))
        assert bool((real.all(dim=1) == (survivors > 0)).all()), name
        assert bool((real.any(dim=1) == real.all(dim=1)).all()), name
        if name == "RandomWithoutReplacementSampler":
            m = torch.minimum(survivors, torch.full_like(survivors, K1))
            srt, _ = torch.sort(torch.where(real, eid, -1 - torch.arange(K1, device=c["dev"]).view(1, -1).expand_as(eid)), dim=1)
            distinct = (srt[:, 1:] != srt[:, :-1]).sum(dim=1) + 1
            assert bool((distinct[survivors > 0] == m[survivors > 0]).all())
        if name == "TopkSampler":
            plain, _ = g.sample("TopkSampler", seeds, K1 + 1)
            hit_free_tail = (plain[:, 1:] != values.view(-1, 1)).all(dim=1) & (deg > K1 + 1) & (survivors == deg - 1)
            # one hit, at slot 0: the hole is refilled with the row's LAST neighbour (filter.cc:83-94)
            last = c["col"][start + deg - 1]
            sel = hit_free_tail
            assert bool((nbr[sel][:, 0] == last[sel]).all()) and bool((nbr[sel][:, 1:] == plain[sel][:, 1:K1]).all())
            assert int(sel.sum()) > 100


def test_random_walk_properties_full_size(c3):
    """RandomWalk at full size: a DeepWalk is the chain of neighbor_count-1 RandomSampler draws; every node2vec step
    lands on one of the first DefaultFullNbrNum neighbours of where it stood (or on the default id at a dead end).
    (The RMAT graph is directed and has hardly any reciprocal edges, so the return bias of p is checked on the small
    graphs of test_gpu_walk.py instead.)"""
    c = c3
    g, seeds, dev = c["g"], c["seeds"], c["dev"]
    L, F = 5, 100
    deep = g.random_walk(seeds, L, seed=9, call_counter=4, default_neighbor_id=0)
    cur = seeds
    for t in range(L):
        nbr, _ = g.sample("RandomSampler", cur, 1, seed=9, call_counter=4 + t)
        assert bool((deep[:, t] == nbr[:, 0]).all())
        cur = nbr[:, 0].contiguous()

    def walk(p, q):
        w = g.random_walk(seeds, L, p=p, q=q, full_nbr_num=F, seed=9, call_counter=4, default_neighbor_id=0)
        torch.cuda.synchronize()
        prev = seeds
        for t in range(L):
            nxt = w[:, t]
            d = torch.clamp(c["deg"][prev], max=F)
            j = torch.arange(F, device=dev).view(1, -1)
            slot = (c["row_ptr"][prev].view(-1, 1) + j).clamp(max=E - 1)
            among = ((c["col"][slot] == nxt.view(-1, 1)) & (j < d.view(-1, 1))).any(dim=1)
            assert bool((among | ((d == 0) & (nxt == 0))).all()), (p, q, t)
            prev = nxt
        return w
    for p, q in ((0.05, 1.0), (20.0, 0.25)):
        w = walk(p, q)
        assert w.shape == (B0, L)
    assert not bool((walk(0.05, 1.0) == walk(20.0, 0.25)).all())

"""GPU tests of the multi-shard path on ONE device: P shard stores live on the
same GPU and the all-to-all is emulated by slicing, so every device kernel of
the sharded pipeline (glx_partition, glx_sample_ex with rng_rows, glx_lookup,
glx_features_view + glx_aggregate, glx_stitch_*) runs for real.  Claim under
test: for every shard count the stitched result is bit-identical to the
single-shard result (which the reference's own distributed Max/Min/Prod is not:
SURVEY.md 8(a) quirk 8).  Also drives dist.ShardedStore over RCCL with
world_size 1."""
import os

import numpy as np
import pytest
import torch

import glx
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    import dist as gdist
    rp, col, eid, w = synth.small_graph(5000, 80000, seed=21, weighted=True, hub_degree=3000)
    X = np.random.default_rng(4).standard_normal((5000, 64)).astype(np.float32)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    whole = glx.Graph(t(rp), t(col), t(eid), t(w))
    feats = glx.Features(t(X))
    shards = {}
    for P in (2, 4, 8):
        gs, fs = [], []
        for r in range(P):
            srp, scol, seid, sw, sids = gdist.shard_graph(t(rp), t(col), t(eid), t(w), r, P)
            gs.append(glx.Graph(srp, scol, seid, sw, ids=sids))
            fs.append(glx.Features(t(X[r::P].copy()), ids=sids))
        shards[P] = (gs, fs)
    return whole, feats, shards, dev


@pytest.mark.parametrize("P", [2, 4, 8])
def test_sharded_sampling_equals_single_shard(world, P):
    whole, _, shards, dev = world
    gs, _ = shards[P]
    rng = np.random.default_rng(P)
    src = torch.from_numpy(np.concatenate([rng.integers(0, 5000, 3000), [0, 0, -1, 5000]]).astype(np.int64)).to(dev)
    cc = 0
    for name in glx.SAMPLER_IDS:
        for k, pad in ((10, 1), (25, 1), (7, 0), (70, 1)):
            cc += 1
            ref_n, ref_e = whole.sample(name, src, k, seed=9, call_counter=cc, padding_mode=pad,
                                        default_neighbor_id=-5)
            bucketed, order, counts = glx.partition(src, P)
            offs = np.concatenate([[0], np.cumsum(counts.cpu().numpy())])
            parts_n, parts_e = [], []
            for p in range(P):
                a, b = int(offs[p]), int(offs[p + 1])
                n, e = gs[p].sample(name, bucketed[a:b].contiguous(), k, seed=9, call_counter=cc,
                                    padding_mode=pad, default_neighbor_id=-5,
                                    rng_rows=order[a:b].contiguous())
                parts_n.append(n)
                parts_e.append(e)
            n = glx.stitch(torch.cat(parts_n), order)
            e = glx.stitch(torch.cat(parts_e), order)
            assert torch.equal(n, ref_n) and torch.equal(e, ref_e), (name, k, pad)


@pytest.mark.parametrize("P", [2, 8])
def test_sharded_aggregation_equals_single_shard(world, P):
    whole, feats, shards, dev = world
    _, fs = shards[P]
    rng = np.random.default_rng(50 + P)
    n = 20000
    ids = torch.from_numpy(rng.integers(-3, 5003, n).astype(np.int64)).to(dev)
    seg = torch.from_numpy((np.arange(n) // 10).astype(np.int32)).to(dev)
    for name in glx.AGGREGATOR_IDS:
        ref_e, ref_c = feats.aggregate(name, ids, seg, n // 10, default_attr=0.5)
        bucketed, order, counts = glx.partition(ids, P)
        offs = np.concatenate([[0], np.cumsum(counts.cpu().numpy())])
        rows = torch.cat([fs[p].lookup(bucketed[int(offs[p]):int(offs[p + 1])].contiguous(), 0.5)
                          for p in range(P)])
        pos = glx.stitch(torch.arange(n, dtype=torch.int64, device=dev).view(n, 1), order).view(n)
        view = glx.Features(rows, view=True)
        e, c = view.aggregate(name, pos, seg, n // 10, default_attr=0.5)
        assert torch.equal(c, ref_c), name
        assert torch.equal(e.view(torch.int32), ref_e.view(torch.int32)), name


def test_device_aggregate_stitch_equals_oracle_on_reference_partials():
    """glx_aggregate_stitch on the partial responses of a 3-server run of the reference
    (tests/golden/agg_stitch.npz): bit-identical to the oracle's fold."""
    from oracle_bindings import Oracle
    orc = Oracle()
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "agg_stitch.npz")))
    dev = torch.device("cuda", 0)
    for c in range(int(g["num_cases"])):
        dflt = float(g["c%d_default" % c])
        for name in glx.AGGREGATOR_IDS:
            parts, cnts = g["c%d_%s_parts" % (c, name)], g["c%d_%s_cnts" % (c, name)]
            oe, oc = orc.aggregate_stitch(name, parts, cnts, dflt)
            e, n = glx.aggregate_stitch(name, torch.from_numpy(parts).to(dev), torch.from_numpy(cnts).to(dev), dflt)
            assert np.array_equal(n.cpu().numpy(), oc), (c, name)
            assert np.array_equal(e.cpu().numpy().view(np.uint32), oe.view(np.uint32)), (c, name)


@pytest.mark.parametrize("P", [2, 8])
def test_partial_reduce_equals_single_shard(world, P):
    """Design R: every shard store reduces its subset, glx_aggregate_stitch folds the
    partials.  Counts and Max/Min are exact; device fold == oracle fold bit for bit."""
    from oracle_bindings import Oracle
    orc = Oracle()
    whole, feats, shards, dev = world
    _, fs = shards[P]
    rng = np.random.default_rng(70 + P)
    n, f = 24000, 12
    ids = torch.from_numpy(rng.integers(-3, 5003, n).astype(np.int64)).to(dev)
    # a few segments whose ids all live on one shard (multiples of P): other shards see nothing
    ids.view(-1, f)[::7] = (ids.view(-1, f)[::7] // P) * P
    seg = torch.from_numpy((np.arange(n) // f).astype(np.int32)).to(dev)
    sg = n // f
    for name in glx.AGGREGATOR_IDS:
        ref_e, ref_c = feats.aggregate(name, ids, seg, sg, default_attr=0.5)
        bucketed, order, counts = glx.partition(ids, P)
        seg_b = seg[order].contiguous()
        offs = np.concatenate([[0], np.cumsum(counts.cpu().numpy())])
        pe, pc = [], []
        for p in range(P):
            a, b = int(offs[p]), int(offs[p + 1])
            e, c = fs[p].aggregate(name, bucketed[a:b].contiguous(), seg_b[a:b].contiguous(), sg, default_attr=0.5)
            pe.append(e)
            pc.append(c)
        parts, cnts = torch.stack(pe), torch.stack(pc)
        e, c = glx.aggregate_stitch(name, parts, cnts, 0.5)
        assert torch.equal(c, ref_c), name
        oe, oc = orc.aggregate_stitch(name, parts.cpu().numpy(), cnts.cpu().numpy(), 0.5)
        assert np.array_equal(e.cpu().numpy().view(np.uint32), oe.view(np.uint32)), name
        if name in ("MaxAggregator", "MinAggregator"):
            assert torch.equal(e.view(torch.int32), ref_e.view(torch.int32)), name
        elif name != "ProdAggregator":
            assert torch.allclose(e, ref_e, rtol=1e-5, atol=1e-4), name
        else:
            assert torch.allclose(e, ref_e, rtol=1e-4, atol=1e-30), name


def test_sharded_store_over_rccl_world1(world):
    import torch.distributed as dist
    import dist as gdist
    whole, feats, _, dev = world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        store = gdist.ShardedStore(gdist.DeviceOps(), whole, feats)
        src = torch.arange(0, 4000, dtype=torch.int64, device=dev)
        for name in glx.SAMPLER_IDS:
            n, e = store.sample(name, src, 10, seed=3, call_counter=1)
            rn, re = whole.sample(name, src, 10, seed=3, call_counter=1)
            assert torch.equal(n, rn) and torch.equal(e, re)
        # a filtered request through the exchange: the values travel with their rows
        vals = whole.sample("TopkSampler", src, 1)[0].view(-1).contiguous()  # every row's first neighbour
        for name in glx.SAMPLER_IDS:
            fn, fe = store.sample_filtered(name, src, 10, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, vals, seed=3, call_counter=2)
            rn2, re2 = whole.sample_filtered(name, src, 10, glx.FILTER_EQUAL, glx.FILTER_FIELD_ID, vals, seed=3, call_counter=2)
            assert torch.equal(fn, rn2) and torch.equal(fe, re2)
        seg = (torch.arange(40000, device=dev) // 10).to(torch.int32)
        emb, cnt = store.aggregate("MeanAggregator", n.view(-1), seg, 4000)
        remb, rcnt = feats.aggregate("MeanAggregator", n.view(-1), seg, 4000)
        assert torch.equal(cnt, rcnt) and torch.equal(emb.view(torch.int32), remb.view(torch.int32))
        emb, cnt = store.aggregate("MeanAggregator", n.view(-1), seg, 4000, dedup=True)
        assert torch.equal(cnt, rcnt) and torch.equal(emb.view(torch.int32), remb.view(torch.int32))
        emb, cnt = store.aggregate("MaxAggregator", n.view(-1), seg, 4000, mode="partial")
        remb, rcnt = feats.aggregate("MaxAggregator", n.view(-1), seg, 4000)
        assert torch.equal(cnt, rcnt) and torch.equal(emb.view(torch.int32), remb.view(torch.int32))
        # Regression for the RCCL large-message limit (profiles/r01/rccl_large_message_probe.txt): a
        # 1.25 GiB peer message must arrive whole -- dist._a2a cuts it into <= 512 MiB rounds.
        rows = (5 << 28) // (64 * 4)
        big = torch.arange(rows * 64, device=dev, dtype=torch.int32).view(rows, 64)
        got = gdist._a2a(big, [rows], [rows], None, rows)
        assert torch.equal(got, big)
        del big, got
    finally:
        dist.destroy_process_group()

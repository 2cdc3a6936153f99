"""world_size-2 (and 3) gloo tests of the sharded execution layer
(graph-learn_amd/dist.py): Partition -> all-to-all -> Process on the owner ->
all-to-all -> Stitch, the replacement of DistributeRunner (op_runner.h:60-152).

The exchange logic is the product's; the local compute is injected: on the GPU
it is DeviceOps (HIP through the C-ABI), here -- CPU only, test infrastructure
-- it is an oracle-backed stand-in with the same interface.  The assertion is
the strongest available: the stitched result of every rank's request equals,
bit for bit, the unpartitioned single-shard result (oracle on the whole graph).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


class OracleOps:
    """CPU stand-in for dist.DeviceOps (tests only)."""

    def __init__(self):
        from oracle_bindings import Oracle
        self.o = Oracle()

    def partition(self, ids, num_shards):
        a = ids.numpy()
        order, counts = self.o.partition(a, num_shards)
        return torch.from_numpy(a[order]), torch.from_numpy(order), torch.from_numpy(counts)

    def stitch(self, rows, order):
        out = torch.empty_like(rows)
        out[order] = rows
        return out

    def sample(self, graph, sampler, ids, rng_rows, k, seed, cc, pad, dflt):
        n, e = self.o.sample(graph, sampler, ids.numpy(), k, seed=seed, call_counter=cc, padding_mode=pad,
                             default_neighbor_id=dflt, rng_rows=rng_rows.numpy())
        return torch.from_numpy(n), torch.from_numpy(e)

    def sample_filtered(self, graph, sampler, ids, rng_rows, k, seed, cc, pad, dflt, ftype, ffield, values, retry,
                        default_ts):
        flt = dict(type=ftype, field=ffield, values=values.numpy(), retry_times=retry, default_timestamp=default_ts)
        n, e = self.o.sample_filtered(graph, sampler, ids.numpy(), k, flt, seed=seed, call_counter=cc, padding_mode=pad,
                                      default_neighbor_id=dflt, rng_rows=rng_rows.numpy())
        return torch.from_numpy(n), torch.from_numpy(e)

    def lookup(self, feats, ids, default_attr):
        X, raw = feats
        row_of = {int(v): i for i, v in enumerate(raw)}
        out = np.full((ids.shape[0], X.shape[1]), default_attr, np.float32)
        for i, v in enumerate(ids.numpy()):
            r = row_of.get(int(v))
            if r is not None:
                out[i] = X[r]
        return torch.from_numpy(out)

    def rows_of(self, feats, ids):
        X, raw = feats
        row_of = {int(v): i for i, v in enumerate(raw)}
        return torch.tensor([row_of.get(int(v), -1) for v in ids.numpy()], dtype=torch.int64)

    def aggregate_rows(self, rows, pos, seg, num_segments, op, default_attr):
        e, c = self.o.aggregate(rows.numpy(), op, pos.numpy(), seg.numpy(), num_segments, default_attr)
        return torch.from_numpy(e), torch.from_numpy(c)

    def aggregate_local(self, feats, op, node_ids, seg, num_segments, default_attr):
        if isinstance(feats, tuple):  # a shard: (rows, their raw ids)
            X, raw = feats
            e, c = self.o.aggregate(X, op, node_ids.numpy(), seg.numpy(), num_segments, default_attr, ids=raw)
        else:
            e, c = self.o.aggregate(feats.numpy(), op, node_ids.numpy(), seg.numpy(), num_segments, default_attr)
        return torch.from_numpy(e), torch.from_numpy(c)

    def aggregate_stitch(self, op, parts, cnts, default_attr):
        e, c = self.o.aggregate_stitch(op, parts.numpy(), cnts.numpy(), default_attr)
        return torch.from_numpy(e), torch.from_numpy(c)


def _world_graph():
    import synth
    rp, col, eid, w = synth.small_graph(600, 9000, seed=12, weighted=True, hub_degree=300)
    X = np.random.default_rng(3).standard_normal((600, 24)).astype(np.float32)
    return rp, col, eid, w, X


def _worker(rank, world, port, q, message_limit=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dist as gdist
        from oracle_bindings import Oracle
        if message_limit:  # force the multi-round exchange (peer messages cut into pieces)
            gdist.MAX_MESSAGE_BYTES = message_limit
        ops = OracleOps()
        orc = Oracle()
        rp, col, eid, w, X = _world_graph()
        whole = dict(row_ptr=rp, col=col, eid=eid, weight=w, alias=orc.alias_build(rp, w))
        t = lambda a: torch.from_numpy(a)  # noqa: E731
        srp, scol, seid, sw, sids = gdist.shard_graph(t(rp), t(col), t(eid), t(w), rank, world)
        shard = dict(row_ptr=srp.numpy(), col=scol.numpy(), eid=seid.numpy(), weight=sw.numpy(),
                     ids=sids.numpy())
        shard["alias"] = orc.alias_build(shard["row_ptr"], shard["weight"])
        feats = (X[rank::world].copy(), sids.numpy())
        store = gdist.ShardedStore(ops, shard, feats)
        rng = np.random.default_rng(100 + rank)
        # each rank drives its OWN request; sizes differ per rank on purpose
        src = np.concatenate([rng.integers(0, 600, 150 + 37 * rank), [0, 0, -4, 600, 10 ** 9]]).astype(np.int64)
        ok = True
        cc = 0
        for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "TopkSampler"):
            for k, pad in ((4, 1), (9, 1), (5, 0)):
                cc += 1
                n1, e1 = store.sample(name, t(src), k, seed=77, call_counter=cc, padding_mode=pad,
                                      default_neighbor_id=-2)
                on, oe = orc.sample(whole, name, src, k, seed=77, call_counter=cc, padding_mode=pad,
                                    default_neighbor_id=-2)
                ok &= np.array_equal(n1.numpy(), on) and np.array_equal(e1.numpy(), oe)
                # hop 2 on the hop-1 output (the NeighborSampler.get loop, neighbor_sampler.py:93-127)
                cc += 1
                n2, _ = store.sample(name, n1.reshape(-1), 3, seed=77, call_counter=cc, padding_mode=pad,
                                     default_neighbor_id=-2)
                on2, _ = orc.sample(whole, name, on.reshape(-1), 3, seed=77, call_counter=cc, padding_mode=pad,
                                    default_neighbor_id=-2)
                ok &= np.array_equal(n2.numpy(), on2)
        # requests with an op::Filter: the values travel with their rows; id filters give the single-store answer
        ts_whole = (np.arange(col.shape[0], dtype=np.int64) * 7919) % 1009  # "timestamps": any per-edge field
        whole_f = dict(whole, ts_slot=ts_whole[eid])
        shard_f = dict(shard, ts_slot=ts_whole[shard["eid"]])
        shard_f["indeg_weight"] = whole_f["indeg_weight"] = None
        store_f = gdist.ShardedStore(ops, shard_f, feats)
        for (ftype, ffield) in ((1, 1), (2, 1), (1, 2)):
            vals = rng.integers(0, 600, src.shape[0]).astype(np.int64) if ffield == 1 else rng.integers(0, 1009, src.shape[0]).astype(np.int64)
            for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "TopkSampler"):
                cc += 1
                n1, e1 = store_f.sample_filtered(name, t(src), 5, ftype, ffield, t(vals), seed=77, call_counter=cc,
                                                 default_neighbor_id=-2, retry_times=2)
                on, oe = orc.sample_filtered(whole_f, name, src, 5, dict(type=ftype, field=ffield, values=vals, retry_times=2),
                                             seed=77, call_counter=cc, default_neighbor_id=-2)
                ok &= np.array_equal(n1.numpy(), on) and np.array_equal(e1.numpy(), oe)
        ids = n2.reshape(-1).numpy().copy()
        seg = (np.arange(ids.shape[0]) // 3).astype(np.int32)
        Sg = ids.shape[0] // 3
        for name in ("SumAggregator", "MeanAggregator", "MaxAggregator", "MinAggregator", "ProdAggregator"):
            emb, cnt = store.aggregate(name, t(ids), t(seg), Sg, default_attr=1.25)
            oemb, ocnt = orc.aggregate(X, name, ids, seg, Sg, 1.25)
            ok &= np.array_equal(cnt.numpy(), ocnt)
            ok &= np.array_equal(emb.numpy().view(np.uint32), oemb.view(np.uint32))
            # the same with every halo id shipped as often as it occurs
            emb, cnt = store.aggregate(name, t(ids), t(seg), Sg, default_attr=1.25, dedup=False)
            ok &= np.array_equal(cnt.numpy(), ocnt)
            ok &= np.array_equal(emb.numpy().view(np.uint32), oemb.view(np.uint32))
            # design R: owners reduce, the requester folds the partials.  Max/Min exact,
            # counts exact, Sum/Mean/Prod re-associated across shards (1e-5 relative).
            emb, cnt = store.aggregate(name, t(ids), t(seg), Sg, default_attr=1.25, mode="partial")
            ok &= np.array_equal(cnt.numpy(), ocnt)
            if name in ("MaxAggregator", "MinAggregator"):
                ok &= np.array_equal(emb.numpy().view(np.uint32), oemb.view(np.uint32))
            else:
                ok &= bool(np.allclose(emb.numpy(), oemb, rtol=1e-5, atol=1e-5))
        # hot-row replica + cold-tail halo exchange (the protocol of glx_dist_aggregate): replica sizes
        # none / partial (top in-degree ids, plus ids nobody knows) / everything
        indeg = np.bincount(col, minlength=600)
        top = np.lexsort((np.arange(600), -indeg))[:60].astype(np.int64)
        for hot in (np.empty(0, np.int64), np.concatenate([top, [700, -7]]), np.arange(600, dtype=np.int64)):
            store_c = gdist.ShardedStore(ops, shard, feats, hot_ids=hot)
            for name in ("SumAggregator", "MeanAggregator", "MaxAggregator", "MinAggregator", "ProdAggregator"):
                emb, cnt = store_c.aggregate(name, t(ids), t(seg), Sg, default_attr=1.25)
                oemb, ocnt = orc.aggregate(X, name, ids, seg, Sg, 1.25)
                ok &= np.array_equal(cnt.numpy(), ocnt)
                ok &= np.array_equal(emb.numpy().view(np.uint32), oemb.view(np.uint32))
            st = store_c.stats()
            ok &= st["from_replica"] + st["from_own_shard"] + st["remote"] == ids.shape[0]
            if hot.shape[0] == 600:
                ok &= st["remote"] == 0 or bool(((ids < 0) | (ids >= 600)).any())
            if hot.shape[0] == 0:
                ok &= st["from_replica"] == 0
        # load-time halo exchange: all-gather the feature shards, then aggregate locally
        full = gdist.replicate_features(t(X[rank::world].copy()), X.shape[0])
        ok &= np.array_equal(full.numpy(), X)
        store2 = gdist.ShardedStore(ops, shard, None, feature_replica=full)
        for name in ("SumAggregator", "MaxAggregator"):
            emb, cnt = store2.aggregate(name, t(ids), t(seg), Sg, default_attr=1.25)
            oemb, ocnt = orc.aggregate(X, name, ids, seg, Sg, 1.25)
            ok &= np.array_equal(cnt.numpy(), ocnt) and np.array_equal(emb.numpy().view(np.uint32), oemb.view(np.uint32))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,message_limit", [(2, None), (3, None), (2, 200)])
def test_sharded_store_equals_single_shard(world, message_limit):
    """message_limit = 200 bytes: every exchange runs in several rounds (dist.MAX_MESSAGE_BYTES keeps
    peer messages under RCCL's 1 GiB all-to-all limit on the GPU; here it is shrunk to exercise that path)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, message_limit)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(world))
    assert got == {r: True for r in range(world)}


def test_shard_graph_partitions_every_row_once():
    sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
    import dist as gdist
    rp, col, eid, w, _ = _world_graph()
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    seen = np.zeros(col.shape[0], np.int32)
    for rank in range(4):
        srp, scol, seid, sw, sids = gdist.shard_graph(t(rp), t(col), t(eid), t(w), rank, 4)
        assert (sids.numpy() % 4 == rank).all()
        for i, v in enumerate(sids.numpy()):
            a, b = srp[i].item(), srp[i + 1].item()
            assert np.array_equal(scol[a:b].numpy(), col[rp[v]:rp[v + 1]])
            assert np.array_equal(seid[a:b].numpy(), eid[rp[v]:rp[v + 1]])
            seen[rp[v]:rp[v + 1]] += 1
    assert (seen == 1).all()


def test_shard_and_row_helpers_on_edge_shapes():
    """More shards than vertices leave empty shards; rows_of_graph cuts any id subset (the shape of a graph replica)."""
    sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
    import dist as gdist
    rp, col, eid, w, _ = _world_graph()
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    V = rp.shape[0] - 1
    srp, scol, seid, sw, sids = gdist.shard_graph(t(rp), t(col), t(eid), t(w), V + 2, V + 5)  # owns nothing
    assert sids.shape[0] == 0 and srp.tolist() == [0] and scol.shape[0] == 0 and sw.shape[0] == 0
    pick = torch.tensor([V - 1, 0, 3], dtype=torch.int64)
    qrp, qcol, qeid, qw, qids = gdist.rows_of_graph(t(rp), t(col), t(eid), t(w), pick)
    assert torch.equal(qids, pick)
    for i, v in enumerate(pick.tolist()):
        a, b = qrp[i].item(), qrp[i + 1].item()
        assert np.array_equal(qcol[a:b].numpy(), col[rp[v]:rp[v + 1]]) and np.array_equal(qeid[a:b].numpy(), eid[rp[v]:rp[v + 1]])
        assert np.array_equal(qw[a:b].numpy(), w[rp[v]:rp[v + 1]])

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


    # -- the partitioned operations beyond the dense samplers (dist.ShardedStore.sample_full / in_degrees / negative_* /
    # -- random_walk): the owner's and the requester's halves, oracle-backed ----------------------------------------
    def sample_full(self, graph, ids, max_limit):
        d, n, e = self.o.sample_full(graph, np.ascontiguousarray(ids.numpy()), max_limit)
        return torch.from_numpy(d), torch.from_numpy(n), torch.from_numpy(e)

    def sample_full_filtered(self, graph, ids, max_limit, ftype, ffield, values, padding_mode, default_neighbor_id,
                             default_timestamp):
        if ids.shape[0] == 0:
            return torch.zeros(0, dtype=torch.int32), torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)
        flt = dict(type=ftype, field=ffield, values=np.ascontiguousarray(values.numpy()), default_timestamp=default_timestamp)
        d, n, e = self.o.sample_full_filtered(graph, np.ascontiguousarray(ids.numpy()), max_limit, flt,
                                              padding_mode=padding_mode, default_neighbor_id=default_neighbor_id)
        return torch.from_numpy(d), torch.from_numpy(n), torch.from_numpy(e)

    def dst_counts(self, graph):
        uniq, cnt = np.unique(graph["col"], return_counts=True)
        return torch.from_numpy(uniq.astype(np.int64)), torch.from_numpy(cnt.astype(np.int64))

    def negative_table(self, ids, weights):
        ids = np.ascontiguousarray(ids.numpy())
        if weights is None:
            return dict(ids=ids, alias=None)
        w = np.ascontiguousarray(weights.numpy(), np.float32)
        return dict(ids=ids, alias=self.o.alias_build(np.array([0, w.shape[0]], np.int64), w))

    def negative_sample(self, table, exclude, graph, src, rows, count, default_neighbor_id, seed, call_counter):
        src = np.ascontiguousarray(src.numpy())
        if rows is None:
            return torch.from_numpy(self.o.negative_sample(table["ids"], table["alias"], exclude, graph, src, count,
                                                           default_neighbor_id, seed, call_counter))
        # row r of the original request draws from the stream (seed, call_counter, r): with the neighbour exclusion the
        # rows are independent, so the part a server is given is answered inside a request long enough to hold its rows
        rows = rows.numpy()
        if rows.shape[0] == 0:
            return torch.zeros((0, count), dtype=torch.int64)
        frame = np.zeros(int(rows.max()) + 1, np.int64)
        frame[rows] = src
        out = self.o.negative_sample(table["ids"], table["alias"], exclude, graph, frame, count, default_neighbor_id, seed,
                                     call_counter)
        return torch.from_numpy(np.ascontiguousarray(out[rows]))

    def full_lists_with_weights(self, graph, ids, limit, default_weight):
        d, n, e = self.o.sample_full(graph, np.ascontiguousarray(ids.numpy()), limit)
        # FullSampler hands out a row's first min(deg, limit) slots in storage order: slot x of row r is edge row_ptr[r] + x
        row_of = {int(v): i for i, v in enumerate(graph["ids"])} if graph.get("ids") is not None else None
        w = np.zeros(n.shape[0], np.float32)
        at = 0
        for v, deg in zip(ids.numpy(), d):
            r = row_of.get(int(v), -1) if row_of is not None else int(v)
            for x in range(int(deg)):
                w[at + x] = graph["weight"][graph["row_ptr"][r] + x] if graph.get("weight") is not None else default_weight
            at += int(deg)
        return torch.from_numpy(d), torch.from_numpy(n), torch.from_numpy(w)

    def node2vec_step(self, parent, deg_c, nbr_c, w_c, deg_p, nbr_p, p, q, seed, call_counter, default_neighbor_id):
        """WeightedRandomWalkKernel (random_walk.cc:192-272) on the lists a step's FullSampler request brought: the biased
        weights restated here, table and draw through the oracle's EdgeWeightSampler on a graph whose rows are the
        walkers (row i, draw 0 of the stream (seed, call_counter, i): alias_method.cc:109-124 on n = min(deg, F) entries)."""
        parent, deg_c, nbr_c, w_c = parent.numpy(), deg_c.numpy(), nbr_c.numpy(), w_c.numpy()
        batch = parent.shape[0]
        off_c = np.zeros(batch + 1, np.int64)
        off_c[1:] = np.cumsum(deg_c)
        biased = np.zeros(nbr_c.shape[0], np.float32)
        inv_p = np.float64(np.float32(p)) + 1e-6
        inv_q = np.float64(np.float32(q)) + 1e-6
        cursor = 0  # into the parents' concatenated lists: NOT advanced behind a walker without out-edges (:214-226)
        for i in range(batch):
            n = int(deg_c[i])
            pn = int(deg_p[i]) if deg_p is not None else 0
            window = nbr_p.numpy()[cursor:cursor + pn] if pn else ()
            for x in range(off_c[i], off_c[i] + n):
                if nbr_c[x] == parent[i]:
                    biased[x] = np.float32(np.float64(w_c[x]) * 1.0 / inv_p)
                elif nbr_c[x] in window:
                    biased[x] = w_c[x]
                else:
                    biased[x] = np.float32(np.float64(w_c[x]) * 1.0 / inv_q)
            if n > 0:
                cursor += pn
        mini = dict(row_ptr=off_c, col=np.ascontiguousarray(nbr_c), eid=np.arange(nbr_c.shape[0], dtype=np.int64),
                    weight=biased, alias=self.o.alias_build(off_c, biased))
        nxt, _ = self.o.sample(mini, "EdgeWeightSampler", np.arange(batch, dtype=np.int64), 1, seed=seed,
                               call_counter=call_counter, padding_mode=1, default_neighbor_id=default_neighbor_id)
        return torch.from_numpy(nxt.reshape(-1).copy())


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


def _worker_round4(rank, world, port, q, message_limit=None):
    """FullSampler (plain and filtered), in-degrees of destination ids, the global negative tables with the three
    exclusion modes, DeepWalk and node2vec over the shards -- the protocols of glx_dist_sample_full / _in_degrees /
    _negative_create / _negative_sample / _random_walk spelled out in dist.py -- against the oracle on the whole graph."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import dist as gdist
        import synth
        from oracle_bindings import Oracle
        if message_limit:  # peer messages cut into rounds (dist.MAX_MESSAGE_BYTES)
            gdist.MAX_MESSAGE_BYTES = message_limit
        ops = OracleOps()
        orc = Oracle()
        V = 240
        rp, col, eid, w = synth.small_graph(V, 2600, seed=31, weighted=True, hub_degree=150)
        col = col.copy()
        col[col % 7 == 3] = 11  # a destination with a large in-degree; dead ends come from vertices without out-edges
        whole = dict(row_ptr=rp, col=col, eid=eid, weight=w, alias=orc.alias_build(rp, w))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
        srp, scol, seid, sw, sids = gdist.shard_graph(t(rp), t(col), t(eid), t(w), rank, world)
        shard = dict(row_ptr=srp.numpy(), col=scol.numpy(), eid=seid.numpy(), weight=sw.numpy(), ids=sids.numpy())
        shard["alias"] = orc.alias_build(shard["row_ptr"], shard["weight"])
        ts_whole = (np.arange(col.shape[0], dtype=np.int64) * 7919) % 1009
        whole["ts_slot"], shard["ts_slot"] = ts_whole[eid], ts_whole[shard["eid"]]
        whole["indeg_weight"] = shard["indeg_weight"] = None
        store = gdist.ShardedStore(ops, shard)
        rng = np.random.default_rng(500 + rank)
        bad = []

        def same(name, got, want):
            for g_, w_ in zip(got if isinstance(got, tuple) else (got,), want if isinstance(want, tuple) else (want,)):
                if not np.array_equal(np.asarray(g_), np.asarray(w_)):
                    bad.append(name)
                    return
        # requests differ per rank, in length too; unknown and negative ids, an empty request on the last rank
        src = np.concatenate([rng.integers(0, V, 90 + 23 * rank), [0, 0, -4, V, 10 ** 9]]).astype(np.int64)
        if rank == world - 1:
            src = src[:0]
        for limit in (0, 1, 4):
            got = store.sample_full(t(src), limit)
            same("full/%d" % limit, tuple(x.numpy() for x in got), orc.sample_full(whole, src, limit))
        for (ftype, ffield) in ((1, 1), (1, 2)):  # id == value, timestamp == value: the single store's answer
            vals = (rng.integers(0, V, src.shape[0]) if ffield == 1 else rng.integers(0, 1009, src.shape[0])).astype(np.int64)
            for limit, pad in ((0, 1), (3, 1), (3, 0)):
                got = store.sample_full(t(src), limit, filter_type=ftype, filter_field=ffield, values=t(vals), padding_mode=pad,
                                        default_neighbor_id=-5)
                want = orc.sample_full_filtered(whole, src, limit, dict(type=ftype, field=ffield, values=vals),
                                                padding_mode=pad, default_neighbor_id=-5) if src.shape[0] else \
                    (np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.int64))
                same("full-filtered/%d/%d/%d/%d" % (ftype, ffield, limit, pad), tuple(x.numpy() for x in got), want)
        # in-degrees: sums over ALL shards; ids nobody points to answer 0
        probe = np.concatenate([rng.integers(0, V, 60), [11, -4, V + 5]]).astype(np.int64)
        same("in-degrees", store.in_degrees(t(probe)).numpy(),
             np.array([np.count_nonzero(col == v) for v in probe], np.int32))
        # the global candidate tables: every destination id, ascending, uniform or weighted by the global in-degree
        uniq, cnt = np.unique(col, return_counts=True)
        tu, tw = store.negative_table(False), store.negative_table(True)
        same("table ids", (tu["ids"], tw["ids"]), (uniq, uniq))
        want_tw = orc.alias_build(np.array([0, uniq.shape[0]], np.int64), cnt.astype(np.float32))
        same("table alias", tuple(tw["alias"]), tuple(want_tw))
        nsrc = np.concatenate([rng.integers(0, V, 70 + 11 * rank), [11, -4, V + 5]]).astype(np.int64)
        for mode in (0, 1, 2):  # no exclusion / the source's neighbours (travels to the owners) / the batch's own ids
            for tab, want_tab in ((tu, None), (tw, want_tw)):
                for count in (1, 6):
                    got = store.negative_sample(tab, t(nsrc), count, exclude=mode, default_neighbor_id=-1, seed=9,
                                                call_counter=100 + rank)
                    want = orc.negative_sample(uniq, want_tab, mode, whole, nsrc, count, -1, 9, 100 + rank)
                    same("negatives/%d/%s/%d" % (mode, "w" if want_tab else "u", count), got.numpy(), want)
        # walks: every rank the same number of steps (a step is a collective), its own walkers
        seeds = np.concatenate([rng.integers(0, V, 40 + 9 * rank), [11, -4, V + 5]]).astype(np.int64)
        for wl in (1, 4):
            got = store.random_walk(t(seeds), wl, default_neighbor_id=-1, seed=11, call_counter=20)
            same("deepwalk/%d" % wl, got.numpy(), orc.random_walk(whole, seeds, wl, default_neighbor_id=-1, seed=11,
                                                                   call_counter=20))
        for (p_, q_, F, dflt) in ((2.0, 0.5, 100, -1), (0.25, 4.0, 3, -1), (0.5, 3.0, 7, 0)):
            # default id 0 is a vertex: a walker that got stuck walks on from it (and the cursor slips behind it)
            got = store.random_walk(t(seeds), 5, p=p_, q=q_, default_neighbor_id=dflt, seed=13, call_counter=40, full_nbr_num=F,
                                    default_weight=0.25)
            want = orc.random_walk(whole, seeds, 5, p=p_, q=q_, full_nbr_num=F, default_weight=0.25, default_neighbor_id=dflt,
                                   seed=13, call_counter=40)
            same("node2vec/%g/%g/%d/%d" % (p_, q_, F, dflt), got.numpy(), want)
        q.put((rank, bad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,message_limit", [(2, None), (3, None), (2, 200)])
def test_partitioned_full_sampler_degrees_negatives_and_walks_equal_single_shard(world, message_limit):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_round4, args=(r, world, port, q, message_limit)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(world))
    assert got == {r: [] for r in range(world)}


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

"""Sharded execution of the hot path over torch.distributed (RCCL on GPUs).

Replaces the reference's DistributeRunner (graphlearn/src/core/runner/
op_runner.h:60-152): Partition (hash_partitioner.h:33-92) -> ship sub-requests
-> Process on the owning shard -> ship results back -> Stitch
(stitcher.h:67-107).  gRPC + protobuf become two all-to-all(v) exchanges per
operator on the xGMI mesh; the bucketing and the stitch are HIP kernels
(glx_partition / glx_stitch_*).

Ownership rule (kept from the reference so results are comparable): the
out-edges and the features of vertex v live on shard llabs(v) % P
(hash_partitioner.h:90-92, graph_update_request.cc:151,234).

Results are bit-identical to the unpartitioned operator for every shard count:
  * sampling ships each row's index in the original request with its id and
    the owner draws from THAT row's random stream (glx_sample_ex);
  * aggregation, design H (the default): ships ids to the owners, gathers feature
    rows there (glx_lookup), ships the rows back (the halo exchange) and reduces
    them on the requester in the original order -- bit-identical, and the
    reference's distributed Max/Min/Prod divergence does
    not occur.  `dedup=True` ships every distinct halo id once.
  * aggregation, design R (`mode="partial"`, the reference's own scheme,
    aggregating_request.cc:117-213): ships (id, segment) to the owners, every owner
    reduces its subset (glx_aggregate), the [Sg, D] partials come back and
    glx_aggregate_stitch folds them in shard order.  Max/Min stay exact; Sum/Mean/
    Prod are re-associated across shards (within 1e-5 relative).  Cheaper than H on
    the wire when the mean segment length exceeds the shard count.

Where things run.  On GPUs (`ops` = DeviceOps) sampling, design H and LookupNodes are ONE
call into the C distributed store (include/glx.h glx_dist_*, csrc/glx_dist.hip): partition,
RCCL send/recv groups (csrc/glx_comm.hip), the owner's kernel and the stitch all happen
behind the C-ABI, and this class is a thin binding.  Design H there also keeps a replica of
hot rows on every GPU (`hot_ids`) and exchanges only the deduplicated cold tail.  The Python
orchestration below is the same protocol spelled out over torch.distributed: it is what the
CPU tests run (world size 2/3 over gloo with an oracle-backed `ops`), and it still carries
design R on GPUs.  Tensors are torch tensors throughout.

The same holds for the other partitioned operations (FullSampler with and without a filter,
in-degrees of destination ids, the global negative candidate tables and strict in-degree
negative sampling, DeepWalk / node2vec): one glx_dist_* call each on GPUs, and below that the
protocol in torch.distributed calls -- tests/test_dist_gloo.py runs it at world size 2 and 3
against the oracle on the unpartitioned graph, bit for bit.
"""
import torch
import torch.distributed as dist


class DeviceOps:
    """Local compute on this rank's GPU through the glx C-ABI."""

    native = True  # ShardedStore hands sampling / design H / lookup to the C distributed store

    def __init__(self):
        import glx
        self.glx = glx

    def partition(self, ids, num_shards):
        return self.glx.partition(ids, num_shards)

    def stitch(self, rows, order):
        return self.glx.stitch(rows, order)

    def sample(self, graph, sampler, ids, rng_rows, k, seed, cc, pad, dflt):
        return graph.sample(sampler, ids, k, seed=seed, call_counter=cc, padding_mode=pad,
                            default_neighbor_id=dflt, rng_rows=rng_rows)

    def sample_filtered(self, graph, sampler, ids, rng_rows, k, seed, cc, pad, dflt, ftype, ffield, values, retry,
                        default_ts):
        return graph.sample_filtered(sampler, ids, k, ftype, ffield, values, seed=seed, call_counter=cc,
                                     padding_mode=pad, default_neighbor_id=dflt, retry_times=retry,
                                     default_timestamp=default_ts, rng_rows=rng_rows)

    def lookup(self, feats, ids, default_attr):
        return feats.lookup(ids, default_attr)

    def aggregate_rows(self, rows, pos, seg, num_segments, op, default_attr):
        view = self.glx.Features(rows, view=True, device=rows.device.index or 0)
        return view.aggregate(op, pos, seg, num_segments, default_attr)

    def aggregate_local(self, feats, op, node_ids, seg, num_segments, default_attr):
        return feats.aggregate(op, node_ids, seg, num_segments, default_attr)

    def aggregate_stitch(self, op, parts, cnts, default_attr):
        return self.glx.aggregate_stitch(op, parts, cnts, default_attr)


def _staged(x, group):
    """gloo cannot move device tensors: when the group is gloo and the tensor lives on a
    GPU (the 2-ranks-on-one-GPU test rig), collectives bounce through host memory.  With
    the production backend (nccl = RCCL) nothing is staged."""
    return x.is_cuda and dist.get_backend(group) == "gloo"


# RCCL (2.26, ROCm 7) delivers only the first half of an all-to-all message larger than
# 1 GiB (measured with world_size 1: bytes beyond message_size / 2 are left untouched;
# scripts/probe_rccl_large_messages.py).  No peer message is allowed to exceed this many bytes; larger
# exchanges are cut into rounds.  all_gather is chunked the same way to stay clear of the limit.
MAX_MESSAGE_BYTES = 512 << 20


def _a2a_once(x, send_counts, recv_counts, group):
    shape = (int(sum(recv_counts)),) + tuple(x.shape[1:])
    if _staged(x, group):
        xc = x.contiguous().cpu()
        out = xc.new_empty(shape)
        dist.all_to_all_single(out, xc, output_split_sizes=list(recv_counts),
                               input_split_sizes=list(send_counts), group=group)
        return out.to(x.device)
    out = x.new_empty(shape)
    dist.all_to_all_single(out, x.contiguous(), output_split_sizes=list(recv_counts),
                           input_split_sizes=list(send_counts), group=group)
    return out


def _a2a(x, send_counts, recv_counts, group, bound):
    """all-to-all(v) of the rows of x: send_counts[p] consecutive rows go to rank p.
    `bound` = the largest row count any rank sends to any rank in this exchange, known to
    EVERY rank (from the gathered count matrix), so that all ranks agree on the number of
    rounds without another collective."""
    width = 1
    for d in x.shape[1:]:
        width *= int(d)
    step = max(1, MAX_MESSAGE_BYTES // max(1, x.element_size() * width))  # rows per peer per round
    most = int(bound)
    if most <= step:
        return _a2a_once(x, send_counts, recv_counts, group)
    out = x.new_empty((int(sum(recv_counts)),) + tuple(x.shape[1:]))
    send_off = [0]
    for c in send_counts:
        send_off.append(send_off[-1] + int(c))
    recv_off = [0]
    for c in recv_counts:
        recv_off.append(recv_off[-1] + int(c))
    for lo in range(0, most, step):  # the same number of rounds on every rank
        sc = [max(0, min(int(c) - lo, step)) for c in send_counts]
        rc = [max(0, min(int(c) - lo, step)) for c in recv_counts]
        part = torch.cat([x[send_off[p] + lo:send_off[p] + lo + sc[p]] for p in range(len(sc))])
        got = _a2a_once(part, sc, rc, group)
        at = 0
        for p in range(len(rc)):
            out[recv_off[p] + lo:recv_off[p] + lo + rc[p]] = got[at:at + rc[p]]
            at += rc[p]
    return out


def comm_for_group(group=None, device=0):
    """A glx communicator with the ranks of a torch.distributed group: RCCL (unique id made by
    rank 0, handed out through the group) when the group's backend is nccl; for gloo -- ranks
    that cannot run RCCL together, e.g. several ranks sharing one GPU in the test rig -- the
    host-staged transport with the group's own all-to-all / all-gather behind it."""
    import glx
    import numpy as np
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) != "gloo":
        box = [glx.Comm.unique_id() if rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group, device=torch.device("cuda", device))
        return glx.Comm.rccl(device, rank, world, box[0])

    def a2a(send, send_counts, recv, recv_counts, eb):
        out = torch.from_numpy(recv)
        dist.all_to_all_single(out, torch.from_numpy(send), output_split_sizes=[int(c) * eb for c in recv_counts],
                               input_split_sizes=[int(c) * eb for c in send_counts], group=group)

    def gather(send, recv):
        parts = list(torch.from_numpy(recv).view(world, -1).unbind(0))
        dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(send)), group=group)
    return glx.Comm.callbacks(device, rank, world, a2a, gather)


class ShardedStore:
    """One rank's view of an edge-cut partitioned graph + feature store."""

    def __init__(self, ops, graph_shard, feature_shard=None, group=None, feature_replica=None, comm=None,
                 hot_ids=None):
        """feature_shard: this rank's rows (halo exchange per request, design H);
        feature_replica: a full copy of the feature table on this GPU (built once by
        `replicate_features`, i.e. the halo exchange done at load time);
        hot_ids: ids (the same list on every rank) whose rows every rank keeps a copy of; design H
        then exchanges only the deduplicated cold tail.  None / empty: no hot-row replica.
        comm: a glx.Comm to use instead of one derived from `group` (device ops only)."""
        self.ops = ops
        self.graph = graph_shard
        self.feats = feature_shard
        self.replica = feature_replica
        self.group = group
        self.native = None
        self.cache = None
        self.last_stats = None
        if getattr(ops, "native", False):
            dev = (graph_shard if graph_shard is not None else feature_shard).device
            self.comm = comm if comm is not None else comm_for_group(group, dev)
            self.world, self.rank = self.comm.world, self.comm.rank
            self.native = ops.glx.DistStore(self.comm, graph=graph_shard, features=feature_shard)
            if hot_ids is not None and feature_shard is not None:
                self.native.set_cache(hot_ids)
            return
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if hot_ids is not None and len(hot_ids) and feature_shard is not None:
            self.cache = self._build_cache(torch.as_tensor(hot_ids, dtype=torch.int64))

    def _build_cache(self, hot):
        """Every rank looks up the hot ids it owns and all-gathers the rows: (sorted ids, rows).  Ids
        their owner does not know stay out (a request's own default applies to them)."""
        owner = hot.abs() % self.world
        mine = hot[owner == self.rank]
        rows_idx = self.ops.rows_of(self.feats, mine)
        known = rows_idx >= 0
        mine = mine[known]
        rows = self.ops.lookup(self.feats, mine, 0.0)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(counts, torch.tensor([mine.shape[0]], dtype=torch.int64), group=self.group)
        counts = [int(c.item()) for c in counts]
        most = max(counts + [1])
        pad_ids = torch.zeros(most, dtype=torch.int64)
        pad_ids[:mine.shape[0]] = mine
        pad_rows = torch.zeros((most, rows.shape[1]), dtype=rows.dtype)
        pad_rows[:mine.shape[0]] = rows
        all_ids = [torch.empty_like(pad_ids) for _ in range(self.world)]
        all_rows = [torch.empty_like(pad_rows) for _ in range(self.world)]
        dist.all_gather(all_ids, pad_ids, group=self.group)
        dist.all_gather(all_rows, pad_rows, group=self.group)
        ids = torch.cat([a[:c] for a, c in zip(all_ids, counts)])
        rows = torch.cat([a[:c] for a, c in zip(all_rows, counts)])
        srt = torch.argsort(ids)
        return ids[srt].contiguous(), rows[srt].contiguous()

    def _route(self, ids):
        """Bucket ids by owner and share the whole count matrix: -> bucketed, order, rows this
        rank sends to each rank, rows it receives from each rank, and the global maximum."""
        bucketed, order, counts = self.ops.partition(ids, self.world)
        mine = counts.cpu() if _staged(counts, self.group) else counts
        rows = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(rows, mine, group=self.group)
        matrix = torch.stack(rows).cpu()  # matrix[q][p] = rows rank q sends to rank p
        return (bucketed, order, matrix[self.rank].tolist(), matrix[:, self.rank].tolist(), int(matrix.max().item()))

    def sample(self, sampler, src, k, seed=0, call_counter=0, padding_mode=1, default_neighbor_id=0):
        if self.native is not None:
            return self.native.sample(sampler, src, k, seed=seed, call_counter=call_counter,
                                      padding_mode=padding_mode, default_neighbor_id=default_neighbor_id)
        bucketed, order, send, recv, most = self._route(src)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        rows_in = _a2a(order, send, recv, self.group, most)  # original row index = random stream
        nbr, eid = self.ops.sample(self.graph, sampler, ids_in, rows_in, k, seed, call_counter,
                                   padding_mode, default_neighbor_id)
        nbr = _a2a(nbr, recv, send, self.group, most)
        eid = _a2a(eid, recv, send, self.group, most)
        return self.ops.stitch(nbr, order), self.ops.stitch(eid, order)

    def _native(self, what):
        if self.native is None:
            raise NotImplementedError("%s runs in the C distributed store (device ops)" % what)
        return self.native

    # ---- the partitioned operations of glx_dist_* beyond the dense samplers.  On GPUs each is ONE call into the C store;
    # ---- below it, the same protocol spelled out over torch.distributed (what the CPU tests run over gloo) -----------
    def _all_max(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def _gather_params(self, values):
        """Every rank's request parameters, requester-major: owners serve each requester with ITS parameters
        (glx_dist_set_params_kernel: they ride with the bucket counts of the count exchange)."""
        mine = torch.tensor([int(v) for v in values], dtype=torch.int64)
        rows = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(rows, mine, group=self.group)
        return [r.tolist() for r in rows]

    def sample_full(self, src, max_limit=0, filter_type=0, filter_field=0, values=None, padding_mode=1,
                    default_neighbor_id=0, default_timestamp=-1):
        """FullSampler over the shards (sparse response, full_sampler.cc:31-78 behind DistributeRunner): -> (degrees,
        nbr, eid) of this rank's rows, rows in request order.  With filter_type / filter_field / values: the filtered
        FullSampler -- every row's value travels with its id and every owner serves each requester's part with ONE call
        (a timestamp > value filter reads the first value of the part a server is given: filter.h:107-111)."""
        if self.native is not None:
            return self.native.sample_full(src, max_limit, filter_type=filter_type, filter_field=filter_field,
                                           values=values, padding_mode=padding_mode,
                                           default_neighbor_id=default_neighbor_id, default_timestamp=default_timestamp)
        filtered = filter_type != 0
        kinds = self._gather_params([filter_type, filter_field, default_timestamp, padding_mode, default_neighbor_id,
                                     max_limit])
        for q, kq in enumerate(kinds):
            if (kq[0] != 0) != filtered:
                raise ValueError("rank %d and this rank disagree on whether the FullSampler request has a filter" % q)
            if kq[5] != max_limit:
                raise ValueError("rank %d asks for at most %d neighbours per row, this rank for %d" % (q, kq[5], max_limit))
        bucketed, order, send, recv, most = self._route(src)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        vals_in = _a2a(values[order].contiguous(), send, recv, self.group, most) if filtered else None
        degs, nbrs, eids, totals = [], [], [], []
        at = 0
        for q in range(self.world):  # one call per requester: its rows, its values, its filter kind / padding / default id
            part = ids_in[at:at + recv[q]]
            if filtered:
                kq = kinds[q]
                d, n, e = self.ops.sample_full_filtered(self.graph, part, max_limit, kq[0], kq[1], vals_in[at:at + recv[q]],
                                                        kq[3], kq[4], kq[2])
            else:
                d, n, e = self.ops.sample_full(self.graph, part, max_limit)
            degs.append(d)
            nbrs.append(n)
            eids.append(e)
            totals.append(int(n.shape[0]))
            at += recv[q]
        deg_b = _a2a(torch.cat(degs), recv, send, self.group, most)  # the sizes half: back in bucketed order
        # the lists half: what every owner sends this rank is the sum of the sizes it just announced
        got, at = [], 0
        for p_ in range(self.world):
            got.append(int(deg_b[at:at + send[p_]].sum().item()))
            at += send[p_]
        bound = self._all_max(max(totals + [0]))
        nbr_b = _a2a(torch.cat(nbrs), totals, got, self.group, bound)
        eid_b = _a2a(torch.cat(eids), totals, got, self.group, bound)
        # stitch: row j of the bucketed request is row order[j] of the caller's
        n = int(src.shape[0])
        i64 = torch.int64
        deg_out = torch.zeros(n, dtype=deg_b.dtype)
        deg_out[order] = deg_b
        off_b = torch.zeros(n + 1, dtype=i64)
        off_b[1:] = torch.cumsum(deg_b.to(i64), 0)
        off_out = torch.zeros(n + 1, dtype=i64)
        off_out[1:] = torch.cumsum(deg_out.to(i64), 0)
        pos_b = torch.empty(n, dtype=i64)
        pos_b[order] = torch.arange(n, dtype=i64)
        total = int(off_out[-1].item())
        row_of = torch.repeat_interleave(torch.arange(n, dtype=i64), deg_out.to(i64), output_size=total)
        from_b = off_b[pos_b[row_of]] + (torch.arange(total, dtype=i64) - off_out[row_of])
        return deg_out, nbr_b[from_b], eid_b[from_b]

    def _owner_table(self):
        """Collective, once: the destination ids this rank OWNS (llabs(id) % P), ascending, with their in-degree summed
        over ALL shards -- every shard run-length-encodes its own destinations, the (id, count) pairs travel to the
        owners, the owners reduce by key (dst_totals in glx_dist.hip; the per-owner half of the unpartitioned storage's
        GetAllDstIds() / GetAllInDegrees(), topo_statics.cc:32-69)."""
        if getattr(self, "_own", None) is None:
            uniq, cnt = self.ops.dst_counts(self.graph)
            bucketed, order, send, recv, most = self._route(uniq)
            ids_in = _a2a(bucketed, send, recv, self.group, most)
            cnt_in = _a2a(cnt[order].contiguous(), send, recv, self.group, most)
            own_id, inverse = torch.unique(ids_in, return_inverse=True)  # ascending
            own_cnt = torch.zeros(own_id.shape[0], dtype=torch.int64)
            own_cnt.index_add_(0, inverse, cnt_in)
            self._own = (own_id, own_cnt)
        return self._own

    def in_degrees(self, ids):
        """Collective: in-degrees of destination ids summed over ALL shards (GetDegree with node_from = dst,
        degree_getter.cc; GraphStorage::GetInDegree, topo_statics.cc:62-69); ids nobody points to answer 0."""
        if self.native is not None:
            return self.native.in_degrees(ids)
        own_id, own_cnt = self._owner_table()
        bucketed, order, send, recv, most = self._route(ids)
        asked = _a2a(bucketed, send, recv, self.group, most)
        if own_id.shape[0]:
            at = torch.searchsorted(own_id, asked).clamp(max=own_id.shape[0] - 1)
            answer = torch.where(own_id[at] == asked, own_cnt[at], torch.zeros_like(asked))
        else:
            answer = torch.zeros_like(asked)
        back = _a2a(answer, recv, send, self.group, most)
        return self.ops.stitch(back, order).to(torch.int32)

    def negative_table(self, by_in_degree=False):
        """Collective: the negative samplers' candidate list over the WHOLE edge type (every shard's destination ids,
        ascending -- the one order every shard count agrees on; uniform or weighted by global in-degree), identical on every
        rank (random_negative_sampler.cc:30-63 / in_degree_negative_sampler.cc:29-135 draw from the storage's
        GetAllDstIds() / GetAllInDegrees())."""
        if self.native is not None:
            return self.native.negative_table(by_in_degree)
        own_id, own_cnt = self._owner_table()
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        dist.all_gather(sizes, torch.tensor([own_id.shape[0]], dtype=torch.int64), group=self.group)
        sizes = [int(x.item()) for x in sizes]
        most = max(sizes + [1])
        pad = torch.zeros((2, most), dtype=torch.int64)
        pad[0, :own_id.shape[0]] = own_id
        pad[1, :own_cnt.shape[0]] = own_cnt
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        ids = torch.cat([a[0, :c] for a, c in zip(parts, sizes)])
        cnt = torch.cat([a[1, :c] for a, c in zip(parts, sizes)])
        srt = torch.argsort(ids)  # owners hold disjoint id sets: no ties
        return self.ops.negative_table(ids[srt].contiguous(), cnt[srt].to(torch.float32) if by_in_degree else None)

    def negative_sample(self, table, src, count, exclude=0, default_neighbor_id=0, seed=0, call_counter=0):
        """Negative sampling over the shards from `table` (negative_table()): what an unpartitioned store with the same
        table answers.  Collective when exclude = 1 (glx.NEG_EXCLUDE_NEIGHBORS, strict in-degree sampling): a row's
        exclusion set is its source vertex's adjacency, which lives with the row's owner -- the rows travel there with
        their index in the request (the random stream they draw from) and the answers travel back."""
        if self.native is not None:
            return self.native.negative_sample(table, src, count, exclude=exclude, default_neighbor_id=default_neighbor_id,
                                               seed=seed, call_counter=call_counter)
        if exclude != 1:
            return self.ops.negative_sample(table, exclude, self.graph, src, None, count, default_neighbor_id, seed,
                                            call_counter)
        params = self._gather_params([seed, call_counter, count, default_neighbor_id])
        for q, pq in enumerate(params):
            if pq[2] != count:
                raise ValueError("rank %d asks for %d negatives per row, this rank for %d" % (q, pq[2], count))
        bucketed, order, send, recv, most = self._route(src)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        rows_in = _a2a(order, send, recv, self.group, most)
        outs, at = [], 0
        for q in range(self.world):  # every requester's rows with ITS seed / call counter / default id
            pq = params[q]
            outs.append(self.ops.negative_sample(table, 1, self.graph, ids_in[at:at + recv[q]], rows_in[at:at + recv[q]],
                                                 count, pq[3], pq[0], pq[1]))
            at += recv[q]
        back = _a2a(torch.cat(outs), recv, send, self.group, most)
        return self.ops.stitch(back, order)

    def random_walk(self, seeds, walk_len, p=1.0, q=1.0, default_neighbor_id=0, seed=0, call_counter=0, full_nbr_num=100,
                    default_weight=0.0):
        """Collective RandomWalk over the shards -> walks[batch, walk_len], the single store's walks.
        DeepWalk (p = q = 1; random_walk.cc:168-190): step t of walker i = RandomSampler's draw 0 of the stream (seed,
        call_counter + t, i) on the vertex it stands on -- one partitioned request with neighbor_count 1 per step.
        node2vec (WeightedRandomWalk, random_walk.cc:192-272): per step ONE partitioned FullSampler request (limit F)
        brings the current vertices' first F neighbours and their weights to the requester; the parents' lists are the
        previous step's, read through the reference's cursor (not advanced for walkers whose current vertex has no
        out-edges: random_walk.cc:214-226); the step itself runs on the requester."""
        if self.native is not None:
            return self.native.random_walk(seeds, walk_len, p=p, q=q, default_neighbor_id=default_neighbor_id, seed=seed,
                                           call_counter=call_counter, full_nbr_num=full_nbr_num,
                                           default_weight=default_weight)
        eps = 32 * 1.1920928955078125e-07  # RandomWalkRequest::IsDeepWalk, random_walk_request.cc:152-160
        f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))  # noqa: E731
        deep = abs(f32(p) - 1.0) < eps and abs(f32(q) - 1.0) < eps
        batch = int(seeds.shape[0])
        walks = torch.empty((batch, walk_len), dtype=torch.int64)
        cur = seeds
        if deep:
            for t in range(walk_len):
                nxt, _ = self.sample("RandomSampler", cur, 1, seed=seed, call_counter=call_counter + t, padding_mode=1,
                                     default_neighbor_id=default_neighbor_id)
                walks[:, t] = nxt.reshape(-1)
                cur = nxt.reshape(-1).contiguous()
            return walks
        if not 1 <= full_nbr_num <= 2048:
            raise ValueError("DefaultFullNbrNum must be in [1, 2048], got %d" % full_nbr_num)
        par = seeds
        prev = None  # (deg, nbr) of the previous step's current vertices, walker by walker
        for t in range(walk_len):
            deg_c, nbr_c, w_c = self._full_lists_with_weights(cur, full_nbr_num, default_weight)
            nxt = self.ops.node2vec_step(par, deg_c, nbr_c, w_c, prev[0] if prev else None, prev[1] if prev else None,
                                         p, q, seed, call_counter + t, default_neighbor_id)
            walks[:, t] = nxt
            # the next step's parent is this step's vertex -- except after step 0, whose parent stays the seed
            # (random_walk_request.cc:120-131: t <= 1 -> the seed): the seed IS this step's vertex then
            par, cur, prev = cur, nxt.contiguous(), (deg_c, nbr_c)
        return walks

    def _full_lists_with_weights(self, ids, limit, default_weight):
        """The partitioned FullSampler request of a node2vec step: -> (degrees, nbr, weight) in request order; the
        answer carries edge weights where the edge ids would be (glx_dist_slot_weights_kernel)."""
        bucketed, order, send, recv, most = self._route(ids)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        d, n, w = self.ops.full_lists_with_weights(self.graph, ids_in, limit, default_weight)
        totals, at = [], 0
        off = torch.zeros(d.shape[0] + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(d.to(torch.int64), 0)
        for q in range(self.world):
            totals.append(int((off[at + recv[q]] - off[at]).item()))
            at += recv[q]
        deg_b = _a2a(d, recv, send, self.group, most)
        got, at = [], 0
        for p_ in range(self.world):
            got.append(int(deg_b[at:at + send[p_]].sum().item()))
            at += send[p_]
        bound = self._all_max(max(totals + [0]))
        nbr_b = _a2a(n, totals, got, self.group, bound)
        w_b = _a2a(w, totals, got, self.group, bound)
        m = int(ids.shape[0])
        i64 = torch.int64
        deg_out = torch.zeros(m, dtype=deg_b.dtype)
        deg_out[order] = deg_b
        off_b = torch.zeros(m + 1, dtype=i64)
        off_b[1:] = torch.cumsum(deg_b.to(i64), 0)
        off_out = torch.zeros(m + 1, dtype=i64)
        off_out[1:] = torch.cumsum(deg_out.to(i64), 0)
        pos_b = torch.empty(m, dtype=i64)
        pos_b[order] = torch.arange(m, dtype=i64)
        total = int(off_out[-1].item())
        row_of = torch.repeat_interleave(torch.arange(m, dtype=i64), deg_out.to(i64), output_size=total)
        from_b = off_b[pos_b[row_of]] + (torch.arange(total, dtype=i64) - off_out[row_of])
        return deg_out, nbr_b[from_b], w_b[from_b]

    def sample_filtered(self, sampler, src, k, filter_type, filter_field, values, seed=0, call_counter=0,
                        padding_mode=1, default_neighbor_id=0, retry_times=5, default_timestamp=-1):
        """sample() for a request with an op::Filter: every row's filter value travels with its id (as
        HashPartitioner copies every tensor of a request, hash_partitioner.h:69-74).  Id filters and
        timestamp == value give the single-store answer draw for draw.  timestamp > value does not: the
        reference's ActOn reads the value of the FIRST row of whatever request a server sees
        (filter.h:107-111), so each shard uses its own part's first value -- as the reference's servers do."""
        if self.native is not None:
            return self.native.sample(sampler, src, k, seed=seed, call_counter=call_counter,
                                      padding_mode=padding_mode, default_neighbor_id=default_neighbor_id,
                                      filter_type=filter_type, filter_field=filter_field, values=values,
                                      retry_times=retry_times, default_timestamp=default_timestamp)
        bucketed, order, send, recv, most = self._route(src)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        rows_in = _a2a(order, send, recv, self.group, most)
        vals_in = _a2a(values[order].contiguous(), send, recv, self.group, most)
        nbr, eid = self.ops.sample_filtered(self.graph, sampler, ids_in, rows_in, k, seed, call_counter, padding_mode,
                                            default_neighbor_id, filter_type, filter_field, vals_in, retry_times,
                                            default_timestamp)
        nbr = _a2a(nbr, recv, send, self.group, most)
        eid = _a2a(eid, recv, send, self.group, most)
        return self.ops.stitch(nbr, order), self.ops.stitch(eid, order)

    def aggregate(self, op, node_ids, segment_ids, num_segments, default_attr=0.0, mode="halo", dedup=True,
                  out=None):
        if self.replica is not None:
            return self.ops.aggregate_local(self.replica, op, node_ids, segment_ids, num_segments,
                                            default_attr)
        if mode == "partial":
            return self._aggregate_partial(op, node_ids, segment_ids, num_segments, default_attr)
        assert mode == "halo", mode
        if self.native is not None:
            res = self.native.aggregate(op, node_ids, segment_ids, num_segments, default_attr, out=out)
            return res
        # ---- the protocol of glx_dist_aggregate, spelled out (CPU tests run this) ----
        n = node_ids.shape[0]
        P, me = self.world, self.rank
        # 1. where does every id's row come from: hot-row replica, own shard, or the halo
        slot = torch.full((n,), -1, dtype=torch.int64)
        if self.cache is not None and self.cache[0].shape[0]:
            cids = self.cache[0]
            at = torch.searchsorted(cids, node_ids).clamp(max=cids.shape[0] - 1)
            slot = torch.where(cids[at] == node_ids, at, slot)
        hit = slot >= 0
        mine = (~hit) & ((node_ids.abs() % P) == me)
        cold = ~(hit | mine)
        # 2. the cold tail: every distinct remote id crosses the links once
        cold_ids = node_ids[cold]
        if dedup:
            distinct, inverse = torch.unique(cold_ids, return_inverse=True)
        else:
            distinct, inverse = cold_ids, torch.arange(cold_ids.shape[0])
        bucketed, order, send, recv, most = self._route(distinct)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        rows_out = self.ops.lookup(self.feats, ids_in, default_attr)
        halo = _a2a(rows_out, recv, send, self.group, most)  # halo rows, in bucketed order
        m = distinct.shape[0]
        pos_of_distinct = torch.empty(m, dtype=torch.int64)
        pos_of_distinct[order] = torch.arange(m, dtype=torch.int64)
        # 3. one table [halo | replica | own rows of this request | default row], reduced in request order
        own_rows = self.ops.lookup(self.feats, node_ids[mine], default_attr)
        parts = [halo]
        base_cache = halo.shape[0]
        if self.cache is not None:
            parts.append(self.cache[1])
        base_own = base_cache + (self.cache[1].shape[0] if self.cache is not None else 0)
        parts.append(own_rows)
        table = torch.cat(parts)
        pos = torch.empty(n, dtype=torch.int64)
        pos[hit] = base_cache + slot[hit]
        pos[mine] = base_own + torch.arange(int(mine.sum()), dtype=torch.int64)
        pos[cold] = pos_of_distinct[inverse]
        self.last_stats = dict(ids=n, from_replica=int(hit.sum()), from_own_shard=int(mine.sum()),
                               remote=int(cold.sum()), remote_distinct=int(m), served_rows=int(ids_in.shape[0]))
        return self.ops.aggregate_rows(table, pos, segment_ids, num_segments, op, default_attr)

    def lookup(self, node_ids, default_attr=0.0):
        """LookupNodes in distributed mode (float attributes)."""
        assert self.native is not None, "device stores only"
        return self.native.lookup(node_ids, default_attr)

    def stats(self):
        return self.native.stats() if self.native is not None else self.last_stats

    def _aggregate_partial(self, op, node_ids, segment_ids, num_segments, default_attr):
        """Design R: owners reduce, the requester folds the partials (AggregatingRequest::
        Partition / AggregatingResponse::Stitch, aggregating_request.cc:117-213)."""
        bucketed, order, send, recv, most = self._route(node_ids)
        # every owner needs each requester's segment count (requests differ per rank)
        mine = torch.tensor([int(num_segments)], dtype=torch.int64)
        if dist.get_backend(self.group) != "gloo":
            mine = mine.to(node_ids.device)
        sgs = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(sgs, mine, group=self.group)
        sg_of = [int(x.item()) for x in sgs]
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        seg_in = _a2a(segment_ids[order].contiguous(), send, recv, self.group, most)  # stays non-decreasing per requester
        embs, cnts = [], []
        at = 0
        for q in range(self.world):
            e, c = self.ops.aggregate_local(self.feats, op, ids_in[at:at + recv[q]], seg_in[at:at + recv[q]],
                                            sg_of[q], default_attr)
            embs.append(e)
            cnts.append(c)
            at += recv[q]
        back = [int(num_segments)] * self.world
        parts = _a2a(torch.cat(embs), sg_of, back, self.group, max(sg_of))   # [P * Sg, D], shard-major
        pcnt = _a2a(torch.cat(cnts), sg_of, back, self.group, max(sg_of))    # [P * Sg]
        return self.ops.aggregate_stitch(op, parts.view(self.world, num_segments, -1),
                                         pcnt.view(self.world, num_segments), default_attr)


def rows_of_graph(row_ptr, col, eid, weight, ids):
    """The rows `ids` (int64 tensor of dense vertex ids) of a (dense-id, torch) CSR, complete and in storage order.
    -> (row_ptr, col, eid, weight, ids) of the sub-graph; col keeps GLOBAL ids."""
    deg = row_ptr[ids + 1] - row_ptr[ids]
    rp = torch.zeros(ids.shape[0] + 1, dtype=torch.int64, device=row_ptr.device)
    rp[1:] = torch.cumsum(deg, 0)
    # slot indices of the kept rows, row-major
    total = int(rp[-1].item())
    row_of_slot = torch.repeat_interleave(torch.arange(ids.shape[0], device=row_ptr.device), deg,
                                          output_size=total)
    slot = row_ptr[ids][row_of_slot] + (torch.arange(total, device=row_ptr.device) - rp[row_of_slot])
    w = weight[slot].contiguous() if weight is not None else None
    return rp, col[slot].contiguous(), eid[slot].contiguous(), w, ids


def shard_graph(row_ptr, col, eid, weight, rank, world):
    """Rows of the (dense-id, torch) CSR owned by `rank`: v % world == rank."""
    V = row_ptr.shape[0] - 1
    ids = (torch.arange(rank, V, world, dtype=torch.int64, device=row_ptr.device) if rank < V
           else torch.empty(0, dtype=torch.int64, device=row_ptr.device))  # more shards than vertices: an empty shard
    return rows_of_graph(row_ptr, col, eid, weight, ids)


def replicate_features(x_shard, num_nodes, group=None):
    """Load-time halo exchange: every rank holds rows rank::world of the [V, D]
    table; one RCCL all-gather gives each GPU the whole table in id order."""
    world = dist.get_world_size(group)
    per = (num_nodes + world - 1) // world
    pad = x_shard
    if x_shard.shape[0] < per:
        pad = torch.cat([x_shard, x_shard.new_zeros((per - x_shard.shape[0], x_shard.shape[1]))])
    if _staged(x_shard, group):
        pc = pad.contiguous().cpu()
        parts = [torch.empty_like(pc) for _ in range(world)]
        dist.all_gather(parts, pc, group=group)
        gathered = torch.stack(parts).to(x_shard.device)
    else:
        gathered = x_shard.new_empty((world, per, x_shard.shape[1]))
        pad = pad.contiguous()
        blk = max(1, MAX_MESSAGE_BYTES // (4 * int(x_shard.shape[1])))  # rows per rank per call
        if per <= blk:
            dist.all_gather_into_tensor(gathered.view(world * per, -1), pad, group=group)
        else:
            for lo in range(0, per, blk):
                hi = min(per, lo + blk)
                piece = x_shard.new_empty((world, hi - lo, x_shard.shape[1]))
                dist.all_gather_into_tensor(piece.view(world * (hi - lo), -1), pad[lo:hi], group=group)
                gathered[:, lo:hi] = piece
    # row v lives at gathered[v % world][v // world]
    return gathered.permute(1, 0, 2).reshape(world * per, -1)[:num_nodes].contiguous()

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

`ops` abstracts the local compute: DeviceOps (HIP, the product) or, in the CPU
tests only, an oracle-backed stand-in.  Tensors are torch tensors throughout.
"""
import torch
import torch.distributed as dist


class DeviceOps:
    """Local compute on this rank's GPU through the glx C-ABI."""

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


class ShardedStore:
    """One rank's view of an edge-cut partitioned graph + feature store."""

    def __init__(self, ops, graph_shard, feature_shard=None, group=None, feature_replica=None):
        """feature_shard: this rank's rows (halo exchange per request, design H);
        feature_replica: a full copy of the feature table on this GPU (built once by
        `replicate_features`, i.e. the halo exchange done at load time) -- the
        MI355X-first placement whenever V*D*4 bytes fit next to the graph in 288 GB."""
        self.ops = ops
        self.graph = graph_shard
        self.feats = feature_shard
        self.replica = feature_replica
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

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
        bucketed, order, send, recv, most = self._route(src)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        rows_in = _a2a(order, send, recv, self.group, most)  # original row index = random stream
        nbr, eid = self.ops.sample(self.graph, sampler, ids_in, rows_in, k, seed, call_counter,
                                   padding_mode, default_neighbor_id)
        nbr = _a2a(nbr, recv, send, self.group, most)
        eid = _a2a(eid, recv, send, self.group, most)
        return self.ops.stitch(nbr, order), self.ops.stitch(eid, order)

    def sample_filtered(self, sampler, src, k, filter_type, filter_field, values, seed=0, call_counter=0,
                        padding_mode=1, default_neighbor_id=0, retry_times=5, default_timestamp=-1):
        """sample() for a request with an op::Filter: every row's filter value travels with its id (as
        HashPartitioner copies every tensor of a request, hash_partitioner.h:69-74).  Id filters and
        timestamp == value give the single-store answer draw for draw.  timestamp > value does not: the
        reference's ActOn reads the value of the FIRST row of whatever request a server sees
        (filter.h:107-111), so each shard uses its own part's first value -- as the reference's servers do."""
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

    def aggregate(self, op, node_ids, segment_ids, num_segments, default_attr=0.0, mode="halo", dedup=False):
        if self.replica is not None:
            return self.ops.aggregate_local(self.replica, op, node_ids, segment_ids, num_segments,
                                            default_attr)
        if mode == "partial":
            return self._aggregate_partial(op, node_ids, segment_ids, num_segments, default_attr)
        assert mode == "halo", mode
        n = node_ids.shape[0]
        inverse = None
        if dedup:
            # every distinct id crosses the links once; the reduce reads the halo table through `inverse`
            node_ids, inverse = torch.unique(node_ids, return_inverse=True)
        bucketed, order, send, recv, most = self._route(node_ids)
        ids_in = _a2a(bucketed, send, recv, self.group, most)
        rows = self.ops.lookup(self.feats, ids_in, default_attr)
        rows = _a2a(rows, recv, send, self.group, most)  # halo rows, in bucketed order
        # pos[i] = where (distinct) element i sits in `rows` (inverse of `order`)
        m = node_ids.shape[0]
        pos = self.ops.stitch(torch.arange(m, dtype=torch.int64, device=node_ids.device).view(m, 1),
                              order).view(m)
        if inverse is not None:
            pos = pos[inverse]
        return self.ops.aggregate_rows(rows, pos, segment_ids, num_segments, op, default_attr)

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


def shard_graph(row_ptr, col, eid, weight, rank, world):
    """Rows of the (dense-id, torch) CSR owned by `rank`: v % world == rank.
    -> (row_ptr, col, eid, weight, ids) of the shard; col keeps GLOBAL ids."""
    V = row_ptr.shape[0] - 1
    ids = torch.arange(rank, V, world, dtype=torch.int64, device=row_ptr.device)
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

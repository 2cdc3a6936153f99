"""numpy bindings of the test oracles.

  * `Oracle`  -- oracle/libglx_oracle.so, the C restatement under the glx
                 seeding contract (bit-exact target of the HIP kernels);
  * `RefLib`  -- oracle/_ref/libglref.so, the reference's OWN sampler /
                 aggregator sources shim-compiled (present when built in a
                 container that has /root/reference; it travels to the GPU box
                 as a prebuilt .so).
Test infrastructure only.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libglx_oracle.so")
REF_SO = os.environ.get("GLX_REF_LIB") or os.path.join(ROOT, "oracle", "_ref", "libglref.so")

VP = ctypes.c_void_p
SAMPLERS = ["RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "TopkSampler"]
ALL_SAMPLERS = SAMPLERS + ["InDegreeSampler"]  # ids of the C-ABI / oracle, in order
AGGREGATORS = ["SumAggregator", "MeanAggregator", "MaxAggregator", "MinAggregator", "ProdAggregator"]


def _p(a):
    return None if a is None else a.ctypes.data_as(VP)


class _CGraph(ctypes.Structure):
    _fields_ = [("V", ctypes.c_int64), ("E", ctypes.c_int64), ("row_ptr", VP), ("col", VP), ("eid", VP),
                ("weight", VP), ("alias_prob", VP), ("alias_idx", VP), ("ids", VP), ("indeg_prob", VP),
                ("indeg_alias", VP)]


class _CFilter(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("field", ctypes.c_int), ("values", VP), ("retry_times", ctypes.c_int32),
                ("ts_slot", VP), ("default_timestamp", ctypes.c_int64), ("indeg_weight", VP)]


FILTER_EQUAL, FILTER_LARGER_THAN = 1, 2  # include/constants.h:135-139
FIELD_ID, FIELD_TIMESTAMP = 1, 2         # include/constants.h:141-145


class Oracle:
    def __init__(self):
        L = ctypes.CDLL(ORACLE_SO)
        i32, i64, u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
        L.glxo_philox4x32_10.argtypes = [VP, VP, VP]
        L.glxo_draw64.argtypes = [u64, u64, ctypes.c_uint32, ctypes.c_uint32]
        L.glxo_draw64.restype = u64
        L.glxo_alias_build.argtypes = [VP, VP, i64, VP, VP]
        L.glxo_sort_rows_by_weight_desc.argtypes = [VP, i64, VP, VP, VP]
        L.glxo_sort_rows_by_timestamp_asc.argtypes = [VP, i64, VP, VP, VP, VP]
        L.glxo_sample.argtypes = [ctypes.POINTER(_CGraph), ctypes.c_int, VP, VP, i32, i32, ctypes.c_int, i64, u64,
                                  u64, VP, VP]
        L.glxo_aggregate.argtypes = [VP, i64, i32, VP, ctypes.c_int, VP, VP, i32, i32, ctypes.c_float, VP, VP]
        L.glxo_in_degree_weights.argtypes = [VP, i64, VP]
        L.glxo_sample_full.argtypes = [ctypes.POINTER(_CGraph), VP, i32, i32, VP, VP, VP, i64]
        L.glxo_sample_full.restype = i64
        L.glxo_partition.argtypes = [VP, i64, i32, VP, VP]
        L.glxo_dst_statics.argtypes = [VP, VP, i64, VP, VP]
        L.glxo_dst_statics.restype = i64
        L.glxo_negative_sample.argtypes = [VP, i64, VP, VP, ctypes.c_int, ctypes.POINTER(_CGraph), VP, i32, i32, i64, u64,
                                           u64, VP]
        L.glxo_aggregate_stitch.argtypes = [ctypes.c_int, i32, VP, VP, i32, i32, ctypes.c_float, ctypes.c_int, VP, VP]
        L.glxo_stitch_i64.argtypes = [VP, VP, i64, i32, VP]
        L.glxo_set_reference_cost_model.argtypes = [ctypes.c_int]
        L.glxo_random_walk.argtypes = [ctypes.POINTER(_CGraph), VP, i32, i32, ctypes.c_float, ctypes.c_float, i32,
                                       ctypes.c_float, i64, u64, u64, VP]
        L.glxo_node2vec_weights.argtypes = [ctypes.POINTER(_CGraph), i64, i64, ctypes.c_int, ctypes.c_float,
                                            ctypes.c_float, i32, ctypes.c_float, VP, VP]
        L.glxo_filter_act_on.argtypes = [ctypes.POINTER(_CFilter), i32, VP, VP, i32, VP]
        L.glxo_filter_act_on.restype = i32
        L.glxo_sample_filtered.argtypes = [ctypes.POINTER(_CGraph), ctypes.c_int, VP, VP, i32, i32, ctypes.c_int, i64,
                                           u64, u64, ctypes.POINTER(_CFilter), VP, VP]
        L.glxo_sample_full_filtered.argtypes = [ctypes.POINTER(_CGraph), VP, i32, i32, ctypes.c_int, i64,
                                                ctypes.POINTER(_CFilter), VP, VP, VP, i64]
        L.glxo_sample_full_filtered.restype = i64
        L.glxo_cond_negative_sample.argtypes = [VP, VP, i64, i32, VP, VP, ctypes.POINTER(_CGraph), VP, VP, VP, i32, i32,
                                                ctypes.c_int, ctypes.c_int, i32, i64, u64, u64, VP, VP]
        L.glxo_subgraph_induce.argtypes = [VP, i32, VP, VP, VP, VP, VP, VP, i64]
        L.glxo_subgraph_induce.restype = i64
        L.glxo_subgraph_dist.argtypes = [i32, VP, VP, i64, VP, VP]
        self.L = L

    def cond_negative_sample(self, ids, weights, cand_keys, props, g, src, dst, dst_keys, count, batch_share=False,
                             unique=False, retry=5, default_neighbor_id=0, seed=0, call_counter=0, with_filled=False):
        """ConditionalNegativeSampler under the contract.  cand_keys [ncols, U] int64, dst_keys [batch, ncols] int64
        (NO_KEY = matches nothing), props [ncols] float32 -> out [batch, count]."""
        ids = np.ascontiguousarray(ids, np.int64)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        ck = np.ascontiguousarray(cand_keys, np.int64).reshape(-1, ids.shape[0])
        ncols = ck.shape[0]
        pr = np.ascontiguousarray(props, np.float32)
        src = np.ascontiguousarray(src, np.int64)
        dst = np.ascontiguousarray(dst, np.int64)
        dk = np.ascontiguousarray(dst_keys, np.int64).reshape(src.shape[0], ncols)
        out = np.zeros((src.shape[0], count), np.int64)
        filled = np.zeros(src.shape[0], np.int32)
        cg = self._cgraph(g) if g is not None else None
        rc = self.L.glxo_cond_negative_sample(_p(ids), _p(w), ids.shape[0], ncols, _p(ck), _p(pr),
                                              ctypes.byref(cg) if cg is not None else None, _p(src), _p(dst), _p(dk),
                                              src.shape[0], count, int(batch_share), int(unique), retry, default_neighbor_id,
                                              seed, call_counter, _p(out), _p(filled))
        assert rc == 0, rc
        return (out, filled) if with_filled else out

    def subgraph_induce(self, nodes, offsets, nbr, eid):
        """InduceSubGraph on FullSampler's rows of `nodes` -> (row[m], col[m], eid[m])."""
        n = nodes.shape[0]
        m = self.L.glxo_subgraph_induce(_p(nodes), n, _p(offsets), _p(nbr), _p(eid), None, None, None, 0)
        row, col, e = np.zeros(m, np.int32), np.zeros(m, np.int32), np.zeros(m, np.int64)
        self.L.glxo_subgraph_induce(_p(nodes), n, _p(offsets), _p(nbr), _p(eid), _p(row), _p(col), _p(e), m)
        return row, col, e

    def subgraph(self, g, seeds, num_nbrs, full_nbr_num=100, need_dist=False):
        """SubGraphSampler::Process (subgraph_sampler.h:36-78) -> dict(nodes, row, col, eid[, dist_src, dist_dst])."""
        nodes = np.ascontiguousarray(seeds, np.int64)
        cur, found = nodes, set()
        for k in num_nbrs:
            if k > 0:
                _, nb, _ = self.sample_full(g, np.ascontiguousarray(cur), k)
                cur = nb
                found.update(int(x) for x in nb)
        nodes = np.concatenate([nodes, np.array(sorted(found), np.int64)]) if found else nodes
        deg, nb, ed = self.sample_full(g, nodes, full_nbr_num)
        off = np.zeros(nodes.shape[0] + 1, np.int64)
        off[1:] = np.cumsum(deg)
        row, col, e = self.subgraph_induce(nodes, off, nb, ed)
        out = dict(nodes=nodes, row=row, col=col, eid=e)
        if need_dist:
            ds, dd = np.zeros(nodes.shape[0], np.int32), np.zeros(nodes.shape[0], np.int32)
            self.L.glxo_subgraph_dist(nodes.shape[0], _p(row), _p(col), row.shape[0], _p(ds), _p(dd))
            out.update(dist_src=ds, dist_dst=dd)
        return out

    def philox(self, ctr, key):
        c = np.asarray(ctr, np.uint32)
        k = np.asarray(key, np.uint32)
        o = np.zeros(4, np.uint32)
        self.L.glxo_philox4x32_10(_p(c), _p(k), _p(o))
        return o

    def draw64(self, seed, cc, row, j):
        return self.L.glxo_draw64(seed, cc, row, j)

    def alias_build(self, row_ptr, weight):
        prob = np.zeros(weight.shape[0], np.float32)
        alias = np.zeros(weight.shape[0], np.int32)
        self.L.glxo_alias_build(_p(row_ptr), _p(weight), row_ptr.shape[0] - 1, _p(prob), _p(alias))
        return prob, alias

    def sort_rows(self, row_ptr, col, eid, weight):
        col, eid, weight = col.copy(), eid.copy(), weight.copy()
        self.L.glxo_sort_rows_by_weight_desc(_p(row_ptr), row_ptr.shape[0] - 1, _p(col), _p(eid), _p(weight))
        return col, eid, weight

    def set_reference_entropy(self, on, seed=0):
        """L1 entropy: glxo_sample draws like the reference (sequential MT19937 per sampler file + libstdc++'s
        distributions) from now on; process-wide, switch it off again."""
        self.L.glxo_set_reference_entropy.argtypes = [ctypes.c_int, ctypes.c_uint32]
        self.L.glxo_set_reference_entropy(1 if on else 0, seed)

    def sample(self, g, sampler, src, k, seed=0, call_counter=0, padding_mode=1, default_neighbor_id=0,
               rng_rows=None):
        """g: dict(row_ptr, col, eid, weight=None, alias=(prob, idx)|None, ids=None)."""
        if isinstance(sampler, str):
            sampler = ALL_SAMPLERS.index(sampler)
        cg = self._cgraph(g)
        batch = src.shape[0]
        nbr = np.zeros((batch, k), np.int64)
        eid = np.zeros((batch, k), np.int64)
        rc = self.L.glxo_sample(ctypes.byref(cg), sampler, _p(src), _p(rng_rows), batch, k, padding_mode, default_neighbor_id,
                                seed, call_counter, _p(nbr), _p(eid))
        assert rc == 0, rc
        return nbr, eid

    @staticmethod
    def _cfilter(g, flt):
        """flt: dict(type, field, values[batch], retry_times=5, default_timestamp=-1); the graph dict supplies
        ts_slot (per CSR slot) and indeg_weight."""
        keep = [np.ascontiguousarray(flt["values"], np.int64)]
        return _CFilter(flt["type"], flt["field"], _p(keep[0]), flt.get("retry_times", 5), _p(g.get("ts_slot")),
                        flt.get("default_timestamp", -1), _p(g.get("indeg_weight"))), keep

    def filter_act_on(self, flt, batch_idx, row_nbr, row_ts=None):
        cf, keep = self._cfilter({"ts_slot": row_ts}, flt)
        out = np.zeros(max(1, row_nbr.shape[0]), np.int32)
        n = self.L.glxo_filter_act_on(ctypes.byref(cf), batch_idx, _p(row_nbr), _p(row_ts), row_nbr.shape[0], _p(out))
        return out[:n].copy()

    def sample_filtered(self, g, sampler, src, k, flt, seed=0, call_counter=0, padding_mode=1, default_neighbor_id=0,
                        rng_rows=None):
        if isinstance(sampler, str):
            sampler = ALL_SAMPLERS.index(sampler)
        cg = self._cgraph(g)
        cf, keep = self._cfilter(g, flt)
        batch = src.shape[0]
        nbr = np.zeros((batch, k), np.int64)
        eid = np.zeros((batch, k), np.int64)
        rc = self.L.glxo_sample_filtered(ctypes.byref(cg), sampler, _p(src), _p(rng_rows), batch, k, padding_mode,
                                         default_neighbor_id, seed, call_counter, ctypes.byref(cf), _p(nbr), _p(eid))
        assert rc == 0, rc
        return nbr, eid

    def sample_full_filtered(self, g, src, max_limit, flt, padding_mode=1, default_neighbor_id=0):
        cg = self._cgraph(g)
        cf, keep = self._cfilter(g, flt)
        batch = src.shape[0]
        deg = np.zeros(batch, np.int32)
        total = self.L.glxo_sample_full_filtered(ctypes.byref(cg), _p(src), batch, max_limit, padding_mode,
                                                 default_neighbor_id, ctypes.byref(cf), _p(deg), None, None, 0)
        nbr = np.zeros(total, np.int64)
        eid = np.zeros(total, np.int64)
        self.L.glxo_sample_full_filtered(ctypes.byref(cg), _p(src), batch, max_limit, padding_mode, default_neighbor_id,
                                         ctypes.byref(cf), _p(deg), _p(nbr), _p(eid), total)
        return deg, nbr, eid

    def random_walk(self, g, seeds, walk_len, p=1.0, q=1.0, full_nbr_num=100, default_weight=0.0, default_neighbor_id=0,
                    seed=0, call_counter=0):
        cg = self._cgraph(g)
        walks = np.zeros((seeds.shape[0], walk_len), np.int64)
        rc = self.L.glxo_random_walk(ctypes.byref(cg), _p(seeds), seeds.shape[0], walk_len, p, q, full_nbr_num,
                                     default_weight, default_neighbor_id, seed, call_counter, _p(walks))
        assert rc == 0, rc
        return walks

    def node2vec_weights(self, g, cur, parent, has_parent_nbrs, p, q, full_nbr_num=100, default_weight=0.0):
        cg = self._cgraph(g)
        w = np.zeros(full_nbr_num, np.float32)
        n = ctypes.c_int32()
        self.L.glxo_node2vec_weights(ctypes.byref(cg), cur, parent, 1 if has_parent_nbrs else 0, p, q, full_nbr_num,
                                     default_weight, _p(w), ctypes.byref(n))
        return w[:n.value].copy()

    def _cgraph(self, g):
        alias = g.get("alias")
        ind = g.get("indeg_alias")
        return _CGraph(g["row_ptr"].shape[0] - 1, g["col"].shape[0], _p(g["row_ptr"]), _p(g["col"]), _p(g["eid"]),
                       _p(g.get("weight")), _p(alias[0]) if alias else None, _p(alias[1]) if alias else None,
                       _p(g.get("ids")), _p(ind[0]) if ind else None, _p(ind[1]) if ind else None)

    def in_degree_alias(self, g):
        """alias tables over float(in-degree of each slot's neighbour) (InDegreeSampler)."""
        w = np.zeros(g["col"].shape[0], np.float32)
        self.L.glxo_in_degree_weights(_p(g["col"]), g["col"].shape[0], _p(w))
        return self.alias_build(g["row_ptr"], w), w

    def sample_full(self, g, src, max_limit):
        cg = self._cgraph(g)
        batch = src.shape[0]
        deg = np.zeros(batch, np.int32)
        total = self.L.glxo_sample_full(ctypes.byref(cg), _p(src), batch, max_limit, _p(deg), None, None, 0)
        nbr = np.zeros(total, np.int64)
        eid = np.zeros(total, np.int64)
        self.L.glxo_sample_full(ctypes.byref(cg), _p(src), batch, max_limit, _p(deg), _p(nbr), _p(eid), total)
        return deg, nbr, eid

    def aggregate(self, X, op, node_ids, segment_ids, num_segments, default_attr=0.0, ids=None):
        if isinstance(op, str):
            op = AGGREGATORS.index(op)
        V, D = X.shape
        emb = np.zeros((num_segments, D), np.float32)
        cnt = np.zeros(num_segments, np.int32)
        rc = self.L.glxo_aggregate(_p(X), V, D, _p(ids), op, _p(node_ids), _p(segment_ids), node_ids.shape[0],
                                   num_segments, default_attr, _p(emb), _p(cnt))
        assert rc == 0, rc
        return emb, cnt

    def dst_statics(self, col, eid):
        """-> (distinct dst ids in first-appearance order, their in-degrees)"""
        E = col.shape[0]
        ids = np.zeros(max(E, 1), np.int64)
        deg = np.zeros(max(E, 1), np.int32)
        U = self.L.glxo_dst_statics(_p(col), _p(eid), E, _p(ids), _p(deg))
        return ids[:U].copy(), deg[:U].copy()

    def negative_sample(self, ids, table, exclude, g, src, count, default_neighbor_id=0, seed=0, call_counter=0):
        """table: (prob, alias) or None (uniform); exclude 0 none / 1 src's neighbours in g / 2 the batch."""
        cg = self._cgraph(g) if g is not None else None
        src = np.ascontiguousarray(src, np.int64)
        out = np.zeros((src.shape[0], count), np.int64)
        rc = self.L.glxo_negative_sample(_p(ids), ids.shape[0], _p(table[0]) if table else None,
                                         _p(table[1]) if table else None, exclude,
                                         ctypes.byref(cg) if cg is not None else None, _p(src), src.shape[0], count,
                                         default_neighbor_id, seed, call_counter, _p(out))
        assert rc == 0, rc
        return out

    def aggregate_stitch(self, op, parts, cnts, default_attr=0.0, reference_fold=False):
        """parts [P, Sg, D] f32, cnts [P, Sg] i32 -> (emb [Sg, D], cnt [Sg])."""
        if isinstance(op, str):
            op = AGGREGATORS.index(op)
        parts = np.ascontiguousarray(parts, np.float32)
        cnts = np.ascontiguousarray(cnts, np.int32)
        P, Sg, D = parts.shape
        emb = np.zeros((Sg, D), np.float32)
        cnt = np.zeros(Sg, np.int32)
        rc = self.L.glxo_aggregate_stitch(op, P, _p(parts), _p(cnts), Sg, D, default_attr, int(reference_fold),
                                          _p(emb), _p(cnt))
        assert rc == 0, rc
        return emb, cnt

    def sort_rows_by_timestamp(self, row_ptr, col, eid, ts_slot, weight=None):
        """-> (col, eid, ts_slot, weight) with every row in timestamp-ascending order (stable)"""
        col, eid, ts = col.copy(), eid.copy(), ts_slot.copy()
        w = None if weight is None else weight.copy()
        self.L.glxo_sort_rows_by_timestamp_asc(_p(row_ptr), row_ptr.shape[0] - 1, _p(col), _p(eid), _p(ts), _p(w))
        return col, eid, ts, w

    def partition(self, ids, P):
        order = np.zeros(ids.shape[0], np.int64)
        counts = np.zeros(P, np.int64)
        self.L.glxo_partition(_p(ids), ids.shape[0], P, _p(order), _p(counts))
        return order, counts

    def stitch(self, shard_major, order):
        n = order.shape[0]
        width = shard_major.size // max(n, 1)
        out = np.zeros_like(shard_major)
        self.L.glxo_stitch_i64(_p(shard_major), _p(order), n, width, _p(out))
        return out


def have_ref():
    return os.path.exists(REF_SO) and not os.environ.get("GLX_NO_REF")


class RefLib:
    """One reference GraphStore (process-global flags: use one instance at a time)."""

    def __init__(self, storage_mode=2, padding_mode=1, default_neighbor_id=0, default_float_attr=0.0):
        L = ctypes.CDLL(REF_SO)
        i32, i64 = ctypes.c_int32, ctypes.c_int64
        cs = ctypes.c_char_p
        L.glref_create.restype = VP
        L.glref_create.argtypes = [ctypes.c_int, ctypes.c_int, i64, ctypes.c_float]
        L.glref_destroy.argtypes = [VP]
        L.glref_set_flags.argtypes = [ctypes.c_int, i64, ctypes.c_float]
        L.glref_set_seed.argtypes = [ctypes.c_uint]
        L.glref_add_edges.argtypes = [VP, cs, VP, VP, VP, i64]
        L.glref_build_graph.argtypes = [VP, cs]
        L.glref_add_nodes.argtypes = [VP, cs, VP, VP, i64, i32]
        L.glref_build_nodes.argtypes = [VP, cs]
        L.glref_get_row.argtypes = [VP, cs, i64, VP, VP, i64]
        L.glref_get_row.restype = i64
        L.glref_edge_weight.argtypes = [VP, cs, i64]
        L.glref_edge_weight.restype = ctypes.c_float
        L.glref_sample.argtypes = [VP, cs, cs, VP, i32, i32, VP, VP, ctypes.c_int]
        L.glref_aggregate.argtypes = [VP, cs, cs, VP, VP, i32, i32, VP, VP, VP]
        L.glref_aggregate_stitch.argtypes = [cs, i32, VP, VP, i32, i32, VP, VP]
        L.glref_add_weighted_nodes.argtypes = [VP, cs, VP, VP, i64]
        L.glref_add_edges_ts.argtypes = [VP, cs, VP, VP, VP, VP, i64]
        L.glref_dst_statics.argtypes = [VP, cs, VP, VP, i64]
        L.glref_dst_statics.restype = i64
        L.glref_hash64.argtypes = [ctypes.c_char_p, i64]
        L.glref_hash64.restype = ctypes.c_uint64
        L.glref_parse_attribute.argtypes = [ctypes.c_char_p, i64, cs, VP, VP, i32, i32, VP, VP, VP, VP, ctypes.c_char_p,
                                            i64, VP]
        L.glref_add_attr_nodes.argtypes = [VP, cs, VP, VP, VP, i32, VP, i32, ctypes.c_char_p, VP, i32, i64]
        L.glref_cond_neg_sample.argtypes = [VP, cs, cs, cs, VP, VP, i32, i32, ctypes.c_int, ctypes.c_int, VP, VP, i32, VP, VP,
                                            i32, VP, VP, i32, i32, VP, i64, VP, ctypes.c_int]
        L.glref_subgraph.argtypes = [VP, cs, VP, i32, VP, i32, ctypes.c_int, i32, VP, i64, VP, VP, VP, i64, VP, VP, VP]
        L.glref_sample_full.argtypes = [VP, cs, VP, i32, i32, VP, VP, VP, i64]
        L.glref_sample_full.restype = i64
        L.glref_in_degree.argtypes = [VP, cs, i64]
        L.glref_in_degree.restype = i32
        L.glref_random_walk.argtypes = [VP, cs, VP, i32, i32, ctypes.c_float, ctypes.c_float, i32, VP, ctypes.c_int]
        L.glref_sample_filtered.argtypes = [VP, cs, cs, VP, i32, i32, ctypes.c_int, ctypes.c_int, VP, i32, i32, VP, VP,
                                            VP, i64, ctypes.c_int]
        L.glref_sample_filtered.restype = i64
        L.glref_alias_build.argtypes = [VP, i32, VP, VP]
        L.glref_time_sample_2hop.argtypes = [VP, cs, cs, VP, i32, i32, i32, i32, i32, VP]
        L.glref_time_sample_2hop.restype = ctypes.c_double
        L.glref_time_aggregate.argtypes = [VP, cs, cs, VP, i32, i32, i32, i32, VP]
        L.glref_time_aggregate.restype = ctypes.c_double
        self.L = L
        self.h = L.glref_create(storage_mode, padding_mode, default_neighbor_id, default_float_attr)

    def close(self):
        if self.h:
            self.L.glref_destroy(self.h)
            self.h = None

    def set_flags(self, padding_mode=1, default_neighbor_id=0, default_float_attr=0.0):
        self.L.glref_set_flags(padding_mode, default_neighbor_id, default_float_attr)

    def set_seed(self, seed):
        self.L.glref_set_seed(seed)

    def add_edges(self, etype, src, dst, weight=None):
        self.L.glref_add_edges(self.h, etype.encode(), _p(src), _p(dst), _p(weight), src.shape[0])
        self.L.glref_build_graph(self.h, etype.encode())

    def add_edges_timestamped(self, etype, src, dst, timestamps, weight=None):
        self.L.glref_add_edges_ts(self.h, etype.encode(), _p(src), _p(dst), _p(weight), _p(timestamps), src.shape[0])
        self.L.glref_build_graph(self.h, etype.encode())

    def add_nodes(self, ntype, ids, feats):
        self.L.glref_add_nodes(self.h, ntype.encode(), _p(ids), _p(feats), ids.shape[0], feats.shape[1])
        self.L.glref_build_nodes(self.h, ntype.encode())

    def export_csr(self, etype, row_ids, max_deg):
        """Post-Build adjacency of the given raw source ids -> (row_ptr, col, eid, weight-by-slot)."""
        rp = [0]
        cols, eids = [], []
        nb = np.zeros(max_deg, np.int64)
        ed = np.zeros(max_deg, np.int64)
        for v in row_ids:
            d = self.L.glref_get_row(self.h, etype.encode(), int(v), _p(nb), _p(ed), max_deg)
            assert d <= max_deg
            cols.append(nb[:d].copy())
            eids.append(ed[:d].copy())
            rp.append(rp[-1] + d)
        col = np.concatenate(cols) if cols else np.zeros(0, np.int64)
        eid = np.concatenate(eids) if eids else np.zeros(0, np.int64)
        w = np.array([self.L.glref_edge_weight(self.h, etype.encode(), int(e)) for e in eid], np.float32)
        return np.array(rp, np.int64), col, eid, w

    def sample(self, etype, strategy, src, k, fresh_thread=True):
        batch = src.shape[0]
        nbr = np.zeros((batch, k), np.int64)
        eid = np.zeros((batch, k), np.int64)
        rc = self.L.glref_sample(self.h, etype.encode(), strategy.encode(), _p(src), batch, k, _p(nbr), _p(eid),
                                 1 if fresh_thread else 0)
        assert rc == 0, rc
        return nbr, eid

    def sample_sequence(self, etype, strategies, src, k):
        """len(strategies) consecutive requests in ONE fresh thread (engines carry over) -> ([calls, batch, k]) x 2."""
        batch, calls = src.shape[0], len(strategies)
        nbr = np.zeros((calls, batch, k), np.int64)
        eid = np.zeros((calls, batch, k), np.int64)
        self.L.glref_sample_sequence.argtypes = [VP, ctypes.c_char_p, ctypes.c_char_p, VP, ctypes.c_int32, ctypes.c_int32,
                                                 ctypes.c_int32, VP, VP]
        rc = self.L.glref_sample_sequence(self.h, etype.encode(), ",".join(strategies).encode(), _p(src), batch, k, calls,
                                          _p(nbr), _p(eid))
        assert rc == 0, rc
        return nbr, eid

    def sample_full(self, etype, src, max_limit, cap=1 << 22):
        deg = np.zeros(src.shape[0], np.int32)
        nbr = np.zeros(cap, np.int64)
        eid = np.zeros(cap, np.int64)
        total = self.L.glref_sample_full(self.h, etype.encode(), _p(src), src.shape[0], max_limit, _p(deg), _p(nbr),
                                         _p(eid), cap)
        assert 0 <= total <= cap, total
        return deg, nbr[:total].copy(), eid[:total].copy()

    def add_attr_nodes(self, ntype, ids, weights=None, int_attrs=None, float_attrs=None, str_attrs=None):
        """int_attrs [n, I] int64, float_attrs [n, F] float32, str_attrs: list of n lists of S byte strings."""
        ids = np.ascontiguousarray(ids, np.int64)
        n = ids.shape[0]
        ia = None if int_attrs is None else np.ascontiguousarray(int_attrs, np.int64).reshape(n, -1)
        fa = None if float_attrs is None else np.ascontiguousarray(float_attrs, np.float32).reshape(n, -1)
        S = len(str_attrs[0]) if str_attrs else 0
        blob = b"".join(x for row in (str_attrs or []) for x in row)
        lens = np.array([len(x) for row in (str_attrs or []) for x in row], np.int32)
        w = None if weights is None else np.ascontiguousarray(weights, np.float32)
        self.L.glref_add_attr_nodes(self.h, ntype.encode(), _p(ids), _p(w), _p(ia), 0 if ia is None else ia.shape[1], _p(fa),
                                    0 if fa is None else fa.shape[1], blob, _p(lens), S, n)
        self.L.glref_build_nodes(self.h, ntype.encode())

    def cond_neg_sample(self, etype, strategy, dst_node_type, src, dst, count, int_cols=(), int_props=(), float_cols=(),
                        float_props=(), str_cols=(), str_props=(), batch_share=False, unique=False, retry=5,
                        fresh_thread=True):
        """The reference's ConditionalNegativeSampler -> flat array of the ids its response holds (may be shorter
        than batch * count: its fill loop never runs)."""
        src = np.ascontiguousarray(src, np.int64)
        dst = np.ascontiguousarray(dst, np.int64)
        cap = src.shape[0] * count + 16
        out = np.zeros(cap, np.int64)
        n = ctypes.c_int64()
        a = lambda x, t: np.ascontiguousarray(x, t)  # noqa: E731
        ic, ip, fc, fp, sc, sp = (a(int_cols, np.int32), a(int_props, np.float32), a(float_cols, np.int32),
                                  a(float_props, np.float32), a(str_cols, np.int32), a(str_props, np.float32))
        rc = self.L.glref_cond_neg_sample(self.h, etype.encode(), strategy.encode(), dst_node_type.encode(), _p(src), _p(dst),
                                          src.shape[0], count, int(batch_share), int(unique), _p(ic), _p(ip), ic.shape[0],
                                          _p(fc), _p(fp), fc.shape[0], _p(sc), _p(sp), sc.shape[0], retry, _p(out), cap,
                                          ctypes.byref(n), 1 if fresh_thread else 0)
        assert rc == 0, rc
        return out[:n.value].copy()

    def subgraph(self, nbr_type, seeds, num_nbrs, full_nbr_num=100, need_dist=False, cap_nodes=1 << 14, cap_edges=1 << 20):
        """The reference's SubGraphSampler operator -> dict(nodes, row, col, eid[, dist_src, dist_dst])."""
        seeds = np.ascontiguousarray(seeds, np.int64)
        nn = np.ascontiguousarray(num_nbrs, np.int32)
        nodes = np.zeros(cap_nodes, np.int64)
        row, col, eid = np.zeros(cap_edges, np.int32), np.zeros(cap_edges, np.int32), np.zeros(cap_edges, np.int64)
        ds, dd = np.zeros(cap_nodes, np.int32), np.zeros(cap_nodes, np.int32)
        sizes = np.zeros(2, np.int64)
        rc = self.L.glref_subgraph(self.h, nbr_type.encode(), _p(seeds), seeds.shape[0], _p(nn), nn.shape[0],
                                   1 if need_dist else 0, full_nbr_num, _p(nodes), cap_nodes, _p(row), _p(col), _p(eid),
                                   cap_edges, _p(ds), _p(dd), _p(sizes))
        assert rc == 0, rc
        n, m = int(sizes[0]), int(sizes[1])
        assert n <= cap_nodes and m <= cap_edges
        out = dict(nodes=nodes[:n].copy(), row=row[:m].copy(), col=col[:m].copy(), eid=eid[:m].copy())
        if need_dist:
            out.update(dist_src=ds[:n].copy(), dist_dst=dd[:n].copy())
        return out

    def sample_filtered(self, etype, strategy, src, k, flt, fresh_thread=True):
        """flt: dict(type, field, values, retry_times=5); values may be shorter than the batch
        (Filter::FillValues repeats each value batch / len(values) times)."""
        values = np.ascontiguousarray(flt["values"], np.int64)
        batch = src.shape[0]
        if strategy == "FullSampler":
            cap = 1 << 22
            deg = np.zeros(batch, np.int32)
            nbr, eid = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
            total = self.L.glref_sample_filtered(self.h, etype.encode(), strategy.encode(), _p(src), batch, k,
                                                 flt["type"], flt["field"], _p(values), values.shape[0],
                                                 flt.get("retry_times", 5), _p(deg), _p(nbr), _p(eid), cap, 0)
            assert 0 <= total <= cap, total
            return deg, nbr[:total].copy(), eid[:total].copy()
        nbr = np.zeros((batch, k), np.int64)
        eid = np.zeros((batch, k), np.int64)
        total = self.L.glref_sample_filtered(self.h, etype.encode(), strategy.encode(), _p(src), batch, k, flt["type"],
                                             flt["field"], _p(values), values.shape[0], flt.get("retry_times", 5),
                                             None, _p(nbr), _p(eid), batch * k, 1 if fresh_thread else 0)
        assert total == batch * k, total
        return nbr, eid

    def random_walk(self, etype, src, walk_len, p, q, full_nbr_num=100, fresh_thread=True):
        walks = np.zeros((src.shape[0], walk_len), np.int64)
        rc = self.L.glref_random_walk(self.h, etype.encode(), _p(src), src.shape[0], walk_len, p, q, full_nbr_num,
                                      _p(walks), 1 if fresh_thread else 0)
        assert rc == 0, rc
        return walks

    def in_degree(self, etype, ids):
        return np.array([self.L.glref_in_degree(self.h, etype.encode(), int(v)) for v in ids], np.int32)

    def alias_build(self, w):
        p = np.zeros(w.shape[0], np.float32)
        a = np.zeros(w.shape[0], np.int32)
        self.L.glref_alias_build(_p(w), w.shape[0], _p(p), _p(a))
        return p, a

    def add_weighted_nodes(self, ntype, ids, weights):
        ids = np.ascontiguousarray(ids, np.int64)
        weights = np.ascontiguousarray(weights, np.float32)
        assert self.L.glref_add_weighted_nodes(self.h, ntype.encode(), _p(ids), _p(weights), ids.shape[0]) == 0
        assert self.L.glref_build_nodes(self.h, ntype.encode()) == 0

    def dst_statics(self, etype, cap=1 << 22):
        ids = np.zeros(cap, np.int64)
        deg = np.zeros(cap, np.int32)
        n = self.L.glref_dst_statics(self.h, etype.encode(), _p(ids), _p(deg), cap)
        assert 0 <= n <= cap
        return ids[:n].copy(), deg[:n].copy()

    def negative_sample(self, type_name, strategy, src, k, fresh_thread=True):
        src = np.ascontiguousarray(src, np.int64)
        out = np.zeros((src.shape[0], k), np.int64)
        rc = self.L.glref_sample(self.h, type_name.encode(), strategy.encode(), _p(src), src.shape[0], k, _p(out), None,
                                 1 if fresh_thread else 0)
        assert rc == 0, rc
        return out

    def hash64(self, data):
        return int(self.L.glref_hash64(data, len(data)))

    def parse_attribute(self, data, delimiter, types, hash_buckets=None):
        """types: io::DataType values (0 int32, 1 int64, 2 float, 3 double, 4 string).
        -> (status code, ints, floats, [bytes])"""
        t = np.asarray(types, np.int32)
        hb = np.asarray(hash_buckets if hash_buckets is not None else [0] * len(types), np.int64)
        ints = np.zeros(max(len(types), 1) + 8, np.int64)
        floats = np.zeros(max(len(types), 1) + 8, np.float32)
        buf = ctypes.create_string_buffer(len(data) + 64)
        ni, nf, ns = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        rc = self.L.glref_parse_attribute(data, len(data), delimiter.encode(), _p(t), _p(hb), len(types),
                                          1 if hash_buckets is not None else 0, _p(ints), ctypes.byref(ni),
                                          _p(floats), ctypes.byref(nf), buf, len(buf), ctypes.byref(ns))
        strings = buf.value.split(b"\n")[:ns.value] if ns.value else []
        return rc, ints[:ni.value].copy(), floats[:nf.value].copy(), strings

    def aggregate_stitch(self, strategy, parts, cnts):
        parts = np.ascontiguousarray(parts, np.float32)
        cnts = np.ascontiguousarray(cnts, np.int32)
        P, Sg, D = parts.shape
        emb = np.zeros((Sg, D), np.float32)
        cnt = np.zeros(Sg, np.int32)
        rc = self.L.glref_aggregate_stitch(strategy.encode(), P, _p(parts), _p(cnts), Sg, D, _p(emb), _p(cnt))
        assert rc == 0, rc
        return emb, cnt

    def aggregate(self, ntype, strategy, node_ids, segment_ids, num_segments, dim):
        emb = np.zeros((num_segments, dim), np.float32)
        cnt = np.zeros(num_segments, np.int32)
        d = ctypes.c_int32()
        rc = self.L.glref_aggregate(self.h, ntype.encode(), strategy.encode(), _p(node_ids), _p(segment_ids),
                                    node_ids.shape[0], num_segments, _p(emb), _p(cnt), ctypes.byref(d))
        assert rc == 0 and d.value == dim, (rc, d.value)
        return emb, cnt

"""ctypes harness over the glx C-ABI (include/glx.h) for tests and bench.py.

This is plumbing, not the product: the product is libglx.so (HIP kernels behind
the C-ABI) and libglx_host.so (the C++ mirror of graphlearn::op).  numpy arrays
are passed as host pointers (GLX_PTR_HOST), torch CUDA tensors as device
pointers (GLX_PTR_DEVICE) on the current torch stream.  There is no CPU
fallback anywhere: a missing library or a missing GPU raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GLX_LIB") or os.path.join(_HERE, "lib", "libglx.so")  # GLX_LIB: A/B builds

PTR_HOST, PTR_DEVICE = 0, 1
RANDOM, RANDOM_WITHOUT_REPLACEMENT, EDGE_WEIGHT, TOPK, IN_DEGREE = 0, 1, 2, 3, 4
SUM, MEAN, MAX, MIN, PROD = 0, 1, 2, 3, 4
PAD_REPLICATE, PAD_CIRCULAR = 0, 1

# registry names of the reference (REGISTER_OPERATOR(...)) -> ids of the C-ABI
SAMPLER_IDS = {
    "RandomSampler": RANDOM,
    "RandomWithoutReplacementSampler": RANDOM_WITHOUT_REPLACEMENT,
    "EdgeWeightSampler": EDGE_WEIGHT,
    "TopkSampler": TOPK,
}
# further registry names served on the device
EXTRA_SAMPLER_IDS = {"InDegreeSampler": IN_DEGREE}
AGGREGATOR_IDS = {
    "SumAggregator": SUM,
    "MeanAggregator": MEAN,
    "MaxAggregator": MAX,
    "MinAggregator": MIN,
    "ProdAggregator": PROD,
}

EXPORTS = [
    "glx_abi_version", "glx_device_count", "glx_last_error", "glx_host_register", "glx_host_unregister",
    "glx_graph_create", "glx_graph_build", "glx_graph_build_ordered", "glx_graph_destroy", "glx_graph_info", "glx_graph_export_alias",
    "glx_graph_degrees", "glx_graph_in_degrees", "glx_sample", "glx_sample_ex", "glx_sample_hops",
    "glx_graph_enable_in_degree", "glx_graph_enable_default_weight", "glx_sample_full_sizes", "glx_sample_full",
    "glx_graph_set_timestamps", "glx_sample_filtered", "glx_sample_full_filtered", "glx_random_walk",
    "glx_features_create", "glx_features_view", "glx_features_destroy", "glx_features_info",
    "glx_aggregate", "glx_lookup",
    "glx_partition", "glx_stitch_i64", "glx_stitch_f32", "glx_aggregate_stitch",
    "glx_negative_create", "glx_negative_from_graph", "glx_negative_destroy", "glx_negative_info",
    "glx_negative_export", "glx_graph_enable_negative", "glx_negative_sample",
    "glx_profile_enable", "glx_profile_collect",
    "glx_comm_unique_id", "glx_comm_init_rccl", "glx_comm_init_local", "glx_comm_init_callbacks", "glx_comm_destroy",
    "glx_comm_info", "glx_comm_set_max_message_bytes", "glx_exchange_v", "glx_comm_allgather_i64", "glx_comm_barrier",
    "glx_dist_store_create", "glx_dist_store_destroy", "glx_dist_store_set_cache", "glx_dist_store_set_graph_replica",
    "glx_dist_build_graph_replica", "glx_dist_sample_full_sizes", "glx_dist_sample_full", "glx_dist_sample_full_filtered",
    "glx_dist_in_degrees", "glx_dist_negative_create", "glx_dist_negative_sample", "glx_dist_random_walk", "glx_dist_random_walk_ex",
    "glx_dist_last_sample_rows", "glx_dist_hot_ids",
    "glx_dist_enable_in_degree",
    "glx_dist_sample", "glx_dist_aggregate", "glx_dist_aggregate_partial", "glx_dist_aggregate_begin", "glx_dist_aggregate_end", "glx_dist_aggregate_end_range", "glx_dist_lookup",
    "glx_dist_last_stats",
    "glx_dist_ledger_create", "glx_dist_ledger_destroy", "glx_dist_store_set_ledger", "glx_dist_confirm",
    "glx_dist_ledger_get_stats", "glx_dist_ledger_set_slack",
    "glx_plan_create", "glx_plan_run", "glx_plan_output", "glx_plan_destroy",
    "glx_probe_bandwidth", "glx_tune", "glx_subgraph_induce",
    "glx_cond_table_create", "glx_cond_table_destroy", "glx_cond_negative_sample",
]


FILTER_NONE, FILTER_EQUAL, FILTER_LARGER_THAN = 0, 1, 2
FILTER_FIELD_NONE, FILTER_FIELD_ID, FILTER_FIELD_TIMESTAMP = 0, 1, 2


class Filter(ctypes.Structure):
    """glx_filter (include/glx.h): type, field, values[batch], retry_times, default_timestamp."""
    _fields_ = [("type", ctypes.c_int32), ("field", ctypes.c_int32), ("values", ctypes.c_void_p),
                ("retry_times", ctypes.c_int32), ("default_timestamp", ctypes.c_int64)]


class DistStats(ctypes.Structure):
    """glx_dist_stats (include/glx.h): where the ids of a store's last aggregate / lookup came from."""
    _fields_ = [(n, ctypes.c_int64) for n in ("ids", "from_replica", "from_own_shard", "remote", "remote_distinct",
                                                "served_rows", "bytes_sent", "bytes_received", "exchange_rounds",
                                                "host_syncs", "host_stall_us")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# int (*)(void* user, const void* send, const int64_t* send_counts, void* recv, const int64_t* recv_counts, int64_t eb)
HOST_ALLTOALLV_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int64)
# int (*)(void* user, const void* send, void* recv, int64_t bytes_per_rank)
HOST_ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64)


ABORTED = 10  # GLX_ABORTED: a speculated exchange did not fit (Ledger); repeat the unconfirmed calls


class GlxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("glx error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Loads libglx.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GlxError(14, "libglx.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
                               " (expected at %s)" % LIB_PATH)
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME
        # libamdhip64.so.7).  It must be in the process BEFORE libglx.so so that
        # libglx's NEEDED libamdhip64.so.7 binds to the same runtime; two HIP
        # runtimes in one process cannot share device pointers.
        if not os.environ.get("GLX_NO_TORCH"):  # (debug knob: run on the system HIP runtime)
            import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, i64, u64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float
        ci = ctypes.c_int
        L.glx_last_error.restype = ctypes.c_char_p
        L.glx_device_count.argtypes = [ctypes.POINTER(ci)]
        L.glx_host_register.argtypes = [vp, u64]
        L.glx_host_unregister.argtypes = [vp]
        L.glx_graph_create.argtypes = [ci, i64, i64, vp, vp, vp, vp, vp, ci, vp, ctypes.POINTER(vp)]
        L.glx_graph_build.argtypes = [ci, i64, vp, vp, vp, vp, ci, ci, vp, ctypes.POINTER(vp)]
        L.glx_graph_destroy.argtypes = [vp]
        L.glx_graph_destroy.restype = None
        L.glx_graph_info.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i64), ctypes.POINTER(ci),
                                     ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.glx_graph_export_alias.argtypes = [vp, vp, vp, ci, vp]
        L.glx_graph_build_ordered.argtypes = [ci, i64, vp, vp, vp, vp, vp, ci, ci, vp, ctypes.POINTER(vp)]
        L.glx_graph_degrees.argtypes = [vp, vp, i64, vp, ci, vp]
        L.glx_graph_in_degrees.argtypes = [vp, vp, i64, vp, ci, vp]
        L.glx_sample.argtypes = [vp, ci, vp, i32, i32, ci, i64, u64, u64, vp, vp, ci, vp]
        L.glx_sample_ex.argtypes = [vp, ci, vp, vp, i32, i32, ci, i64, u64, u64, vp, vp, ci, vp]
        L.glx_sample_hops.argtypes = [vp, i32, ci, vp, i32, vp, ci, i64, u64, u64, vp, vp, ci, vp]
        L.glx_graph_enable_in_degree.argtypes = [vp, vp]
        L.glx_graph_enable_default_weight.argtypes = [vp, ctypes.c_float, vp]
        L.glx_sample_full_sizes.argtypes = [vp, vp, i32, i32, vp, vp, ci, vp]
        L.glx_sample_full.argtypes = [vp, vp, i32, i32, vp, vp, vp, ci, vp]
        L.glx_graph_set_timestamps.argtypes = [vp, vp, ci, vp]
        L.glx_random_walk.argtypes = [vp, vp, i32, i32, f32, f32, i32, f32, i64, u64, u64, vp, ci, vp]
        L.glx_sample_filtered.argtypes = [vp, ci, vp, vp, i32, i32, ci, i64, u64, u64, ctypes.POINTER(Filter), vp, vp, ci,
                                          vp]
        L.glx_sample_full_filtered.argtypes = [vp, vp, i32, i32, vp, ci, i64, ctypes.POINTER(Filter), vp, vp, ci, vp]
        L.glx_features_view.argtypes = [ci, i64, i32, vp, ctypes.POINTER(vp)]
        L.glx_features_create.argtypes = [ci, i64, i32, vp, vp, ci, vp, ctypes.POINTER(vp)]
        L.glx_features_destroy.argtypes = [vp]
        L.glx_features_destroy.restype = None
        L.glx_features_info.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i32), ctypes.POINTER(ci),
                                        ctypes.POINTER(ci)]
        L.glx_aggregate.argtypes = [vp, ci, vp, vp, i32, i32, f32, vp, vp, ci, vp]
        L.glx_lookup.argtypes = [vp, vp, i64, f32, vp, ci, vp]
        L.glx_partition.argtypes = [ci, vp, i64, i32, vp, vp, vp, vp]
        L.glx_stitch_i64.argtypes = [ci, vp, vp, i64, i32, vp, vp]
        L.glx_stitch_f32.argtypes = [ci, vp, vp, i64, i32, vp, vp]
        L.glx_aggregate_stitch.argtypes = [ci, ci, i32, vp, vp, i32, i32, f32, vp, vp, vp]
        L.glx_negative_create.argtypes = [ci, i64, vp, vp, ci, vp, ctypes.POINTER(vp)]
        L.glx_negative_from_graph.argtypes = [vp, ci, vp, ctypes.POINTER(vp)]
        L.glx_negative_destroy.argtypes = [vp]
        L.glx_negative_destroy.restype = None
        L.glx_negative_info.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(ci)]
        L.glx_negative_export.argtypes = [vp, vp, vp, vp, vp]
        L.glx_graph_enable_negative.argtypes = [vp, vp]
        L.glx_negative_sample.argtypes = [vp, ci, vp, vp, i32, i32, i64, u64, u64, vp, ci, vp]
        L.glx_profile_enable.argtypes = [ci]
        L.glx_profile_collect.argtypes = [ci, vp, i32, ctypes.POINTER(i32)]
        L.glx_comm_unique_id.argtypes = [vp]
        L.glx_comm_init_rccl.argtypes = [ci, ci, ci, vp, ctypes.POINTER(vp)]
        L.glx_comm_init_local.argtypes = [i64, ci, ci, ci, ctypes.POINTER(vp)]
        L.glx_comm_init_callbacks.argtypes = [ci, ci, ci, HOST_ALLTOALLV_FN, HOST_ALLGATHER_FN, vp, ctypes.POINTER(vp)]
        L.glx_comm_destroy.argtypes = [vp]
        L.glx_comm_destroy.restype = None
        L.glx_comm_info.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.glx_comm_set_max_message_bytes.argtypes = [vp, i64]
        L.glx_comm_set_max_message_bytes.restype = i64
        L.glx_exchange_v.argtypes = [vp, vp, vp, vp, vp, i64, ci, vp]
        L.glx_comm_allgather_i64.argtypes = [vp, vp, i32, vp, ci, vp]
        L.glx_comm_barrier.argtypes = [vp, vp]
        L.glx_dist_store_create.argtypes = [vp, vp, vp, ctypes.POINTER(vp)]
        L.glx_dist_store_destroy.argtypes = [vp]
        L.glx_dist_store_destroy.restype = None
        L.glx_dist_store_set_cache.argtypes = [vp, vp, i64, f32, ci, vp]
        L.glx_dist_store_set_graph_replica.argtypes = [vp, vp]
        L.glx_dist_build_graph_replica.argtypes = [vp, vp, i64, ci, vp, vp]
        L.glx_dist_random_walk.argtypes = [vp, vp, i32, i32, f32, f32, i64, u64, u64, vp, ci, vp]
        L.glx_dist_random_walk_ex.argtypes = [vp, vp, i32, i32, f32, f32, i32, f32, i64, u64, u64, vp, ci, vp]
        L.glx_dist_sample_full_sizes.argtypes = [vp, vp, i32, i32, vp, vp, ci, vp]
        L.glx_dist_sample_full.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i64, ci, vp]
        L.glx_dist_sample_full_filtered.argtypes = [vp, vp, i32, i32, vp, vp, ci, i64, vp, vp, vp, i64, ci, vp]
        L.glx_dist_in_degrees.argtypes = [vp, vp, i32, vp, ci, vp]
        L.glx_dist_negative_create.argtypes = [vp, ci, vp, ctypes.POINTER(vp)]
        L.glx_dist_negative_sample.argtypes = [vp, vp, ci, vp, i32, i32, i64, u64, u64, vp, ci, vp]
        L.glx_dist_last_sample_rows.argtypes = [vp, vp, vp, vp]
        L.glx_dist_hot_ids.argtypes = [vp, i64, vp, ctypes.POINTER(i64), vp]
        L.glx_dist_enable_in_degree.argtypes = [vp, vp, vp]
        L.glx_dist_sample.argtypes = [vp, ci, vp, i32, i32, ci, i64, u64, u64, ctypes.POINTER(Filter), vp, vp, ci, vp]
        L.glx_dist_aggregate.argtypes = [vp, ci, vp, vp, i32, i32, f32, vp, vp, ci, vp]
        L.glx_dist_aggregate_partial.argtypes = [vp, ci, vp, vp, i32, i32, f32, vp, vp, ci, vp]
        L.glx_dist_lookup.argtypes = [vp, vp, i64, f32, vp, ci, vp]
        L.glx_dist_aggregate_begin.argtypes = [vp, i32, vp, i32, f32, vp]
        L.glx_dist_aggregate_end.argtypes = [vp, i32, ci, vp, i32, vp, vp, vp]
        L.glx_dist_aggregate_end_range.argtypes = [vp, i32, i32, i32, ci, ci, vp, i32, vp, vp, vp]
        L.glx_dist_last_stats.argtypes = [vp, ctypes.POINTER(DistStats)]
        L.glx_dist_ledger_create.argtypes = [ci, ctypes.POINTER(vp)]
        L.glx_dist_ledger_destroy.argtypes = [vp]
        L.glx_dist_ledger_destroy.restype = None
        L.glx_dist_store_set_ledger.argtypes = [vp, vp]
        L.glx_dist_confirm.argtypes = [vp, vp]
        L.glx_dist_ledger_get_stats.argtypes = [vp, ctypes.POINTER(LedgerStats)]
        L.glx_dist_ledger_set_slack.argtypes = [vp, ctypes.c_double, i64]
        L.glx_plan_create.argtypes = [vp, i32, ci, vp, i32, ci, i64, u64, vp, ci, f32, ctypes.POINTER(vp)]
        L.glx_plan_run.argtypes = [vp, vp, u64, vp]
        L.glx_plan_output.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp),
                                      ctypes.POINTER(vp), ctypes.POINTER(i64), ctypes.POINTER(i32)]
        L.glx_plan_destroy.argtypes = [vp]
        L.glx_cond_table_create.argtypes = [ci, i64, vp, vp, i32, vp, ci, vp, ctypes.POINTER(vp)]
        L.glx_cond_table_destroy.argtypes = [vp]
        L.glx_cond_table_destroy.restype = None
        L.glx_cond_negative_sample.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, ci, ci, i32, i64, u64, u64, vp, ci, vp]
        L.glx_subgraph_induce.argtypes = [ci, vp, i32, vp, vp, vp, vp, vp, vp, i64, ctypes.POINTER(i64), ci, vp]
        L.glx_probe_bandwidth.argtypes = [ci, ci, i64, i64, i32, i32, ctypes.POINTER(ctypes.c_double),
                                          ctypes.POINTER(ctypes.c_double), vp]
        L.glx_tune.argtypes = [ctypes.c_char_p, i32]
        L.glx_plan_destroy.restype = None
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise GlxError(rc, lib().glx_last_error().decode())


def device_count():
    n = ctypes.c_int(0)
    _check(lib().glx_device_count(ctypes.byref(n)))
    return n.value


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _ptr(x, dtype=None):
    """(pointer, kind) of a numpy array or a torch CUDA tensor; None -> NULL."""
    if x is None:
        return None, None
    if _is_torch(x):
        assert x.is_cuda and x.is_contiguous(), "torch inputs must be contiguous CUDA tensors"
        return ctypes.c_void_p(x.data_ptr()), PTR_DEVICE
    assert isinstance(x, np.ndarray) and x.flags["C_CONTIGUOUS"]
    if dtype is not None:
        assert x.dtype == dtype, (x.dtype, dtype)
    return ctypes.c_void_p(x.ctypes.data), PTR_HOST


def _kind(*xs):
    kinds = set(k for _, k in xs if k is not None)
    assert len(kinds) == 1, "all data pointers of a call must be host (numpy) or device (torch)"
    return kinds.pop()


def _stream(kind, device=None):
    """torch's current stream ON THE DEVICE THE HANDLE LIVES ON (not on torch's current device: a process
    that drives several GPUs, or set_device_id() without torch.cuda.set_device(), would otherwise pass a
    stream of another device)."""
    if kind == PTR_DEVICE:
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return None


class Graph:
    """Device-resident CSR of one edge type (glx_graph)."""

    def __init__(self, row_ptr, col, eid, weight=None, ids=None, device=0):
        ptrs = [_ptr(row_ptr), _ptr(col), _ptr(eid), _ptr(weight), _ptr(ids)]
        kind = _kind(*ptrs)
        self.num_rows = int(row_ptr.shape[0]) - 1
        self.num_edges = int(col.shape[0])
        self.device = device
        h = ctypes.c_void_p()
        _check(lib().glx_graph_create(device, self.num_rows, self.num_edges, ptrs[0][0], ptrs[1][0],
                                      ptrs[2][0], ptrs[3][0], ptrs[4][0], kind, _stream(kind, self.device),
                                      ctypes.byref(h)))
        self._h = h

    @classmethod
    def from_edges(cls, src, dst, weight=None, edge_ids=None, sort_by_weight=True, device=0, timestamp=None):
        """Device-side build from a raw edge list in insertion order (edge id = index,
        or edge_ids[i] for a shard's subset of a global edge list).  Rows are ordered by
        timestamp ascending when timestamps are given (timestamped edge types), else by
        weight descending (sort_by_weight), else left in insertion order."""
        self = cls.__new__(cls)
        ptrs = [_ptr(src), _ptr(dst), _ptr(weight), _ptr(edge_ids), _ptr(timestamp)]
        kind = _kind(*ptrs)
        h = ctypes.c_void_p()
        order = 2 if timestamp is not None else (1 if sort_by_weight else 0)
        _check(lib().glx_graph_build_ordered(device, int(src.shape[0]), ptrs[0][0], ptrs[1][0], ptrs[2][0],
                                             ptrs[3][0], ptrs[4][0], order, kind, _stream(kind, device), ctypes.byref(h)))
        self._h = h
        self.device = device
        v, e = ctypes.c_int64(), ctypes.c_int64()
        _check(lib().glx_graph_info(h, ctypes.byref(v), ctypes.byref(e), None, None, None))
        self.num_rows, self.num_edges = v.value, e.value
        return self

    @classmethod
    def from_handle(cls, handle, device=0):
        """Borrow a glx_graph* owned by someone else (e.g. the C++ host layer's GraphStore)."""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p(int(handle))
        self._borrowed = True
        v, e, d = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        _check(lib().glx_graph_info(self._h, ctypes.byref(v), ctypes.byref(e), None, None, ctypes.byref(d)))
        self.num_rows, self.num_edges = v.value, e.value
        self.device = d.value  # the device the handle lives on, whatever the caller passed
        return self

    def close(self):
        if getattr(self, "_borrowed", False):
            self._h = None
        if getattr(self, "_h", None):
            try:
                lib().glx_graph_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    __del__ = close

    def enable_negative(self):
        """Sort every row's neighbour ids (the exclusion test of strict negative sampling)."""
        _check(lib().glx_graph_enable_negative(self._h, None))

    enable_id_index = enable_negative  # the same per-row id-sorted index serves id == value filters

    def enable_in_degree(self):
        """Build the in-degree alias tables InDegreeSampler needs (once, on the device)."""
        _check(lib().glx_graph_enable_in_degree(self._h, None))
        return self

    def enable_default_weight(self, default_weight=0.0):
        """EdgeWeightSampler on an unweighted graph: every slot weighs `default_weight` (the reference's
        GLOBAL_FLAG(DefaultWeight)); builds the alias tables of that.  No effect on a weighted graph."""
        _check(lib().glx_graph_enable_default_weight(self._h, ctypes.c_float(default_weight), None))
        return self

    def sample_full(self, src, max_limit=0):
        """FullSampler: -> (degrees[batch] int32, nbr[total], eid[total]) (sparse response)."""
        batch = int(src.shape[0])
        if _is_torch(src):
            import torch
            deg = torch.empty(batch, dtype=torch.int32, device=src.device)
            off = torch.empty(batch + 1, dtype=torch.int64, device=src.device)
            _check(lib().glx_sample_full_sizes(self._h, _ptr(src)[0], batch, max_limit, _ptr(deg)[0], _ptr(off)[0],
                                               PTR_DEVICE, _stream(PTR_DEVICE, self.device)))
            total = int(off[-1].item())
            nbr = torch.empty(total, dtype=torch.int64, device=src.device)
            eid = torch.empty(total, dtype=torch.int64, device=src.device)
            _check(lib().glx_sample_full(self._h, _ptr(src)[0], batch, max_limit, _ptr(off)[0], _ptr(nbr)[0],
                                         _ptr(eid)[0], PTR_DEVICE, _stream(PTR_DEVICE, self.device)))
            return deg, nbr, eid
        deg = np.empty(batch, np.int32)
        off = np.empty(batch + 1, np.int64)
        _check(lib().glx_sample_full_sizes(self._h, _ptr(src)[0], batch, max_limit, _ptr(deg)[0], _ptr(off)[0],
                                           PTR_HOST, None))
        total = int(off[-1])
        nbr = np.empty(total, np.int64)
        eid = np.empty(total, np.int64)
        _check(lib().glx_sample_full(self._h, _ptr(src)[0], batch, max_limit, _ptr(off)[0], _ptr(nbr)[0],
                                     _ptr(eid)[0], PTR_HOST, None))
        return deg, nbr, eid

    def set_timestamps(self, ts_slot):
        """Per-slot edge timestamps (CSR order) for a handle made from a CSR; mutates the handle."""
        p, kind = _ptr(ts_slot)
        _check(lib().glx_graph_set_timestamps(self._h, p, kind, _stream(kind, self.device)))

    def sample_filtered(self, sampler, src, k, filter_type, filter_field, values, seed=0, call_counter=0,
                        padding_mode=PAD_CIRCULAR, default_neighbor_id=0, retry_times=5, default_timestamp=-1,
                        rng_rows=None):
        """Sampling with a Filter: values[batch] is the expanded filter tensor (same kind as src)."""
        if isinstance(sampler, str):
            sampler = SAMPLER_IDS[sampler] if sampler in SAMPLER_IDS else EXTRA_SAMPLER_IDS[sampler]
        batch = int(src.shape[0])
        if _is_torch(src):
            import torch
            nbr = torch.empty((batch, k), dtype=torch.int64, device=src.device)
            eid = torch.empty((batch, k), dtype=torch.int64, device=src.device)
        else:
            nbr = np.empty((batch, k), np.int64)
            eid = np.empty((batch, k), np.int64)
        ps, pn, pe, pr, pv = _ptr(src), _ptr(nbr), _ptr(eid), _ptr(rng_rows), _ptr(values)
        kind = _kind(ps, pn, pe, pr, pv)
        flt = Filter(filter_type, filter_field, pv[0], retry_times, default_timestamp)
        _check(lib().glx_sample_filtered(self._h, sampler, ps[0], pr[0], batch, k, padding_mode, default_neighbor_id,
                                         seed, call_counter, ctypes.byref(flt), pn[0], pe[0], kind, _stream(kind, self.device)))
        return nbr, eid

    def sample_full_filtered(self, src, max_limit, filter_type, filter_field, values, padding_mode=PAD_CIRCULAR,
                             default_neighbor_id=0, default_timestamp=-1):
        """FullSampler with a Filter: -> (degrees, nbr, eid); the segments are the unfiltered ones."""
        batch = int(src.shape[0])
        torch_in = _is_torch(src)
        if torch_in:
            import torch
            deg = torch.empty(batch, dtype=torch.int32, device=src.device)
            off = torch.empty(batch + 1, dtype=torch.int64, device=src.device)
        else:
            deg = np.empty(batch, np.int32)
            off = np.empty(batch + 1, np.int64)
        kind = PTR_DEVICE if torch_in else PTR_HOST
        _check(lib().glx_sample_full_sizes(self._h, _ptr(src)[0], batch, max_limit, _ptr(deg)[0], _ptr(off)[0], kind,
                                           _stream(kind, self.device)))
        total = int(off[-1].item()) if torch_in else int(off[-1])
        if torch_in:
            nbr = torch.empty(total, dtype=torch.int64, device=src.device)
            eid = torch.empty(total, dtype=torch.int64, device=src.device)
        else:
            nbr = np.empty(total, np.int64)
            eid = np.empty(total, np.int64)
        flt = Filter(filter_type, filter_field, _ptr(values)[0], 0, default_timestamp)
        _check(lib().glx_sample_full_filtered(self._h, _ptr(src)[0], batch, max_limit, _ptr(off)[0], padding_mode,
                                              default_neighbor_id, ctypes.byref(flt), _ptr(nbr)[0], _ptr(eid)[0], kind,
                                              _stream(kind, self.device)))
        return deg, nbr, eid

    def random_walk(self, seeds, walk_len, p=1.0, q=1.0, full_nbr_num=100, default_weight=0.0, default_neighbor_id=0,
                    seed=0, call_counter=0):
        """RandomWalk operator: -> walks[batch, walk_len] (DeepWalk when p = q = 1, node2vec otherwise)."""
        batch = int(seeds.shape[0])
        if _is_torch(seeds):
            import torch
            walks = torch.empty((batch, walk_len), dtype=torch.int64, device=seeds.device)
        else:
            walks = np.empty((batch, walk_len), np.int64)
        ps, pw = _ptr(seeds), _ptr(walks)
        kind = _kind(ps, pw)
        _check(lib().glx_random_walk(self._h, ps[0], batch, walk_len, p, q, full_nbr_num, default_weight,
                                     default_neighbor_id, seed, call_counter, pw[0], kind, _stream(kind, self.device)))
        return walks

    def export_alias(self):
        prob = np.empty(self.num_edges, np.float32)
        alias = np.empty(self.num_edges, np.int32)
        _check(lib().glx_graph_export_alias(self._h, _ptr(prob)[0], _ptr(alias)[0], PTR_HOST, None))
        return prob, alias

    def degrees(self, src):
        if _is_torch(src):
            import torch
            out = torch.empty(src.shape[0], dtype=torch.int64, device=src.device)
        else:
            out = np.empty(src.shape[0], np.int64)
        (ps, k1), (po, _) = _ptr(src), _ptr(out)
        _check(lib().glx_graph_degrees(self._h, ps, src.shape[0], po, k1, _stream(k1, self.device)))
        return out

    def in_degrees(self, ids):
        """In-degrees of raw destination ids (needs enable_in_degree())."""
        if _is_torch(ids):
            import torch
            out = torch.empty(ids.shape[0], dtype=torch.int64, device=ids.device)
        else:
            out = np.empty(ids.shape[0], np.int64)
        (ps, k1), (po, _) = _ptr(ids), _ptr(out)
        _check(lib().glx_graph_in_degrees(self._h, ps, ids.shape[0], po, k1, _stream(k1, self.device)))
        return out

    def sample(self, sampler, src, k, seed=0, call_counter=0, padding_mode=PAD_CIRCULAR,
               default_neighbor_id=0, out=None, rng_rows=None):
        """-> (nbr[batch, k], eid[batch, k]) int64; numpy in -> numpy out, torch in -> torch out."""
        if isinstance(sampler, str):
            sampler = SAMPLER_IDS[sampler] if sampler in SAMPLER_IDS else EXTRA_SAMPLER_IDS[sampler]
        batch = int(src.shape[0])
        if out is not None:
            nbr, eid = out
        elif _is_torch(src):
            import torch
            nbr = torch.empty((batch, k), dtype=torch.int64, device=src.device)
            eid = torch.empty((batch, k), dtype=torch.int64, device=src.device)
        else:
            nbr = np.empty((batch, k), np.int64)
            eid = np.empty((batch, k), np.int64)
        ps, pn, pe, pr = _ptr(src), _ptr(nbr), _ptr(eid), _ptr(rng_rows)
        kind = _kind(ps, pn, pe, pr)
        _check(lib().glx_sample_ex(self._h, sampler, ps[0], pr[0], batch, k, padding_mode,
                                   default_neighbor_id, seed, call_counter, pn[0], pe[0], kind,
                                   _stream(kind, self.device)))
        return nbr, eid


class Features:
    """Device-resident [V, D] float32 node features of one node type (glx_features)."""

    def __init__(self, X, ids=None, device=0, view=False):
        """view=True: non-owning view of a torch CUDA matrix (dense ids); keeps X alive."""
        if view:
            assert ids is None and _is_torch(X)
            self.num_rows, self.dim = int(X.shape[0]), int(X.shape[1])
            self.device = device
            self._keep = X
            h = ctypes.c_void_p()
            _check(lib().glx_features_view(device, self.num_rows, self.dim, _ptr(X)[0], ctypes.byref(h)))
            self._h = h
            return
        ptrs = [_ptr(X), _ptr(ids)]
        kind = _kind(*ptrs)
        self.num_rows, self.dim = int(X.shape[0]), int(X.shape[1])
        self.device = device
        h = ctypes.c_void_p()
        _check(lib().glx_features_create(device, self.num_rows, self.dim, ptrs[0][0], ptrs[1][0], kind,
                                         _stream(kind, self.device), ctypes.byref(h)))
        self._h = h

    @classmethod
    def from_handle(cls, handle, device=0):
        """Borrow a glx_features* owned by someone else (the C++ host layer's Noder)."""
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p(int(handle))
        self._borrowed = True
        v, d, dv = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int()
        _check(lib().glx_features_info(self._h, ctypes.byref(v), ctypes.byref(d), None, ctypes.byref(dv)))
        self.num_rows, self.dim = v.value, d.value
        self.device = dv.value  # the device the handle lives on, whatever the caller passed
        return self

    def close(self):
        if getattr(self, "_borrowed", False):
            self._h = None
        if getattr(self, "_h", None):
            try:
                lib().glx_features_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    __del__ = close

    def aggregate(self, op, node_ids, segment_ids, num_segments, default_attr=0.0, out=None):
        """-> (emb[num_segments, D] float32, counts[num_segments] int32)."""
        if isinstance(op, str):
            op = AGGREGATOR_IDS[op]
        n = int(node_ids.shape[0])
        if out is not None:
            emb, cnt = out
        elif _is_torch(node_ids):
            import torch
            emb = torch.empty((num_segments, self.dim), dtype=torch.float32, device=node_ids.device)
            cnt = torch.empty((num_segments,), dtype=torch.int32, device=node_ids.device)
        else:
            emb = np.empty((num_segments, self.dim), np.float32)
            cnt = np.empty((num_segments,), np.int32)
        # segment_ids=None: num_segments equal segments of n / num_segments ids (a dense sampler response)
        pi, pg, pe, pc = _ptr(node_ids), _ptr(segment_ids), _ptr(emb), _ptr(cnt)
        kind = _kind(pi, pg, pe, pc)
        _check(lib().glx_aggregate(self._h, op, pi[0], pg[0], n, num_segments, default_attr, pe[0],
                                   pc[0], kind, _stream(kind, self.device)))
        return emb, cnt

    def lookup(self, node_ids, default_attr=0.0):
        n = int(node_ids.shape[0])
        if _is_torch(node_ids):
            import torch
            out = torch.empty((n, self.dim), dtype=torch.float32, device=node_ids.device)
        else:
            out = np.empty((n, self.dim), np.float32)
        pi, po = _ptr(node_ids), _ptr(out)
        kind = _kind(pi, po)
        _check(lib().glx_lookup(self._h, pi[0], n, default_attr, po[0], kind, _stream(kind, self.device)))
        return out


def sample_hops(graphs, sampler, seeds, fanouts, seed=0, call_counter=0, padding_mode=PAD_CIRCULAR,
                default_neighbor_id=0):
    """Multi-hop driver (NeighborSampler.get): hop h from graphs[h] with fanouts[h].
    -> list of (nbr[rows_h, fanouts[h]], eid[...]) per hop."""
    if isinstance(sampler, str):
        sampler = SAMPLER_IDS[sampler]
    L = len(fanouts)
    assert len(graphs) == L
    rows = int(seeds.shape[0])
    outs = []
    torch_mode = _is_torch(seeds)
    for k in fanouts:
        if torch_mode:
            import torch
            nbr = torch.empty((rows, k), dtype=torch.int64, device=seeds.device)
            eid = torch.empty((rows, k), dtype=torch.int64, device=seeds.device)
        else:
            nbr = np.empty((rows, k), np.int64)
            eid = np.empty((rows, k), np.int64)
        outs.append((nbr, eid))
        rows *= k
    kind = PTR_DEVICE if torch_mode else PTR_HOST
    gh = (ctypes.c_void_p * L)(*[g._h for g in graphs])
    fo = (ctypes.c_int32 * L)(*fanouts)
    pn = (ctypes.c_void_p * L)(*[_ptr(o[0])[0] for o in outs])
    pe = (ctypes.c_void_p * L)(*[_ptr(o[1])[0] for o in outs])
    _check(lib().glx_sample_hops(gh, L, sampler, _ptr(seeds)[0], int(seeds.shape[0]), fo, padding_mode,
                                 default_neighbor_id, seed, call_counter, pn, pe, kind, _stream(kind, graphs[0].device)))
    return outs


def partition(ids, num_shards):
    """Device HashPartitioner: torch int64 CUDA ids -> (bucketed, order, counts)."""
    import torch
    n = int(ids.shape[0])
    bucketed = torch.empty_like(ids)
    order = torch.empty_like(ids)
    counts = torch.empty(num_shards, dtype=torch.int64, device=ids.device)
    dev = ids.device.index or 0
    _check(lib().glx_partition(dev, _ptr(ids)[0], n, num_shards, _ptr(bucketed)[0], _ptr(order)[0],
                               _ptr(counts)[0], _stream(PTR_DEVICE, dev)))
    return bucketed, order, counts


def stitch(rows, order):
    """Device Stitcher: out[order[i]] = rows[i]; rows is [n, width] int64 or float32 (torch CUDA)."""
    import torch
    n = int(rows.shape[0])
    width = int(rows.numel() // max(n, 1)) if n else 1
    out = torch.empty_like(rows)
    dev = rows.device.index or 0
    fn = lib().glx_stitch_i64 if rows.dtype == torch.int64 else lib().glx_stitch_f32
    assert rows.dtype in (torch.int64, torch.float32)
    _check(fn(dev, _ptr(rows)[0], _ptr(order)[0], n, width, _ptr(out)[0], _stream(PTR_DEVICE, dev)))
    return out


NEG_EXCLUDE_NONE, NEG_EXCLUDE_NEIGHBORS, NEG_EXCLUDE_BATCH = 0, 1, 2


class Negative:
    """Candidate list of the negative samplers in HBM (glx_negative)."""

    def __init__(self, ids, weights=None, device=0):
        ptrs = [_ptr(ids), _ptr(weights)]
        kind = _kind(*ptrs)
        h = ctypes.c_void_p()
        _check(lib().glx_negative_create(device, int(ids.shape[0]), ptrs[0][0], ptrs[1][0], kind, _stream(kind, device),
                                         ctypes.byref(h)))
        self._h = h
        self.device = device
        self._info()

    @classmethod
    def from_graph(cls, graph, by_in_degree=False):
        """Distinct destination ids of the edge type in first-appearance order, uniform or
        weighted by in-degree."""
        self = cls.__new__(cls)
        h = ctypes.c_void_p()
        _check(lib().glx_negative_from_graph(graph._h, 1 if by_in_degree else 0, None, ctypes.byref(h)))
        self._h = h
        self.device = graph.device
        self._info()
        return self

    def _info(self):
        n, w = ctypes.c_int64(), ctypes.c_int()
        _check(lib().glx_negative_info(self._h, ctypes.byref(n), ctypes.byref(w)))
        self.num_ids, self.weighted = n.value, bool(w.value)

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().glx_negative_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    __del__ = close

    def export(self):
        """-> (ids, prob or None, alias or None) as numpy arrays."""
        ids = np.zeros(self.num_ids, np.int64)
        prob = np.zeros(self.num_ids, np.float32) if self.weighted else None
        alias = np.zeros(self.num_ids, np.int32) if self.weighted else None
        _check(lib().glx_negative_export(self._h, _ptr(ids)[0], _ptr(prob)[0], _ptr(alias)[0], None))
        return ids, prob, alias

    def sample(self, src, count, exclude=NEG_EXCLUDE_NONE, graph=None, default_neighbor_id=0, seed=0,
               call_counter=0):
        """-> out[batch, count] candidate ids (numpy in -> numpy out, torch CUDA in -> torch CUDA out)."""
        batch = int(src.shape[0])
        if _is_torch(src):
            import torch
            out = torch.empty((batch, count), dtype=torch.int64, device=src.device)
        else:
            out = np.empty((batch, count), np.int64)
        ps, po = _ptr(src), _ptr(out)
        kind = _kind(ps, po)
        _check(lib().glx_negative_sample(self._h, exclude, graph._h if graph is not None else None, ps[0], batch,
                                         count, default_neighbor_id, seed, call_counter, po[0], kind, _stream(kind, self.device)))
        return out


def aggregate_stitch(op, parts, cnts, default_attr=0.0):
    """Device AggregatingResponse::Stitch: parts [P, Sg, D] float32 + cnts [P, Sg] int32
    (torch CUDA, contiguous: the receive buffers of one all-to-all) -> (emb [Sg, D], cnt [Sg])."""
    import torch
    if isinstance(op, str):
        op = AGGREGATOR_IDS[op]
    P, sg, dim = (int(x) for x in parts.shape)
    assert parts.dtype == torch.float32 and cnts.dtype == torch.int32 and tuple(cnts.shape) == (P, sg)
    assert parts.is_contiguous() and cnts.is_contiguous()
    emb = torch.empty((sg, dim), dtype=torch.float32, device=parts.device)
    cnt = torch.empty((sg,), dtype=torch.int32, device=parts.device)
    dev = parts.device.index or 0
    _check(lib().glx_aggregate_stitch(dev, op, P, _ptr(parts)[0], _ptr(cnts)[0], sg, dim, default_attr,
                                      _ptr(emb)[0], _ptr(cnt)[0], _stream(PTR_DEVICE, dev)))
    return emb, cnt


COMM_RCCL, COMM_LOCAL, COMM_CALLBACKS = 0, 1, 2
UNIQUE_ID_BYTES = 128


class Comm:
    """Shard communicator (glx_comm): RCCL over xGMI, in-process threads, or host-staged callbacks."""

    def __init__(self, handle, keep=None):
        self._h = handle
        self._keep = keep  # ctypes callback objects must outlive the handle
        r, w, d, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(lib().glx_comm_info(handle, ctypes.byref(r), ctypes.byref(w), ctypes.byref(d), ctypes.byref(t)))
        self.rank, self.world, self.device, self.transport = r.value, w.value, d.value, t.value

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        _check(lib().glx_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def rccl(cls, device, rank, world, unique_id):
        h = ctypes.c_void_p()
        assert len(unique_id) == UNIQUE_ID_BYTES
        _check(lib().glx_comm_init_rccl(device, rank, world, ctypes.c_char_p(unique_id), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def local(cls, fabric_key, device, rank, world):
        """Ranks = threads of this process (ctypes releases the GIL inside the collectives)."""
        h = ctypes.c_void_p()
        _check(lib().glx_comm_init_local(fabric_key, device, rank, world, ctypes.byref(h)))
        return cls(h)

    @classmethod
    def callbacks(cls, device, rank, world, alltoallv, allgather):
        """Host-staged transport.  alltoallv(send_u8, send_counts, recv_u8, recv_counts, elem_bytes) and
        allgather(send_u8, recv_u8) get numpy views of the pinned staging buffers."""
        def _a2a(_user, send, send_counts, recv, recv_counts, eb):
            try:
                sc = np.ctypeslib.as_array((ctypes.c_int64 * world).from_address(send_counts)).copy()
                rc = np.ctypeslib.as_array((ctypes.c_int64 * world).from_address(recv_counts)).copy()
                ns, nr = int(sc.sum()) * eb, int(rc.sum()) * eb
                sb = np.ctypeslib.as_array((ctypes.c_uint8 * max(ns, 1)).from_address(send))[:ns] if send else np.empty(0, np.uint8)
                rb = np.ctypeslib.as_array((ctypes.c_uint8 * max(nr, 1)).from_address(recv))[:nr] if recv else np.empty(0, np.uint8)
                alltoallv(sb, sc, rb, rc, int(eb))
                return 0
            except Exception:  # noqa: BLE001 -- an exception must not unwind through C
                import traceback
                traceback.print_exc()
                return 1

        def _ag(_user, send, recv, nbytes):
            try:
                sb = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(send))
                rb = np.ctypeslib.as_array((ctypes.c_uint8 * (nbytes * world)).from_address(recv))
                allgather(sb, rb)
                return 0
            except Exception:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1
        c1, c2 = HOST_ALLTOALLV_FN(_a2a), HOST_ALLGATHER_FN(_ag)
        h = ctypes.c_void_p()
        _check(lib().glx_comm_init_callbacks(device, rank, world, c1, c2, None, ctypes.byref(h)))
        return cls(h, keep=(c1, c2))

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().glx_comm_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    __del__ = close

    def set_max_message_bytes(self, nbytes):
        return int(lib().glx_comm_set_max_message_bytes(self._h, int(nbytes)))

    def exchange_v(self, send, send_counts, recv_counts):
        """all-to-all(v) of the rows of `send` (numpy or torch CUDA, packed peer-major) -> received rows."""
        sc = np.ascontiguousarray(send_counts, dtype=np.int64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.int64)
        n = int(rc.sum())
        width = 1
        for d in send.shape[1:]:
            width *= int(d)
        if _is_torch(send):
            import torch
            out = torch.empty((n,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
            eb = send.element_size() * width
        else:
            out = np.empty((n,) + tuple(send.shape[1:]), send.dtype)
            eb = send.dtype.itemsize * width
        ps, po = _ptr(send), _ptr(out)
        kind = _kind(ps, po)
        _check(lib().glx_exchange_v(self._h, ps[0], _ptr(sc)[0], po[0], _ptr(rc)[0], eb, kind, _stream(kind, self.device)))
        return out

    def allgather_i64(self, vals):
        """vals[nvals] int64 (numpy or torch CUDA) of every rank -> [world, nvals] of the same kind."""
        n = int(vals.shape[0])
        if _is_torch(vals):
            import torch
            out = torch.empty((self.world, n), dtype=torch.int64, device=vals.device)
        else:
            out = np.empty((self.world, n), np.int64)
        pv, po = _ptr(vals), _ptr(out)
        kind = _kind(pv, po)
        _check(lib().glx_comm_allgather_i64(self._h, pv[0], n, po[0], kind, _stream(kind, self.device)))
        return out

    def barrier(self):
        _check(lib().glx_comm_barrier(self._h, _stream(PTR_DEVICE, self.device)))


class DistStore:
    """One rank's shard of an edge-cut partitioned graph + feature table behind a communicator
    (glx_dist_store): the device-resident DistributeRunner.  Every method except stats() is
    COLLECTIVE: all ranks of the communicator call it together, each with its own request."""

    def __init__(self, comm, graph=None, features=None):
        h = ctypes.c_void_p()
        _check(lib().glx_dist_store_create(comm._h, graph._h if graph is not None else None,
                                           features._h if features is not None else None, ctypes.byref(h)))
        self._h = h
        self.comm, self.graph, self.features = comm, graph, features
        self.dim = features.dim if features is not None else None

    def close(self):
        if getattr(self, "_h", None):
            try:
                lib().glx_dist_store_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    __del__ = close

    def set_cache(self, hot_ids, default_attr=0.0):
        """Replicate the rows of hot_ids (same list on every rank) on this GPU; empty list drops the replica."""
        n = int(hot_ids.shape[0])
        p, kind = _ptr(hot_ids) if n else (None, PTR_HOST)
        _check(lib().glx_dist_store_set_cache(self._h, p, n, default_attr, kind, _stream(kind, self.comm.device)))

    def set_graph_replica(self, replica):
        """Serve request rows of the vertices `replica` (a glx.Graph built with explicit ids, holding their complete
        adjacency rows) on this GPU instead of sending them to their owners; None detaches it.  The store keeps a
        reference so the replica outlives its use."""
        _check(lib().glx_dist_store_set_graph_replica(self._h, replica._h if replica is not None else None))
        self._graph_replica = replica

    def build_graph_replica(self, hot_ids, attach=True):
        """Collective: the complete adjacency rows of hot_ids (same list on every rank; unique ids), cut out of the
        shards by their owners and all-gathered -> a glx.Graph on this GPU (owned by the caller).  attach=True also
        makes the store serve those vertices' sampling requests from it (set_graph_replica)."""
        n = int(hot_ids.shape[0])
        p, kind = _ptr(hot_ids)
        h = ctypes.c_void_p()
        _check(lib().glx_dist_build_graph_replica(self._h, p, n, kind, _stream(kind, self.comm.device), ctypes.byref(h)))
        g = Graph.from_handle(h.value)
        g._borrowed = False  # ours to destroy
        if attach:
            self.set_graph_replica(g)
        return g

    def sample_full(self, src, max_limit=0, filter_type=FILTER_NONE, filter_field=FILTER_FIELD_NONE, values=None,
                    padding_mode=PAD_CIRCULAR, default_neighbor_id=0, default_timestamp=-1):
        """Collective FullSampler over the shards: -> (degrees[batch] int32, nbr[total], eid[total]) for this rank's
        rows, as Graph.sample_full / sample_full_filtered answer on one store (numpy in -> numpy out, torch CUDA in ->
        torch CUDA out).  With a filter: values[batch], same kind as src; every rank passes the same kind of filter."""
        batch = int(src.shape[0])
        if _is_torch(src):
            import torch
            mk = lambda n, dt: torch.empty(n, dtype=dt, device=src.device)  # noqa: E731
            deg, off, i64t = mk(batch, torch.int32), mk(batch + 1, torch.int64), torch.int64
        else:
            mk = lambda n, dt: np.empty(n, dt)  # noqa: E731
            deg, off, i64t = mk(batch, np.int32), mk(batch + 1, np.int64), np.int64
        ps, kind = _ptr(src) if batch else (None, PTR_DEVICE if _is_torch(src) else PTR_HOST)
        stream = _stream(kind, self.comm.device)
        _check(lib().glx_dist_sample_full_sizes(self._h, ps, batch, max_limit, _ptr(deg)[0] if batch else None, _ptr(off)[0],
                                                kind, stream))
        total = int(off[-1])
        nbr, eid = mk(total, i64t), mk(total, i64t)
        if filter_type != FILTER_NONE:
            flt = Filter(filter_type, filter_field, _ptr(values)[0] if batch else None, 0, default_timestamp)
            _check(lib().glx_dist_sample_full_filtered(self._h, ps, batch, max_limit, _ptr(deg)[0] if batch else None,
                                                       _ptr(off)[0], padding_mode, default_neighbor_id, ctypes.byref(flt),
                                                       _ptr(nbr)[0] if total else None, _ptr(eid)[0] if total else None, total,
                                                       kind, stream))
            return deg, nbr, eid
        _check(lib().glx_dist_sample_full(self._h, ps, batch, max_limit, _ptr(deg)[0] if batch else None, _ptr(off)[0],
                                          _ptr(nbr)[0] if total else None, _ptr(eid)[0] if total else None, total, kind, stream))
        return deg, nbr, eid

    def in_degrees(self, ids):
        """Collective GetDegree for destination ids: in-degrees summed over all shards (int32; numpy or torch CUDA)."""
        n = int(ids.shape[0])
        if _is_torch(ids):
            import torch
            out, kind = torch.empty(n, dtype=torch.int32, device=ids.device), PTR_DEVICE
        else:
            out, kind = np.empty(n, np.int32), PTR_HOST
        _check(lib().glx_dist_in_degrees(self._h, _ptr(ids)[0] if n else None, n, _ptr(out)[0] if n else None, kind,
                                         _stream(kind, self.comm.device)))
        return out

    def negative_table(self, by_in_degree=False):
        """Collective: the negative samplers' candidate list over the WHOLE edge type (every shard's destination ids,
        ascending, uniform or weighted by global in-degree) -> a Negative, identical on every rank."""
        t = Negative.__new__(Negative)
        h = ctypes.c_void_p()
        _check(lib().glx_dist_negative_create(self._h, 1 if by_in_degree else 0, _stream(PTR_DEVICE, self.comm.device),
                                              ctypes.byref(h)))
        t._h, t.device = h, self.comm.device
        t._info()
        return t

    def negative_sample(self, table, src, count, exclude=NEG_EXCLUDE_NONE, default_neighbor_id=0, seed=0, call_counter=0):
        """glx_negative_sample across the shards (collective for NEG_EXCLUDE_NEIGHBORS): -> out[batch, count]."""
        batch = int(src.shape[0])
        if _is_torch(src):
            import torch
            out, kind = torch.empty((batch, count), dtype=torch.int64, device=src.device), PTR_DEVICE
        else:
            out, kind = np.empty((batch, count), np.int64), PTR_HOST
        _check(lib().glx_dist_negative_sample(self._h, table._h, exclude, _ptr(src)[0] if batch else None, batch, count,
                                              default_neighbor_id, seed, call_counter,
                                              _ptr(out)[0] if batch * count else None, kind, _stream(kind, self.comm.device)))
        return out

    def random_walk(self, seeds, walk_len, p=1.0, q=1.0, default_neighbor_id=0, seed=0, call_counter=0, full_nbr_num=100,
                    default_weight=0.0):
        """Collective RandomWalk over the shards (DeepWalk when p = q = 1, node2vec otherwise): -> walks[batch, walk_len],
        Graph.random_walk's draws."""
        batch = int(seeds.shape[0])
        if _is_torch(seeds):
            import torch
            walks = torch.empty((batch, walk_len), dtype=torch.int64, device=seeds.device)
            kind = PTR_DEVICE
        else:
            walks = np.empty((batch, walk_len), np.int64)
            kind = PTR_HOST
        _check(lib().glx_dist_random_walk_ex(self._h, _ptr(seeds)[0] if batch else None, batch, walk_len, p, q, full_nbr_num,
                                             default_weight, default_neighbor_id, seed, call_counter,
                                             _ptr(walks)[0] if batch * walk_len else None, kind,
                                             _stream(kind, self.comm.device)))
        return walks

    def last_sample_rows(self):
        """{'rows', 'from_graph_replica', 'remote'} of the last sample() on this rank."""
        a, b, c = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        _check(lib().glx_dist_last_sample_rows(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"rows": a.value, "from_graph_replica": b.value, "remote": c.value}

    def hot_ids(self, want):
        """The `want` destination ids with the largest global in-degree (numpy int64, same on every rank)."""
        out = np.empty(max(int(want), 1), np.int64)
        n = ctypes.c_int64(0)
        _check(lib().glx_dist_hot_ids(self._h, int(want), _ptr(out)[0], ctypes.byref(n), _stream(PTR_DEVICE, self.comm.device)))
        return out[:n.value].copy()

    def enable_in_degree(self):
        """Collective: build the shard's InDegreeSampler tables from in-degrees summed over all shards."""
        _check(lib().glx_dist_enable_in_degree(self._h, self.graph._h, _stream(PTR_DEVICE, self.comm.device)))

    def sample(self, sampler, src, k, seed=0, call_counter=0, padding_mode=PAD_CIRCULAR, default_neighbor_id=0,
               out=None, filter_type=FILTER_NONE, filter_field=FILTER_FIELD_NONE, values=None, retry_times=5,
               default_timestamp=-1):
        if isinstance(sampler, str):
            sampler = SAMPLER_IDS[sampler] if sampler in SAMPLER_IDS else EXTRA_SAMPLER_IDS[sampler]
        batch = int(src.shape[0])
        if out is not None:
            nbr, eid = out
        elif _is_torch(src):
            import torch
            nbr = torch.empty((batch, k), dtype=torch.int64, device=src.device)
            eid = torch.empty((batch, k), dtype=torch.int64, device=src.device)
        else:
            nbr = np.empty((batch, k), np.int64)
            eid = np.empty((batch, k), np.int64)
        ps, pn, pe, pv = _ptr(src), _ptr(nbr), _ptr(eid), _ptr(values)
        kind = _kind(ps, pn, pe, pv)
        flt = None
        if values is not None and filter_type != FILTER_NONE:
            flt = ctypes.byref(Filter(filter_type, filter_field, pv[0], retry_times, default_timestamp))
        _check(lib().glx_dist_sample(self._h, sampler, ps[0], batch, k, padding_mode, default_neighbor_id, seed,
                                     call_counter, flt, pn[0], pe[0], kind, _stream(kind, self.comm.device)))
        return nbr, eid

    def aggregate(self, op, node_ids, segment_ids, num_segments, default_attr=0.0, out=None, partial=False):
        """segment_ids=None: num_segments equal segments (a dense sampler response).  partial=True: design R -- owners
        reduce, the requester folds the partial results (glx_dist_aggregate_partial) instead of the halo-row exchange."""
        if isinstance(op, str):
            op = AGGREGATOR_IDS[op]
        n = int(node_ids.shape[0])
        if out is not None:
            emb, cnt = out
        elif _is_torch(node_ids):
            import torch
            emb = torch.empty((num_segments, self.dim), dtype=torch.float32, device=node_ids.device)
            cnt = torch.empty((num_segments,), dtype=torch.int32, device=node_ids.device)
        else:
            emb = np.empty((num_segments, self.dim), np.float32)
            cnt = np.empty((num_segments,), np.int32)
        pi, pg, pe, pc = _ptr(node_ids), _ptr(segment_ids), _ptr(emb), _ptr(cnt)
        kind = _kind(pi, pg, pe, pc)
        fn = lib().glx_dist_aggregate_partial if partial else lib().glx_dist_aggregate
        _check(fn(self._h, op, pi[0], pg[0], n, num_segments, default_attr, pe[0], pc[0], kind, _stream(kind, self.comm.device)))
        return emb, cnt

    def aggregate_begin(self, slot, node_ids, default_attr=0.0):
        """Collective half of aggregate(): resolve the ids and fetch the halo rows into buffer set `slot`
        (torch CUDA ids, on the current stream) -- may run ahead of aggregate_end, beside an earlier reduce."""
        assert _is_torch(node_ids) and node_ids.is_cuda and node_ids.is_contiguous()
        _check(lib().glx_dist_aggregate_begin(self._h, slot, _ptr(node_ids)[0], int(node_ids.shape[0]), default_attr,
                                              _stream(PTR_DEVICE, self.comm.device)))

    def aggregate_end(self, slot, op, segment_ids, num_segments, out):
        """Local half: the segmented reduce over the rows aggregate_begin(slot, ...) prepared."""
        if isinstance(op, str):
            op = AGGREGATOR_IDS[op]
        emb, cnt = out
        _check(lib().glx_dist_aggregate_end(self._h, slot, op, _ptr(segment_ids)[0], num_segments, _ptr(emb)[0],
                                            _ptr(cnt)[0], _stream(PTR_DEVICE, self.comm.device)))
        return emb, cnt

    def aggregate_end_range(self, slot, first_id, num_ids, op, segment_ids, num_segments, out, release=False):
        """The reduce over ids [first_id, first_id + num_ids) of the request begun in `slot` (one begin, several
        aggregating requests: glx_dist_aggregate_end_range); release=True ends the slot's request."""
        if isinstance(op, str):
            op = AGGREGATOR_IDS[op]
        emb, cnt = out
        _check(lib().glx_dist_aggregate_end_range(self._h, slot, int(first_id), int(num_ids), 1 if release else 0, op,
                                                  _ptr(segment_ids)[0], num_segments, _ptr(emb)[0], _ptr(cnt)[0],
                                                  _stream(PTR_DEVICE, self.comm.device)))
        return emb, cnt

    def lookup(self, node_ids, default_attr=0.0):
        n = int(node_ids.shape[0])
        if _is_torch(node_ids):
            import torch
            out = torch.empty((n, self.dim), dtype=torch.float32, device=node_ids.device)
        else:
            out = np.empty((n, self.dim), np.float32)
        pi, po = _ptr(node_ids), _ptr(out)
        kind = _kind(pi, po)
        _check(lib().glx_dist_lookup(self._h, pi[0], n, default_attr, po[0], kind, _stream(kind, self.comm.device)))
        return out

    def stats(self):
        st = DistStats()
        _check(lib().glx_dist_last_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def confirm(self):
        """A confirmation point for speculated sample calls (Ledger): one count exchange; raises GlxError(ABORTED)
        on every rank when a speculated message overflowed."""
        _check(lib().glx_dist_confirm(self._h, _stream(PTR_DEVICE, self.comm.device)))


class LedgerStats(ctypes.Structure):
    _fields_ = [("speculated", ctypes.c_int64), ("learned", ctypes.c_int64), ("aborted", ctypes.c_int64),
                ("holding", ctypes.c_int64), ("largest_share", ctypes.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Ledger:
    """glx_dist_ledger: lets glx_dist_sample skip its count exchange for requests whose shape repeats (fixed-capacity
    messages; whether they fitted is confirmed by the next count exchange of any store attached to the ledger, which
    raises GlxError with code ABORTED on every rank when they did not).  One per rank."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        _check(lib().glx_dist_ledger_create(int(device), ctypes.byref(h)))
        self._h = h
        self._stores = []

    def attach(self, *stores):
        for st in stores:
            _check(lib().glx_dist_store_set_ledger(st._h, self._h))
            st._ledger = self  # the ledger outlives the stores attached to it
            self._stores.append(st)
        return self

    def detach(self, *stores):
        for st in stores:
            if getattr(st, "_h", None):  # (a store closed before its ledger has nothing left to detach)
                _check(lib().glx_dist_store_set_ledger(st._h, None))
            st._ledger = None
            self._stores = [x for x in self._stores if x is not st]

    def set_slack(self, slack=1.25, pad_rows=1024):
        _check(lib().glx_dist_ledger_set_slack(self._h, float(slack), int(pad_rows)))

    def stats(self):
        st = LedgerStats()
        _check(lib().glx_dist_ledger_get_stats(self._h, ctypes.byref(st)))
        return st.as_dict()

    def close(self):
        if self._h:
            self.detach(*list(self._stores))
            lib().glx_dist_ledger_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _DevArray:
    """A device buffer owned by a C handle, exposed to torch without a copy (CUDA array interface)."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner  # keeps the plan alive while a tensor view exists


class Plan:
    """A fixed-shape multi-hop sample (+ aggregate) request captured into one hipGraph (glx_plan): the
    launch-bound small-batch path.  run(seeds, call_counter) replays it on torch's current stream; the outputs
    are views of plan-owned device buffers, overwritten by the next run."""

    def __init__(self, graphs, sampler, fanouts, batch, features=None, agg=None, seed=0, padding_mode=PAD_CIRCULAR,
                 default_neighbor_id=0, default_attr=0.0):
        import torch
        if isinstance(sampler, str):
            sampler = SAMPLER_IDS[sampler] if sampler in SAMPLER_IDS else EXTRA_SAMPLER_IDS[sampler]
        L = len(fanouts)
        assert len(graphs) == L and (features is None or len(features) == L)
        self.device = graphs[0].device
        self._keep = (list(graphs), list(features) if features is not None else None)
        gh = (ctypes.c_void_p * L)(*[g._h for g in graphs])
        fo = (ctypes.c_int32 * L)(*[int(f) for f in fanouts])
        fh = (ctypes.c_void_p * L)(*[f._h for f in features]) if features is not None else None
        op = AGGREGATOR_IDS[agg] if isinstance(agg, str) else (agg if agg is not None else 0)
        h = ctypes.c_void_p()
        _check(lib().glx_plan_create(gh, L, sampler, fo, int(batch), padding_mode, default_neighbor_id, seed, fh, op,
                                     default_attr, ctypes.byref(h)))
        self._h = h
        self.batch, self.num_hops = int(batch), L
        self.hops = []
        dev = torch.device("cuda", self.device)
        for hop in range(L):
            pn, pe, pm, pc = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            rows, fan = ctypes.c_int64(), ctypes.c_int32()
            _check(lib().glx_plan_output(h, hop, ctypes.byref(pn), ctypes.byref(pe), ctypes.byref(pm), ctypes.byref(pc),
                                         ctypes.byref(rows), ctypes.byref(fan)))
            r, k = rows.value, fan.value
            out = {"nbr": torch.as_tensor(_DevArray(pn.value, (r, k), "<i8", self), device=dev),
                   "eid": torch.as_tensor(_DevArray(pe.value, (r, k), "<i8", self), device=dev)}
            if features is not None:
                out["emb"] = torch.as_tensor(_DevArray(pm.value, (r, features[hop].dim), "<f4", self), device=dev)
                out["cnt"] = torch.as_tensor(_DevArray(pc.value, (r,), "<i4", self), device=dev)
            self.hops.append(out)

    def run(self, seeds, call_counter=0):
        assert _is_torch(seeds) and seeds.is_cuda and seeds.is_contiguous() and int(seeds.shape[0]) == self.batch
        _check(lib().glx_plan_run(self._h, _ptr(seeds)[0], call_counter, _stream(PTR_DEVICE, self.device)))
        return self.hops

    def close(self):
        """Frees the plan's device buffers and lets go of the stores it was captured over.  The output tensors keep their
        plan alive through the array interface (a reference cycle that runs through torch's C++ side, which the garbage
        collector cannot see): a caller that wants the memory back calls close() -- dropping the last Python name is not
        enough."""
        if getattr(self, "_h", None):
            try:
                lib().glx_plan_destroy(self._h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None
        self.hops = []
        self._keep = None

    __del__ = close


KERNEL_SAMPLE, KERNEL_AGGREGATE, KERNEL_LOOKUP = 0, 1, 2


i64_t = ctypes.c_int64
NO_KEY = -(1 << 63)  # a dst key that matches no group


class CondTable:
    """glx_cond_table: ids [U], weights [U] | None, cand_keys [ncols, U] int64 (numpy or torch-on-device)."""

    def __init__(self, ids, weights, cand_keys, device=0):
        self.device = device
        self.num_ids = int(ids.shape[0])
        self.num_cols = 0 if cand_keys is None else int(cand_keys.reshape(-1, max(self.num_ids, 1)).shape[0]) if self.num_ids else 0
        pi, pw, pk = _ptr(ids), _ptr(weights), _ptr(cand_keys)
        kind = _kind(pi, pw, pk)
        h = ctypes.c_void_p()
        _check(lib().glx_cond_table_create(device, self.num_ids, pi[0], pw[0], self.num_cols, pk[0], kind, _stream(kind, device),
                                           ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().glx_cond_table_destroy(self._h)
            self._h = None

    __del__ = close

    def sample(self, graph, src, dst, dst_keys, props, count, batch_share=False, unique=False, retry=5,
               default_neighbor_id=0, seed=0, call_counter=0):
        """-> out [batch, count] int64 (numpy in -> numpy out, torch in -> torch out)."""
        batch = int(src.shape[0])
        if _is_torch(src):
            import torch
            out = torch.empty((batch, count), dtype=torch.int64, device=src.device)
        else:
            out = np.empty((batch, count), np.int64)
        props = np.ascontiguousarray(props, np.float32)
        ps, pd, pk, po = _ptr(src), _ptr(dst), _ptr(dst_keys), _ptr(out)
        kind = _kind(ps, pd, pk, po)
        _check(lib().glx_cond_negative_sample(self._h, graph._h if graph is not None else None, ps[0], pd[0], pk[0],
                                              ctypes.c_void_p(props.ctypes.data), batch, count, int(batch_share), int(unique),
                                              retry, default_neighbor_id, seed, call_counter, po[0], kind,
                                              _stream(kind, self.device)))
        return out


def subgraph_induce(nodes, offsets, nbr, eid, device=0):
    """glx_subgraph_induce on FullSampler's response rows of `nodes` -> (row[m] int32, col[m] int32, eid[m] int64);
    numpy in -> numpy out, torch (device) in -> torch out."""
    n = int(nodes.shape[0])
    pn, po, pb, pe = _ptr(nodes), _ptr(offsets), _ptr(nbr), _ptr(eid)
    kind = _kind(pn, po, pb, pe)
    total = i64_t()
    _check(lib().glx_subgraph_induce(device, pn[0], n, po[0], pb[0], pe[0], None, None, None, 0, ctypes.byref(total), kind,
                                     _stream(kind, device)))
    m = int(total.value)
    if _is_torch(nodes):
        import torch
        row = torch.empty(m, dtype=torch.int32, device=nodes.device)
        col = torch.empty(m, dtype=torch.int32, device=nodes.device)
        out = torch.empty(m, dtype=torch.int64, device=nodes.device)
    else:
        row, col, out = np.empty(m, np.int32), np.empty(m, np.int32), np.empty(m, np.int64)
    if m:
        _check(lib().glx_subgraph_induce(device, pn[0], n, po[0], pb[0], pe[0], _ptr(row)[0], _ptr(col)[0], _ptr(out)[0], m,
                                         ctypes.byref(total), kind, _stream(kind, device)))
    return row, col, out


PROBES = {"stream_read": 0, "copy": 1, "triad": 2, "gather32": 3, "gather_rows": 4}


def probe_bandwidth(kind, nbytes, units=0, unit_bytes=0, reps=10, device=0):
    """glx_probe_bandwidth: -> dict(gbps, ms, moved_bytes) of one hand-written streaming / gather kernel."""
    moved, ms = ctypes.c_double(), ctypes.c_double()
    _check(lib().glx_probe_bandwidth(device, PROBES[kind] if isinstance(kind, str) else kind, int(nbytes), int(units),
                                     int(unit_bytes), int(reps), ctypes.byref(moved), ctypes.byref(ms),
                                     _stream(PTR_DEVICE, device)))
    return {"gbps": moved.value / (ms.value * 1e-3) / 1e9, "ms": ms.value, "moved_bytes": moved.value}


def tune(name, value):
    """glx_tune: set a measurement knob of the aggregation launch (agg_unroll, agg_legacy, agg_segs, ...)."""
    _check(lib().glx_tune(name.encode(), int(value)))


def profile_enable(on=True):
    _check(lib().glx_profile_enable(1 if on else 0))


def profile_collect(kind, cap=1 << 16):
    """Durations (ms) of the timed dominant-kernel launches of `kind`, oldest first."""
    buf = np.zeros(cap, np.float32)
    n = ctypes.c_int32(0)
    _check(lib().glx_profile_collect(kind, _ptr(buf)[0], cap, ctypes.byref(n)))
    return buf[:n.value].copy()

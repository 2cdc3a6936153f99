"""gl.Graph for the sampling path in local deploy mode.

Same construction protocol as graphlearn/python/graph.py:
    g = gl.Graph().node(path, "user", decoder).edge(path, ("user", "item", "buy"), decoder).init()
    g.neighbor_sampler(["buy"], expand_factor=[10]).get(ids)
`init()` parses the sources on the host (C++), builds CSR / alias tables / the feature
matrix on the GPU and keeps them resident there; every sampler / aggregator call then
goes Python -> pywrap -> Operator::Process -> HIP.  Multi-GPU is the SPMD mode:
init(task_index=r, task_count=P) in each of P processes (one per GPU) keeps the records that
hash to shard r, and sharded_store() serves sampling / aggregation across the shards over RCCL.
The reference's client/server deploy modes (init(cluster=...)), GSL (`V()/E()`), subgraph /
conditional-negative samplers and the vineyard backend are outside the path this engine
replaces and raise NotImplementedError.
"""
import os
import sys

import numpy as np

from graphlearn import pywrap_graphlearn as pywrap
from graphlearn import errors
from graphlearn import values as data
from graphlearn.decoder import Decoder
from graphlearn.topology import Topology
from graphlearn.utils import Mask, get_mask_type

# glx.py (the ctypes face of the C-ABI) lives next to the python/ directory
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

_PYWRAP_TYPE = {"int": pywrap.DataType.INT64, "float": pywrap.DataType.FLOAT, "string": pywrap.DataType.STRING}


def _fill_source(source, path, decoder, option):
  """Decoder -> the format bits / AttributeInfo the loader checks the file against."""
  source.path = path
  source.format = int(decoder.data_format)
  if option is None:
    option = pywrap.IndexOption()
    option.name = "sort"  # rows ordered like the reference's default Build()
  source.option = option
  info = source.attr_info
  if decoder.attributed:
    info.delimiter = decoder.attr_delimiter
    for spec in decoder.attr_types:
      name, bucket, multi = Decoder.parse(spec)
      info.append_hash_bucket(0 if (multi or bucket is None) else int(bucket))
      info.append_type(_PYWRAP_TYPE[name])
  source.attr_info = info
  return source


class Graph(object):

  def __init__(self):
    self._node_sources, self._edge_sources = [], []
    self._node_decoders, self._edge_decoders = {}, {}
    self._topology = Topology()
    self._undirected_edges = []
    self._client = None
    self._server = None

  # -- construction ---------------------------------------------------------------
  def node(self, source, node_type, decoder, option=None, mask=Mask.NONE):
    if not isinstance(source, str):
      raise ValueError("source for node() must be string.")
    if not isinstance(node_type, str):
      raise ValueError("node_type for node() must be string.")
    if not isinstance(decoder, Decoder):
      raise ValueError("decoder must be an instance of `Decoder`, got {}".format(type(decoder)))
    stored = get_mask_type(node_type, mask)
    self._node_decoders[stored] = decoder
    for path in (p.strip() for p in source.split(",")):
      s = pywrap.NodeSource()
      s.id_type = stored
      self._node_sources.append(_fill_source(s, path, decoder, option))
    return self

  def _edge_source(self, path, types, decoder, direction, option):
    s = pywrap.EdgeSource()
    s.src_id_type, s.dst_id_type, s.edge_type = types
    s.direction = direction
    return _fill_source(s, path, decoder, option)

  def edge(self, source, edge_type, decoder=None, directed=True, option=None, mask=Mask.NONE):
    if not isinstance(source, str):
      raise ValueError("source for edge() must be a string.")
    if not isinstance(edge_type, tuple) or len(edge_type) != 3:
      raise ValueError("edge_type for edge() must be a tuple of (src_type, dst_tye, edge_type).")
    decoder = decoder or Decoder()
    if not isinstance(decoder, Decoder):
      raise ValueError("decoder must be an instance of Decoder, got {}".format(type(decoder)))
    src, dst, name = edge_type
    stored = get_mask_type(name, mask)
    self._edge_decoders[stored] = decoder
    self._topology.add(stored, src, dst)
    paths = [p.strip() for p in source.split(",")]
    for path in paths:
      self._edge_sources.append(self._edge_source(path, (src, dst, stored), decoder, pywrap.Direction.ORIGIN, option))
    if not directed:
      # graph.py:357-380: a heterogeneous edge type gets a twin `<type>_reverse` with swapped
      # end points; a homogeneous one gets the reversed records appended to the SAME type.
      self._undirected_edges.append(name)
      if src != dst:
        twin = name + "_reverse"
        self._edge_decoders[twin] = decoder
        self._topology.add(twin, dst, src)
        types = (dst, src, twin)
      else:
        types = (src, dst, name)
      for path in paths:
        self._edge_sources.append(self._edge_source(path, types, decoder, pywrap.Direction.REVERSED, option))
    return self

  def init(self, task_index=0, task_count=1, cluster="", job_name="", **kwargs):
    """Load + build on this process' GPU.  `tracker=` is accepted and ignored.

    task_count > 1 is the SPMD mode of this engine: one process per GPU (torchrun), every process
    reads the same sources and keeps shard `task_index` -- the edges whose source id, and the nodes
    whose id, satisfy llabs(id) % task_count == task_index, the rule the reference routes requests by
    (hash_partitioner.h:88-90).  The numpy request path of this object (sampler.get(), lookups) then sees the
    shard only; the view over ALL shards -- requests exchanged between the GPUs over RCCL, every rank calling at
    the same time with its own batch -- is `sharded_store()`, and on CUDA tensors `neighbor_sampler(...).get_device()`
    and `random_walk()` go through it by themselves."""
    if cluster or kwargs.get("hosts") is not None:
      raise NotImplementedError(
          "the reference's client/server RPC deploy mode is not served by this engine; multi-GPU execution is one "
          "process per GPU: init(task_index=rank, task_count=world_size) + sharded_store()")
    task_index, task_count = int(task_index), int(task_count)
    if task_count < 1 or not 0 <= task_index < task_count:
      raise ValueError("task_index {} is not in [0, task_count {})".format(task_index, task_count))
    self._shard = (task_index, task_count)
    self._client = pywrap.in_memory_client()
    self._server = pywrap.server(task_index, task_count, "", "")
    self._server.start()
    self._server.init(self._edge_sources, self._node_sources)
    errors.raise_exception_on_not_ok_status(self._server.init_status())
    return self

  def sharded_store_cached(self, edge_type, node_type=None):
    """sharded_store(edge_type, node_type) on the default process group, created once per (edge type, node
    type): a store owns a communicator, so the samplers of the Python API share it."""
    stores = self.__dict__.setdefault("_sharded_stores", {})
    key = (edge_type, node_type)
    if key not in stores:
      stores[key] = self.sharded_store(edge_type, node_type)
    return stores[key]

  def sharded_store(self, edge_type, node_type=None, group=None, replicate_features=False, hot_nodes=0):
    """The edge type (and optionally a node type's float attributes) across all shards of an
    init(task_index, task_count) job, as a dist.ShardedStore: `store.sample(sampler, cuda_ids, k, ...)`
    and `store.aggregate(op, node_ids, segment_ids, num_segments)` are collective calls -- every rank
    passes its own batch, rows are routed to their owners by llabs(id) % world over RCCL
    (HashPartitioner / Stitcher on the device, glx_dist_*) and the answers equal a single store's, draw
    for draw.  Needs an initialised torch.distributed process group whose size is task_count.
    hot_nodes=K keeps a replica of the K vertices with the largest in-degree (over all shards of
    `edge_type`) on every GPU -- their float attributes and their adjacency rows; aggregation then fetches only
    the remaining remote rows per request and sampling requests for those vertices stay local.
    replicate_features (a full copy of the table on every GPU) is not offered here: use hot_nodes."""
    import torch.distributed as torch_dist
    import dist as glx_dist
    if replicate_features:
      raise NotImplementedError("replicate_features is not offered by sharded_store(); hot_nodes=K replicates the "
                                "K hottest rows (K = the node count replicates everything that has in-edges)")
    if not torch_dist.is_initialized():
      raise RuntimeError("sharded_store needs torch.distributed.init_process_group (backend 'nccl' = RCCL)")
    world = torch_dist.get_world_size(group)
    shard = getattr(self, "_shard", (0, 1))
    if (torch_dist.get_rank(group), world) != shard:
      raise ValueError("the process group says rank {} of {}, the graph was initialised as shard {} of {}".format(
          torch_dist.get_rank(group), world, shard[0], shard[1]))
    feats = self.device_features(node_type) if node_type is not None else None
    store = glx_dist.ShardedStore(glx_dist.DeviceOps(), self.device_graph(edge_type), feats, group)
    if hot_nodes:
      hot = store.native.hot_ids(int(hot_nodes))
      if feats is not None:
        store.native.set_cache(hot)
      if hot.shape[0] > 0:
        # the same vertices' adjacency rows on every GPU (cut out of the shards, all-gathered once): their sampling
        # requests are then served locally, with the same draws
        store.graph_replica = store.native.build_graph_replica(hot)
    return store

  def add_dataset(self, ds):
    """graph.py:510-511: datasets register here (gsl.Dataset does it itself) so that close() can stop them -- with
    prefetch=True a dataset owns a background thread that keeps issuing requests."""
    if not hasattr(self, "_datasets"):
      self._datasets = []
    import weakref
    self._datasets.append(weakref.ref(ds))

  def close(self):
    for ref in getattr(self, "_datasets", []):
      ds = ref()
      if ds is not None:
        ds.close()
    self._datasets = []
    if self._client is not None:
      self._client.stop()
      self._client = None
    if self._server is not None:
      self._server.stop()
      self._server = None

  wait_for_close = close

  # -- the reference's other deployment entry points: its RPC client / server modes, the vineyard storage backend and the
  # faiss KNN operator are not part of this engine; they fail by name, not by AttributeError
  def _not_served(self, what):
    raise NotImplementedError(what + " is not served by this engine: it runs in process, one process per GPU "
                              "(init(task_index=rank, task_count=world_size) for several GPUs)")

  def deploy_in_local_mode(self, *a, **k):
    return self.init()

  def deploy_in_server_mode(self, *a, **k):
    self._not_served("the client / server deploy mode (deploy_in_server_mode)")

  def deploy_in_worker_mode(self, *a, **k):
    self._not_served("the client / server deploy mode (deploy_in_worker_mode)")

  def vineyard(self, *a, **k):
    self._not_served("the vineyard storage backend")

  def init_vineyard(self, *a, **k):
    self._not_served("the vineyard storage backend")

  def node_view(self, *a, **k):
    self._not_served("node_view (vineyard backend only in the reference too)")

  def search(self, *a, **k):
    self._not_served("the KNN operator (contrib/knn over faiss)")

  # -- introspection ----------------------------------------------------------------
  def get_client(self):
    return self._client

  def get_server(self):
    return self._server

  def get_topology(self):
    return self._topology

  def get_node_decoder(self, node_type):
    if node_type not in self._node_decoders:
      raise ValueError("node type {} not exist in graph.".format(node_type))
    return self._node_decoders[node_type]

  def get_edge_decoder(self, edge_type):
    if edge_type not in self._edge_decoders:
      raise ValueError("edge type {} not exist in graph.".format(edge_type))
    return self._edge_decoders[edge_type]

  def get_node_decoders(self):
    return self._node_decoders

  def get_edge_decoders(self):
    return self._edge_decoders

  @property
  def undirected_edges(self):
    return self._undirected_edges

  def is_directed(self, edge_type):
    self.get_edge_decoder(edge_type)
    return edge_type not in self._undirected_edges

  # -- values ---------------------------------------------------------------------
  def get_nodes(self, node_type, ids, offsets=None, shape=None):
    if offsets is None:
      return data.Nodes(ids, node_type, shape=shape, graph=self)
    width = shape[1] if shape and shape[1] and shape[1] > 0 else max(offsets, default=0)
    return data.SparseNodes(ids, offsets, (len(offsets), width), node_type, graph=self)

  def get_edges(self, edge_type, src_ids, dst_ids, edge_ids=None, offsets=None, shape=None, reverse=False):
    if reverse:
      edge_type = edge_type + "_reverse"
    src_type = self._topology.get_src_type(edge_type)
    dst_type = self._topology.get_dst_type(edge_type)
    if offsets is None:
      return data.Edges(src_ids, src_type, dst_ids, dst_type, edge_type, edge_ids, shape=shape, graph=self)
    width = shape[1] if shape and shape[1] and shape[1] > 0 else max(offsets, default=0)
    return data.SparseEdges(src_ids, src_type, dst_ids, dst_type, edge_type, offsets, (len(offsets), width),
                            edge_ids, graph=self)

  def _run_lookup(self, kind, type_name, decoder, set_request):
    """One LookupNodes / LookupEdges call -> Values with the columns the decoder declares."""
    req = getattr(pywrap, "new_lookup_%ss_request" % kind)(type_name)
    set_request(req)
    res = getattr(pywrap, "new_lookup_%ss_response" % kind)()
    status = getattr(self._client, "lookup_%ss" % kind)(req, res)
    cols = {}
    if status.ok():
      fetch = lambda what: getattr(pywrap, "get_%s_%s" % (kind, what))(res)  # noqa: E731
      cols["weights"] = fetch("weights") if decoder.weighted else None
      cols["labels"] = fetch("labels") if decoder.labeled else None
      cols["timestamps"] = fetch("timestamps") if decoder.timestamped else None
      if decoder.attributed:
        cols["int_attrs"], cols["float_attrs"], cols["string_attrs"] = decoder.format_attrs(
            fetch("int_attributes"), fetch("float_attributes"), fetch("string_attributes"))
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return data.Values(graph=self, **cols)

  def lookup_nodes(self, node_type, ids):
    ids = np.ascontiguousarray(np.array(ids).reshape(-1), dtype=np.int64)
    return self._run_lookup("node", node_type, self.get_node_decoder(node_type),
                            lambda req: pywrap.set_lookup_nodes_request(req, ids))

  def lookup_edges(self, edge_type, src_ids, edge_ids):
    src_ids = np.ascontiguousarray(np.array(src_ids).reshape(-1), dtype=np.int64)
    edge_ids = np.ascontiguousarray(np.array(edge_ids).reshape(-1), dtype=np.int64)
    if src_ids.size != edge_ids.size:
      raise ValueError("src_ids and edge_ids for lookup edges must be same, got {} and {}"
                       .format(src_ids.size, edge_ids.size))
    return self._run_lookup("edge", edge_type, self.get_edge_decoder(edge_type),
                            lambda req: pywrap.set_lookup_edges_request(req, src_ids, edge_ids))

  def get_stats(self):
    """Number of nodes / edges held per type, one entry per server (graph.py:1083-1093 in the reference):
    the "GetStats" operator through the client."""
    req = pywrap.new_get_stats_request()
    res = pywrap.new_get_stats_response()
    status = self._client.get_stats(req, res)
    stats = pywrap.get_stats(res) if status.ok() else None
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return stats

  def _degrees(self, ids, edge_type, node_from):
    ids = np.array(ids)
    req = pywrap.new_get_degree_request(edge_type, node_from)
    pywrap.set_degree_request(req, np.ascontiguousarray(ids.reshape(-1), dtype=np.int64))
    res = pywrap.new_get_degree_response()
    status = self._client.get_degree(req, res)
    out = pywrap.get_degree(res).reshape(ids.shape) if status.ok() else None
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return out

  def out_degrees(self, ids, edge_type):
    """Out-degrees of source ids of `edge_type` (0 for unknown ids)."""
    return self._degrees(ids, edge_type, pywrap.NodeFrom.EDGE_SRC)

  def in_degrees(self, ids, edge_type):
    """In-degrees of destination ids of `edge_type` (0 for ids no edge points to).  In SPMD mode (init(task_index,
    task_count)) the call is collective and the counts are sums over ALL shards (glx_dist_in_degrees)."""
    if getattr(self, "_shard", (0, 1))[1] > 1:
      import numpy as np
      arr = np.ascontiguousarray(np.asarray(ids).reshape(-1), dtype=np.int64)
      out = self.sharded_store_cached(edge_type).in_degrees(arr)
      return out.reshape(np.asarray(ids).shape)
    return self._degrees(ids, edge_type, pywrap.NodeFrom.EDGE_DST)

  def global_negative_table(self, object_type, by_in_degree=False, node_weights=False):
    """SPMD mode: the negative samplers' candidate list over the WHOLE type, the same glx.Negative on every rank
    (collective on first use): an edge type's destination ids of all shards (ascending; uniform or weighted by the
    in-degree summed over all shards), or -- node_weights=True -- a node type's ids and node weights of all shards."""
    import numpy as np
    import glx
    tables = self.__dict__.setdefault("_negative_tables", {})
    key = (object_type, bool(by_in_degree), bool(node_weights))
    if key not in tables:
      if node_weights:
        import torch.distributed as torch_dist
        mine = self._server.node_ids(object_type)
        weights = self.get_nodes(object_type, mine).weights
        parts = [None] * torch_dist.get_world_size()
        torch_dist.all_gather_object(parts, (np.asarray(mine, np.int64), np.asarray(weights, np.float32)))
        ids = np.concatenate([p[0] for p in parts])
        w = np.concatenate([p[1] for p in parts])
        order = np.argsort(ids, kind="stable")
        tables[key] = glx.Negative(np.ascontiguousarray(ids[order]), np.ascontiguousarray(w[order]),
                                   device=self.device_features(object_type).device
                                   if self._server.device_features(object_type) else 0)
      else:
        tables[key] = self.sharded_store_cached(object_type).negative_table(by_in_degree)
    return tables[key]

  # -- samplers -------------------------------------------------------------------
  def neighbor_sampler(self, meta_path, expand_factor, strategy="random"):
    """strategy: "random", "random_without_replacement", "topk", "in_degree", "edge_weight", "full"."""
    from graphlearn import sampler
    cls = getattr(sampler, "".join(w.capitalize() for w in strategy.split("_")) + "NeighborSampler", None)
    if cls is None:
      raise ValueError("unknown neighbor sampling strategy {!r}".format(strategy))
    return cls(self, meta_path, expand_factor, strategy=strategy)

  # -- device-resident access (new) ----------------------------------------------
  def device_graph(self, edge_type):
    """The edge type's CSR in HBM as a borrowed glx.Graph (torch-tensor in / out, zero copies)."""
    import glx
    h = self._server.device_graph(edge_type)
    if not h:
      raise ValueError("edge type {} is not built on the device".format(edge_type))
    return glx.Graph.from_handle(h)

  def device_features(self, node_type):
    """The node type's float attributes in HBM as a borrowed glx.Features."""
    import glx
    h = self._server.device_features(node_type)
    if not h:
      raise ValueError("node type {} has no float attributes on the device".format(node_type))
    return glx.Features.from_handle(h)

  def random_walk(self, edge_type, ids, walk_len, p=1.0, q=1.0, call_counter=None):
    """Random walks over one edge type -- the reference's "RandomWalk" operator
    (core/operator/random_walk/random_walk.cc), which its own Python API reaches through GSL's
    .random_walk() only: `walk_len` steps from every id.  p = q = 1: DeepWalk, every step one uniform
    neighbour draw.  Otherwise node2vec: a step weighs the first `default_full_nbr_num` edges of the
    current vertex by 1/p (back to the parent), 1 (to a neighbour of the parent) or 1/q and draws
    from their alias table.  A vertex without out-edges yields the default neighbour id.
    ids: numpy array -> numpy walks [len(ids), walk_len] through the operator; torch CUDA tensor ->
    CUDA walks straight from the device graph (call_counter defaults to 0 there)."""
    import torch
    from graphlearn import settings
    if isinstance(ids, torch.Tensor) and getattr(self, "_shard", (0, 1))[1] > 1:
      # SPMD mode: a collective walk over the shards (DeepWalk and node2vec; glx_dist_random_walk_ex), the single store's draws
      flags = settings._MIRROR  # pylint: disable=protected-access
      return self.sharded_store_cached(edge_type).native.random_walk(
          ids, int(walk_len), p=float(p), q=float(q), default_neighbor_id=flags["default_neighbor_id"],
          seed=flags["sampling_seed"], call_counter=call_counter or 0, full_nbr_num=flags["default_full_nbr_num"],
          default_weight=flags["default_weight"])
    if isinstance(ids, torch.Tensor):
      flags = settings._MIRROR  # pylint: disable=protected-access
      return self.device_graph(edge_type).random_walk(
          ids, int(walk_len), p=float(p), q=float(q), full_nbr_num=flags["default_full_nbr_num"],
          default_weight=flags["default_weight"], default_neighbor_id=flags["default_neighbor_id"],
          seed=flags["sampling_seed"], call_counter=call_counter or 0)
    self.get_edge_decoder(edge_type)
    src = np.ascontiguousarray(np.array(ids).reshape(-1), dtype=np.int64)
    req = pywrap.new_random_walk_request(edge_type, float(p), float(q), int(walk_len))
    pywrap.set_random_walk_request(req, src)
    if call_counter is not None:
      pywrap.set_random_walk_call_counter(req, int(call_counter))
    res = pywrap.new_random_walk_response()
    status = self._client.run_op(req, res)
    walks = pywrap.get_random_walks(res).reshape(src.size, int(walk_len)) if status.ok() else None
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return walks

  # -- GSL (python/graph.py:645-718 in the reference; steps and Dataset: gsl.py) ------------------
  def V(self, t, feed=None, node_from=pywrap.NodeFrom.NODE, mask=Mask.NONE):  # pylint: disable=invalid-name
    """Starts a query at batches of vertices: of node type `t`, or (node_from = EDGE_SRC / EDGE_DST) of the end points
    of edge type `t`.  -> a step to chain .batch() / .shuffle() / .alias() / .outV() ... on."""
    from graphlearn import gsl
    if feed is not None:
      raise NotImplementedError("feeding a query from a generator is not served: batch the ids through the sampler "
                                "objects (Graph.neighbor_sampler ...) instead")
    return gsl.VertexSource(gsl.Query(self), t, node_from=node_from, mask=mask)

  def E(self, edge_type, feed=None, reverse=False, mask=Mask.NONE):  # pylint: disable=invalid-name
    """Starts a query at batches of edges of `edge_type` (reverse=True: of its reversed twin, for undirected types)."""
    from graphlearn import gsl
    if feed is not None:
      raise NotImplementedError("feeding a query from a generator is not served")
    return gsl.EdgeSource(gsl.Query(self), edge_type + "_reverse" if reverse else edge_type, mask=mask)

  def SubGraph(self, seed_type, nbr_type, batch_size=64, strategy="random_node", num_nbrs=(0,), feed=None):  # pylint: disable=invalid-name
    """graph.py:629-671: sub-graph sampling as a GSL entry -- batches of `seed_type` vertices ("random_node" /
    "in_order_node") or edges ("random_edge" / "in_order_edge") and the sub-graph `nbr_type` induces around each batch
    (+ num_nbrs sampled neighbours per hop).  -> the SubGraph step: .alias(name).values() closes the query."""
    if feed is not None:
      raise NotImplementedError("`feed` is not supported for now.")
    if strategy not in ("random_node", "random_edge", "in_order_node", "in_order_edge"):
      raise ValueError("strategy must be random_node / random_edge / in_order_node / in_order_edge, got {!r}".format(strategy))
    source = self.V(seed_type) if strategy.endswith("node") else self.E(seed_type)
    source = source.batch(batch_size)
    if strategy.startswith("random"):
      source = source.shuffle(traverse=True)
    return source.SubGraph(nbr_type, num_nbrs=tuple(num_nbrs))

  def node_sampler(self, t, batch_size=64, strategy="by_order", node_from=pywrap.NodeFrom.NODE, mask=Mask.NONE):
    """Batches of seed vertices: strategy "by_order" | "shuffle" | "random" (see traversal.py)."""
    from graphlearn import traversal
    return traversal.NodeSampler(self, t, batch_size, strategy=strategy, node_from=node_from, mask=mask)

  def edge_sampler(self, edge_type, batch_size=64, strategy="by_order", mask=Mask.NONE):
    """Batches of seed edges: strategy "by_order" | "shuffle" | "random"."""
    from graphlearn import traversal
    return traversal.EdgeSampler(self, edge_type, batch_size, strategy=strategy, mask=mask)

  def negative_sampler(self, object_type, expand_factor, strategy="random", conditional=False, **kwargs):
    """strategy: "random", "in_degree", "soft_in_degree" (object_type = an edge type) or
    "node_weight" (object_type = a node type)."""
    from graphlearn import sampler
    if conditional:
      return sampler.ConditionalNegativeSampler(self, object_type, expand_factor, strategy=strategy, **kwargs)
    cls = getattr(sampler, "".join(w.capitalize() for w in strategy.split("_")) + "NegativeSampler", None)
    if cls is None:
      raise ValueError("unknown negative sampling strategy {!r}".format(strategy))
    return cls(self, object_type, expand_factor, strategy=strategy)

  def subgraph_sampler(self, nbr_type, num_nbrs=(0,), need_dist=False):
    """graph.py:1060-1081 in the reference (without its unused seed_type): a sampler of induced sub-graphs."""
    from graphlearn import sampler
    return sampler.SubGraphSampler(self, nbr_type, num_nbrs=num_nbrs, need_dist=need_dist)

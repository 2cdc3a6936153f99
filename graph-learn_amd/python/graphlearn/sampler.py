"""Neighbor samplers of the Python API (graphlearn/python/sampler/neighbor_sampler.py).

`get(ids)` walks the meta path hop by hop: one sampling request per hop, the hop's
output ids are the next hop's input.  Results come back as numpy arrays inside
Layers, like the reference.  `get_device(ids)` is the MI355X-first variant: all hops
run in ONE call (glx_sample_hops), frontiers never leave HBM, and the result is a list
of torch CUDA tensors.
"""
import numpy as np

from graphlearn import pywrap_graphlearn as pywrap
from graphlearn import errors
from graphlearn.utils import strategy2op
from graphlearn.values import Layer, Layers

__all__ = ["NeighborSampler", "RandomNeighborSampler", "RandomWithoutReplacementNeighborSampler",
           "EdgeWeightNeighborSampler", "TopkNeighborSampler", "InDegreeNeighborSampler", "FullNeighborSampler",
           "NegativeSampler", "RandomNegativeSampler", "InDegreeNegativeSampler", "SoftInDegreeNegativeSampler",
           "NodeWeightNegativeSampler"]


def _as_list(value, what):
  if isinstance(value, (list, tuple)):
    return list(value)
  if isinstance(value, (int, str)):
    return [value]
  raise ValueError("`%s` must be a value or a list of values." % what)


class NeighborSampler(object):

  def __init__(self, graph, meta_path, expand_factor, strategy="random"):
    self._graph = graph
    self._meta_path = _as_list(meta_path, "meta_path")
    if isinstance(expand_factor, str):
      raise ValueError("`expand_factor` must be int or list of int.")
    self._expand_factor = _as_list(expand_factor, "expand_factor")
    self._strategy = strategy
    self._op = strategy2op(strategy, "Sampler")
    topo = graph.get_topology()
    self._dst_types = [topo.get_dst_type(e) for e in self._meta_path]
    self._call_counter = None
    self._filter = (pywrap.FilterType.OPERATOR_UNSPECIFIED, pywrap.FilterField.FIELD_UNSPECIFIED)

  def set_filter(self, filter_type, filter_field):
    """Sample around neighbours whose field hits the filter value of their seed (the
    reference's op::Filter, core/operator/sampler/filter.h, which its Python API only reaches
    through GSL's .filter() / timestamped traversals): filter_type "equal" | "larger_than" |
    None, filter_field "id" | "timestamp".  get(ids, filter_values=...) then takes one int64
    value per seed id; every hop reuses a seed's value for all of its descendants, like
    Filter::FillValues (filter.cc:53-67)."""
    types = {None: pywrap.FilterType.OPERATOR_UNSPECIFIED, "equal": pywrap.FilterType.EQUAL,
             "larger_than": pywrap.FilterType.LARGER_THAN}
    fields = {None: pywrap.FilterField.FIELD_UNSPECIFIED, "id": pywrap.FilterField.ID,
              "timestamp": pywrap.FilterField.TIMESTAMP}
    if filter_type not in types or filter_field not in fields:
      raise ValueError("unknown filter ({!r}, {!r})".format(filter_type, filter_field))
    self._filter = (types[filter_type], fields[filter_field])
    return self

  def _filtered(self):
    return self._filter[0] != pywrap.FilterType.OPERATOR_UNSPECIFIED

  def _filter_values(self, ids, filter_values):
    if not self._filtered():
      if filter_values is not None:
        raise ValueError("filter_values given but no filter set: call set_filter() first")
      return None
    if filter_values is None:
      raise ValueError("the sampler has a filter: pass filter_values (one per id)")
    values = np.ascontiguousarray(np.array(filter_values).reshape(-1), dtype=np.int64)
    if values.size != np.array(ids).size:
      raise ValueError("filter_values must hold one value per id")
    return values

  def set_call_counter(self, value):
    """Pin the random stream of the next get(): hop h uses counter value + h.  Unset,
    every request draws a fresh stream from the operator's own counter."""
    self._call_counter = value

  def _check(self):
    if len(self._meta_path) != len(self._expand_factor):
      raise ValueError("The length of meta_path must be same with hop count.")

  def _sample(self, hop, src_ids, filter_values=None):
    """-> (neighbor ids, edge ids, per-row counts) of one hop, flat."""
    req = pywrap.new_sampling_request(self._meta_path[hop], self._op, int(self._expand_factor[hop]),
                                      self._filter[0], self._filter[1])
    pywrap.set_sampling_request(req, np.ascontiguousarray(src_ids.reshape(-1), dtype=np.int64))
    if filter_values is not None:
      pywrap.set_sampling_filter_values(req, filter_values)
    if self._call_counter is not None:
      pywrap.set_sampling_call_counter(req, int(self._call_counter) + hop)
    res = pywrap.new_sampling_response()
    status = self._graph.get_client().sample_neighbor(req, res)
    out = None
    if status.ok():
      out = (pywrap.get_sampling_node_ids(res), pywrap.get_sampling_edge_ids(res),
             pywrap.get_sampling_node_degrees(res))
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return out

  def get(self, ids, filter_values=None):
    """ids: 1-D int64 array.  -> Layers; layer h holds [len(previous layer), expand_factor[h]] nodes/edges."""
    self._check()
    src = np.array(ids)
    values = self._filter_values(src, filter_values)
    layers = Layers()
    for hop, edge_type in enumerate(self._meta_path):
      k = self._expand_factor[hop]
      nbr, eid, _ = self._sample(hop, src, values)
      values = None if values is None else np.repeat(values, k)
      shape = (src.size, k)
      nodes = self._graph.get_nodes(self._dst_types[hop], nbr, shape=shape)
      edges = self._graph.get_edges(edge_type, np.repeat(src.reshape(-1), k), nbr, shape=shape)
      edges.edge_ids = eid
      layers.append_layer(Layer(nodes, edges))
      src = nbr
    return layers


  def get_device(self, ids, seed=None, call_counter=0):
    """All hops in one engine call on torch CUDA tensors (glx_sample_hops): the hop-h frontier is
    hop h-1's output and never leaves HBM.  ids: int64 CUDA tensor [B].
    -> [(neighbor ids [rows_h, k_h], edge ids [rows_h, k_h]) per hop], CUDA tensors.
    Same draws as get() with set_call_counter(call_counter) under the same seed.  In SPMD mode the call is
    collective: every rank passes its own batch (torch.distributed must be initialised)."""
    import glx
    self._check()
    if self._strategy == "full":
      raise ValueError("the full sampler returns ragged rows; use get()")
    if self._filtered():
      raise ValueError("filtered sampling goes hop by hop; use get(ids, filter_values=...)")
    seed = _flag("sampling_seed") if seed is None else seed
    if getattr(self._graph, "_shard", (0, 1))[1] > 1:
      # SPMD mode (Graph.init(task_index, task_count)): every hop is a collective request to the partitioned store
      # (rows travel to their owners over RCCL, glx_dist_sample); same draws as one store, for any shard count
      out, frontier = [], ids
      for hop, edge_type in enumerate(self._meta_path):
        store = self._graph.sharded_store_cached(edge_type)
        nbr, eid = store.sample(self._op, frontier.reshape(-1), int(self._expand_factor[hop]), seed=seed,
                                call_counter=call_counter + hop, padding_mode=_flag("padding_mode"),
                                default_neighbor_id=_flag("default_neighbor_id"))
        out.append((nbr, eid))
        frontier = nbr
      return out
    graphs = [self._graph.device_graph(e) for e in self._meta_path]
    return glx.sample_hops(graphs, self._op, ids, [int(k) for k in self._expand_factor], seed=seed,
                           call_counter=call_counter, padding_mode=_flag("padding_mode"),
                           default_neighbor_id=_flag("default_neighbor_id"))


def _flag(name):
  from graphlearn import settings
  return settings._MIRROR[name]  # pylint: disable=protected-access


class RandomNeighborSampler(NeighborSampler):
  pass


class RandomWithoutReplacementNeighborSampler(NeighborSampler):
  pass


class EdgeWeightNeighborSampler(NeighborSampler):
  pass


class TopkNeighborSampler(NeighborSampler):
  pass


class InDegreeNeighborSampler(NeighborSampler):
  pass


class FullNeighborSampler(NeighborSampler):
  """All neighbours (at most expand_factor per vertex when it is > 0): ragged results,
  returned as SparseNodes / SparseEdges."""

  def get(self, ids, filter_values=None):
    self._check()
    src = np.array(ids).reshape(-1)
    values = self._filter_values(src, filter_values)
    layers = Layers()
    for hop, edge_type in enumerate(self._meta_path):
      nbr, eid, counts = self._sample(hop, src, values)
      counts = [int(c) for c in counts]
      values = None if values is None else np.repeat(values, counts)
      dense = (src.size, max(counts) if counts else 0)
      nodes = self._graph.get_nodes(self._dst_types[hop], nbr, offsets=counts, shape=dense)
      edges = self._graph.get_edges(edge_type, np.repeat(src, counts), nbr, offsets=counts, shape=dense)
      edges.edge_ids = eid
      layers.append_layer(Layer(nodes, edges))
      src = nbr
    return layers


class SubGraph(object):
  """python/data/values.py:819-843: edge_index [2, m] (positions in `nodes`), nodes, edges + free attributes."""

  def __init__(self, edge_index, nodes, edges=None, **kwargs):
    self._nodes = nodes
    self._edge_index = edge_index
    self._edges = edges
    for key, item in kwargs.items():
      setattr(self, key, item)

  @property
  def nodes(self):
    return self._nodes

  @property
  def edge_index(self):
    return self._edge_index

  @property
  def edges(self):
    return self._edges

  def __getitem__(self, key):
    return getattr(self, key, None)

  def __setitem__(self, key, value):
    setattr(self, key, value)


class SubGraphSampler(object):
  """python/sampler/subgraph_sampler.py:26-104: the seeds' `num_nbrs`-hop neighbourhood (FullSampler with a limit per
  hop) and the edges of `nbr_type` it induces; need_dist adds every node's distance to the first two nodes."""

  def __init__(self, graph, nbr_type, num_nbrs=(0,), need_dist=False):
    self._graph = graph
    self._nbr_type = nbr_type
    self._num_nbrs = [int(x) for x in num_nbrs]
    self._need_dist = bool(need_dist)
    self._node_type = graph.get_topology().get_src_type(nbr_type)

  def get(self, ids, dst_ids=None):
    ids = np.ascontiguousarray(np.array(ids).reshape(-1), dtype=np.int64)
    if dst_ids is not None:
      dst_ids = np.ascontiguousarray(np.array(dst_ids).reshape(-1), dtype=np.int64)
    req = pywrap.new_subgraph_request(self._nbr_type, self._num_nbrs, self._need_dist)
    pywrap.set_subgraph_request(req, ids, dst_ids)
    res = pywrap.new_subgraph_response()
    status = self._graph.get_client().sample_subgraph(req, res)
    out = None
    if status.ok():
      out = (pywrap.get_node_set(res), pywrap.get_row_idx(res), pywrap.get_col_idx(res), pywrap.get_edge_set(res),
             pywrap.get_dist_to_src(res) if self._need_dist else None,
             pywrap.get_dist_to_dst(res) if self._need_dist else None)
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    node_ids, row_idx, col_idx, edge_ids, to_src, to_dst = out
    nodes = self._graph.get_nodes(self._node_type, node_ids)
    sub = SubGraph(np.stack([row_idx, col_idx], axis=0), nodes, edge_ids)
    sub.dist_to_src = to_src
    sub.dist_to_dst = to_dst
    return sub


class NegativeSampler(object):
  """Negative sampling (graphlearn/python/sampler/negative_sampler.py): for every given id,
  `expand_factor` candidate destination ids.  object_type is an edge type ("random",
  "in_degree", "soft_in_degree": candidates = the type's destination ids) or a node type
  ("node_weight": candidates = the type's ids, weighted by node weight)."""

  _needs = "edge"

  def __init__(self, graph, object_type, expand_factor, strategy="random"):
    self._graph = graph
    self._object_type = object_type
    self._expand_factor = int(expand_factor)
    self._op = strategy2op(strategy, "NegativeSampler")
    self._call_counter = None
    if object_type in graph.get_node_decoders():
      kind, self._dst_type = "node", object_type
    elif object_type in graph.get_edge_decoders():
      kind, self._dst_type = "edge", graph.get_topology().get_dst_type(object_type)
    else:
      raise ValueError("node or edge type {} is not in the graph".format(object_type))
    if kind != self._needs:
      raise ValueError("{} is not type of {}.".format(object_type, self._needs))

  def set_call_counter(self, value):
    self._call_counter = value

  def _get_spmd(self, ids):
    """SPMD mode (Graph.init(task_index, task_count)): candidates = the WHOLE type's list, the same table on every rank
    (Graph.global_negative_table); strict in-degree sampling is a collective request to the owners of the source ids
    (glx_dist_negative_sample).  What an unpartitioned store with that table draws, for every shard count."""
    import glx
    kind = {"RandomNegativeSampler": (False, glx.NEG_EXCLUDE_NONE), "SoftInDegreeNegativeSampler": (True, glx.NEG_EXCLUDE_NONE),
            "InDegreeNegativeSampler": (True, glx.NEG_EXCLUDE_NEIGHBORS), "NodeWeightNegativeSampler": (None, glx.NEG_EXCLUDE_BATCH)}
    by_in_degree, exclude = kind[self._op]
    if self._call_counter is None:
      self._spmd_calls = getattr(self, "_spmd_calls", 0) + 1
    cc = int(self._call_counter) if self._call_counter is not None else self._spmd_calls
    seed, default = _flag("sampling_seed"), _flag("default_neighbor_id")
    if by_in_degree is None:  # node weights: nothing is needed from another shard once the table is global
      table = self._graph.global_negative_table(self._object_type, node_weights=True)
      out = table.sample(ids, self._expand_factor, exclude=exclude, default_neighbor_id=default, seed=seed, call_counter=cc)
    else:
      if exclude == glx.NEG_EXCLUDE_NEIGHBORS:
        self._graph.device_graph(self._object_type).enable_negative()
      table = self._graph.global_negative_table(self._object_type, by_in_degree=by_in_degree)
      out = self._graph.sharded_store_cached(self._object_type).negative_sample(
          table, ids, self._expand_factor, exclude=exclude, default_neighbor_id=default, seed=seed, call_counter=cc)
    return self._graph.get_nodes(self._dst_type, out, shape=(ids.shape[0], self._expand_factor))

  def get(self, ids):
    """-> Nodes of shape [len(ids), expand_factor]"""
    ids = np.ascontiguousarray(np.array(ids).reshape(-1), dtype=np.int64)
    if getattr(self._graph, "_shard", (0, 1))[1] > 1 and type(self) is not ConditionalNegativeSampler:  # pylint: disable=unidiomatic-typecheck
      return self._get_spmd(ids)
    req = pywrap.new_sampling_request(self._object_type, self._op, self._expand_factor,
                                      pywrap.FilterType.OPERATOR_UNSPECIFIED, pywrap.FilterField.FIELD_UNSPECIFIED)
    pywrap.set_sampling_request(req, ids)
    if self._call_counter is not None:
      pywrap.set_sampling_call_counter(req, int(self._call_counter))
    res = pywrap.new_sampling_response()
    status = self._graph.get_client().sample_neighbor(req, res)
    out = pywrap.get_sampling_node_ids(res) if status.ok() else None
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return self._graph.get_nodes(self._dst_type, out, shape=(ids.shape[0], self._expand_factor))


class ConditionalNegativeSampler(NegativeSampler):
  """python/sampler/negative_sampler.py:119-229: negatives for (src, dst) pairs that share the selected attribute
  columns with the dst.  kwargs: batch_share, unique, int_cols / int_props, float_cols / float_props, str_cols /
  str_props (column indices into the dst node type's int / float / string attributes and the share of the
  `expand_factor` slots each column fills; the rest comes from the unconditioned `strategy`)."""

  _needs = None  # an edge type ("random", "in_degree") or a node type ("node_weight")

  def __init__(self, graph, object_type, expand_factor, strategy="random", **kwargs):
    self._needs = "node" if strategy == "node_weight" else "edge"
    super(ConditionalNegativeSampler, self).__init__(graph, object_type, expand_factor, strategy=strategy)
    self._strategy = strategy
    self._batch_share = bool(kwargs.get("batch_share", False))
    self._unique = bool(kwargs.get("unique", False))
    self._cols = [[int(c) for c in (kwargs.get(k + "_cols") or [])] for k in ("int", "float", "str")]
    self._props = [[float(p) for p in (kwargs.get(k + "_props") or [])] for k in ("int", "float", "str")]
    for cols, props in zip(self._cols, self._props):
      if len(cols) != len(props):
        raise ValueError("Condition columns and props must be the same size.")
    decoder = graph.get_node_decoder(self._dst_type)
    for cols, n in zip(self._cols, (decoder.int_attr_num, decoder.float_attr_num, decoder.string_attr_num)):
      if any(not (0 <= c < n) for c in cols):
        raise ValueError("Condition columns index out of range.")
    if sum(sum(p) for p in self._props) > 1:
      raise ValueError("Condition props sum is greater than 1.")

  def get(self, src_ids, dst_ids):
    """-> Nodes of shape [len(src_ids), expand_factor]"""
    src_ids = np.ascontiguousarray(np.array(src_ids).reshape(-1), dtype=np.int64)
    dst_ids = np.ascontiguousarray(np.array(dst_ids).reshape(-1), dtype=np.int64)
    req = pywrap.new_conditional_sampling_request(self._object_type, self._strategy, self._expand_factor, self._dst_type,
                                                  self._batch_share, self._unique)
    pywrap.set_conditional_sampling_request_ids(req, src_ids, dst_ids)
    pywrap.set_conditional_sampling_request_cols(req, self._cols[0], self._props[0], self._cols[1], self._props[1],
                                                 self._cols[2], self._props[2])
    if self._call_counter is not None:
      pywrap.set_sampling_call_counter(req, int(self._call_counter))
    res = pywrap.new_sampling_response()
    status = self._graph.get_client().cond_neg_sample(req, res)
    out = pywrap.get_sampling_node_ids(res) if status.ok() else None
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return self._graph.get_nodes(self._dst_type, out, shape=(dst_ids.shape[0], self._expand_factor))


class RandomNegativeSampler(NegativeSampler):
  pass


class InDegreeNegativeSampler(NegativeSampler):
  pass


class SoftInDegreeNegativeSampler(NegativeSampler):
  pass


class NodeWeightNegativeSampler(NegativeSampler):
  _needs = "node"

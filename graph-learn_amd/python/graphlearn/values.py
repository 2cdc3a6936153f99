"""Result containers of the sampling path: Nodes, Edges, their sparse variants,
Layer and Layers.  Same attribute surface as graphlearn/python/data/values.py
(`ids`, `src_ids`, `dst_ids`, `edge_ids`, `int_attrs`, `float_attrs`,
`string_attrs`, `weights`, `labels`, `timestamps`, `shape`, `type`,
`embedding_agg`, `get_out_degrees` ...), written around one idea: a container owns
ids + a shape, and everything else is a *column* fetched from the engine on first
access (one LookupNodes / LookupEdges call) and viewed through that shape.
"""
import numpy as np

from graphlearn import pywrap_graphlearn as pywrap
from graphlearn import errors
from graphlearn.utils import strategy2op

_COLUMNS = ("int_attrs", "float_attrs", "string_attrs", "weights", "labels", "timestamps")
_PER_ELEMENT = ("weights", "labels", "timestamps")  # shape `shape`; attrs get a trailing axis


def _ro(attr):
  """A read-only attribute view (the reference spells these out as @property methods)."""
  return property(lambda self: getattr(self, attr))


def _rw(attr):
  """A read-write attribute view."""
  return property(lambda self: getattr(self, attr), lambda self, value: setattr(self, attr, value))



class Values(object):
  """Columns of a batch of nodes or edges, viewed through `shape`."""

  def __init__(self, int_attrs=None, float_attrs=None, string_attrs=None, weights=None, labels=None,
               timestamps=None, shape=None, graph=None):
    self._shape = shape
    self._graph = graph
    self._cols = {}
    given = dict(int_attrs=int_attrs, float_attrs=float_attrs, string_attrs=string_attrs, weights=weights,
                 labels=labels, timestamps=timestamps)
    self._fetched = any(v is not None for v in given.values())
    for name, v in given.items():
      self._store(name, v)

  # -- shape handling ---------------------------------------------------------
  def _view(self, value, trailing_axis=False):
    if value is None or not isinstance(value, np.ndarray) or value.size == 0 or not self._shape:
      return value
    if not isinstance(self._shape, tuple):
      raise ValueError("shape must be a tuple, got {}.".format(self._shape))
    return value.reshape(self._shape + (-1,)) if trailing_axis else value.reshape(self._shape)

  def _store(self, name, value):
    self._cols[name] = self._view(value, trailing_axis=name not in _PER_ELEMENT)

  # -- lazy fetch -------------------------------------------------------------
  def _get_decoder(self):
    raise NotImplementedError

  def _lookup(self):
    raise NotImplementedError

  def _column(self, name):
    if not self._fetched:
      self._fetched = True
      decoder = None
      try:
        decoder = self._get_decoder()
      except (AttributeError, KeyError, ValueError):
        pass
      if decoder is not None and decoder.has_property:
        got = self._lookup()
        for col in _COLUMNS:
          self._store(col, getattr(got, col))
    return self._cols.get(name)

  shape = _rw('_shape')

  graph = _rw('_graph')


def _column_property(name):
  def getter(self):
    return self._column(name)

  def setter(self, value):
    self._store(name, value)
  return property(getter, setter)


for _name in _COLUMNS:
  setattr(Values, _name, _column_property(_name))


class _Ragged(object):
  """Row structure of a sparse (ragged 2-D) result: `offsets[i]` values on row i."""

  def __init__(self, offsets, dense_shape):
    self._offsets = [int(x) for x in offsets]
    self._dense_shape = dense_shape
    self._starts = np.concatenate([[0], np.cumsum(self._offsets)]).astype(np.int64)
    self._row = 0

  offsets = _ro('_offsets')

  dense_shape = _ro('_dense_shape')

  @property
  def indices(self):
    """[row, position-in-row] of every value (a COO index list)."""
    return [[r, c] for r, n in enumerate(self._offsets) for c in range(n)]

  def _row_slice(self):
    if self._row >= len(self._offsets):
      raise StopIteration
    lo, hi = int(self._starts[self._row]), int(self._starts[self._row + 1])
    self._row += 1
    return lo, hi

  def __iter__(self):
    return self

  def next(self):
    return self.__next__()


def _resolve_shape(ids, shape):
  """The reference's rule (values.py:268-276): keep `shape` if it fits the ids,
  otherwise keep its trailing dimension and infer the leading one."""
  if shape is None:
    return ids.shape
  if int(np.prod(ids.shape)) == int(np.prod(shape)):
    return tuple(shape)
  if len(shape) == 1:
    return ids.shape
  return (int(np.prod(ids.shape) // np.prod(shape[1:])),) + tuple(shape[-1:])


class Nodes(Values):
  """A batch of vertices of one type."""

  def __init__(self, ids, node_type, int_attrs=None, float_attrs=None, string_attrs=None, weights=None,
               labels=None, timestamps=None, shape=None, graph=None):
    ids = np.asarray(ids)
    super(Nodes, self).__init__(int_attrs, float_attrs, string_attrs, weights, labels, timestamps,
                                _resolve_shape(ids, shape), graph)
    self._ids = self._view(ids)
    self._type = node_type
    self._degrees = {"in": {}, "out": {}}

  def _get_decoder(self):
    return self._graph.get_node_decoder(self._type)

  def _lookup(self):
    return self._graph.lookup_nodes(self._type, self._ids)

  ids = _ro('_ids')

  @ids.setter
  def ids(self, ids):
    self._ids = self._view(ids)

  type = _ro('_type')

  @type.setter
  def type(self, node_type):  # pylint: disable=redefined-builtin
    self._type = node_type

  # -- degrees ------------------------------------------------------------------
  def _degree(self, way, edge_type):
    topo = self._graph.get_topology()
    mine = topo.get_src_type(edge_type) if way == "out" else topo.get_dst_type(edge_type)
    if mine != self._type:
      raise ValueError("Nodes {} has no {} edge with type {}".format(self._type, way, edge_type))
    cache = self._degrees[way]
    if edge_type not in cache:
      fetch = self._graph.out_degrees if way == "out" else self._graph.in_degrees
      cache[edge_type] = fetch(self._ids, edge_type)
    return cache[edge_type]

  def get_out_degrees(self, edge_type):
    return self._degree("out", edge_type)

  def get_in_degrees(self, edge_type):
    return self._degree("in", edge_type)

  def add_out_degrees(self, edge_type, degrees):
    self._degrees["out"][edge_type] = degrees

  def add_in_degrees(self, edge_type, degrees):
    self._degrees["in"][edge_type] = degrees

  @property
  def out_degrees(self):
    return self._degrees["out"] or None

  @property
  def in_degrees(self):
    return self._degrees["in"] or None

  # -- aggregation --------------------------------------------------------------
  def _agg(self, func, segment_ids, num_segments):
    """One Sum/Mean/Max/Min/Prod aggregator call over these ids' float attributes."""
    req = pywrap.new_aggregating_request(self._type, strategy2op(func, "Aggregator"))
    pywrap.set_aggregating_request(req, np.ascontiguousarray(self._ids, np.int64).reshape(-1),
                                   np.asarray(segment_ids, dtype=np.int32), int(num_segments))
    res = pywrap.new_aggregating_response()
    status = self._graph.get_client().agg_nodes(req, res)
    out = pywrap.get_aggregating_nodes(res) if status.ok() else None
    pywrap.del_op_response(res)
    pywrap.del_op_request(req)
    errors.raise_exception_on_not_ok_status(status)
    return out

  def embedding_agg(self, func="sum"):
    """[batch, k] nodes -> [batch, float_attr_num]: reduce each row's k embeddings."""
    if len(self.shape) != 2:
      raise ValueError("embedding_agg is for Nodes with 2 dimension, and the default aggregated dimension is axis=1")
    rows, k = self.shape
    out = self._agg(func, np.repeat(np.arange(rows, dtype=np.int32), k), rows)
    return out.reshape(rows, self._get_decoder().float_attr_num)


class DeviceNodes(object):
  """A batch of vertices of one type whose ids -- and everything derived from them -- stay on the GPU: what
  gl.Dataset(query, fuse_hops=True, device=True) yields for the hops of a fused chain (the reference keeps a query's
  values in server-side tapes until the client asks, dag_node.py:183-210; here they never leave HBM until the caller
  asks with .to_host()).  Torch CUDA tensors throughout:
    ids [rows, k] int64, shape, type;
    float_attrs [rows, k, D] float32 (one glx_lookup on the device, on first access);
    embedding_agg(func) [rows, D] (one device aggregation over each row's k vertices -- Nodes.embedding_agg's result);
    to_host() -> the ordinary Nodes value (one copy)."""

  def __init__(self, ids, node_type, graph):
    self._ids, self._type, self._graph = ids, node_type, graph
    self._float_attrs = None

  ids = property(lambda self: self._ids)
  type = property(lambda self: self._type)
  shape = property(lambda self: tuple(self._ids.shape))

  def _features(self):
    return self._graph.device_features(self._type)

  @property
  def float_attrs(self):
    if self._float_attrs is None:
      from graphlearn import settings
      default = float(settings._MIRROR.get("default_float_attr", 0.0))  # pylint: disable=protected-access
      flat = self._features().lookup(self._ids.reshape(-1), default)
      self._float_attrs = flat.reshape(self.shape + (-1,))
    return self._float_attrs

  def embedding_agg(self, func="sum"):
    if len(self.shape) != 2:
      raise ValueError("embedding_agg is for Nodes with 2 dimension, and the default aggregated dimension is axis=1")
    from graphlearn import settings
    default = float(settings._MIRROR.get("default_float_attr", 0.0))  # pylint: disable=protected-access
    rows = self.shape[0]
    emb, _ = self._features().aggregate(strategy2op(func, "Aggregator"), self._ids.reshape(-1), None, rows,
                                        default_attr=default)
    return emb

  def to_host(self):
    return self._graph.get_nodes(self._type, self._ids.cpu().numpy(), shape=self.shape)


class SparseNodes(Nodes, _Ragged):
  """Ragged 2-D Nodes (FullNeighborSampler): `ids` is flat, row i owns `offsets[i]` of them."""

  def __init__(self, ids, offsets, dense_shape, node_type, int_attrs=None, float_attrs=None,
               string_attrs=None, weights=None, labels=None, timestamps=None, graph=None):
    Nodes.__init__(self, ids, node_type, int_attrs, float_attrs, string_attrs, weights, labels, timestamps,
                   None, graph)
    _Ragged.__init__(self, offsets, dense_shape)
    if np.asarray(ids).shape[0] != sum(self._offsets):
      raise ValueError("Ids must be the same length of indices")

  def __next__(self):
    lo, hi = self._row_slice()
    cut = lambda a: None if a is None else a[lo:hi]  # noqa: E731
    return Nodes(self._ids[lo:hi], self._type, graph=self._graph, int_attrs=cut(self.int_attrs),
                 float_attrs=cut(self.float_attrs), string_attrs=cut(self.string_attrs),
                 weights=cut(self.weights), labels=cut(self.labels), timestamps=cut(self.timestamps))

  def embedding_agg(self, func="sum"):
    rows = len(self._offsets)
    out = self._agg(func, np.repeat(np.arange(rows, dtype=np.int32), self._offsets), rows)
    return out.reshape(rows, self._get_decoder().float_attr_num)


class Edges(Values):
  """A batch of edges of one type."""

  def __init__(self, src_ids=None, src_type=None, dst_ids=None, dst_type=None, edge_type=None, edge_ids=None,
               src_nodes=None, dst_nodes=None, int_attrs=None, float_attrs=None, string_attrs=None,
               weights=None, labels=None, timestamps=None, shape=None, graph=None):
    if src_nodes is not None:
      src_ids, src_type = src_nodes.ids, src_nodes.type
    if dst_nodes is not None:
      dst_ids, dst_type = dst_nodes.ids, dst_nodes.type
    src_ids = np.asarray(src_ids)
    super(Edges, self).__init__(int_attrs, float_attrs, string_attrs, weights, labels, timestamps,
                                _resolve_shape(src_ids, shape), graph)
    self._src_ids = self._view(src_ids)
    self._dst_ids = self._view(np.asarray(dst_ids))
    self._edge_ids = self._view(None if edge_ids is None else np.asarray(edge_ids))
    self._src_type, self._dst_type, self._edge_type = src_type, dst_type, edge_type
    self._src_nodes, self._dst_nodes = src_nodes, dst_nodes

  def _get_decoder(self):
    return self._graph.get_edge_decoder(self._edge_type)

  def _lookup(self):
    return self._graph.lookup_edges(self._edge_type, self._src_ids, self._edge_ids)

  @property
  def src_nodes(self):
    if self._src_nodes is None:
      self._src_nodes = self._graph.get_nodes(self._src_type, self._src_ids, shape=self._shape)
    return self._src_nodes

  @src_nodes.setter
  def src_nodes(self, nodes):
    self._src_nodes = nodes

  @property
  def dst_nodes(self):
    if self._dst_nodes is None:
      self._dst_nodes = self._graph.get_nodes(self._dst_type, self._dst_ids, shape=self._shape)
    return self._dst_nodes

  @dst_nodes.setter
  def dst_nodes(self, nodes):
    self._dst_nodes = nodes

  src_ids = property(lambda self: self._src_ids)
  dst_ids = property(lambda self: self._dst_ids)
  src_type = property(lambda self: self._src_type)
  dst_type = property(lambda self: self._dst_type)
  edge_type = property(lambda self: self._edge_type)

  edge_ids = _ro('_edge_ids')

  @edge_ids.setter
  def edge_ids(self, edge_ids):
    self._edge_ids = self._view(np.asarray(edge_ids))

  @property
  def type(self):  # pylint: disable=redefined-builtin
    return (self._src_type, self._dst_type, self._edge_type)


class SparseEdges(Edges, _Ragged):
  """Ragged 2-D Edges (FullNeighborSampler)."""

  def __init__(self, src_ids, src_type, dst_ids, dst_type, edge_type, offsets, dense_shape, edge_ids=None,
               int_attrs=None, float_attrs=None, string_attrs=None, weights=None, labels=None, timestamps=None,
               graph=None):
    Edges.__init__(self, src_ids, src_type, dst_ids, dst_type, edge_type, edge_ids, None, None, int_attrs,
                   float_attrs, string_attrs, weights, labels, timestamps, None, graph)
    _Ragged.__init__(self, offsets, dense_shape)

  def __next__(self):
    lo, hi = self._row_slice()
    cut = lambda a: None if a is None else a[lo:hi]  # noqa: E731
    return Edges(self._src_ids[lo:hi], self._src_type, self._dst_ids[lo:hi], self._dst_type, self._edge_type,
                 cut(self._edge_ids), graph=self._graph, int_attrs=cut(self.int_attrs),
                 float_attrs=cut(self.float_attrs), string_attrs=cut(self.string_attrs),
                 weights=cut(self.weights), labels=cut(self.labels), timestamps=cut(self.timestamps))


class Layer(object):
  """One hop of a neighbor sample: the reached nodes and the traversed edges."""

  def __init__(self, nodes, edges=None, shape=None):
    self.nodes = nodes
    self.edges = edges
    self.shape = shape if shape is not None else getattr(nodes, "shape", None)


class Layers(object):
  """All hops of a neighbor sample; layer ids are 1-based (hop number)."""

  def __init__(self, layers=None):
    self.layers = list(layers) if layers else []

  def _at(self, layer_id):
    layer_id -= 1
    if not 0 <= layer_id < len(self.layers):
      raise ValueError("layer id beyond the layers length.")
    return self.layers[layer_id]

  def layer(self, layer_id):
    return self._at(layer_id)

  def layer_size(self, layer_id):
    return self._at(layer_id).shape

  def layer_nodes(self, layer_id):
    return self._at(layer_id).nodes

  def layer_edges(self, layer_id):
    return self._at(layer_id).edges

  def set_layer_nodes(self, layer_id, nodes):
    self._at(layer_id).nodes = nodes

  def set_layer_edges(self, layer_id, edges):
    self._at(layer_id).edges = edges

  def append_layer(self, layer):
    self.layers.append(layer)

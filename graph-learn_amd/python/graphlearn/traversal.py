"""Batch traversal samplers: g.node_sampler(...) / g.edge_sampler(...)
(graphlearn/python/sampler/{node,edge}_sampler.py over the GetNodes / GetEdges operators,
core/operator/graph/node_getter.cc:62-91, node_generator.h).

They iterate the store's id lists -- a node type's ids in insertion order, an edge type's
edges in edge-id order, or an edge type's distinct source / destination ids in
first-appearance order (GetAllSrcIds / GetAllDstIds) -- in one of three ways:
  by_order  consecutive batches; the last batch of an epoch may be short; the call after
            it raises gl.OutOfRangeError and the next epoch starts
  shuffle   the same over a fresh permutation every epoch
  random    independent uniform draws, never out of range
Samplers over the same (type, node_from) share one cursor, like the reference's server-side
state map.  This is host-side bookkeeping (seed selection for the device samplers): the id
lists are fetched from the engine once and kept as numpy arrays.
"""
import numpy as np

from graphlearn import pywrap_graphlearn as pywrap
from graphlearn import errors
from graphlearn.utils import Mask, get_mask_type

__all__ = ["NodeSampler", "RandomNodeSampler", "ByOrderNodeSampler", "ShuffleNodeSampler", "EdgeSampler",
           "RandomEdgeSampler", "ByOrderEdgeSampler", "ShuffleEdgeSampler"]


def _first_appearance(ids):
  _, first = np.unique(ids, return_index=True)
  return ids[np.sort(first)]


class _Cursor(object):
  """Shared epoch state of one id list (node_generator.h State/StateMap)."""

  def __init__(self, size, seed):
    self.size = size
    self.at = 0
    self.epoch = 0
    self.rng = np.random.default_rng(seed)
    self.perm = None

  def take(self, batch_size, strategy):
    """-> positions into the id list"""
    if strategy == "random":
      return self.rng.integers(0, self.size, batch_size)
    if strategy == "shuffle" and self.perm is None:
      self.perm = self.rng.permutation(self.size)
    if self.at >= self.size:  # nothing left: begin the next epoch, report the boundary
      self.at = 0
      self.epoch += 1
      self.perm = None
      raise errors.OutOfRangeError("No more nodes exist.", pywrap.ErrorCode.OUT_OF_RANGE)
    pos = np.arange(self.at, min(self.at + batch_size, self.size))
    self.at += pos.size
    return self.perm[pos] if strategy == "shuffle" else pos


class _Traversal(object):
  _STRATEGIES = ("by_order", "random", "shuffle")

  def __init__(self, graph, batch_size, strategy):
    if strategy not in self._STRATEGIES:
      raise ValueError("strategy must be one of {}".format(self._STRATEGIES))
    self._graph = graph
    self._batch_size = int(batch_size)
    self._strategy = strategy

  def _cursor(self, key, size):
    states = self._graph.__dict__.setdefault("_traversal_state", {})
    full = key + (self._strategy == "shuffle",)
    if full not in states:
      from graphlearn import settings
      states[full] = _Cursor(size, settings._MIRROR["sampling_seed"] + len(states))  # pylint: disable=protected-access
    return states[full]


class NodeSampler(_Traversal):

  def __init__(self, graph, t, batch_size, strategy="by_order", node_from=pywrap.NodeFrom.NODE, mask=Mask.NONE):
    super(NodeSampler, self).__init__(graph, batch_size, strategy)
    self._node_from = node_from
    stored = get_mask_type(t, mask)
    server = graph.get_server()
    if node_from == pywrap.NodeFrom.NODE:
      if stored not in graph.get_node_decoders():
        raise ValueError("Graph has no node type of {}".format(stored))
      self._node_type = t
      self._key = ("node", stored, int(node_from))
      self._load = lambda: server.node_ids(stored)
    else:
      topo = graph.get_topology()
      src_type, dst_type = topo.get_src_type(stored), topo.get_dst_type(stored)
      from_src = node_from == pywrap.NodeFrom.EDGE_SRC
      self._node_type = src_type if from_src else dst_type
      self._key = ("edge", stored, int(node_from))
      self._load = lambda: _first_appearance(server.edge_src_ids(stored) if from_src else server.edge_dst_ids(stored))
    self._ids = None

  def get(self):
    """-> Nodes of shape [batch_size] (shorter at the end of an epoch)"""
    if self._ids is None:
      cache = self._graph.__dict__.setdefault("_traversal_ids", {})
      if self._key not in cache:
        cache[self._key] = self._load()
      self._ids = cache[self._key]
    pos = self._cursor(self._key, self._ids.shape[0]).take(self._batch_size, self._strategy)
    return self._graph.get_nodes(self._node_type, self._ids[pos])


class RandomNodeSampler(NodeSampler):
  pass


class ByOrderNodeSampler(NodeSampler):
  pass


class ShuffleNodeSampler(NodeSampler):
  pass


class EdgeSampler(_Traversal):

  def __init__(self, graph, edge_type, batch_size, strategy="by_order", mask=Mask.NONE):
    super(EdgeSampler, self).__init__(graph, batch_size, strategy)
    self._edge_type = edge_type
    self._stored = get_mask_type(edge_type, mask)
    if self._stored not in graph.get_edge_decoders():
      raise ValueError("Graph has no edge type of {}".format(self._stored))
    self._lists = None

  def get(self):
    """-> Edges of shape [batch_size]; edge ids are positions in load order"""
    if self._lists is None:
      cache = self._graph.__dict__.setdefault("_traversal_ids", {})
      key = ("edges", self._stored)
      if key not in cache:
        server = self._graph.get_server()
        cache[key] = (server.edge_src_ids(self._stored), server.edge_dst_ids(self._stored))
      self._lists = cache[key]
    src, dst = self._lists
    pos = self._cursor(("edges", self._stored, 0), src.shape[0]).take(self._batch_size, self._strategy)
    edges = self._graph.get_edges(self._stored, src[pos], dst[pos])
    edges.edge_ids = np.asarray(pos, dtype=np.int64)
    return edges


class RandomEdgeSampler(EdgeSampler):
  pass


class ByOrderEdgeSampler(EdgeSampler):
  pass


class ShuffleEdgeSampler(EdgeSampler):
  pass

"""Batch traversal samplers: g.node_sampler(...) / g.edge_sampler(...)
(graphlearn/python/sampler/{node,edge}_sampler.py over the "GetNodes" / "GetEdges" operators,
core/operator/graph/node_getter.cc:62-91, node_generator.h).

They walk the store's id lists -- a node type's ids in insertion order, an edge type's edges
in edge-id order, or an edge type's distinct source / destination ids in first-appearance
order -- in one of three ways:
  by_order  consecutive batches; the last batch of an epoch may be short; the call after it
            raises gl.OutOfRangeError and the next epoch starts
  shuffle   the same over a fresh permutation every epoch
  random    independent uniform draws, never out of range
The cursor lives with the engine's operator, one per (type, node_from), so samplers over the
same type share it, like the reference's server-side state; the Graph object remembers the
epoch it has seen per type (graph.py node_state / edge_state in the reference).  This is
host-side bookkeeping: seed selection for the device samplers.
"""
from graphlearn import pywrap_graphlearn as pywrap
from graphlearn import errors
from graphlearn.utils import Mask, get_mask_type

__all__ = ["NodeSampler", "RandomNodeSampler", "ByOrderNodeSampler", "ShuffleNodeSampler", "EdgeSampler",
           "RandomEdgeSampler", "ByOrderEdgeSampler", "ShuffleEdgeSampler"]

_STRATEGIES = ("by_order", "random", "shuffle")


def _epochs(graph):
  return graph.__dict__.setdefault("_traversal_epochs", {})


def _run(graph, key, make_request, new_response, call, read):
  """One GetNodes / GetEdges call with the caller-side epoch protocol."""
  epochs = _epochs(graph)
  req = make_request(epochs.get(key, 0))
  res = new_response()
  status = call(req, res)
  out = read(res) if status.ok() else None
  if not status.ok() and status.code() == pywrap.ErrorCode.OUT_OF_RANGE:
    epochs[key] = epochs.get(key, 0) + 1  # the engine has moved on to the next epoch
  pywrap.del_op_response(res)
  pywrap.del_op_request(req)
  errors.raise_exception_on_not_ok_status(status)
  return out


class NodeSampler(object):

  def __init__(self, graph, t, batch_size, strategy="by_order", node_from=pywrap.NodeFrom.NODE, mask=Mask.NONE):
    if strategy not in _STRATEGIES:
      raise ValueError("strategy must be one of {}".format(_STRATEGIES))
    self._graph = graph
    self._batch_size = int(batch_size)
    self._strategy = strategy
    self._node_from = node_from
    self._stored = get_mask_type(t, mask)
    if node_from == pywrap.NodeFrom.NODE:
      if self._stored not in graph.get_node_decoders():
        raise ValueError("Graph has no node type of {}".format(self._stored))
      self._node_type = t
    else:
      topo = graph.get_topology()
      src_type, dst_type = topo.get_src_type(self._stored), topo.get_dst_type(self._stored)
      self._node_type = src_type if node_from == pywrap.NodeFrom.EDGE_SRC else dst_type

  def get(self):
    """-> Nodes of shape [batch_size] (shorter at the end of an epoch)"""
    client = self._graph.get_client()
    ids = _run(self._graph, ("nodes", self._stored, int(self._node_from), self._strategy == "shuffle"),
               lambda epoch: pywrap.new_get_nodes_request(self._stored, self._strategy, self._node_from,
                                                          self._batch_size, epoch),
               pywrap.new_get_nodes_response, client.get_nodes, pywrap.get_node_ids)
    return self._graph.get_nodes(self._node_type, ids)


class RandomNodeSampler(NodeSampler):
  pass


class ByOrderNodeSampler(NodeSampler):
  pass


class ShuffleNodeSampler(NodeSampler):
  pass


class EdgeSampler(object):

  def __init__(self, graph, edge_type, batch_size, strategy="by_order", mask=Mask.NONE):
    if strategy not in _STRATEGIES:
      raise ValueError("strategy must be one of {}".format(_STRATEGIES))
    self._graph = graph
    self._batch_size = int(batch_size)
    self._strategy = strategy
    self._stored = get_mask_type(edge_type, mask)
    if self._stored not in graph.get_edge_decoders():
      raise ValueError("Graph has no edge type of {}".format(self._stored))

  def get(self):
    """-> Edges of shape [batch_size]; edge ids are positions in load order"""
    client = self._graph.get_client()
    src, dst, eid = _run(self._graph, ("edges", self._stored, self._strategy == "shuffle"),
                         lambda epoch: pywrap.new_get_edges_request(self._stored, self._strategy, self._batch_size, epoch),
                         pywrap.new_get_edges_response, client.get_edges,
                         lambda res: (pywrap.get_edge_src_id(res), pywrap.get_edge_dst_id(res), pywrap.get_edge_id(res)))
    edges = self._graph.get_edges(self._stored, src, dst)
    edges.edge_ids = eid
    return edges


class RandomEdgeSampler(EdgeSampler):
  pass


class ByOrderEdgeSampler(EdgeSampler):
  pass


class ShuffleEdgeSampler(EdgeSampler):
  pass

"""SubGraph / HeteroSubGraph (graphlearn/python/nn/{subgraph,hetero_subgraph}.py): what an induce function hands a
model -- edge_index [2, m] into `nodes` (a Data or an id array) and optional edges, homogeneous or per type; both can
carry any further attribute (subgraph["y"] = labels)."""
from graphlearn.nn.data import Data


class _Attrs(object):

  @property
  def keys(self):
    return [k for k in self.__dict__ if self[k] is not None and not (k[:2] == "__" and k[-2:] == "__")]

  def __getitem__(self, key):
    return getattr(self, key, None)

  def __setitem__(self, key, value):
    setattr(self, key, value)


def _count(nodes):
  return nodes.ids.size if isinstance(nodes, Data) else nodes.size


class SubGraph(_Attrs):

  def __init__(self, edge_index, nodes, edges=None, **kwargs):
    self._edge_index, self._nodes, self._edges = edge_index, nodes, edges
    for key, item in kwargs.items():
      self[key] = item

  num_nodes = property(lambda self: _count(self._nodes))
  num_edges = property(lambda self: self._edge_index.shape[1])
  nodes = property(lambda self: self._nodes)
  edge_index = property(lambda self: self._edge_index)
  edges = property(lambda self: self._edges)


class HeteroSubGraph(_Attrs):
  """edge_index_dict: (src_type, edge_type, dst_type) -> [2, m]; nodes_dict: node type -> Data / ids."""

  def __init__(self, edge_index_dict, nodes_dict, edges_dict=None, **kwargs):
    self._edge_index_dict, self._nodes_dict, self._edges_dict = edge_index_dict, nodes_dict, edges_dict
    for key, item in kwargs.items():
      self[key] = item

  def num_nodes(self, node_type):
    return _count(self._nodes_dict[node_type])

  def num_edges(self, edge_type):
    return self._edge_index_dict[edge_type].shape[1]

  nodes_dict = property(lambda self: self._nodes_dict)
  edge_index_dict = property(lambda self: self._edge_index_dict)
  edges_dict = property(lambda self: self._edges_dict)
  node_types = property(lambda self: list(self._nodes_dict.keys()))
  edge_types = property(lambda self: list(self._edge_index_dict.keys()))

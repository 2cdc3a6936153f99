"""Edge type -> (source node type, destination node type)
(graphlearn/python/data/topology.py)."""
import warnings


class EdgeInfo(object):

  def __init__(self, src_type, dst_type):
    self._src_type, self._dst_type = src_type, dst_type

  src_type = property(lambda self: self._src_type)
  dst_type = property(lambda self: self._dst_type)


class Topology(object):

  def __init__(self):
    self._ends = {}

  def add(self, edge_type, src_type, dst_type):
    """topology.py:27-30 refuses a second declaration of an edge type; here the same (src, dst) again is accepted --
    Graph.edge() may be called once per source file of one type -- a DIFFERENT pair is the error."""
    had = self._ends.get(edge_type)
    if had is not None and (had.src_type, had.dst_type) != (src_type, dst_type):
      raise ValueError("edge_type {} has existed.".format(edge_type))
    self._ends[edge_type] = EdgeInfo(src_type, dst_type)

  def get_edge_info(self, edge_type):
    if edge_type not in self._ends:
      raise ValueError("edge type {} not exist in graph.".format(edge_type))
    return self._ends[edge_type]

  _of = get_edge_info

  def get_src_type(self, edge_type):
    return self.get_edge_info(edge_type).src_type

  def get_dst_type(self, edge_type):
    return self.get_edge_info(edge_type).dst_type

  def is_exist(self, edge_type):
    return edge_type in self._ends

  def print_all(self):
    for edge_type, info in self._ends.items():
      print("edge_type:{}, src_type:{}, dst_type:{}\n".format(edge_type, info.src_type, info.dst_type))

  def print_one(self, edge_type):
    info = self._ends.get(edge_type)
    if info is None:
      warnings.warn("edge_type {} not exists in the graph.".format(edge_type))
      return
    print("edge_type:{}, src_type:{}, dst_type:{}\n".format(edge_type, info.src_type, info.dst_type))

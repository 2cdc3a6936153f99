"""Edge type -> (source node type, destination node type)
(graphlearn/python/data/topology.py)."""


class Topology(object):

  def __init__(self):
    self._ends = {}

  def add(self, edge_type, src_type, dst_type):
    self._ends[edge_type] = (src_type, dst_type)

  def _of(self, edge_type):
    if edge_type not in self._ends:
      raise ValueError("edge type {} not exist in graph.".format(edge_type))
    return self._ends[edge_type]

  def get_src_type(self, edge_type):
    return self._of(edge_type)[0]

  def get_dst_type(self, edge_type):
    return self._of(edge_type)[1]

  def is_exist(self, edge_type):
    return edge_type in self._ends

  def print_all(self):
    for edge_type, (src, dst) in self._ends.items():
      print("edge_type:{}, src_type:{}, dst_type:{}\n".format(edge_type, src, dst))

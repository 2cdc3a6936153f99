"""Data (graphlearn/python/nn/data.py:20-77): a batch of Nodes or Edges as plain attributes --
ids [N], int_attrs / float_attrs / string_attrs [N, num], labels / weights / timestamps [N], dst_ids [N] for edges,
offsets / indices / dense_shape for the ragged results of a full-neighbour hop -- numpy arrays, or whatever apply()
turned them into (torch tensors in graphlearn.nn.pytorch)."""


class Data(object):

  def __init__(self, ids=None, ints=None, floats=None, strings=None, labels=None, weights=None, timestamps=None,
               dst_ids=None, offsets=None, indices=None, dense_shape=None, **kwargs):
    self.ids = ids
    self.int_attrs = ints
    self.float_attrs = floats
    self.string_attrs = strings
    self.labels = labels
    self.weights = weights
    self.timestamps = timestamps
    self.dst_ids = dst_ids
    self.offsets = offsets
    self.indices = indices
    self.dense_shape = dense_shape
    for key, item in kwargs.items():
      self[key] = item

  def apply(self, func):
    """func over every attribute that is set (data.py:62-68)."""
    for k, v in list(self.__dict__.items()):
      if v is not None and not k.startswith("_"):
        self.__dict__[k] = func(v)
    return self

  def __getitem__(self, key):
    return getattr(self, key, None)

  def __setitem__(self, key, value):
    setattr(self, key, value)

  def keys(self):
    return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("_")]

  def __repr__(self):
    return "Data({})".format(", ".join("{}={}".format(k, getattr(getattr(self, k), "shape", "...")) for k in self.keys()))

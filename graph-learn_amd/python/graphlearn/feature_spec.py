"""FeatureSpec (graphlearn/python/data/feature_spec.py): how a model is to encode the attributes a Decoder describes --
per attribute one spec, in the int / float / string list it ends up in: a dense value, an id to embed (a fixed number
of buckets, or a dynamic vocabulary), or a multi-valued string.  `dimension` = the width of the encoded feature."""


class SparseSpec(object):
  def __init__(self, bucket_size, dimension, need_hash):
    self.bucket_size, self.dimension, self.need_hash = bucket_size, dimension, need_hash


class DynamicSparseSpec(object):
  def __init__(self, dimension, need_hash):
    self.dimension, self.need_hash = dimension, need_hash


class DenseSpec(object):
  pass


class MultivalSpec(object):
  def __init__(self, bucket_size, dimension, delimiter):
    self.bucket_size, self.dimension, self.delimiter = bucket_size, dimension, delimiter


class DynamicMultivalSpec(object):
  def __init__(self, dimension, delimiter):
    self.dimension, self.delimiter = dimension, delimiter


class FeatureSpec(object):

  def __init__(self, feature_num, weighted=False, labeled=False, timestamped=False):
    self._feature_num = feature_num
    self._weighted, self._labeled, self._timestamped = weighted, labeled, timestamped
    self._total_dim = 0
    self._int_spec_list, self._float_spec_list, self._string_spec_list = [], [], []

  weighted = property(lambda self: self._weighted)
  labeled = property(lambda self: self._labeled)
  timestamped = property(lambda self: self._timestamped)
  int_specs = property(lambda self: self._int_spec_list)
  float_specs = property(lambda self: self._float_spec_list)
  string_specs = property(lambda self: self._string_spec_list)
  dimension = property(lambda self: self._total_dim)

  def append_sparse(self, bucket_size, dimension, need_hash=False):
    if bucket_size is not None:
      self._int_spec_list.append(SparseSpec(bucket_size, dimension, need_hash))
    elif need_hash:
      self._int_spec_list.append(DynamicSparseSpec(dimension, need_hash))
    else:
      self._string_spec_list.append(DynamicSparseSpec(dimension, need_hash))
    self._total_dim += dimension

  def append_dense(self, is_float=True):
    (self._float_spec_list if is_float else self._int_spec_list).append(DenseSpec())
    self._total_dim += 1

  def append_multival(self, bucket_size, dimension, delimiter=","):
    if bucket_size is not None:
      self._string_spec_list.append(MultivalSpec(bucket_size, dimension, delimiter))
    else:
      self._string_spec_list.append(DynamicMultivalSpec(dimension, delimiter))
    self._total_dim += dimension

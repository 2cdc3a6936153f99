"""Decoder: the schema of one data source (graphlearn/python/data/decoder.py).

A source line is `id [weight] [label] [timestamp] [attributes]` (edges: src_id,
dst_id first); the attribute column packs typed values separated by
`attr_delimiter`.  `attr_types` entries are "int", "float", "string", or
("string", buckets) -- a string hashed into an int attribute -- and
("string", buckets_or_None, True) for a multi-valued string kept as a string.
"""

_BASE_TYPES = ("int", "float", "string")


class Decoder(object):

  def __init__(self, weighted=False, labeled=False, timestamped=False, attr_types=None,
               attr_delimiter=":", attr_dims=None):
    if attr_types is not None and not isinstance(attr_types, list):
      raise ValueError("attr_types for Decoder must be a list, got {}.".format(type(attr_types)))
    self._weighted = bool(weighted)
    self._labeled = bool(labeled)
    self._timestamped = bool(timestamped)
    self._attr_types = list(attr_types) if attr_types else []
    self._attr_delimiter = attr_delimiter
    self._attr_dims = list(attr_dims) if attr_dims else []
    counts = {"int": 0, "float": 0, "string": 0}
    for spec in self._attr_types:
      counts[self.stored_kind(spec)] += 1
    self._counts = counts

  @staticmethod
  def parse(attr_type):
    """-> (type_name, bucket_size, is_multival)"""
    if isinstance(attr_type, (tuple, list)):
      name = attr_type[0]
      bucket = attr_type[1] if len(attr_type) > 1 else None
      multi = bool(attr_type[2]) if len(attr_type) > 2 else False
    else:
      name, bucket, multi = attr_type, None, False
    if name not in _BASE_TYPES:
      raise ValueError("attribute type must be one of %s, got %r" % (_BASE_TYPES, name))
    if multi and name != "string":
      raise ValueError("multi-value attribute must be string type.")
    return name, bucket, multi

  @classmethod
  def stored_kind(cls, attr_type):
    """Which of the three attribute arrays the value ends up in."""
    name, bucket, multi = cls.parse(attr_type)
    if name == "string" and bucket is not None and not multi:
      return "int"  # hashed into `bucket` buckets
    return name

  weighted = property(lambda self: self._weighted)
  labeled = property(lambda self: self._labeled)
  timestamped = property(lambda self: self._timestamped)
  attributed = property(lambda self: bool(self._attr_types))
  attr_types = property(lambda self: self._attr_types)
  attr_delimiter = property(lambda self: self._attr_delimiter)
  attr_dims = property(lambda self: self._attr_dims)
  int_attr_num = property(lambda self: self._counts["int"])
  float_attr_num = property(lambda self: self._counts["float"])
  string_attr_num = property(lambda self: self._counts["string"])

  @property
  def has_property(self):
    return self._weighted or self._labeled or self._timestamped or self.attributed

  @property
  def data_format(self):
    """io::DataFormat bits (graphlearn/src/include/constants.h)."""
    return 2 * self._weighted + 4 * self._labeled + 8 * self._timestamped + 16 * self.attributed

  def format_attrs(self, int_attrs, float_attrs, string_attrs):
    """Flat lookup results -> [n, num] arrays."""
    def shaped(a, num):
      return None if (a is None or num == 0) else a.reshape(-1, num)
    return (shaped(int_attrs, self.int_attr_num), shaped(float_attrs, self.float_attr_num),
            shaped(string_attrs, self.string_attr_num))

"""Dataset (graphlearn/python/nn/dataset.py:30-255): every batch of a GSL query as {alias: Data}.

Which attributes an alias carries is decided once, from its decoder (get_mask, dataset.py:182-214): int / float /
string attributes, labels, weights, timestamps; ids (src_ids + dst_ids for an edge step); offsets / indices /
dense_shape for a by("full") hop.  Attributes of a [batch, count] hop are flattened to [batch * count, num] and its
labels / weights to [batch * count] (_reformat_features, dataset.py:216-255).  SubGraph() steps have no Data form and
are passed through untouched."""
from collections import OrderedDict

import numpy as np

from graphlearn import gsl
from graphlearn.errors import OutOfRangeError
from graphlearn.nn.data import Data

_FEATS = ("int_attr_num", "float_attr_num", "string_attr_num", "labeled", "weighted", "timestamped")


class Dataset(object):

  def __init__(self, query, window=10, batch_size=1, drop_last=False, fuse_hops=False, device=False, prefetch=False):
    """fuse_hops / device (new): gsl.Dataset's -- chains of dense hops as one engine call, and their values left on
    the GPU: such an alias' ids and float_attrs are torch CUDA tensors (the gather of the float attributes runs on the
    device too); any other column of it is looked up through the host as usual."""
    if not isinstance(query, gsl.Query) or query.values_func is None:
      raise ValueError("Dataset takes a GSL query that ends with .values()")
    self._dag = query
    self._ds = gsl.Dataset(query, window=window, drop_last=drop_last, fuse_hops=fuse_hops, device=device,
                           prefetch=prefetch)
    self._device = bool(device)
    self.batch_size = batch_size
    self.drop_last = drop_last
    self._masks = OrderedDict()
    self._passthrough = []
    for alias, step in query.aliases.items():
      kind = step._describe()  # pylint: disable=protected-access
      if kind is None:
        self._passthrough.append(alias)
        continue
      decoder, is_edge, is_sparse = kind
      self._masks[alias] = self.get_mask(decoder, is_edge=is_edge, is_sparse=is_sparse)

  def __iter__(self):
    def iterator():
      while True:
        try:
          yield self.get_data_dict()
        except OutOfRangeError:
          break
    return iterator()

  def close(self):
    self._ds.close()

  @property
  def iterator(self):
    return self.__iter__()

  @property
  def masks(self):
    """alias -> (feature masks [6], id masks [2], sparse masks [3])."""
    return self._masks

  @staticmethod
  def get_mask(node_decoder, is_edge=False, is_sparse=False):
    feat_masks = []
    for feat in _FEATS:
      spec = getattr(node_decoder, feat)
      feat_masks.append(bool(spec) if isinstance(spec, bool) else spec > 0)
    return feat_masks, [True, bool(is_edge)], [bool(is_sparse)] * 3

  @staticmethod
  def _reformat_features(value, feat_masks):
    def rows(feat):
      return None if feat is None else np.reshape(feat, (-1, feat.shape[-1]))

    def flat(feat):
      return None if feat is None else np.asarray(feat).reshape(-1)
    names = ("int_attrs", "float_attrs", "string_attrs", "labels", "weights", "timestamps")
    out = []
    for name, wanted, shaper in zip(names, feat_masks, (rows, rows, rows, flat, flat, flat)):
      out.append(shaper(getattr(value, name)) if wanted else None)  # an unwanted column is never looked up
    return out

  def _on_device(self, nodes):
    import torch
    from graphlearn.values import DeviceNodes
    graph = self._dag.graph
    dev = torch.device("cuda", graph.device_features(nodes.type).device)
    ids = torch.from_numpy(np.ascontiguousarray(nodes.ids, dtype=np.int64)).to(dev)
    return DeviceNodes(ids, nodes.type, graph)

  def _device_row(self, value, feat_masks, host=None):
    other = [m for i, m in enumerate(feat_masks) if i != 1]
    if any(other):
      host = self._reformat_features(host if host is not None else value.to_host(),
                                     [m and i != 1 for i, m in enumerate(feat_masks)])
    else:
      host = [None] * 6
    if feat_masks[1]:
      floats = value.float_attrs
      host[1] = floats.reshape(-1, floats.shape[-1])
    return [v for v, m in zip(host + [value.ids.reshape(-1)], feat_masks + [True]) if m]

  def get_flatten_values(self):
    """The raw arrays of a batch, alias by alias, in the order build_data_dict consumes them."""
    values = self._ds.next()
    res = []
    for alias, (feat_masks, id_masks, sparse_masks) in self._masks.items():
      value = values[alias]
      if hasattr(value, "to_host"):  # values.DeviceNodes: ids and float attributes never leave the GPU
        res.extend(self._device_row(value, feat_masks))
        continue
      if self._device and not id_masks[1] and not sparse_masks[-1] and feat_masks[1]:
        # a host-produced vertex batch (source, end points, negatives, walks) in device mode: its float attributes
        # are gathered on the GPU from the ids instead of being looked up through the host and copied twice
        res.extend(self._device_row(self._on_device(value), feat_masks, host=value))
        continue
      row = self._reformat_features(value, feat_masks)
      if id_masks[1]:
        row.extend([np.asarray(value.src_ids).reshape(-1), np.asarray(value.dst_ids).reshape(-1)])
      else:
        row.extend([np.asarray(value.ids).reshape(-1), None])
      if sparse_masks[-1]:
        row.extend([np.asarray(value.offsets), np.asarray(value.indices), np.asarray(value.dense_shape)])
      else:
        row.extend([None, None, None])
      res.extend(v for v, m in zip(row, feat_masks + id_masks + sparse_masks) if m)
    self._last_passthrough = {alias: values[alias] for alias in self._passthrough}
    return res

  def build_data_dict(self, flatten_values):
    data_dict = {}
    cursor = [-1]

    def pop(mask):
      if mask:
        cursor[0] += 1
        return flatten_values[cursor[0]]
      return None
    for alias, masks in self._masks.items():
      ints, floats, strings, labels, weights, timestamps, ids, dst_ids, offsets, indices, dense_shape = \
          [pop(m) for m in sum(masks, [])]
      data_dict[alias] = Data(ids, ints, floats, strings, labels, weights, timestamps, dst_ids=dst_ids,
                              offsets=offsets, indices=indices, dense_shape=dense_shape)
    return data_dict

  def get_data_dict(self):
    data = self.build_data_dict(self.get_flatten_values())
    data.update(self._last_passthrough)
    return data

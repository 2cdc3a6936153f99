"""Dataset (graphlearn/python/nn/pytorch/data/dataset.py:32-112): a GSL query as a torch IterableDataset -- each item
is {alias: Data of torch tensors}, or after as_dict() {alias: {name: tensor}} for torch's own DataLoader, or whatever
`induce_func(data_dict)` returns (the reference hands it to a PyG Data builder).  String attributes stay numpy arrays
of bytes (torch has no string tensor); as_dict() leaves them out so that the default collate function works.

`device="cuda"`: every tensor on the GPU the engine runs on; chains of dense hops then run as ONE engine call and their
ids and float attributes are produced there and never visit the host (gsl.Dataset fuse_hops + device).  The lazy /
client-server constructor arguments of the reference (graph=, cluster=) belong to its RPC deploy modes and raise."""
import numpy as np
import torch as th

from graphlearn.nn.dataset import Dataset as RawDataset


class Dataset(th.utils.data.IterableDataset):

  def __init__(self, query, window=10, induce_func=None, graph=None, cluster=None, device=None, prefetch=False):
    super(Dataset, self).__init__()
    if graph is not None or cluster is not None:
      raise NotImplementedError("lazy initialisation against a running server is a client / server deploy mode; "
                                "this engine runs in process (pass the query of an initialised Graph)")
    self._device = th.device(device) if device is not None else None
    on_gpu = self._device is not None and self._device.type == "cuda"
    # on the GPU: chains of dense hops run as one engine call and their ids / float attributes stay in HBM
    # prefetch: up to `window` batches are sampled ahead on a background thread (gsl.Dataset), overlapping the model step
    self._rds = RawDataset(query, window=window, fuse_hops=on_gpu, device=on_gpu, prefetch=prefetch)
    self._induce_func = induce_func
    self._format = lambda x: x
    self._client_id = 0

  def __iter__(self):
    for value in self._rds:
      if self._induce_func is not None:
        yield self._induce_func(value)
        continue
      out = {}
      for k, v in value.items():
        if hasattr(v, "apply"):
          v.apply(self._convert_func)
          out[k] = self._format(v)
        else:
          out[k] = v
      yield out

  def as_dict(self):
    def func(x):
      return {k: v for k, v in x.__dict__.items() if isinstance(v, th.Tensor)}
    self._format = func
    return self

  def lazy_init(self):
    return False

  def close(self):
    self._rds.close()

  def _convert_func(self, data):
    if isinstance(data, dict):
      return {k: self._convert_func(v) for k, v in data.items()}
    if isinstance(data, th.Tensor):
      return data.to(self._device) if self._device is not None else data
    arr = np.asarray(data)
    if arr.dtype.kind in "OSU":  # strings
      return arr
    t = th.from_numpy(np.ascontiguousarray(arr))
    return t.to(self._device) if self._device is not None else t

  @property
  def client_id(self):
    return self._client_id

  @client_id.setter
  def client_id(self, value):
    if not isinstance(value, int):
      raise ValueError("client_id should be int type")
    self._client_id = value

"""Device-resident mini-batch loader: the caller that sits right above the hot path in a
training loop (the role of graphlearn/python/nn/pytorch/data/{dataset,pyg_dataloader}.py,
which drive the same sample -> lookup steps through GSL and numpy).

Every batch is produced on the GPU and stays there: seed ids -> all hops in one
glx_sample_hops call -> float attributes of every frontier gathered by glx_lookup.  No
request objects, no host round trips, no worker processes: one iteration is a handful of
kernel launches on the current torch stream.

    loader = gl.NeighborLoader(g, "item", ["i2i", "i2i"], [15, 10], batch_size=1024,
                               strategy="edge_weight", shuffle=True)
    for batch in loader:                 # one epoch
        batch.seeds                      # [B] int64 cuda
        batch.nbr[h], batch.eid[h]       # hop h+1: [rows_h, fanout_h] int64 cuda
        batch.x[h]                       # float attributes of frontier h ([B, D], [B*f1, D], ...)
        src, dst = batch.edge_index(h)   # COO of hop h+1 in frontier-local positions

Seeding: batch i of epoch e uses call counter (e * batches_per_epoch + i) * hops, so a
(sampling seed, epoch, batch) triple always reproduces the same sample.
"""
import numpy as np

__all__ = ["NeighborLoader", "NeighborBatch"]


class NeighborBatch(object):
  """Tensors of one mini-batch (all on the GPU)."""

  def __init__(self, seeds, nbr, eid, x):
    self.seeds, self.nbr, self.eid, self.x = seeds, nbr, eid, x

  @property
  def num_hops(self):
    return len(self.nbr)

  def frontier(self, h):
    """ids of frontier h: the seeds (h = 0) or hop h's sampled neighbours, flattened"""
    return self.seeds if h == 0 else self.nbr[h - 1].reshape(-1)

  def edge_index(self, h):
    """(src, dst): position of each hop-(h+1) edge's source in frontier h and of its
    destination in frontier h+1 -- the dense fan-out layout makes both closed-form."""
    import torch
    rows, k = self.nbr[h].shape
    dst = torch.arange(rows * k, device=self.nbr[h].device)
    return torch.div(dst, k, rounding_mode="floor"), dst


class NeighborLoader(object):

  def __init__(self, graph, node_type, meta_path, fanouts, batch_size, strategy="random", shuffle=True,
               drop_last=False, with_features=True, seed_ids=None):
    import torch
    from graphlearn import settings
    self._graph = graph
    self._sampler = graph.neighbor_sampler(meta_path, fanouts, strategy=strategy)
    self._hops = len(fanouts)
    self._batch_size = int(batch_size)
    self._shuffle = shuffle
    self._drop_last = drop_last
    self._device = torch.device("cuda", settings._MIRROR.get("device_id", 0))  # pylint: disable=protected-access
    ids = graph.get_server().node_ids(node_type) if seed_ids is None else np.asarray(seed_ids, np.int64)
    self._ids = torch.from_numpy(np.ascontiguousarray(ids)).to(self._device)
    self._epoch = 0
    topo = graph.get_topology()
    types = [node_type] + [topo.get_dst_type(e) for e in (meta_path if isinstance(meta_path, (list, tuple)) else [meta_path])]
    self._feats = None
    if with_features:
      self._feats = []
      for t in types:
        try:
          self._feats.append(graph.device_features(t))
        except ValueError:
          self._feats.append(None)  # a type without float attributes

  def __len__(self):
    n = self._ids.shape[0]
    return n // self._batch_size if self._drop_last else (n + self._batch_size - 1) // self._batch_size

  def __iter__(self):
    import torch
    from graphlearn import settings
    n = self._ids.shape[0]
    order = self._ids
    if self._shuffle:
      gen = torch.Generator(device=self._device)
      gen.manual_seed(int(settings._MIRROR["sampling_seed"]) * 1000003 + self._epoch)  # pylint: disable=protected-access
      order = self._ids[torch.randperm(n, generator=gen, device=self._device)]
    batches = len(self)
    default_attr = float(settings._MIRROR.get("default_float_attr", 0.0))  # pylint: disable=protected-access
    for i in range(batches):
      seeds = order[i * self._batch_size:(i + 1) * self._batch_size].contiguous()
      cc = (self._epoch * batches + i) * self._hops
      hops = self._sampler.get_device(seeds, call_counter=cc)
      nbr = [h[0] for h in hops]
      eid = [h[1] for h in hops]
      x = None
      if self._feats is not None:
        x = []
        for h, f in enumerate(self._feats):
          ids = seeds if h == 0 else nbr[h - 1].reshape(-1)
          x.append(f.lookup(ids, default_attr) if f is not None else None)
      yield NeighborBatch(seeds, nbr, eid, x)
    self._epoch += 1

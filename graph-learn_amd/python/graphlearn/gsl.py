"""GSL: the reference's graph sampling language (graphlearn/python/gsl/{dag,dag_node,dag_edge,dag_dataset}.py),
the other documented way into the sampling path beside the sampler objects:

  q = g.V("user").batch(64).alias("seed") \\
       .outV("buy").sample(10).by("edge_weight").alias("hop1") \\
       .outV("similar").sample(5).by("topk").alias("hop2") \\
       .values()
  ds = gl.Dataset(q)
  while True:
    try: batch = ds.next()          # {"seed": Nodes [64], "hop1": Nodes [64, 10], "hop2": Nodes [640, 5]}
    except gl.OutOfRangeError: break

The reference turns a query into a DagDef protobuf, ships it to the servers once, and a scheduler thread there runs the
DAG node by node into bounded tapes (core/dag/*, core/runner/dag_node_runner.cc:32-109, dag_scheduler.cc) that
Dataset.next() pops.  Here a query is a list of steps in creation order -- every step's upstream comes before it --
and Dataset.next() evaluates them on the spot, one operator request per step through the same client the sampler
objects use (every traversal IS one of them: NodeSampler / EdgeSampler / NeighborSampler / NegativeSampler /
ConditionalNegativeSampler / Graph.random_walk), so a GSL step draws exactly what the matching sampler object draws.
Results are the same Nodes / Edges / SparseNodes values, attributes looked up lazily.

Supported: V / E sources (batch, shuffle, mask, node_from), outV / inV / outE / inE with sample().by(), filter(),
outNeg / inNeg / Neg with where(), each(), random_walk(), SubGraph(), alias(), values(func), Dataset(window,
drop_last, fuse_hops).  Not here: feeding a query from a generator.
"""
import numpy as np

from graphlearn import pywrap_graphlearn as pywrap
from graphlearn.utils import Mask

__all__ = ["Dataset"]


class Query(object):
  """What dag.py's Dag is to the reference: the steps of one query and what .values() asked for."""

  def __init__(self, graph):
    self.graph = graph
    self.steps = []
    self.aliases = {}
    self.values_func = None

  def add(self, step):
    if self.values_func is not None:
      raise ValueError("the query is closed: values() has been called")
    self.steps.append(step)

  def name(self, step, alias):
    if not isinstance(alias, str) or not alias:
      raise ValueError("alias must be a non-empty string")
    if alias in self.aliases:
      raise ValueError("alias {!r} is already used in this query".format(alias))
    self.aliases[alias] = step

  def step_of(self, target):
    if isinstance(target, Step):
      return target
    if target not in self.aliases:
      raise ValueError("no step is called {!r}".format(target))
    return self.aliases[target]

  def close(self, func):
    if self.values_func is not None:
      raise ValueError("values() has already been called on this query")
    if not callable(func):
      raise ValueError("values() takes a function of the result dictionary")
    if not any(isinstance(s, (VertexSource, EdgeSource)) for s in self.steps):
      raise ValueError("a query starts with g.V() or g.E()")
    self.values_func = func
    return self


class Step(object):
  """One traversal (dag_node.py DagNode).  `upstream` is the step whose output this one consumes."""

  def __init__(self, query, upstream=None):
    self._query = query
    self._upstream = upstream
    self._alias = None
    query.add(self)

  # -- the chainable surface (dag_node.py:165-306) ---------------------------------------------
  def alias(self, alias):
    self._query.name(self, alias)
    self._alias = alias
    return self

  def each(self, func):
    """func(this step) builds any number of branches from it (dag_node.py:294-296)."""
    func(self)
    return self

  def values(self, func=lambda x: x):
    """Closes the query; Dataset.next() returns func({alias: Nodes | Edges})."""
    return self._query.close(func)

  def _off(self, what):
    raise ValueError("{} does not apply to {}".format(what, type(self).__name__))

  def batch(self, batch_size):
    self._off("batch()")

  def shuffle(self, traverse=False):
    self._off("shuffle()")

  def sample(self, count):
    self._off("sample()")

  def by(self, strategy):
    self._off("by()")

  def filter(self, target):
    self._off("filter()")

  def where(self, target, condition=None):
    self._off("where()")

  def _evaluate(self, results):
    raise NotImplementedError

  def _describe(self):
    """(decoder, is_edge, is_sparse) of the value this step yields, or None when it is not a Nodes / Edges batch:
    what graphlearn.nn.Dataset derives an alias' Data layout from (nn/dataset.py:52-57)."""
    return None

  def _node_kind(self, sparse=False):
    return self._query.graph.get_node_decoder(self._vertex_type()), False, sparse


class _VertexTraversals(object):
  """What can follow a step that yields vertices (dag_node.py TraverseVertexDagNode:462-530)."""

  def _vertex_type(self):
    raise NotImplementedError

  def _hop(self, cls, edge_type, reverse):
    topo = self._query.graph.get_topology()
    stored = edge_type + "_reverse" if reverse else edge_type
    if stored not in self._query.graph.get_edge_decoders():
      raise ValueError("edge type {} is not in the graph{}".format(
          edge_type, " as an undirected type (inV / inE / inNeg walk its reversed twin)" if reverse else ""))
    if topo.get_src_type(stored) != self._vertex_type():
      raise ValueError("{} starts at {} vertices, this step yields {}".format(stored, topo.get_src_type(stored),
                                                                           self._vertex_type()))
    return cls(self._query, self, stored)

  def outV(self, edge_type=None):  # pylint: disable=invalid-name
    return self._hop(NeighborStep, edge_type, False)

  def inV(self, edge_type=None):  # pylint: disable=invalid-name
    return self._hop(NeighborStep, edge_type, True)

  def outE(self, edge_type):  # pylint: disable=invalid-name
    return self._hop(NeighborEdgeStep, edge_type, False)

  def inE(self, edge_type):  # pylint: disable=invalid-name
    return self._hop(NeighborEdgeStep, edge_type, True)

  def outNeg(self, edge_type):  # pylint: disable=invalid-name
    return self._hop(NegativeStep, edge_type, False)

  def inNeg(self, edge_type):  # pylint: disable=invalid-name
    return self._hop(NegativeStep, edge_type, True)

  def Neg(self, node_type):  # pylint: disable=invalid-name
    """Negatives drawn from a node type by node weight (dag_node.py:509-515)."""
    if node_type not in self._query.graph.get_node_decoders():
      raise ValueError("node type {} is not in the graph".format(node_type))
    return NegativeStep(self._query, self, node_type, from_nodes=True)

  def random_walk(self, edge_type, walk_len=1, p=1.0, q=1.0):
    if edge_type not in self._query.graph.get_edge_decoders():
      raise ValueError("edge type {} is not in the graph".format(edge_type))
    return WalkStep(self._query, self, edge_type, int(walk_len), float(p), float(q))

  def SubGraph(self, nbr_type, num_nbrs=(0,), need_dist=False):  # pylint: disable=invalid-name
    """The sub-graph `nbr_type` induces among this step's vertices (+ num_nbrs sampled neighbours per hop):
    dag_node.py:532-556 -> SubGraphSampler.  -> a SubGraph value (nodes, edge_index, edges)."""
    return SubGraphStep(self._query, self, nbr_type, num_nbrs, need_dist)


class VertexSource(Step, _VertexTraversals):
  """g.V(t): batches of vertices of a node type, or of the end points of an edge type (node_from)."""

  def __init__(self, query, t, node_from=pywrap.NodeFrom.NODE, mask=Mask.NONE):
    Step.__init__(self, query)
    self._t, self._node_from, self._mask = t, node_from, mask
    self._batch_size, self._strategy = 64, "by_order"
    self._sampler = None
    self._sampler = self._make()  # validates the type now, like the reference's Graph.V

  def _make(self):
    return self._query.graph.node_sampler(self._t, batch_size=self._batch_size, strategy=self._strategy,
                                          node_from=self._node_from, mask=self._mask)

  def _vertex_type(self):
    return self._sampler._node_type  # pylint: disable=protected-access

  def batch(self, batch_size):
    if int(batch_size) <= 0:
      raise ValueError("batch size must be positive")
    self._batch_size = int(batch_size)
    self._sampler = self._make()
    return self

  def shuffle(self, traverse=False):
    """traverse=True: every vertex once per epoch in random order; False: independent random draws (no epochs)."""
    self._strategy = "shuffle" if traverse else "random"
    self._sampler = self._make()
    return self

  def _evaluate(self, results):
    return self._sampler.get()

  def _describe(self):
    return self._node_kind()


class EdgeSource(Step):
  """g.E(edge_type): batches of edges; outV() / inV() give their end points."""

  def __init__(self, query, edge_type, mask=Mask.NONE):
    Step.__init__(self, query)
    self._edge_type, self._mask = edge_type, mask
    self._batch_size, self._strategy = 64, "by_order"
    self._sampler = self._make()

  def _make(self):
    return self._query.graph.edge_sampler(self._edge_type, batch_size=self._batch_size, strategy=self._strategy,
                                          mask=self._mask)

  def batch(self, batch_size):
    if int(batch_size) <= 0:
      raise ValueError("batch size must be positive")
    self._batch_size = int(batch_size)
    self._sampler = self._make()
    return self

  def shuffle(self, traverse=False):
    self._strategy = "shuffle" if traverse else "random"
    self._sampler = self._make()
    return self

  def outV(self):  # pylint: disable=invalid-name
    return EndpointStep(self._query, self, "src")

  def inV(self):  # pylint: disable=invalid-name
    return EndpointStep(self._query, self, "dst")

  def SubGraph(self, nbr_type, num_nbrs=(0,), need_dist=False):  # pylint: disable=invalid-name
    """The sub-graph around the batch's (src, dst) pairs (dag_node.py:647-674; SEAL-style with need_dist)."""
    return SubGraphStep(self._query, self, nbr_type, num_nbrs, need_dist, pairs=True)

  def _stored_edge_type(self):
    return self._sampler._stored  # pylint: disable=protected-access

  def _evaluate(self, results):
    return self._sampler.get()

  def _describe(self):
    return self._query.graph.get_edge_decoder(self._stored_edge_type()), True, False


class EndpointStep(Step, _VertexTraversals):
  """The source or destination vertices of the edges an upstream step yields (dag_node.py:583-593, 633-645)."""

  def __init__(self, query, upstream, end):
    Step.__init__(self, query, upstream)
    self._end = end
    topo = query.graph.get_topology()
    edge_type = upstream._stored_edge_type()  # pylint: disable=protected-access
    self._type = topo.get_src_type(edge_type) if end == "src" else topo.get_dst_type(edge_type)

  def _vertex_type(self):
    return self._type

  def _evaluate(self, results):
    edges = results[self._upstream]
    ids = edges.src_ids if self._end == "src" else edges.dst_ids
    offsets = getattr(edges, "offsets", None)
    if offsets is not None:
      return self._query.graph.get_nodes(self._type, ids, offsets=offsets, shape=edges.dense_shape)
    return self._query.graph.get_nodes(self._type, ids, shape=edges.shape)

  def _describe(self):
    up = self._upstream._describe()  # pylint: disable=protected-access
    return self._node_kind(sparse=bool(up and up[2]))


class _Sampled(Step):
  """A step that needs .sample(count).by(strategy)."""

  _strategies = ()

  def __init__(self, query, upstream, stored):
    Step.__init__(self, query, upstream)
    self._stored = stored
    self._count = None
    self._strategy = "random"
    self._filter_target = None

  def sample(self, count):
    if int(count) < 0:
      raise ValueError("sample() takes a non-negative count")
    self._count = int(count)
    return self

  def by(self, strategy):
    if strategy not in self._strategies:
      raise ValueError("by(): strategy must be one of {}, got {!r}".format(self._strategies, strategy))
    self._strategy = strategy
    return self

  def _need_count(self):
    if self._count is None:
      raise ValueError("{} over {} needs .sample(count)".format(type(self).__name__, self._stored))


class NeighborStep(_Sampled, _VertexTraversals):
  """outV / inV with sample().by(): one hop of a NeighborSampler -> Nodes [upstream size, count] (SparseNodes for
  by("full")).  filter(target): neighbours equal to the target step's id of the same row are never drawn
  (dag_node.py:212-231 -> op::Filter EQUAL on ID)."""

  _strategies = ("random", "random_without_replacement", "topk", "in_degree", "edge_weight", "full")

  def _vertex_type(self):
    return self._query.graph.get_topology().get_dst_type(self._stored)

  def filter(self, target):
    self._filter_target = self._query.step_of(target)
    return self

  def _layer(self, results):
    self._need_count()
    src = results[self._upstream].ids.reshape(-1)
    sampler = self._query.graph.neighbor_sampler(self._stored, self._count, strategy=self._strategy)
    values = None
    if self._filter_target is not None:
      values = results[self._filter_target].ids.reshape(-1)
      if values.size != src.size:
        raise ValueError("filter(): the target step yields {} ids, this step has {} rows".format(values.size, src.size))
      sampler.set_filter("equal", "id")
    return sampler.get(src, filter_values=values)

  def _evaluate(self, results):
    return self._layer(results).layer_nodes(1)

  def _describe(self):
    return self._node_kind(sparse=self._strategy == "full")


class NeighborEdgeStep(NeighborStep):
  """outE / inE: the same hop, the sampled EDGES as the value; inV() / outV() then give their end points."""

  def outV(self):  # pylint: disable=invalid-name,arguments-differ
    return EndpointStep(self._query, self, "src")

  def inV(self):  # pylint: disable=invalid-name,arguments-differ
    return EndpointStep(self._query, self, "dst")

  def _stored_edge_type(self):
    return self._stored

  def _evaluate(self, results):
    return self._layer(results).layer_edges(1)

  def _describe(self):
    return self._query.graph.get_edge_decoder(self._stored), True, self._strategy == "full"


class NegativeStep(_Sampled, _VertexTraversals):
  """outNeg / inNeg / Neg: count negatives per upstream vertex; where(target, condition) makes them conditional on
  the target step's attributes (dag_node.py:233-292 -> ConditionalNegativeSampler)."""

  _strategies = ("random", "in_degree", "soft_in_degree", "node_weight")

  def __init__(self, query, upstream, stored, from_nodes=False):
    _Sampled.__init__(self, query, upstream, stored)
    self._from_nodes = from_nodes
    if from_nodes:
      self._strategy = "node_weight"
    self._where = None

  def _vertex_type(self):
    if self._from_nodes:
      return self._stored
    return self._query.graph.get_topology().get_dst_type(self._stored)

  def where(self, target, condition=None):
    condition = dict(condition or {})
    allowed = {"batch_share", "unique", "int_cols", "int_props", "float_cols", "float_props", "str_cols", "str_props"}
    if set(condition) - allowed:
      raise ValueError("where(): unknown condition keys {}".format(sorted(set(condition) - allowed)))
    self._where = (self._query.step_of(target), condition)
    return self

  def _evaluate(self, results):
    self._need_count()
    graph = self._query.graph
    src = results[self._upstream].ids.reshape(-1)
    if self._where is None:
      return graph.negative_sampler(self._stored, self._count, strategy=self._strategy).get(src)
    target, condition = self._where
    dst = results[target].ids.reshape(-1)
    if dst.size != src.size:
      raise ValueError("where(): the target step yields {} ids, this step has {} rows".format(dst.size, src.size))
    sampler = graph.negative_sampler(self._stored, self._count, strategy=self._strategy, conditional=True, **condition)
    return sampler.get(src, dst)

  def _describe(self):
    return self._node_kind()


class WalkStep(Step, _VertexTraversals):
  """random_walk(edge_type, walk_len, p, q) -> Nodes [upstream size, walk_len]."""

  def __init__(self, query, upstream, edge_type, walk_len, p, q):
    Step.__init__(self, query, upstream)
    self._edge_type, self._walk_len, self._p, self._q = edge_type, walk_len, p, q

  def _vertex_type(self):
    return self._query.graph.get_topology().get_dst_type(self._edge_type)

  def _evaluate(self, results):
    src = np.ascontiguousarray(results[self._upstream].ids.reshape(-1), dtype=np.int64)
    walks = self._query.graph.random_walk(self._edge_type, src, self._walk_len, p=self._p, q=self._q)
    return self._query.graph.get_nodes(self._vertex_type(), walks.reshape(-1), shape=(src.size, self._walk_len))

  def _describe(self):
    return self._node_kind()


class SubGraphStep(Step):
  """SubGraph(): one SubGraphSampler request per batch; its value is a sampler.SubGraph, not Nodes, so nothing chains
  from it (as in the reference, where SubGraphDagNode has no traversals)."""

  def __init__(self, query, upstream, nbr_type, num_nbrs, need_dist, pairs=False):
    Step.__init__(self, query, upstream)
    self._pairs = pairs
    self._sampler = query.graph.subgraph_sampler(nbr_type, num_nbrs=num_nbrs, need_dist=need_dist)

  def _evaluate(self, results):
    up = results[self._upstream]
    if self._pairs:
      return self._sampler.get(up.src_ids.reshape(-1), up.dst_ids.reshape(-1))
    return self._sampler.get(up.ids.reshape(-1))


class _HostView(dict):
  """results, with DeviceNodes values converted to Nodes on access (once per value)."""

  def __init__(self, results):
    dict.__init__(self, results)
    self._host = {}

  def __getitem__(self, step):
    value = dict.__getitem__(self, step)
    if hasattr(value, "to_host"):
      if step not in self._host:
        self._host[step] = value.to_host()
      return self._host[step]
    return value


class Dataset(object):
  """dag_dataset.py Dataset: next() -> the query's values for one more batch of its source; raises OutOfRangeError
  at the end of an epoch (the following next() starts the next one).  `window` is the reference's prefetch depth and
  has nothing to size here: a batch is produced when it is asked for.  drop_last: a final batch shorter than the
  source's batch size is skipped."""

  _DENSE = ("random", "random_without_replacement", "topk", "in_degree", "edge_weight")

  def __init__(self, query, window=10, drop_last=False, fuse_hops=False, device=False, prefetch=False):
    """prefetch (new): produce batches ahead of next() on a background thread, up to `window` of them -- what the
    reference's tapes do for a query (dag_dataset.py / core/dag/tape.h: the scheduler fills a bounded queue that
    next() pops).  The engine calls release the GIL and run on the GPU, so sampling batch i + 1 overlaps whatever the
    caller does with batch i; batches arrive in the order a plain next() would produce them, OutOfRangeError included
    (one per epoch, the following next() belongs to the next epoch).  close() -- or garbage collection -- stops the thread.
    device (new, with fuse_hops): the hops of a fused chain yield values.DeviceNodes -- ids, float attributes and
    aggregates as torch CUDA tensors that never visit the host (`.to_host()` gives the ordinary Nodes).  A step
    downstream of such a hop that is NOT part of the chain sees its upstream through .to_host().
    fuse_hops (new): a chain .outV(e1).sample(k1).by(s).outV(e2).sample(k2).by(s)... of dense, unfiltered hops with
    one strategy runs as ONE engine call (glx_sample_hops through NeighborSampler.get_device: the frontiers stay in HBM,
    one copy back per hop) instead of one request per hop; values, shapes and types are the same, the random streams
    are the Dataset's own (seed = gl.set_sampling_seed's, a fresh call counter per batch)."""
    if not isinstance(query, Query) or query.values_func is None:
      raise ValueError("Dataset takes a query closed with .values()")
    self._query = query
    self._window = int(window)
    self._drop_last = bool(drop_last)
    self._source = next(s for s in query.steps if isinstance(s, (VertexSource, EdgeSource)))
    if device and not fuse_hops:
      raise ValueError("device=True keeps the values of FUSED chains on the GPU: pass fuse_hops=True as well")
    self._device_values = bool(device)
    self._chains = self._find_chains() if fuse_hops else {}
    self._fused_calls = int.from_bytes(__import__("os").urandom(6), "little") << 8
    self._prefetch = bool(prefetch) and self._window > 0
    if self._prefetch and getattr(query.graph, "_shard", (0, 1))[1] > 1:
      # a prefetch thread issues requests beside the consumer's own: in SPMD mode every rank must issue its
      # requests in the same order (partitioned requests are collectives), which two threads cannot promise
      raise ValueError("prefetch=True is not available on a sharded graph (requests of all ranks must stay in order)")
    if hasattr(query.graph, "add_dataset"):
      query.graph.add_dataset(self)  # dag_dataset.py:59: Graph.close() stops its datasets
    self._queue = None
    self._thread = None
    self._stop = None

  def _find_chains(self):
    """first step -> the steps of a fusable chain of two or more hops."""
    chains, taken = {}, set()
    for step in self._query.steps:
      if step in taken or type(step) is not NeighborStep:  # pylint: disable=unidiomatic-typecheck
        continue
      chain = [step]
      while True:
        nxt = [s for s in self._query.steps if type(s) is NeighborStep and s._upstream is chain[-1]]  # noqa: E721 pylint: disable=unidiomatic-typecheck,protected-access
        if len(nxt) != 1 or nxt[0]._strategy != step._strategy:  # pylint: disable=protected-access
          break
        chain.append(nxt[0])
      ok = all(s._filter_target is None and s._count is not None and s._count > 0 for s in chain)  # pylint: disable=protected-access
      if len(chain) >= 2 and ok and step._strategy in self._DENSE:  # pylint: disable=protected-access
        chains[step] = chain
        taken.update(chain)
    return chains

  def _run_chain(self, chain, results):
    import torch
    graph = self._query.graph
    src = results[chain[0]._upstream].ids.reshape(-1)  # pylint: disable=protected-access
    sampler = graph.neighbor_sampler([s._stored for s in chain], [s._count for s in chain],  # pylint: disable=protected-access
                                     strategy=chain[0]._strategy)  # pylint: disable=protected-access
    device = torch.device("cuda", graph.device_graph(chain[0]._stored).device)  # pylint: disable=protected-access
    ids = torch.from_numpy(np.ascontiguousarray(src, dtype=np.int64)).to(device)
    self._fused_calls += len(chain)
    hops = sampler.get_device(ids, call_counter=self._fused_calls)
    rows = src.size
    for step, (nbr, _) in zip(chain, hops):
      if self._device_values:
        from graphlearn.values import DeviceNodes
        results[step] = DeviceNodes(nbr.reshape(rows, step._count), step._vertex_type(), graph)  # pylint: disable=protected-access
      else:
        results[step] = graph.get_nodes(step._vertex_type(), nbr.cpu().numpy(), shape=(rows, step._count))  # pylint: disable=protected-access
      rows *= step._count  # pylint: disable=protected-access

  def next(self):
    if not self._prefetch:
      return self._produce()
    if self._thread is None:
      self._start()
    kind, item = self._queue.get()
    if kind == "raise":
      raise item
    return item

  # -- prefetching -----------------------------------------------------------------------------
  def _start(self):
    import queue
    import threading
    import weakref
    self._queue = queue.Queue(maxsize=self._window)
    self._stop = threading.Event()
    self._thread = threading.Thread(target=Dataset._worker, args=(weakref.ref(self), self._queue, self._stop),
                                    name="gsl-prefetch", daemon=True)
    self._thread.start()

  @staticmethod
  def _worker(ref, out, stop):
    import queue
    from graphlearn.errors import OutOfRangeError
    while not stop.is_set():
      self = ref()
      if self is None:
        return
      try:
        item = ("value", self._produce())
        if self._device_values:  # CUDA tensors cross to the consumer's thread: finish them first
          import torch
          torch.cuda.current_stream().synchronize()
      except OutOfRangeError as e:
        item = ("raise", e)  # the end of an epoch is an item of the stream; production goes on with the next epoch
      except BaseException as e:  # pylint: disable=broad-except
        item = ("raise", e)
        stop.set()
      del self
      while True:
        try:
          out.put(item, timeout=0.1)
          break
        except queue.Full:
          if stop.is_set():
            return
      if item[0] == "raise" and not isinstance(item[1], OutOfRangeError):
        return

  def close(self):
    """Stops the prefetch thread (a no-op without one); batches already produced are dropped."""
    if self._thread is not None:
      import time
      import warnings
      self._stop.set()
      deadline = time.time() + 30.0
      while self._thread.is_alive() and time.time() < deadline:
        try:
          self._queue.get_nowait()
        except Exception:  # pylint: disable=broad-except
          pass
        self._thread.join(timeout=0.05)
      if self._thread.is_alive():
        # blocked inside an engine call (e.g. a collective whose peers have stopped): give up on it rather than
        # hang Graph.close(); the thread is a daemon and holds no lock of this object
        warnings.warn("Dataset.close(): the prefetch thread did not stop within 30 s and was abandoned")
      self._thread = None
      self._queue = None

  def __del__(self):
    try:
      if self._stop is not None:
        self._stop.set()
    except Exception:  # pylint: disable=broad-except
      pass

  def _produce(self):
    results = {}
    for step in self._query.steps:
      if step in results:  # a later hop of a fused chain
        continue
      if step in self._chains:
        self._run_chain(self._chains[step], results)
        continue
      # a step outside a fused chain runs through host requests: it sees device-resident values as ordinary Nodes
      value = step._evaluate(_HostView(results) if self._device_values else results)  # pylint: disable=protected-access
      if step is self._source and self._drop_last:
        rows = value.src_ids.size if isinstance(step, EdgeSource) else value.ids.size
        if rows < step._batch_size:  # pylint: disable=protected-access
          # the short tail of the epoch: skip it; the source's next request reports the end of the epoch
          value = step._evaluate(results)  # pylint: disable=protected-access
      results[step] = value
    named = {alias: results[step] for alias, step in self._query.aliases.items()}
    return self._query.values_func(named)

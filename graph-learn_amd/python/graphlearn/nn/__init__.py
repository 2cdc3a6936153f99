"""graphlearn.nn: what model code reads a GSL query through (graphlearn/python/nn/{data,dataset}.py):
`Data` -- a batch of Nodes / Edges as plain arrays -- and `Dataset`, which turns every Dataset.next() of a query
into {alias: Data}.  `graphlearn.nn.pytorch` wraps it as a torch IterableDataset."""
from graphlearn.nn.data import Data  # noqa: F401
from graphlearn.nn.dataset import Dataset  # noqa: F401
from graphlearn.nn.subgraph import HeteroSubGraph, SubGraph  # noqa: F401

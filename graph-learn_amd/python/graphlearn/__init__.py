"""graphlearn -- the Python API of graph-learn's sampling / aggregation path, served by
the glx MI355X engine (HIP kernels behind the C-ABI of include/glx.h).

`import graphlearn as gl` gives the names the reference's python/__init__.py exports
for this path: Graph, Decoder, Mask, Nodes/Edges/SparseNodes/SparseEdges/Layer/Layers,
the neighbor samplers, the set_* flag setters, REPLICATE/CIRCULAR, the error classes
and `pywrap` (the pybind11 module with the reference's pywrap_graphlearn names).
"""
# torch bundles its own HIP runtime: load it first so that libglx.so binds to the same one.
import torch as _torch  # noqa: F401

from graphlearn import pywrap_graphlearn as pywrap
from graphlearn.pywrap_graphlearn import IndexOption  # noqa: F401
from graphlearn.settings import *  # noqa: F401,F403
from graphlearn.errors import *  # noqa: F401,F403
from graphlearn.utils import Mask, get_mask_type, strategy2op  # noqa: F401
from graphlearn.decoder import Decoder  # noqa: F401
from graphlearn.topology import Topology  # noqa: F401
from graphlearn.values import Values, Nodes, Edges, SparseNodes, SparseEdges, Layer, Layers  # noqa: F401
from graphlearn.sampler import *  # noqa: F401,F403
from graphlearn.traversal import *  # noqa: F401,F403
from graphlearn.graph import Graph  # noqa: F401
from graphlearn.loader import NeighborLoader, NeighborBatch  # noqa: F401
from graphlearn.gsl import Dataset  # noqa: F401
from graphlearn.sampler import SubGraph  # noqa: F401  (python/data/values.py SubGraph: what subgraph_sampler().get() returns)
import graphlearn.nn as nn  # noqa: F401,E402  (gl.nn.Dataset / Data / SubGraph / HeteroSubGraph)

NODE = pywrap.NodeFrom.NODE
EDGE_SRC = pywrap.NodeFrom.EDGE_SRC
EDGE_DST = pywrap.NodeFrom.EDGE_DST
REPLICATE = pywrap.PaddingMode.REPLICATE
CIRCULAR = pywrap.PaddingMode.CIRCULAR

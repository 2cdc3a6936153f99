"""graphlearn.nn.pytorch (graphlearn/python/nn/pytorch): the torch side of graphlearn.nn."""
from graphlearn.nn.pytorch.data.dataset import Dataset  # noqa: F401

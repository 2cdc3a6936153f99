"""`graphlearn.python.*` import paths of the reference (graphlearn/python/ is where its modules live; its tests and
user code import e.g. `graphlearn.python.nn.pytorch`): aliases of this package's modules."""
import importlib
import sys

_ALIASES = {
    "nn": "graphlearn.nn",
    "nn.data": "graphlearn.nn.data",
    "nn.dataset": "graphlearn.nn.dataset",
    "nn.subgraph": "graphlearn.nn.subgraph",
    "nn.pytorch": "graphlearn.nn.pytorch",
    "nn.pytorch.data": "graphlearn.nn.pytorch.data",
    "nn.pytorch.data.dataset": "graphlearn.nn.pytorch.data.dataset",
    "errors": "graphlearn.errors",
    "utils": "graphlearn.utils",
}
for _name, _target in _ALIASES.items():
  sys.modules[__name__ + "." + _name] = importlib.import_module(_target)
nn = sys.modules[__name__ + ".nn"]
errors = sys.modules[__name__ + ".errors"]
utils = sys.modules[__name__ + ".utils"]

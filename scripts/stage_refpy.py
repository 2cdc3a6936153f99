#!/usr/bin/env python3
"""Stage the REFERENCE's own Python layer on top of this engine's pywrap_graphlearn module.

Test infrastructure, like oracle/_ref: `graphlearn/__init__.py` and every .py file under `graphlearn/python/` are
copied AS THEY LIE from /root/reference into graph-learn_amd/python/_refpy/graphlearn/ (git-ignored: never in
history; not gpurun-ignored: it travels to the GPU box, where /root/reference does not exist), and the built
pywrap_graphlearn extension is copied beside them.  `PYTHONPATH=graph-learn_amd/python/_refpy` then gives
`import graphlearn` = the reference's package, running on libglx_host.so / libglx.so.

  tests/test_refpy_names.py   (CPU)  every `pywrap.<name>` the staged tree uses exists in the module
  tests/test_gpu_refpy.py     (GPU)  the reference's python/sampler/tests, gsl/tests and python/tests run unchanged

Nothing here is imported by the product.  Usage: python scripts/stage_refpy.py [--check]
"""
import filecmp
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/graphlearn"
STAGE = os.path.join(ROOT, "graph-learn_amd", "python", "_refpy")
PACKAGE = os.path.join(STAGE, "graphlearn")


def built_module():
  found = glob.glob(os.path.join(ROOT, "graph-learn_amd", "python", "graphlearn", "pywrap_graphlearn*.so"))
  return found[0] if found else None


def copy_module():
  """The extension beside the staged package (its rpath finds graph-learn_amd/lib from either place)."""
  module = built_module()
  if module is None:
    raise RuntimeError("pywrap_graphlearn is not built: make -C graph-learn_amd")
  os.makedirs(PACKAGE, exist_ok=True)
  target = os.path.join(PACKAGE, os.path.basename(module))
  if not os.path.exists(target) or not filecmp.cmp(module, target, shallow=False):
    shutil.copy2(module, target)
  return target


def stage():
  """-> number of reference .py files staged; 0 when there is no reference here (the GPU box)."""
  if not os.path.isdir(REFERENCE):
    if os.path.isdir(PACKAGE):
      copy_module()
    return 0
  if os.path.isdir(PACKAGE):
    shutil.rmtree(PACKAGE)
  os.makedirs(PACKAGE)
  shutil.copy2(os.path.join(REFERENCE, "__init__.py"), os.path.join(PACKAGE, "__init__.py"))
  count = 1
  for base, dirs, files in os.walk(os.path.join(REFERENCE, "python")):
    dirs[:] = [d for d in dirs if d != "__pycache__" and d != "c"]  # c/: the reference's own binding sources
    rel = os.path.relpath(base, REFERENCE)
    for name in files:
      if name.endswith(".py"):
        os.makedirs(os.path.join(PACKAGE, rel), exist_ok=True)
        shutil.copy2(os.path.join(base, name), os.path.join(PACKAGE, rel, name))
        count += 1
  copy_module()
  return count


def staged():
  return os.path.isfile(os.path.join(PACKAGE, "__init__.py")) and bool(
      glob.glob(os.path.join(PACKAGE, "pywrap_graphlearn*.so")))


if __name__ == "__main__":
  if "--check" in sys.argv:
    print("staged" if staged() else "not staged")
    sys.exit(0 if staged() else 1)
  n = stage()
  print("staged %d reference python files under %s" % (n, PACKAGE) if n else
        "no reference tree here; extension refreshed" if staged() else "nothing to stage")

#!/usr/bin/env python3
"""Stage the REFERENCE's own Python layer for the tests that run it on this engine's pywrap_graphlearn module.

Test infrastructure, like oracle/_ref: `graphlearn/__init__.py` and every .py file under `graphlearn/python/` are packed
AS THEY LIE from /root/reference into ONE archive, graph-learn_amd/python/_refpy.zip (git-ignored: never in history;
not gpurun-ignored: it travels to the GPU box, where /root/reference does not exist).  tests/refpy.py unpacks it into
a directory under /tmp beside a copy of the built extension; `PYTHONPATH=<that dir>` then gives `import graphlearn` =
the reference's package, running on libglx_host.so / libglx.so.  No reference source file ever sits in this tree.

  tests/test_refpy_names.py   (CPU)  every `pywrap.<name>` the reference's tree uses exists in the module
  tests/test_gpu_refpy.py     (GPU)  the reference's python/sampler/tests, gsl/tests, python/tests and
                                     nn/pytorch/data/test run unchanged

Nothing here is imported by the product.  Usage: python scripts/stage_refpy.py [--check]
"""
import os
import sys
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/graphlearn"
ARCHIVE = os.path.join(ROOT, "graph-learn_amd", "python", "_refpy.zip")


def stage():
    """-> number of reference .py files packed; 0 when there is no reference here (the GPU box: the archive travelled)."""
    if not os.path.isdir(REFERENCE):
        return 0
    names = [(os.path.join(REFERENCE, "__init__.py"), "graphlearn/__init__.py")]
    for base, dirs, files in os.walk(os.path.join(REFERENCE, "python")):
        dirs[:] = sorted(d for d in dirs if d != "__pycache__" and d != "c")  # c/: the reference's own binding sources
        rel = os.path.relpath(base, REFERENCE)
        for name in sorted(files):
            if name.endswith(".py"):
                names.append((os.path.join(base, name), "graphlearn/%s/%s" % (rel, name)))
    tmp = ARCHIVE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for path, arc in names:
            info = zipfile.ZipInfo(arc, date_time=(2020, 1, 1, 0, 0, 0))  # reproducible: same sources, same bytes
            info.compress_type = zipfile.ZIP_DEFLATED
            with open(path, "rb") as f:
                z.writestr(info, f.read())
    os.replace(tmp, ARCHIVE)
    return len(names)


def staged():
    return os.path.isfile(ARCHIVE)


if __name__ == "__main__":
    if "--check" in sys.argv:
        print("staged" if staged() else "not staged")
        sys.exit(0 if staged() else 1)
    n = stage()
    print("packed %d reference python files into %s" % (n, ARCHIVE) if n else
          "no reference tree here; archive %s" % ("present" if staged() else "absent"))

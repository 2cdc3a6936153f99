"""Where the staged reference Python layer lives and how its test files are run (tests/test_refpy_names.py,
tests/test_gpu_refpy.py).  Staging itself is scripts/stage_refpy.py -- test infrastructure, like oracle/_ref."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "graph-learn_amd", "python", "_refpy")
PACKAGE = os.path.join(STAGE, "graphlearn")

# The reference's own test files, relative to graphlearn/python/.  Not listed: nn/ (TensorFlow / torch_geometric
# are not installed) and tests that need a second process of the RPC deployment.
TEST_GLOBS = ["sampler/tests/test_*.py", "gsl/tests/test_*.py", "tests/test_*.py"]


def staged():
    return os.path.isfile(os.path.join(PACKAGE, "__init__.py")) and bool(glob.glob(os.path.join(PACKAGE, "pywrap_graphlearn*.so")))


def test_files():
    out = []
    for pattern in TEST_GLOBS:
        out += sorted(glob.glob(os.path.join(PACKAGE, "python", pattern)))
    # the base class module has no tests of its own
    return [os.path.relpath(p, os.path.join(PACKAGE, "python")) for p in out if not p.endswith("sampler/tests/test_sampling.py")]


def env():
    e = dict(os.environ)
    e["PYTHONPATH"] = STAGE + (os.pathsep + e["PYTHONPATH"] if e.get("PYTHONPATH") else "")
    return e


def run_file(rel, cwd, timeout=600, extra=()):
    """One reference test file in its own process and its own working directory (the files write `.data_path/`
    and `.tracker_path/` into the cwd), the way the reference's test_python_ut.sh runs them: one file, one process."""
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", os.path.join(PACKAGE, "python", rel)] + list(extra)
    return subprocess.run(cmd, cwd=cwd, env=env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)

"""Where the staged reference Python layer lives and how its test files are run (tests/test_refpy_names.py,
tests/test_gpu_refpy.py).  Staging itself is scripts/stage_refpy.py -- test infrastructure, like oracle/_ref: ONE
git-ignored archive in the tree, unpacked here into a directory under /tmp beside a copy of the built extension."""
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import zipfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCHIVE = os.path.join(ROOT, "graph-learn_amd", "python", "_refpy.zip")
LIB = os.path.join(ROOT, "graph-learn_amd", "lib")


def _module():
    if os.environ.get("GLX_REFPY_MODULE"):  # an instrumented build of the extension (scripts/asan_refpy.sh)
        return os.environ["GLX_REFPY_MODULE"]
    found = glob.glob(os.path.join(ROOT, "graph-learn_amd", "python", "graphlearn", "pywrap_graphlearn*.so"))
    return found[0] if found else None


def _unpack():
    """-> the directory to put on PYTHONPATH: the archive's tree + the extension as built NOW (keyed by both)."""
    mod = _module()
    with open(ARCHIVE, "rb") as f:
        digest = hashlib.sha1(f.read())
    digest.update(("%s:%d:%d" % (mod, os.path.getsize(mod), os.stat(mod).st_mtime_ns)).encode())
    key = digest.hexdigest()[:16]
    stage = os.path.join(tempfile.gettempdir(), "glx_refpy_" + key)
    if not os.path.isfile(os.path.join(stage, ".complete")):
        tmp = tempfile.mkdtemp(prefix="glx_refpy_tmp_")
        with zipfile.ZipFile(ARCHIVE) as z:
            z.extractall(tmp)
        shutil.copy2(mod, os.path.join(tmp, "graphlearn", os.path.basename(mod)))
        open(os.path.join(tmp, ".complete"), "w").close()
        try:
            os.rename(tmp, stage)
        except OSError:  # another process got there first
            shutil.rmtree(tmp, ignore_errors=True)
    return stage


class _Lazy(object):
    """STAGE / PACKAGE as strings that are made on first use (importing this module must not unpack anything)."""

    def __init__(self, *tail):
        self.tail = tail

    def __str__(self):
        return os.path.join(_unpack(), *self.tail)

    __fspath__ = __str__

    def __repr__(self):
        return repr(str(self))


STAGE = _Lazy()
PACKAGE = _Lazy("graphlearn")

# The reference's own test files, relative to graphlearn/python/.  Not listed: nn/tf (TensorFlow is not installed).
TEST_GLOBS = ["sampler/tests/test_*.py", "gsl/tests/test_*.py", "tests/test_*.py", "nn/pytorch/data/test/test_*.py"]

# Tests that cannot pass on ANY engine, the reference's own included -- each with the defect in the reference's
# sources.  A file maps to the test methods that ARE run (everything else in it is the defect), or to None when the
# whole file is affected.
_REPLICATE_OOB = (
    "setUpClass selects REPLICATE padding (sampler/tests/test_sampling.py:114) and the test samples k = 6 (2-hop: 3) "
    "neighbours of vertices with 1..4: under REPLICATE the alias samplers hand the padder `indices` of size k, which it "
    "ignores, reading neighbors_[0..k) (core/operator/sampler/padder/replicate_padder.h:37-50) through an unchecked "
    "io::Array::operator[] (core/graph/storage/types.h:73-75) -- out of bounds for every row shorter than k; the test "
    "then requires each value read there to be a neighbour.  SURVEY 8(a) quirk 3: this engine default-fills instead of "
    "reading past the row.")
KNOWN_BROKEN_IN_REFERENCE = {
    "sampler/tests/test_in_degree_neighbor_sampling.py": (
        ["InDegreeNeighborSamplingTestCase.test_1hop_with_neighbor_missing"], _REPLICATE_OOB),
    "sampler/tests/test_edge_weight_neighbor_sampling.py": (
        ["EdgeWeightNeighborSamplingTestCase.test_1hop_with_neighbor_missing"], _REPLICATE_OOB),
    "sampler/tests/test_subgraph_sampling.py": (
        None, "the test calls g.subgraph_sampler(nbr_type=...) (test_subgraph_sampling.py:34) but Graph.subgraph_sampler "
              "requires seed_type (python/graph.py:1059-1063): a TypeError inside the reference's own Python, before any "
              "engine call."),
}


def staged():
    return os.path.isfile(ARCHIVE) and _module() is not None


def test_files():
    out = []
    for pattern in TEST_GLOBS:
        out += sorted(glob.glob(os.path.join(PACKAGE, "python", pattern)))
    # the base class module has no tests of its own
    return [os.path.relpath(p, os.path.join(PACKAGE, "python")) for p in out if not p.endswith("sampler/tests/test_sampling.py")]


def defines_tests(rel):
    """Does the file hold a test method outside a comment?"""
    with open(os.path.join(PACKAGE, "python", rel)) as f:
        return any(line.lstrip().startswith("def test") for line in f)


def env():
    e = dict(os.environ)
    e["PYTHONPATH"] = str(STAGE) + (os.pathsep + e["PYTHONPATH"] if e.get("PYTHONPATH") else "")
    # the extension's copy sits outside the tree: its $ORIGIN-relative rpath no longer finds the engine
    e["LD_LIBRARY_PATH"] = LIB + (os.pathsep + e["LD_LIBRARY_PATH"] if e.get("LD_LIBRARY_PATH") else "")
    if os.environ.get("GLX_REFPY_PRELOAD"):  # the sanitizer's runtime must be the first library of the interpreter
        e["LD_PRELOAD"] = os.environ["GLX_REFPY_PRELOAD"]
    return e


def run_file(rel, cwd, timeout=600, tests=()):
    """One reference test file in its own process and its own working directory (the files write `.data_path/`
    and `.tracker_path/` into the cwd), the way the reference's test_python_ut.sh runs them: `python <file>`, one
    file per process (the files' __main__ blocks matter: gsl/tests/test_gsl_sampling.py raises the samplers' retry
    count there).  `tests`: unittest names (Class.method) to run instead of the whole file."""
    cmd = [sys.executable, os.path.join(PACKAGE, "python", rel), "-v"] + list(tests)
    return subprocess.run(cmd, cwd=cwd, env=env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, text=True)

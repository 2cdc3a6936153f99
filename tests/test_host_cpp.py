"""The C++ host layer (graph-learn_amd/host: the mirror of graphlearn::op's
registry / request / operator / partition-stitch API) exercised by C++ test programs
that restate the reference's sampler_unittest.cpp, aggregating_op_unittest.cpp and
partition_stitch_unittest.cpp."""
import ctypes
import os
import subprocess

import pytest

import glx

LIB = os.path.join(os.path.dirname(glx.LIB_PATH))
BINARIES = ["sampler_unittest", "aggregating_op_unittest", "partition_stitch_unittest", "graph_op_unittest"]


def run(name):
    return subprocess.run([os.path.join(LIB, name)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                          timeout=300)


@pytest.mark.gpu
@pytest.mark.parametrize("name", BINARIES)
def test_host_unittests_on_gpu(name):
    r = run(name)
    assert r.returncode == 0, r.stdout
    assert "0 failure(s)" in r.stdout, r.stdout


def _no_gpu():
    n = ctypes.c_int(-1)
    return glx.lib().glx_device_count(ctypes.byref(n)) != 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
@pytest.mark.parametrize("name", BINARIES)
def test_host_layer_fails_loudly_without_gpu(name):
    """No silent CPU path: building the store without a GPU reports UNAVAILABLE."""
    r = run(name)
    assert r.returncode == 2, r.stdout
    assert "no usable HIP device" in r.stdout and "no CPU fallback" in r.stdout


def test_loader_unittest_on_cpu(tmp_path):
    """edge_loader_unittest.cpp / node_loader_unittest.cpp restated: parsing and staging are host
    work, so this one runs everywhere."""
    r = subprocess.run([os.path.join(LIB, "loader_unittest")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout
    assert "6 test(s), 0 failure(s)" in r.stdout, r.stdout


def test_request_unittest_on_cpu():
    """Request classes, Filter::FillValues, partitioning of filter values, RandomWalkRequest: no device involved."""
    r = run("request_unittest")
    assert r.returncode == 0, r.stdout
    assert "6 test(s), 0 failure(s)" in r.stdout, r.stdout


def test_dag_unittest_on_cpu():
    """The query-DAG machinery (host dag.h: compile + hop fusion, tapes, the scheduler thread, Dataset) with stand-in
    operators: core/dag/test/*_unittest.cpp and core/runner/test/dag_scheduler_unittest.cpp restated.  No device."""
    r = run("dag_unittest")
    assert r.returncode == 0, r.stdout
    assert "9 test(s), 0 failure(s)" in r.stdout, r.stdout


@pytest.mark.skipif(os.environ.get("GLX_TSAN") != "1", reason="opt-in (GLX_TSAN=1): a minute of instrumented compilation")
def test_dag_unittest_under_thread_sanitizer():
    """scripts/tsan_dag.sh: the same eight tests in a ThreadSanitizer build of the host sources -- the tape store, the
    scheduler's query threads, Dataset's prefetch thread and close-while-starved leave no report."""
    r = subprocess.run(["bash", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "tsan_dag.sh")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "ThreadSanitizer: no report" in r.stdout, r.stdout


def test_host_library_exports_registry():
    import subprocess as sp
    out = sp.run(["nm", "-DC", os.path.join(LIB, "libglx_host.so")], stdout=sp.PIPE, text=True).stdout
    for sym in ["graphlearn::op::OpFactory::Create", "graphlearn::op::OpRegistry::Register",
                "graphlearn::RequestFactory::NewRequest", "graphlearn::SamplingRequest::Set",
                "graphlearn::AggregatingRequest::Set", "graphlearn::GraphStore::GetGraph"]:
        assert sym in out, sym
    ldd = sp.run(["ldd", os.path.join(LIB, "libglx_host.so")], stdout=sp.PIPE, text=True).stdout
    assert "libglx.so" in ldd and "oracle" not in ldd

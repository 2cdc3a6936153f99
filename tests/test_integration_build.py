"""CPU tests of integration/: INTEGRATION.md quotes the compiled sources, the library built from the reference's own
translation units + the glx operator bodies loads and fails loudly without a GPU, and it is test infrastructure only."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "integration", "_build", "libgl_glx.so")
HAVE_REF = os.path.isdir("/root/reference/graphlearn/src")


def test_integration_md_quotes_the_compiled_sources():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "sync_integration_md.py"), "--check"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    assert "4 snippet(s) in sync" in r.stdout


@pytest.mark.skipif(not HAVE_REF, reason="needs /root/reference")
def test_integration_library_builds_from_the_reference_tree_and_is_up_to_date():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "integration"), "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert os.path.exists(LIB)
    # the five replaced reference TUs are NOT in the library; the registry / factory / requests / storages ARE
    objs = subprocess.run(["find", os.path.join(ROOT, "integration", "_build", "obj", "ref"), "-name", "*.o"],
                          stdout=subprocess.PIPE, text=True).stdout
    for gone in ("random_sampler.o", "random_without_replacement_sampler.o", "edge_weight_sampler.o", "topk_sampler.o",
                 "/aggregator.o"):
        assert gone not in objs, gone
    for kept in ("op_registry.o", "op_factory.o", "sampling_request.o", "aggregating_request.o", "memory_graph_storage.o",
                 "sum_aggregator.o", "mean_aggregator.o", "full_sampler.o"):
        assert kept in objs, kept


@pytest.mark.skipif(not os.path.exists(LIB), reason="integration/_build not built")
def test_integration_library_links_libglx_only_and_nothing_of_the_product_links_it():
    out = subprocess.run(["ldd", LIB], stdout=subprocess.PIPE, text=True).stdout
    assert "libglx.so" in out and "libglref" not in out and "libglx_oracle" not in out and "libglx_host" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "graph-learn_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".cc", ".cpp", ".py")) or f == "Makefile":
                assert "libgl_glx" not in open(os.path.join(root, f)).read(), f


def _no_gpu():
    import glx
    n = ctypes.c_int(-1)
    return glx.lib().glx_device_count(ctypes.byref(n)) != 0


@pytest.mark.skipif(not os.path.exists(LIB), reason="integration/_build not built")
def test_glx_backed_operators_fail_loudly_without_a_gpu(tmp_path):
    """Without a GPU the glx-backed operators return the C-ABI's UNAVAILABLE through the reference's Status; operators
    left to the reference's own bodies (FullSampler) still answer.  In a process of its own (one driver library per
    process)."""
    if not _no_gpu():
        pytest.skip("a GPU is visible")
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from oracle_bindings import RefLib, _p
ref = RefLib()
ref.add_edges("e", np.array([1, 1, 2], np.int64), np.array([5, 6, 7], np.int64), np.array([.5, .25, 1.], np.float32))
nbr = np.zeros((2, 2), np.int64); eid = np.zeros((2, 2), np.int64)
src = np.array([1, 2], np.int64)
rc = ref.L.glref_sample(ref.h, b"e", b"TopkSampler", _p(src), 2, 2, _p(nbr), _p(eid), 0)
deg, fn, fe = ref.sample_full("e", src, 5)
print("RC", rc, "FULL", deg.tolist(), sorted(fn.tolist()))
''' % os.path.join(ROOT, "tests")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GLX_REF_LIB=LIB), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "RC 14 FULL [2, 1] [5, 6, 7]" in r.stdout, r.stdout[-500:]  # 14 = UNAVAILABLE: no CPU path inside glx

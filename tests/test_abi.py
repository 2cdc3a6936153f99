"""CPU tests of the drop-in boundary: libglx.so loads, exports every symbol that
include/glx.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

import glx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "glx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    assert set(syms) == set(glx.EXPORTS), set(syms) ^ set(glx.EXPORTS)


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(glx.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), "libglx.so does not export %s" % s
    assert L.glx_abi_version() == 5


def test_product_does_not_link_oracle():
    """The product library must not depend on anything under oracle/."""
    import subprocess
    out = subprocess.run(["ldd", glx.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out and "glref" not in out
    nm = subprocess.run(["nm", "-D", glx.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    assert "glxo_" not in nm and "glref_" not in nm
    for root, _, files in os.walk(os.path.join(ROOT, "graph-learn_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".cc", ".cpp", ".py")):
                src = open(os.path.join(root, f)).read()
                assert "glx_oracle" not in src and "oracle_bindings" not in src, f


def _no_gpu():
    n = ctypes.c_int(-1)
    rc = glx.lib().glx_device_count(ctypes.byref(n))
    return rc != 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_fails_loudly_without_gpu():
    n = ctypes.c_int(-1)
    rc = glx.lib().glx_device_count(ctypes.byref(n))
    assert rc == 14 and n.value == 0  # UNAVAILABLE
    assert b"no CPU fallback" in glx.lib().glx_last_error()
    rp = np.array([0, 1], np.int64)
    with pytest.raises(glx.GlxError) as e:
        glx.Graph(rp, np.array([0], np.int64), np.array([0], np.int64))
    assert e.value.code == 14
    with pytest.raises(glx.GlxError):
        glx.Features(np.zeros((2, 4), np.float32))


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_distributed_entry_points_fail_loudly_without_gpu():
    """The communicator / distributed store / request plan are part of the same library: without a GPU they
    report UNAVAILABLE like everything else (librccl itself loads: a unique id can be made on any box)."""
    L = glx.lib()
    h = ctypes.c_void_p()
    assert L.glx_comm_init_local(7, 0, 0, 1, ctypes.byref(h)) == 14 and not h.value
    uid = ctypes.create_string_buffer(128)
    rc = L.glx_comm_unique_id(uid)
    assert rc in (0, 14)
    if rc == 0:
        assert L.glx_comm_init_rccl(0, 0, 1, uid, ctypes.byref(h)) == 14 and not h.value
    assert L.glx_dist_store_create(None, None, None, ctypes.byref(h)) == 3  # INVALID_ARGUMENT: no communicator
    assert L.glx_plan_create(None, 1, 0, None, 1, 1, 0, 0, None, 0, 0.0, ctypes.byref(h)) == 3
    a = np.zeros(16, np.int64)
    assert L.glx_host_register(ctypes.c_void_p(a.ctypes.data), a.nbytes) == 14


def test_argument_validation_needs_no_gpu():
    L = glx.lib()
    assert L.glx_graph_info(None, None, None, None, None, None) == 3  # INVALID_ARGUMENT
    assert L.glx_features_info(None, None, None, None, None) == 3
    assert L.glx_sample(None, 0, None, 1, 1, 1, 0, 0, 0, None, None, 0, None) == 3
    assert L.glx_aggregate(None, 0, None, None, 0, 0, 0.0, None, None, 0, None) == 3
    assert b"NULL" in L.glx_last_error()


def test_every_entry_point_cites_the_reference_interface_it_replaces():
    """include/glx.h is the drop-in boundary: each block of entry points must say which reference file:line it
    stands in for (or that it is an addition of this engine)."""
    text = open(os.path.join(ROOT, "include", "glx.h")).read()
    # sections start at the "/* ---- title ..." comments; a citation anywhere in a section's comments covers the
    # entry points declared in that section
    starts = [m.start() for m in re.finditer(r"/\* ----", text)] + [len(text)]
    cited = set()
    for a, b in zip(starts[:-1], starts[1:]):
        section = text[a:b]
        comments = " ".join(re.findall(r"/\*.*?\*/", section, flags=re.S))
        names = re.findall(r"\b(glx_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", section, flags=re.S))
        if re.search(r"\w+\.(?:cc|h|py):\d+", comments):
            cited.update(names)
    uncited = [s for s in declared_symbols() if s not in cited]
    # housekeeping entry points that have no counterpart in the reference
    allowed = {"glx_abi_version", "glx_device_count", "glx_last_error", "glx_graph_destroy", "glx_graph_info",
               "glx_features_destroy", "glx_features_info", "glx_negative_destroy", "glx_negative_info",
               "glx_profile_collect"}
    assert set(uncited) <= allowed, sorted(set(uncited) - allowed)


def test_host_mirror_registers_every_reference_operator_name():
    """Every registry name of the reference (REGISTER_OPERATOR in graphlearn/src/core/operator/**: 27 of them, from the
    samplers down to the loader's UpdateEdges / UpdateNodes) -- a request by any of these names must find an operator
    (OpFactory::Create, op_factory.cc:36-66)."""
    want = {"RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "TopkSampler", "InDegreeSampler",
            "FullSampler", "RandomNegativeSampler", "InDegreeNegativeSampler", "SoftInDegreeNegativeSampler",
            "NodeWeightNegativeSampler", "ConditionalNegativeSampler", "SubGraphSampler", "RandomWalk",
            "SumAggregator", "MeanAggregator", "MaxAggregator", "MinAggregator", "ProdAggregator",
            "LookupNodes", "LookupEdges", "GetNodes", "GetEdges", "GetDegree", "GetCount", "GetStats",
            "UpdateEdges", "UpdateNodes"}
    have = set()
    src_dir = os.path.join(ROOT, "graph-learn_amd", "host", "src")
    for f in os.listdir(src_dir):
        text = open(os.path.join(src_dir, f)).read()
        have.update(re.findall(r'REGISTER_OPERATOR\("(\w+)"', text))
        have.update(re.findall(r"DEFINE_AGGREGATOR\((\w+),", text))
    assert want <= have, sorted(want - have)
    if os.path.isdir("/root/reference/graphlearn/src/core/operator"):
        ref = set()
        for root, _, files in os.walk("/root/reference/graphlearn/src/core/operator"):
            for f in files:
                if f.endswith(".cc"):
                    ref.update(re.findall(r'REGISTER_OPERATOR\(\s*"(\w+)"', open(os.path.join(root, f)).read()))
        assert ref == want, sorted(ref ^ want)

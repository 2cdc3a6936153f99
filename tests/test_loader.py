"""CPU tests of the data-format side of the path: the TSV loader's
primitives against golden vectors produced by the reference's own code
(tests/golden/loader.json: common/base/hash.cc Hash64, core/io/parser.cc ParseAttribute,
generated through oracle/_ref by tests/golden/make_golden.py), the schema check of
edge_loader.cc:110-141 / node_loader.cc, and the Python surface around them."""
import ctypes
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import graphlearn as gl  # noqa: E402
import glx  # noqa: E402
import pyapi_fixture as fx  # noqa: E402
from oracle_bindings import RefLib, have_ref  # noqa: E402

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "loader.json")))
PYWRAP_TYPES = [gl.pywrap.DataType.INT32, gl.pywrap.DataType.INT64, gl.pywrap.DataType.FLOAT,
                gl.pywrap.DataType.DOUBLE, gl.pywrap.DataType.STRING]


def _no_gpu():
    n = ctypes.c_int(-1)
    return glx.lib().glx_device_count(ctypes.byref(n)) != 0


def test_hash64_matches_reference_golden():
    for hexed, want in GOLD["hash64"]:
        assert gl.pywrap.hash64(bytes.fromhex(hexed)) == int(want), hexed


def _parse(case):
    info = gl.pywrap.AttributeInfo()
    info.delimiter = case["delimiter"]
    for t in case["types"]:
        info.append_type(PYWRAP_TYPES[t])
    for b in case["hash_buckets"] or []:
        info.append_hash_bucket(b)
    return gl.pywrap.parse_attribute(bytes.fromhex(case["data"]), info)


def test_parse_attribute_matches_reference_golden():
    for case in GOLD["parse_attribute"]:
        code, ints, floats, strings = _parse(case)
        assert code == case["code"], case
        if code != 0:
            continue
        assert list(ints) == case["ints"], case
        assert [int(x) for x in np.asarray(floats, np.float32).view(np.uint32)] == case["floats_bits"], case
        assert [s.hex() for s in strings] == case["strings"], case


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_loader_primitives_live_reference():
    ref = RefLib()
    rng = np.random.default_rng(5)
    try:
        for _ in range(300):
            b = bytes(rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8).tolist())
            assert gl.pywrap.hash64(b) == ref.hash64(b)
        alphabet = b"0123456789.-e:, ax"
        for _ in range(400):
            n = int(rng.integers(1, 5))
            types = [int(t) for t in rng.integers(0, 5, n)]
            buckets = [int(x) for x in rng.integers(0, 4, n)] if rng.random() < 0.5 else None
            data = bytes(alphabet[i] for i in rng.integers(0, len(alphabet), int(rng.integers(0, 20))))
            want = ref.parse_attribute(data, ":,", types, buckets)
            got = _parse(dict(data=data.hex(), delimiter=":,", types=types, hash_buckets=buckets))
            assert got[0] == want[0], (data, types)
            if want[0] == 0:
                assert list(got[1]) == list(want[1]) and [s for s in got[3]] == list(want[3]), (data, types)
                assert np.array_equal(np.asarray(got[2], np.float32).view(np.uint32), want[2].view(np.uint32))
        # the loader's fast number paths must be EXACTLY strtof / strtol: decimal strings of every
        # shape (up to 17 significant digits, values near float rounding midpoints included)
        for _ in range(6000):
            nd = int(rng.integers(1, 18))
            digits = "".join(str(int(x)) for x in rng.integers(0, 10, nd))
            cut = int(rng.integers(0, nd + 1))
            text = ("-" if rng.random() < 0.3 else "") + digits[:cut] + ("." + digits[cut:] if rng.random() < 0.8 else digits[cut:])
            data = (text + ":" + text.replace(".", "")[:18] or "0").encode()
            want = ref.parse_attribute(data, ":", [2, 1])
            got = _parse(dict(data=data.hex(), delimiter=":", types=[2, 1], hash_buckets=None))
            assert got[0] == want[0], text
            if want[0] == 0:
                assert list(got[1]) == list(want[1]), text
                assert np.array_equal(np.asarray(got[2], np.float32).view(np.uint32), want[2].view(np.uint32)), text
        for base in (16777217, 33554433, 8388609.5, 1.00000005960464477539, 0.100000001490116119384765625):
            for text in (repr(base), "%.17g" % base, "%.9f" % base, "%.15g" % base):
                data = text.encode()
                want = ref.parse_attribute(data, ":", [2])
                got = _parse(dict(data=data.hex(), delimiter=":", types=[2], hash_buckets=None))
                assert got[0] == want[0] == 0 and np.array_equal(
                    np.asarray(got[2], np.float32).view(np.uint32), want[2].view(np.uint32)), text
    finally:
        ref.close()


def test_decoder_counts_and_format_bits():
    d = gl.Decoder(weighted=True, labeled=True, attr_types=fx.ATTR_TYPES)
    # ('string', 10) is hashed into an int attribute (parser.h:50-57)
    assert (d.int_attr_num, d.float_attr_num, d.string_attr_num) == (2, 1, 1)
    assert d.data_format == 2 + 4 + 16 and d.has_property and d.attributed
    assert gl.Decoder().data_format == 0 and not gl.Decoder().has_property
    assert gl.Decoder(attr_types=["int", ("string", 8, True)]).string_attr_num == 1  # multi-valued stays a string
    with pytest.raises(ValueError):
        gl.Decoder(attr_types=[("int", 3, True)])
    assert gl.strategy2op("random_without_replacement", "Sampler") == "RandomWithoutReplacementSampler"
    assert [gl.get_mask_type("u", m) for m in gl.Mask] == ["u", "MASK*u", "MASK**u", "MASK***u"]


def test_schema_mismatch_is_invalid_argument(tmp_path):
    """The header must spell the optional columns the decoder declares, in order."""
    weighted = fx.write_nodes(str(tmp_path), "w_nodes", (0, 4), [fx.WEIGHTED])
    g = gl.Graph().node(weighted, "n", gl.Decoder(attr_types=["int"]))
    with pytest.raises(gl.InvalidArgumentError, match="Invalid node table schema"):
        g.init()
    g.close()
    edges = fx.write_edges(str(tmp_path), "e", (0, 4), (10, 20), [fx.WEIGHTED])
    g = gl.Graph().edge(edges, ("a", "b", "e"), gl.Decoder(weighted=True, labeled=True))
    with pytest.raises(gl.InvalidArgumentError, match="Invalid edge table schema"):
        g.init()
    g.close()
    g = gl.Graph().node(os.path.join(str(tmp_path), "missing"), "n", gl.Decoder())
    with pytest.raises(gl.NotFoundError):
        g.init()
    g.close()


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_python_api_fails_loudly_without_gpu(tmp_path):
    """Parsing is host work, the store lives in HBM: without a GPU init() raises, nothing falls back."""
    nodes = fx.write_nodes(str(tmp_path), "nodes", (0, 10), [fx.ATTRIBUTED])
    edges = fx.write_edges(str(tmp_path), "edges", (0, 10), (0, 10), [fx.WEIGHTED])
    g = gl.Graph().node(nodes, "n", gl.Decoder(attr_types=fx.ATTR_TYPES)) \
        .edge(edges, ("n", "n", "e"), gl.Decoder(weighted=True))
    with pytest.raises(gl.UnavailableError, match="no CPU fallback"):
        g.init()
    g.close()


def test_deploy_modes():
    """The reference's client/server RPC deploy mode is not served; init(task_index, task_count) is the SPMD mode
    (one process per GPU, every process keeps its shard of the sources)."""
    with pytest.raises(NotImplementedError):
        gl.Graph().init(cluster={"server_count": 1, "client_count": 1})
    with pytest.raises(NotImplementedError):
        gl.Graph().init(hosts="127.0.0.1:8888")
    with pytest.raises(ValueError):
        gl.Graph().init(task_index=2, task_count=2)
    g = gl.Graph().init(task_index=1, task_count=2)  # nothing to load: no device needed
    with pytest.raises(RuntimeError):
        g.sharded_store("e")  # no torch.distributed process group
    g.close()


def test_package_exports_the_reference_names():
    """graphlearn/__init__.py + python/data/__init__.py + python/nn/__init__.py of the reference: the names user code
    imports from the package root resolve here too (the RPC deploy helpers aside)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
    import numpy as np
    import graphlearn as gl
    for name in ("Graph", "Dataset", "Decoder", "Topology", "Values", "Nodes", "Edges", "SparseNodes", "SparseEdges", "Layer", "Layers",
                 "SubGraph", "IndexOption", "pywrap", "nn", "EDGE_SRC", "EDGE_DST", "NODE", "REPLICATE", "CIRCULAR", "Mask",
                 "OutOfRangeError", "set_padding_mode", "set_shuffle_buffer_size", "set_default_neighbor_id"):
        assert hasattr(gl, name), name
    assert gl.nn.Dataset is not None and gl.nn.Data is not None
    import graphlearn.python.nn as ref_path
    assert ref_path.SubGraph is gl.nn.SubGraph and ref_path.HeteroSubGraph is gl.nn.HeteroSubGraph
    sg = gl.nn.SubGraph(np.array([[0, 1, 2], [2, 3, 4]]), gl.nn.Data(ids=np.arange(5)), y=np.ones(5))
    assert sg.num_nodes == 5 and sg.num_edges == 3 and "y" in sg.keys and sg["nope"] is None
    hg = gl.nn.HeteroSubGraph({("user", "click", "item"): np.zeros((2, 4), np.int64)},
                              {"user": np.arange(3), "item": gl.nn.Data(ids=np.arange(7))})
    assert hg.num_nodes("item") == 7 and hg.num_nodes("user") == 3 and hg.num_edges(("user", "click", "item")) == 4
    assert hg.node_types == ["user", "item"] and hg.edge_types == [("user", "click", "item")]


def test_error_and_topology_surface():
    """python/errors.py: BaseError, one class per status code incl. RequestStopError, and the two lookups between codes and
    classes; python/data/topology.py: EdgeInfo, get_edge_info, print_one, a conflicting re-declaration refused."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
    import pytest
    import graphlearn as gl
    from graphlearn import errors
    assert issubclass(gl.OutOfRangeError, gl.BaseError) and issubclass(gl.RequestStopError, errors.BaseError)
    for name in ("CANCELLED", "OUT_OF_RANGE", "UNAVAILABLE", "REQUEST_STOP", "DATA_LOSS"):
        code = getattr(gl.pywrap.ErrorCode, name)
        cls = errors.exception_type_from_error_code(code)
        assert errors.error_code_from_exception_type(cls) == code
    assert errors.exception_type_from_error_code(gl.pywrap.ErrorCode.OUT_OF_RANGE) is gl.OutOfRangeError
    e = gl.NotFoundError("nope", gl.pywrap.ErrorCode.NOT_FOUND)
    assert e.message == "nope" and e.error_code == gl.pywrap.ErrorCode.NOT_FOUND and str(e) == "nope"
    t = gl.Topology()
    t.add("buy", "user", "item")
    t.add("buy", "user", "item")  # the same declaration again (a second source file of the type)
    with pytest.raises(ValueError):
        t.add("buy", "item", "user")
    info = t.get_edge_info("buy")
    assert (info.src_type, info.dst_type) == ("user", "item") == (t.get_src_type("buy"), t.get_dst_type("buy"))
    assert t.is_exist("buy") and not t.is_exist("sell")
    with pytest.raises(ValueError):
        t.get_edge_info("sell")
    t.print_one("buy")
    with pytest.warns(UserWarning):
        t.print_one("sell")

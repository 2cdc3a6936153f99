"""The reference's Python layer (graphlearn/__init__.py + graphlearn/python/, staged unchanged by
scripts/stage_refpy.py) finds every `pywrap.<name>` it uses in this engine's pywrap_graphlearn module, and imports
on it.  CPU: nothing here launches a kernel.  VERDICT r04 'Next round' item 1 (198 names, 85 were missing)."""
import os
import re
import subprocess
import sys

import pytest

import refpy

pytestmark = pytest.mark.skipif(not refpy.staged(), reason="reference python layer not staged (scripts/stage_refpy.py)")

# Names that appear only in the reference's docstrings and exist in no pywrap module, the reference's included
# (python/errors.py:32 "`pywrap.ErrorCode.Code`", python/graph.py:951 "Construct pywrap.Source").
DOCSTRING_ONLY = {"pywrap.ErrorCode.Code", "pywrap.Source"}


def _names():
    found = set()
    for base, _, files in os.walk(refpy.PACKAGE):
        for f in files:
            if f.endswith(".py"):
                with open(os.path.join(base, f)) as fh:
                    found |= set(re.findall(r"pywrap\.\w+(?:\.\w+)?", fh.read()))
    return sorted(found - DOCSTRING_ONLY)


def test_every_pywrap_name_the_reference_python_uses_exists():
    names = _names()
    assert len(names) >= 190, "the staged tree looks incomplete: %d names" % len(names)
    code = (
        "import sys; from graphlearn import pywrap_graphlearn as pw\n"
        "missing = []\n"
        "for n in sys.argv[1:]:\n"
        "    o = pw\n"
        "    for part in n.split('.')[1:]:\n"
        "        if not hasattr(o, part): missing.append(n); break\n"
        "        o = getattr(o, part)\n"
        "print('MISSING', missing)\n")
    out = subprocess.run([sys.executable, "-c", code] + names, env=refpy.env(), cwd="/tmp", stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout
    assert "MISSING []" in out.stdout, out.stdout


def test_reference_package_imports_on_this_module_and_builds_a_query():
    """`import graphlearn` is the reference's package; the flag setters, the key constants and the DAG definition
    calls a GSL query makes all resolve (the query is only defined here -- running it needs the GPU)."""
    code = (
        "import graphlearn as gl, os\n"
        "assert os.path.realpath(gl.__file__).startswith(os.path.realpath(%r)), gl.__file__\n"
        "from graphlearn import pywrap_graphlearn as pw\n"
        "assert (pw.kNodeIds, pw.kSrcIds, pw.kFloatAttrKey, pw.kFilterValues, pw.kDegrees) == ('nid', 'sid', 'fa', 'filt', 'dg')\n"
        "pw.set_deploy_mode(pw.DeployMode.LOCAL); pw.set_tape_capacity(4); pw.set_dataset_capacity(3)\n"
        "pw.set_tracker('x'); pw.set_server_hosts('h'); pw.set_knn_metric(1)\n"
        "d = pw.new_dag(); pw.set_dag_id(d, 7)\n"
        "e = pw.new_dag_edge(); pw.set_dag_edge_id(e, 1); pw.set_dag_edge_src_output(e, 'nid'); pw.set_dag_edge_dst_input(e, 'sid')\n"
        "n = pw.new_dag_node(); pw.set_dag_node_id(n, 1); pw.set_dag_node_op_name(n, 'GetNodes')\n"
        "pw.add_dag_node_string_params(n, pw.kNodeType, 'u'); pw.add_dag_node_int_params(n, pw.kBatchSize, 4)\n"
        "pw.add_dag_node_int_vector_params(n, pw.kNeighborCount, [1, 2]); pw.add_dag_node_float_vector_params(n, pw.kSideInfo, [1.0, 2.0])\n"
        "pw.add_dag_node_out_edge(n, e); pw.add_dag_node(d, n)\n"
        "s = pw.debug_string(d)\n"
        "assert 'op_name: \"GetNodes\"' in s and 'id: 7' in s and 'src_output: \"nid\"' in s, s\n"
        "try:\n"
        "    pw.rpc_client(0, True); raise SystemExit('rpc_client did not fail')\n"
        "except RuntimeError: pass\n"
        "print('OK')\n" % refpy.STAGE)
    out = subprocess.run([sys.executable, "-c", code], env=refpy.env(), cwd="/tmp", stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout

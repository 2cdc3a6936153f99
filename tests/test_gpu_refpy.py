"""The reference's own Python unit tests -- graphlearn/python/sampler/tests, gsl/tests and python/tests, staged
UNCHANGED by scripts/stage_refpy.py together with the reference's whole Python layer -- run on this engine's
pywrap_graphlearn module: Graph.init() loads the TSV sources into HBM, the samplers are the HIP kernels, and every
GSL query is a DAG compiled and run by the host layer's scheduler (dag.h).  SURVEY 8(f)-3 as the survey states it:
"lets GL/python/sampler/tests/* ... run against the new engine unchanged".  One file = one process = one working
directory, like the reference's test_python_ut.sh."""
import pytest

import refpy

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not refpy.staged(), reason="reference python layer not staged (scripts/stage_refpy.py)")]

FILES = refpy.test_files() if refpy.staged() else []


@pytest.mark.parametrize("rel", FILES)
def test_reference_python_test_file(rel, tmp_path):
    run, reason = refpy.KNOWN_BROKEN_IN_REFERENCE.get(rel, ((), None))
    if run is None:
        pytest.skip("broken in the reference itself: " + reason)
    out = refpy.run_file(rel, str(tmp_path), tests=run)
    assert out.returncode == 0, "%s\n%s" % (rel, out.stdout[-6000:])
    if refpy.defines_tests(rel):  # a few files are base classes or entirely commented out (test_random_node_subgraph_sampling.py)
        # unittest's own verdict: "OK" (possibly with skips) after "Ran N tests"
        assert "\nOK" in out.stdout and "FAILED" not in out.stdout, out.stdout[-3000:]


def test_the_reference_layer_really_ran_on_this_engine(tmp_path):
    """Guards the harness itself: the process that ran the reference's tests had the REFERENCE's graphlearn package
    and THIS engine's libglx.so mapped."""
    import subprocess
    import sys
    code = ("import graphlearn as gl, os\n"
            "maps = open('/proc/self/maps').read()\n"
            "assert 'libglx.so' in maps and 'libglx_host.so' in maps\n"
            "assert '_refpy' in gl.__file__ and hasattr(gl.Graph, 'V')\n"
            "print('OK')\n")
    out = subprocess.run([sys.executable, "-c", code], env=refpy.env(), cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout


def _run_script(name, args, cwd):
    import os
    import subprocess
    import sys
    script = os.path.join(refpy.ROOT, "tests", "scripts", name)
    out = subprocess.run([sys.executable, script] + list(args), env=refpy.env(), cwd=cwd, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-4000:]
    return out.stdout


def test_fused_hop_chain_of_a_query_equals_the_separate_requests(tmp_path):
    """The DAG runner's hop fusion (a chain of dense sampling nodes -> one glx_sample_hops call, host dag.h) against the
    same hops as separate SamplingRequests, both driven by the reference's Python layer: deterministic (topk) in one
    process; random across two fresh processes, whose operators' call counters start at zero alike."""
    for sub in ("t", "d1", "d2"):
        (tmp_path / sub).mkdir()
    assert "TOPK_OK" in _run_script("refpy_dag_fusion.py", ["topk"], str(tmp_path / "t"))
    fused = [l for l in _run_script("refpy_dag_fusion.py", ["dag"], str(tmp_path / "d1")).splitlines() if l.startswith("VALUES")]
    apart = [l for l in _run_script("refpy_dag_fusion.py", ["direct"], str(tmp_path / "d2")).splitlines() if l.startswith("VALUES")]
    assert fused and fused == apart, (fused, apart)

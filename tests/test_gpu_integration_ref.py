"""The drop-in inside graphlearn::op itself: integration/_build/libgl_glx.so is the reference's
OWN OpRegistry / OpFactory / RequestFactory / SamplingRequest / AggregatingRequest / storages (compiled from
/root/reference by integration/Makefile) with the glx-backed operator bodies of integration/src/ -- INTEGRATION.md's
snippets, compiled -- behind Operator::Process.

  * integration_unittest: the cases of the reference's sampler_unittest.cpp:76-273 and
    aggregating_op_unittest.cpp:237-357 against that library;
  * the same seeded workload through that library and through oracle/_ref/libglref.so (the reference as it is), each
    in a process of its own: deterministic operators (Topk, with and without filters, the five aggregators, Stitch,
    FullSampler) must agree bit for bit; the random samplers must equal the oracle's L2 contract (glx_oracle.c) under
    the same (seed, call counter) on the CSR the reference's own storage exports.
Both libraries are prebuilt (they need /root/reference) and travel to the GPU box like oracle/_ref."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "integration", "_build", "libgl_glx.so")
EXE = os.path.join(ROOT, "integration", "_build", "integration_unittest")
REF = os.path.join(ROOT, "oracle", "_ref", "libglref.so")
needs_build = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(EXE) and os.path.exists(REF)),
                                 reason="integration/_build or oracle/_ref not prebuilt (they need /root/reference)")


@needs_build
def test_reference_operator_unit_tests_pass_on_the_glx_backed_registry():
    r = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "13 test(s), 0 failure(s)" in r.stdout, r.stdout[-1500:]
    for name in ("SamplerTest.Topk", "SamplerTest.RandomWithoutReplacement", "AggregationOpTest.MeanAggregator"):
        assert "[  OK  ] " + name in r.stdout


def _run(lib, path, seed):
    env = dict(os.environ, GLX_REF_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scripts", "integration_run.py"), path, str(seed)], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return dict(np.load(path))


@needs_build
@pytest.mark.parametrize("seed", [1, 2])
def test_glx_backed_reference_registry_agrees_with_the_reference_and_the_oracle(tmp_path, seed):
    got = _run(LIB, str(tmp_path / "glx.npz"), seed)
    ref = _run(REF, str(tmp_path / "ref.npz"), seed)
    assert str(got["library"]) == "libgl_glx.so" and str(ref["library"]) == "libglref.so"
    # the storages are the reference's in both: same post-Build adjacency
    for k in ("rows", "row_ptr", "col", "eid", "weight"):
        assert np.array_equal(got[k], ref[k]), k
    beq = lambda a, b: np.array_equal(a.view(np.uint32), b.view(np.uint32))  # noqa: E731
    for tag in ("pad1_", "pad0_"):
        # deterministic operators: bit for bit the reference's answers
        for k in ("TopkSampler_k3", "TopkSampler_k40", "topk_flt_id", "topk_flt_ts"):
            assert np.array_equal(got[tag + k + "_nbr"], ref[tag + k + "_nbr"]), (tag, k)
            assert np.array_equal(got[tag + k + "_eid"], ref[tag + k + "_eid"]), (tag, k)
        for k in ("full_deg", "full_nbr", "full_eid"):
            assert np.array_equal(got[tag + k], ref[tag + k]), (tag, k)
        for name in ("SumAggregator", "MeanAggregator", "MaxAggregator", "MinAggregator", "ProdAggregator"):
            assert np.array_equal(got[tag + name + "_cnt"], ref[tag + name + "_cnt"]), (tag, name)
            assert beq(got[tag + name + "_emb"], ref[tag + name + "_emb"]), (tag, name)
        for name in ("SumAggregator", "MeanAggregator", "MaxAggregator"):
            assert beq(got[tag + name + "_stitch_emb"], ref[tag + name + "_stitch_emb"]), (tag, name)
            assert np.array_equal(got[tag + name + "_stitch_cnt"], ref[tag + name + "_stitch_cnt"]), (tag, name)
    # random operators: the oracle's contract under the stream the run set (seed 1234, call counters 10, 11, ... in
    # call order), on the CSR exported from the reference's storage
    orc = Oracle()
    g = dict(row_ptr=got["row_ptr"], col=got["col"], eid=got["eid"], weight=got["weight"], ids=got["rows"])
    g["alias"] = orc.alias_build(g["row_ptr"], g["weight"])
    sys.path.insert(0, os.path.join(ROOT, "tests", "scripts"))
    from integration_run import workload
    seeds = workload(seed)["seeds"]
    for tag, padding in (("pad1_", 1), ("pad0_", 0)):
        cc = 10
        for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "TopkSampler"):
            for k in (3, 40):
                want_n, want_e = orc.sample(g, name, seeds, k, seed=1234, call_counter=cc, padding_mode=padding,
                                            default_neighbor_id=-3)
                assert np.array_equal(got[tag + "%s_k%d_nbr" % (name, k)], want_n), (tag, name, k)
                assert np.array_equal(got[tag + "%s_k%d_eid" % (name, k)], want_e), (tag, name, k)
                cc += 1
                # and against the reference's own random draws: same support per row (SURVEY 8(c) acceptance)
                if name != "TopkSampler" and tag + "%s_k%d_nbr" % (name, k) in ref:
                    ref_n = ref[tag + "%s_k%d_nbr" % (name, k)]
                    for i in range(0, seeds.shape[0], 17):
                        lo, hi = None, None
                        r = np.searchsorted(got["rows"], seeds[i])
                        if r < got["rows"].shape[0] and got["rows"][r] == seeds[i]:
                            lo, hi = got["row_ptr"][r], got["row_ptr"][r + 1]
                        row = set(got["col"][lo:hi].tolist()) if lo is not None else set()
                        allowed = row | {-3}
                        assert set(want_n[i].tolist()) <= allowed and set(ref_n[i].tolist()) <= allowed, (name, k, i)

"""CPU tests: the oracle's SubGraphSampler restatement (oracle/glx_oracle.c glxo_subgraph_induce / glxo_subgraph_dist,
tests/oracle_bindings.py Oracle.subgraph) against golden outputs of the reference's own operator
(core/operator/subgraph/subgraph_sampler.{h,cc}, subgraph_utils.cc; tests/golden/subgraph.npz made by
make_golden.py from oracle/_ref) and, when the reference library is here, live against it."""
import os

import numpy as np
import pytest

from oracle_bindings import Oracle, RefLib, have_ref

GOLD = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "subgraph.npz")))
CASES = [str(c) for c in GOLD["cases"]]


def graph():
    return dict(row_ptr=GOLD["row_ptr"], col=GOLD["col"], eid=GOLD["eid"], weight=GOLD["w_slot"], ids=GOLD["rows"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_equals_reference_golden(case):
    need_dist = case + "_dist_src" in GOLD
    r = Oracle().subgraph(graph(), GOLD[case + "_seeds"], GOLD[case + "_num_nbrs"], full_nbr_num=int(GOLD[case + "_full"]),
                          need_dist=need_dist)
    for k in ("nodes", "row", "col", "eid") + (("dist_src", "dist_dst") if need_dist else ()):
        assert np.array_equal(r[k], GOLD[case + "_" + k]), (case, k)


def test_later_slot_wins_for_multi_edges():
    # node 0 -> 1 twice (edge ids 7 then 9): node2edge[1] ends up 9 (subgraph_sampler.cc:60-64)
    nodes = np.array([0, 1], np.int64)
    off = np.array([0, 2, 2], np.int64)
    row, col, eid = Oracle().subgraph_induce(nodes, off, np.array([1, 1], np.int64), np.array([7, 9], np.int64))
    assert row.tolist() == [0, 1] and col.tolist() == [1, 0] and eid.tolist() == [9, 9]


@pytest.mark.skipif(not have_ref(), reason="reference library not built")
def test_oracle_equals_live_reference_on_random_requests():
    import os
    rng = np.random.default_rng(3 + int(os.environ.get("GLX_FUZZ_FIRST", "0")))
    trials = 40 * max(1, int(os.environ.get("GLX_FUZZ_CASES", "1")))  # GLX_FUZZ_CASES=100: 4,000 random requests
    ref = RefLib(storage_mode=2)
    try:
        ref.add_edges("sub", GOLD["src"], GOLD["dst"], GOLD["w"])
        ref.set_flags(1, 0, 0.0)
        orc = Oracle()
        for _ in range(trials):
            seeds = rng.integers(0, 64, int(rng.integers(1, 9))).astype(np.int64)
            nn = [int(rng.integers(0, 7))]
            full = int(rng.choice([1, 2, 5, 100]))
            dist = bool(rng.integers(0, 2)) and seeds.shape[0] >= 2
            a = ref.subgraph("sub", seeds, nn, full_nbr_num=full, need_dist=dist)
            b = orc.subgraph(graph(), seeds, nn, full_nbr_num=full, need_dist=dist)
            for k in a:
                assert np.array_equal(a[k], b[k]), (seeds, nn, full, k)
    finally:
        ref.close()

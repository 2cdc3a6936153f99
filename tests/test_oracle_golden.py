"""CPU tests: pin the oracle (oracle/glx_oracle.c) against the reference.

  * golden vectors generated from the real reference (tests/golden/*.npz, made
    by tests/golden/make_golden.py from oracle/_ref) and the reference's own
    known-answer tests (sampler_unittest.cpp, aggregating_op_unittest.cpp,
    python/tests/utils.py topk expectations);
  * when oracle/_ref/libglref.so is present, also live against it.
Deterministic ops must match bit-for-bit; the three random samplers are pinned
distributionally (the reference is unseeded: random_sampler.cc:46-47).
"""
import os

import numpy as np
import pytest
from scipy import stats

from oracle_bindings import AGGREGATORS, SAMPLERS, Oracle, RefLib, have_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def graph_of(g, with_alias=True, orc=None):
    d = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
    if with_alias and orc is not None:
        d["alias"] = orc.alias_build(g["row_ptr"], g["w_slot"])
    return d


def beq(a, b):
    """bit equality for float arrays (NaN-safe)."""
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ------------------------------------------------------------------- Philox ---
def test_philox_known_answers(orc):
    # Random123 kat_vectors, philox4x32-10
    assert list(orc.philox([0, 0, 0, 0], [0, 0])) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(orc.philox([0xffffffff] * 4, [0xffffffff] * 2)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6,
                                                                     0x6d5451fd]
    assert list(orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0])) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_draw_layout(orc):
    # draw j = words {2(j&1), 2(j&1)+1} of block j>>1; key=(seed lo, hi); ctr=(blk,row,cc lo,cc hi)
    seed, cc, row = 0x1122334455667788, 0x99aabbccddeeff00, 77
    for j in range(6):
        o = orc.philox([j >> 1, row, cc & 0xffffffff, cc >> 32], [seed & 0xffffffff, seed >> 32])
        w = (j & 1) * 2
        assert orc.draw64(seed, cc, row, j) == (int(o[w + 1]) << 32) | int(o[w])


# ------------------------------------------------------------- deterministic ---
def test_topk_kat_reference_unittest(orc):
    """sampler_unittest.cpp:190-195: Topk of ids {0,1}, k=2 -> {20,10,21,11}."""
    g = load("kat_sampler.npz")
    og = graph_of(g)
    nbr, eid = orc.sample(og, "TopkSampler", np.array([0, 1], np.int64), 2)
    assert nbr.reshape(-1).tolist() == [20, 10, 21, 11]
    # and the sorted adjacency the oracle builds from the raw edge list matches
    # the reference's post-Build order
    rp = g["row_ptr"]
    order = np.argsort(g["src"], kind="stable")
    col, eid2, w = orc.sort_rows(rp, g["dst"][order], order.astype(np.int64), g["w"][order])
    assert np.array_equal(col, g["col"]) and np.array_equal(eid2, g["eid"]) and beq(w, g["w_slot"])


def test_topk_kat_all_modes(orc):
    g = load("kat_sampler.npz")
    og = graph_of(g)
    for pad in (0, 1):
        for dflt in (0, -1):
            for k in (2, 4):
                nbr, eid = orc.sample(og, "TopkSampler", g["query"], k, padding_mode=pad,
                                      default_neighbor_id=dflt)
                key = "topk_p%d_d%d_k%d" % (pad, dflt + 1, k)
                assert np.array_equal(nbr, g[key + "_nbr"]), key
                assert np.array_equal(eid, g[key + "_eid"]), key


def test_python_fixture_topk(orc):
    """GL/python/sampler/tests/test_topk_neighbor_sampling.py via utils.check_topk_edge_ids."""
    g = load("pyfixture_topk.npz")
    og = graph_of(g)
    for pad in (0, 1):
        nbr, eid = orc.sample(og, "TopkSampler", g["query"], 6, padding_mode=pad, default_neighbor_id=-1)
        assert np.array_equal(nbr, g["topk_p%d_nbr" % pad])
        assert np.array_equal(eid, g["topk_p%d_eid" % pad])
        # the expectation the python test itself computes (utils.py:304-326)
        for i, s in enumerate(g["query"][:3]):
            dsts = sorted([int(s) * it % 100 for it in range(1, int(s) % 5 + 1)], reverse=True)
            real = min(int(s) % 5, 6)
            if pad == 0:
                exp = dsts[:real] + [-1] * (6 - real)
            else:
                exp = (dsts[:real] * 6)[:6]
            assert nbr[i].tolist() == exp
    nbr, eid = orc.sample(og, "RandomWithoutReplacementSampler", g["query"], 6, padding_mode=0,
                          default_neighbor_id=-1)
    assert np.array_equal(nbr, g["rwor_p0_nbr"]) and np.array_equal(eid, g["rwor_p0_eid"])


def test_rand_graph_topk_and_padding(orc):
    g = load("rand_graph.npz")
    og = graph_of(g)
    for pad in (0, 1):
        for k in (1, 3, 10, 33, 70):
            nbr, eid = orc.sample(og, "TopkSampler", g["query"], k, padding_mode=pad, default_neighbor_id=-7)
            assert np.array_equal(nbr, g["topk_p%d_k%d_nbr" % (pad, k)])
            assert np.array_equal(eid, g["topk_p%d_k%d_eid" % (pad, k)])
    for k in (3, 33):
        nbr, eid = orc.sample(og, "RandomWithoutReplacementSampler", g["query"], k, padding_mode=0,
                              default_neighbor_id=-7)
        assert np.array_equal(nbr, g["rwor_p0_k%d_nbr" % k])
        assert np.array_equal(eid, g["rwor_p0_k%d_eid" % k])


def test_row_sort_matches_reference_build(orc):
    """MemoryAdjMatrix::Sort (memory_adj_matrix.cc:105-125) on tie-free weights."""
    g = load("rand_graph.npz")
    rows = g["rows"]
    row_of = {int(v): i for i, v in enumerate(rows)}
    r = np.array([row_of[int(s)] for s in g["src"]])
    order = np.argsort(r, kind="stable")
    rp = np.zeros(rows.shape[0] + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp)
    assert np.array_equal(rp, g["row_ptr"])
    col, eid, w = orc.sort_rows(rp, g["dst"][order], order.astype(np.int64), g["w"][order])
    assert np.array_equal(col, g["col"]) and np.array_equal(eid, g["eid"]) and beq(w, g["w_slot"])


def test_alias_tables_bit_exact(orc):
    """AliasMethod::Build (alias_method.cc:57-107): probs_/alias_ of the reference itself."""
    g = load("rand_graph.npz")
    prob, alias = orc.alias_build(g["row_ptr"], g["w_slot"])
    assert beq(prob, g["alias_prob"])
    assert np.array_equal(alias, g["alias_idx"])


def test_aggregator_kat_reference_unittest(orc):
    """aggregating_op_unittest.cpp:237-364 expectations, verbatim."""
    a = load("agg.npz")
    X = np.arange(100, dtype=np.float32).reshape(100, 1)
    expect = {
        "SumAggregator": [0, 0, 3, 12, 30],
        "MeanAggregator": [0, 0, 1.5, 4, 7.5],
        "MinAggregator": [0, 0, 1, 3, 6],
        "MaxAggregator": [0, 0, 2, 5, 9],
        "ProdAggregator": [0, 0, 2, 60, 3024],
    }
    for name in AGGREGATORS:
        emb, cnt = orc.aggregate(X, name, a["kat_ids"], a["kat_seg"], 5)
        assert emb.reshape(-1).tolist() == [float(x) for x in expect[name]], name
        assert cnt.tolist() == [0, 1, 2, 3, 4]
        assert beq(emb, a["kat_%s_emb" % name]) and np.array_equal(cnt, a["kat_%s_cnt" % name])


def test_aggregators_golden_bit_exact(orc):
    a = load("agg.npz")
    for c in range(int(a["num_cases"])):
        for name in AGGREGATORS:
            emb, cnt = orc.aggregate(a["c%d_X" % c], name, a["c%d_ids" % c], a["c%d_seg" % c],
                                     int(a["c%d_num_segments" % c]), float(a["c%d_default" % c]),
                                     ids=a["c%d_raw" % c])
            assert np.array_equal(cnt, a["c%d_%s_cnt" % (c, name)]), (c, name)
            assert beq(emb, a["c%d_%s_emb" % (c, name)]), (c, name)


def test_max_init_is_minus_37(orc):
    """max_aggregator.cc:28 initialises with FLT_MIN_10_EXP (-37), not -FLT_MAX."""
    X = np.full((4, 2), -100.0, np.float32)
    emb, cnt = orc.aggregate(X, "MaxAggregator", np.array([0, 1], np.int64), np.array([0, 0], np.int32), 1)
    assert emb.tolist() == [[-37.0, -37.0]] and cnt.tolist() == [2]


# --------------------------------------------------------------- distributions ---
def _positions(g, eid_out, r):
    rp = g["row_ptr"]
    pos_of = {int(x): i for i, x in enumerate(g["eid"][rp[r]:rp[r + 1]])}
    return np.vectorize(pos_of.get)(eid_out)


def _two_sample_p(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    keep = (a + b) > 0
    if keep.sum() < 2:
        return 1.0
    return stats.chi2_contingency(np.stack([a[keep], b[keep]]))[1]


@pytest.mark.parametrize("name", SAMPLERS[:3])
@pytest.mark.parametrize("k", [2, 6])
def test_random_samplers_match_reference_distribution(orc, name, k):
    """Per (row, slot) marginals and the slot-(0,1) joint of the contract samplers
    vs the real reference's (40000 requests each; two-sample chi-square)."""
    g = load("dist.npz")
    og = graph_of(g, orc=orc)
    T = int(g["T"])
    degs = g["degs"]
    q = np.tile(g["rows"], T)
    nbr, eid = orc.sample(og, name, q, k, seed=99, call_counter=3)
    eid = eid.reshape(T, len(degs), k)
    ref_hist = g["%s_k%d_hist" % (name, k)]
    ref_pair = g["%s_k%d_pair" % (name, k)]
    for r, d in enumerate(degs):
        pos = _positions(g, eid[:, r, :], r)
        # support: every emitted neighbour belongs to the row (set membership, as
        # sampler_unittest.cpp:116-121,154-159,223-231 assert)
        assert pos.min() >= 0 and pos.max() < d
        for j in range(k):
            h = np.bincount(pos[:, j], minlength=d)
            p = _two_sample_p(h, ref_hist[r, j, :d])
            assert p > 1e-4, (name, k, r, j, p, h, ref_hist[r, j, :d])
        pair = np.zeros((d, d), np.int64)
        np.add.at(pair, (pos[:, 0], pos[:, 1]), 1)
        p = _two_sample_p(pair, ref_pair[r, :d, :d])
        assert p > 1e-4, (name, k, r, "pair", p)
        if name == "RandomWithoutReplacementSampler":
            # circular padding of a permutation: first min(k,d) distinct, then it repeats
            m = min(k, d)
            assert all(len(set(row[:m])) == m for row in pos[:2000])
            assert np.array_equal(pos[:, :k], pos[:, np.arange(k) % m])


def test_alias_draw_never_starts_at_last_slot(orc):
    """alias_method.cc:117 draws from [0, deg-1): with all-equal weights (alias = identity)
    the last slot is unreachable -- a reference quirk the contract keeps."""
    rp = np.array([0, 5], np.int64)
    w = np.ones(5, np.float32)
    g = dict(row_ptr=rp, col=np.arange(5, dtype=np.int64), eid=np.arange(5, dtype=np.int64), weight=w,
             alias=orc.alias_build(rp, w))
    nbr, _ = orc.sample(g, "EdgeWeightSampler", np.zeros(20000, np.int64), 4, seed=1)
    assert set(np.unique(nbr).tolist()) == {0, 1, 2, 3}


# ----------------------------------------------------------- partition / stitch ---
def test_partition_stitch(orc):
    """hash_partitioner.h:90-92 (llabs(id) % P, stable) and stitcher.h:82-93."""
    ids = np.array([0, 1, 2, 3, 4, 5, -1, -2, 7, 9, 8], np.int64)
    order, counts = orc.partition(ids, 2)
    assert counts.tolist() == [5, 6]
    assert order.tolist() == [0, 2, 4, 7, 10, 1, 3, 5, 6, 8, 9]
    rows = np.stack([ids[order] * 10, ids[order] * 10 + 1], 1)
    out = orc.stitch(rows, order)
    assert np.array_equal(out, np.stack([ids * 10, ids * 10 + 1], 1))


# ------------------------------------------------------------ live vs reference ---
@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_csr_mode_and_samplers(orc):
    """StorageMode=3 (the reference's CSR) exposes the same adjacency and results."""
    g = load("rand_graph.npz")
    ref = RefLib(storage_mode=3, padding_mode=1, default_neighbor_id=-7)
    try:
        ref.add_edges("rnd3", g["src"], g["dst"], g["w"])
        rp, col, eid, ws = ref.export_csr("rnd3", g["rows"], 4096)
        assert np.array_equal(rp, g["row_ptr"]) and np.array_equal(col, g["col"]) and np.array_equal(eid, g["eid"])
        og = graph_of(g)
        for k in (3, 33):
            n, e = ref.sample("rnd3", "TopkSampler", g["query"], k)
            on, oe = orc.sample(og, "TopkSampler", g["query"], k, default_neighbor_id=-7)
            assert np.array_equal(n, on) and np.array_equal(e, oe)
        # random samplers: support-set equality per row (what the reference's tests assert)
        og["alias"] = orc.alias_build(g["row_ptr"], g["w_slot"])
        rows = g["rows"][:20]
        for name in SAMPLERS[:3]:
            n, _ = ref.sample("rnd3", name, np.repeat(rows, 400), 8)
            on, _ = orc.sample(og, name, np.repeat(rows, 400), 8, seed=4, default_neighbor_id=-7)
            for i, r in enumerate(rows):
                a = set(n[i * 400:(i + 1) * 400].reshape(-1).tolist())
                b = set(on[i * 400:(i + 1) * 400].reshape(-1).tolist())
                full = set(g["col"][g["row_ptr"][i]:g["row_ptr"][i + 1]].tolist())
                assert a <= full and b <= full
                if name != "EdgeWeightSampler" and len(full) <= 8:
                    assert a == b == full
    finally:
        ref.close()


def test_partition_stitch_reference_unittest_layout(orc):
    """partition_stitch_unittest.cpp DenseReq_DenseRes: ids {1,2,3,4} on 2 servers ->
    shard 0 serves {2,4} (stickers 1,3), shard 1 serves {1,3} (stickers 0,2); the
    stitched dense [4,6] response has every row back at its request position."""
    ids = np.array([1, 2, 3, 4], np.int64)
    order, counts = orc.partition(ids, 2)
    assert counts.tolist() == [2, 2] and order.tolist() == [1, 3, 0, 2]
    shard_rows = np.stack([ids[order] * 100 + j for j in range(6)], 1)  # what each shard returns
    out = orc.stitch(shard_rows, order)
    assert np.array_equal(out, np.stack([ids * 100 + j for j in range(6)], 1))


def test_python_fixture_random_without_replacement_sets(orc):
    """test_random_worepl_neighbor_sampling.py:31-63: with expand_factor >= degree the
    sampled set of every seed equals its full neighbour set (circular padding), and
    additionally contains the default id (-1) under replicate padding."""
    g = load("pyfixture_topk.npz")
    og = graph_of(g)
    seeds = np.array([102, 107, 108], np.int64)
    for pad in (1, 0):
        nbr, eid = orc.sample(og, "RandomWithoutReplacementSampler", seeds, 4, seed=9, call_counter=1,
                              padding_mode=pad, default_neighbor_id=-1)
        for i, s in enumerate(seeds):
            full = set(int(s) * it % 100 for it in range(1, int(s) % 5 + 1))
            want = full if pad == 1 else full | {-1}
            assert set(nbr[i].tolist()) == want, (pad, s)


def test_alias_method_unittest_membership(orc):
    """alias_method_unittest.cpp:33-60: dist {1.2, .3, .4}, every draw is a valid index."""
    rp = np.array([0, 3], np.int64)
    w = np.array([1.2, 0.3, 0.4], np.float32)
    g = dict(row_ptr=rp, col=np.arange(3, dtype=np.int64), eid=np.arange(3, dtype=np.int64), weight=w,
             alias=orc.alias_build(rp, w))
    nbr, _ = orc.sample(g, "EdgeWeightSampler", np.zeros(1000, np.int64), 10, seed=3)
    assert set(np.unique(nbr).tolist()) <= {0, 1, 2}


# ------------------------------------------------ next rows: Full / InDegree samplers ---
def test_full_sampler_golden(orc):
    """FullSampler (full_sampler.cc:28-97): sparse response, truncated at neighbor_count."""
    g = load("rand_graph.npz")
    og = graph_of(g)
    for lim in (0, 3, 33):
        d, n, e = orc.sample_full(og, g["query"], lim)
        assert np.array_equal(d, g["full_l%d_deg" % lim])
        assert np.array_equal(n, g["full_l%d_nbr" % lim]) and np.array_equal(e, g["full_l%d_eid" % lim])


def test_in_degree_tables_golden(orc):
    """InDegreeSampler's per-row weights (GetInDegree) and alias tables, vs the reference's own."""
    g = load("rand_graph.npz")
    og = graph_of(g)
    (ip, ia), w = orc.in_degree_alias(og)
    assert np.array_equal(w, g["indeg_w"])
    assert beq(ip, g["indeg_alias_prob"]) and np.array_equal(ia, g["indeg_alias_idx"])


def test_in_degree_sampler_distribution(orc):
    g = load("dist_indegree.npz")
    og = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], ids=g["rows"])
    (ip, ia), w = orc.in_degree_alias(og)
    assert np.array_equal(w, g["indeg_w"])
    og["indeg_alias"] = (ip, ia)
    T, degs, k = int(g["T"]), g["degs"], 4
    _, eid = orc.sample(og, "InDegreeSampler", np.tile(g["rows"][:len(degs)], T), k, seed=5, call_counter=8)
    eid = eid.reshape(T, len(degs), k)
    for r, d in enumerate(degs):
        pos = _positions(g, eid[:, r, :], r)
        for j in range(k):
            p = _two_sample_p(np.bincount(pos[:, j], minlength=d), g["hist"][r, j, :d])
            assert p > 1e-4, (r, j, p)


def test_aggregate_stitch_golden(orc):
    """a9: AggregatingResponse::Stitch (aggregating_request.cc:172-213).  The golden file
    holds a 3-server run of the reference: per-server partial responses, the reference's
    stitched response and the single-server answer for the same request."""
    g = load("agg_stitch.npz")
    diverged = set()
    for c in range(int(g["num_cases"])):
        dflt = float(g["c%d_default" % c])
        for name in AGGREGATORS:
            parts, cnts = g["c%d_%s_parts" % (c, name)], g["c%d_%s_cnts" % (c, name)]
            ref_emb, single = g["c%d_%s_stitched" % (c, name)], g["c%d_%s_single" % (c, name)]
            # 1. the restatement with the reference's fold == the reference, bit for bit
            e, n = orc.aggregate_stitch(name, parts, cnts, dflt, reference_fold=True)
            assert beq(e, ref_emb) and np.array_equal(n, g["c%d_%s_cnt" % (c, name)]), (c, name)
            # 2. the contract fold (skip empty partials): bit-identical to the reference
            #    wherever no partial of the segment is empty ...
            e, n = orc.aggregate_stitch(name, parts, cnts, dflt)
            full = (cnts > 0).all(axis=0) | (cnts.sum(axis=0) == 0)
            assert beq(e[full], ref_emb[full]), (c, name)
            # ... and equal to the SINGLE-server answer everywhere (Max/Min exactly; the
            # sums are re-associated across servers)
            if name in ("MaxAggregator", "MinAggregator"):
                assert beq(e, single), (c, name)
            else:
                scale = np.abs(parts).max() * (cnts.sum(axis=0).max() if name != "ProdAggregator" else 1)
                tol = 1e-5 * (np.abs(single) + (scale if name != "ProdAggregator" else 0.0))
                assert (np.abs(e - single) <= tol + 1e-30).all(), (c, name)
            if not beq(ref_emb, e):
                diverged.add((name, dflt != 0.0))
    # the reference's own distributed Max/Min/Prod fold the empty
    # servers' DefaultFloatAttribute rows in.  Sum is only safe while that default is 0
    # (x + 0); Mean always is (an empty partial has weight 0).
    assert ("SumAggregator", False) not in diverged and ("SumAggregator", True) in diverged
    assert not {d for d in diverged if d[0] == "MeanAggregator"}
    assert {("MaxAggregator", False), ("MinAggregator", False), ("ProdAggregator", False)} <= diverged


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_aggregate_stitch_live_reference(orc):
    rng = np.random.default_rng(8)
    ref = RefLib()
    try:
        for P, Sg, D in ((2, 30, 5), (8, 17, 32)):
            parts = (rng.standard_normal((P, Sg, D)) * 5).astype(np.float32)
            parts[0, 0, 0] = -0.0
            cnts = rng.integers(0, 4, (P, Sg)).astype(np.int32)
            parts[cnts == 0] = 0.0
            for name in AGGREGATORS:
                e, n = orc.aggregate_stitch(name, parts, cnts, 0.0, reference_fold=True)
                re_, rn = ref.aggregate_stitch(name, parts, cnts)
                assert beq(e, re_) and np.array_equal(n, rn), (P, name)
    finally:
        ref.close()


def test_negative_candidates_and_tables_golden(orc):
    """Negative samplers' inputs as the reference builds them: destination ids in
    first-appearance order + in-degrees (topo_statics.cc:32-55) and the ONE alias table over
    the whole list (AliasMethodFactory::LookupOrCreate), bit for bit."""
    g = load("negative.npz")
    ids, deg = orc.dst_statics(g["col"], g["eid"])
    assert np.array_equal(ids, g["dst_ids"]) and np.array_equal(deg, g["in_degrees"])
    p, a = orc.alias_build(np.array([0, ids.shape[0]], np.int64), deg.astype(np.float32))
    assert beq(p, g["indeg_prob"]) and np.array_equal(a, g["indeg_alias"])
    p, a = orc.alias_build(np.array([0, g["node_ids"].shape[0]], np.int64), g["node_weights"])
    assert beq(p, g["node_prob"]) and np.array_equal(a, g["node_alias"])


def test_negative_retry_schedule(orc):
    """in_degree_negative_sampler.cc:57-92: blocks of `count` draws, accepted candidates keep
    their order, the exclusion set is dropped from the 4th block on."""
    g = load("negative.npz")
    graph = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
    ids = g["dst_ids"]
    table = (g["indeg_prob"], g["indeg_alias"])
    src = g["rows"][:40].copy()
    k = 7
    strict = orc.negative_sample(ids, table, 1, graph, src, k, seed=5, call_counter=2)
    free = [orc.negative_sample(ids, table, 0, graph, src, k, seed=5, call_counter=2)]
    # the strict result is the unrestricted candidate stream of blocks 0.. with neighbours removed
    pos = {int(v): i for i, v in enumerate(g["rows"])}
    for i, s in enumerate(src):
        nb = set(g["col"][g["row_ptr"][pos[int(s)]]:g["row_ptr"][pos[int(s)] + 1]].tolist())
        stream = free[0][i].tolist()
        want = [c for c in stream if c not in nb]
        assert strict[i].tolist()[:len(want)] == want[:k]
    # all candidates excluded: only the 4th block (no exclusion) can deliver
    few = ids[:3]
    tab = orc.alias_build(np.array([0, 3], np.int64), np.ones(3, np.float32))
    out = orc.negative_sample(few, tab, 2, None, few, 5, seed=1, call_counter=1)
    assert set(out.reshape(-1).tolist()) <= set(few.tolist())
    assert (orc.negative_sample(few[:0], None, 0, None, few, 4, default_neighbor_id=-9) == -9).all()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name,exclude,weighted", [("RandomNegativeSampler", 0, False),
                                                   ("SoftInDegreeNegativeSampler", 0, True),
                                                   ("InDegreeNegativeSampler", 1, True)])
def test_negative_samplers_match_reference_distribution(orc, name, exclude, weighted):
    g = load("negative.npz")
    ref = RefLib()
    try:
        ref.add_edges("neg", g["src"], g["dst"], g["w"])
        graph = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
        ids = g["dst_ids"]
        table = (g["indeg_prob"], g["indeg_alias"]) if weighted else None
        src = np.repeat(g["rows"][:10], 600)
        got_ref = ref.negative_sample("neg", name, src, 6)
        got = orc.negative_sample(ids, table, exclude, graph, src, 6, seed=31, call_counter=4)
        index = {int(v): i for i, v in enumerate(ids)}
        hr = np.bincount([index[int(v)] for v in got_ref.reshape(-1)], minlength=ids.shape[0])
        ho = np.bincount([index[int(v)] for v in got.reshape(-1)], minlength=ids.shape[0])
        keep = (hr + ho) >= 10
        assert stats.chi2_contingency(np.stack([hr[keep], ho[keep]]))[1] > 1e-4, name
        if exclude == 1:  # neighbours slip through equally rarely (only via the 4th block)
            pos = {int(v): i for i, v in enumerate(g["rows"])}

            def leaks(res):
                n = 0
                for s, row in zip(src, res):
                    nb = g["col"][g["row_ptr"][pos[int(s)]]:g["row_ptr"][pos[int(s)] + 1]]
                    n += int(np.isin(row, nb).sum())
                return n
            a, b = leaks(got), leaks(got_ref)
            assert abs(a - b) <= 6 * np.sqrt(a + b + 1), (a, b)  # same leak rate (hub destinations exhaust 3 blocks)
    finally:
        ref.close()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_node_weight_negative_sampler_matches_reference_distribution(orc):
    g = load("negative.npz")
    ref = RefLib()
    try:
        nid, nw = g["node_ids"], g["node_weights"]
        ref.add_weighted_nodes("nw", nid, nw)
        batch = np.tile(nid[np.random.default_rng(1).integers(0, nid.shape[0], 50)], 60)
        got_ref = ref.negative_sample("nw", "NodeWeightNegativeSampler", batch, 5)
        got = orc.negative_sample(nid, (g["node_prob"], g["node_alias"]), 2, None, batch, 5, seed=8, call_counter=3)
        assert not set(got.reshape(-1).tolist()) & set(batch.tolist())
        assert not set(got_ref.reshape(-1).tolist()) & set(batch.tolist())
        index = {int(v): i for i, v in enumerate(nid)}
        hr = np.bincount([index[int(v)] for v in got_ref.reshape(-1)], minlength=nid.shape[0])
        ho = np.bincount([index[int(v)] for v in got.reshape(-1)], minlength=nid.shape[0])
        keep = (hr + ho) >= 10
        assert stats.chi2_contingency(np.stack([hr[keep], ho[keep]]))[1] > 1e-4
    finally:
        ref.close()


def test_timestamped_rows_sort_by_timestamp_golden(orc):
    """Quirk 10: a timestamped edge type is ordered by timestamp ascending at Build(), even when it is
    weighted too (memory_adj_matrix.cc:60-66,129-148).  tests/golden/timestamped.npz = the reference's
    post-Build adjacency and its Topk answer."""
    g = load("timestamped.npz")
    rows = g["rows"]
    row_of = {int(v): i for i, v in enumerate(rows)}
    r = np.array([row_of[int(s)] for s in g["src"]])
    order = np.argsort(r, kind="stable")
    rp = np.zeros(rows.shape[0] + 1, np.int64)
    np.add.at(rp, r + 1, 1)
    rp = np.cumsum(rp)
    assert np.array_equal(rp, g["row_ptr"])
    col, eid, ts, w = orc.sort_rows_by_timestamp(rp, g["dst"][order], order.astype(np.int64), g["ts"][order],
                                                 g["w"][order])
    assert np.array_equal(col, g["col"]) and np.array_equal(eid, g["eid"]) and beq(w, g["w_slot"])
    graph = dict(row_ptr=rp, col=col, eid=eid, weight=w, ids=rows)
    n, e = orc.sample(graph, "TopkSampler", rows, 4)
    assert np.array_equal(n, g["topk_nbr"]) and np.array_equal(e, g["topk_eid"])  # "top" = earliest here


# ------------------------------------------------------------ sampling filters ---
FILTERS = {"id_eq": (1, 1), "id_gt": (2, 1), "ts_eq": (1, 2), "ts_gt": (2, 2)}  # (FilterType, FilterField)


def filtered_graph(g, orc=None):
    d = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"], ts_slot=g["ts_slot"])
    return d


def test_filtered_topk_and_full_match_reference(orc):
    """a6: Filter::ActOn's reserved order (the in-place partition of filter.cc:83-94, the descending
    binary-search prefix of :74-82 that reads values[0] for every row) and the padders behind
    it, for TopkSampler / FullSampler: bit-identical to the reference on all 32 golden cases."""
    g = load("filtered.npz")
    og = filtered_graph(g)
    for name in g["cases"]:
        kind, strategy, k, pad = str(name).rsplit("_", 3)[0], str(name).split("_")[2], int(str(name).split("_k")[1][0]), \
            int(str(name)[-1])
        ft, ff = FILTERS[kind]
        ids, vals = g[name + "_ids"], g[name + "_values"]
        flt = dict(type=ft, field=ff, values=vals)
        if strategy == "FullSampler":
            deg, nbr, eid = orc.sample_full_filtered(og, ids, k, flt, padding_mode=pad, default_neighbor_id=-7)
            assert np.array_equal(deg, g[name + "_deg"]), name
        else:
            nbr, eid = orc.sample_filtered(og, "TopkSampler", ids, k, flt, padding_mode=pad, default_neighbor_id=-7)
        assert np.array_equal(nbr, g[name + "_nbr"]), name
        assert np.array_equal(eid, g[name + "_eid"]), name


def test_filter_values_expand_like_fill_values(orc):
    """Filter::FillValues (filter.cc:53-67): each value covers batch / len(values) request rows."""
    g = load("filtered.npz")
    vals = np.repeat(g["fill_values"], g["fill_ids"].shape[0] // g["fill_values"].shape[0])
    nbr, eid = orc.sample_filtered(filtered_graph(g), "TopkSampler", g["fill_ids"], 5, dict(type=1, field=1, values=vals),
                                   default_neighbor_id=-7)
    assert np.array_equal(nbr, g["fill_nbr"]) and np.array_equal(eid, g["fill_eid"])


def test_filter_quirks_are_kept(orc):
    g = load("filtered.npz")
    nb = np.array([5, 7, 5, 9, 5, 3], np.int64)
    # the partition pulls survivors from the right end into the holes, rightmost first
    assert orc.filter_act_on(dict(type=1, field=1, values=[5]), 0, nb).tolist() == [5, 1, 3]
    assert orc.filter_act_on(dict(type=2, field=1, values=[5]), 0, nb).tolist() == [0, 5, 2, 4]
    ts = np.array([1, 2, 2, 4, 8, 9], np.int64)
    # timestamp > v: a descending prefix; a single-neighbour row always comes back empty (filter.cc:208-210)
    assert orc.filter_act_on(dict(type=2, field=2, values=[3]), 0, nb, ts).tolist() == [2, 1, 0]
    assert orc.filter_act_on(dict(type=2, field=2, values=[100]), 0, nb[:1], ts[:1]).tolist() == []
    # ... and reads values[0] whatever the request row
    assert orc.filter_act_on(dict(type=2, field=2, values=[3, 100]), 1, nb, ts).tolist() == [2, 1, 0]


@pytest.mark.parametrize("name", ["RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler",
                                  "InDegreeSampler"])
def test_filtered_random_samplers_match_reference_distribution(orc, name):
    """ID == value filter, retry budget 1: per (row, slot) histograms vs 20000 reference requests."""
    g = load("filtered.npz")
    og = dict(row_ptr=g["d_row_ptr"], col=g["d_col"], eid=g["d_eid"], weight=g["d_w_slot"], ids=g["d_rows"],
              indeg_weight=g["d_indeg_w"])
    T, degs, k = int(g["d_T"]), g["d_degs"], 4
    flt = dict(type=1, field=1, values=np.tile(g["d_values"], T), retry_times=1)
    _, eid = orc.sample_filtered(og, name, np.tile(g["d_rows"][:len(degs)], T), k, flt, seed=77, call_counter=2)
    eid = eid.reshape(T, len(degs), k)
    ref_hist = g["d_%s_hist" % name]
    rp = g["d_row_ptr"]
    for r, d in enumerate(degs):
        pos_of = {int(x): i for i, x in enumerate(g["d_eid"][rp[r]:rp[r + 1]])}
        pos = np.vectorize(pos_of.get)(eid[:, r, :])
        for j in range(k):
            h = np.bincount(pos[:, j], minlength=d)
            assert _two_sample_p(h, ref_hist[r, j, :d]) > 1e-4, (name, r, j, h, ref_hist[r, j, :d])
        hit = int(np.flatnonzero(g["d_col"][rp[r]:rp[r + 1]] == g["d_values"][r])[0])
        if name != "RandomSampler":  # ActOn removes the neighbour for good; RandomSampler leaks it (1/d)^2 of the time
            assert ref_hist[r, :, hit].sum() == 0 and (pos == hit).sum() == 0


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_filtered_samplers_live_against_reference(orc):
    """Fresh random requests (every filter, both paddings, Topk + Full) against the live reference."""
    g = load("filtered.npz")
    og = filtered_graph(g)
    ref = RefLib()
    try:
        ref.add_edges_timestamped("flt", g["src"], g["dst"], g["ts"], g["w"])
        rng = np.random.default_rng(8)
        rows, rp = g["rows"], g["row_ptr"]
        for trial in range(40):
            ids = rng.choice(rows, 24)
            kind = list(FILTERS)[trial % 4]
            ft, ff = FILTERS[kind]
            if ff == 1:
                vals = rng.integers(898, 913, ids.shape[0])
            else:
                vals = rng.choice(g["ts"], ids.shape[0]) + rng.integers(-1, 2, ids.shape[0])
            pad = (trial // 4) % 2
            ref.set_flags(pad, -3, 0.0)
            flt = dict(type=ft, field=ff, values=vals.astype(np.int64))
            k = int(rng.integers(1, 9))
            a = orc.sample_filtered(og, "TopkSampler", ids, k, flt, padding_mode=pad, default_neighbor_id=-3)
            b = ref.sample_filtered("flt", "TopkSampler", ids, k, flt)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (trial, kind, pad)
            if pad == 0:  # replicate: no FillWith(dim2) on emptied rows
                a = orc.sample_full_filtered(og, ids, k - 1, flt, padding_mode=pad, default_neighbor_id=-3)
                b = ref.sample_filtered("flt", "FullSampler", ids, k - 1, flt)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (trial, kind)
    finally:
        ref.close()


def test_refseq_golden(orc):
    """The oracle's random samplers against the reference DRAW FOR DRAW, from committed vectors (refseq.npz: the
    reference's own samplers with random_device pinned, make_golden.py::gen_refseq): with glxo_set_reference_entropy
    the row code consumes a sequential MT19937 through libstdc++ 11's distributions (restated in glx_oracle.c) and
    must reproduce every neighbour and edge id -- rows of degree 0 .. 257 and one of 70,000 (std::shuffle's
    one-position-per-variate branch), unknown ids, k = 1, 4, 9.  tests/test_oracle_refseq.py widens this live."""
    g = load("refseq.npz")
    og = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
    og["alias"] = orc.alias_build(g["row_ptr"], g["w_slot"])
    og["indeg_alias"] = orc.in_degree_alias(og)[0]
    try:
        for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "InDegreeSampler"):
            for seed in g["seeds"]:
                for k in g["ks"]:
                    orc.set_reference_entropy(True, int(seed))
                    n, e = orc.sample(og, name, g["query"], int(k), padding_mode=1, default_neighbor_id=-3)
                    key = "%s_s%d_k%d" % (name, seed, k)
                    assert np.array_equal(n, g[key + "_nbr"]) and np.array_equal(e, g[key + "_eid"]), key
    finally:
        orc.set_reference_entropy(False)


def _fuzz_cases(n):
    first = int(os.environ.get("GLX_FUZZ_FIRST", "0"))
    return list(range(first, first + int(os.environ.get("GLX_FUZZ_CASES", str(n)))))


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("case", _fuzz_cases(10))
def test_aggregators_live_fuzz_against_reference(orc, case):
    """Random feature tables (values that overflow Prod, -50s below Max's -37 start, NaN-free), ragged non-decreasing
    segment ids with empty and out-of-range tails (the cursor stalls there, aggregating_request.cc:86-105), unknown
    ids, sparse raw ids, random default attributes: the five aggregators of the oracle against the reference's own
    Aggregator::Aggregate, bit for bit."""
    rng = np.random.default_rng(64000 + case)
    V, D = int(rng.integers(1, 60)), int(rng.choice([1, 2, 3, 8, 17]))
    X = (rng.standard_normal((V, D)) * float(rng.choice([1, 30]))).astype(np.float32)
    X[rng.random((V, D)) < 0.05] = -50.0
    raw = (rng.permutation(V * 3)[:V] - V).astype(np.int64)
    dflt = float(rng.choice([0.0, 1.5, -2.0]))
    ref = RefLib(default_float_attr=dflt)
    try:
        ref.add_nodes("agg_fz%d" % case, raw, X)
        Sg = int(rng.integers(1, 30))
        sizes = rng.integers(0, 9, Sg)
        seg = np.repeat(np.arange(Sg, dtype=np.int32), sizes)
        if rng.random() < 0.3 and seg.size:
            seg = np.concatenate([seg, np.full(int(rng.integers(1, 4)), Sg, np.int32)])  # ids past the last segment
        ids = raw[rng.integers(0, V, seg.shape[0])].copy() if seg.size else np.zeros(0, np.int64)
        ids[rng.random(ids.shape[0]) < 0.1] = 10 ** 9  # unknown ids: the default row
        for name in AGGREGATORS:
            want = ref.aggregate("agg_fz%d" % case, name, ids, seg, Sg, D)
            got = orc.aggregate(X, name, ids, seg, Sg, dflt, ids=raw)
            assert np.array_equal(got[1], want[1]), (case, name)
            assert beq(got[0], want[0]), (case, name)
    finally:
        ref.close()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("case", _fuzz_cases(10))
def test_deterministic_samplers_live_fuzz_against_reference(orc, case):
    """Random timestamped + weighted graphs built by the reference itself (rows in its post-Build order), random
    requests with unknown ids: TopkSampler under both padders, FullSampler with and without every filter kind and
    limit -- the oracle against the reference's operators, id for id."""
    rng = np.random.default_rng(73000 + case)
    V = int(rng.integers(2, 50))
    E = int(rng.integers(1, 12 * V))
    src = rng.integers(0, V, E).astype(np.int64) * 3 - 7
    dst = rng.integers(0, V + 5, E).astype(np.int64) * 3 - 7
    w = (rng.random(E) * 0.9 + 0.05 + np.arange(E) * 2.0 ** -21).astype(np.float32)  # tie-free: topk order is defined
    ts = rng.permutation(E).astype(np.int64) + 1000
    ref = RefLib(default_neighbor_id=-3)
    try:
        tag = "det%d" % case
        ref.add_edges_timestamped(tag, src, dst, ts, w)
        rows = np.unique(src)
        rp, col, eid, ws = ref.export_csr(tag, rows, E + 1)
        ts_slot = ts[eid]
        og = dict(row_ptr=rp, col=col, eid=eid, weight=ws, ids=rows, ts_slot=ts_slot)
        q = np.concatenate([rows[rng.integers(0, rows.shape[0], int(rng.integers(1, 40)))], [10 ** 6, -10 ** 6]]).astype(np.int64)
        for pad in (0, 1):
            ref.set_flags(pad, -3, 0.0)
            k = int(rng.integers(1, 12))
            a = orc.sample(og, "TopkSampler", q, k, padding_mode=pad, default_neighbor_id=-3)
            b = ref.sample(tag, "TopkSampler", q, k)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (case, "topk", pad, k)
            limit = int(rng.choice([0, 1, 3, 20]))
            a = orc.sample_full(og, q, limit)
            b = ref.sample_full(tag, q, limit)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (case, "full", limit)
            kind = list(FILTERS)[int(rng.integers(0, 4))]
            ft, ff = FILTERS[kind]
            vals = (q if ff == 1 else rng.choice(ts, q.shape[0]) + rng.integers(-1, 2, q.shape[0])).astype(np.int64)
            if ff == 1:
                vals = col[rng.integers(0, col.shape[0], q.shape[0])]
            flt = dict(type=ft, field=ff, values=vals)
            a = orc.sample_filtered(og, "TopkSampler", q, k, flt, padding_mode=pad, default_neighbor_id=-3)
            b = ref.sample_filtered(tag, "TopkSampler", q, k, flt)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (case, "topk", kind, pad, k)
            if pad == 0:  # replicate: no FillWith(dim2) on emptied rows (quirk 12)
                a = orc.sample_full_filtered(og, q, limit, flt, padding_mode=pad, default_neighbor_id=-3)
                b = ref.sample_filtered(tag, "FullSampler", q, limit, flt)
                assert all(np.array_equal(x, y) for x, y in zip(a, b)), (case, "full", kind, limit)
    finally:
        ref.set_flags(1, 0, 0.0)
        ref.close()


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_alias_build_live_fuzz_degenerate_weights(orc):
    """AliasMethod::Build (alias_method.cc:57-107) on rows the goldens do not hold: all-zero weights (sum 0, NaN
    probabilities), partly zero, all equal, 60 orders of magnitude apart, small integers -- the oracle's tables equal the
    reference's private probs_ / alias_ bit for bit, NaNs included.  GLX_FUZZ_CASES multiplies the 2,000 rows."""
    rng = np.random.default_rng(5 + int(os.environ.get("GLX_FUZZ_FIRST", "0")))
    ref = RefLib()
    try:
        for trial in range(2000 * max(1, int(os.environ.get("GLX_FUZZ_CASES", "1")))):
            n = int(rng.integers(1, 40))
            kind = trial % 6
            if kind == 0:
                w = rng.random(n)
            elif kind == 1:
                w = np.zeros(n)
            elif kind == 2:
                w = rng.random(n) * (rng.random(n) < 0.5)
            elif kind == 3:
                w = np.full(n, rng.random())
            elif kind == 4:
                w = 10.0 ** rng.integers(-30, 30, n)
            else:
                w = rng.integers(1, 4, n)
            w = w.astype(np.float32)
            p, a = ref.alias_build(w)
            op, oa = orc.alias_build(np.array([0, n], np.int64), w)
            assert beq(p, op) and np.array_equal(a, oa), (trial, kind, w)
    finally:
        ref.close()

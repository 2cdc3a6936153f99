"""The oracle's random samplers against the reference's OWN code, draw for draw.

The reference seeds one thread_local std::mt19937 per sampler source file from std::random_device and has no seed API
(random_sampler.cc:46-47, random_without_replacement_sampler.cc:53-54, alias_method.cc:114-115), so the contract the
HIP kernels are bit-exact to (oracle/glx_oracle.c: Philox, one stream per request row) can only agree with it in
distribution -- tests/test_oracle_golden.py checks that with chi-square tests and support sets.  This file closes the
rest of the gap at the oracle: with glxo_set_reference_entropy() the SAME row algorithms take their variates the way
the reference does (sequential MT19937 + libstdc++ 11's uniform_int_distribution / std::shuffle /
uniform_real_distribution, restated in glx_oracle.c), and every output then equals oracle/_ref -- the reference's own
translation units with random_device pinned -- bit for bit: row lookup, default fill, the alias compare on the
float-cast variate, both padders, the order of requests within a thread.  What differs between the reference and the
contract is therefore exactly the entropy source and its two documented mappings (bounded integer, unit real), nothing
in the sampling logic.  libstdc++-version-specific like oracle/_ref itself (gcc 11.4)."""
import numpy as np
import pytest

from oracle_bindings import Oracle, RefLib, have_ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")

RANDOM, RWOR, EDGE_WEIGHT, IN_DEGREE = ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler",
                                        "InDegreeSampler")


def _graph(seed, degrees, tag):
    """Source v has degrees[v] out-edges to random destinations, tie-free weights; + the reference's view of it."""
    rng = np.random.default_rng(seed)
    src = np.repeat(np.arange(len(degrees), dtype=np.int64), degrees)
    dst = rng.integers(0, max(8, len(degrees) // 2), src.shape[0]).astype(np.int64)
    w = (rng.random(src.shape[0]) * 0.9 + 0.05 + np.arange(src.shape[0]) * 2.0 ** -22).astype(np.float32)
    ref = RefLib(storage_mode=2, padding_mode=1, default_neighbor_id=-3)
    ref.add_edges(tag, src, dst, w)
    rows = np.arange(len(degrees), dtype=np.int64)
    rows = rows[np.asarray(degrees) > 0]
    rp, col, eid, ws = ref.export_csr(tag, rows, int(max(degrees)) + 1)
    return ref, dict(row_ptr=rp, col=col, eid=eid, weight=ws, ids=rows)


@pytest.fixture(scope="module")
def orc():
    o = Oracle()
    yield o
    o.set_reference_entropy(False)


@pytest.fixture(scope="module")
def small(orc):
    # degrees 0 (default fill), 1 and 2 (the shuffle's first branches), odd / even, one beyond 256
    degrees = [0, 1, 2, 3, 4, 5, 8, 13, 16, 33, 64, 257, 0, 7, 1, 100]
    ref, g = _graph(11, degrees, "small")
    g["alias"] = orc.alias_build(g["row_ptr"], g["weight"])
    g["indeg_alias"] = orc.in_degree_alias(g)[0]
    yield ref, g, len(degrees)
    ref.close()


@pytest.mark.parametrize("name", [RANDOM, RWOR, EDGE_WEIGHT, IN_DEGREE])
@pytest.mark.parametrize("k", [1, 4, 9])
@pytest.mark.parametrize("padding", [1, 0])  # CIRCULAR, REPLICATE
def test_single_requests_equal_the_reference_draw_for_draw(orc, small, name, k, padding):
    ref, g, V = small
    if padding == 0 and name in (EDGE_WEIGHT, IN_DEGREE):
        pytest.skip("quirk 3: the reference's replicate padder reads neighbours [0, k) of a row shorter than k")
    ref.set_flags(padding_mode=padding, default_neighbor_id=-3)
    rng = np.random.default_rng(k * 7 + padding)
    src = np.concatenate([np.arange(V + 3, dtype=np.int64), rng.integers(0, V, 300).astype(np.int64)])  # + unknown ids
    for seed in (1, 20240923, 0xfffffffe):
        ref.set_seed(seed)
        want_n, want_e = ref.sample("small", name, src, k, fresh_thread=True)
        orc.set_reference_entropy(True, seed)
        got_n, got_e = orc.sample(g, name, src, k, padding_mode=padding, default_neighbor_id=-3)
        assert np.array_equal(got_n, want_n) and np.array_equal(got_e, want_e), (name, k, padding, seed)
    ref.set_flags(padding_mode=1, default_neighbor_id=-3)


def test_engines_carry_over_between_requests_and_are_per_source_file(orc, small):
    """Six requests in one thread: each sampler continues ITS engine; EdgeWeight and InDegree share AliasMethod's."""
    ref, g, V = small
    src = np.random.default_rng(5).integers(0, V, 200).astype(np.int64)
    order = [RANDOM, EDGE_WEIGHT, RWOR, IN_DEGREE, RANDOM, EDGE_WEIGHT]
    ref.set_seed(77)
    want_n, want_e = ref.sample_sequence("small", order, src, 5)
    orc.set_reference_entropy(True, 77)
    for c, name in enumerate(order):
        got_n, got_e = orc.sample(g, name, src, 5, padding_mode=1, default_neighbor_id=-3)
        assert np.array_equal(got_n, want_n[c]) and np.array_equal(got_e, want_e[c]), (c, name)
    # ... and the carried-over state matters: the second RandomSampler request differs from the first
    assert not np.array_equal(want_n[0], want_n[4])


def test_rows_beyond_the_pairwise_shuffle_and_rejected_variates(orc):
    """A 70,000-neighbour row: std::shuffle draws one position per variate there (n * n > 2^32 - 1), and ranges of
    that size make Lemire's rejection step fire; 400 rows of ~2,000 neighbours keep the pairwise branch busy with
    ranges near 2^22 (rejection probability ~1e-3 per draw, thousands of draws)."""
    degrees = [70000] + [1900 + 7 * i for i in range(40)]
    ref, g = _graph(3, degrees, "wide")
    try:
        g["alias"] = orc.alias_build(g["row_ptr"], g["weight"])
        src = np.tile(np.arange(len(degrees), dtype=np.int64), 10)
        for name in (RWOR, RANDOM, EDGE_WEIGHT):
            ref.set_seed(123)
            want_n, want_e = ref.sample("wide", name, src, 6, fresh_thread=True)
            orc.set_reference_entropy(True, 123)
            got_n, got_e = orc.sample(g, name, src, 6, padding_mode=1, default_neighbor_id=-3)
            assert np.array_equal(got_n, want_n) and np.array_equal(got_e, want_e), name
    finally:
        ref.close()


def test_the_contract_path_is_untouched_by_the_switch(orc, small):
    _, g, V = small
    src = np.arange(V, dtype=np.int64)
    orc.set_reference_entropy(False)
    a = orc.sample(g, RANDOM, src, 4, seed=9, call_counter=2)
    orc.set_reference_entropy(True, 5)
    orc.sample(g, RANDOM, src, 4)
    orc.set_reference_entropy(False)
    b = orc.sample(g, RANDOM, src, 4, seed=9, call_counter=2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


# ------------------------------------------------- the other operators that draw: filters, negatives, walks ---
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILTERS = {"id_eq": (1, 1), "id_gt": (2, 1), "ts_eq": (1, 2), "ts_gt": (2, 2)}  # (FilterType, FilterField)


def _load(name):
    return dict(np.load(os.path.join(GOLD, name)))


@pytest.mark.parametrize("name", [RANDOM, RWOR, EDGE_WEIGHT, IN_DEGREE])
def test_filtered_samplers_equal_the_reference_draw_for_draw(orc, name):
    """op::Filter in front of the random samplers (a6): HitAll + the retry budget that is NOT reset between rows
    (random_sampler.cc:50,65-70), ActOn's reserved list under std::shuffle / a per-row alias table -- every filter
    kind, the reference's graph of tests/golden/filtered.npz (timestamped + weighted), 30 fresh requests each."""
    g = _load("filtered.npz")
    og = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"], ts_slot=g["ts_slot"])
    og["indeg_weight"] = orc.in_degree_alias(og)[1]
    ref = RefLib()
    try:
        ref.add_edges_timestamped("flt", g["src"], g["dst"], g["ts"], g["w"])
        rng = np.random.default_rng(21)
        for trial in range(30):
            ids = rng.choice(g["rows"], 40)
            kind = list(FILTERS)[trial % 4]
            ft, ff = FILTERS[kind]
            vals = rng.integers(898, 913, ids.shape[0]) if ff == 1 else \
                rng.choice(g["ts"], ids.shape[0]) + rng.integers(-1, 2, ids.shape[0])
            pad = 1 if name in (EDGE_WEIGHT, IN_DEGREE) else (trial // 4) % 2
            ref.set_flags(pad, -3, 0.0)
            flt = dict(type=ft, field=ff, values=vals.astype(np.int64), retry_times=int(rng.integers(0, 4)))
            k = int(rng.integers(1, 9))
            ref.set_seed(1000 + trial)
            want = ref.sample_filtered("flt", name, ids, k, flt, fresh_thread=True)
            orc.set_reference_entropy(True, 1000 + trial)
            got = orc.sample_filtered(og, name, ids, k, flt, padding_mode=pad, default_neighbor_id=-3)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (name, trial, kind, pad, k)
    finally:
        ref.close()
        orc.set_reference_entropy(False)


@pytest.mark.parametrize("name,exclude,weighted", [("RandomNegativeSampler", 0, False),
                                                   ("SoftInDegreeNegativeSampler", 0, True),
                                                   ("InDegreeNegativeSampler", 1, True)])
def test_negative_samplers_equal_the_reference_draw_for_draw(orc, name, exclude, weighted):
    """One candidate list + one alias table per edge type (negative.npz: the reference's own), blocks of `count`
    draws, accepted candidates in draw order, the exclusion set dropped from the 4th block."""
    g = _load("negative.npz")
    ref = RefLib()
    try:
        ref.add_edges("neg", g["src"], g["dst"], g["w"])
        graph = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
        table = (g["indeg_prob"], g["indeg_alias"]) if weighted else None
        rng = np.random.default_rng(4)
        for trial, count in enumerate((1, 3, 6, 17)):
            src = rng.choice(g["rows"], 150)
            ref.set_seed(50 + trial)
            want = ref.negative_sample("neg", name, src, count, fresh_thread=True)
            orc.set_reference_entropy(True, 50 + trial)
            got = orc.negative_sample(g["dst_ids"], table, exclude, graph, src, count)
            assert np.array_equal(got, want), (name, count)
    finally:
        ref.close()
        orc.set_reference_entropy(False)


def test_node_weight_negative_sampler_equals_the_reference_draw_for_draw(orc):
    g = _load("negative.npz")
    ref = RefLib()
    try:
        nid, nw = g["node_ids"], g["node_weights"]
        ref.add_weighted_nodes("nw", nid, nw)
        batch = nid[np.random.default_rng(1).integers(0, nid.shape[0], 80)]
        for seed, count in ((9, 2), (10, 5), (11, 12)):
            ref.set_seed(seed)
            want = ref.negative_sample("nw", "NodeWeightNegativeSampler", batch, count, fresh_thread=True)
            orc.set_reference_entropy(True, seed)
            got = orc.negative_sample(nid, (g["node_prob"], g["node_alias"]), 2, None, batch, count)
            assert np.array_equal(got, want), count
    finally:
        ref.close()
        orc.set_reference_entropy(False)


@pytest.mark.parametrize("p,q,full_nbr_num", [(1.0, 1.0, 100), (0.5, 2.0, 100), (4.0, 0.25, 3)])
def test_random_walks_equal_the_reference_draw_for_draw(orc, p, q, full_nbr_num):
    """RandomWalk (random_walk.cc): DeepWalk steps draw from the operator's own engine, node2vec steps build a biased
    alias table per walker per step and draw through AliasMethod's.  Every vertex of this graph has out-edges; graphs
    with dead ends -- where the reference's cursor slips, :214-226 -- are test_node2vec_walks_with_dead_ends_equal_the_reference's."""
    rng = np.random.default_rng(12)
    V = 60
    deg = rng.integers(1, 9, V)
    src = np.repeat(np.arange(V, dtype=np.int64), deg)
    dst = rng.integers(0, V, src.shape[0]).astype(np.int64)
    w = (rng.random(src.shape[0]) * 0.9 + 0.05 + np.arange(src.shape[0]) * 2.0 ** -20).astype(np.float32)
    ref = RefLib(default_neighbor_id=-3)
    try:
        ref.add_edges("walk", src, dst, w)
        rows = np.arange(V, dtype=np.int64)
        rp, col, eid, ws = ref.export_csr("walk", rows, 16)
        og = dict(row_ptr=rp, col=col, eid=eid, weight=ws, ids=rows)
        seeds = rng.integers(0, V, 120).astype(np.int64)
        for seed in (3, 4):
            ref.set_seed(seed)
            want = ref.random_walk("walk", seeds, 6, p, q, full_nbr_num=full_nbr_num, fresh_thread=True)
            orc.set_reference_entropy(True, seed)
            got = orc.random_walk(og, seeds, 6, p=p, q=q, full_nbr_num=full_nbr_num, default_neighbor_id=-3)
            assert np.array_equal(got, want), (p, q, seed)
    finally:
        ref.close()
        orc.set_reference_entropy(False)


@pytest.mark.parametrize("name,strategy,share,unique", [
    ("random", "random", False, False), ("random_unique", "random", False, True), ("random_share", "random", True, False),
    ("in_degree", "in_degree", False, False), ("node_weight", "node_weight", False, True)])
def test_conditional_negative_sampler_equals_the_reference_draw_for_draw(orc, name, strategy, share, unique):
    """ConditionalNegativeSampler (conditional_negative_sampler.cc, condition_table.cc, attribute_nodes_map.h): one
    alias table per attribute group + the default table, every draw through AliasMethod::Sample's engine; 60 seeded
    requests of the fixture tests/golden/cond_negative.npz was generated from.  Requests in which a condition column
    comes up short are skipped: the reference's response is misaligned there (its fill loop is dead code, quirk 14)."""
    from test_oracle_cond_negative import GOLD as CG, setup
    cand, w, keys, dk, g, _ = setup(strategy)
    props = np.concatenate([CG["int_props"], CG["float_props"], CG["str_props"]])
    count = int(CG["count"])
    ref = RefLib()
    compared = 0
    try:
        ref.add_attr_nodes("item", CG["items"], weights=CG["item_w"], int_attrs=CG["int_attr"].reshape(-1, 1),
                           float_attrs=CG["float_attr"].reshape(-1, 1), str_attrs=[[bytes(x)] for x in CG["str_attr"]])
        ref.set_flags(1, 0, 0.0)
        etype = "item"
        if strategy != "node_weight":
            etype = "buy_rs_" + name
            ref.add_edges(etype, CG["src"], CG["dst"], None)
        for t in range(60):
            ref.set_seed(7000 + t)
            want = ref.cond_neg_sample(etype, strategy, "item", CG["req_src"], CG["req_dst"], count, int_cols=[0],
                                       int_props=CG["int_props"], float_cols=[0], float_props=CG["float_props"],
                                       str_cols=[0], str_props=CG["str_props"], batch_share=share, unique=unique)
            orc.set_reference_entropy(True, 7000 + t)
            got, filled = orc.cond_negative_sample(cand, w, keys, props, g, CG["req_src"], CG["req_dst"], dk, count,
                                                   batch_share=share, unique=unique, with_filled=True)
            if want.shape[0] != CG["req_src"].shape[0] * count:
                continue
            assert np.all(filled == count)
            assert np.array_equal(got.reshape(-1), want), (name, t)
            compared += 1
        assert compared >= 50
    finally:
        ref.close()
        orc.set_reference_entropy(False)


def _fuzz_cases(n):
    first = int(os.environ.get("GLX_FUZZ_FIRST", "0"))
    return list(range(first, first + int(os.environ.get("GLX_FUZZ_CASES", str(n)))))


@pytest.mark.parametrize("case", _fuzz_cases(12))
def test_reference_entropy_fuzz(orc, case):
    """Random graphs (degree 0 .. 300, a few hubs of thousands), random sequences of 1-5 requests of random samplers, k,
    seeds and padding in ONE reference thread -- the oracle under the reference's entropy must reproduce every id."""
    rng = np.random.default_rng(31000 + case)
    V = int(rng.integers(1, 60))
    degrees = rng.integers(0, int(rng.choice([3, 12, 300])), V)
    if rng.random() < 0.3:
        degrees[int(rng.integers(0, V))] = int(rng.integers(1000, 9000))
    if degrees.sum() == 0:
        degrees[0] = 1
    ref, g = _graph(case, [int(d) for d in degrees], "fz%d" % case)
    try:
        g["alias"] = orc.alias_build(g["row_ptr"], g["weight"])
        g["indeg_alias"] = orc.in_degree_alias(g)[0]
        pad = int(rng.integers(0, 2))
        ref.set_flags(padding_mode=pad, default_neighbor_id=-3)
        pool = [RANDOM, RWOR] if pad == 0 else [RANDOM, RWOR, EDGE_WEIGHT, IN_DEGREE]  # quirk 3: see above
        order = [str(x) for x in rng.choice(pool, int(rng.integers(1, 6)))]
        k = int(rng.integers(1, 40))
        src = rng.integers(-1, V + 1, int(rng.integers(1, 80))).astype(np.int64)
        seed = int(rng.integers(0, 2 ** 32))
        ref.set_seed(seed)
        want_n, want_e = ref.sample_sequence("fz%d" % case, order, src, k)
        orc.set_reference_entropy(True, seed)
        for c, name in enumerate(order):
            got_n, got_e = orc.sample(g, name, src, k, padding_mode=pad, default_neighbor_id=-3)
            assert np.array_equal(got_n, want_n[c]) and np.array_equal(got_e, want_e[c]), (case, c, name, k, pad)
    finally:
        ref.set_flags(padding_mode=1, default_neighbor_id=-3)
        ref.close()
        orc.set_reference_entropy(False)


@pytest.fixture(scope="module")
def families(orc):
    """The reference stores the wider sweeps below draw from, built once: the timestamped graph of filtered.npz, the
    candidate graph of negative.npz (+ its weighted node type), a walk graph where every vertex has out-edges."""
    f = _load("filtered.npz")
    n = _load("negative.npz")
    ref = RefLib(default_neighbor_id=-3)
    ref.add_edges_timestamped("fam_flt", f["src"], f["dst"], f["ts"], f["w"])
    ref.add_edges("fam_neg", n["src"], n["dst"], n["w"])
    ref.add_weighted_nodes("fam_nw", n["node_ids"], n["node_weights"])
    rng = np.random.default_rng(77)
    V = 80
    deg = rng.integers(1, 12, V)
    src = np.repeat(np.arange(V, dtype=np.int64), deg)
    dst = rng.integers(0, V, src.shape[0]).astype(np.int64)
    w = (rng.random(src.shape[0]) * 0.9 + 0.05 + np.arange(src.shape[0]) * 2.0 ** -20).astype(np.float32)
    ref.add_edges("fam_walk", src, dst, w)
    rows = np.arange(V, dtype=np.int64)
    rp, col, eid, ws = ref.export_csr("fam_walk", rows, 16)
    og_f = dict(row_ptr=f["row_ptr"], col=f["col"], eid=f["eid"], weight=f["w_slot"], ids=f["rows"], ts_slot=f["ts_slot"])
    og_f["indeg_weight"] = orc.in_degree_alias(og_f)[1]
    og_n = dict(row_ptr=n["row_ptr"], col=n["col"], eid=n["eid"], weight=n["w_slot"], ids=n["rows"])
    yield dict(ref=ref, f=f, n=n, og_f=og_f, og_n=og_n, og_w=dict(row_ptr=rp, col=col, eid=eid, weight=ws, ids=rows), V=V)
    ref.close()


@pytest.mark.parametrize("case", _fuzz_cases(12))
def test_reference_entropy_fuzz_filters_negatives_walks(orc, families, case):
    """One random request per family and case -- a filtered sampler (any filter kind, retry budget 0-5), a negative
    sampler (any of the four, counts 1-24), a walk (DeepWalk or node2vec, lengths 1-7, caps 1-100) -- each in a fresh
    reference thread under a random seed: the oracle under the reference's entropy returns the same ids."""
    rng = np.random.default_rng(52000 + case)
    ref, f, n = families["ref"], families["f"], families["n"]
    try:
        # filtered sampler
        name = str(rng.choice([RANDOM, RWOR, EDGE_WEIGHT, IN_DEGREE]))
        kind = list(FILTERS)[int(rng.integers(0, 4))]
        ft, ff = FILTERS[kind]
        ids = rng.choice(f["rows"], int(rng.integers(1, 60)))
        vals = rng.integers(898, 913, ids.shape[0]) if ff == 1 else rng.choice(f["ts"], ids.shape[0]) + rng.integers(-1, 2, ids.shape[0])
        pad = 1 if name in (EDGE_WEIGHT, IN_DEGREE) else int(rng.integers(0, 2))
        ref.set_flags(pad, -3, 0.0)
        flt = dict(type=ft, field=ff, values=vals.astype(np.int64), retry_times=int(rng.integers(0, 6)))
        k, seed = int(rng.integers(1, 14)), int(rng.integers(0, 2 ** 32))
        ref.set_seed(seed)
        want = ref.sample_filtered("fam_flt", name, ids, k, flt, fresh_thread=True)
        orc.set_reference_entropy(True, seed)
        got = orc.sample_filtered(families["og_f"], name, ids, k, flt, padding_mode=pad, default_neighbor_id=-3)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), (case, name, kind, pad, k)
        ref.set_flags(1, -3, 0.0)
        # negative sampler
        neg = [("RandomNegativeSampler", 0, None), ("SoftInDegreeNegativeSampler", 0, "indeg"),
               ("InDegreeNegativeSampler", 1, "indeg"), ("NodeWeightNegativeSampler", 2, "node")][int(rng.integers(0, 4))]
        count, seed = int(rng.integers(1, 25)), int(rng.integers(0, 2 ** 32))
        ref.set_seed(seed)
        orc.set_reference_entropy(True, seed)
        if neg[2] == "node":
            batch = n["node_ids"][rng.integers(0, n["node_ids"].shape[0], int(rng.integers(1, 90)))]
            want = ref.negative_sample("fam_nw", neg[0], batch, count, fresh_thread=True)
            got = orc.negative_sample(n["node_ids"], (n["node_prob"], n["node_alias"]), 2, None, batch, count)
        else:
            batch = rng.choice(n["rows"], int(rng.integers(1, 120)))
            table = (n["indeg_prob"], n["indeg_alias"]) if neg[2] else None
            want = ref.negative_sample("fam_neg", neg[0], batch, count, fresh_thread=True)
            got = orc.negative_sample(n["dst_ids"], table, neg[1], families["og_n"], batch, count)
        assert np.array_equal(got, want), (case, neg[0], count)
        # walk
        p, q = [(1.0, 1.0), (0.5, 2.0), (4.0, 0.25), (1.0, 3.0)][int(rng.integers(0, 4))]
        F, L, seed = int(rng.choice([1, 2, 5, 100])), int(rng.integers(1, 8)), int(rng.integers(0, 2 ** 32))
        seeds = rng.integers(0, families["V"], int(rng.integers(1, 100))).astype(np.int64)
        ref.set_seed(seed)
        want = ref.random_walk("fam_walk", seeds, L, p, q, full_nbr_num=F, fresh_thread=True)
        orc.set_reference_entropy(True, seed)
        got = orc.random_walk(families["og_w"], seeds, L, p=p, q=q, full_nbr_num=F, default_neighbor_id=-3)
        assert np.array_equal(got, want), (case, p, q, F, L)
    finally:
        ref.set_flags(1, -3, 0.0)
        orc.set_reference_entropy(False)


@pytest.fixture(scope="module")
def cond_world():
    from test_oracle_cond_negative import GOLD as CG
    ref = RefLib()
    ref.add_attr_nodes("item", CG["items"], weights=CG["item_w"], int_attrs=CG["int_attr"].reshape(-1, 1),
                       float_attrs=CG["float_attr"].reshape(-1, 1), str_attrs=[[bytes(x)] for x in CG["str_attr"]])
    ref.set_flags(1, 0, 0.0)
    for strategy in ("random", "in_degree"):
        ref.add_edges("buy_fz_" + strategy, CG["src"], CG["dst"], None)
    yield ref, CG
    ref.close()


@pytest.mark.parametrize("case", _fuzz_cases(12))
def test_reference_entropy_fuzz_conditional_negative(orc, cond_world, case):
    """Random ConditionalNegativeSampler requests -- strategy, batch_share, unique, count, 1-12 random (src, dst) rows --
    against the reference draw for draw (the columns and proportions stay those of the fixture: the reference caches
    its condition table per type).  Requests the reference answers short (quirk 13c) are skipped."""
    from test_oracle_cond_negative import float_key, setup
    ref, CG = cond_world
    rng = np.random.default_rng(88000 + case)
    strategy = str(rng.choice(["random", "in_degree", "node_weight"]))
    share, unique = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    count = int(rng.choice([4, 8, 12, 16]))
    cand, w, keys, _, g, attr = setup(strategy)
    n = int(rng.integers(1, 13))
    req_src = rng.integers(0, 80, n).astype(np.int64)
    req_dst = rng.choice(CG["items"], n).astype(np.int64)
    sdict = {b"A": 0, b"B": 1}
    dk = np.stack([np.array([attr[int(d)][0] for d in req_dst], np.int64), float_key([attr[int(d)][1] for d in req_dst]),
                   np.array([sdict[attr[int(d)][2]] for d in req_dst], np.int64)], axis=1)
    props = np.concatenate([CG["int_props"], CG["float_props"], CG["str_props"]])
    etype = "item" if strategy == "node_weight" else "buy_fz_" + strategy
    seed = int(rng.integers(0, 2 ** 32))
    try:
        ref.set_seed(seed)
        want = ref.cond_neg_sample(etype, strategy, "item", req_src, req_dst, count, int_cols=[0], int_props=CG["int_props"],
                                   float_cols=[0], float_props=CG["float_props"], str_cols=[0], str_props=CG["str_props"],
                                   batch_share=share, unique=unique)
        orc.set_reference_entropy(True, seed)
        got, filled = orc.cond_negative_sample(cand, w, keys, props, g, req_src, req_dst, dk, count, batch_share=share,
                                               unique=unique, with_filled=True)
        if want.shape[0] == n * count:
            assert np.all(filled == count)
            assert np.array_equal(got.reshape(-1), want), (case, strategy, share, unique, count)
    finally:
        orc.set_reference_entropy(False)


@pytest.mark.parametrize("case", _fuzz_cases(16))
def test_node2vec_walks_with_dead_ends_equal_the_reference(orc, case):
    """node2vec on graphs WITH vertices that have no out-edges: the reference walks the concatenated parent lists with
    a cursor it only advances for walkers that can move (random_walk.cc:214-226), so every walker behind a stuck one
    compares its neighbours with a window that starts too early -- reproduced since round 4 (oracle, kernels), and the
    walks must match vertex for vertex.  The default neighbour id is a vertex that has out-edges, so stuck walkers keep
    walking (and keep shifting the windows) on later steps."""
    rng = np.random.default_rng(91000 + case)
    V = int(rng.integers(6, 50))
    deg = rng.integers(0, 9, V)
    deg[rng.random(V) < 0.3] = 0  # dead ends
    deg[0] = max(int(deg[0]), 2)
    src = np.repeat(np.arange(V, dtype=np.int64), deg)
    dst = rng.integers(0, V, src.shape[0]).astype(np.int64)
    w = (rng.random(src.shape[0]) * 0.9 + 0.05 + np.arange(src.shape[0]) * 2.0 ** -20).astype(np.float32)
    ref = RefLib(default_neighbor_id=0)
    try:
        tag = "dead%d" % case
        ref.add_edges(tag, src, dst, w)
        rows = np.flatnonzero(deg > 0).astype(np.int64)
        rp, col, eid, ws = ref.export_csr(tag, rows, 16)
        og = dict(row_ptr=rp, col=col, eid=eid, weight=ws, ids=rows)
        seeds = rng.integers(0, V, int(rng.integers(2, 60))).astype(np.int64)
        p, q = [(0.5, 2.0), (4.0, 0.25), (1.0, 3.0), (0.1, 0.1)][int(rng.integers(0, 4))]
        F, L, seed = int(rng.choice([1, 2, 4, 100])), int(rng.integers(2, 8)), int(rng.integers(0, 2 ** 32))
        ref.set_seed(seed)
        want = ref.random_walk(tag, seeds, L, p, q, full_nbr_num=F, fresh_thread=True)
        orc.set_reference_entropy(True, seed)
        got = orc.random_walk(og, seeds, L, p=p, q=q, full_nbr_num=F, default_neighbor_id=0)
        assert np.array_equal(got, want), (case, p, q, F, L)
    finally:
        ref.close()
        orc.set_reference_entropy(False)

"""CPU tests of the oracle's RandomWalk restatement (core/operator/random_walk/random_walk.cc).

The reference's operator cannot be built here without its RPC client and runner, so the walk is pinned through
its parts: the biased weights against the formula of WeightedRandomWalkKernel (:228-272) on hand-made cases, the
alias tables through the (reference-pinned) AliasMethod restatement, and the walk's transition frequencies
against the exact distribution those tables imply."""
import numpy as np
from scipy import stats

from oracle_bindings import Oracle

orc = Oracle()


def csr(edges, weights=None):
    """edges: list of (src, dst) in row order -> oracle graph dict with dense ids."""
    V = max(max(s, d) for s, d in edges) + 1
    rows = [[] for _ in range(V)]
    for n, (s, d) in enumerate(edges):
        rows[s].append((d, n))
    rp = np.zeros(V + 1, np.int64)
    col, eid = [], []
    for v in range(V):
        rp[v + 1] = rp[v] + len(rows[v])
        col += [d for d, _ in rows[v]]
        eid += [n for _, n in rows[v]]
    eid = np.array(eid, np.int64)
    g = dict(row_ptr=rp, col=np.array(col, np.int64), eid=eid)
    if weights is not None:
        g["weight"] = np.asarray(weights, np.float32)[eid]
    return g


def pick_distribution(orc, w):
    """Exact distribution of AliasMethod::Sample(1) over weights w (alias_method.cc:109-124):
    start slot uniform in [0, n-1), keep it with probability probs[slot], else its alias."""
    n = w.shape[0]
    if n == 1:
        return np.ones(1)
    prob, alias = orc.alias_build(np.array([0, n], np.int64), w)
    out = np.zeros(n)
    for ix in range(n - 1):
        keep = min(max(float(prob[ix]), 0.0), 1.0)
        out[ix] += keep / (n - 1)
        out[alias[ix]] += (1.0 - keep) / (n - 1)
    return out


def test_biased_weights_follow_the_reference_formula():
    # 0 -> {1, 2, 3}; 1 -> {0, 2, 4}: from 1 with parent 0: back to 0 -> w/p, 2 is shared with 0 -> w, 4 -> w/q
    g = csr([(0, 1), (0, 2), (0, 3), (1, 0), (1, 2), (1, 4), (2, 0), (3, 0), (4, 1)],
            [1.0, 2.0, 3.0, 0.5, 0.25, 4.0, 1.0, 1.0, 1.0])
    p, q = np.float32(0.25), np.float32(4.0)
    w = orc.node2vec_weights(g, 1, 0, True, p, q)
    want = [np.float32(0.5 * 1.0 / (float(p) + 1e-6)), np.float32(0.25), np.float32(4.0 * 1.0 / (float(q) + 1e-6))]
    assert w.tolist() == [float(x) for x in want]
    # first step: the parent is the vertex itself and has no neighbour list -> everything is w/q
    w0 = orc.node2vec_weights(g, 0, 0, False, p, q)
    assert w0.tolist() == [float(np.float32(x * 1.0 / (float(q) + 1e-6))) for x in (1.0, 2.0, 3.0)]
    # only the first DefaultFullNbrNum neighbours take part, on both sides
    assert orc.node2vec_weights(g, 1, 0, True, p, q, full_nbr_num=2).shape[0] == 2
    w2 = orc.node2vec_weights(g, 1, 0, True, p, q, full_nbr_num=1)  # parent list = {1}: nothing shared
    assert w2.tolist() == [float(want[0])]
    # an unweighted type weighs with DefaultWeight
    gu = dict(g)
    del gu["weight"]
    assert orc.node2vec_weights(gu, 1, 0, True, np.float32(1.0), np.float32(2.0), default_weight=0.5)[1] == 0.5


def test_walk_transitions_match_the_alias_distribution():
    rng = np.random.default_rng(4)
    V = 12
    edges = [(s, int(d)) for s in range(V) for d in rng.choice(V, int(rng.integers(2, 7)), replace=False)]
    g = csr(edges, rng.random(len(edges)) + 0.05)
    p, q = np.float32(0.5), np.float32(2.0)
    T = 60000
    rp, col = g["row_ptr"], g["col"]
    start = int(np.argmax(np.diff(rp)))  # the vertex with the most neighbours
    seeds = np.full(T, start, np.int64)
    walks = orc.random_walk(g, seeds, 2, p, q, seed=9, call_counter=1)
    first = col[rp[start]:rp[start + 1]]
    want1 = pick_distribution(orc, orc.node2vec_weights(g, start, start, False, p, q))
    got1 = np.array([(walks[:, 0] == v).sum() for v in first])
    assert (want1 > 0).sum() > 2
    assert stats.chisquare(got1[want1 > 0], want1[want1 > 0] * T)[1] > 1e-4 and got1[want1 == 0].sum() == 0
    for v1 in first:
        sel = walks[walks[:, 0] == v1, 1]
        if sel.shape[0] == 0:
            continue
        nb = col[rp[v1]:rp[v1 + 1]]
        want2 = pick_distribution(orc, orc.node2vec_weights(g, int(v1), start, True, p, q))
        got2 = np.array([(sel == v).sum() for v in nb])
        assert got2.sum() == sel.shape[0]
        keep = want2 > 0
        if keep.sum() > 1:
            assert stats.chisquare(got2[keep], want2[keep] * sel.shape[0])[1] > 1e-4, (v1, got2, want2)
        assert got2[~keep].sum() == 0


def test_deepwalk_and_dead_ends():
    g = csr([(0, 1), (0, 2), (1, 2), (2, 5)])  # 5 has no out-edges; id 3, 4 unknown rows without edges
    walks = orc.random_walk(g, np.array([0, 5, 7], np.int64), 4, default_neighbor_id=0, seed=2)
    assert walks.shape == (3, 4)
    # a stuck walker yields the default id and walks on from it (vertex 0 here), like the reference
    assert walks[1, 0] == 0 and walks[1, 1] in (1, 2) and walks[2, 0] == 0
    # p = q = 1 is the uniform walk over whole rows: the RandomSampler draw with neighbor_count 1
    n, _ = orc.sample(dict(g), "RandomSampler", np.array([0, 5, 7], np.int64), 1, seed=2, call_counter=0)
    assert np.array_equal(walks[:, 0], n[:, 0])
    # node2vec from a stuck vertex: default id as well
    w2 = orc.random_walk(g, np.array([5], np.int64), 2, np.float32(2.0), np.float32(0.5), default_neighbor_id=-1, seed=2)
    assert w2.tolist() == [[-1, -1]]


# ------------------------------------------------- against the reference's own operator ---
import os  # noqa: E402

import pytest  # noqa: E402

from oracle_bindings import RefLib, have_ref  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "walk.npz")


def _two_sample_p(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    keep = (a + b) > 0
    return 1.0 if keep.sum() < 2 else stats.chi2_contingency(np.stack([a[keep], b[keep]]))[1]


def _walk_stats(walks, V):
    h1 = np.bincount(walks[:, 0], minlength=V)
    h12 = np.zeros((V, V), np.int64)
    h23 = np.zeros((V, V), np.int64)
    np.add.at(h12, (walks[:, 0], walks[:, 1]), 1)
    np.add.at(h23, (walks[:, 1], walks[:, 2]), 1)
    return h1, h12, h23


def test_walks_match_the_reference_operator_distribution():
    """tests/golden/walk.npz holds 40000 three-step walks per (p, q, DefaultFullNbrNum) of the reference's OWN
    RandomWalk operator (oracle/_ref: random_walk.cc + random_walk_request.cc compiled where they lie).  The
    restatement must produce the same first-step, (step 1, step 2) and (step 2, step 3) frequencies: that pins the
    1/p, 1, 1/q weighting, the self-parent first step, the neighbour cap and the single alias draw end to end."""
    g = dict(np.load(GOLD))
    og = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
    T, V = int(g["T"]), 14
    for name in g["cases"]:
        name = str(name)
        p, q, F = float(name.split("_")[0][1:]), float(name.split("_")[1][1:]), int(name.split("_F")[1])
        walks = orc.random_walk(og, np.full(T, 5, np.int64), 3, np.float32(p), np.float32(q), full_nbr_num=F, seed=31,
                                call_counter=6)
        for got, want, what in zip(_walk_stats(walks, V), (g[name + "_h1"], g[name + "_h12"], g[name + "_h23"]),
                                   ("step 1", "steps 1-2", "steps 2-3")):
            assert ((got > 0) == (want > 0)).all() or _two_sample_p(got, want) > 1e-4, (name, what)  # same support ...
            assert _two_sample_p(got, want) > 1e-4, (name, what, got, want)                          # ... same law


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_walks_live_against_the_reference_operator():
    g = dict(np.load(GOLD))
    og = dict(row_ptr=g["row_ptr"], col=g["col"], eid=g["eid"], weight=g["w_slot"], ids=g["rows"])
    ref = RefLib()
    try:
        ref.add_edges("walk", g["src"], g["dst"], g["w"])
        ref.set_seed(99)
        T = 30000
        for start, p, q, F in ((2, 0.25, 1.0, 100), (9, 1.0, 3.0, 4)):
            seeds = np.full(T, start, np.int64)
            a = _walk_stats(ref.random_walk("walk", seeds, 3, p, q, F), 14)
            b = _walk_stats(orc.random_walk(og, seeds, 3, np.float32(p), np.float32(q), full_nbr_num=F, seed=1), 14)
            for x, y in zip(a, b):
                assert _two_sample_p(x, y) > 1e-4, (start, p, q, F)
    finally:
        ref.close()

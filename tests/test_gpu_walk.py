"""GPU parity of the RandomWalk operator (core/operator/random_walk/random_walk.cc): glx_random_walk through the
C-ABI, bit for bit against the oracle (tests/test_oracle_walk.py pins the oracle)."""
import os

import numpy as np
import pytest

import glx
from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    return Oracle()


def graph(rng, weighted=True, V=300, E=5000):
    src = (rng.zipf(1.5, E) % V).astype(np.int64) * 3 - 50
    dst = src[rng.integers(0, E, E)].copy()  # destinations are sources too, so walks keep going
    dst[:200] = rng.integers(10 ** 6, 10 ** 6 + 50, 200)  # ... except into these dead ends
    w = (rng.random(E) + 0.02).astype(np.float32) if weighted else None
    dev = glx.Graph.from_edges(src, dst, w)
    rows = np.unique(src)
    deg, col, eid = dev.sample_full(rows, 0)
    rp = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
    og = dict(row_ptr=rp, col=col, eid=eid, ids=rows)
    if weighted:
        og["weight"] = w[eid]
    return dev, og, rows


@pytest.mark.parametrize("weighted", [True, False])
def test_random_walk_bit_exact_with_oracle(orc, weighted):
    import torch
    rng = np.random.default_rng(21 + weighted)
    dev, og, rows = graph(rng, weighted)
    seeds = np.concatenate([rng.choice(rows, 900), [10 ** 6 + 3, -77]]).astype(np.int64)
    for p, q, F, L in [(1.0, 1.0, 100, 5), (0.5, 2.0, 100, 4), (4.0, 0.25, 7, 6), (2.0, 2.0, 1, 3), (0.3, 1.0, 300, 1)]:
        kw = dict(p=np.float32(p), q=np.float32(q), full_nbr_num=F, default_weight=0.5, default_neighbor_id=int(rows[0]),
                  seed=11, call_counter=40)
        want = orc.random_walk(og, seeds, L, **kw)
        got = dev.random_walk(seeds, L, **kw)
        assert np.array_equal(got, want), (p, q, F, L)
        got_dev = dev.random_walk(torch.from_numpy(seeds).cuda(), L, **kw)
        assert np.array_equal(got_dev.cpu().numpy(), want), (p, q, F, L)
    # the DeepWalk case is the RandomSampler draw with neighbor_count 1, step by step
    walks = dev.random_walk(seeds, 3, seed=5, call_counter=9, default_neighbor_id=-1)
    cur = seeds
    for t in range(3):
        nbr, _ = dev.sample("RandomSampler", cur, 1, seed=5, call_counter=9 + t, default_neighbor_id=-1)
        assert np.array_equal(walks[:, t], nbr[:, 0])
        cur = np.ascontiguousarray(nbr[:, 0])
    dev.close()


def test_node2vec_prefers_what_p_and_q_say(orc):
    """Small p: walks keep returning to where they came from; small q: they move outward."""
    rng = np.random.default_rng(5)
    dev, og, rows = graph(rng)
    seeds = rng.choice(rows, 20000).astype(np.int64)

    def back_rate(p, q):
        w = dev.random_walk(seeds, 3, p=p, q=q, seed=3, default_neighbor_id=-1)
        return float((w[:, 1] == seeds).mean())  # the second step returns to the seed
    assert back_rate(0.05, 1.0) > 2 * back_rate(1.0, 1.0) > 4 * back_rate(20.0, 1.0)
    dev.close()


def _fuzz_cases(n):
    first = int(os.environ.get("GLX_FUZZ_FIRST", "0"))
    return list(range(first, first + int(os.environ.get("GLX_FUZZ_CASES", str(n)))))


@pytest.mark.parametrize("case", _fuzz_cases(16))
def test_random_walk_fuzz(orc, case):
    """Random graphs (tiny to 400 vertices, weighted or not, dead ends, hubs beyond DefaultFullNbrNum), random p / q
    (incl. the DeepWalk pair), neighbour caps 1 .. 2048, walk lengths 1 .. 9, unknown seeds: device == oracle."""
    rng = np.random.default_rng(5100 + case)
    V = int(rng.choice([2, 9, 60, 400]))
    E = int(rng.integers(1, 25 * V))
    weighted = bool(rng.integers(0, 2))
    src = rng.integers(0, V, E).astype(np.int64) * 2 - V // 2
    dst = rng.integers(0, V + V // 4 + 1, E).astype(np.int64) * 2 - V // 2  # some destinations have no out-edges
    if V >= 60:
        src[: E // 3] = src[0]  # a hub with far more neighbours than the cap
    w = (rng.random(E) + 0.02 + np.arange(E) * 1e-7).astype(np.float32) if weighted else None
    dev = glx.Graph.from_edges(src, dst, w)
    rows = np.unique(src)
    deg, col, eid = dev.sample_full(rows, 0)
    og = dict(row_ptr=np.concatenate([[0], np.cumsum(deg)]).astype(np.int64), col=col, eid=eid, ids=rows)
    if weighted:
        og["weight"] = w[eid]
    seeds = np.concatenate([rng.choice(rows, int(rng.integers(1, 300))), [10 ** 7, -10 ** 7]]).astype(np.int64)
    p, q = [(1.0, 1.0), (0.25, 4.0), (3.0, 0.5), (1.0, 2.0)][int(rng.integers(0, 4))]
    kw = dict(p=np.float32(p), q=np.float32(q), full_nbr_num=int(rng.choice([1, 2, 5, 100, 2048])),
              default_weight=float(rng.choice([0.0, 0.5])), default_neighbor_id=int(rng.choice([-1, int(rows[0])])),
              seed=int(rng.integers(0, 1 << 40)), call_counter=int(rng.integers(0, 1 << 30)))
    L = int(rng.integers(1, 10))
    if not weighted and kw["default_weight"] == 0.0 and (p, q) != (1.0, 1.0):
        kw["default_weight"] = 0.5  # all-zero weights: the reference's alias build divides by zero there
    want = orc.random_walk(og, seeds, L, **kw)
    got = dev.random_walk(seeds, L, **kw)
    assert np.array_equal(got, want), (case, p, q, kw["full_nbr_num"], L)
    dev.close()

"""CPU tests: the oracle's ConditionalNegativeSampler restatement (oracle/glx_oracle.c glxo_cond_negative_sample) against
the reference's own operator (conditional_negative_sampler.cc, condition_table.cc, attribute_nodes_map.h).

The reference is unseeded, so -- as for the other random samplers -- parity is pinned distributionally: golden
counts[row, slot, candidate] of the reference over 3,000 seeded requests (tests/golden/cond_negative.npz) against the
oracle's counts over as many requests, per (row, condition column), by a chi-square homogeneity test; plus the
deterministic structure every response must have (slot groups carry the dst's attribute, nothing from the exclusion set,
no repeats when unique).  Requests whose condition columns came up short are left out on both sides: there the
reference's response is misaligned (its fill loop is dead code) and glx fills the row."""
import os

import numpy as np
import pytest
from scipy import stats

from oracle_bindings import Oracle

GOLD = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cond_negative.npz")))
CASES = (("random", "random", False, False), ("random_unique", "random", False, True), ("random_share", "random", True, False),
         ("in_degree", "in_degree", False, False), ("node_weight", "node_weight", False, True))
NO_KEY = np.iinfo(np.int64).min


def float_key(x):
    x = np.asarray(x, np.float32) + np.float32(0.0)  # -0.0 -> +0.0, as unordered_map<float> compares
    return x.view(np.int32).astype(np.int64)


def setup(strategy):
    """-> candidates, weights, keys [3, U], graph (or None) exactly as the reference derives them."""
    items, src, dst = GOLD["items"], GOLD["src"], GOLD["dst"]
    attr = {int(i): (int(a), float(f), bytes(s)) for i, a, f, s in zip(items, GOLD["int_attr"], GOLD["float_attr"], GOLD["str_attr"])}
    if strategy == "node_weight":  # candidates = the node type's ids, weights = node weights; no neighbour exclusion
        cand, w, g = items, GOLD["item_w"], None
    else:  # candidates = GetAllDstIds(): distinct destinations in first-appearance order
        _, first = np.unique(dst, return_index=True)
        cand = dst[np.sort(first)]
        w = None
        if strategy == "in_degree":
            w = np.array([np.sum(dst == c) for c in cand], np.float32)
        order = np.argsort(src, kind="stable")
        rp = np.zeros(int(src.max()) + 2, np.int64)
        np.add.at(rp, src + 1, 1)
        g = dict(row_ptr=np.cumsum(rp), col=np.ascontiguousarray(dst[order]), eid=np.ascontiguousarray(order.astype(np.int64)))
    sdict = {b"A": 0, b"B": 1}
    keys = np.stack([np.array([attr[int(c)][0] for c in cand], np.int64), float_key([attr[int(c)][1] for c in cand]),
                     np.array([sdict[attr[int(c)][2]] for c in cand], np.int64)])
    dk = np.stack([np.array([attr[int(d)][0] for d in GOLD["req_dst"]], np.int64), float_key([attr[int(d)][1] for d in GOLD["req_dst"]]),
                   np.array([sdict[attr[int(d)][2]] for d in GOLD["req_dst"]], np.int64)], axis=1)
    return cand, w, keys, dk, g, attr


@pytest.mark.parametrize("name,strategy,share,unique", CASES)
def test_distribution_and_structure_equal_the_reference(name, strategy, share, unique):
    orc = Oracle()
    cand, w, keys, dk, g, attr = setup(strategy)
    props = np.concatenate([GOLD["int_props"], GOLD["float_props"], GOLD["str_props"]])
    count, T = int(GOLD["count"]), int(GOLD["T"])
    req_src, req_dst = GOLD["req_src"], GOLD["req_dst"]
    nums = [int(np.float32(count) * p) for p in props]
    counts = np.zeros((req_src.shape[0], count, GOLD["items"].shape[0]), np.int64)
    used = 0
    for t in range(T):
        out, filled = orc.cond_negative_sample(cand, w, keys, props, g, req_src, req_dst, dk, count, batch_share=share,
                                               unique=unique, seed=77, call_counter=t, with_filled=True)
        if not np.all(filled == count):
            continue
        used += 1
        # structure (every complete response): slot groups carry the dst's attribute; exclusion set; uniqueness
        S = set(int(x) for x in req_dst) if share else set()
        for r in range(req_src.shape[0]):
            if not share:
                if g is not None:
                    S |= set(int(x) for x in g["col"][g["row_ptr"][req_src[r]]:g["row_ptr"][req_src[r] + 1]])
                S.add(int(req_dst[r]))
            lo = 0
            for c, n in enumerate(nums):
                for x in out[r, lo:lo + n]:
                    assert attr[int(x)][c] == attr[int(req_dst[r])][c], (name, t, r, c)
                    assert int(x) not in S, (name, t, r, int(x))
                    if unique:
                        S.add(int(x))
                lo += n
            counts[r, np.arange(count), out[r] - 100] += 1
    assert abs(used - int(GOLD[name + "_trials"])) < 0.02 * T  # both sides come up short about as often
    ref = GOLD[name + "_counts"].astype(np.int64)
    # homogeneity per (row, column): slots of one column pooled (the reference draws them from one table)
    lo = 0
    pvals = []
    for c, n in enumerate(nums):
        for r in range(req_src.shape[0]):
            a = counts[r, lo:lo + n].sum(axis=0)
            b = ref[r, lo:lo + n].sum(axis=0)
            keep = (a + b) >= 10
            assert np.array_equal((a + b) > 0, (a + b) > 0) and keep.sum() >= 2
            # ids only one side ever produced would be a support mismatch, not noise
            assert not np.any((a == 0) & (b >= 25)) and not np.any((b == 0) & (a >= 25)), (name, r, c)
            _, p, _, _ = stats.chi2_contingency(np.stack([a[keep], b[keep]]))
            pvals.append(p)
        lo += n
    assert min(pvals) > 1e-4 / len(pvals) * 10 and np.median(pvals) > 0.05, (name, sorted(pvals)[:4])


def test_short_columns_are_filled_by_the_default_sampler():
    """A condition column whose group is (almost) excluded comes up short; the row is then completed from the default
    alias table and, when even that runs dry (unique, tiny candidate list), with the default neighbour id."""
    orc = Oracle()
    ids = np.array([10, 11, 12, 13], np.int64)
    keys = np.array([[0, 0, 1, 1]], np.int64)
    out, filled = orc.cond_negative_sample(ids, None, keys, np.array([1.0], np.float32), None, np.array([0], np.int64),
                                           np.array([10], np.int64), np.array([[0]], np.int64), 3, unique=True, seed=1,
                                           call_counter=2, default_neighbor_id=-1, with_filled=True)
    # group {10, 11}: 10 is the dst, and the alias draw over [0, n - 1) only ever starts at slot 0 -> the column gives
    # nothing; the fill draws from {10, 11, 12} (slot 3 is never drawn) against the set, then -- the set dropped --
    # whatever comes, repeats included (nbr_set.clear(), conditional_negative_sampler.cc:140)
    assert filled[0] == 0
    assert set(out[0].tolist()) <= {10, 11, 12, -1} and out[0, 0] in (11, 12)
    # a key no candidate has: the column is skipped, the whole row comes from the default table
    out, filled = orc.cond_negative_sample(ids, None, keys, np.array([1.0], np.float32), None, np.array([0], np.int64),
                                           np.array([99], np.int64), np.array([[NO_KEY]], np.int64), 4, seed=1, call_counter=3,
                                           with_filled=True)
    assert filled[0] == 0 and set(out[0].tolist()) <= {10, 11, 12, 13}

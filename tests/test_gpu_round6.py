"""Round 6 additions on the device path."""
import os

import numpy as np
import pytest

import glx
from oracle_bindings import Oracle

pytestmark = pytest.mark.gpu


def _requests(rng, V, n, sg):
    """(what, ids, segment_ids) of the same length: the three answers the segment bookkeeping can give (level 0 / 1 / 2)."""
    f = n // sg
    ids = lambda: rng.integers(-2, V + 2, n).astype(np.int64)  # noqa: E731
    dense = (np.arange(n) // f).astype(np.int32)
    swapped = dense.copy()
    swapped[f - 1] = 1  # divisible, not the dense layout: arithmetic segment bounds would be wrong
    bad = np.sort(rng.integers(0, sg, n)).astype(np.int32)
    bad[n // 3] = -1  # the cursor stalls here: nothing behind it may be counted
    return [("dense spelled out", ids(), dense), ("divisible but not uniform", ids(), swapped), ("violation", ids(), bad),
            ("ragged", ids(), np.sort(rng.integers(0, sg, n)).astype(np.int32))]


def test_segment_words_survive_the_wrap_of_their_epoch_counter():
    """the two words a segment scan raises are tagged with an epoch; the tag used to come from one
    process-wide 32-bit counter, so after 2^32 calls stale words out-tagged new calls, atomicMax never landed and a
    non-uniform or invalid segment_ids request whose length divides evenly was reduced with arithmetic bounds.  The
    counter is now the buffer's own and the words are cleared (stream-ordered) when it wraps: requests issued across the
    wrap -- the test knob leaves the calling thread's buffers three epochs short of it -- still equal the oracle."""
    import torch
    orc = Oracle()
    rng = np.random.default_rng(66)
    V, dim, n, sg = 2000, 64, 6000, 600
    X = rng.standard_normal((V, dim)).astype(np.float32)
    feats = glx.Features(X)
    dev = torch.device("cuda", 0)
    reqs = _requests(rng, V, n, sg)

    def check(what, ids, seg):
        e, c = feats.aggregate("SumAggregator", torch.from_numpy(ids).to(dev), torch.from_numpy(seg).to(dev), sg, default_attr=0.5)
        we, wc = orc.aggregate(X, "SumAggregator", ids, seg, sg, default_attr=0.5)
        assert np.array_equal(c.cpu().numpy(), wc), what
        assert np.array_equal(e.cpu().numpy().view(np.uint32), we.view(np.uint32)), what

    check(*reqs[1])  # makes this (thread, stream)'s words
    glx.tune("seg_epochs_before_wrap", 3)
    for lap in range(3):  # 12 calls: the counter wraps during the first lap
        for what, ids, seg in reqs:
            check("%s (lap %d)" % (what, lap), ids, seg)
    # words raised just BEFORE the wrap (level 2, with the highest tags a buffer can carry) must not leak into the calls
    # right after it
    glx.tune("seg_epochs_before_wrap", 1)
    check(*reqs[2])
    check(*reqs[0])
    check(*reqs[1])
    feats.close()


def test_a_captured_aggregate_with_segment_ids_keeps_its_words_in_its_own_scratch():
    """the first explicit-segment_ids call on a (thread, stream) allocated its two words -- illegal
    while the stream is being captured.  A captured call now takes them from its own scratch, cleared by a node of the
    graph: capture on a stream that never ran such a call before, replay with three different inputs."""
    import torch
    orc = Oracle()
    rng = np.random.default_rng(67)
    V, dim, n, sg = 2000, 64, 6000, 600
    X = rng.standard_normal((V, dim)).astype(np.float32)
    feats = glx.Features(X)
    dev = torch.device("cuda", 0)
    reqs = _requests(rng, V, n, sg)
    ids_t = torch.zeros(n, dtype=torch.int64, device=dev)
    seg_t = torch.zeros(n, dtype=torch.int32, device=dev)
    emb = torch.empty((sg, dim), dtype=torch.float32, device=dev)
    cnt = torch.empty(sg, dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        # warm-up WITHOUT segment ids: the stream's scratch workspace exists, its segment words do not
        feats.aggregate("SumAggregator", ids_t, None, sg, out=(emb, cnt))
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        feats.aggregate("SumAggregator", ids_t, seg_t, sg, default_attr=0.5, out=(emb, cnt))
    for what, ids, seg in reqs[1:] + reqs[:1]:
        ids_t.copy_(torch.from_numpy(ids))
        seg_t.copy_(torch.from_numpy(seg))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        we, wc = orc.aggregate(X, "SumAggregator", ids, seg, sg, default_attr=0.5)
        assert np.array_equal(cnt.cpu().numpy(), wc), what
        assert np.array_equal(emb.cpu().numpy().view(np.uint32), we.view(np.uint32)), what
    del g
    feats.close()


def test_conditional_negative_sampler_starts_over_on_a_new_store(tmp_path):
    """The operator is a process-wide singleton that caches one condition table per edge type -- with a pointer to the
    store's device graph.  A second Graph in the same process with the same type name must get tables of its OWN
    candidates: with the cache of the first store (round 6: a sporadic 'graph and condition table live on different
    devices' when two test modules built their fixtures one after the other) every negative would come from the first
    graph's candidate ids."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "graph-learn_amd", "python"))
    import graphlearn as gl
    import pyapi_fixture as fx

    def build(sub, nodes, edges):
        d = str(tmp_path / sub)
        os.makedirs(d)
        cond = fx.write_cond_nodes(d, "cond_item", count=nodes)
        rel = fx.write_relation_edges(d, "relation", count=edges)  # i -> i + 2, i + 3, i + 5 for i < edges
        g = gl.Graph() \
            .node(cond, "cond_item", gl.Decoder(attr_types=["int", "int", "float", "string"], weighted=True)) \
            .edge(rel, ("cond_item", "cond_item", "cond_sim"), gl.Decoder(weighted=True), directed=True)
        g.init(tracker=d)
        return g

    def negatives(g):
        ns = g.negative_sampler("cond_sim", expand_factor=4, strategy="random", conditional=True, unique=False,
                                batch_share=False, int_cols=[0, 1], int_props=[0.25, 0.25], str_cols=[0], str_props=[0.5])
        out = []
        src, dst = np.array([1, 2, 3, 4, 5]), np.array([12, 34, 2, 67, 88])
        for cc in range(20):
            ns.set_call_counter(cc)
            ids = ns.get(src, dst).ids
            assert ids.shape == (5, 4)
            for i in range(5):
                assert ids[i, 0] % 5 == dst[i] % 5 and ids[i, 1] % 4 == dst[i] % 4 and ids[i, 2] % 3 == dst[i] % 3
            out.append(ids)
        return np.concatenate(out).reshape(-1)

    first = build("a", 200, 100)   # candidates: the distinct destinations 2 .. 104
    a = negatives(first)
    assert a.max() <= 104
    first.close()
    second = build("b", 400, 300)  # candidates: 2 .. 304
    b = negatives(second)
    assert b.max() > 104 and b.max() <= 304, "the second graph's negatives come from the first graph's candidates"
    second.close()


def test_response_pool_beside_numpy_arrays_and_pageable_copies():
    """The C++ layer's response pool registers its blocks with the GPU runtime; numpy marks every array of 4 MiB or more
    MADV_HUGEPAGE.  Registered ranges of the malloc HEAP beside such memory make later pageable host-to-device copies of the
    process fault on ROCm 7.0 within a few dozen rounds (scripts/r06/repro, the runtime alone; a test that registered numpy
    page ranges killed one GPU test run in ~10 that way: profiles/r06/crash_hunt.txt).  The pool's blocks are anonymous
    mappings of their own (explicitly since round 6; by the allocator's grace before): 60 rounds of pool-backed NeighborSampler responses beside glx.Graph builds from fresh numpy arrays of
    0.1 - 8 MB, in a process of their own, end without an error."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "r06", "pool_pageable_stress.py"), "60"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=540)
    assert r.returncode == 0 and "no error" in r.stdout, r.stdout[-3000:]

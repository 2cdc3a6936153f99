"""GPU tests of the speculation ledger (include/glx.h ABI 4, csrc/glx_dist.hip): partitioned sampling without a count
exchange.  DistributeRunner::Run waits for every shard before it stitches (graphlearn/src/core/runner/op_runner.h:86-117);
with a ledger a repeated request shape travels in fixed-capacity messages and the wait moves to the next count exchange
(the aggregation's).  Claims under test: confirmed results are bit-identical to the unpartitioned operator's; a step of
two hops + one aggregation blocks the host once instead of three times; a message that does not fit aborts the
confirmation on every rank, and the repeat fits; ranks that speculate on different requests are caught.
Ranks are threads over one GPU (glx_comm_init_local), as in test_gpu_dist_store.py."""
import numpy as np
import pytest
import torch

import glx
from test_gpu_dist_store import V, _run_ranks, world  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

B, K1, K2 = 2000, 10, 5


def _seeds(rank, step, dev):
    rng = np.random.default_rng(1000 * step + rank)
    src = rng.integers(0, V, B).astype(np.int64)
    src[:4] = [0, -1, V + 3, 10 ** 9]  # a hub, and ids no shard knows
    return torch.from_numpy(src).to(dev)


def _step(st, whole, feats, r, i, dev, sampler="EdgeWeightSampler"):
    """Two hops through the store + the aggregation that confirms them; -> True when equal to the unpartitioned result."""
    src = _seeds(r, i, dev)
    n1, e1 = st.sample(sampler, src, K1, seed=5, call_counter=2 * i)
    n2, e2 = st.sample(sampler, n1.view(-1), K2, seed=5, call_counter=2 * i + 1)
    emb, cnt = st.aggregate("MeanAggregator", n2.view(-1), None, B * K1)  # its count exchange confirms both hops
    w1, we1 = whole.sample(sampler, src, K1, seed=5, call_counter=2 * i)
    w2, we2 = whole.sample(sampler, w1.view(-1), K2, seed=5, call_counter=2 * i + 1)
    wemb, wcnt = feats.aggregate("MeanAggregator", w2.view(-1), None, B * K1)
    return bool(torch.equal(n1, w1) and torch.equal(e1, we1) and torch.equal(n2, w2) and torch.equal(e2, we2)
                and torch.equal(cnt, wcnt) and torch.equal(emb.view(torch.int32), wemb.view(torch.int32)))


@pytest.mark.parametrize("P", [1, 2, 3, 8])
@pytest.mark.parametrize("replica", [False, True])
def test_speculated_steps_equal_unpartitioned_and_block_the_host_once(world, P, replica, monkeypatch):
    monkeypatch.setenv("GLX_DIST_NO_SHORTCUT", "1")  # world 1 takes the exchange path too
    whole, feats, dev = world["whole"], world["feats"], world["dev"]
    gs, fs = world["shards"][P]
    hot = np.argsort(-world["indeg"])[:400].astype(np.int64)

    def body(r, comm):
        st = glx.DistStore(comm, graph=gs[r], features=fs[r])
        if replica:
            st.build_graph_replica(hot)
            st.set_cache(hot)
        lg = glx.Ledger(0).attach(st)
        syncs = []
        for i in range(5):
            before = st.stats()["host_syncs"]
            assert _step(st, whole, feats, r, i, dev), (r, i)
            syncs.append(st.stats()["host_syncs"] - before)
        assert syncs[0] == 3  # the first step learns the two request shapes: both hops exchange their counts
        assert syncs[1:] == [1, 1, 1, 1], syncs  # then only the aggregation does
        s = lg.stats()
        assert s["learned"] == 2 and s["speculated"] == 8 and s["aborted"] == 0 and s["holding"] == 0, s
        assert 0.0 <= s["largest_share"] <= 1.0
        # other samplers ride the same shapes
        for name in ("RandomSampler", "TopkSampler", "RandomWithoutReplacementSampler"):
            assert _step(st, whole, feats, r, 7, dev, sampler=name), (r, name)
        st.confirm()
        assert lg.stats()["speculated"] == 14
        lg.close()
        st.close()
    _run_ranks(P, body)


@pytest.mark.parametrize("P", [2, 3])
def test_overflowing_message_aborts_every_rank_and_the_repeat_fits(world, P):
    whole, feats, dev = world["whole"], world["feats"], world["dev"]
    gs, fs = world["shards"][P]

    def body(r, comm):
        st = glx.DistStore(comm, graph=gs[r], features=fs[r])
        lg = glx.Ledger(0).attach(st)
        assert _step(st, whole, feats, r, 0, dev)
        lg.set_slack(0.5, 0)  # messages half as large as the learned shares: every bucket overflows
        with pytest.raises(glx.GlxError) as ei:
            _step(st, whole, feats, r, 1, dev)
        assert ei.value.code == glx.ABORTED and "did not fit" in str(ei.value)
        assert lg.stats()["aborted"] == 1
        lg.set_slack(1.25, 1024)
        before = st.stats()["host_syncs"]
        assert _step(st, whole, feats, r, 1, dev)  # the repeat: same call counters, same answers
        assert st.stats()["host_syncs"] - before == 1  # ... still speculated
        # a lone overflow on ONE rank aborts all of them: rank 0 asks for one vertex B times
        lg.set_slack(1.0, 0)
        src = _seeds(r, 2, dev)
        if r == 0:
            src = torch.full_like(src, 1)
        st.sample("RandomSampler", src, K1, seed=5, call_counter=40)
        with pytest.raises(glx.GlxError) as ei:
            st.confirm()
        assert ei.value.code == glx.ABORTED
        lg.set_slack(1.25, 1024)
        n1, e1 = st.sample("RandomSampler", src, K1, seed=5, call_counter=40)
        st.confirm()
        w1, we1 = whole.sample("RandomSampler", src, K1, seed=5, call_counter=40)
        assert torch.equal(n1, w1) and torch.equal(e1, we1)
        lg.close()
        st.close()
    _run_ranks(P, body)


def test_ranks_that_speculate_on_different_requests_are_caught(world):
    whole, feats, dev = world["whole"], world["feats"], world["dev"]
    gs, fs = world["shards"][2]

    def body(r, comm):
        st = glx.DistStore(comm, graph=gs[r], features=fs[r])
        lg = glx.Ledger(0).attach(st)
        assert _step(st, whole, feats, r, 0, dev)
        src = _seeds(r, 1, dev)
        st.sample("RandomSampler", src, K1, seed=5, call_counter=10 + r)  # the owners answered with THEIR counter
        with pytest.raises(glx.GlxError) as ei:
            st.confirm()
        assert ei.value.code == glx.ABORTED and "differ between the ranks" in str(ei.value)
        assert lg.stats()["holding"] == 1
        # from here on every request takes its count exchange, and per-rank parameters are served as such
        before = st.stats()["host_syncs"]
        n1, e1 = st.sample("RandomSampler", src, K1, seed=5, call_counter=10 + r)
        assert st.stats()["host_syncs"] - before == 1
        w1, we1 = whole.sample("RandomSampler", src, K1, seed=5, call_counter=10 + r)
        assert torch.equal(n1, w1) and torch.equal(e1, we1)
        lg.close()
        st.close()
    _run_ranks(2, body)


def test_two_stores_share_one_ledger(world):
    """bench.py's layout: a sampling store and an aggregation store, one communicator each; the aggregation store's
    count exchange confirms the sampling store's calls."""
    whole, feats, dev = world["whole"], world["feats"], world["dev"]
    gs, fs = world["shards"][2]
    key2 = 990001

    def body(r, comm):
        comm2 = glx.Comm.local(key2, 0, r, 2)
        st_s = glx.DistStore(comm, graph=gs[r])
        st_a = glx.DistStore(comm2, features=fs[r])
        lg = glx.Ledger(0).attach(st_s, st_a)
        for i in range(3):
            src = _seeds(r, i, dev)
            b_s, b_a = st_s.stats()["host_syncs"], st_a.stats()["host_syncs"]
            n1, e1 = st_s.sample("EdgeWeightSampler", src, K1, seed=5, call_counter=2 * i)
            n2, e2 = st_s.sample("EdgeWeightSampler", n1.view(-1), K2, seed=5, call_counter=2 * i + 1)
            emb, cnt = st_a.aggregate("SumAggregator", n2.view(-1), None, B * K1)
            w1, _ = whole.sample("EdgeWeightSampler", src, K1, seed=5, call_counter=2 * i)
            w2, _ = whole.sample("EdgeWeightSampler", w1.view(-1), K2, seed=5, call_counter=2 * i + 1)
            wemb, _ = feats.aggregate("SumAggregator", w2.view(-1), None, B * K1)
            assert torch.equal(n2, w2) and torch.equal(emb.view(torch.int32), wemb.view(torch.int32))
            assert st_s.stats()["host_syncs"] - b_s == (2 if i == 0 else 0)
            assert st_a.stats()["host_syncs"] - b_a == 1
        lg.close()
        st_s.close()
        st_a.close()
    _run_ranks(2, body)


@pytest.mark.parametrize("P", [2, 3])
def test_a_rank_with_a_shorter_tail_batch_speculates_along_with_its_peers(world, P):
    """the decision to speculate must not depend on a rank's own request length -- one rank with a tail
    batch (or an empty request) would otherwise take the count exchange while its peers send fixed-capacity messages:
    mismatched collectives.  Keyed by the call's position in the step, every rank takes the same path; the short rank's
    answers still equal the unpartitioned operator's."""
    whole, feats, dev = world["whole"], world["feats"], world["dev"]
    gs, fs = world["shards"][P]

    def body(r, comm):
        st = glx.DistStore(comm, graph=gs[r], features=fs[r])
        lg = glx.Ledger(0).attach(st)
        assert _step(st, whole, feats, r, 0, dev)  # learns positions 0 and 1
        for i, n_tail in ((1, 137), (2, 0), (3, B)):  # a short batch, an empty one, a full one again -- on rank 0 only
            src = _seeds(r, i, dev)
            if r == 0:
                src = src[:n_tail].contiguous()
            before = st.stats()["host_syncs"]
            n1, e1 = st.sample("EdgeWeightSampler", src, K1, seed=5, call_counter=2 * i)
            n2, e2 = st.sample("EdgeWeightSampler", n1.view(-1), K2, seed=5, call_counter=2 * i + 1)
            st.confirm()
            assert st.stats()["host_syncs"] - before == 1, (r, i)  # both hops speculated on every rank
            w1, we1 = whole.sample("EdgeWeightSampler", src, K1, seed=5, call_counter=2 * i)
            w2, we2 = whole.sample("EdgeWeightSampler", w1.view(-1), K2, seed=5, call_counter=2 * i + 1)
            assert torch.equal(n1, w1) and torch.equal(e1, we1) and torch.equal(n2, w2) and torch.equal(e2, we2), (r, i)
        s = lg.stats()
        assert s["aborted"] == 0 and s["holding"] == 0 and s["speculated"] == 6, s
        # a LONGER request than the position has seen may not fit: that is an abort on every rank, and the repeat fits
        src = torch.cat([_seeds(r, 9, dev)] * 6) if r == 0 else _seeds(r, 9, dev)
        try:
            n1, e1 = st.sample("RandomSampler", src, K1, seed=5, call_counter=77)
            st.confirm()
        except glx.GlxError as ex:
            assert ex.code == glx.ABORTED
            n1, e1 = st.sample("RandomSampler", src, K1, seed=5, call_counter=77)
            st.confirm()
        w1, we1 = whole.sample("RandomSampler", src, K1, seed=5, call_counter=77)
        assert torch.equal(n1, w1) and torch.equal(e1, we1)
        lg.close()
        st.close()
    _run_ranks(P, body)


def _fuzz_cases(n):
    import os
    first = int(os.environ.get("GLX_FUZZ_FIRST", "0"))
    return list(range(first, first + int(os.environ.get("GLX_FUZZ_CASES", str(n)))))


@pytest.mark.parametrize("case", _fuzz_cases(12))
def test_ledger_fuzz_ragged_windows(world, case):
    """Windows of 1-3 speculated hops closed by a confirmation, with request lengths that differ from rank to rank and
    from step to step -- empty requests, tails, requests several times longer than anything the position has seen (an
    overflow: GLX_ABORTED on EVERY rank, the window is repeated until it fits).  Every rank issues the same sequence
    of calls; what each ends up with equals the unpartitioned operator's answer, and nobody waits for a collective a
    peer never enters."""
    whole, dev = world["whole"], world["dev"]
    rng = np.random.default_rng(6200 + case)
    P = int(rng.choice([2, 3, 8]))
    gs, _ = world["shards"][P]
    hops = int(rng.integers(1, 4))
    fan = [int(x) for x in rng.choice([1, 2, 5, 10], hops)]
    steps = int(rng.integers(3, 7))
    names = [str(x) for x in rng.choice(["RandomSampler", "EdgeWeightSampler", "TopkSampler", "RandomWithoutReplacementSampler"], hops)]
    base = int(rng.choice([40, 600, 2500]))
    # lengths[step][rank]: mostly around `base`, sometimes empty, sometimes several times longer
    lengths = []
    for i in range(steps):
        row = []
        for r in range(P):
            u = rng.random()
            row.append(0 if u < 0.12 else int(base * rng.integers(3, 7)) if u > 0.9 else int(rng.integers(1, base + 1)))
        lengths.append(row)

    def body(r, comm):
        st = glx.DistStore(comm, graph=gs[r])
        lg = glx.Ledger(0).attach(st)
        for i in range(steps):
            rr = np.random.default_rng(97 * case + 13 * i + r)
            src = torch.from_numpy(rr.integers(-2, V + 2, lengths[i][r]).astype(np.int64)).to(dev)
            for attempt in range(8):
                try:
                    cur, outs = src, []
                    for h in range(hops):
                        n, e = st.sample(names[h], cur, fan[h], seed=case, call_counter=10 * i + h, default_neighbor_id=-1)
                        outs.append((n, e))
                        cur = n.view(-1)
                    st.confirm()
                    break
                except glx.GlxError as ex:
                    assert ex.code == glx.ABORTED, str(ex)  # on every rank alike: all repeat the window
            else:
                raise AssertionError("the window never fitted")
            cur = src
            for h in range(hops):
                wn, we = whole.sample(names[h], cur, fan[h], seed=case, call_counter=10 * i + h, default_neighbor_id=-1)
                assert torch.equal(outs[h][0], wn) and torch.equal(outs[h][1], we), (case, r, i, h)
                cur = wn.view(-1)
        assert lg.stats()["holding"] == 0
        lg.close()
        st.close()
    _run_ranks(P, body)


@pytest.mark.parametrize("P", [2, 3])
def test_random_walks_never_speculate_even_with_taught_positions(world, P):
    """a DeepWalk step is a neighbor_count-1 sampling request, and position keys do not depend on the request
    length -- so after two training hops have taught positions 0 and 1, a walk's first steps would have skipped their
    count exchange on capacities learned from those hops.  glx.h promises walks always exchange counts: the steps run
    with the ledger detached.  Walks equal the unpartitioned operator's, no call of the walk is counted as speculated,
    every step blocks the host once, and the ledger keeps working for the training hops afterwards."""
    whole, feats, dev = world["whole"], world["feats"], world["dev"]
    gs, fs = world["shards"][P]
    walk_len = 6

    def body(r, comm):
        st = glx.DistStore(comm, graph=gs[r], features=fs[r])
        lg = glx.Ledger(0).attach(st)
        for i in range(2):
            assert _step(st, whole, feats, r, i, dev), (r, i)  # positions 0 and 1 are learned and speculated on
        before = lg.stats()
        assert before["learned"] == 2 and before["speculated"] == 2
        seeds = _seeds(r, 9, dev)[:500].contiguous()
        syncs0 = st.stats()["host_syncs"]
        walks = st.random_walk(seeds, walk_len, seed=5, call_counter=100)
        syncs1 = st.stats()["host_syncs"]
        want = whole.random_walk(seeds, walk_len, seed=5, call_counter=100)
        assert torch.equal(walks, want)
        after = lg.stats()
        assert after["speculated"] == before["speculated"] and after["learned"] == before["learned"], (before, after)
        assert syncs1 - syncs0 == walk_len  # one count exchange per step
        assert _step(st, whole, feats, r, 3, dev)  # the ledger is attached again: the hops speculate as before
        assert lg.stats()["speculated"] == before["speculated"] + 2 and lg.stats()["aborted"] == 0
        lg.close()
        st.close()
    _run_ranks(P, body)

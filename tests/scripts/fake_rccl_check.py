"""glx's RCCL transport (csrc/glx_comm.hip, RcclComm) with world size > 1 on ONE GPU: librccl is replaced by the
in-process stand-in of tests/fake_rccl (GLX_RCCL_LIBRARY, set by the caller), ranks are host threads.  What is under
test is glx's own call pattern -- groups of ncclSend / ncclRecv with per-peer offsets and counts, messages cut into
rounds, several segments per group, the count all-gather -- through the distributed store on top: every rank's
sampling / aggregation / lookup answers must equal the unpartitioned operators', bit for bit.

  GLX_RCCL_LIBRARY=tests/fake_rccl/libfakerccl.so python tests/scripts/fake_rccl_check.py [P ...]
"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
assert os.environ.get("GLX_RCCL_LIBRARY"), "set GLX_RCCL_LIBRARY to the stand-in library"
import numpy as np  # noqa: E402
import torch  # noqa: E402
import glx  # noqa: E402
import dist as gdist  # noqa: E402
import synth  # noqa: E402

V, D = 5000, 64
dev = torch.device("cuda", 0)
rp, col, eid, w = (torch.from_numpy(a).to(dev) for a in synth.small_graph(V, 80000, seed=21, weighted=True, hub_degree=3000))
X = torch.from_numpy(np.random.default_rng(4).standard_normal((V, D)).astype(np.float32)).to(dev)
whole, feats = glx.Graph(rp, col, eid, w), glx.Features(X)
indeg = torch.bincount(col, minlength=V)
hot = torch.topk(indeg, 500).indices.to(torch.int64)
rrp, rcol, reid, rw, rids = gdist.rows_of_graph(rp, col, eid, w, hot)
replica = glx.Graph(rrp, rcol, reid, rw, ids=rids)


def run(P, max_message_bytes):
    uid = glx.Comm.unique_id()
    shards = []
    for r in range(P):
        srp, scol, seid, sw, sids = gdist.shard_graph(rp, col, eid, w, r, P)
        shards.append((glx.Graph(srp, scol, seid, sw, ids=sids), glx.Features(X[r::P].contiguous(), ids=sids)))
    errors = [None] * P

    def main(r):
        try:
            comm = glx.Comm.rccl(0, r, P, uid)
            assert comm.transport == glx.COMM_RCCL, comm.transport
            if max_message_bytes:
                comm.set_max_message_bytes(max_message_bytes)
            with torch.cuda.stream(torch.cuda.Stream(device=0)):
                g, f = shards[r]
                st = glx.DistStore(comm, graph=g, features=f)
                rng = np.random.default_rng(100 + r)
                src = torch.from_numpy(np.concatenate([rng.integers(0, V, 3000 + 37 * r), [0, -1, V, 10 ** 9]]).astype(np.int64)).to(dev)
                for with_replicas in (False, True):
                    if with_replicas:
                        got_hot = st.hot_ids(400)
                        assert got_hot.shape[0] == 400
                        st.set_cache(got_hot)
                        built = st.build_graph_replica(hot)  # collective: rows cut out of the shards, all-gathered
                        assert built.num_edges == replica.num_edges
                    cc = 0
                    for name in glx.SAMPLER_IDS:
                        for k, pad in ((10, 1), (7, 0)):
                            cc += 1
                            n1, e1 = st.sample(name, src, k, seed=9, call_counter=cc, padding_mode=pad, default_neighbor_id=-5)
                            rn, re = whole.sample(name, src, k, seed=9, call_counter=cc, padding_mode=pad, default_neighbor_id=-5)
                            assert torch.equal(n1, rn) and torch.equal(e1, re), (name, k, pad, r, with_replicas)
                    ids = n1.reshape(-1).contiguous()
                    for op in ("SumAggregator", "MaxAggregator", "MeanAggregator"):
                        e, c = st.aggregate(op, ids, None, src.shape[0], default_attr=0.5)
                        re_, rc_ = feats.aggregate(op, ids, None, src.shape[0], default_attr=0.5)
                        assert torch.equal(c, rc_) and torch.equal(e.view(torch.int32), re_.view(torch.int32)), (op, r)
                    for a, b in zip(st.sample_full(src, 4), whole.sample_full(src, 4)):  # ragged values back
                        assert torch.equal(a, b), ("full", r)
                    rows = st.lookup(src, default_attr=-1.0)
                    assert torch.equal(rows.view(torch.int32), feats.lookup(src, -1.0).view(torch.int32)), r
                    # a request only SOME ranks have rows for: the others take part with nothing to send
                    part = src[: (0 if r % 2 else 50)]
                    pn, _ = st.sample("TopkSampler", part, 3, seed=1, call_counter=5)
                    assert torch.equal(pn, whole.sample("TopkSampler", part, 3, seed=1, call_counter=5)[0])
                    stats = st.stats()
                    assert stats["exchange_rounds"] >= 1
                # equal-length requests in lockstep with a speculation ledger: fixed-size send / recv groups, no count
                # all-gather for the sampling calls, the aggregation's confirms them
                lg = glx.Ledger(0).attach(st)
                for i in range(4):
                    eq = torch.from_numpy(np.random.default_rng(7 * i + r).integers(0, V, 2000).astype(np.int64)).to(dev)
                    before = st.stats()["host_syncs"]
                    n1, e1 = st.sample("EdgeWeightSampler", eq, 8, seed=3, call_counter=2 * i)
                    n2, e2 = st.sample("EdgeWeightSampler", n1.view(-1), 4, seed=3, call_counter=2 * i + 1)
                    e, c = st.aggregate("MaxAggregator", n2.view(-1), None, 16000)
                    assert st.stats()["host_syncs"] - before == (3 if i == 0 else 1), i
                    w1, _ = whole.sample("EdgeWeightSampler", eq, 8, seed=3, call_counter=2 * i)
                    w2, we2 = whole.sample("EdgeWeightSampler", w1.view(-1), 4, seed=3, call_counter=2 * i + 1)
                    we, wc = feats.aggregate("MaxAggregator", w2.view(-1), None, 16000)
                    assert torch.equal(n2, w2) and torch.equal(e2, we2) and torch.equal(e.view(torch.int32), we.view(torch.int32)), (r, i)
                assert lg.stats()["speculated"] == 6 and lg.stats()["aborted"] == 0
                lg.close()
                torch.cuda.current_stream().synchronize()
                st.close()
            comm.close()
        except BaseException as ex:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            errors[r] = ex
    ts = [threading.Thread(target=main, args=(r,)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(400)
    assert not any(t.is_alive() for t in ts), "a rank hung"
    for e in errors:
        if e is not None:
            raise e


for P in [int(a) for a in sys.argv[1:]] or [2, 3, 8]:
    run(P, 0)
    run(P, 4096)  # every message beyond 4 KiB goes out in several rounds
    print("fake-rccl ok: world size %d" % P, flush=True)

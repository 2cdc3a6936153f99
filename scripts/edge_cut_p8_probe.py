"""The edge-cut path at its real shape on ONE GPU: P ranks = P threads of this process (in-process transport,
device-to-device copies instead of xGMI), full C3 (RMAT 10M / 100M, EdgeWeight [25,10] + Max, dim 256), every
rank driving its own 65,536-seed batch like bench.py --gpus P.  All ranks share the GPU, so wall time per step
/ P is an UPPER BOUND on a rank's cost without link time (the in-process transport synchronises the stream and
meets the other threads at two host barriers per exchange, and the eight threads share one Python interpreter).  Also checks every rank's answer against the
unpartitioned operators, bit for bit, and prints where the ids came from (replica / own shard / halo).

  python scripts/edge_cut_p8_probe.py [P=8] [hot_fraction=0.10] [steps=6] [solo]

solo: only rank 0 asks; the other ranks take part in every collective with empty requests and serve rank 0's.
The wall time per step is then rank 0's own work plus the service work all owners do for it -- in the symmetric
real case exactly one GPU's share -- with no kernels of other ranks' requests running beside it.
"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import numpy as np, torch, glx, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hot_fraction = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
solo = len(sys.argv) > 4 and sys.argv[4] == "solo"
# sym: every rank asks (the symmetric real case) and ALL ranks enqueue on ONE stream, so no two kernels overlap and
# every kernel runs at its own duration; a rocprofv3 kernel trace filtered by rank 0's host thread (printed below,
# scripts/r03/p8_solo_step.py --thread) is then exactly ONE rank's step at P = 8: its own request plus its service of
# the seven peers' rows -- one sampling launch per hop and one row gather for all peers, as a real rank issues them
# (solo spreads that service over eight owner "ranks": eight launches per hop of an eighth of the rows each).
sym = len(sys.argv) > 4 and sys.argv[4] == "sym"
dev = torch.device("cuda", 0)
V, E, D, B0, k1, k2 = 10_000_000, 100_000_000, 256, 65536, 25, 10
src, dst, w = synth.rmat_edges_torch(V, E, 4, dev)
pool = torch.unique(src)
whole = glx.Graph.from_edges(src, dst, w)
X = synth.features_torch(V, D, 5, dev)
feats = glx.Features(X)
graphs, fshards = [], []
for r in range(P):
    own = (src % P) == r
    graphs.append(glx.Graph.from_edges(src[own].contiguous(), dst[own].contiguous(), w[own].contiguous(),
                                       edge_ids=torch.nonzero(own).view(-1)))
    ids = torch.arange(r, V, P, dtype=torch.int64, device=dev)
    fshards.append(glx.Features(X[r::P].contiguous(), ids=ids))
    del own, ids
graph_replica_on = os.environ.get("GRAPH_REPLICA", "0") == "1"  # also replicate the hot vertices' adjacency rows
del src, dst, w
del X
torch.cuda.empty_cache()
n1, n2 = B0 * k1, B0 * k1 * k2
bar = threading.Barrier(P)
shared_stream = torch.cuda.Stream(device=0) if sym else None
hot_by = os.environ.get("HOT_BY", "indegree")
acc_all = torch.zeros(V, dtype=torch.int32, device=dev)
acc_lock = threading.Lock()
hot_shared = [None]
times, stats, ok = [None] * P, [None] * P, [True] * P


def rank_main(r):
    try:
        comm = glx.Comm.local(4242, 0, r, P)
        if sym and r == 0:
            print("rank 0 host thread id: %d" % threading.get_native_id(), flush=True)
        with torch.cuda.stream(shared_stream if sym else torch.cuda.Stream(device=0)):
            st_s = glx.DistStore(comm, graph=graphs[r])
            st_a = glx.DistStore(comm, features=fshards[r])
            if hot_by == "access" and hot_fraction > 0:
                # rows ranked by how often profiling requests touch them (bench.py --hot-by access)
                pg = torch.Generator(device=dev)
                pg.manual_seed(77 + r)
                for j in range(4):
                    ps = pool[torch.randint(0, pool.shape[0], (B0,), generator=pg, device=dev)]
                    p1, _ = st_s.sample("EdgeWeightSampler", ps, k1, seed=4242, call_counter=2 * j)
                    p2, _ = st_s.sample("EdgeWeightSampler", p1.view(-1), k2, seed=4242, call_counter=2 * j + 1)
                    c = torch.bincount(p2.view(-1), minlength=V) + torch.bincount(p1.view(-1), minlength=V)
                    torch.cuda.current_stream().synchronize()
                    with acc_lock:
                        acc_all.add_(c.to(torch.int32))
                        torch.cuda.current_stream().synchronize()
                bar.wait()
                if r == 0:
                    hot_shared[0] = torch.topk(acc_all, int(V * hot_fraction)).indices.to(torch.int64).cpu().numpy()
                bar.wait()
                hot = hot_shared[0]
            else:
                hot = st_s.hot_ids(int(V * hot_fraction))
            st_a.set_cache(hot)
            if graph_replica_on:
                # collective: every owner cuts its hot vertices' rows out of its shard, the pieces are all-gathered
                mine = st_s.build_graph_replica(hot)
                if r == 0:
                    print("graph replica built from the shards: %d of %d edges" % (mine.num_edges, E), flush=True)
            gen = torch.Generator(device=dev)
            gen.manual_seed(1000 + r)
            b0 = B0 if (r == 0 or not solo) else 0
            m1, m2 = b0 * k1, b0 * k1 * k2
            seeds = pool[torch.randint(0, pool.shape[0], (steps + 2, b0), generator=gen, device=dev)]
            nb1 = torch.empty((b0, k1), dtype=torch.int64, device=dev); ed1 = torch.empty_like(nb1)
            nb2 = torch.empty((m1, k2), dtype=torch.int64, device=dev); ed2 = torch.empty_like(nb2)
            emb2 = torch.empty((m1, D), dtype=torch.float32, device=dev); cnt2 = torch.empty(m1, dtype=torch.int32, device=dev)
            emb1 = torch.empty((b0, D), dtype=torch.float32, device=dev); cnt1 = torch.empty(b0, dtype=torch.int32, device=dev)

            ledger = glx.Ledger(0).attach(st_s, st_a) if os.environ.get("LEDGER", "0") == "1" else None
            merged = os.environ.get("MERGED", "0") == "1"  # bench.py's step: ONE aggregate_begin for both id sets
            both = torch.empty(m2 + m1, dtype=torch.int64, device=dev)
            if merged:
                nb2 = both[:m2].view(m1, k2)
                nb1 = both[m2:].view(b0, k1)

            def step_merged(i):
                st_s.sample("EdgeWeightSampler", seeds[i], k1, seed=42, call_counter=4 * i, out=(nb1, ed1))
                st_s.sample("EdgeWeightSampler", nb1.view(-1), k2, seed=42, call_counter=4 * i + 1, out=(nb2, ed2))
                st_a.aggregate_begin(0, both)
                s2 = st_a.stats()
                st_a.aggregate_end_range(0, 0, m2, "MaxAggregator", None, m1, out=(emb2, cnt2))
                st_a.aggregate_end_range(0, m2, m1, "MaxAggregator", None, b0, out=(emb1, cnt1), release=True)
                return s2

            def step(i):
                if merged:
                    return step_merged(i)
                st_s.sample("EdgeWeightSampler", seeds[i], k1, seed=42, call_counter=4 * i, out=(nb1, ed1))
                st_s.sample("EdgeWeightSampler", nb1.view(-1), k2, seed=42, call_counter=4 * i + 1, out=(nb2, ed2))
                st_a.aggregate("MaxAggregator", nb2.view(-1), None, m1, out=(emb2, cnt2))
                s2 = st_a.stats()
                st_a.aggregate("MaxAggregator", nb1.view(-1), None, b0, out=(emb1, cnt1))
                return s2
            for i in range(2):
                step(i)
            torch.cuda.current_stream().synchronize()
            bar.wait()
            sync0 = (st_s.stats()["host_syncs"], st_a.stats()["host_syncs"])
            t0 = time.perf_counter()
            for i in range(2, steps + 2):
                s2 = step(i)
            torch.cuda.current_stream().synchronize()
            bar.wait()
            times[r] = (time.perf_counter() - t0) / steps
            if r == 0:
                print("count exchanges per step: sampling store %.1f, aggregation store %.1f%s" % (
                    (st_s.stats()["host_syncs"] - sync0[0]) / steps, (st_a.stats()["host_syncs"] - sync0[1]) / steps,
                    "; ledger: %s" % ledger.stats() if ledger else ""), flush=True)
            stats[r] = dict(s2, sampling_hop2=st_s.last_sample_rows())
            # bit-identical to the unpartitioned operators (last step)
            i = steps + 1
            wa, wae = whole.sample("EdgeWeightSampler", seeds[i], k1, seed=42, call_counter=4 * i)
            wb, wbe = whole.sample("EdgeWeightSampler", wa.view(-1), k2, seed=42, call_counter=4 * i + 1)
            we2, wc2 = feats.aggregate("MaxAggregator", wb.view(-1), None, m1)
            torch.cuda.current_stream().synchronize()
            ok[r] = bool(torch.equal(nb1, wa) and torch.equal(ed1, wae) and torch.equal(nb2, wb) and torch.equal(ed2, wbe)
                         and torch.equal(cnt2, wc2) and torch.equal(emb2.view(torch.int32), we2.view(torch.int32)))
            if ledger:
                ledger.close()
            st_s.close(); st_a.close()
        comm.close()
    except BaseException:
        import traceback
        traceback.print_exc()
        ok[r] = False
        try:
            bar.abort()
        except Exception:
            pass


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in ts: t.start()
for t in ts: t.join(600)
if sym:
    print("P = %d ranks on one GPU, hot fraction %.2f, every rank asks, ONE stream (kernels never overlap): %.2f ms per step of "
          "all ranks = %.2f ms per rank-step; answers equal the unpartitioned operators: %s"
          % (P, hot_fraction, max(x or 0 for x in times) * 1e3, max(x or 0 for x in times) * 1e3 / P, all(ok)))
elif solo:
    print("P = %d ranks on one GPU, hot fraction %.2f, ONLY rank 0 asks: %.2f ms per step = one rank's own work + the service "
          "work of all owners for it (= one GPU's share in the symmetric case), no link time; answers equal the unpartitioned "
          "operators: %s" % (P, hot_fraction, max(x or 0 for x in times) * 1e3, all(ok)))
else:
    print("hot rows chosen by", hot_by)
    print("P = %d ranks on one GPU, hot fraction %.2f: %.2f ms per step with all ranks running (%.2f ms of wall time per rank-step: host barriers and one GPU shared by all ranks included); "
          "all answers equal the unpartitioned operators: %s" % (P, hot_fraction, max(x or 0 for x in times) * 1e3,
                                                                  max(x or 0 for x in times) * 1e3 / P, all(ok)))
for r in (0, P - 1):
    print("rank %d hop-2 request:" % r, stats[r])
import json
print(json.dumps({"p8_probe": {"ranks": P, "mode": "sym" if sym else ("solo" if solo else "all"), "hot_fraction": hot_fraction,
                               "steps": steps, "ms_per_step_all_ranks": max(x or 0 for x in times) * 1e3,
                               "ms_per_rank_step": max(x or 0 for x in times) * 1e3 / (1 if solo else P),
                               "answers_equal_unpartitioned": bool(all(ok)),
                               "merged_aggregation": os.environ.get("MERGED", "0") == "1",
                               "ledger": os.environ.get("LEDGER", "0") == "1",
                               "graph_replica": graph_replica_on}}))
sys.exit(0 if all(ok) else 1)

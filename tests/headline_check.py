"""Checker (test infrastructure): oracle parity of the HIP path AT THE SIZE THE METRIC IS QUOTED ON.

The bit-exact suites (test_gpu_parity.py ...) run on graphs the oracle can hold whole.  Here the
graph is BASELINE.json's (100 M edges) and the oracle answers a SUBSET of the request rows of one
real 2-hop step, on a sub-graph cut from the RAW EDGE LIST -- not from the device CSR -- so the
device build (row order incl. weight ties, edge ids, alias tables) is inside what is checked:

  raw edges  --mask(src in needed ids)-->  host sub-CSR (insertion order)
             --oracle sort_rows (MemoryAdjMatrix::Sort, memory_adj_matrix.cc:105-125)
             --oracle alias_build (AliasMethod::Build, alias_method.cc:57-107)
             --oracle sample with rng_rows = the rows' indices in the full request
  == the HIP outputs at those rows, bit for bit (edge_weight_sampler.cc:31-92,
     random_sampler.cc:33-76, random_without_replacement_sampler.cc:31-75, topk_sampler.cc:29-68);
  oracle aggregate over the selected segments (aggregator.cc:25-59, max_aggregator.cc:26-40,
     mean_aggregator.cc:26-61, sum_aggregator.cc:25-33) on the feature rows they touch
  == the HIP embeddings / counts of those segments, bit for bit.

Used by tests/test_gpu_fullsize_oracle.py and by `bench.py --verify` (outside the timed region).
torch is plumbing here (masking the edge list on the device, moving the subset to the host).
"""
import numpy as np
import torch

from oracle_bindings import Oracle


def sub_csr(orc, src, dst, weight, need_ids, sort_by_weight=True, with_alias=True):
    """Rows of `need_ids` (sorted unique int64 numpy) cut from the raw edge list (device tensors,
    insertion order = edge id order) -> oracle graph dict with an id map."""
    dev = src.device
    need = torch.from_numpy(need_ids).to(dev)
    hi = int(max(int(src.max().item()), int(need_ids.max()) if need_ids.size else 0)) + 1
    flag = torch.zeros(hi, dtype=torch.bool, device=dev)
    flag[need[(need >= 0) & (need < hi)]] = True
    step = 1 << 30  # torch.nonzero takes at most 2^31 elements: the 2.3 B-edge list goes through in pieces
    idx = torch.cat([torch.nonzero(flag[src[lo:lo + step]]).view(-1) + lo for lo in range(0, max(1, src.shape[0]), step)])
    # ascending = insertion order
    s = src[idx].cpu().numpy()
    d = dst[idx].cpu().numpy()
    e = idx.cpu().numpy().astype(np.int64)
    w = weight[idx].cpu().numpy() if weight is not None else None
    del flag, idx
    order = np.argsort(s, kind="stable")  # rows grouped, insertion order kept inside a row
    s, d, e = s[order], d[order], e[order]
    if w is not None:
        w = np.ascontiguousarray(w[order])
    row_ptr = np.zeros(need_ids.shape[0] + 1, np.int64)
    np.add.at(row_ptr, np.searchsorted(need_ids, s) + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    col, eid = np.ascontiguousarray(d), np.ascontiguousarray(e)
    if w is not None and sort_by_weight:
        col, eid, w = orc.sort_rows(row_ptr, col, eid, w)
    g = dict(row_ptr=row_ptr, col=col, eid=eid, weight=w, ids=np.ascontiguousarray(need_ids))
    if w is not None and with_alias:
        g["alias"] = orc.alias_build(row_ptr, w)
    return g


def _pick(n, want, gen):
    want = min(want, n)
    return np.sort(gen.choice(n, want, replace=False)).astype(np.int64)


def check_sample(orc, edges, sampler, k, request, nbr, eid, seed, call_counter, rows, padding_mode=1,
                 default_neighbor_id=0):
    """Request rows `rows` (sorted int64 numpy indices into `request`, a device tensor of source ids) of one sampler
    call against the oracle on the sub-graph of their source ids, cut from the raw edge list.
    -> (equal, edges in the sub-graph)"""
    src, dst, weight = edges
    t = torch.from_numpy(rows).to(request.device)
    ids = np.ascontiguousarray(request.reshape(-1)[t].cpu().numpy())
    g = sub_csr(orc, src, dst, weight, np.unique(ids), with_alias=sampler == "EdgeWeightSampler")
    o, oe = orc.sample(g, sampler, ids, k, seed=seed, call_counter=call_counter, rng_rows=rows,
                       padding_mode=padding_mode, default_neighbor_id=default_neighbor_id)
    ok = np.array_equal(o, nbr.reshape(-1, k)[t].cpu().numpy()) and np.array_equal(oe, eid.reshape(-1, k)[t].cpu().numpy())
    return ok, int(g["col"].shape[0])


def check_aggregate(orc, features_of, aggregator, ids_2d, emb, cnt, segs, default_attr=0.0):
    """Segments `segs` (sorted int64 numpy indices) of one aggregate call whose segment i reduces ids_2d[i, :]
    (a dense sampler response) against the oracle on the feature rows those segments touch."""
    f = ids_2d.shape[1]
    t = torch.from_numpy(segs).to(ids_2d.device)
    ids = ids_2d[t].reshape(-1)  # the segments' ids, in request order
    uniq = torch.unique(ids)
    X = features_of(uniq).cpu().numpy()
    ids_h = ids.cpu().numpy()
    seg = (np.arange(ids_h.shape[0]) // f).astype(np.int32)
    oemb, ocnt = orc.aggregate(X, aggregator, ids_h, seg, segs.shape[0], default_attr=default_attr,
                               ids=np.ascontiguousarray(uniq.cpu().numpy()))
    return bool(np.array_equal(ocnt, cnt[t].cpu().numpy())
                and np.array_equal(oemb.view(np.uint32), emb[t].cpu().numpy().view(np.uint32)))


def check_step(edges, features_of, sampler, fanout, aggregator, seeds, out, seed, call_counters,
               rows_hop1=4096, rows_hop2=8192, segments=16384, hub_ids=None, hub_rows=2048, rng_seed=99,
               padding_mode=1, default_neighbor_id=0, default_attr=0.0):
    """One real 2-hop step against the oracle on a row subset.

    edges        (src, dst, weight|None) device tensors of the raw edge list (edge id = index)
    features_of  callable(ids: int64 device tensor) -> [len, D] float32 device tensor of the raw rows
                 (the generator's, NOT glx_lookup)
    seeds        [B0] device tensor;  out = dict(n1, e1, n2, e2 [, emb2, cnt2, emb1, cnt1]) device tensors
    call_counters (cc of hop 1, cc of hop 2)
    -> dict(ok, rows_hop1, rows_hop2, segments_hop2, segments_hop1, edges_in_subgraph, mismatches)
    """
    orc = Oracle()
    k1, k2 = fanout
    gen = np.random.default_rng(rng_seed)
    B0 = seeds.shape[0]
    n1, e1, n2, e2 = out["n1"], out["e1"], out["n2"], out["e2"]
    req2 = n1.reshape(-1)
    r1 = _pick(B0, rows_hop1, gen)
    r2 = _pick(req2.shape[0], rows_hop2, gen)
    if hub_ids is not None and len(hub_ids):
        # request rows that ask for a hub (the longest alias tables, the rows with weight ties)
        hub = torch.from_numpy(np.asarray(hub_ids, np.int64)).to(req2.device)
        pos = torch.nonzero(torch.isin(req2, hub)).view(-1).cpu().numpy()
        if pos.shape[0] > hub_rows:
            pos = pos[gen.choice(pos.shape[0], hub_rows, replace=False)]
        r2 = np.unique(np.concatenate([r2, pos.astype(np.int64)]))
    bad = []
    kw = dict(padding_mode=padding_mode, default_neighbor_id=default_neighbor_id)
    ok1, ne1 = check_sample(orc, edges, sampler, k1, seeds, n1, e1, seed, call_counters[0], r1, **kw)
    if not ok1:
        bad.append("hop-1 sample")
    ok2, ne2 = check_sample(orc, edges, sampler, k2, req2, n2, e2, seed, call_counters[1], r2, **kw)
    if not ok2:
        bad.append("hop-2 sample")
    res = dict(rows_hop1=int(r1.shape[0]), rows_hop2=int(r2.shape[0]), edges_in_subgraph=ne1 + ne2,
               segments_hop2=0, segments_hop1=0)
    if aggregator is not None and "emb2" in out:
        for name, ids_t, emb, cnt, want in (("hop-2", n2, out["emb2"], out["cnt2"], segments),
                                            ("hop-1", n1, out["emb1"], out["cnt1"], max(1, segments // 8))):
            sg = _pick(ids_t.shape[0], want, gen)
            if not check_aggregate(orc, features_of, aggregator, ids_t, emb, cnt, sg, default_attr=default_attr):
                bad.append(name + " aggregate")
            res["segments_" + name.replace("-", "")] = int(sg.shape[0])
    res["mismatches"] = bad
    res["ok"] = not bad
    return res


def hub_rows_equal(orc, graph, edges, hub_ids):
    """The device store's rows of `hub_ids` (FullSampler = storage order) and their alias tables against the oracle's
    build of the same rows from the raw edge list.  -> (rows_equal, alias_equal, edges)."""
    src, dst, weight = edges
    hub = np.unique(np.asarray(hub_ids, np.int64))
    g = sub_csr(orc, src, dst, weight, hub)
    deg, nbr, eid = graph.sample_full(torch.from_numpy(hub).to(src.device))
    deg, nbr, eid = (x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x) for x in (deg, nbr, eid))
    rows_ok = (np.array_equal(deg.astype(np.int64), np.diff(g["row_ptr"])) and np.array_equal(nbr, g["col"])
               and np.array_equal(eid, g["eid"]))
    alias_ok = None
    if weight is not None:
        prob, alias = graph.export_alias()
        # slot offsets of vertex v in a source-ordered CSR = number of edges whose source id is < v
        starts = torch.cumsum(torch.bincount(src, minlength=int(hub.max()) + 2), 0)
        starts = torch.cat([torch.zeros(1, dtype=starts.dtype, device=starts.device), starts])[torch.from_numpy(hub).to(src.device)]
        starts = starts.cpu().numpy()
        alias_ok = True
        for r in range(hub.shape[0]):
            a, b = g["row_ptr"][r], g["row_ptr"][r + 1]
            s = starts[r]
            if not (np.array_equal(prob[s:s + b - a].view(np.uint32), g["alias"][0][a:b].view(np.uint32))
                    and np.array_equal(alias[s:s + b - a], g["alias"][1][a:b])):
                alias_ok = False
                break
    return rows_ok, alias_ok, int(g["col"].shape[0])

"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libglref.so =
the reference's own sampler / aggregator / storage sources, shim-compiled by
oracle/Makefile).  Run in a container that has /root/reference:

    python tests/golden/make_golden.py

The GPU box has no /root/reference; the committed .npz files are what travels.
Everything here is deterministic (fixed numpy seeds, pinned mt19937 seed).
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_bindings import RefLib, VP, SAMPLERS, AGGREGATORS  # noqa: E402

OUT_DIR = HERE  # --check regenerates into a scratch directory instead


def ref_alias(ref, w):
    ref.L.glref_alias_build.argtypes = [VP, ctypes.c_int32, VP, VP]
    p = np.zeros(w.shape[0], np.float32)
    a = np.zeros(w.shape[0], np.int32)
    ref.L.glref_alias_build(w.ctypes.data_as(VP), w.shape[0], p.ctypes.data_as(VP), a.ctypes.data_as(VP))
    return p, a


def first_appearance(ids):
    """Row order the reference's AutoIndex assigns (auto_indexing.cc:21-24)."""
    _, idx = np.unique(ids, return_index=True)
    return ids[np.sort(idx)]


def gen_kat(ref):
    """sampler_unittest.cpp:76-83 graph; Topk KAT is {20,10,21,11} (:190-195)."""
    src = np.array([0, 0, 0, 1, 1], np.int64)
    dst = np.array([10, 20, 30, 11, 21], np.int64)
    w = np.array([0.8, 1.0, 0.5, 0.88, 1.2], np.float32)
    ref.add_edges("kat", src, dst, w)
    rows = first_appearance(src)
    rp, col, eid, ws = ref.export_csr("kat", rows, 8)
    out = dict(src=src, dst=dst, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws)
    q = np.array([0, 1, 2], np.int64)
    out["query"] = q
    for pad in (0, 1):
        for dflt in (0, -1):
            for k in (2, 4):
                ref.set_flags(pad, dflt, 0.0)
                n, e = ref.sample("kat", "TopkSampler", q, k)
                out["topk_p%d_d%d_k%d_nbr" % (pad, dflt + 1, k)] = n
                out["topk_p%d_d%d_k%d_eid" % (pad, dflt + 1, k)] = e
    np.savez_compressed(os.path.join(OUT_DIR, "kat_sampler.npz"), **out)


def gen_pyfixture(ref):
    """GL/python/tests/utils.py:96,126-133: src 100..199, dst = src*it % 100 for
    it in 1..src%5, weight = float('%f' % ((src + 0.1*dst)/10)); topk expectations
    are pinned by test_topk_neighbor_sampling.py:31-66 / utils.py:304-326."""
    src, dst, w = [], [], []
    for s in range(100, 200):
        for it in range(1, s % 5 + 1):
            d = s * it % 100
            src.append(s)
            dst.append(d)
            w.append(float("%f" % ((s + 0.1 * d) / 10.0)))
    src = np.array(src, np.int64)
    dst = np.array(dst, np.int64)
    w = np.array(w, np.float32)
    ref.add_edges("edge2", src, dst, w)
    rows = first_appearance(src)
    rp, col, eid, ws = ref.export_csr("edge2", rows, 8)
    out = dict(src=src, dst=dst, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws)
    q = np.array([102, 107, 108, 105, 110, 5], np.int64)
    out["query"] = q
    for pad in (0, 1):
        ref.set_flags(pad, -1, 0.0)
        n, e = ref.sample("edge2", "TopkSampler", q, 6)
        out["topk_p%d_nbr" % pad] = n
        out["topk_p%d_eid" % pad] = e
        # replicate mode makes RWoR deterministic too (ReplicatePadder ignores indices)
        if pad == 0:
            n, e = ref.sample("edge2", "RandomWithoutReplacementSampler", q, 6)
            out["rwor_p0_nbr"] = n
            out["rwor_p0_eid"] = e
    np.savez_compressed(os.path.join(OUT_DIR, "pyfixture_topk.npz"), **out)


def gen_rand_graph(ref):
    """Random weighted multigraph with sparse, partly negative raw ids."""
    rng = np.random.default_rng(11)
    V, E = 300, 6000
    raw = (np.arange(V, dtype=np.int64) * 7 + 1000)
    raw[::13] *= -1
    s_idx = np.minimum((rng.pareto(1.2, E) * 6).astype(np.int64), V - 1)
    src = raw[s_idx]
    dst = raw[rng.integers(0, V, E)]
    w = (rng.random(E) * 0.99 + 0.01).astype(np.float32)
    w = (w + np.arange(E, dtype=np.float32) * np.float32(2.0 ** -20)).astype(np.float32)  # tie-free
    ref.add_edges("rnd", src, dst, w)
    rows = first_appearance(src)
    maxdeg = int(np.bincount(s_idx).max())
    rp, col, eid, ws = ref.export_csr("rnd", rows, maxdeg)
    prob = np.zeros(E, np.float32)
    alias = np.zeros(E, np.int32)
    for r in range(rows.shape[0]):
        a, b = rp[r], rp[r + 1]
        prob[a:b], alias[a:b] = ref_alias(ref, ws[a:b])
    out = dict(src=src, dst=dst, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws,
               alias_prob=prob, alias_idx=alias)
    q = np.concatenate([rows[:40], np.array([5, -5, 123456789], np.int64), rows[-10:]])
    out["query"] = q
    for pad in (0, 1):
        for k in (1, 3, 10, 33, 70):
            ref.set_flags(pad, -7, 0.0)
            n, e = ref.sample("rnd", "TopkSampler", q, k)
            out["topk_p%d_k%d_nbr" % (pad, k)] = n
            out["topk_p%d_k%d_eid" % (pad, k)] = e
    ref.set_flags(0, -7, 0.0)
    for k in (3, 33):
        n, e = ref.sample("rnd", "RandomWithoutReplacementSampler", q, k)
        out["rwor_p0_k%d_nbr" % k] = n
        out["rwor_p0_k%d_eid" % k] = e
    # FullSampler (sparse response) and the in-degree weights / alias tables InDegreeSampler uses
    ref.set_flags(1, -7, 0.0)
    for lim in (0, 3, 33):
        d, n, e = ref.sample_full("rnd", q, lim)
        out["full_l%d_deg" % lim] = d
        out["full_l%d_nbr" % lim] = n
        out["full_l%d_eid" % lim] = e
    indeg = ref.in_degree("rnd", col).astype(np.float32)
    ip = np.zeros(E, np.float32)
    ia = np.zeros(E, np.int32)
    for r in range(rows.shape[0]):
        a, b = rp[r], rp[r + 1]
        ip[a:b], ia[a:b] = ref_alias(ref, indeg[a:b].copy())
    out["indeg_w"] = indeg
    out["indeg_alias_prob"] = ip
    out["indeg_alias_idx"] = ia
    np.savez_compressed(os.path.join(OUT_DIR, "rand_graph.npz"), **out)


def gen_dist(ref):
    """Reference sampling distributions (the reference is unseedable, so random
    samplers are pinned distributionally): per (row, slot) histograms of the
    picked row-local position over T independent requests."""
    rng = np.random.default_rng(5)
    degs = [1, 2, 3, 5, 8, 20]
    src, dst, w = [], [], []
    for r, d in enumerate(degs):
        for j in range(d):
            src.append(r)
            dst.append(1000 * (r + 1) + j)
            w.append(rng.random() * 0.99 + 0.01)
    src = np.array(src, np.int64)
    dst = np.array(dst, np.int64)
    w = np.array(w, np.float32)
    ref.add_edges("dist", src, dst, w)
    rows = np.arange(len(degs), dtype=np.int64)
    rp, col, eid, ws = ref.export_csr("dist", rows, max(degs))
    out = dict(src=src, dst=dst, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws,
               degs=np.array(degs, np.int64))
    T = 40000
    ref.set_flags(1, 0, 0.0)
    ref.set_seed(12345)
    for name in SAMPLERS[:3]:
        for k in (2, 6):
            q = np.tile(rows, T)
            _, e = ref.sample("dist", name, q, k, fresh_thread=True)
            e = e.reshape(T, len(degs), k)
            hist = np.zeros((len(degs), k, max(degs)), np.int64)
            pair = np.zeros((len(degs), max(degs), max(degs)), np.int64)  # joint of slots 0,1
            for r, d in enumerate(degs):
                pos_of = {int(x): i for i, x in enumerate(eid[rp[r]:rp[r + 1]])}
                pos = np.vectorize(pos_of.get)(e[:, r, :])
                for j in range(k):
                    hist[r, j, :d] = np.bincount(pos[:, j], minlength=d)
                np.add.at(pair[r], (pos[:, 0], pos[:, 1]), 1)
            out["%s_k%d_hist" % (name, k)] = hist
            out["%s_k%d_pair" % (name, k)] = pair
    out["T"] = np.array(T)
    np.savez_compressed(os.path.join(OUT_DIR, "dist.npz"), **out)


def gen_dist_indegree(ref):
    """InDegreeSampler distribution: neighbours with repeated destination ids so that
    in-degrees differ (weights = in-degree of the neighbour, in_degree_sampler.cc:79-92)."""
    rng = np.random.default_rng(6)
    degs = [1, 2, 4, 7, 12]
    pool = np.arange(500, 512, dtype=np.int64)
    src, dst = [], []
    for r, d in enumerate(degs):
        src += [r] * d
        dst += rng.choice(pool, d, replace=False).tolist()
    # extra edges from other sources to skew the in-degrees of the pool
    extra = rng.choice(pool, 60, p=np.arange(1, 13) / 78.0)
    src += list(range(100, 160))
    dst += extra.tolist()
    src = np.array(src, np.int64)
    dst = np.array(dst, np.int64)
    ref.add_edges("indeg", src, dst)
    rows = first_appearance(src)  # ALL rows: in-degrees count every edge of the type
    rp, col, eid, _ = ref.export_csr("indeg", rows, max(degs))
    out = dict(src=src, dst=dst, rows=rows, row_ptr=rp, col=col, eid=eid, degs=np.array(degs, np.int64),
               indeg_w=ref.in_degree("indeg", col).astype(np.float32))
    T = 40000
    ref.set_flags(1, 0, 0.0)
    ref.set_seed(777)
    k = 4
    _, e = ref.sample("indeg", "InDegreeSampler", np.tile(rows[:len(degs)], T), k, fresh_thread=True)
    e = e.reshape(T, len(degs), k)
    hist = np.zeros((len(degs), k, max(degs)), np.int64)
    for r, d in enumerate(degs):
        pos_of = {int(x): i for i, x in enumerate(eid[rp[r]:rp[r + 1]])}
        pos = np.vectorize(pos_of.get)(e[:, r, :])
        for j in range(k):
            hist[r, j, :d] = np.bincount(pos[:, j], minlength=d)
    out["hist"] = hist
    out["T"] = np.array(T)
    np.savez_compressed(os.path.join(OUT_DIR, "dist_indegree.npz"), **out)


def gen_agg(ref):
    """aggregating_op_unittest.cpp:216-364 KAT (100 nodes, attr = id, segment
    sizes {0,1,2,3,4}) + random cases with unknown ids, empty segments and a
    cursor-stalling (unsorted) tail."""
    out = {}
    ids = np.arange(100, dtype=np.int64)
    feats = np.arange(100, dtype=np.float32).reshape(100, 1).copy()
    ref.set_flags(1, 0, 0.0)
    ref.add_nodes("kat_user", ids, feats)
    nid = np.arange(10, dtype=np.int64)
    seg = np.array([1, 2, 2, 3, 3, 3, 4, 4, 4, 4], np.int32)
    out["kat_ids"] = nid
    out["kat_seg"] = seg
    for name in AGGREGATORS:
        emb, cnt = ref.aggregate("kat_user", name, nid, seg, 5, 1)
        out["kat_%s_emb" % name] = emb
        out["kat_%s_cnt" % name] = cnt
    rng = np.random.default_rng(21)
    case = 0
    for D in (1, 3, 8, 128, 260):
        for dflt in (0.0, 999.9):
            V = 200
            raw = np.arange(V, dtype=np.int64) * 5 - 300
            X = (rng.standard_normal((V, D)) * 10).astype(np.float32)
            ntype = "n%d" % case
            ref.set_flags(1, 0, dflt)
            ref.add_nodes(ntype, raw, X)
            Sg = 37
            sizes = rng.integers(0, 9, Sg)
            sizes[[0, 5, 6, Sg - 1]] = 0
            sizes[10] = 40
            seg = np.repeat(np.arange(Sg, dtype=np.int32), sizes)
            nid = raw[rng.integers(0, V, seg.shape[0])].copy()
            nid[rng.random(seg.shape[0]) < 0.1] = 7777777  # unknown ids -> default row
            if case % 2 == 1:  # stall the cursor: an out-of-order tail is never consumed
                seg = np.concatenate([seg, np.array([3, 4, 50, -1], np.int32)])
                nid = np.concatenate([nid, raw[:4]])
            out["c%d_raw" % case] = raw
            out["c%d_X" % case] = X
            out["c%d_ids" % case] = nid
            out["c%d_seg" % case] = seg
            out["c%d_default" % case] = np.float32(dflt)
            out["c%d_num_segments" % case] = np.array(Sg)
            for name in AGGREGATORS:
                emb, cnt = ref.aggregate(ntype, name, nid, seg, Sg, D)
                out["c%d_%s_emb" % (case, name)] = emb
                out["c%d_%s_cnt" % (case, name)] = cnt
            case += 1
    out["num_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT_DIR, "agg.npz"), **out)


def gen_agg_stitch(ref):
    """The reference's distributed aggregation end to end: node table split over 3
    "servers" by llabs(id) % 3, every server aggregates its part of the request
    (AggregatingRequest::Partition keeps the segment ids), AggregatingResponse::Stitch
    folds the partial responses.  Stored next to the single-server answer."""
    out = {}
    rng = np.random.default_rng(33)
    P, V = 3, 240
    case = 0
    for D in (4, 7, 64):
        for dflt in (0.0, 2.5):
            raw = np.arange(V, dtype=np.int64) * 7 - 500
            X = (rng.standard_normal((V, D)) * 10).astype(np.float32)
            ref.set_flags(1, 0, dflt)
            ref.add_nodes("st%d_all" % case, raw, X)
            owner = np.abs(raw) % P
            for p in range(P):
                ref.add_nodes("st%d_p%d" % (case, p), raw[owner == p], X[owner == p])
            Sg = 41
            sizes = rng.integers(0, 10, Sg)
            sizes[[0, 9, Sg - 1]] = 0
            seg = np.repeat(np.arange(Sg, dtype=np.int32), sizes)
            nid = raw[rng.integers(0, V, seg.shape[0])].copy()
            nid[rng.random(seg.shape[0]) < 0.05] = 123456789  # unknown id: its owner contributes a default row
            own = np.abs(nid) % P
            out["c%d_X" % case] = X
            out["c%d_raw" % case] = raw
            out["c%d_ids" % case] = nid
            out["c%d_seg" % case] = seg
            out["c%d_default" % case] = np.float32(dflt)
            out["c%d_num_segments" % case] = np.array(Sg)
            for name in AGGREGATORS:
                parts = np.zeros((P, Sg, D), np.float32)
                cnts = np.zeros((P, Sg), np.int32)
                for p in range(P):
                    parts[p], cnts[p] = ref.aggregate("st%d_p%d" % (case, p), name, nid[own == p], seg[own == p], Sg, D)
                emb, cnt = ref.aggregate_stitch(name, parts, cnts)
                one, one_cnt = ref.aggregate("st%d_all" % case, name, nid, seg, Sg, D)
                assert np.array_equal(cnt, one_cnt)
                out["c%d_%s_parts" % (case, name)] = parts
                out["c%d_%s_cnts" % (case, name)] = cnts
                out["c%d_%s_stitched" % (case, name)] = emb
                out["c%d_%s_single" % (case, name)] = one
                out["c%d_%s_cnt" % (case, name)] = cnt
            case += 1
    out["num_cases"] = np.array(case)
    np.savez_compressed(os.path.join(OUT_DIR, "agg_stitch.npz"), **out)


def gen_loader(ref):
    """Loader primitives of the reference: Hash64
    (common/base/hash.cc) on byte strings of every tail length, and ParseAttribute
    (core/io/parser.cc:39-104) on well-formed and malformed attribute columns."""
    import json
    rng = np.random.default_rng(55)
    hashes = []
    samples = [b"", b"hehe", b"0", b"12345678", b"123456789", "\u00e9t\u00e9".encode()]
    for n in range(0, 41):
        samples.append(bytes(rng.integers(0, 256, n, dtype=np.uint8).tolist()))
    for b in samples:
        hashes.append([b.hex(), str(ref.hash64(b))])
    I32, I64, F, D, S = 0, 1, 2, 3, 4
    cases = []
    for data, delim, types, buckets in [
        (b"3:2.5:7:hehe", ":", [I64, F, S, S], [0, 0, 0, 10]),   # python/tests/utils.py ATTR_TYPES
        (b"3:2.5:7:hehe", ":", [I64, F, S, S], None),
        (b"3:x:7:hehe", ":", [I64, F, S, S], [0, 0, 0, 10]),      # bad float
        (b"3:2.5:7", ":", [I64, F, S, S], [0, 0, 0, 10]),          # too few
        (b"3:2.5:7:a:b", ":", [I64, F, S, S], [0, 0, 0, 10]),      # too many
        (b"", ":", [], None),
        (b"", ":", [S], None),                                      # empty input = zero tokens
        (b"3::", ":", [I64, S, S], None),                           # empty tokens are kept
        (b"1,2;3", ",;", [I64, I64, I64], None),                    # the delimiter is a SET of characters
        (b" 12 :  1e3 :-7", ":", [I64, F, I32], None),              # blanks around numbers
        (b"12a", ":", [I64], None),
        (b"99999999999", ":", [I32], None),                          # int32 overflow
        (b"99999999999", ":", [I64], None),
        (b"0.1:1.5e-3", ":", [D, D], None),
        (b"7:abc", ":", [S, S], [5, 0]),                             # hash the first, keep the second
        (b"-0.0:nan:inf", ":", [F, F, F], None),
    ]:
        rc, ints, floats, strings = ref.parse_attribute(data, delim, types, buckets)
        cases.append(dict(data=data.hex(), delimiter=delim, types=types, hash_buckets=buckets, code=int(rc),
                          ints=[int(x) for x in ints], floats_bits=[int(x) for x in floats.view(np.uint32)],
                          strings=[x.hex() for x in strings]))
    with open(os.path.join(OUT_DIR, "loader.json"), "w") as f:
        json.dump(dict(hash64=hashes, parse_attribute=cases), f, indent=1)


def gen_negative(ref):
    """Candidate lists of the negative samplers as the reference holds them
    (GetAllDstIds / GetAllInDegrees, topo_statics.cc:32-55) and its global alias tables
    (AliasMethod over the in-degrees / node weights), for a graph with repeated and
    late-appearing destinations."""
    out = {}
    rng = np.random.default_rng(77)
    E = 3000
    src = rng.integers(0, 150, E).astype(np.int64) * 2 + 1
    dst = (rng.zipf(1.4, E) % 300).astype(np.int64) * 5 - 200
    w = (rng.random(E) * 0.99 + 0.01 + np.arange(E) * 1e-7).astype(np.float32)
    ref.add_edges("neg", src, dst, w)
    rows = first_appearance(src)
    rp, col, eid, ws = ref.export_csr("neg", rows, 4000)
    ids, deg = ref.dst_statics("neg")
    prob, alias = ref.alias_build(deg.astype(np.float32))
    out.update(src=src, dst=dst, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws, dst_ids=ids, in_degrees=deg,
               indeg_prob=prob, indeg_alias=alias)
    nid = np.arange(400, dtype=np.int64) * 3 - 50
    nw = (rng.random(400) + 0.02).astype(np.float32)
    np_, na = ref.alias_build(nw)
    out.update(node_ids=nid, node_weights=nw, node_prob=np_, node_alias=na)
    np.savez_compressed(os.path.join(OUT_DIR, "negative.npz"), **out)


def gen_timestamped(ref):
    """A timestamped (and weighted) edge type: after Build() every row is in timestamp-ascending order,
    not weight order (memory_adj_matrix.cc:60-66,129-148).  Timestamps are distinct inside a row."""
    out = {}
    rng = np.random.default_rng(91)
    E = 2500
    src = rng.integers(0, 120, E).astype(np.int64) * 3
    dst = rng.integers(0, 500, E).astype(np.int64)
    ts = rng.permutation(E).astype(np.int64) * 7 + 1_600_000_000
    w = (rng.random(E) + 0.01).astype(np.float32)
    ref.add_edges_timestamped("ts", src, dst, ts, w)
    rows = first_appearance(src)
    rp, col, eid, ws = ref.export_csr("ts", rows, 4000)
    tk, te = ref.sample("ts", "TopkSampler", rows, 4)
    out.update(src=src, dst=dst, ts=ts, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws, topk_nbr=tk, topk_eid=te)
    np.savez_compressed(os.path.join(OUT_DIR, "timestamped.npz"), **out)


def gen_filtered(ref):
    """Sampling with a Filter (core/operator/sampler/filter.{h,cc}).
    Deterministic part: TopkSampler / FullSampler answers of the reference for ID and TIMESTAMP
    filters of both types under both padding modes, on a timestamped multigraph whose rows have
    1..40 neighbours with repeated destination ids.  Random part: per (row, slot) histograms
    of RandomSampler / RandomWithoutReplacementSampler / EdgeWeightSampler / InDegreeSampler
    with an ID == value filter."""
    from oracle_bindings import FILTER_EQUAL, FILTER_LARGER_THAN, FIELD_ID, FIELD_TIMESTAMP, Oracle
    orc = Oracle()
    out = {}
    rng = np.random.default_rng(123)
    degs = np.concatenate([[1, 1, 2, 2, 3], rng.integers(1, 41, 55)])
    src = np.repeat(np.arange(degs.shape[0], dtype=np.int64) * 5 - 20, degs)
    dst = rng.integers(0, 12, src.shape[0]).astype(np.int64) + 900
    ts = rng.permutation(src.shape[0]).astype(np.int64) * 3 + 1000
    w = (rng.random(src.shape[0]) + 0.05).astype(np.float32)
    ref.add_edges_timestamped("flt", src, dst, ts, w)
    rows = first_appearance(src)
    rp, col, eid, ws = ref.export_csr("flt", rows, 64)
    ts_slot = ts[eid]
    out.update(src=src, dst=dst, ts=ts, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws, ts_slot=ts_slot)
    q = np.concatenate([rows, [77777, rows[3]]]).astype(np.int64)  # + an unknown id and a repeat
    pos = {int(v): i for i, v in enumerate(rows)}

    def row_of(v):
        i = pos.get(int(v))
        return (col[rp[i]:rp[i + 1]], ts_slot[rp[i]:rp[i + 1]]) if i is not None else (np.zeros(0, np.int64),) * 2

    def values_for(kind):
        vals = []
        for n, v in enumerate(q):
            nb, t = row_of(v)
            if nb.shape[0] == 0:
                vals.append(5)
            elif kind == "id_eq":
                vals.append(int(nb[n % nb.shape[0]]) if n % 3 else 899)
            elif kind == "id_gt":
                vals.append(int(np.sort(nb)[nb.shape[0] // 2]))
            elif kind == "ts_eq":
                vals.append(int(t[n % t.shape[0]]))
            else:  # ts_gt: per-row thresholds; the ActOn path only ever reads values[0]
                vals.append(int(np.sort(t)[t.shape[0] // 2]) + (n % 2))
        return np.array(vals, np.int64)

    filters = {"id_eq": (FILTER_EQUAL, FIELD_ID), "id_gt": (FILTER_LARGER_THAN, FIELD_ID),
               "ts_eq": (FILTER_EQUAL, FIELD_TIMESTAMP), "ts_gt": (FILTER_LARGER_THAN, FIELD_TIMESTAMP)}
    cases = []
    for kind, (ft, ff) in filters.items():
        vals = values_for(kind)
        if kind == "ts_gt":
            vals[0] = int(np.median(ts))  # the shared threshold: about half of every row survives
        out["values_" + kind] = vals
        flt = dict(type=ft, field=ff, values=vals)
        for pad in (1, 0):
            ref.set_flags(pad, -7, 0.0)
            for strategy, k in (("TopkSampler", 6), ("TopkSampler", 1), ("FullSampler", 0), ("FullSampler", 3)):
                ids = q
                if strategy == "FullSampler" and pad == 1:
                    # a row filtered to nothing breaks the reference's ragged layout (FillWith(dim2)): leave those out
                    keep = []
                    for n, v in enumerate(q):
                        nb, t = row_of(v)
                        one = dict(flt, values=vals if kind == "ts_gt" else vals[n:n + 1])
                        keep.append(nb.shape[0] == 0 or orc.filter_act_on(one, 0, nb, t).shape[0] > 0)
                    ids = q[np.array(keep)]
                    flt_case = dict(flt, values=vals[np.array(keep)] if kind != "ts_gt" else np.concatenate(
                        [vals[:1], vals[np.array(keep)][1:]]))
                else:
                    flt_case = flt
                got = ref.sample_filtered("flt", strategy, ids, k, flt_case)
                name = "%s_%s_k%d_pad%d" % (kind, strategy, k, pad)
                out[name + "_ids"] = ids
                out[name + "_values"] = np.ascontiguousarray(flt_case["values"], np.int64)
                if strategy == "FullSampler":
                    out[name + "_deg"], out[name + "_nbr"], out[name + "_eid"] = got
                else:
                    out[name + "_nbr"], out[name + "_eid"] = got
                cases.append(name)
    # Filter::FillValues: values shorter than the batch repeat batch / len(values) times
    ref.set_flags(1, -7, 0.0)
    two = np.array([rows[10], rows[11], rows[12], rows[13]], np.int64)
    short = np.array([int(row_of(rows[10])[0][0]), int(row_of(rows[12])[0][0])], np.int64)
    out["fill_ids"], out["fill_values"] = two, short
    out["fill_nbr"], out["fill_eid"] = ref.sample_filtered("flt", "TopkSampler", two, 5,
                                                          dict(type=FILTER_EQUAL, field=FIELD_ID, values=short))
    out["cases"] = np.array(cases)

    # ---- distributions --------------------------------------------------------------
    rng = np.random.default_rng(124)
    ddegs = [2, 3, 5, 8, 12]
    pool = np.arange(700, 716, dtype=np.int64)
    dsrc, ddst = [], []
    for r, d in enumerate(ddegs):
        dsrc += [r] * d
        ddst += rng.choice(pool, d, replace=False).tolist()
    extra = rng.choice(pool, 80, p=np.arange(1, 17) / 136.0)
    dsrc += list(range(100, 180))
    ddst += extra.tolist()
    dsrc, ddst = np.array(dsrc, np.int64), np.array(ddst, np.int64)
    dw = (rng.random(dsrc.shape[0]) * 0.95 + 0.05).astype(np.float32)
    ref.add_edges("fdist", dsrc, ddst, dw)
    drows = first_appearance(dsrc)
    drp, dcol, deid, dws = ref.export_csr("fdist", drows, max(ddegs))
    out.update(d_src=dsrc, d_dst=ddst, d_w=dw, d_rows=drows, d_row_ptr=drp, d_col=dcol, d_eid=deid, d_w_slot=dws,
               d_degs=np.array(ddegs, np.int64), d_indeg_w=ref.in_degree("fdist", dcol).astype(np.float32))
    T, k = 20000, 4
    dvals = np.array([int(dcol[drp[r] + (r % d)]) for r, d in enumerate(ddegs)], np.int64)  # one neighbour of each row
    out["d_values"] = dvals
    flt = dict(type=FILTER_EQUAL, field=FIELD_ID, values=np.tile(dvals, T), retry_times=1)
    ref.set_flags(1, 0, 0.0)
    ref.set_seed(4242)
    for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "InDegreeSampler"):
        _, e = ref.sample_filtered("fdist", name, np.tile(drows[:len(ddegs)], T), k, flt, fresh_thread=True)
        e = e.reshape(T, len(ddegs), k)
        hist = np.zeros((len(ddegs), k, max(ddegs)), np.int64)
        for r, d in enumerate(ddegs):
            pos_of = {int(x): i for i, x in enumerate(deid[drp[r]:drp[r + 1]])}
            p = np.vectorize(pos_of.get)(e[:, r, :])
            for j in range(k):
                hist[r, j, :d] = np.bincount(p[:, j], minlength=d)
        out["d_%s_hist" % name] = hist
    out["d_T"] = np.array(T)
    np.savez_compressed(os.path.join(OUT_DIR, "filtered.npz"), **out)


def gen_walk(ref):
    """The reference's RandomWalk operator (random_walk.cc) on a weighted graph without dead ends: node2vec
    walks of length 3 from one seed, 40000 walkers per (p, q): histograms of the first step and of the
    (step 1, step 2) and (step 2, step 3) pairs.  Also a run with DefaultFullNbrNum = 3."""
    rng = np.random.default_rng(77)
    V = 14
    src, dst = [], []
    for v in range(V):
        for d in rng.choice(V, int(rng.integers(3, 9)), replace=False):
            src.append(v)
            dst.append(int(d))
    src, dst = np.array(src, np.int64), np.array(dst, np.int64)
    w = (rng.random(src.shape[0]) + 0.05).astype(np.float32)
    ref.add_edges("walk", src, dst, w)
    rows = first_appearance(src)
    rp, col, eid, ws = ref.export_csr("walk", rows, 16)
    out = dict(src=src, dst=dst, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws)
    T = 40000
    ref.set_flags(1, 0, 0.0)
    ref.set_seed(2024)
    cases = []
    for p, q, F in ((0.5, 2.0, 100), (4.0, 0.25, 100), (0.5, 2.0, 3), (1.0, 1.0, 100)):
        walks = ref.random_walk("walk", np.full(T, 5, np.int64), 3, p, q, F)
        name = "p%g_q%g_F%d" % (p, q, F)
        h1 = np.bincount(walks[:, 0], minlength=V)
        h12 = np.zeros((V, V), np.int64)
        h23 = np.zeros((V, V), np.int64)
        np.add.at(h12, (walks[:, 0], walks[:, 1]), 1)
        np.add.at(h23, (walks[:, 1], walks[:, 2]), 1)
        out[name + "_h1"], out[name + "_h12"], out[name + "_h23"] = h1, h12, h23
        cases.append(name)
    out["cases"] = np.array(cases)
    out["T"] = np.array(T)
    np.savez_compressed(os.path.join(OUT_DIR, "walk.npz"), **out)


def gen_subgraph(ref):
    """The reference's SubGraphSampler (core/operator/subgraph/subgraph_sampler.{h,cc}) on a small weighted graph with
    multi-edges (the later slot's edge id wins, :60-64) and duplicate seeds.  One-hop and zero-hop requests only: with
    two or more hops the reference feeds hop h+1 from the already destroyed response of hop h (subgraph_sampler.h:52-68:
    `nodes = res.GetNeighborIds()` outlives `res`), i.e. its output is undefined there."""
    rng = np.random.default_rng(77)
    V = 60
    src, dst = [], []
    for v in range(V):
        for d in rng.choice(V, int(rng.integers(0, 12)), replace=True):  # replace=True: multi-edges
            src.append(v)
            dst.append(int(d))
    src, dst = np.array(src, np.int64), np.array(dst, np.int64)
    w = (rng.random(src.shape[0]) + 0.05).astype(np.float32)
    ref.add_edges("sub", src, dst, w)
    rows = first_appearance(src)
    rp, col, eid, ws = ref.export_csr("sub", rows, 16)
    out = dict(src=src, dst=dst, w=w, rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws)
    ref.set_flags(1, 0, 0.0)
    cases = []
    for name, seeds, nn, full, dist in (
            ("hop1", [3, 17, 40, 41], [4], 100, False),
            ("hop1_limit3", [3, 17, 40, 41], [4], 3, False),
            ("hop0", [5, 6, 7, 8, 9, 10], [0], 100, False),
            ("dup_seeds", [3, 3, 17], [2], 100, False),
            ("pair_dist", [3, 17], [5], 100, True),
            ("pair_dist_far", [0, 59], [3], 100, True),
            ("unknown_seed", [3, 1000], [4], 100, False)):
        r = ref.subgraph("sub", np.array(seeds, np.int64), nn, full_nbr_num=full, need_dist=dist)
        out[name + "_seeds"] = np.array(seeds, np.int64)
        out[name + "_num_nbrs"] = np.array(nn, np.int32)
        out[name + "_full"] = np.array(full)
        for k, v in r.items():
            out[name + "_" + k] = v
        cases.append(name)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT_DIR, "subgraph.npz"), **out)


COND_CASES = (  # name, strategy, batch_share, unique
    ("random", "random", False, False), ("random_unique", "random", False, True), ("random_share", "random", True, False),
    ("in_degree", "in_degree", False, False), ("node_weight", "node_weight", False, True))


def cond_fixture():
    """The graph the conditional negative sampling fixtures share: 120 item nodes (int attribute id % 4, float attribute
    (id % 3) / 2, string attribute "A" / "B" by id parity, node weights), 80 users buying 4 items each (the candidates
    of the edge-type strategies are the items somebody bought, in first-appearance order: GetAllDstIds)."""
    rng = np.random.default_rng(2024)
    U = 120
    items = np.arange(100, 100 + U, dtype=np.int64)
    w = (rng.random(U) + 0.1).astype(np.float32)
    src, dst = [], []
    for u in range(80):
        for d in rng.choice(items, 4, replace=False):
            src.append(u)
            dst.append(int(d))
    src, dst = np.array(src, np.int64), np.array(dst, np.int64)
    req_src = np.arange(4, dtype=np.int64)
    req_dst = np.array([dst[4 * u] for u in range(4)], np.int64)
    return dict(items=items, item_w=w, int_attr=(items % 4).astype(np.int64), float_attr=((items % 3) * 0.5).astype(np.float32),
                str_attr=np.array([b"A" if i % 2 == 0 else b"B" for i in items]), src=src, dst=dst, req_src=req_src,
                req_dst=req_dst, count=np.array(8), int_props=np.array([0.5], np.float32),
                float_props=np.array([0.25], np.float32), str_props=np.array([0.25], np.float32))


def gen_cond_negative(ref):
    """The reference's ConditionalNegativeSampler (conditional_negative_sampler.cc, condition_table.cc,
    attribute_nodes_map.h): per case T requests with fresh pinned seeds; counts[row, slot, candidate] of the ids its
    response holds.  Groups are large and exclusions few, so every response is complete (asserted): the reference's dead
    fill loop is not exercised."""
    fx = cond_fixture()
    ref.add_attr_nodes("item", fx["items"], weights=fx["item_w"], int_attrs=fx["int_attr"].reshape(-1, 1),
                       float_attrs=fx["float_attr"].reshape(-1, 1), str_attrs=[[bytes(x)] for x in fx["str_attr"]])
    out = dict(fx)
    T = 3000
    ref.set_flags(1, 0, 0.0)
    count = int(fx["count"])
    for name, strategy, share, unique in COND_CASES:
        etype = "buy_" + name  # the reference caches its condition tables per type name
        if strategy != "node_weight":
            ref.add_edges(etype, fx["src"], fx["dst"], None)
        else:
            etype = "item"
        counts = np.zeros((fx["req_src"].shape[0], count, fx["items"].shape[0]), np.int32)
        skipped = 0
        for t in range(T):
            ref.set_seed(1000 + t)
            ids = ref.cond_neg_sample(etype, strategy, "item", fx["req_src"], fx["req_dst"], count, int_cols=[0],
                                      int_props=fx["int_props"], float_cols=[0], float_props=fx["float_props"], str_cols=[0],
                                      str_props=fx["str_props"], batch_share=share, unique=unique)
            if ids.shape[0] != fx["req_src"].shape[0] * count:
                skipped += 1  # a column came up short: the reference's response is misaligned (its fill loop is dead code)
                continue
            ix = ids.reshape(-1, count) - 100
            for r in range(ix.shape[0]):
                counts[r, np.arange(count), ix[r]] += 1
        assert skipped <= T // 20, (name, skipped)
        out[name + "_counts"] = counts
        out[name + "_trials"] = np.array(T - skipped)
    out["T"] = np.array(T)
    ref.set_seed(0)
    np.savez_compressed(os.path.join(OUT_DIR, "cond_negative.npz"), **out)


def gen_refseq(ref):
    """The reference's random samplers under a PINNED random_device (fixed_rd.h), each request in a thread of its own:
    the draw-for-draw vectors tests/test_oracle_golden.py::test_refseq_golden holds the oracle's reference-entropy mode to
    (tests/test_oracle_refseq.py does the same live, and wider, where oracle/_ref exists)."""
    rng = np.random.default_rng(99)
    degrees = np.array([0, 1, 2, 3, 4, 5, 8, 13, 16, 33, 64, 257, 0, 7, 1, 100, 70000])
    src = np.repeat(np.arange(len(degrees), dtype=np.int64), degrees)
    dst = rng.integers(0, 40, src.shape[0]).astype(np.int64)
    w = (rng.random(src.shape[0]) * 0.9 + 0.05 + np.arange(src.shape[0]) * 2.0 ** -22).astype(np.float32)
    ref.set_flags(1, -3, 0.0)
    ref.add_edges("refseq", src, dst, w)
    rows = np.flatnonzero(degrees > 0).astype(np.int64)
    rp, col, eid, ws = ref.export_csr("refseq", rows, int(degrees.max()) + 1)
    query = np.concatenate([np.arange(len(degrees) + 2, dtype=np.int64), rng.integers(0, len(degrees), 120)])
    out = dict(rows=rows, row_ptr=rp, col=col, eid=eid, w_slot=ws, query=query, seeds=np.array([1, 20240923], np.int64),
               ks=np.array([1, 4, 9], np.int64))
    for name in ("RandomSampler", "RandomWithoutReplacementSampler", "EdgeWeightSampler", "InDegreeSampler"):
        for seed in out["seeds"]:
            for k in out["ks"]:
                ref.set_seed(int(seed))
                n, e = ref.sample("refseq", name, query, int(k), fresh_thread=True)
                out["%s_s%d_k%d_nbr" % (name, seed, k)] = n
                out["%s_s%d_k%d_eid" % (name, seed, k)] = e
    ref.set_seed(0)
    ref.set_flags(1, 0, 0.0)
    np.savez_compressed(os.path.join(OUT_DIR, "refseq.npz"), **out)


def generate():
    ref = RefLib(storage_mode=2)
    gen_kat(ref)
    gen_pyfixture(ref)
    gen_rand_graph(ref)
    gen_dist(ref)
    gen_dist_indegree(ref)
    gen_agg(ref)
    gen_agg_stitch(ref)
    gen_loader(ref)
    gen_negative(ref)
    gen_timestamped(ref)
    gen_filtered(ref)
    gen_walk(ref)
    gen_subgraph(ref)
    gen_cond_negative(ref)
    gen_refseq(ref)
    ref.close()


def check():
    """Regenerate every fixture from the reference library as built NOW and compare
    with the committed files: same keys, dtypes, shapes and bytes for each .npz
    (array contents, not the zip container), same text for loader.json.  Returns the
    list of differences (empty = the committed goldens are what the recipe makes)."""
    import tempfile
    global OUT_DIR
    diffs = []
    with tempfile.TemporaryDirectory() as tmp:
        OUT_DIR = tmp
        try:
            generate()
        finally:
            OUT_DIR = HERE
        made = sorted(f for f in os.listdir(tmp))
        have = sorted(f for f in os.listdir(HERE) if f.endswith(".npz") or f.endswith(".json"))
        if made != have:
            diffs.append("file sets differ: made %s, committed %s" % (made, have))
        for f in made:
            a, b = os.path.join(tmp, f), os.path.join(HERE, f)
            if not os.path.exists(b):
                continue
            if f.endswith(".json"):
                if open(a).read() != open(b).read():
                    diffs.append(f + ": text differs")
                continue
            za, zb = np.load(a), np.load(b)
            if sorted(za.files) != sorted(zb.files):
                diffs.append("%s: keys differ (%s)" % (f, sorted(set(za.files) ^ set(zb.files))))
                continue
            for k in za.files:
                x, y = za[k], zb[k]
                if x.dtype != y.dtype or x.shape != y.shape or x.tobytes() != y.tobytes():
                    diffs.append("%s[%s]" % (f, k))
    return diffs


def main():
    if "--check" in sys.argv[1:]:
        diffs = check()
        for d in diffs:
            print("DIFF", d)
        print("golden check:", "FAILED (%d)" % len(diffs) if diffs else "OK -- every committed fixture regenerates identically")
        sys.exit(1 if diffs else 0)
    generate()
    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("  %-22s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()

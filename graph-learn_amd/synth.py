"""Synthetic inputs for tests and bench.py (there is no network for Cora / OGB).

RMAT power-law graphs with the parameters SURVEY.md 8(d) fixes
((a,b,c,d) = (0.57,0.19,0.19,0.05), dedup off), tie-free U(0.01,1) weights, rows
sorted by weight descending (the order the reference's Build() leaves:
memory_adj_matrix.cc:105-125), edge ids = insertion indices.  torch is used as
plumbing to generate on the GPU; the small numpy variant serves the CPU tests.
"""
import numpy as np


def rmat_edges_numpy(scale, num_edges, num_nodes, seed, a=0.57, b=0.19, c=0.19):
    rng = np.random.default_rng(seed)
    src = np.zeros(num_edges, np.int64)
    dst = np.zeros(num_edges, np.int64)
    for _ in range(scale):
        r = rng.random(num_edges)
        sb = (r >= a + b).astype(np.int64)
        db = (((r >= a) & (r < a + b)) | (r >= a + b + c)).astype(np.int64)
        src = (src << 1) | sb
        dst = (dst << 1) | db
    return src % num_nodes, dst % num_nodes


def csr_numpy(src, dst, weight, num_nodes):
    """Insertion-order edge ids, rows sorted by weight descending (stable)."""
    E = src.shape[0]
    eid = np.arange(E, dtype=np.int64)
    if weight is not None:
        order = np.lexsort((eid, -weight.astype(np.float64), src))
    else:
        order = np.argsort(src, kind="stable")
    row_ptr = np.zeros(num_nodes + 1, np.int64)
    np.add.at(row_ptr, src + 1, 1)
    row_ptr = np.cumsum(row_ptr)
    return row_ptr, dst[order].copy(), eid[order].copy(), (weight[order].copy() if weight is not None else None)


def small_graph(num_nodes=2000, num_edges=30000, seed=0, weighted=True, hub_degree=0):
    """numpy power-law graph for parity tests; optional extra hub row 0."""
    scale = int(np.ceil(np.log2(max(num_nodes, 2))))
    src, dst = rmat_edges_numpy(scale, num_edges, num_nodes, seed)
    rng = np.random.default_rng(seed + 1)
    if hub_degree:
        src = np.concatenate([src, np.zeros(hub_degree, np.int64)])
        dst = np.concatenate([dst, rng.integers(0, num_nodes, hub_degree)])
    E = src.shape[0]
    w = None
    if weighted:
        w = (rng.random(E) * 0.99 + 0.01).astype(np.float32)
        w = (w + np.arange(E, dtype=np.float32) * np.float32(2.0 ** -26)).astype(np.float32)
    return csr_numpy(src, dst, w, num_nodes)


def rmat_edges_torch(num_nodes, num_edges, seed, device, weighted=True, a=0.57, b=0.19, c=0.19,
                     chunk=1 << 25, scramble=True):
    """RMAT edge list in generation (= insertion) order: src, dst (int64), weight
    (float32 U(0.01, 1) or None).  Edge id = index.  Feed it to glx_graph_build.

    scramble: relabel vertices with a fixed random permutation, as Graph500 does.  Raw
    RMAT ids are bit-skewed (every id bit is 0 with probability a+b = 0.76), so the
    reference's `llabs(id) % P` ownership rule would put 44 % of all edges on shard 0
    of 8; real-world ids (and scrambled ones) spread evenly."""
    import torch
    scale = int(np.ceil(np.log2(max(num_nodes, 2))))
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    srcs, dsts = [], []
    for lo in range(0, num_edges, chunk):
        n = min(chunk, num_edges - lo)
        s = torch.zeros(n, dtype=torch.int64, device=device)
        d = torch.zeros(n, dtype=torch.int64, device=device)
        for _ in range(scale):
            r = torch.rand(n, generator=gen, device=device)
            sb = (r >= a + b).to(torch.int64)
            db = (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)
            s = (s << 1) | sb
            d = (d << 1) | db
        srcs.append(s % num_nodes)
        dsts.append(d % num_nodes)
    src = torch.cat(srcs)
    dst = torch.cat(dsts)
    del srcs, dsts
    weight = None
    if weighted:
        weight = torch.rand(num_edges, generator=gen, device=device) * 0.99 + 0.01
    if scramble:
        perm = torch.randperm(num_nodes, generator=gen, device=device)
        src = perm[src]
        dst = perm[dst]
    return src, dst, weight


def rmat_graph_torch(num_nodes, num_edges, seed, device, weighted=True, a=0.57, b=0.19, c=0.19,
                     chunk=1 << 25):
    """The same RMAT graph as rmat_edges_torch, CSR-sorted with torch (tests use this
    independent construction to cross-check glx_graph_build).

    -> row_ptr[V+1], col[E], eid[E] (int64), weight[E] (float32 or None), all on `device`;
    rows ordered by weight descending, edge id = generation index."""
    import torch
    src, dst, weight = rmat_edges_torch(num_nodes, num_edges, seed, device, weighted, a, b, c, chunk)
    eid = torch.arange(num_edges, dtype=torch.int64, device=device)
    if weighted:
        # sort key: (src asc, weight desc, eid asc) -- two stable sorts
        o = torch.sort(weight, descending=True, stable=True).indices
        src, dst, eid, weight = src[o], dst[o], eid[o], weight[o]
    o = torch.sort(src, stable=True).indices
    src, dst, eid = src[o], dst[o], eid[o]
    if weighted:
        weight = weight[o].contiguous()
    row_ptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(torch.bincount(src, minlength=num_nodes), 0)
    return row_ptr, dst.contiguous(), eid.contiguous(), weight


def features_torch(num_nodes, dim, seed, device):
    """U(-1, 1) float32 features."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    return torch.rand((num_nodes, dim), generator=gen, device=device, dtype=torch.float32) * 2 - 1

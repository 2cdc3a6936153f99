"""SPMD mode of the Python API, run under torch.distributed.run: every rank loads ITS shard of the same TSV
sources (Graph.init(task_index, task_count)), the ranks answer each other's sampling / aggregation requests
through Graph.sharded_store(), and every rank checks its answers against an unsharded load of the same files.
Prints one line `spmd ok rank R` per rank.

  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 scripts/spmd_pyapi_check.py DIR [--share-device]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as td  # noqa: E402
import graphlearn as gl  # noqa: E402

data_dir = sys.argv[1]
share = "--share-device" in sys.argv
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = 0 if share else int(os.environ.get("LOCAL_RANK", "0"))
td.init_process_group(backend="gloo" if share else "nccl")
torch.cuda.set_device(local)
gl.set_device_id(local)
gl.set_padding_mode(gl.CIRCULAR)
gl.set_default_neighbor_id(-1)
gl.set_sampling_seed(77)

edges, nodes = os.path.join(data_dir, "edges"), os.path.join(data_dir, "nodes")
if rank == 0:
    rng = np.random.default_rng(3)
    V, E = 400, 6000
    src = rng.integers(0, V, E) * 3 - 200  # sparse ids, some negative
    dst = rng.integers(0, V, E) * 3 - 200
    with open(edges, "w") as f:
        f.write("src_id:int64\tdst_id:int64\tweight:float\n")
        for s, d, w in zip(src, dst, rng.random(E) + 0.01):
            f.write("%d\t%d\t%.6f\n" % (s, d, w))
    with open(nodes, "w") as f:
        f.write("id:int64\tattribute:string\n")
        for v in range(V):
            f.write("%d\t%s\n" % (v * 3 - 200, ":".join("%.3f" % x for x in rng.standard_normal(8))))
td.barrier()


def build(**kw):
    return gl.Graph().node(nodes, "n", gl.Decoder(attr_types=["float"] * 8)) \
        .edge(edges, ("n", "n", "e"), gl.Decoder(weighted=True)).init(**kw)


whole = build()
whole_graph, whole_feats = whole.device_graph("e"), whole.device_features("n")
shard = build(task_index=rank, task_count=world)
store = shard.sharded_store("e", "n")
dev = torch.device("cuda", local)

# every rank holds its part, and the parts add up
mine = torch.tensor([shard.device_graph("e").num_edges], dtype=torch.int64)
td.all_reduce(mine)
assert int(mine) == whole_graph.num_edges, (int(mine), whole_graph.num_edges)
assert shard.device_graph("e").num_edges < whole_graph.num_edges

gen = np.random.default_rng(100 + rank)  # every rank asks for its own batch
for trial, name in enumerate(["TopkSampler", "EdgeWeightSampler", "RandomSampler", "RandomWithoutReplacementSampler"]):
    ids = torch.from_numpy(np.concatenate([gen.integers(0, 400, 300) * 3 - 200, [5, 10 ** 6]])).to(dev)
    nbr, _ = store.sample(name, ids, 6, seed=77, call_counter=10 + trial, default_neighbor_id=-1)
    want, _ = whole_graph.sample(name, ids, 6, seed=77, call_counter=10 + trial, default_neighbor_id=-1)
    assert torch.equal(nbr, want), (rank, name)
    seg = torch.arange(ids.shape[0], device=dev, dtype=torch.int32).repeat_interleave(6)
    for op in ("MaxAggregator", "SumAggregator"):
        emb, cnt = store.aggregate(op, nbr.reshape(-1), seg, ids.shape[0])
        wemb, wcnt = whole_feats.aggregate(op, nbr.reshape(-1).contiguous(), seg, ids.shape[0])
        assert torch.equal(cnt, wcnt), (rank, name, op)
        # the halo design reduces on the requester in request order: bit-identical for every aggregator
        assert torch.equal(emb.view(torch.int32), wemb.view(torch.int32)), (rank, name, op)

# the sampler of the Python API: in SPMD mode get_device() is the collective multi-hop request
for strategy in ("edge_weight", "topk", "random_without_replacement"):
    ids = torch.from_numpy(gen.integers(0, 400, 200) * 3 - 200).to(dev)
    got = shard.neighbor_sampler(["e", "e"], [4, 3], strategy=strategy).get_device(ids, call_counter=60)
    want = whole.neighbor_sampler(["e", "e"], [4, 3], strategy=strategy).get_device(ids, call_counter=60)
    for (gn, _), (wn, _) in zip(got, want):  # edge ids are per-shard insertion indices, as on the reference's servers
        assert torch.equal(gn.reshape(-1), wn.reshape(-1)), (rank, strategy)

# walks: collective DeepWalk over the shards
ids = torch.from_numpy(gen.integers(0, 400, 150) * 3 - 200).to(dev)
assert torch.equal(shard.random_walk("e", ids, 5, call_counter=80), whole.random_walk("e", ids, 5, call_counter=80)), rank

# node2vec: one partitioned FullSampler request per step + the step on the requester -- the single store's walks
assert torch.equal(shard.random_walk("e", ids, 4, p=0.5, q=2.0, call_counter=84),
                   whole.random_walk("e", ids, 4, p=0.5, q=2.0, call_counter=84)), rank

# in-degrees of destination ids are sums over ALL shards (collective in SPMD mode)
probe = np.concatenate([gen.integers(0, 400, 120) * 3 - 200, [7, 10 ** 7]]).astype(np.int64)
# (the expected counts come from the file: the operators of this process are bound to the LAST store initialised,
# the shard's, so `whole.in_degrees` would count the shard's edges only)
all_dst = np.loadtxt(edges, skiprows=1, usecols=1, dtype=np.int64)
uniq_dst, uniq_cnt = np.unique(all_dst, return_counts=True)
pos = np.searchsorted(uniq_dst, probe)
hit = (pos < uniq_dst.shape[0]) & (uniq_dst[np.minimum(pos, uniq_dst.shape[0] - 1)] == probe)
want_deg = np.where(hit, uniq_cnt[np.minimum(pos, uniq_dst.shape[0] - 1)], 0)
got_deg = shard.in_degrees(probe, "e")
assert np.array_equal(got_deg, want_deg), (rank, probe[got_deg != want_deg][:8], got_deg[got_deg != want_deg][:8])

# the negative samplers draw from the WHOLE type's candidate list: every rank holds the same table (every shard's
# destination ids, ascending, global in-degrees) and answers what one store answers from that table
import glx  # noqa: E402
cand_ids, indeg = uniq_dst, uniq_cnt.astype(np.float32)
ref_tables = {False: glx.Negative(cand_ids, device=local), True: glx.Negative(cand_ids, indeg, device=local)}
whole_graph.enable_negative()
src_ids = np.ascontiguousarray(gen.integers(0, 400, 90) * 3 - 200, dtype=np.int64)
for strategy, by_deg, mode in (("random", False, glx.NEG_EXCLUDE_NONE), ("soft_in_degree", True, glx.NEG_EXCLUDE_NONE),
                               ("in_degree", True, glx.NEG_EXCLUDE_NEIGHBORS)):
    smp = shard.negative_sampler("e", 6, strategy=strategy)
    smp.set_call_counter(300 + rank)
    got = smp.get(src_ids).ids
    want = ref_tables[by_deg].sample(src_ids, 6, exclude=mode, graph=whole_graph, default_neighbor_id=-1, seed=77,
                                     call_counter=300 + rank)
    assert np.array_equal(got, want), (rank, strategy)
t_u = shard.global_negative_table("e")
assert np.array_equal(t_u.export()[0], cand_ids), rank

# a replica of the hottest rows on every GPU changes where rows come from, never the answer
hot_store = shard.sharded_store("e", "n", hot_nodes=50)
emb, cnt = hot_store.aggregate("MeanAggregator", nbr.reshape(-1), seg, ids.shape[0])
wemb, wcnt = whole_feats.aggregate("MeanAggregator", nbr.reshape(-1).contiguous(), seg, ids.shape[0])
assert torch.equal(cnt, wcnt) and torch.equal(emb.view(torch.int32), wemb.view(torch.int32)), rank
st = hot_store.stats()
assert st["from_replica"] > 0 and st["from_replica"] + st["from_own_shard"] + st["remote"] == st["ids"], st
# ... and the same vertices' adjacency rows are on every GPU too: their sampling requests stay local, same draws
ids = torch.from_numpy(gen.integers(0, 400, 300) * 3 - 200).to(dev)
h1, _ = whole_graph.sample("RandomSampler", ids, 5, seed=77, call_counter=70, default_neighbor_id=-1)
for name in ("EdgeWeightSampler", "TopkSampler", "RandomWithoutReplacementSampler"):
    nbr2, _ = hot_store.sample(name, h1.reshape(-1), 4, seed=77, call_counter=71, default_neighbor_id=-1)
    want2, _ = whole_graph.sample(name, h1.reshape(-1), 4, seed=77, call_counter=71, default_neighbor_id=-1)
    assert torch.equal(nbr2, want2), (rank, name)
assert hot_store.native.last_sample_rows()["from_graph_replica"] > 0

# InDegreeSampler weighs neighbours by their in-degree over ALL shards; a shard's tables only count its own
# edges, so a partitioned store refuses it instead of drawing from a different distribution
if world > 1:
    try:
        store.sample("InDegreeSampler", ids, 3)
        raise AssertionError("InDegreeSampler must be refused on a partitioned store")
    except Exception as ex:  # noqa: BLE001
        assert "InDegreeSampler" in str(ex), ex
torch.cuda.synchronize()
td.barrier()
shard.close()
whole.close()
print("spmd ok rank %d" % rank, flush=True)
td.destroy_process_group()

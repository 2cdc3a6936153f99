import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch, torch.distributed as dist, glx, synth
import dist as gdist
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
V, D = int(sys.argv[1]), int(sys.argv[2]); n = int(sys.argv[3]); f = 10
X = synth.features_torch(V, D, 5, dev)
replica = glx.Features(X)
ids = torch.arange(0, V, dtype=torch.int64, device=dev)
store_h = gdist.ShardedStore(gdist.DeviceOps(), None, glx.Features(X, ids=ids))
nid = torch.randint(0, V, (n,), device=dev); seg = (torch.arange(n, device=dev) // f).to(torch.int32)
want, wc = replica.aggregate("MaxAggregator", nid, seg, n // f)
for kw in (dict(mode="halo"), dict(mode="halo", dedup=True), dict(mode="partial")):
    e, c = store_h.aggregate("MaxAggregator", nid, seg, n // f, **kw)
    torch.cuda.synchronize()
    bad = (e != want).any(dim=1)
    print(kw, "bad segments:", int(bad.sum()), "of", n // f, "counts equal:", bool(torch.equal(c, wc)),
          "first bad:", bad.nonzero()[:3].view(-1).tolist(), "last bad:", bad.nonzero()[-3:].view(-1).tolist() if bad.any() else None)
dist.destroy_process_group()

import os, torch, torch.distributed as dist
dev = torch.device("cuda", 0)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for mb in (256, 512, 768, 1023, 1024, 1025, 1536, 2048, 4096):
    n = mb * (1 << 20) // 4
    x = torch.arange(n, device=dev, dtype=torch.int32)
    out = torch.zeros_like(x)
    dist.all_to_all_single(out, x, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    bad = (out != x)
    nb = int(bad.sum())
    first = int(bad.nonzero()[0]) if nb else -1
    g = torch.zeros_like(x)
    dist.all_gather_into_tensor(g, x)
    torch.cuda.synchronize()
    gb = int((g != x).sum())
    x2 = x.view(-1, 64)
    o2 = torch.zeros_like(x2)
    dist.all_to_all_single(o2, x2, output_split_sizes=[x2.shape[0]], input_split_sizes=[x2.shape[0]])
    torch.cuda.synchronize()
    b2 = int((o2 != x2).sum())
    print("%5d MiB: a2a bad %d (first at element %d = byte %d) | all_gather bad %d | a2a 2-D bad %d" % (mb, nb, first, first * 4, gb, b2))
dist.destroy_process_group()

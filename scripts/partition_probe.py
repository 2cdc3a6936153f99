"""Where does glx_partition's time go?  (world-1 edge-cut trace: glx_part_scan_kernel averaged 182 us.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-learn_amd"))
import torch, glx
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
for n in (65536, 1638400, 16384000):
    ids = torch.randint(0, 10_000_000, (n,), generator=gen, device=dev)
    for P in (1, 2, 8):
        for _ in range(3): glx.partition(ids, P)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): glx.partition(ids, P)
        b.record(); torch.cuda.synchronize()
        print("n=%9d P=%d: %.3f ms per partition" % (n, P, a.elapsed_time(b) / 10), flush=True)

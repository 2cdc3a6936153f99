"""What the PCIe link of this box delivers device -> host: copy engine into pinned / pageable memory, and a
kernel writing straight into pinned (mapped) host memory -- the three ways a response can reach the caller."""
import time, torch
dev = torch.device("cuda", 0)
n = 64 << 20  # 256 MiB of float32
x = torch.randn(n, device=dev)
pin = torch.empty(n, dtype=torch.float32).pin_memory()
page = torch.empty(n, dtype=torch.float32)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for name, dst in (("copy engine -> pinned", pin), ("copy engine -> pageable", page)):
    dt = t(lambda: dst.copy_(x, non_blocking=True))
    print("%-28s %.1f GB/s" % (name, n * 4 / dt / 1e9))

"""Achievable HBM bandwidth on this box: device-to-device copy (read + write bytes / time)."""
import torch, time, json
x = torch.empty(1 << 30, dtype=torch.float32, device="cuda")  # 4 GiB
y = torch.empty_like(x)
x.normal_()
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): y.copy_(x)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(json.dumps({"copy_GiB": 4, "ms": ms, "read_plus_write_GBps": 2 * x.numel() * 4 / (ms * 1e-3) / 1e9}))

"""Write / copy bandwidth of plain torch kernels at activation sizes (diagnostic yardstick)."""
import torch
dev = torch.device("cuda")
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for mb in (33.5, 134, 537):
    n = int(mb * 1e6 / 2)
    a = torch.empty(n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, device=dev).bfloat16()
    tf = t(lambda: a.zero_())
    tc = t(lambda: a.copy_(b))
    tr = t(lambda: b.float().sum()) if mb < 200 else 0
    print(f"{mb:6.1f} MB: fill {tf:7.1f} us = {mb/tf*1e0:5.2f} TB/s | copy {tc:7.1f} us = {2*mb/tc:5.2f} TB/s (r+w)")

"""Streaming yardsticks for the first-layer kernels: write-only, read-only and copy bandwidth at the sizes of
the conv1_1 map (33.5 MB at 512^2, 134 MB at 1024^2) and at 1 GB."""
import torch
dev = torch.device("cuda")
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mb in (33.5, 134.2, 1024):
    n = int(mb * 1e6) // 2
    a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.empty_like(a)
    w = t(lambda: a.fill_(1.0)); r = t(lambda: a.sum()); c = t(lambda: b.copy_(a))
    print(f"{mb:7.1f} MB: fill {w:7.1f} us ({mb / w * 1e-3 * 1e3:5.2f} TB/s)   sum {r:7.1f} us ({mb / r:5.2f} TB/s)   copy {c:7.1f} us ({2 * mb / c:5.2f} TB/s r+w)")

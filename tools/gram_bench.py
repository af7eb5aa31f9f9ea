"""Time the Gram partial / finish kernels per tap shape (diagnostic). usage: gram_bench.py [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda")
def t(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
tot_p = tot_f = 0.0
for C, k in ((64, 1), (128, 2), (256, 4), (512, 8), (512, 16)):
    N = (S // k) ** 2
    F = torch.randn(N, C, device=dev).bfloat16()
    T = torch.zeros(C, C, device=dev)
    parts = ops.gram_partial(F)
    lp = torch.empty(ops.gram_loss_parts(C), device=dev)
    sg = torch.empty(C, C, device=dev, dtype=torch.bfloat16)
    tp = t(lambda: ops.gram_partial(F, parts))
    tf = t(lambda: ops.gram_finish(parts, N, C, target=T, loss_part=lp, sgrad=sg, dtype=torch.bfloat16))
    tot_p += tp; tot_f += tf
    print(f"C={C:4d} N={N:8d} ksplit={parts.shape[0]:4d} partial {tp:6.1f} us ({N*C*2/tp/1e6:5.2f} TB/s)  finish {tf:6.1f} us")
print(f"size {S}: partial {tot_p:.1f} us  finish {tot_f:.1f} us")

"""Does sweep B of the L-BFGS step (solve + second history sweep) run faster when part of the history was read just before it
(Infinity Cache)?  flush -> sweep A (stv_lbfgsc_dots) -> [touch k slots of S and Y] -> time stv_lbfgsc_apply.  If it does, a
prefetch kernel beside the one-workgroup solve kernel would be free time (diagnostic).
usage: mall_probe.py [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
n, hist = 3 * size * size, 100
nn = (n + 4095) // 4096 * 4096
state, work = ops.lbfgs_alloc(n, hist, dev, compact=True)
x = torch.randn(n, device=dev) * 0.1
big = torch.empty(256 << 20, device=dev, dtype=torch.float32)
gen = torch.Generator(device=dev).manual_seed(0)
for k in range(hist + 5):                      # fill the history with well-conditioned pairs
    g = x * (1.0 + 0.01 * k) + 0.01 * torch.randn(n, device=dev, generator=gen)
    ops.lbfgs_step(x, g, state, work, hist, min(k, hist), 1.0, compact=True)
torch.cuda.synchronize()
S0 = 2 * nn
def touch(k):
    if k:
        a = work[S0: S0 + k * nn].sum()
        b = work[S0 + (hist + 1) * nn: S0 + (hist + 1) * nn + k * nn].sum()
        return a + b
sink = []
for k in (0, 10, 25, 40, 60, 101):
    ts = []
    for rep in range(8):
        g = x * 1.3 + 0.01 * torch.randn(n, device=dev, generator=gen)
        big.add_(1.0)
        ops.lbfgs_dots(g, state, work, hist, hist)
        sink.append(touch(k))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.lbfgs_apply(x, g, state, work, hist, 1.0)
        e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"size {size}: {k:3d} slots of S and Y read first ({2 * k * nn * 4 / 1e6:6.0f} MB): solve + sweep B {ts[len(ts) // 2]:7.1f} us (min {ts[0]:.1f})", flush=True)

"""Sweep A (+ reduce) and solve + sweep B of the device L-BFGS timed separately at m = history (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
dev = torch.device("cuda")
for size in [int(a) for a in sys.argv[1:]] or [512]:
    n, hist = 3 * size * size, 100
    x = torch.zeros(n, device=dev)
    state, work = ops.lbfgs_alloc(n, hist, dev, compact=True)
    gen = torch.Generator(device=dev).manual_seed(0)
    a = torch.rand(n, device=dev, generator=gen) * 9 + 1
    for k in range(hist + 10):
        g = a * x - 1.0 + 0.01 * torch.randn(n, device=dev, generator=gen)
        ops.lbfgs_step(x, g, state, work, hist, min(k, hist), 1.0, compact=True)
    g = a * x - 1.0
    def timed(fn, reps=30):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    ta = timed(lambda: ops.lbfgs_dots(g, state, work, hist, hist))
    tb = timed(lambda: ops.lbfgs_apply(x, g, state, work, hist, 1.0))
    print(f"size {size}: sweep A + reduce {ta:.1f} us, solve + sweep B {tb:.1f} us   (TILE={os.environ.get('STV_LBFGS_TILE','-')} PG={os.environ.get('STV_LBFGS_PGROUPS','-')} TILE_B={os.environ.get('STV_LBFGS_TILE_B','-')})")

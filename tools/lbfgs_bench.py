"""Steady-state (m = history) time of one device L-BFGS step, isolated (diagnostic).

python tools/lbfgs_bench.py [size ...]   ->  ms per stv_lbfgsc_step at n = 3*size^2, history 100, ring full.
A/B switches are read by the library at load time (e.g. STV_LBFGS_ACC=f32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops

dev = torch.device("cuda")
for size in [int(a) for a in sys.argv[1:]] or [512, 1024]:
    n, hist = 3 * size * size, 100
    x = torch.zeros(n, device=dev)
    state, work = ops.lbfgs_alloc(n, hist, dev, compact=True)
    gen = torch.Generator(device=dev).manual_seed(0)
    a = torch.rand(n, device=dev, generator=gen) * 9 + 1
    for k in range(hist + 10):                     # fill the ring with valid pairs (convex quadratic: y.s > 0)
        g = a * x - 1.0 + 0.01 * torch.randn(n, device=dev, generator=gen)
        ops.lbfgs_step(x, g, state, work, hist, min(k, hist), 1.0, compact=True)
    torch.cuda.synchronize()
    st = state.cpu().view(torch.int32)
    assert int(st[1]) == hist, f"history {int(st[1])}"
    g = a * x - 1.0
    reps = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.lbfgs_step(x, g, state, work, hist, hist, 1.0, compact=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = (4 * hist + 8) * n * 4 / 1e9
    print(f"size {size}: {ms:.4f} ms per L-BFGS step at m={hist}  ({gb / ms:.2f} TB/s of the (4m+8) n 4 B model)  acc={os.environ.get('STV_LBFGS_ACC', 'f64')}")

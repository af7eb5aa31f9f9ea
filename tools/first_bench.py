"""Time the first-layer kernels (VALU vs matrix-core) at a given size (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
SW = int(sys.argv[2]) if len(sys.argv) > 2 else S
dev = torch.device("cuda")
x = torch.randn(1, 3, S, SW, device=dev)
w = torch.randn(64, 3, 3, 3) * 0.2
wf = ops.pack_weights_fwd(w).to(dev)
pk = ops.conv_first_pack(wf)
b = torch.zeros(64, device=dev)
y = torch.empty(S, SW, 64, device=dev, dtype=torch.bfloat16)
dy = torch.randn(S, SW, 64, device=dev).bfloat16()
dx = torch.empty(1, 3, S, SW, device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
print(S, SW, "fwd  valu %.1f us  mfma %.1f us" % (t(lambda: ops.conv_first_fwd(x, wf, b, torch.bfloat16, out=y)),
                                            t(lambda: ops.conv_first_fwd(x, wf, b, torch.bfloat16, out=y, packed=pk))))
print(S, SW, "dgrad valu %.1f us  mfma %.1f us" % (t(lambda: ops.conv_first_dgrad(dy, wf, 3, out=dx)),
                                             t(lambda: ops.conv_first_dgrad(dy, wf, 3, out=dx, packed=pk))))

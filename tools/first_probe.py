"""Times the bf16 first-layer kernels (forward, forward + Gram slabs, dgrad) at 512^2 and 1024^2 - with a diagnostic
build (STV_LIB_PATH, -DSTV_FIRST_DIAG=n) this is the knock-out table of DESIGN §3.6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
dev = torch.device("cuda")
def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for S in (512, 1024):
    x = torch.randn(1, 3, S, S, device=dev)
    wf = ops.pack_weights_fwd(torch.randn(64, 3, 3, 3) * 0.2).to(dev)
    b = torch.zeros(64, device=dev)
    pk = ops.conv_first_pack(wf)
    y = torch.empty(S, S, 64, device=dev, dtype=torch.bfloat16)
    slabs = torch.empty(ops.gram_ksplit(S * S, 64), 64, 64, device=dev)
    dy = (torch.randn(S, S, 64, device=dev) * 0.1).bfloat16()
    dx = torch.empty(1, 3, S, S, device=dev)
    f = t(lambda: ops.conv_first_fwd(x, wf, b, torch.bfloat16, out=y, packed=pk))
    fg = t(lambda: ops.conv_first_fwd(x, wf, b, torch.bfloat16, out=y, packed=pk, gram_partials=slabs))
    d = t(lambda: ops.conv_first_dgrad(dy, wf, 3, out=dx, packed=pk))
    mb = S * S * 64 * 2 / 1e6
    print(f"{S}^2: fwd {f:6.1f} us ({mb / f:4.2f} TB/s written)   fwd+gram {fg:6.1f} us   dgrad {d:6.1f} us ({mb / d:4.2f} TB/s read)", flush=True)

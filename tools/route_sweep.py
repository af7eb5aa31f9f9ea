"""Time the dgrad-with-pool-routing epilogue (stv_conv_igemm_route) per tile config on the four layers that use it,
next to dgrad + maxpool_bwd as two passes.  usage: route_sweep.py [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for k, cin, cout in ((2, 128, 64), (4, 256, 128), (8, 512, 256), (16, 512, 512)):
    H = S // k
    dy = (torch.randn(H, H, cin, device=dev) * 0.5).bfloat16()
    w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.02).bfloat16())
    idx = torch.randint(0, 8, (H, H, cout), device=dev, dtype=torch.uint8)
    out = torch.empty(2 * H, 2 * H, cout, device=dev, dtype=torch.bfloat16)
    mid = torch.empty(H, H, cout, device=dev, dtype=torch.bfloat16)
    line = []
    for cfg in range(11):
        if cout <= 64 and cfg in (0, 2):
            continue
        os.environ["STV_CONV_CFG"] = str(cfg)
        if cfg in (7, 8):
            continue
        routed = t(lambda: ops.conv_igemm_route(dy, w, idx, out=out, flags=ops.MASK))
        def two():
            ops.conv_igemm(dy, w, None, out=mid)
            ops.maxpool_bwd_idx(idx, mid, 2 * H, 2 * H, out=out, flags=ops.MASK)
        line.append(f"cfg{cfg}: {routed:6.1f}/{t(two):6.1f}")
    os.environ.pop("STV_CONV_CFG")
    pick = ops.conv_tune(H, H, cin, cout, ops.TUNE_ROUTE, torch.bfloat16)
    print(f"{H}^2 {cin}->{cout} (tuner: cfg {pick}) routed/two-pass us:  " + "  ".join(line), flush=True)

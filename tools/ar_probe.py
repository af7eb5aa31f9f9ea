"""Tiles with and without A-column reuse (CfgAR) on the 512^2-net layer shapes; STV_CONV_CFG forced per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_visualizer_amd import ops
dev = "cuda"
def t(fn, n=40):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(64, 512, 512), (128, 256, 256), (128, 128, 256), (64, 256, 512), (256, 128, 128), (256, 256, 256), (32, 512, 512)]
pairs = [(4, 9), (1, 10), (5, 11), (3, 12)]
for (H, cin, cout) in shapes:
    x = (torch.randn(H, H, cin, device=dev) * 0.5).bfloat16()
    w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.03).bfloat16())
    b = torch.zeros(cout, device=dev); y = torch.empty(H, H, cout, device=dev, dtype=torch.bfloat16)
    row = []
    for base, ar in pairs:
        ts = []
        for cfg in (base, ar):
            os.environ["STV_CONV_CFG"] = str(cfg)
            ts.append(t(lambda: ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)))
        row.append(f"cfg{base}/{ar}: {ts[0]:5.1f}/{ts[1]:5.1f} us ({(ts[0]/ts[1]-1)*100:+.0f}%)")
    print(f"{H}^2 {cin}->{cout}: " + " | ".join(row))

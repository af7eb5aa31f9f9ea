"""Hot-loop time of a few conv shapes with the library named by STV_LIB_PATH (knock-out builds: -DSTV_DIAG=1 no
DMA traffic, 2 no output stores, 3 both; results are wrong, timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
dev = torch.device("cuda")
shapes = [(64, 64, 512, 512, 4), (128, 128, 256, 256, 1), (256, 256, 128, 128, 0), (128, 128, 512, 512, 0), (256, 256, 256, 256, 0), (512, 512, 128, 128, 0)]
def span(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
out = []
for (H, W, cin, cout, cfg) in shapes:
    os.environ["STV_CONV_CFG"] = str(cfg)
    x = torch.randn(H, W, cin, device=dev).bfloat16()
    w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.02).bfloat16())
    b = torch.zeros(cout, device=dev)
    y = torch.empty(H, W, cout, device=dev, dtype=torch.bfloat16)
    out.append(f"{H}x{W} {cin}->{cout} cfg{cfg}: {span(lambda: ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)):6.1f}")
print(os.path.basename(os.environ.get("STV_LIB_PATH", "libstv_hip.so")), " | ".join(out))

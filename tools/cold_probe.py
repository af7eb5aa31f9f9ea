"""Hot-loop vs in-step conditions for one conv launch: the same launch timed back to back (weights and input in
L2) and after 1 GB of unrelated traffic has gone through the caches (what the step's other layers do to it).
usage: cold_probe.py [H W cin cout cfg]..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
dev = torch.device("cuda")
shapes = [(64, 64, 512, 512, 3), (64, 64, 512, 512, 4), (32, 32, 512, 512, 8), (128, 128, 256, 256, 1), (64, 64, 512, 256, 7),
          (256, 256, 128, 128, 0), (128, 128, 512, 512, 0), (256, 256, 256, 256, 0)]
big = torch.empty(512 * 2 ** 20, device=dev, dtype=torch.uint8)
def span(fn, n):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def one(H, W, cin, cout, cfg):
    os.environ["STV_CONV_CFG"] = str(cfg)
    x = torch.randn(H, W, cin, device=dev).bfloat16()
    w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.02).bfloat16())
    b = torch.zeros(cout, device=dev)
    y = torch.empty(H, W, cout, device=dev, dtype=torch.bfloat16)
    conv = lambda: ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)
    flush = lambda: big.add_(1)                   # 512 MB read + 512 MB written: L2 and Infinity Cache turned over
    hot = span(conv, 30)
    t_flush = span(flush, 10)
    def both():
        flush(); conv()
    cold = span(both, 10) - t_flush
    return hot, cold
for (H, W, cin, cout, cfg) in shapes:
    hot, cold = one(H, W, cin, cout, cfg)
    print(f"{H}x{W} {cin}->{cout} cfg {cfg}: back to back {hot:6.1f} us   behind 1 GB of other traffic {cold:6.1f} us", flush=True)

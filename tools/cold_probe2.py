"""Which operand's coldness costs a conv launch inside the step?  After 1 GB of unrelated traffic, touch
(read) the weights, the input, or both before the timed launch (diagnostic).
usage: cold_probe2.py [H W cin cout cfg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
dev = torch.device("cuda")
shapes = [(64, 64, 512, 512, 4), (32, 32, 512, 512, 8), (128, 128, 256, 256, 1), (256, 256, 128, 128, 0)]
if len(sys.argv) == 6:
    shapes = [tuple(int(a) for a in sys.argv[1:6])]
big = torch.empty(512 * 2 ** 20, device=dev, dtype=torch.uint8)
def span(fn, n):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (H, W, cin, cout, cfg) in shapes:
    os.environ["STV_CONV_CFG"] = str(cfg)
    x = torch.randn(H, W, cin, device=dev).bfloat16()
    w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.02).bfloat16())
    b = torch.zeros(cout, device=dev)
    y = torch.empty(H, W, cout, device=dev, dtype=torch.bfloat16)
    sink = torch.zeros(4, device=dev)
    conv = lambda: ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)
    flush = lambda: big.add_(1)
    tw = lambda: sink[0:1].copy_(w.view(torch.int16).max().float().reshape(1))     # reads every weight byte
    tx = lambda: sink[1:2].copy_(x.view(torch.int16).max().float().reshape(1))
    ty = lambda: y.zero_()
    hot = span(conv, 30)
    res = {}
    # one 2-byte read per 4 KiB (address translation warmed, cache lines not): every 2048th bf16 element
    wp = lambda: sink[2:3].copy_(w.view(torch.int16).reshape(-1)[::2048].max().float().reshape(1))
    yp = lambda: sink[3:4].copy_(y.view(torch.int16).reshape(-1)[::2048].max().float().reshape(1))
    xp = lambda: sink[1:2].copy_(x.view(torch.int16).reshape(-1)[::2048].max().float().reshape(1))
    for name, pre in (("nothing", []), ("weights", [tw]), ("weights 1/4KiB", [wp]), ("input", [tx]), ("output written", [ty]), ("output 1/4KiB read", [yp]),
                      ("w+out 1/4KiB", [wp, yp]), ("w+in+out 1/4KiB", [wp, xp, yp]), ("weights+input", [tw, tx]), ("all three", [tw, tx, ty])):
        def base():
            flush()
            for p in pre: p()
        def full():
            base(); conv()
        res[name] = span(full, 10) - span(base, 10)
    print(f"{H}x{W} {cin}->{cout} cfg {cfg}: back to back {hot:5.1f} us | after a cache turn-over, having touched: " +
          ", ".join(f"{k} {v:5.1f}" for k, v in res.items()), flush=True)

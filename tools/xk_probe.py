"""One small-grid 3x3 layer with and without the K split across workgroups (STV_CONV_XK): event-timed,
alternating, on the shapes the rule in conv_igemm.hip::xk_wanted accepts.  usage: xk_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda")
ws = torch.zeros(ops.conv_workspace_bytes(), dtype=torch.uint8, device=dev)
ops.set_conv_workspace(ws)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timed(fn, cold):
    for _ in range(3):
        fn()
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tot = 0.0
    for _ in range(reps // 5):                     # behind 512 MB of unrelated traffic: operands where a step leaves them
        flush.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / (reps // 5) * 1e3


for (H, W, cin, cout) in ((64, 64, 512, 512), (64, 64, 256, 512), (64, 64, 512, 384)):
    x = torch.randn(H, W, cin, device=dev).bfloat16()
    w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.02).bfloat16())
    b = torch.zeros(cout, device=dev)
    y = torch.empty(H, W, cout, device=dev, dtype=torch.bfloat16)
    gf = 2 * 9 * cin * cout * H * W / 1e9
    for cold in (False, True):
        row = []
        for rnd in range(2):
            for xk in ("0", "1"):
                os.environ["STV_CONV_XK"] = xk
                row.append((xk, timed(lambda: ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT), cold)))
        print(f"{H}x{W} {cin}->{cout} {'cold' if cold else 'hot '}: " + "  ".join(f"xk={k} {us:6.1f} us ({gf / us * 1e3:5.0f} TF/s)" for k, us in row))
ops.set_conv_workspace(None)

"""Launch one conv shape a few times (for rocprofv3 --pmc runs). usage: conv_one.py H W cin cout cfg [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops
H, W, cin, cout, cfg = map(int, sys.argv[1:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
os.environ["STV_CONV_CFG"] = str(cfg)
dev = torch.device("cuda")
x = torch.randn(H, W, cin, device=dev).bfloat16()
w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.02).bfloat16())
b = torch.zeros(cout, device=dev)
y = torch.empty(H, W, cout, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"done: {H}x{W} {cin}->{cout} tile {cfg}: {us:.1f} us per launch (back to back), {2 * 9 * cin * cout * H * W / us / 1e6:.0f} TFLOP/s")

"""Times the weight-stationary conv kernel on the three short-K layer shapes (optionally with STV_WS_DIAG knock-outs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_visualizer_amd import ops
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
HH = int(os.environ.get("WS_H", "1024"))
for (H, cout, relu, pool) in ((HH, 64, True, True), (HH // 2, 128, False, False), (HH, 64, False, False)):
    x = (torch.randn(H, H, 64, device=dev) * 0.5).bfloat16()
    w = ops.block_weights((torch.randn(9, cout, 64, device=dev) * 0.06).bfloat16())
    b = torch.zeros(cout, device=dev)
    y = torch.empty(H, H, cout, device=dev, dtype=torch.bfloat16)
    fl = (ops.RELU_IN | ops.RELU_OUT) if relu else 0
    if pool:
        yp = torch.empty(H // 2, H // 2, cout, device=dev, dtype=torch.bfloat16)
        idx = torch.empty(H // 2, H // 2, cout, device=dev, dtype=torch.uint8)
        us = t(lambda: ops.conv_igemm_pool(x, w, b, flags=fl, out=y, pool_out=yp, pool_idx=idx))
    else:
        us = t(lambda: ops.conv_igemm(x, w, b, out=y, flags=fl))
    gf = 2 * 9 * 64 * cout * H * H / 1e9
    print(f"ws2={os.environ.get('STV_CONV_WS2','0')} skew={os.environ.get('STV_WS2_SKEW','1')} diag={os.environ.get('STV_WS_DIAG','0')} fwd {H}^2 64->{cout} relu={relu} pool={pool}: {us:7.1f} us  {gf / us * 1e3:7.0f} TFLOP/s")
H = HH
dy = (torch.randn(H, H, 64, device=dev) * 0.5).bfloat16(); z = (torch.randn(H, H, 64, device=dev)).bfloat16()
wb = ops.block_weights((torch.randn(9, 64, 64, device=dev) * 0.06).bfloat16()); S = (torch.randn(64, 64, device=dev) * 0.01).bfloat16()
out = torch.empty(H, H, 64, device=dev, dtype=torch.bfloat16)
us = t(lambda: ops.conv_igemm_dual(dy, wb, z, S, ref=z, out=out, flags=ops.MASK))
print(f"diag={os.environ.get('STV_WS_DIAG','0')} dgrad+mask+gram 1024^2 64->64: {us:7.1f} us  {2*(9*64+64)*64*H*H/1e9/us*1e3:7.0f} TFLOP/s")

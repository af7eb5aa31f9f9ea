"""conv2_2's two launches (128 -> 128 at half the image size: forward with ReLU-on-load + pool + arg-max map, pooled map
only; backward = masked dgrad + Gram term) on the weight-stationary kernel (STV_CONV_WS128=2) and on the general kernel
(=0), interleaved in one process, back to back and behind 1 GB of unrelated traffic (the state the step leaves them in).

    python tools/ws128_probe.py [image size, default 1024]
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from style_transfer_visualizer_amd import ops
dev = "cuda"
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = S // 2
C = 128
x = (torch.randn(H, H, C, device=dev) * 0.5).bfloat16()
w = ops.block_weights((torch.randn(9, C, C, device=dev) * 0.04).bfloat16())
wb = ops.block_weights((torch.randn(9, C, C, device=dev) * 0.04).bfloat16())
b = torch.zeros(C, device=dev)
y = torch.empty(H, H, C, device=dev, dtype=torch.bfloat16)
yp = torch.empty(H // 2, H // 2, C, device=dev, dtype=torch.bfloat16)
idx = torch.empty(H // 2, H // 2, C, device=dev, dtype=torch.uint8)
dy = (torch.randn(H, H, C, device=dev) * 0.5).bfloat16()
z = torch.randn(H, H, C, device=dev).bfloat16()
Sm = (torch.randn(C, C, device=dev) * 0.01).bfloat16()
out = torch.empty(H, H, C, device=dev, dtype=torch.bfloat16)
junk = torch.empty(256 << 20, device=dev, dtype=torch.float32)      # 1 GiB


def fwd():
    ops.conv_igemm_pool(x, w, b, flags=ops.RELU_IN | ops.RELU_OUT | ops.POOL_ONLY, out=y, pool_out=yp, pool_idx=idx)


def bwd():
    ops.conv_igemm_dual(dy, wb, z, Sm, ref=z, out=out, flags=ops.MASK)


def timed(fn, n=20, cold=False):
    for _ in range(3):
        fn()
    tot = 0.0
    if not cold:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    for _ in range(n):
        junk.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


gf_f = 2 * 9 * C * C * H * H / 1e9
gf_b = 2 * (9 * C + C) * C * H * H / 1e9
for rnd in range(3):
    for ws in ("2", "0"):
        os.environ["STV_CONV_WS128"] = ws
        os.environ["STV_CONV_WS"] = "1"
        uses = ops.conv_uses_ws(H, H, C, C, torch.bfloat16, flags=ops.RELU_IN | ops.RELU_OUT | ops.W_BLOCKED, has_pool=True)
        f_hot, b_hot = timed(fwd), timed(bwd)
        f_cold, b_cold = timed(fwd, 10, True), timed(bwd, 10, True)
        print(f"round {rnd} {H}^2 128->128 ws128={ws} (ws kernel: {uses}): fwd+pool {f_hot:6.1f} us ({gf_f / f_hot * 1e3:5.0f} TF/s), cold {f_cold:6.1f};  "
              f"dgrad+mask+gram {b_hot:6.1f} us ({gf_b / b_hot * 1e3:5.0f} TF/s), cold {b_cold:6.1f}", flush=True)

"""Time the fused conv variants (conv+pool+argmax map, dgrad + Gram 1x1 term) per tile config next to the
plain conv the tuner measures (diagnostic: does the plain-conv choice carry over?).  usage: conv_variant_sweep.py [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import _lib, ops
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
lib = _lib.load()
for k, C in ((1, 64), (2, 128), (4, 256), (8, 512)):
    H = S // k
    x = torch.randn(H, H, C, device=dev).bfloat16()
    w = ops.block_weights((torch.randn(9, C, C, device=dev) * 0.02).bfloat16())
    b = torch.zeros(C, device=dev)
    y = torch.empty(H, H, C, device=dev, dtype=torch.bfloat16)
    yp = torch.empty(H // 2, H // 2, C, device=dev, dtype=torch.bfloat16)
    idx = torch.empty(H // 2, H // 2, C, device=dev, dtype=torch.uint8)
    w2 = (torch.randn(C, C, device=dev) * 0.02).bfloat16()
    rows = {}
    for cfg in range(11):
        if C <= 64 and cfg in (0, 2):
            continue
        os.environ["STV_CONV_CFG"] = str(cfg)
        plain = t(lambda: ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT))
        pool = t(lambda: ops.conv_igemm_pool(x, w, b, flags=ops.RELU_OUT | ops.RELU_IN, out=y, pool_out=yp, pool_idx=idx)) \
            if cfg not in (7, 8) else float("nan")
        dual = t(lambda: ops.conv_igemm_dual(x, w, x, w2, ref=x, out=y, flags=ops.MASK))
        rows[cfg] = (plain, pool, dual)
    os.environ.pop("STV_CONV_CFG")
    pick = lib.stv_conv_tune(H, H, C, C, 9, 1, None)
    best = [min(rows, key=lambda c: rows[c][i] if rows[c][i] == rows[c][i] else 1e9) for i in range(3)]
    print(f"{H}x{H} C={C}: tuner picks {pick}; best plain/pool/dual = {best}")
    for cfg, (a, p, d) in rows.items():
        print(f"   cfg {cfg}: plain {a:7.1f}  pool {p:7.1f}  dual {d:7.1f} us")

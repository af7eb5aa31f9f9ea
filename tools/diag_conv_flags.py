"""Check conv_igemm flag combinations on small shapes against torch (diagnostic)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from style_transfer_visualizer_amd import ops
DEV = torch.device("cuda")
torch.manual_seed(0)
bad = 0
for dtype in (torch.float32, torch.bfloat16):
    for (cin, cout, H, W) in [(32, 32, 16, 16), (16, 16, 32, 32), (8, 8, 64, 64), (64, 64, 8, 8), (32, 64, 16, 16), (128, 128, 40, 40)]:
        for taps in (9, 1):
            if taps == 1 and cin != cout:
                continue
            for flags in (0, ops.MASK, ops.ACCUM, ops.MASK | ops.ACCUM):
                for trial in range(3):
                    x = torch.randn(1, cin, H, W) * (10.0 ** (trial - 1))
                    w = torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1) * 0.1
                    z = torch.randn(1, cout, H, W)
                    prev = torch.randn(1, cout, H, W)
                    q = lambda t: t.to(dtype).float()
                    ref = F.conv2d(q(x), q(w), None, padding=1 if taps == 9 else 0)
                    if flags & ops.MASK:
                        ref = ref * (q(z) > 0).float()
                    if flags & ops.ACCUM:
                        ref = ref + q(prev)
                    out = ops.to_nhwc(prev, dtype).to(DEV)
                    wp = (ops.pack_weights_fwd(w) if taps == 9 else w.reshape(1, cout, cin)).to(dtype).to(DEV)
                    ops.conv_igemm(ops.to_nhwc(x, dtype).to(DEV), wp, None, ref=ops.to_nhwc(z, dtype).to(DEV), out=out, flags=flags)
                    got = ops.from_nhwc(out).cpu()
                    err = float((got - ref).abs().max() / ref.abs().max())
                    lim = 2e-5 if dtype == torch.float32 else 2e-2
                    if err > lim:
                        bad += 1
                        idx = (got - ref).abs().flatten().argmax()
                        print(f"BAD {dtype} cin={cin} cout={cout} {H}x{W} taps={taps} flags={flags} trial={trial} err={err:.2e} at {idx}")
print("bad:", bad)

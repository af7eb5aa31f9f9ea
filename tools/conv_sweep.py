"""Time stv_conv_igemm per tile config (STV_CONV_CFG) on VGG layer shapes (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import ops

dev = torch.device("cuda")
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024     # image size: VGG19 forward and dgrad shapes up to conv5_1
shapes = [(S, S, 64, 64), (S // 2, S // 2, 64, 128), (S // 2, S // 2, 128, 64), (S // 2, S // 2, 128, 128),
          (S // 4, S // 4, 128, 256), (S // 4, S // 4, 256, 128), (S // 4, S // 4, 256, 256),
          (S // 8, S // 8, 256, 512), (S // 8, S // 8, 512, 256), (S // 8, S // 8, 512, 512), (S // 16, S // 16, 512, 512)]
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
names = ["8x128", "8x64", "4x128", "4x64", "4x64/K2", "8x64/2buf", "4x64/2buf", "2x64/K2", "1x64/K2", "16x64", "16x64/2buf", "2x32/K2", "4x32/K2",
         "8x64/m16", "16x64/m16", "4x64/m16", "2x32/K2/m16", "4x32/K2/m16", "16x128/2buf"]
only = [int(c) for c in os.environ.get("SWEEP_CFGS", "").split(",") if c] or list(range(len(names)))
for (H, W, cin, cout) in shapes:
    x = torch.randn(H, W, cin, device=dev).to(dtype)
    w = ops.block_weights((torch.randn(9, cout, cin, device=dev) * 0.02).to(dtype))
    b = torch.zeros(cout, device=dev)
    y = torch.empty(H, W, cout, device=dev, dtype=dtype)
    row = []
    for cfg in only:
        os.environ["STV_CONV_CFG"] = str(cfg)
        if cout <= 64 and cfg in (0, 2, 18):
            row.append("   -  ")
            continue
        for _ in range(3):
            ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            ops.conv_igemm(x, w, b, out=y, flags=ops.RELU_OUT)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        row.append(f"{2*9*cin*cout*H*W/ms/1e9:6.0f}")
    print(f"{H:5d}x{W:<5d} {cin:4d}->{cout:<4d} TF/s by cfg " + "  ".join(f"{names[c]}:{v}" for c, v in zip(only, row)), flush=True)

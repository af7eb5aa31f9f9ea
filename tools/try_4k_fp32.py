import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["STV_SYNTHETIC_WEIGHTS"] = "0"
import torch
from style_transfer_visualizer_amd import core_model, synthetic, optimizers
dev = torch.device("cuda")
H, W = 2160, 3840
content = synthetic.synthetic_image(0, H, W).to(dev)
style = synthetic.synthetic_image(1, 512, 512).to(dev)
model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21], precision="fp32").to(dev)
model.set_targets(style, content)
x = torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(0)).to(dev).requires_grad_(True)
adam = optimizers.HipAdam([x], lr=1e-2)
t0 = time.time()
vals = [float(adam.step(lambda: model.loss_and_grad(x, 1e5, 1.0)[2])) for _ in range(3)]
torch.cuda.synchronize()
print("4K fp32 totals", vals, "finite", bool(torch.isfinite(x).all()), f"{time.time() - t0:.2f} s", "mem GB", torch.cuda.max_memory_allocated() / 1e9)

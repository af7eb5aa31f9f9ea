import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from style_transfer_visualizer_amd import config as stv_config, core_model, optimization, synthetic
size, steps = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda")
cfg = stv_config.StyleTransferConfig.model_validate({})
oc = cfg.optimization
oc.steps, oc.init_method = steps, "random"
cfg.hardware.precision = "bf16"; cfg.video.create_video = False; cfg.video.final_only = True; cfg.output.log_every = 100
torch.manual_seed(0)
content = synthetic.synthetic_image(0, size, size).to(dev); style = synthetic.synthetic_image(1, size, size).to(dev)
model, x, opt = core_model.prepare_model_and_input(content, style, dev, oc, precision="bf16")
class Bar:
    def update(self, n): pass
    def set_postfix(self, *a, **k): pass
    def close(self): pass
t0 = time.time()
out, hist, _ = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=Bar()).run()
torch.cuda.synchronize()
tl = hist["total_loss"]
st = opt.device_state()
print(f"optimizer state at the end: {st}")
print("every 250th logged total loss:", " ".join(f"{v:.4e}" for v in tl[::250]))
nondecr = sum(1 for a, b in zip(tl[:-1], tl[1:]) if b > a)
print(f"steps whose loss went up: {nondecr} of {len(tl) - 1} (L-BFGS without line search is not monotone)")
print(f"{size}^2 {steps} steps in {time.time()-t0:.1f}s; loss first {tl[0]:.4e} min {min(tl):.4e} last {tl[-1]:.4e}; finite {all(math.isfinite(v) for v in tl)}; image finite {bool(torch.isfinite(out).all())} range [{float(out.min()):.2f},{float(out.max()):.2f}]")

"""Steady-state step time through OptimizationRunner (history full), nothing else (diagnostic A/B harness).
usage: step_time.py [size] [timed_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from style_transfer_visualizer_amd import config as stv_config, core_model, optimization, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda")
class Bar:
    def update(self, n=1): pass
    def set_postfix(self, *a, **k): pass
    def close(self): pass
cfg = stv_config.StyleTransferConfig.model_validate({})
oc = cfg.optimization
oc.steps, oc.init_method = 110 + timed, "random"
cfg.hardware.precision = os.environ.get("STV_PRECISION", "bf16")
cfg.video.create_video = False
cfg.output.log_every = int(os.environ.get("LOG_EVERY", "10"))
torch.manual_seed(0)
content = synthetic.synthetic_image(0, size, size).to(dev); style = synthetic.synthetic_image(1, size, size).to(dev)
model, x, opt = core_model.prepare_model_and_input(content, style, dev, oc, precision=cfg.hardware.precision)
marks = {}
def on_end(m):
    if m.step == 110:
        torch.cuda.synchronize(); marks["t0"] = time.perf_counter()
    elif m.step == 110 + timed:
        marks["h1"] = time.perf_counter()          # the host has ISSUED the timed steps (it may be ahead of the GPU)
        torch.cuda.synchronize(); marks["t1"] = time.perf_counter()
runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=Bar(), callbacks=optimization.OptimizationCallbacks(on_step_end=on_end))
_, hist, _ = runner.run()
ms = (marks["t1"] - marks["t0"]) / timed * 1e3
host = (marks["h1"] - marks["t0"]) / timed * 1e3
print(f"size {size}: {ms:.4f} ms/step = {1e3 / ms:.1f} steps/s at m=100   host issue {host:.4f} ms/step   final loss {hist['total_loss'][-1]:.5e}   "
      f"graph={os.environ.get('STV_HIP_GRAPH', '1')} next_w={os.environ.get('STV_NEXT_W', '1')}")

"""Host-side cost of one optimisation step (CPU seconds of this process per step) beside its wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from style_transfer_visualizer_amd import config as stv_config, core_model, optimization, synthetic
size, steps = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda")
cfg = stv_config.StyleTransferConfig.model_validate({})
oc = cfg.optimization
oc.steps, oc.init_method = steps, "random"
cfg.hardware.precision = "bf16"; cfg.video.create_video = False; cfg.video.final_only = True; cfg.output.log_every = 10
content = synthetic.synthetic_image(0, size, size).to(dev); style = synthetic.synthetic_image(1, size, size).to(dev)
model, x, opt = core_model.prepare_model_and_input(content, style, dev, oc, precision="bf16")
class Bar:
    def update(self, n): pass
    def set_postfix(self, *a, **k): pass
    def close(self): pass
marks = {}
def on_end(m):
    if m.step == 100:
        torch.cuda.synchronize(); marks["w0"] = time.perf_counter(); marks["c0"] = time.process_time(); marks["t0"] = time.thread_time()
    if m.step == steps:
        marks["c1"] = time.process_time(); marks["t1"] = time.thread_time(); torch.cuda.synchronize(); marks["w1"] = time.perf_counter()
optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=Bar(),
                                callbacks=optimization.OptimizationCallbacks(on_step_end=on_end)).run()
n = steps - 100
print(f"{size}^2: wall {1e6*(marks['w1']-marks['w0'])/n:.0f} us/step; main-thread CPU {1e6*(marks['t1']-marks['t0'])/n:.0f} us/step; process CPU {1e6*(marks['c1']-marks['c0'])/n:.0f} us/step")

"""Two (or K) independent images on ONE GPU, each on its own host thread and stream: does one image's L-BFGS update
(HBM-bound) overlap the other's closure (matrix-core / issue-bound)?  Aggregate steps/s against K = 1 (diagnostic).
usage: two_images_probe.py [size] [timed_steps] [K]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from style_transfer_visualizer_amd import config as stv_config, core_model, optimization, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
timed = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
class Bar:
    def update(self, n=1): pass
    def set_postfix(self, *a, **k): pass
    def close(self): pass
FILL = 110
gate = threading.Barrier(K)
marks = [dict() for _ in range(K)]
def work(i):
    torch.cuda.set_device(dev)
    cfg = stv_config.StyleTransferConfig.model_validate({})
    oc = cfg.optimization
    oc.steps, oc.init_method = FILL + timed, "random"
    cfg.hardware.precision = "bf16"
    cfg.video.create_video = False
    content = synthetic.synthetic_image(2 * i, size, size).to(dev); style = synthetic.synthetic_image(2 * i + 1, size, size).to(dev)
    model, x, opt = core_model.prepare_model_and_input(content, style, dev, oc, precision="bf16")
    def on_end(m):
        if m.step == FILL:
            torch.cuda.synchronize(); gate.wait(); marks[i]["t0"] = time.perf_counter()
        elif m.step == FILL + timed:
            torch.cuda.synchronize(); marks[i]["t1"] = time.perf_counter()
    runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=Bar(), callbacks=optimization.OptimizationCallbacks(on_step_end=on_end))
    _, hist, _ = runner.run()
    marks[i]["loss"] = hist["total_loss"][-1]
ths = [threading.Thread(target=work, args=(i,)) for i in range(K)]
for t in ths: t.start()
for t in ths: t.join()
t0 = min(m["t0"] for m in marks); t1 = max(m["t1"] for m in marks)
print(f"size {size} K={K}: {K * timed / (t1 - t0):.1f} steps/s aggregate ({(t1 - t0) / timed * 1e3:.4f} ms per step of each image); "
      f"final losses {['%.4e' % m['loss'] for m in marks]}")

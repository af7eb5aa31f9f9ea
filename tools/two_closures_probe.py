"""K independent CLOSURES (no optimizer) in flight on one GPU, one host thread + stream each: what part of the gain of
several images in flight (tools/two_images_probe.py) comes from the closure's own kernel boundaries (ramp / tail of
35 dependent launches) rather than from overlapping an HBM-bound L-BFGS update with a matrix-core-bound closure?
usage: two_closures_probe.py [size] [closures] [K]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from style_transfer_visualizer_amd import core_model, synthetic
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
gate = threading.Barrier(K)
marks = [dict() for _ in range(K)]
setup = threading.Lock()
def work(i):
    torch.cuda.set_device(dev)
    with setup:
        content = synthetic.synthetic_image(2 * i, size, size).to(dev); style = synthetic.synthetic_image(2 * i + 1, size, size).to(dev)
        model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21], precision="bf16").to(dev)
        model.set_targets(style, content)
        x = torch.randn(1, 3, size, size, device=dev).requires_grad_(True)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        for _ in range(20):
            model.loss_and_grad(x, 1e5, 1.0)
        side.synchronize(); gate.wait(); marks[i]["t0"] = time.perf_counter()
        for _ in range(reps):
            model.loss_and_grad(x, 1e5, 1.0)
        side.synchronize(); marks[i]["t1"] = time.perf_counter()
ths = [threading.Thread(target=work, args=(i,)) for i in range(K)]
for t in ths: t.start()
for t in ths: t.join()
span = max(m["t1"] for m in marks) - min(m["t0"] for m in marks)
print(f"size {size} K={K}: {K * reps / span:.1f} closures/s aggregate ({span / reps * 1e3:.4f} ms per closure of each image)")

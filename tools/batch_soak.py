"""main.style_transfer_batch end to end (PNG in, PNG out): N pairs at SIZE for STEPS L-BFGS steps with 1 and with K images in
flight on the one GPU - wall time of both and whether every result image is the same, bit for bit (diagnostic / soak).
usage: batch_soak.py [size] [steps] [pairs] [K]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from PIL import Image
from style_transfer_visualizer_amd import config as stv_config, main as stv_main, synthetic
from style_transfer_visualizer_amd.type_defs import InputPaths
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pairs_n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
tmp = tempfile.mkdtemp(prefix="stv_batch_")
for i in range(pairs_n + 1):
    img = synthetic.synthetic_image(i, size, size, normalize=False)[0].permute(1, 2, 0).mul(255).byte().numpy()
    Image.fromarray(img).save(os.path.join(tmp, f"img{i}.png"))
pairs = [InputPaths(content_path=os.path.join(tmp, f"img{i}.png"), style_path=os.path.join(tmp, f"img{pairs_n}.png")) for i in range(pairs_n)]
def run(k):
    cfg = stv_config.StyleTransferConfig.model_validate({})
    cfg.optimization.steps, cfg.optimization.init_method = steps, "random"
    cfg.video.create_video, cfg.video.final_only = False, True
    cfg.hardware.device, cfg.hardware.precision = "cuda", "bf16"
    cfg.output.output = os.path.join(tmp, f"out{k}")
    t0 = time.perf_counter()
    out = stv_main.style_transfer_batch(pairs, cfg, images_per_gpu=k)
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0
run(1)                                   # first touch of the process (library load, page-in): not timed
seq, t1 = run(1)
par, tk = run(K)
same = all(torch.equal(a, b) for a, b in zip(seq, par))
print(f"size {size}, {pairs_n} pairs x {steps} steps: one at a time {t1:.2f} s ({pairs_n * steps / t1:.0f} steps/s incl. set-up and PNG I/O), "
      f"{K} in flight {tk:.2f} s ({pairs_n * steps / tk:.0f} steps/s); results identical: {same}")

"""Where does the time of ONE style_transfer() call go outside its optimisation steps?  (diagnostic: cProfile of the second of
two calls - the first pays one-off costs such as importing torch, loading the library and creating the HIP context)
usage: setup_profile.py [size] [steps]"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from PIL import Image
from style_transfer_visualizer_amd import config as stv_config, main as stv_main, synthetic
from style_transfer_visualizer_amd.type_defs import InputPaths
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tmp = tempfile.mkdtemp(prefix="stv_setup_")
for i in range(3):
    img = synthetic.synthetic_image(i, size, size, normalize=False)[0].permute(1, 2, 0).mul(255).byte().numpy()
    Image.fromarray(img).save(os.path.join(tmp, f"img{i}.png"))
def run(i):
    cfg = stv_config.StyleTransferConfig.model_validate({})
    cfg.optimization.steps, cfg.optimization.init_method = steps, "random"
    cfg.video.create_video, cfg.video.final_only = False, True
    cfg.hardware.device, cfg.hardware.precision = "cuda", "bf16"
    cfg.output.output = os.path.join(tmp, f"out{i}")
    t0 = time.perf_counter()
    stv_main.style_transfer(InputPaths(content_path=os.path.join(tmp, f"img{i}.png"), style_path=os.path.join(tmp, "img2.png")), cfg)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
print(f"first call {run(0):.3f} s")
pr = cProfile.Profile()
pr.enable()
t = run(1)
pr.disable()
print(f"second call {t:.3f} s ({steps} steps at {size}^2)")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

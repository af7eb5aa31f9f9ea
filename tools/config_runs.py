"""BASELINE.json configs[1] / configs[2] end to end through the CLI (synthetic weights and images): 300 L-BFGS steps
at 512^2 and 500 at 1024^2, bf16, --no-video - wall time, steps/s including image load / PNG save, finite and
decreasing losses in the CSV."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
from style_transfer_visualizer_amd import cli, synthetic
for size, steps in ((512, 300), (1024, 500)):
    with tempfile.TemporaryDirectory() as d:
        for name, seed in (("content", 0), ("style", 1)):
            img = synthetic.synthetic_image(seed, size, size, normalize=False)[0].permute(1, 2, 0).mul(255).byte().numpy()
            Image.fromarray(img).save(os.path.join(d, f"{name}.png"))
        csv = os.path.join(d, "loss.csv")
        t0 = time.time()
        cli.main(["--content", os.path.join(d, "content.png"), "--style", os.path.join(d, "style.png"), "--steps", str(steps),
                  "--init", "random", "--device", "cuda", "--no-video", "--final-only", "--seed", "0", "--precision", "bf16",
                  "--output", os.path.join(d, "out"), "--log-loss", csv, "--log-every", "50"])
        wall = time.time() - t0
        rows = [r.split(",") for r in open(csv).read().strip().splitlines()[1:]]
        totals = [float(r[3]) for r in rows]
        ok = bool(np.all(np.isfinite(totals))) and totals[-1] < totals[0]
        png = os.path.join(d, "out", "stylized_content_x_style.png")
        print(f"{size}x{size}, {steps} steps: wall {wall:.2f} s ({steps / wall:.0f} steps/s incl. model build, tuning, I/O); "
              f"loss {totals[0]:.4g} -> {totals[-1]:.4g}; finite+decreasing={ok}; png={Image.open(png).size}", flush=True)

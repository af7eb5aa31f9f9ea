"""bf16 mode late in a run: HIP stored activations vs the rounding-faithful oracle, layer by layer, both
free-running from the same image and op-by-op on the HIP path's own inputs (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
import torch
from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import config as stv_config, core_model, optimization, ops, synthetic
dev = torch.device("cuda")
size, steps = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 150
class Bar:
    def update(self, n=1): pass
    def set_postfix(self, *a, **k): pass
    def close(self): pass
cfg = stv_config.StyleTransferConfig.model_validate({})
oc = cfg.optimization; oc.steps, oc.init_method = steps, "random"
cfg.hardware.precision = "bf16"; cfg.video.create_video = False
content = synthetic.synthetic_image(0, size, size); style = synthetic.synthetic_image(1, size, size)
torch.manual_seed(0)
model, x, opt = core_model.prepare_model_and_input(content.to(dev), style.to(dev), dev, oc, precision="bf16")
optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=Bar()).run()
s, c, t = model.loss_and_grad(x, 1e5, 1.0)
torch.cuda.synchronize()
print("HIP losses at the step-%d image: style %.6e content %.6e" % (steps, float(s), float(c)))
eng = next(iter(model._engines.values()))
weights = synthetic.synthetic_conv_weights(0)
oracle = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), [0, 5, 10, 19, 28], [21], bf16_storage=True)
oracle.set_targets(style, content)
xc = x.detach().cpu()
def r16(v): return v.bfloat16().float()
h_free = xc
acts_free = {}
with torch.no_grad():
    for li in range(29):
        layer = oracle.program[li]
        h_free = ocm.run_layer(layer, h_free)
        if layer[0] in ("conv", "pool"): h_free = r16(h_free)
        acts_free[li] = h_free
print(f"{'node':14s} {'differs(free)':>13s} {'mean d/ulp(free)':>16s} {'rms rel(free)':>13s} | own-input op check: differs, mean d/ulp")
prev_hip = xc
for nd in eng.sched.nodes:
    li = nd.layer + (1 if nd.dst.relu_fused else 0)
    a_hip = ops.from_nhwc(nd.dst.act).cpu()
    a_free = acts_free[li]
    ulp = (a_free.abs().clamp_min(1e-30)).log2().floor().exp2() * 2.0 ** -7
    d = (a_hip - a_free)
    nz = d != 0
    # op on the HIP path's own stored input
    with torch.no_grad():
        src = xc if nd.src is None else ops.from_nhwc(nd.src.act).cpu()
        if nd.kind in ("conv", "conv_first"):
            inp = torch.relu(src) if nd.relu_in else src
            w, b = oracle.program[nd.layer][1], oracle.program[nd.layer][2]
            z = torch.nn.functional.conv2d(inp, w, b, padding=1)
            if nd.dst.relu_fused: z = torch.relu(z)
            own = r16(z)
        elif nd.kind == "pool":
            own = torch.nn.functional.max_pool2d(src, 2, 2)
        else:
            own = torch.relu(src)
    d2 = a_hip - own
    print(f"{nd.kind:10s} L{nd.layer:2d} {float(nz.float().mean()):13.3e} {float((d / ulp)[nz].mean()) if nz.any() else 0:16.3e} "
          f"{float(d.norm() / a_free.norm()):13.3e} | {float((d2 != 0).float().mean()):.3e} {float((d2 / ulp)[d2 != 0].mean()) if (d2 != 0).any() else 0:.3e}")
# content target and loss
T_hip = model.content_targets[0].float().cpu()
T_or = oracle.content_targets[0]
F_hip = ops.from_nhwc(next(n for n in eng.sched.nodes if n.layer == 21).dst.act).cpu()
F_or = acts_free[21]
print("content target: differs %.3e, mean signed diff %.3e, rms rel %.3e" % (float((T_hip != T_or).float().mean()), float((T_hip - T_or).mean()), float((T_hip - T_or).norm() / T_or.norm())))
for nm, F, T in (("HIP F, HIP T", F_hip, T_hip), ("HIP F, oracle T", F_hip, T_or), ("oracle F, HIP T", F_or, T_hip), ("oracle F, oracle T", F_or, T_or)):
    print(f"content mse {nm:20s} {float(((F - T) ** 2).mean()):.6e}")
print("feature rms at conv4_2: %.3f, |F-T| rms %.3f" % (float(F_or.pow(2).mean().sqrt()), float((F_or - T_or).pow(2).mean().sqrt())))

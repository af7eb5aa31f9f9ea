"""Layer-by-layer activation/gradient comparison HIP vs torch-CPU autograd at one image (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from tests.conftest import GoldenCase
from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import core_model, ops

DEV = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "vgg19_content_lbfgs"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
case = GoldenCase(name); m = case.meta
weights = case.weights()
core_model.initialize_vgg = lambda: core_model.build_vgg_features(weights, case.cfg).eval()
model = core_model.StyleContentModel(m["style_layers"], m["content_layers"]).to(DEV)
content, style = case.images()
model.set_targets(style.to(DEV), content.to(DEV))
x = case.tensor("x0").to(DEV).requires_grad_(True)
compact = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
state, work = ops.lbfgs_alloc(x.numel(), 100, DEV, compact=compact)
for step in range(nsteps):
    model.loss_and_grad(x, m["style_w"], m["content_w"])
    ops.lbfgs_step(x.detach(), x.grad, state, work, 100, min(step, 100), 1.0, compact=compact)
model.loss_and_grad(x, m["style_w"], m["content_w"])
torch.cuda.synchronize()
eng = next(iter(model._engines.values()))

for dt, label in ((torch.float32, "cpu-fp32"), (torch.float64, "cpu-fp64")):
    prog = ocm.vgg_program([(w.to(dt), b.to(dt)) for w, b in weights], case.cfg)
    oracle = ocm.OracleModel(prog, m["style_layers"], m["content_layers"])
    oracle.set_targets(style.to(dt), content.to(dt))
    xr = x.detach().cpu().to(dt).requires_grad_(True)
    acts = []
    h = xr
    last = max(list(m["style_layers"]) + list(m["content_layers"]))
    for li in range(last + 1):
        h = ocm.run_layer(prog[li], h)
        h.retain_grad()
        acts.append(h)
    s_losses, c_losses = [], []
    for j, blk in enumerate(oracle.blocks):
        f = acts[blk[-1]]
        if j in oracle.style_ids:
            s_losses.append(F.mse_loss(ocm.gram_matrix(f), oracle.style_targets[oracle.style_ids.index(j)]))
        if j in oracle.content_ids:
            c_losses.append(F.mse_loss(f, oracle.content_targets[oracle.content_ids.index(j)]))
    total = m["style_w"] * torch.stack(s_losses).sum() + m["content_w"] * torch.stack(c_losses).sum()
    total.backward()
    print(f"== vs {label}: total {float(total):.8e}")
    for nd in eng.sched.nodes:
        li = nd.layer + (1 if nd.dst.relu_fused else 0)
        a_ref = acts[li].detach()
        g_ref = acts[nd.layer].grad
        a_hip = ops.from_nhwc(nd.dst.act).cpu().double()
        g_hip = ops.from_nhwc(nd.dst.grad).cpu().double()
        ea = float((a_hip - a_ref.double()).abs().max() / (a_ref.abs().max() + 1e-30))
        eg = float((g_hip - g_ref.double()).norm() / (g_ref.double().norm() + 1e-30))
        egm = float((g_hip - g_ref.double()).abs().max() / (g_ref.abs().max() + 1e-30))
        print(f"  {nd.kind:10s} layer {nd.layer:2d} {tuple(a_ref.shape)}  act max-err {ea:.2e} | grad rel-rms {eg:.2e} max {egm:.2e}")
    gx = x.grad.cpu().double()
    print(f"  image grad rel-rms {float((gx - xr.grad.double()).norm() / xr.grad.double().norm()):.2e}")

# ---- where do pooling decisions differ? ----
prog = ocm.vgg_program(weights, case.cfg)
xr = x.detach().cpu()
h = xr
acts = []
for li in range(max(list(m["style_layers"]) + list(m["content_layers"])) + 1):
    h = ocm.run_layer(prog[li], h)
    acts.append(h)
for nd in eng.sched.nodes:
    if nd.kind != "pool":
        continue
    a_hip = ops.from_nhwc(nd.src.act).cpu()
    a_cpu = acts[nd.layer - 1]
    _, i_hip = F.max_pool2d(a_hip, 2, 2, return_indices=True)
    _, i_cpu = F.max_pool2d(a_cpu, 2, 2, return_indices=True)
    diff = (i_hip != i_cpu).nonzero()
    print(f"pool layer {nd.layer}: {len(diff)} windows with a different argmax")
    for d in diff[:6]:
        _, c, oy, ox = d.tolist()
        wh = a_hip[0, c, 2 * oy:2 * oy + 2, 2 * ox:2 * ox + 2].flatten().tolist()
        wc = a_cpu[0, c, 2 * oy:2 * oy + 2, 2 * ox:2 * ox + 2].flatten().tolist()
        print("   c", c, "oy", oy, "ox", ox, "hip", [f"{v:.9g}" for v in wh], "cpu", [f"{v:.9g}" for v in wc])

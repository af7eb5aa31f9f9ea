"""HIP fp32 vs CPU fp32 vs CPU fp64, layer by layer at one image: activation / gradient error against fp64 for
both fp32 paths and the number of ReLU sign decisions each takes differently from fp64 (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from tests.conftest import GoldenCase
from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import core_model, ops

DEV = torch.device("cuda")
name = sys.argv[1] if len(sys.argv) > 1 else "vgg19_random_adam"
case = GoldenCase(name); m = case.meta
weights = case.weights()
core_model.initialize_vgg = lambda: core_model.build_vgg_features(weights, case.cfg).eval()
model = core_model.StyleContentModel(m["style_layers"], m["content_layers"]).to(DEV)
content, style = case.images()
model.set_targets(style.to(DEV), content.to(DEV))
x = case.tensor("x0").to(DEV).requires_grad_(True)
model.loss_and_grad(x, m["style_w"], m["content_w"])
torch.cuda.synchronize()
eng = next(iter(model._engines.values()))

def cpu_eval(dt):
    prog = ocm.vgg_program([(w.to(dt), b.to(dt)) for w, b in weights], case.cfg)
    oracle = ocm.OracleModel(prog, m["style_layers"], m["content_layers"])
    oracle.set_targets(style.to(dt), content.to(dt))
    xr = x.detach().cpu().to(dt).requires_grad_(True)
    acts, h = [], xr
    last = max(list(m["style_layers"]) + list(m["content_layers"]))
    for li in range(last + 1):
        h = ocm.run_layer(prog[li], h); h.retain_grad(); acts.append(h)
    s_l, c_l = [], []
    for j, blk in enumerate(oracle.blocks):
        f = acts[blk[-1]]
        if j in oracle.style_ids: s_l.append(F.mse_loss(ocm.gram_matrix(f), oracle.style_targets[oracle.style_ids.index(j)]))
        if j in oracle.content_ids: c_l.append(F.mse_loss(f, oracle.content_targets[oracle.content_ids.index(j)]))
    total = m["style_w"] * torch.stack(s_l).sum() + m["content_w"] * torch.stack(c_l).sum()
    total.backward()
    return acts, xr.grad, float(total)

a32, g32, t32 = cpu_eval(torch.float32)
a64, g64, t64 = cpu_eval(torch.float64)
print(f"total: fp64 {t64:.9e}  cpu32 {t32:.9e}")
def rel(a, b): return float((a.double() - b).norm() / (b.norm() + 1e-300))
print(f"{'node':22s} {'act hip':>9s} {'act cpu':>9s} | {'grad hip':>9s} {'grad cpu':>9s} | flips hip cpu (of)")
for nd in eng.sched.nodes:
    li = nd.layer + (1 if nd.dst.relu_fused else 0)
    ah = ops.from_nhwc(nd.dst.act).cpu()
    gh = ops.from_nhwc(nd.dst.grad).cpu()
    z64 = a64[nd.layer].detach()
    fh = fc = -1
    if nd.kind in ("conv", "conv_first"):
        on64 = z64 > 0
        fh = int(((ah > 0) != on64).sum()) if nd.dst.relu_fused else int(((ah > 0) != on64).sum())
        fc = int(((a32[nd.layer].detach() > 0) != on64).sum())
    print(f"{nd.kind:10s} L{nd.layer:2d} C{nd.dst.C:4d} {rel(ah, a64[li].detach()):9.2e} {rel(a32[li].detach(), a64[li].detach()):9.2e} | "
          f"{rel(gh, a64[nd.layer].grad):9.2e} {rel(a32[nd.layer].grad, a64[nd.layer].grad):9.2e} | {fh:6d} {fc:6d} ({z64.numel()})")
print(f"image gradient: hip {rel(x.grad.cpu(), g64):.3e}  cpu32 {rel(g32, g64):.3e}   hip vs cpu32 max/scale {float((x.grad.cpu()-g32).abs().max()/g32.abs().max()):.2e}")

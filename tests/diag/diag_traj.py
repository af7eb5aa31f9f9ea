"""Per-step: HIP gradient vs CPU-oracle gradient evaluated at the SAME x, and x drift vs the oracle trajectory."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.conftest import GoldenCase
from oracle import core_model_ref as ocm, optim_ref
from style_transfer_visualizer_amd import core_model, ops, config as stv_config

DEV = torch.device("cuda")
torch.set_num_threads(16)
names = sys.argv[1:] or ["mini_random_lbfgs_nonorm", "mini_content_lbfgs"]
for name in names:
    case = GoldenCase(name); m = case.meta
    weights = case.weights()
    prog = ocm.vgg_program(weights, case.cfg)
    oracle = ocm.OracleModel(prog, m["style_layers"], m["content_layers"])
    content, style = case.images(); oracle.set_targets(style, content)
    core_model.initialize_vgg = lambda: core_model.build_vgg_features(weights, case.cfg).eval()
    model = core_model.StyleContentModel(m["style_layers"], m["content_layers"]).to(DEV)
    model.set_targets(style.to(DEV), content.to(DEV))
    x0 = case.tensor("x0")
    x_ref = x0.clone()
    ref = optim_ref.LbfgsRef(x_ref.view(-1), lr=1.0)
    x = x0.clone().to(DEV).requires_grad_(True)
    state, work = ops.lbfgs_alloc(x.numel(), 100, DEV)
    for step in range(m["steps"]):
        def closure():
            s, c, t, g = ocm.loss_and_grad(oracle, x_ref, m["style_w"], m["content_w"])
            return t, g
        ref.step(closure)
        s, c, t = model.loss_and_grad(x, m["style_w"], m["content_w"])
        so, co, to, go = ocm.loss_and_grad(oracle, x.detach().cpu(), m["style_w"], m["content_w"])
        g = x.grad.cpu()
        gmax = float((g - go).abs().max() / go.abs().max()); grms = float((g - go).norm() / go.norm())
        ops.lbfgs_step(x.detach(), x.grad, state, work, 100, min(step, 100), 1.0)
        err = float((x.detach().cpu() - x_ref).abs().max()) / float(x_ref.abs().max())
        print(f"{name} step {step+1}: grad(HIP vs oracle @same x) max {gmax:.2e} rms {grms:.2e} | loss rel {abs(float(t)-float(to))/float(to):.1e} | x drift {err:.2e}")

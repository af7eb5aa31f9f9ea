"""Isolate the device L-BFGS from the conv kernels: CPU-oracle gradients fed to both optimizers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.conftest import GoldenCase
from oracle import core_model_ref as ocm, optim_ref
from style_transfer_visualizer_amd import ops

DEV = torch.device("cuda")
torch.set_num_threads(16)
for name in ["mini_random_lbfgs_nonorm", "mini_content_lbfgs", "mini_white_lbfgs"]:
    case = GoldenCase(name); m = case.meta
    prog = ocm.vgg_program(case.weights(), case.cfg)
    model = ocm.OracleModel(prog, m["style_layers"], m["content_layers"])
    content, style = case.images(); model.set_targets(style, content)
    x0 = case.tensor("x0")
    x_ref = x0.clone()
    ref = optim_ref.LbfgsRef(x_ref.view(-1), lr=1.0)
    x = x0.clone().to(DEV)
    state, work = ops.lbfgs_alloc(x.numel(), 100, DEV)
    for step in range(m["steps"]):
        def closure():
            s, c, t, g = ocm.loss_and_grad(model, x_ref, m["style_w"], m["content_w"])
            return t, g
        ref.step(closure)
        s, c, t, g = ocm.loss_and_grad(model, x.cpu(), m["style_w"], m["content_w"])
        ops.lbfgs_step(x, g.to(DEV).contiguous(), state, work, 100, min(step, 100), 1.0)
        err = float((x.cpu() - x_ref).abs().max()) / float(x_ref.abs().max())
        st = state.cpu()
        ints, fl = st.view(torch.int32), st.view(torch.float32)
        print(f"{name} step {step+1}: x err {err:.2e} | dev n_iter {int(ints[0])} m {int(ints[1])} t {float(fl[6]):.6e} H {float(fl[7]):.6e} gtd {float(fl[8]):.6e} ys {float(fl[10]):.6e} yy {float(fl[11]):.6e}"
              f" | ref n_iter {ref.n_iter} m {len(ref.old_dirs)} t {float(ref.t):.6e} H {float(ref.H_diag):.6e}")

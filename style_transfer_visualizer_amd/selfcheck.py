"""``__graft_entry__.smoke()``: one small invocation of the hot path on cuda:0, checked
against the CPU oracle (the oracle is imported here only as the checker)."""
from __future__ import annotations

import torch


def smoke() -> None:
    from oracle import core_model_ref as ocm  # checker only

    from . import core_model, synthetic
    from .optimizers import HipLBFGS

    if not torch.cuda.is_available():
        msg = "smoke() needs a GPU: the hot path has no CPU fallback"
        raise RuntimeError(msg)
    dev = torch.device("cuda:0")
    cfg = (8, 8, "M", 16, 16, "M", 32, 32, 32, 32, "M", 64, 64, 64, 64, "M", 64, 64, 64, 64, "M")
    weights = synthetic.synthetic_conv_weights(3, cfg)
    S, C = [0, 5, 10, 19, 28], [21]
    content = synthetic.synthetic_image(0, 64, 64)
    style = synthetic.synthetic_image(1, 80, 64)
    x0 = synthetic.synthetic_image(2, 64, 64)

    oracle = ocm.OracleModel(ocm.vgg_program(weights, cfg), S, C)
    oracle.set_targets(style, content)
    s_ref, c_ref, t_ref, g_ref = ocm.loss_and_grad(oracle, x0, 1e5, 1.0)

    saved = core_model.initialize_vgg
    core_model.initialize_vgg = lambda: core_model.build_vgg_features(weights, cfg).eval()
    try:
        model = core_model.StyleContentModel(S, C).to(dev)
    finally:
        core_model.initialize_vgg = saved
    model.set_targets(style.to(dev), content.to(dev))
    x = x0.to(dev).requires_grad_(True)
    opt = HipLBFGS([x], lr=1.0)

    def closure():
        _, _, total = model.loss_and_grad(x, 1e5, 1.0)
        return total

    loss = opt.step(closure)
    torch.cuda.synchronize()
    rel = abs(float(loss) - float(t_ref)) / abs(float(t_ref))
    gerr = float((x.grad.cpu() - g_ref).abs().max() / g_ref.abs().max())
    if rel > 1e-4 or gerr > 1e-4:
        msg = f"smoke parity failed: loss rel err {rel:.2e}, grad err {gerr:.2e}"
        raise RuntimeError(msg)
    print(f"smoke ok: total {float(loss):.6e} (oracle {float(t_ref):.6e}), grad err {gerr:.1e}")

"""Print actual GPU-vs-golden errors per fixture (diagnostic, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.conftest import GoldenCase, GOLDEN_CASES
from style_transfer_visualizer_amd import core_model, optimization, optimizers, config as stv_config

DEV = torch.device("cuda")
class Bar:
    def update(self, n=1): pass
    def set_postfix(self, *a, **k): pass
    def close(self): pass

for name in GOLDEN_CASES:
    case = GoldenCase(name); m = case.meta
    weights = case.weights()
    core_model.initialize_vgg = lambda: core_model.build_vgg_features(weights, case.cfg).eval()
    cfg = stv_config.StyleTransferConfig.model_validate({})
    oc = cfg.optimization
    oc.steps, oc.style_w, oc.content_w = m["steps"], m["style_w"], m["content_w"]
    oc.init_method = m["init_method"]; oc.style_layers = list(m["style_layers"]); oc.content_layers = list(m["content_layers"])
    oc.normalize = m["normalize"]; cfg.output.log_every = 2; cfg.video.create_video = False
    content, style = case.images()
    model, x, opt = core_model.prepare_model_and_input(content.to(DEV), style.to(DEV), DEV, oc)
    with torch.no_grad(): x.copy_(case.tensor("x0").to(DEV))
    s, c, t = model.loss_and_grad(x, m["style_w"], m["content_w"])
    g = x.grad.cpu().numpy(); gr = case.arrays["grad_step1"]
    gerr_max = np.abs(g-gr).max()/np.abs(gr).max(); gerr_rms = np.sqrt(((g-gr)**2).mean())/np.sqrt((gr**2).mean())
    lerr = abs(float(t)-case.arrays["total_loss"][0])/case.arrays["total_loss"][0]
    if m["optimizer"] == "adam": opt = optimizers.HipAdam([x], lr=m["adam_lr"])
    runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=Bar())
    out, hist, _ = runner.run()
    xf = case.arrays["x_final"]
    xerr = np.abs(out.detach().cpu().numpy()-xf).max()/np.abs(xf).max()
    herr = np.max(np.abs(np.array(hist["total_loss"])-case.arrays["total_loss"])/case.arrays["total_loss"])
    print(f"{name:28s} grad err max {gerr_max:.2e} rms {gerr_rms:.2e} | loss1 {lerr:.2e} | hist max rel {herr:.2e} | x_final {xerr:.2e}")

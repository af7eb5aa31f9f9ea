"""BASELINE.json configs[1] and configs[2] at their REAL length, through ``OptimizationRunner``:

* configs[1]: one 512x512 image, 300 L-BFGS steps, VGG19 bf16 storage - and the same run in fp32 parity mode,
  where the comparison with the oracle is at rounding level;
* configs[2]: one 1024x1024 image, 500 steps, fp32 Gram / bf16 conv

(the reference loop: optimization.py:162-202, one ``optimizer.step(closure)`` per step).  The bench's inputs:
synthetic weights seed 0, content seed 0, style seed 1, ``init_method=random`` on seed 0, default layers and
weights.  What is asserted:

* integer bookkeeping, bit-exact: step ids 1..N, one closure per step, history length, the logging steps;
* every 50 (512^2) / 250 (1024^2) steps the CPU oracle - rounding to bf16 exactly where the kernels do - is
  evaluated AT THE IMAGE THE HIP PATH HOLDS and must give the loss the HIP path logged for it (chaos-free:
  nothing is compared between two free-running trajectories);
* the device L-BFGS state (``n_iter``, history length, skip / no-update flags) equals, at each of the first 110
  steps, that of an ``oracle.optim_ref.LbfgsRef`` twin fed the same gradients: the ramp to 100 pairs, the first
  ten evictions; at the end ``hist_len == 100`` and ``n_iter == steps``;
* the run optimises: the loss falls and stays finite.

The loss curves go to ``gpurun_out/r03_loss_curve_<size>_<precision>.csv`` (copied to ``profiles/``).
"""
from __future__ import annotations

import os
import time

import numpy as np
import pytest
import torch

from oracle import core_model_ref as ocm
from oracle import optim_ref
from style_transfer_visualizer_amd import config as stv_config
from style_transfer_visualizer_amd import core_model, optimization, synthetic
from tests.conftest import ROOT, record_parity
from tests.test_gpu_fullsize import _fused_style_taps

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
S_LAYERS, C_LAYERS = [0, 5, 10, 19, 28], [21]
TWIN_STEPS = 110


class _Bar:
    def update(self, n=1):
        return None

    def set_postfix(self, *a, **k):
        return None

    def close(self):
        return None


@pytest.mark.parametrize("size,steps,every,precision", [(512, 300, 50, "bf16"), (512, 300, 50, "fp32"), (1024, 500, 250, "bf16")])
def test_config_runs_at_full_length(size, steps, every, precision, monkeypatch):
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    bf16 = precision == "bf16"
    # losses vs the oracle at the same image.  fp32 (parity mode, the reference's arithmetic): rounding level.
    # bf16: the oracle rounds where the kernels round (and is given the model's targets, see below); fp32 sums
    # in another order still flip roundings layer by layer (tests/test_gpu_bf16_layerwise.py bounds every stored
    # tensor to one ulp), which moves the losses by ~1e-4: same 2e-3 as tests/test_gpu_fullsize.py.
    ltol = 2e-3 if bf16 else 1e-4
    case = f"configs[{1 if size == 512 else 2}] {size}x{size} x{steps} {precision}"
    cfg = stv_config.StyleTransferConfig.model_validate({})
    oc = cfg.optimization
    oc.steps, oc.init_method = steps, "random"
    cfg.hardware.precision = precision
    cfg.output.log_every = 10
    cfg.video.create_video = False
    content = synthetic.synthetic_image(0, size, size)
    style = synthetic.synthetic_image(1, size, size)
    torch.manual_seed(0)
    model, x, opt = core_model.prepare_model_and_input(content.to(DEV), style.to(DEV), DEV, oc, precision=precision)
    n = x.numel()

    seen, images, grads, dev_states = [], {}, [], []

    def on_end(mt):
        seen.append((mt.step, mt.has_values))
        if mt.step % every == every - 1 or mt.step == steps - 1:
            images[mt.step + 1] = x.detach().cpu().clone()        # the image step (mt.step + 1) evaluates
        if mt.step <= TWIN_STEPS and size <= 512:        # (1024^2: 35 s of host time for the same state machine)
            grads.append(x.grad.detach().cpu().clone().view(-1))
            st = opt.device_state()
            dev_states.append((st["n_iter"], st["hist_len"], st["skip"], st["no_update"]))
    runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=_Bar(),
                                             callbacks=optimization.OptimizationCallbacks(on_step_end=on_end))
    t0 = time.time()
    out, history, _ = runner.run()
    torch.cuda.synchronize()
    wall = time.time() - t0

    # ---- integer bookkeeping: bit-exact (reference optimization.py:162-202, loss_accumulator.py:95-125) ----------
    assert [s for s, _ in seen] == list(range(1, steps + 1))
    assert [s for s, has in seen if has] == list(range(10, steps + 1, 10))
    assert runner._closure_calls == steps
    assert len(history["total_loss"]) == len(history["style_loss"]) == len(history["content_loss"]) == steps
    end = opt.device_state()
    assert end["n_iter"] == steps and end["hist_len"] == 100 and end["skip"] == 0, end
    totals = np.asarray(history["total_loss"])
    assert np.isfinite(totals).all() and np.isfinite(np.asarray(history["style_loss"])).all()
    assert totals[-1] < 0.5 * totals[0], f"{case}: loss {totals[0]:.4e} -> {totals[-1]:.4e}"
    assert torch.isfinite(out).all()

    # ---- device L-BFGS state vs the oracle optimizer fed the same gradients (first 110 steps) -----------------------
    twin_x = torch.zeros(n)
    twin = optim_ref.LbfgsRef(twin_x, lr=1.0)
    zero = torch.tensor(0.0)
    for k, g in enumerate(grads):
        n_before = twin.n_iter
        twin.step(lambda: (zero, g))
        want = (twin.n_iter, len(twin.old_dirs), int(twin.n_iter == n_before), 0)
        assert dev_states[k][:3] == want[:3], f"{case} step {k + 1}: device state {dev_states[k]} vs oracle optimizer {want}"
        assert dev_states[k][3] == 0
    assert size > 512 or (len(twin.old_dirs) == 100 and twin.n_iter == TWIN_STEPS)
    del grads, twin

    # ---- the oracle at the same image --------------------------------------------------------------------
    fused = _fused_style_taps(model)
    weights = synthetic.synthetic_conv_weights(0)
    oracle = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), S_LAYERS, C_LAYERS, bf16_storage=bf16,
                             fused_style_taps=fused if bf16 else None)
    oracle.set_targets(style, content)
    if bf16:
        # In bf16 the TARGETS are rounded tensors too, and which way each element rounded is part of the problem
        # the run solved: 70 % of the content target's elements differ by an ulp between two correct bf16
        # evaluations (tests/diag/diag_bf16_late.py), and after 100+ steps the image has been fitted to the HIP
        # path's realisation of that rounding - against another realisation its content score is ~0.7 % higher,
        # for the HIP features and the oracle's alike.  "Oracle at the same image" therefore means: same image,
        # same targets (the model's public ``content_targets`` / ``style_targets``, reference core_model.py:218-232);
        # that the targets themselves are right is the business of tests/test_gpu_fullsize.py.
        oracle.content_targets = [t.float().cpu().contiguous() for t in model.content_targets]
        oracle.style_targets = [t.float().cpu() for t in model.style_targets]
    t1 = time.time()
    worst = 0.0
    def oracle_losses(img):
        with torch.no_grad():
            s_l, c_l = oracle(img)
        s_v, c_v = float(torch.stack(s_l).sum()), float(torch.stack(c_l).sum())
        return s_v, c_v, oc.style_w * s_v + oc.content_w * c_v
    for step, img in sorted(images.items()):
        s_ref, c_ref, t_ref = oracle_losses(img)
        got = (history["style_loss"][step - 1], history["content_loss"][step - 1], history["total_loss"][step - 1])
        tol_k = [ltol] * 3
        # each weighted term relative to itself - or, once the optimisation has made it a small part of the total
        # (the style score is a squared DIFFERENCE of nearly equal Grams by then, and bf16 rounding flips move it
        # by percents of itself), within `floor` of the total
        floor = (2e-3 if bf16 else 1e-6) * abs(t_ref)
        for i, (nm, wgt, a, b) in enumerate((("style", oc.style_w, got[0], s_ref), ("content", oc.content_w, got[1], c_ref), ("total", 1.0, got[2], t_ref))):
            rel = abs(a - b) / abs(b)
            if wgt * abs(a - b) > floor:
                worst = max(worst, rel)
                assert rel <= tol_k[i], f"{case} step {step}: {nm} loss {a!r} vs oracle at the same image {b!r} (tolerance {tol_k[i]:.1e})"
            elif nm == "total":
                worst = max(worst, rel)
    record_parity(case, f"losses vs oracle at the same image, {len(images)} steps (rel)", worst, ltol,
                  ("oracle given the model's (bf16) targets; " if bf16 else "") +
                  f"steps {sorted(images)}; loss {totals[0]:.3e} -> {totals[-1]:.3e}; {steps / wall:.0f} steps/s incl. test "
                  f"callbacks; oracle {time.time() - t1:.0f} s")

    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, f"r03_loss_curve_{size}_{precision}.csv"), "w") as fh:
            fh.write("step,style_loss,content_loss,total_loss\n")
            for k in range(steps):
                fh.write(f"{k + 1},{history['style_loss'][k]!r},{history['content_loss'][k]!r},{history['total_loss'][k]!r}\n")
    del model, x, opt, runner
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_a_run_is_bit_reproducible(precision, monkeypatch):
    """Same seed, same inputs, two fresh models: the image after 120 L-BFGS steps (history full, ring wrapped) and
    every logged loss must be BIT-identical.  Nothing on the path sums in a timing-dependent order: split-K slabs
    and per-wave partial sums are reduced in fixed order, no float atomics, and the tile of every conv shape comes
    from the persisted table (style_transfer_visualizer_amd/conv_tiles_gfx950.json), not from a measurement."""
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    monkeypatch.delenv("STV_CONV_TUNE", raising=False)          # the product's default: tiles from the table
    size, steps = 512, 120
    outs = []
    for _ in range(2):
        cfg = stv_config.StyleTransferConfig.model_validate({})
        oc = cfg.optimization
        oc.steps, oc.init_method = steps, "random"
        cfg.hardware.precision = precision
        cfg.video.create_video = False
        torch.manual_seed(0)
        content = synthetic.synthetic_image(0, size, size).to(DEV)
        style = synthetic.synthetic_image(1, size, size).to(DEV)
        model, x, opt = core_model.prepare_model_and_input(content, style, DEV, oc, precision=precision)
        out, hist, _ = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=_Bar()).run()
        outs.append((out.detach().cpu().clone(), hist["total_loss"], opt.device_state()))
        del model, x, opt
        torch.cuda.empty_cache()
    assert torch.equal(outs[0][0], outs[1][0]), "final images differ between two identical runs"
    assert outs[0][1] == outs[1][1], "loss histories differ between two identical runs"
    assert outs[0][2] == outs[1][2]

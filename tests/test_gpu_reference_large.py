"""The HIP path against the two LARGE runs of the unmodified reference (tests/golden, oracle/make_golden.py):

* ``cfg0_256_content_lbfgs50`` - BASELINE.json configs[0] as the reference itself runs it: 256x256, ``--init content``,
  50 L-BFGS steps, default layers / weights, ``log_every`` 10 (reference cli.py:317-343 -> main.py:72-132 ->
  optimization.py:162-202 driving torch.optim.LBFGS); inputs as its image loader returns them for the 8-bit PNGs;
* ``vgg19_128_random_lbfgs12`` - VGG19 width at 128x128 from the reference's default start (random), 12 L-BFGS steps,
  the image after every step.

Three kinds of rows, none of which runs a CPU trajectory on the GPU box (the oracle is pinned to these fixtures bit for
bit by tests/test_oracle_golden.py in the build container):

* integers, bit for bit: step ids, closure count, logged steps, and the optimizer's (n_iter, history length) after every
  step as torch.optim.LBFGS itself held them in the reference run;
* CHAOS-FREE, against the reference's own numbers: the fixtures hold FULL images from inside the reference's trajectory
  (after step 49 of configs[0]; after steps 5 and 11 of the 128^2 run) together with what the reference computed there
  (the next step's loss triple and gradient).  The HIP model evaluated AT THOSE IMAGES must give those losses to
  north_star's 1e-4 outright and that gradient to 2e-4 of scale (a ReLU / max-pool near-tie decided the other way is the
  one legitimate excuse, and is checked: few decisions, float64 near-ties, and the oracle with the HIP decisions imposed
  reproduces the HIP gradient);
* the FREE-RUNNING trajectory (L-BFGS without a line search amplifies rounding): every step is held to
  max(1e-4, 4x the spread the reference arithmetic itself shows under gradient noise of the size by which the HIP gradient
  measurably differs from the reference's at step 1) - ``GoldenCase.spread_level``: the fixture holds the spreads for
  3e-7 (two ulps), 3e-6, 3e-5 and 3e-4 (a few ReLU / pool near-ties decided the other way).  Where the HIP gradient is
  the reference's up to two ulps, the steps the reference reproduces to 2.5e-5 are thereby held to 1e-4 outright.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import config as stv_config
from style_transfer_visualizer_amd import core_model, optimization
from tests import parity_util as pu
from tests.conftest import LARGE_CASES, GoldenCase, record_parity

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
PIXEL_TOL_CAP = 5e-2          # as tests/test_gpu_model.py: beyond this an image row is a ceiling, not a rounding-level claim


class _Bar:
    def update(self, n=1):
        return None

    def set_postfix(self, *a, **k):
        return None

    def close(self):
        return None


def _gradient_row(case: GoldenCase, model, name: str, tag: str, g_hip: torch.Tensor, g_ref_sub: np.ndarray, gscale: float,
                  x_eval: torch.Tensor) -> float:
    """HIP gradient vs the reference's (subsampled) at one image: 2e-4 of scale, or - across a near-tie decided the
    other way - the same against the oracle with the HIP decisions imposed.  Returns the relative rms deviation."""
    m, k = case.meta, case.meta["compact"]
    sub = g_hip.numpy()[..., ::k, ::k]
    dev = float(np.abs(sub - g_ref_sub).max() / gscale)
    rms = float(np.sqrt(np.mean(np.square((sub - g_ref_sub).astype(np.float64)))) / np.sqrt(np.mean(np.square(g_ref_sub.astype(np.float64)))))
    if dev <= 2e-4:
        record_parity(name, f"{tag} gradient vs reference (of scale, subsampled)", dev, 2e-4, f"relative rms {rms:.1e}")
        return rms
    nl = pu.n_program_layers(m["style_layers"], m["content_layers"])
    content, style = case.images()
    oracle = ocm.OracleModel(ocm.vgg_program(case.weights(), case.cfg), m["style_layers"], m["content_layers"])
    oracle.set_targets(style, content)
    d_hip, d_cpu = pu.hip_decisions(model), pu.oracle_decisions(oracle.program, x_eval, nl)
    flips = pu.count_flips(d_hip, d_cpu)
    prog64 = ocm.vgg_program([(w.double(), b.double()) for w, b in case.weights()], case.cfg)
    gap = pu.flip_gaps(d_hip, d_cpu, prog64, x_eval.double(), nl)
    g_locked = ocm.loss_and_grad(pu.lock(oracle, d_hip), x_eval, m["style_w"], m["content_w"])[3]
    dev_locked = float((g_hip - g_locked).abs().max() / gscale)
    assert 0 < flips <= 64 and gap < 1e-5, f"{name} {tag}: {flips} decisions differ, largest float64 gap {gap:.2e}"
    note = f"plain comparison {dev:.1e} (subsampled): {flips} ReLU/pool decision(s) differ, float64 gap <= {gap:.1e} of the layer rms"
    if dev_locked <= 2e-4:
        record_parity(name, f"{tag} gradient vs reference arithmetic on the HIP path's branch (of scale)", dev_locked, 2e-4, note)
        return rms
    # Two fp32 evaluations on the SAME branch further apart than 2e-4 of scale: at a nearly converged image the style
    # gradient is proportional to G - T, a small difference of fp32 Gram sums - neither fp32 path is accurate to 2e-4
    # there.  The yardstick is float64 on that branch (as tests/test_gpu_fullsize.py): the HIP gradient may be as far
    # from it as the reference arithmetic is (x4).
    oracle64 = ocm.OracleModel(prog64, m["style_layers"], m["content_layers"])
    oracle64.set_targets(style.double(), content.double())
    g64 = ocm.loss_and_grad(pu.lock(oracle64, d_hip), x_eval.double(), m["style_w"], m["content_w"])[3]
    err_cpu = float((g_locked.double() - g64).abs().max() / gscale)
    err_hip = float((g_hip.double() - g64).abs().max() / gscale)
    bound = max(2e-4, 4.0 * err_cpu)
    record_parity(name, f"{tag} gradient vs FLOAT64 on the HIP path's branch (of scale)", err_hip, bound,
                  note + f"; the reference arithmetic on that branch is {err_cpu:.1e} from float64 (the two fp32 paths: {dev_locked:.1e} apart)")
    assert err_hip <= bound, f"{name} {tag}: HIP {err_hip:.2e} from float64 on its branch, the reference arithmetic {err_cpu:.2e}"
    return rms


@pytest.mark.parametrize("name", LARGE_CASES)
def test_large_reference_trajectory(name, monkeypatch):
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    case = GoldenCase(name)
    m, k, steps = case.meta, case.meta["compact"], case.meta["steps"]
    monkeypatch.setattr(core_model, "initialize_vgg", lambda: core_model.build_vgg_features(case.weights(), case.cfg).eval())
    cfg = stv_config.StyleTransferConfig.model_validate({})
    oc = cfg.optimization
    oc.steps, oc.style_w, oc.content_w, oc.init_method = steps, m["style_w"], m["content_w"], m["init_method"]
    oc.style_layers, oc.content_layers, oc.normalize = list(m["style_layers"]), list(m["content_layers"]), m["normalize"]
    cfg.hardware.precision = "fp32"
    cfg.output.log_every = m["log_every"]
    cfg.video.create_video = False
    content, style = case.images()
    torch.manual_seed(0)
    model, x, opt = core_model.prepare_model_and_input(content.to(DEV), style.to(DEV), DEV, oc, precision="fp32")
    # the start image IS the reference's, bit for bit (random: the CPU generator's draw after the network's constructors)
    x0 = case.start_image()
    assert torch.equal(x.detach().cpu(), x0)

    # ---- targets against the reference's --------------------------------------------------------------------------------
    for i, t in enumerate(model.style_targets):
        got = t.cpu().numpy()
        if f"style_target_{i}" in case.arrays:
            ref = case.arrays[f"style_target_{i}"]
        else:
            got, ref = got[::16, ::16], case.arrays[f"style_target_{i}_sub16"]
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())
    for i, t in enumerate(model.content_targets):
        assert float(t.double().abs().sum().cpu()) == pytest.approx(float(case.arrays[f"content_target_{i}_abs_sum"]), rel=1e-4)

    seen, snaps_all, states, grads = [], [], [], {}

    def on_end(mt):
        seen.append((mt.step, mt.has_values))
        if "x_steps_sub" in case.arrays:
            snaps_all.append(x.detach().cpu().clone())
        st = opt.device_state()
        states.append((st["n_iter"], st["hist_len"]))
        if mt.step == 1:
            grads["g1"] = x.grad.detach().cpu().clone()
            grads["rms"] = _gradient_row(case, model, name, "step-1", grads["g1"], case.arrays["grad_step1_sub"],
                                         float(case.arrays["grad_step1_absmax"]), x0)
    runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=_Bar(),
                                             callbacks=optimization.OptimizationCallbacks(on_step_end=on_end))
    out, history, _ = runner.run()

    # ---- integers, bit for bit --------------------------------------------------------------------------------------
    assert [s for s, _ in seen] == list(range(1, steps + 1))
    assert [s for s, has in seen if has] == case.arrays["logged_steps"].tolist()
    assert runner._closure_calls == int(case.arrays["closure_calls"])
    assert len(history["total_loss"]) == steps
    assert states == [tuple(r) for r in case.arrays["lbfgs_state"].tolist()], "L-BFGS (n_iter, history length) differ from torch.optim.LBFGS's in the reference run"

    # ---- the free-running trajectory --------------------------------------------------------------------------------
    level = case.spread_level(grads["rms"])
    eps = float(case.arrays["sens_eps"][level])
    xtol, ltol = case.step_tolerances(level)
    _, ltol0 = case.step_tolerances(0)
    total_ref = np.abs(np.asarray(case.arrays["total_loss"]))
    bad = []
    for j, (key, wgt) in enumerate((("style", m["style_w"]), ("content", m["content_w"]), ("total", 1.0))):
        got, want = np.asarray(history[f"{key}_loss"]), np.asarray(case.arrays[f"{key}_loss"])
        rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
        # a weighted term that is numerically negligible in the total (the content loss right after a content start is
        # ~1e-9 against 7e-2) is compared against the total: 1e-6 of it, as tests/test_gpu_model.py does
        rel = np.where(wgt * np.abs(got - want) <= 1e-6 * total_ref, 0.0, rel)
        tol = ltol[:, j]
        i_w = int(np.argmax(rel / tol))
        record_parity(name, f"free-running {key} loss, {steps} steps, worst vs its tolerance (rel)", float(rel[i_w]), float(tol[i_w]),
                      f"step {i_w + 1}; per step max(1e-4, 4x the reference's own spread under gradient noise of {eps:.0e}: the HIP "
                      f"step-1 gradient differs from the reference's by {grads['rms']:.1e} rms)")
        if key == "total":
            n_out = int((ltol0[:, j] <= 1e-4).sum())
            n_ok = int(((ltol0[:, j] <= 1e-4) & (rel <= 1e-4)).sum())
            n_any = int((rel <= 1e-4).sum())
            # asserted only where the HIP gradient is the reference's up to two-ulp noise (level 0: no near-tie went the
            # other way at step 1); otherwise reported - the chaos-free rows below carry the 1e-4 claim
            record_parity(name, "free-running steps within north_star 1e-4 outright (count of misses among the reproducible steps)",
                          float(n_out - n_ok), 0.0 if level == 0 else float("nan"),
                          f"{n_ok} of the {n_out} steps the reference itself reproduces to 2.5e-5 under two-ulp gradient noise are within 1e-4 "
                          f"({n_any} of all {steps} steps are)")
            if level == 0 and n_ok < n_out:
                bad.append(f"{n_out - n_ok} reproducible steps beyond 1e-4")
        over = rel > tol
        if over.any():
            bad.append(f"{key} loss at steps {(np.nonzero(over)[0] + 1).tolist()}: worst {float((rel / tol).max()):.1f}x its tolerance")
    if "x_steps_sub" in case.arrays:
        snaps = list(enumerate(snaps_all))
        ref_sub, ref_absmax = list(case.arrays["x_steps_sub"]), list(case.arrays["x_steps_absmax"])
    else:
        snaps = [(steps - 1, out.detach().cpu())]
        ref_sub = [None] * (steps - 1) + [case.arrays["x_final_sub"]]
        ref_absmax = [None] * (steps - 1) + [float(case.arrays["x_final_absmax"])]
    for s_i, got in snaps:
        dev = float(np.abs(got.numpy()[..., ::k, ::k] - ref_sub[s_i]).max() / float(ref_absmax[s_i]))
        tol, sens = float(xtol[s_i]), float(case.arrays["x_steps_sensitivity"][level][s_i])
        if tol > PIXEL_TOL_CAP:
            tol, note = min(8.0 * sens, 1.0), f"ceiling only (8x the reference's own spread {sens:.1e} at noise {eps:.0e})"
        else:
            note = "meets north_star 1e-4 outright" if tol <= 1e-4 else f"reference's own spread {sens:.1e} at noise {eps:.0e}, x4"
        record_parity(name, f"free-running image after step {s_i + 1} per pixel (of range)", dev, tol, note)
        if not dev <= tol:
            bad.append(f"image after step {s_i + 1}: {dev:.2e} > {tol:.2e}")

    # ---- chaos-free: the HIP model at images from INSIDE the reference's trajectory -----------------------------------
    for k_full in m["full_steps"]:
        x_ref = torch.from_numpy(case.arrays[f"x_after_step_{k_full}"])
        with torch.no_grad():
            x.copy_(x_ref.to(DEV))
        s, c, t = model.loss_and_grad(x, m["style_w"], m["content_w"])
        got = (float(s), float(c), float(t))
        want = tuple(float(case.arrays[f"{key}_loss"][k_full]) for key in ("style", "content", "total"))   # step k_full + 1
        for key, a, b, wgt in zip(("style", "content", "total"), got, want, (m["style_w"], m["content_w"], 1.0), strict=True):
            rel = abs(a - b) / max(abs(b), 1e-30)
            if wgt * abs(a - b) <= 1e-6 * abs(want[2]):
                rel = min(rel, 1e-6)
            record_parity(name, f"{key} loss AT the reference's image after step {k_full} (rel)", rel, 1e-4,
                          f"the reference's own step-{k_full + 1} value: chaos-free, north_star 1e-4 outright")
            assert rel <= 1e-4, f"{name}: {key} loss at the reference's image after step {k_full}: {a!r} vs {b!r}"
        _gradient_row(case, model, name, f"step-{k_full + 1} (at the reference's image)", x.grad.detach().cpu().clone(),
                      case.arrays[f"grad_at_step_{k_full + 1}_sub"], float(case.arrays[f"grad_at_step_{k_full + 1}_absmax"]), x_ref)
    assert not bad, f"{name}: free-running trajectory beyond its per-step tolerances: {bad}"

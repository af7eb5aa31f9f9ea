"""The HIP path against the two LARGE runs of the unmodified reference (tests/golden, oracle/make_golden.py):

* ``cfg0_256_content_lbfgs50`` - BASELINE.json configs[0] as the reference itself runs it: 256x256, ``--init content``,
  50 L-BFGS steps, default layers / weights, ``log_every`` 10 (reference cli.py:317-343 -> main.py:72-132 ->
  optimization.py:162-202 driving torch.optim.LBFGS); inputs as its image loader returns them for the 8-bit PNGs;
* ``vgg19_128_random_lbfgs12`` - VGG19 width at 128x128 from the reference's default start (random), 12 L-BFGS steps,
  the image after every step.

Both runs are FREE-RUNNING trajectories of L-BFGS without a line search, so each step carries the tolerance the
reference's own arithmetic earns there: north_star's 1e-4 OUTRIGHT at every step whose measured spread under 2-ulp
gradient noise is <= 2.5e-5 (``loss_sensitivity`` / ``x_steps_sensitivity`` of the fixture), 4x that spread elsewhere
(``GoldenCase.step_tolerances``).  Integer state - step ids, closure count, logged steps, the optimizer's
(n_iter, history length) after every step as torch.optim.LBFGS itself held them - is compared bit for bit.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import core_model_ref as ocm
from oracle import optim_ref
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


def _compare(case: GoldenCase, history: dict, snaps: list, ref_hist: dict, ref_sub: list, ref_absmax: list, what: str):
    """Losses of every step and the images (subsampled as the fixture stores them) against one reference run.
    Returns (failures, parity rows)."""
    m, k = case.meta, case.meta["compact"]
    xtol, ltol = case.step_tolerances()
    steps = m["steps"]
    bad, rows = [], []
    total = np.abs(np.asarray(ref_hist["total_loss"]))
    outright = 0
    worst = {key: (0.0, 0.0) for key in ("style", "content", "total")}
    for j, (key, wgt) in enumerate((("style", m["style_w"]), ("content", m["content_w"]), ("total", 1.0))):
        got, want = np.asarray(history[f"{key}_loss"]), np.asarray(ref_hist[f"{key}_loss"])
        rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
        # a weighted term that is numerically negligible in the total (the content loss right after a content start is
        # ~1e-9 against 7e-2) is compared against the total: 1e-6 of it, as tests/test_gpu_model.py does
        small = wgt * np.abs(got - want) <= 1e-6 * total
        rel = np.where(small, 0.0, rel)
        over = rel > ltol[:, j]
        if key == "total":
            outright = int(((ltol[:, j] <= 1e-4) & ~over).sum())
        i_w = int(np.argmax(rel / ltol[:, j]))
        worst[key] = (float(rel[i_w]), float(ltol[i_w, j]))
        rows.append((m["name"], f"{key} loss, {steps} steps, worst vs its tolerance (rel){what}", float(rel[i_w]), float(ltol[i_w, j]),
                     f"step {i_w + 1}; tolerance per step = max(1e-4, 4x the reference's own spread there)"))
        if over.any():
            bad.append(f"{key} loss at steps {(np.nonzero(over)[0] + 1).tolist()}: {rel[over].max():.2e}")
    n_outright = int((ltol[:, 2] <= 1e-4).sum())
    rows.append((m["name"], f"steps held to north_star 1e-4 outright (count){what}", float(n_outright - outright), 0.0,
                 f"{outright} of {n_outright} such steps within 1e-4 (deviation = how many are not)"))
    for s_i, got in snaps:
        want, scale = ref_sub[s_i], float(ref_absmax[s_i])
        dev = float(np.abs(got.numpy()[..., ::k, ::k] - want).max() / scale)
        tol = float(xtol[s_i])
        sens = float(case.arrays["x_steps_sensitivity"][s_i])
        if tol > PIXEL_TOL_CAP:
            ceiling = min(8.0 * sens, 1.0)
            rows.append((m["name"], f"image after step {s_i + 1} per pixel (of range){what}", dev, ceiling,
                         f"ceiling only (8x the reference's own spread {sens:.1e})"))
            if not dev <= ceiling:
                bad.append(f"image after step {s_i + 1}: {dev:.2e} > ceiling {ceiling:.2e}")
            continue
        rows.append((m["name"], f"image after step {s_i + 1} per pixel (of range){what}", dev, tol,
                     "meets north_star 1e-4 outright" if tol <= 1e-4 else f"reference's own spread {sens:.1e} x4"))
        if not dev <= tol:
            bad.append(f"image after step {s_i + 1}: {dev:.2e} > {tol:.2e}")
    return bad, rows


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
    assert torch.equal(x.detach().cpu(), case.start_image())

    # ---- targets + the first evaluation against the reference -------------------------------------------------------
    for i, t in enumerate(model.style_targets):
        got = t.cpu().numpy()
        if f"style_target_{i}" in case.arrays:
            ref = case.arrays[f"style_target_{i}"]
        else:
            got, ref = got[::16, ::16], case.arrays[f"style_target_{i}_sub16"]
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())
    for i, t in enumerate(model.content_targets):
        assert float(t.double().abs().sum().cpu()) == pytest.approx(float(case.arrays[f"content_target_{i}_abs_sum"]), rel=1e-4)

    seen, snaps_all, states, decisions, grads = [], [], [], [], {}

    def on_end(mt):
        seen.append((mt.step, mt.has_values))
        snaps_all.append(x.detach().cpu().clone())
        st = opt.device_state()
        states.append((st["n_iter"], st["hist_len"]))
        decisions.append(pu.hip_decisions(model))
        if mt.step == 1:
            grads["g1"] = x.grad.detach().cpu().clone()
    runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=_Bar(),
                                             callbacks=optimization.OptimizationCallbacks(on_step_end=on_end))
    out, history, _ = runner.run()

    # ---- integers, bit for bit --------------------------------------------------------------------------------------
    assert [s for s, _ in seen] == list(range(1, steps + 1))
    assert [s for s, has in seen if has] == case.arrays["logged_steps"].tolist()
    assert runner._closure_calls == int(case.arrays["closure_calls"])
    assert len(history["total_loss"]) == steps
    assert states == [tuple(r) for r in case.arrays["lbfgs_state"].tolist()], "L-BFGS (n_iter, history length) differ from torch.optim.LBFGS's in the reference run"

    # ---- step-1 gradient -------------------------------------------------------------------------------------------
    g_ref, gscale = case.arrays["grad_step1_sub"], float(case.arrays["grad_step1_absmax"])
    gdev = float(np.abs(grads["g1"].numpy()[..., ::k, ::k] - g_ref).max() / gscale)
    record_parity(name, "step-1 gradient vs reference (of scale, subsampled)", gdev, 2e-4)

    # ---- the trajectory -------------------------------------------------------------------------------------------
    ref_hist = {key: case.arrays[key] for key in ("style_loss", "content_loss", "total_loss")}
    if "x_steps_sub" in case.arrays:
        snaps = list(enumerate(snaps_all))
        ref_sub, ref_absmax = list(case.arrays["x_steps_sub"]), list(case.arrays["x_steps_absmax"])
    else:
        snaps = [(steps - 1, out.detach().cpu())]
        ref_sub = [None] * (steps - 1) + [case.arrays["x_final_sub"]]
        ref_absmax = [None] * (steps - 1) + [float(case.arrays["x_final_absmax"])]
    bad, rows = _compare(case, history, snaps, ref_hist, ref_sub, ref_absmax, "")
    if not bad and gdev <= 2e-4:
        for row in rows:
            record_parity(*row)
        return

    # ---- Beyond tolerance of the stored trajectory: legitimate only where the two fp32 evaluations decide a ReLU /
    # max-pool near-tie differently (tests/parity_util.py).  Checked as tests/test_gpu_model.py does: the first differing
    # decisions must be few and genuine float64 near-ties, and the reference arithmetic REPLAYED with the HIP path's
    # decisions imposed at every step must reproduce this run within the same per-step tolerances.
    nl = pu.n_program_layers(m["style_layers"], m["content_layers"])
    oracle = ocm.OracleModel(ocm.vgg_program(case.weights(), case.cfg), m["style_layers"], m["content_layers"])
    oracle.set_targets(style, content)
    x0 = case.start_image()
    free = optim_ref.run_loop(lambda xx: ocm.loss_and_grad(oracle, xx, m["style_w"], m["content_w"]), x0, steps, keep_steps=True)
    evaluated = [x0] + free["x_steps"][:-1]
    first_flip = None
    for s_i in range(steps):
        d_ref = pu.oracle_decisions(oracle.program, evaluated[s_i], nl)
        flips = pu.count_flips(decisions[s_i], d_ref)
        if flips:
            prog64 = ocm.vgg_program([(w.double(), b.double()) for w, b in case.weights()], case.cfg)
            first_flip = (s_i + 1, flips, pu.flip_gaps(decisions[s_i], d_ref, prog64, evaluated[s_i].double(), nl))
            break
    assert first_flip is not None, f"{name}: {bad} (step-1 gradient {gdev:.1e}) - and no ReLU/pool decision differs from the reference's"
    step_f, flips, gap = first_flip
    assert flips <= 64 and gap < 1e-5, f"{name}: step {step_f}: {flips} decisions differ, float64 gap {gap:.2e} of the layer rms"
    calls = []

    def locked_eval(xx):
        calls.append(len(calls))
        return ocm.loss_and_grad(pu.lock(oracle, decisions[calls[-1]]), xx, m["style_w"], m["content_w"])
    replay = optim_ref.run_loop(locked_eval, x0, steps, keep_steps=True)
    rep_hist = {"total_loss": replay["history"]["total"], "style_loss": replay["history"]["style"], "content_loss": replay["history"]["content"]}
    rep_sub = [xs.numpy()[..., ::k, ::k] for xs in replay["x_steps"]]
    rep_absmax = [float(xs.abs().max()) for xs in replay["x_steps"]]
    note = f" [reference arithmetic on the HIP path's branch: first differing decision at step {step_f} ({flips}, float64 gap <= {gap:.1e})]"
    gdev_locked = float((grads["g1"] - replay["first_grad"]).abs().max() / gscale)
    record_parity(name, "step-1 gradient vs reference on the HIP path's branch (of scale)", gdev_locked, 2e-4, f"plain comparison {gdev:.1e}")
    for row in rows:                                     # the plain rows: reported with what they measured
        record_parity(row[0], row[1], row[2], float("nan"), row[4] + f"; plain comparison across the decision flip at step {step_f}: reported, the rows marked [branch] are compared")
    bad2, rows2 = _compare(case, history, snaps, rep_hist, rep_sub, rep_absmax, note)
    for row in rows2:
        record_parity(*row)
    assert gdev_locked <= 2e-4
    assert not bad2, f"{name}: differs from the reference arithmetic even on its own branch: {bad2} (plain comparison: {bad})"

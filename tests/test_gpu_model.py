"""End-to-end parity of the HIP path on MI355X against the golden vectors captured
from the unmodified reference (tests/golden, produced by oracle/make_golden.py)
and against the CPU oracle on the same seeded inputs.

north_star tolerance: 1e-4 relative per pixel (fp32 parity mode), bit-exact step
indices / history lengths / logging cadence.
"""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import config as stv_config
from style_transfer_visualizer_amd import core_model, optimization, optimizers, synthetic
from tests import parity_util as pu
from tests.conftest import GoldenCase, record_parity

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


class _Bar:
    def update(self, n=1):
        return None

    def set_postfix(self, *a, **k):
        return None

    def close(self):
        return None


def _build(case: GoldenCase, monkeypatch, precision="fp32"):
    m = case.meta
    weights = case.weights()
    monkeypatch.setattr(core_model, "initialize_vgg",
                        lambda: core_model.build_vgg_features(weights, case.cfg).eval())
    cfg = stv_config.StyleTransferConfig.model_validate({})
    oc = cfg.optimization
    oc.steps, oc.style_w, oc.content_w = m["steps"], m["style_w"], m["content_w"]
    oc.init_method = m["init_method"]
    oc.style_layers, oc.content_layers = list(m["style_layers"]), list(m["content_layers"])
    oc.normalize = m["normalize"]
    cfg.hardware.precision = precision
    cfg.output.log_every = 2
    cfg.video.create_video = False
    content, style = case.images()
    torch.manual_seed(0)        # as oracle/make_golden.py seeded the reference (runtime/device.py:31-42 does it for the CLI)
    model, input_img, opt = core_model.prepare_model_and_input(content.to(DEV), style.to(DEV), DEV, oc,
                                                               precision=precision)
    # Nothing is copied in: the start image must BE the reference's, bit for bit - also for init_method=random,
    # whose draw comes from the CPU generator after the network's module constructions consumed it
    # (reference core_model.py:66-100 on its --device cpu path).
    assert torch.equal(input_img.detach().cpu(), case.tensor("x0")), f"{m['init_method']} start image differs from the reference's"
    return cfg, model, input_img, opt


def test_targets_and_first_step_match_reference(golden_case: GoldenCase, monkeypatch):
    case = golden_case
    m = case.meta
    cfg, model, x, _ = _build(case, monkeypatch)
    assert len(model.vgg_blocks) == m["block_count"]
    assert model.style_ids == m["style_ids"] and model.content_ids == m["content_ids"]
    for i, t in enumerate(model.style_targets):
        got = t.cpu().numpy()
        if f"style_target_{i}" in case.arrays:
            ref = case.arrays[f"style_target_{i}"]
        else:
            got, ref = got[::16, ::16], case.arrays[f"style_target_{i}_sub16"]
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=2e-5 * np.abs(ref).max())
    for i, t in enumerate(model.content_targets):
        assert float(t.double().abs().sum().cpu()) == pytest.approx(
            float(case.arrays[f"content_target_{i}_abs_sum"]), rel=1e-4)

    g_ref = case.arrays["grad_step1"]
    gscale = np.abs(g_ref).max()
    name = m.get("name", "fixture")
    # fused path
    s, c, tot = model.loss_and_grad(x, m["style_w"], m["content_w"])
    assert float(s) == pytest.approx(case.arrays["style_loss"][0], rel=2e-4)
    assert float(c) == pytest.approx(case.arrays["content_loss"][0], rel=2e-4)
    assert float(tot) == pytest.approx(case.arrays["total_loss"][0], rel=2e-4)
    fused_grad = x.grad.clone()
    dev = float(np.abs(fused_grad.cpu().numpy() - g_ref).max() / gscale)
    if dev <= 2e-4:
        record_parity(name, "step-1 gradient vs reference (of scale)", dev, 2e-4)
    else:
        # Not rounding: the two fp32 evaluations sit on different sides of a ReLU / max-pool near-tie
        # (tests/parity_util.py).  That claim is checked, not assumed: the differing decisions must be few and
        # genuine float64 near-ties, and on the branch the HIP path took the oracle must reproduce its gradient.
        nl = pu.n_program_layers(m["style_layers"], m["content_layers"])
        content, style = case.images()
        oracle = ocm.OracleModel(ocm.vgg_program(case.weights(), case.cfg), m["style_layers"], m["content_layers"])
        oracle.set_targets(style, content)
        x0 = case.tensor("x0")
        d_hip, d_cpu = pu.hip_decisions(model), pu.oracle_decisions(oracle.program, x0, nl)
        flips = pu.count_flips(d_hip, d_cpu)
        prog64 = ocm.vgg_program([(w.double(), b.double()) for w, b in case.weights()], case.cfg)
        gap = pu.flip_gaps(d_hip, d_cpu, prog64, x0.double(), nl)
        _, _, _, g_locked = ocm.loss_and_grad(pu.lock(oracle, d_hip), x0, m["style_w"], m["content_w"])
        dev_locked = float((fused_grad.cpu() - g_locked).abs().max() / gscale)
        record_parity(name, "step-1 gradient vs reference on the HIP path's branch (of scale)", dev_locked, 2e-4,
                      f"plain comparison {dev:.1e}: {flips} ReLU/pool decision(s) differ, float64 gap <= {gap:.1e} of the layer rms")
        assert 0 < flips <= 16 and gap < 1e-5, f"{name}: {flips} decisions differ, largest float64 gap {gap:.2e}"
        assert dev_locked <= 2e-4, f"{name}: gradient differs by {dev_locked:.2e} of scale on the same branch"
        g_ref = g_locked.numpy()
    # autograd path: model(x) -> lists of 0-d tensors -> loss.backward()
    x.grad = None
    s_losses, c_losses = model(x)
    assert len(s_losses) == len(m["style_layers"]) and len(c_losses) == len(m["content_layers"])
    assert all(t.dim() == 0 for t in s_losses + c_losses)
    loss = m["style_w"] * torch.stack(s_losses).sum() + m["content_w"] * torch.stack(c_losses).sum()
    loss.backward()
    assert float(loss) == pytest.approx(case.arrays["total_loss"][0], rel=2e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g_ref, rtol=0, atol=2e-4 * gscale)
    np.testing.assert_allclose(x.grad.cpu().numpy(), fused_grad.cpu().numpy(), rtol=0, atol=1e-5 * gscale)


def test_trajectory_matches_reference(golden_case: GoldenCase, monkeypatch):
    case = golden_case
    m = case.meta
    cfg, model, x, opt = _build(case, monkeypatch)
    if m["optimizer"] == "adam":
        opt = optimizers.HipAdam([x], lr=m["adam_lr"])
    seen, snaps, decisions = [], [], []

    def on_end(mt):
        seen.append((mt.step, mt.has_values))
        snaps.append(x.detach().cpu().clone())
        decisions.append(pu.hip_decisions(model))         # of the evaluation this step's update was built from
    runner = optimization.OptimizationRunner(
        model, x, cfg, optimizer=opt, progress_bar=_Bar(),
        callbacks=optimization.OptimizationCallbacks(on_step_end=on_end))
    out, history, _ = runner.run()
    steps = m["steps"]
    # integer bookkeeping: bit-exact
    assert [s for s, _ in seen] == list(range(1, steps + 1))
    assert [s for s, has in seen if has] == case.arrays["logged_steps"].tolist()
    assert runner._closure_calls == int(case.arrays["closure_calls"])
    assert len(history["total_loss"]) == steps
    name = case.meta.get("name", request_name(case))
    # Tolerances.  Losses: steps 1-2 are evaluated at x0 and x0 - t*g1 (no curvature estimate involved yet), so
    # every fixture - chaotic or not - must match the reference there at the plain 1e-4; later steps 1e-3, or ten
    # times the fixture's per-pixel tolerance where its trajectory amplifies rounding noise, NEVER above 1e-2.
    # Images: north_star's 1e-4 per pixel, widened to 4x the spread the REFERENCE arithmetic shows for that image
    # under 2-ulp gradient noise (oracle/make_golden.py); an image whose tolerance would exceed 5e-2 of its range
    # says nothing and is reported, not compared.  Fixtures with a chaotic or overshooting trajectory store the
    # image after EVERY step, so they are pinned wherever the reference itself is reproducible (e.g. before and
    # AFTER the overshoot of mini_clamp_lbfgs / tiny_taps_lbfgs: 5.6e8 -> 8.3e29 -> 5.6e8).
    ltol = min(max(1e-3, 10 * case.pixel_tolerance()), LOSS_TOL_CAP)
    if "x_steps" in case.arrays:
        assert np.array_equal(case.arrays["x_steps"][-1], case.arrays["x_final"])
        images = [(f"image after step {k + 1} per pixel (of range)", snaps[k], case.arrays["x_steps"][k],
                   float(case.arrays["x_steps_sensitivity"][k])) for k in range(steps)]
    else:
        images = [("x_final per pixel (of range)", out.detach().cpu(), case.arrays["x_final"],
                   float(case.arrays["x_final_sensitivity"]))]

    def check(ref_hist: dict, ref_images: list, what: str) -> tuple[list[str], list[tuple]]:
        """Compare losses and images with one reference run; returns the failures (empty: all within tolerance)
        and the parity-table rows (recorded by the caller: a plain comparison that a decision flip explains is
        reported, not asserted)."""
        bad = []
        rows: list[tuple] = []

        def record_parity(*row):                         # (shadows the module-level function inside check)
            rows.append(row)
        ref_total = np.asarray(ref_hist["total_loss"])
        first = min(2, steps)
        dev_first = float(np.abs(np.asarray(history["total_loss"][:first]) / ref_total[:first] - 1.0).max())
        record_parity(name, f"total loss, first {first} steps (rel){what}", dev_first, 1e-4)
        if not dev_first <= 1e-4:
            bad.append(f"first-step losses {dev_first:.2e}")
        dev_loss = float(np.abs(np.asarray(history["total_loss"]) / ref_total - 1.0).max())
        record_parity(name, f"total loss, {steps} steps (rel){what}", dev_loss, ltol)
        if not dev_loss <= ltol:
            bad.append(f"losses {dev_loss:.2e} > {ltol:.1e}")
        # the two terms: relative, with an absolute floor of 1e-6 of the total for a weighted term that is
        # numerically negligible in it (the content loss of a content-initialised image is ~1e-5 of the total)
        for key, wgt in (("style_loss", m["style_w"]), ("content_loss", m["content_w"])):
            got, ref = wgt * np.asarray(history[key]), wgt * np.asarray(ref_hist[key])
            if not np.allclose(got, ref, rtol=ltol, atol=1e-6 * float(np.abs(ref_total).max())):
                bad.append(f"{key} differs")
        compared = 0
        for (tag, got, _, sens), ref in zip(images, ref_images, strict=True):
            scale = float(np.abs(ref).max())
            dev = float(np.abs(got.numpy() - ref).max() / scale)
            tol = max(1e-4, 4.0 * sens)
            if not np.isfinite(tol) or tol > PIXEL_TOL_CAP:
                # An image the reference itself does not reproduce to 5e-2 of its range under 2-ulp gradient noise pins
                # nothing at rounding level - but it is still bounded: 8x the reference's own spread, never more than the
                # range itself (a regression that moved it tenfold fails; measured 2-4x).  Not counted as "compared".
                ceiling = min(CEILING_SPREADS * sens, 1.0) if np.isfinite(sens) else 1.0
                record_parity(name, tag + what, dev, ceiling,
                              f"ceiling only ({CEILING_SPREADS:g}x the reference's own spread of {sens:.1e} of the range; x4 > {PIXEL_TOL_CAP:g}: no rounding-level claim)")
                if not dev <= ceiling:
                    bad.append(f"{tag}: {dev:.2e} > ceiling {ceiling:.2e}")
                continue
            compared += 1
            record_parity(name, tag + what, dev, tol,
                          "meets north_star 1e-4 outright" if tol <= 1e-4 else f"reference's own spread {sens:.1e} x4")
            if not dev <= tol:                   # (a NaN deviation is a failure)
                bad.append(f"{tag}: {dev:.2e} > {tol:.2e}")
        if compared == 0:
            bad.append("no image of this fixture could be compared")
        return bad, rows

    golden_hist = {k: case.arrays[k] for k in ("total_loss", "style_loss", "content_loss")}
    from tests.conftest import record_parity as emit
    bad, rows = check(golden_hist, [ref for _, _, ref, _ in images], "")
    if not bad:
        for row in rows:
            emit(*row)
        return
    # ---- Not within tolerance of the stored trajectory.  The one legitimate cause is a ReLU / max-pool near-tie
    # that the two fp32 evaluations decide differently (tests/parity_util.py) - from there on the trajectories
    # are on different branches of a piecewise-linear network and nothing bounds their distance.  Checked, not
    # assumed: (1) up to the first step where decisions differ the plain comparison must hold; (2) the decisions
    # that differ there must be few and genuine float64 near-ties; (3) the oracle REPLAYED with the HIP path's
    # decisions imposed at every step must reproduce this run's losses and images within the same tolerances.
    from oracle import optim_ref
    nl = pu.n_program_layers(m["style_layers"], m["content_layers"])
    content, style = case.images()
    oracle = ocm.OracleModel(ocm.vgg_program(case.weights(), case.cfg), m["style_layers"], m["content_layers"])
    oracle.set_targets(style, content)
    x0 = case.tensor("x0")
    kw = dict(optimizer=m["optimizer"], lr=m["adam_lr"] if m["optimizer"] == "adam" else None, keep_steps=True)
    free = optim_ref.run_loop(lambda xx: ocm.loss_and_grad(oracle, xx, m["style_w"], m["content_w"]), x0, steps, **kw)
    evaluated = [x0] + free["x_steps"][:-1]               # the image step k + 1 of the free-running oracle evaluates
    first_flip = None
    for k in range(steps):
        d_ref = pu.oracle_decisions(oracle.program, evaluated[k], nl)
        flips = pu.count_flips(decisions[k], d_ref)
        if flips:
            prog64 = ocm.vgg_program([(w.double(), b.double()) for w, b in case.weights()], case.cfg)
            gap = pu.flip_gaps(decisions[k], d_ref, prog64, evaluated[k].double(), nl)
            first_flip = (k + 1, flips, gap)
            break
    assert first_flip is not None, f"{name}: {bad} - and no ReLU/pool decision differs from the reference's"
    step_f, flips, gap = first_flip
    assert flips <= 16 and gap < 1e-5, f"{name}: step {step_f}: {flips} decisions differ, float64 gap {gap:.2e} of the layer rms"
    # (1): the loss is continuous across a near-tie, so up to and including the step of the first flip the stored
    # trajectory's losses must still be matched
    upto = np.asarray(history["total_loss"][:step_f]) / case.arrays["total_loss"][:step_f] - 1.0
    assert float(np.abs(upto).max()) <= ltol, f"{name}: losses differ from the reference before any decision does"
    calls = []

    def locked_eval(xx):
        k = len(calls)
        calls.append(k)
        return ocm.loss_and_grad(pu.lock(oracle, decisions[k]), xx, m["style_w"], m["content_w"])
    replay = optim_ref.run_loop(locked_eval, x0, steps, **kw)
    rep_hist = {"total_loss": replay["history"]["total"], "style_loss": replay["history"]["style"],
                "content_loss": replay["history"]["content"]}
    rep_images = [xs.numpy() for xs in replay["x_steps"]] if "x_steps" in case.arrays else [replay["x"].numpy()]
    note = (f" [reference replayed on the HIP path's branch: at step {step_f} {flips} ReLU/pool decision(s) differ, "
            f"float64 gap <= {gap:.1e} of the layer rms]")
    # The plain comparison across the flip is not a rounding-level claim (the replay rows below are), but it is BOUNDED:
    # the reference's own arithmetic, moved onto the HIP path's branch by the same flipped decisions (`replay`), departs
    # from the stored trajectory by some amount - this run may depart by at most twice that plus the row's tolerance.
    own: dict[str, float] = {}
    gold_total = np.asarray(golden_hist["total_loss"])
    rep_total = np.asarray(rep_hist["total_loss"])
    first = min(2, steps)
    own[f"total loss, first {first} steps (rel)"] = float(np.abs(rep_total[:first] / gold_total[:first] - 1.0).max())
    own[f"total loss, {steps} steps (rel)"] = float(np.abs(rep_total / gold_total - 1.0).max())
    for (tag, _, ref, _), rep in zip(images, rep_images, strict=True):
        own[tag] = float(np.abs(rep - ref).max() / float(np.abs(ref).max()))
    over_ceiling = []
    for row in rows:
        case_, qty, dev_, tol_, *rest = row
        if dev_ <= tol_:
            emit(*row)
            continue
        ceiling = 2.0 * own.get(qty, 0.0) + tol_
        emit(case_, qty, dev_, ceiling,
             f"across the decision flip at step {step_f} (beyond the plain {tol_:.1e}): ceiling = 2x the reference arithmetic's own "
             f"departure when the same decisions are flipped ({own.get(qty, float('nan')):.1e}) + the plain tolerance; the rounding-level rows are the replayed ones")
        if not dev_ <= ceiling:
            over_ceiling.append(f"{qty}: {dev_:.2e} > {ceiling:.2e}")
    assert not over_ceiling, f"{name}: beyond what the flipped decisions explain: {over_ceiling}"
    bad2, rows2 = check(rep_hist, rep_images, note)
    for row in rows2:
        emit(*row)
    assert not bad2, f"{name}: differs from the reference even on its own branch: {bad2} (plain comparison: {bad})"


LOSS_TOL_CAP = 1e-2
PIXEL_TOL_CAP = 5e-2
CEILING_SPREADS = 8.0      # ceiling of an image row that is beyond PIXEL_TOL_CAP: this many times the reference's own spread


def request_name(case: GoldenCase) -> str:
    m = case.meta
    return f"{m.get('net', 'net')}_{m['init_method']}_{m['optimizer']}_{m['hw_content'][0]}x{m['hw_content'][1]}"


def _decision_flips(model, oracle64, x64):
    """Count ReLU-sign / max-pool-argmax decisions where the HIP activations disagree with the fp64
    oracle, and the largest fp64 gap (relative to the layer's scale) among those positions."""
    import torch.nn.functional as F
    from style_transfer_visualizer_amd import ops
    eng = next(iter(model._engines.values()))
    acts = []
    h = x64
    last = max(n.layer for n in eng.sched.nodes) + 1
    for li in range(min(last + 1, len(oracle64.program))):
        h = ocm.run_layer(oracle64.program[li], h)
        acts.append(h)
    flips, worst = 0, 0.0
    for nd in eng.sched.nodes:
        # a conv output meets a ReLU either in its own epilogue (relu_fused) or, when it is tapped
        # pre-activation, in its consumer (ReLU-on-load forward, MASK on the way back)
        relu_later = any(n.src is nd.dst and (n.relu_in or n.kind == "relu") for n in eng.sched.nodes)
        if nd.kind in ("conv", "conv_first") and (nd.dst.relu_fused or relu_later):
            z = acts[nd.layer]                                  # fp64 pre-activation
            hip_on = ops.from_nhwc(nd.dst.act).cpu() > 0
            diff = hip_on != (z > 0)
            if diff.any():
                flips += int(diff.sum())
                worst = max(worst, float(z[diff].abs().max() / z.abs().max()))
        if nd.kind == "pool":
            src = acts[nd.layer - 1]
            a_hip = ops.from_nhwc(nd.src.act).cpu().double()
            v_ref, i_ref = F.max_pool2d(src, 2, 2, return_indices=True)
            _, i_hip = F.max_pool2d(a_hip, 2, 2, return_indices=True)
            diff = i_hip != i_ref
            if diff.any():
                flips += int(diff.sum())
                alt = src.flatten(2).gather(2, i_hip.flatten(2)).reshape(v_ref.shape)   # fp64 value HIP picked
                worst = max(worst, float((v_ref - alt)[diff].abs().max() / src.abs().max()))
    return flips, worst


def _locked_oracle(model, oracle64):
    """Copy of the fp64 oracle whose ReLU masks and max-pool argmaxes are the HIP path's."""
    import copy
    import torch.nn.functional as F
    from style_transfer_visualizer_amd import ops
    eng = next(iter(model._engines.values()))
    prog = list(oracle64.program)
    for nd in eng.sched.nodes:
        if nd.kind in ("conv", "conv_first") and nd.layer + 1 < len(prog) and prog[nd.layer + 1][0] == "relu":
            prog[nd.layer + 1] = ("relu_mask", ops.from_nhwc(nd.dst.act).cpu() > 0)    # fused or not: act > 0 <=> z > 0
        elif nd.kind == "relu":
            prog[nd.layer] = ("relu_mask", ops.from_nhwc(nd.dst.act).cpu() > 0)
        elif nd.kind == "pool":
            _, idx = F.max_pool2d(ops.from_nhwc(nd.src.act).cpu().double(), 2, 2, return_indices=True)
            prog[nd.layer] = ("pool_idx", idx)
    locked = copy.copy(oracle64)
    locked.program = prog
    return locked


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("name", ["mini_content_lbfgs", "mini_random_lbfgs_nonorm", "vgg19_content_lbfgs"])
def test_every_step_matches_oracle_at_same_image(name, compact, monkeypatch):
    """Chaos-free parity: at every step of a HIP-driven run, evaluate the CPU oracle at the SAME
    image and compare losses and gradient (fp32 rounding level), and compare the device L-BFGS
    update with the oracle optimizer fed the same gradients."""
    from oracle import optim_ref
    from style_transfer_visualizer_amd import ops
    case = GoldenCase(name)
    m = case.meta
    cfg, model, x, _ = _build(case, monkeypatch)
    oracle = ocm.OracleModel(ocm.vgg_program(case.weights(), case.cfg), m["style_layers"], m["content_layers"])
    content, style = case.images()
    oracle.set_targets(style, content)
    # fp64 evaluation of the same algorithm: the yardstick both fp32 paths are measured against
    w64 = [(w.double(), b.double()) for w, b in case.weights()]
    oracle64 = ocm.OracleModel(ocm.vgg_program(w64, case.cfg), m["style_layers"], m["content_layers"])
    oracle64.set_targets(style.double(), content.double())
    x_twin = x.detach().cpu().clone()                      # oracle optimizer fed HIP gradients
    twin = optim_ref.LbfgsRef(x_twin.view(-1), lr=1.0)
    state, work = ops.lbfgs_alloc(x.numel(), 100, DEV, compact=compact)
    for step in range(m["steps"]):
        s, c, t = model.loss_and_grad(x, m["style_w"], m["content_w"])
        xc = x.detach().cpu()
        so, co, to, go = ocm.loss_and_grad(oracle, xc, m["style_w"], m["content_w"])
        _, _, t64, g64 = ocm.loss_and_grad(oracle64, xc.double(), m["style_w"], m["content_w"])
        # each term to 1e-5 relative, with an absolute floor of 1e-6 of the total for terms that
        # are pure rounding noise (content loss of a content-initialised image is ~0)
        floor = 1e-6 * abs(float(to))
        assert float(t) == pytest.approx(float(to), rel=1e-5)
        assert m["style_w"] * float(s) == pytest.approx(m["style_w"] * float(so), rel=1e-5, abs=floor)
        assert m["content_w"] * float(c) == pytest.approx(m["content_w"] * float(co), rel=1e-5, abs=floor)
        assert abs(float(t) - float(t64)) <= max(3 * abs(float(to) - float(t64)), 1e-5 * abs(float(t64)))
        # gradient: the HIP fp32 path must be as close to the fp64 truth as the reference's own
        # fp32 CPU path is (x3), with a floor at fp32 rounding level.  A max-pool / ReLU decision
        # that a 1-ulp activation difference flips moves BOTH fp32 paths away from fp64, so this
        # criterion does not depend on which side of such a discontinuity a path lands.
        g = x.grad.detach().cpu()
        err_hip = float((g.double() - g64).norm() / g64.norm())
        err_cpu = float((go.double() - g64).norm() / g64.norm())
        flips, worst_gap = _decision_flips(model, oracle64, xc.double())
        if flips == 0:
            assert err_hip <= max(3 * err_cpu, 2e-5), f"step {step + 1}: HIP {err_hip:.2e} vs CPU-fp32 {err_cpu:.2e}"
        else:
            # every differing ReLU / max-pool decision must be a genuine near-tie of the fp64 values
            assert worst_gap < 1e-4, f"step {step + 1}: decision differs at a gap of {worst_gap:.2e} of the layer scale"
            assert flips <= 8
            # on the branch of the network the HIP path actually took (its ReLU masks and pool
            # argmaxes imposed on the fp64 oracle) the gradient must again be fp32-accurate
            _, _, _, g64_locked = ocm.loss_and_grad(_locked_oracle(model, oracle64), xc.double(), m["style_w"],
                                                    m["content_w"])
            err_locked = float((g.double() - g64_locked).norm() / g64_locked.norm())
            assert err_locked <= max(3 * err_cpu, 2e-5), \
                f"step {step + 1}: HIP {err_locked:.2e} on its own branch vs CPU-fp32 {err_cpu:.2e}"
        twin.step(lambda: (t.cpu(), g))
        ops.lbfgs_step(x.detach(), x.grad, state, work, 100, min(step, 100), 1.0, compact=compact)
        drift = float((x.detach().cpu() - x_twin).abs().max() / x_twin.abs().max())
        assert drift < 5e-6, f"device L-BFGS left the oracle optimizer at step {step + 1}: {drift:.2e}"


def test_bf16_storage_tracks_fp32(monkeypatch):
    """bf16 activation storage (perf mode): losses within bf16 rounding of the fp32 reference."""
    case = GoldenCase("vgg19_white_lbfgs")
    m = case.meta
    cfg, model, x, _ = _build(case, monkeypatch, precision="bf16")
    s, c, tot = model.loss_and_grad(x, m["style_w"], m["content_w"])
    assert float(tot) == pytest.approx(case.arrays["total_loss"][0], rel=5e-2)
    g_ref = case.arrays["grad_step1"]
    err = np.abs(x.grad.cpu().numpy() - g_ref).max() / np.abs(g_ref).max()
    assert err < 0.1, f"bf16 gradient drifted {err:.3f} of scale"


def test_first_layer_gram_fusion_is_equivalent(monkeypatch):
    """bf16 mode, conv1_1 tapped: the first layer leaving its own Gram slabs (stv_conv_first_fwd_gram, forced
    on at this small size) against the separate Gram pass - same targets and losses up to the fp32 summation
    order of the slabs, same gradient up to what a last-bit change of the bf16 seed S moves."""
    case = GoldenCase("vgg19_white_lbfgs")
    m = case.meta
    out = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("STV_FUSE_GRAM_FIRST", mode)
        _, model, x, _ = _build(case, monkeypatch, precision="bf16")
        s, c, tot = model.loss_and_grad(x, m["style_w"], m["content_w"])
        first = next(iter(model._engines.values())).sched.style_taps[0]
        assert first.partials_fused == (mode == "2")
        out[mode] = (float(s), float(c), x.grad.clone(), model.style_targets[0].clone())
    assert torch.allclose(out["0"][3], out["2"][3], rtol=0, atol=2e-6 * float(out["0"][3].abs().max()))
    assert out["2"][0] == pytest.approx(out["0"][0], rel=2e-5)
    assert out["2"][1] == out["0"][1]                                   # the content term does not see the Gram chain
    err = float((out["2"][2] - out["0"][2]).abs().max() / out["0"][2].abs().max())
    record_parity("vgg19_white_lbfgs bf16", "gradient, fused first-layer Gram vs separate pass (of scale)", err, 2e-2)
    assert err < 2e-2


def test_late_gram_finish_of_the_large_taps_changes_nothing(monkeypatch):
    """1024x1024, bf16: conv1_1 (134 MB) and conv2_1 (67 MB) are too large to wait for the batched Gram launch, so
    their partial-sum passes stay behind their producers - but their FINISH passes (a few MB of fp32 slabs) join the
    batched finish at the end of the forward pass (round 4).  Against STV_GRAM_FIN_LATE=0 (a finish launch of its own
    right behind each): two launches fewer, the same sums in another fixed order."""
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    size = 1024
    content = synthetic.synthetic_image(0, size, size).to(DEV)
    style = synthetic.synthetic_image(1, size, size).to(DEV)
    x0 = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(0)).to(DEV)
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("STV_GRAM_FIN_LATE", mode)
        model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21], precision="bf16").to(DEV)
        model.set_targets(style, content)
        x = x0.clone().requires_grad_(True)
        scores = tuple(float(v) for v in model.loss_and_grad(x, 1e5, 1.0))
        eng = next(iter(model._engines.values()))
        prog = next(p for k, p in eng._programs.items() if k[0] == "fused")
        seeds = [t.sgrad.detach().clone() for t in eng.sched.style_taps]
        out[mode] = (scores, x.grad.detach().clone(), prog.n_ops, seeds, [t.detach().clone() for t in model.style_targets])
        del model, x
        torch.cuda.empty_cache()
    assert out["1"][2] == out["0"][2] - 2, (out["0"][2], out["1"][2])
    for a, b in zip(out["0"][4], out["1"][4], strict=True):               # the targets do not go through this path
        assert torch.equal(a, b)
    # The batched finish adds a tap's split-K slabs in groups of 8, the stand-alone launch of a many-slab tap in groups
    # of 32 (gram.hip): the same fp32 sums in another fixed order.  Scores to fp32 rounding; a Gram seed S (bf16) may
    # differ in its last bit where the sum sat on a rounding boundary, and the gradient by what such a bit moves
    # (the bound of test_first_layer_gram_fusion_is_equivalent).
    for a, b in zip(out["0"][0], out["1"][0], strict=True):
        assert b == pytest.approx(a, rel=2e-6)
    for k, (a, b) in enumerate(zip(out["0"][3], out["1"][3], strict=True)):
        d = (a.float() - b.float()).abs()
        assert float((d / (2.0 ** -7 * torch.maximum(a.float().abs(), b.float().abs()) + 1e-30)).max()) <= 1.0, f"seed of tap {k}"
        assert float((d > 0).float().mean()) <= 2e-2
    err = float((out["1"][1] - out["0"][1]).abs().max() / out["0"][1].abs().max())
    record_parity("vgg19_1024x1024_bf16", "gradient, Gram finish of the large taps batched vs stand-alone (of scale)", err, 2e-2)
    assert err <= 2e-2


def test_oracle_agreement_at_larger_size(monkeypatch):
    """Same seeded inputs through oracle (CPU) and HIP at 160x128 with the mini net."""
    case = GoldenCase("mini_white_lbfgs")
    weights = case.weights()
    monkeypatch.setattr(core_model, "initialize_vgg",
                        lambda: core_model.build_vgg_features(weights, case.cfg).eval())
    content = synthetic.synthetic_image(4, 160, 128)
    style = synthetic.synthetic_image(5, 96, 200)
    S, C = case.meta["style_layers"], case.meta["content_layers"]
    prog = ocm.vgg_program(weights, case.cfg)
    oracle = ocm.OracleModel(prog, S, C)
    oracle.set_targets(style, content)
    x0 = synthetic.synthetic_image(6, 160, 128)
    s_ref, c_ref, t_ref, g_ref = ocm.loss_and_grad(oracle, x0, 1e5, 1.0)
    model = core_model.StyleContentModel(S, C).to(DEV)
    model.set_targets(style.to(DEV), content.to(DEV))
    x = x0.to(DEV).requires_grad_(True)
    s, c, t = model.loss_and_grad(x, 1e5, 1.0)
    assert float(t) == pytest.approx(float(t_ref), rel=2e-4)
    assert float(s) == pytest.approx(float(s_ref), rel=2e-4)
    assert float(c) == pytest.approx(float(c_ref), rel=2e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g_ref.numpy(), rtol=0, atol=2e-4 * float(g_ref.abs().max()))


def test_model_errors_match_reference_texts(monkeypatch):
    case = GoldenCase("tiny_taps_lbfgs")
    weights = case.weights()
    monkeypatch.setattr(core_model, "initialize_vgg",
                        lambda: core_model.build_vgg_features(weights, case.cfg).eval())
    model = core_model.StyleContentModel([0], [1]).to(DEV)
    x = torch.randn(1, 3, 16, 16, device=DEV)
    model.content_targets = [x]
    with pytest.raises(RuntimeError, match="style_targets must be set"):
        model(x)
    model.style_targets, model.content_targets = [x], None
    with pytest.raises(RuntimeError, match="content_targets must be set"):
        model(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        core_model.StyleContentModel([0], [1]).set_targets(x.cpu(), x.cpu())


def test_cli_end_to_end_writes_png(tmp_path, monkeypatch):
    """BASELINE.json configs[0] plumbing through the CLI on the GPU: PNG in, PNG out, frame counts."""
    from PIL import Image

    from style_transfer_visualizer_amd import cli
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    for name, seed in (("content", 0), ("style", 1)):
        img = synthetic.synthetic_image(seed, 256, 256, normalize=False)[0].permute(1, 2, 0).mul(255).byte().numpy()
        Image.fromarray(img).save(tmp_path / f"{name}.png")
    out_dir = tmp_path / "out"
    csv_path = tmp_path / "loss.csv"
    cli.main(["--content", str(tmp_path / "content.png"), "--style", str(tmp_path / "style.png"), "--steps", "12",
              "--init", "random", "--device", "cuda", "--no-video", "--final-only", "--seed", "0", "--output",
              str(out_dir), "--log-loss", str(csv_path), "--log-every", "4"])
    png = out_dir / "stylized_content_x_style.png"
    assert png.is_file()
    assert Image.open(png).size == (256, 256)
    rows = csv_path.read_text().strip().splitlines()
    assert rows[0] == "step,style_loss,content_loss,total_loss"
    assert [r.split(",")[0] for r in rows[1:]] == ["4", "8", "12"]
    totals = [float(r.split(",")[3]) for r in rows[1:]]
    assert all(np.isfinite(totals)) and totals[-1] < totals[0]


def test_odd_sizes_match_oracle(monkeypatch):
    """Non-multiple-of-tile image sizes (floor pooling, ragged tiles, differing style size)."""
    case = GoldenCase("mini_white_lbfgs")
    weights = case.weights()
    monkeypatch.setattr(core_model, "initialize_vgg",
                        lambda: core_model.build_vgg_features(weights, case.cfg).eval())
    S, C = case.meta["style_layers"], case.meta["content_layers"]
    content = synthetic.synthetic_image(7, 75, 101)
    style = synthetic.synthetic_image(8, 50, 67)
    x0 = synthetic.synthetic_image(9, 75, 101)
    # fp32 mode against the reference arithmetic; bf16 storage against the oracle with the same
    # storage rounding emulated (bf16 rounding moves this small net's gradient by ~20 % rms on the
    # CPU too - the comparison separates that from kernel errors)
    for precision, ltol, gtol in (("fp32", 2e-4, 2e-4), ("bf16", 3e-3, 1e-1)):
        oracle = ocm.OracleModel(ocm.vgg_program(weights, case.cfg), S, C, bf16_storage=precision == "bf16")
        oracle.set_targets(style, content)
        s_ref, c_ref, t_ref, g_ref = ocm.loss_and_grad(oracle, x0, 1e5, 1.0)
        model = core_model.StyleContentModel(S, C, precision=precision).to(DEV)
        model.set_targets(style.to(DEV), content.to(DEV))
        x = x0.to(DEV).requires_grad_(True)
        s, c, t = model.loss_and_grad(x, 1e5, 1.0)
        g = x.grad.cpu()
        err = float((g - g_ref).abs().max() / g_ref.abs().max()) if precision == "fp32" else \
            float((g - g_ref).norm() / g_ref.norm())
        print(f"odd-size {precision}: loss rel {abs(float(t) - float(t_ref)) / float(t_ref):.2e} grad err {err:.2e}")
        assert float(t) == pytest.approx(float(t_ref), rel=ltol)
        assert err < gtol, f"{precision}: gradient error {err:.2e}"


def test_4k_image_runs(monkeypatch):
    """BASELINE configs[4] size on ONE GPU (bf16 storage): exercises >1 GiB activations and the
    32-bit offset guards; three Adam steps must run and reduce the loss."""
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    from style_transfer_visualizer_amd import optimizers as opt
    H, W = 2160, 3840
    content = synthetic.synthetic_image(0, H, W).to(DEV)
    style = synthetic.synthetic_image(1, 1024, 1024).to(DEV)
    model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21], precision="bf16").to(DEV)
    model.set_targets(style, content)
    x = torch.randn(1, 3, H, W, device=DEV).requires_grad_(True)
    adam = opt.HipAdam([x], lr=1e-2)
    totals = []
    for _ in range(3):
        totals.append(adam.step(lambda: model.loss_and_grad(x, 1e5, 1.0)[2]))
    vals = [float(t) for t in totals]
    assert all(np.isfinite(vals)) and vals[-1] < vals[0]
    assert torch.isfinite(x).all()
    del model, x, adam
    torch.cuda.empty_cache()


# (The whole 3840x2160 image in fp32 against the CPU oracle - losses 1e-5, gradient incl. the last rows at the largest
#  32-bit byte offsets - is part of tests/test_gpu_configs.py::test_configs4_200_adam_steps since round 4: there the
#  image is the one the four-strip run holds after 100 Adam steps, and the strips are checked against it too.)

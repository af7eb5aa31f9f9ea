"""Parity at the REAL sizes of BASELINE.json configs[1] (512x512) and configs[2] (1024x1024):
full-width VGG19, the benchmark's synthetic weights/images (bench.py seeds), default layers and
weights - what only exists at full size (the tile configurations picked there, Gram split-K over
2.6e5..1e6 pixels, the batched Gram chain's 48-MiB decision, fused conv+pool on 1024-wide rows).

Per size and precision: the five Gram targets, then one closure + three L-BFGS steps with the CPU
oracle re-evaluated AT THE SAME IMAGE (chaos-free: reference optimization.py:286-327,
core_model.py:297-328) - at the first and the last evaluation - and at every step
the device L-BFGS update against the oracle optimizer fed the same gradients.  Float64 evaluations (5 s at 512^2,
20-25 s at 1024^2 each) are spent where they carry information.  At the START image the float64 gradient is a property of
the oracle and the seeded inputs alone: it is computed once (oracle/make_fullsize_ref.py -> tests/golden/
fullsize_fp64_<size>.npz: the gradient subsampled, and the oracle's own fp32 distance from it) and gives the PLAIN fp64
row at both sizes for both tile tables without any float64 time on the box.  Live float64 goes to the 512^2 run's last
evaluation (an image L-BFGS has moved): plain row + the same-branch comparison of both paths.  Wherever no float64
gradient exists for an evaluation, the HIP gradient is compared PER PIXEL with the fp32 CPU oracle given the HIP path's own
ReLU / pool decisions (``pu.lock``: measured 7.6e-7 of scale, bound 2e-5) - a row that decision near-ties cannot
loosen - instead of a rel-rms bound that includes both paths' near-ties (round 4's 5e-3).  fp32 = the parity mode (reference arithmetic); bf16 = the measured mode, against the
oracle that rounds to bf16 exactly where the kernels do (oracle/core_model_ref.py).

Tolerances (measured values are printed in the parity table at the end of the run):
* fp32: losses 1e-5 relative.  Gradient: a Gram entry is an fp32 sum over 2.6e5..1e6 pixels
  (relative error ~sqrt(N)*eps = 3e-5..6e-5 whatever the summation order) and the style gradient
  is proportional to G - T, where those errors no longer cancel: two CORRECT fp32 evaluations -
  the reference's own CPU path with 1 and with 16 threads, or MKL and these kernels - differ by
  ~3e-4 of the gradient's scale at these sizes (measured: HIP vs CPU-fp32 2.7e-4 rms at 512^2).
  The yardstick is therefore the same algorithm in float64: the HIP gradient must be as close
  to it as the reference's fp32 CPU path is - at full size in the two-part form below (same branch: x2, floor
  1e-6; plain rows: x4, floor GRAD_FLOOR = 2.5e-3); tests/test_gpu_model.py::test_every_step_matches_oracle_at_same_image
  applies the plain criterion at fixture sizes.
  MFMA K-loops add the 9*Cin products of an output one after the other, oneDNN's kernels in 16
  SIMD lanes.  Round 3: the fp32 kernels sum every K-stage in a fresh accumulator (blocked summation,
  csrc/conv_igemm.hip) and every stored activation / gradient of the HIP path is now CLOSER to float64
  than the CPU path's (tests/diag/diag_three_way.py: e.g. 4.2e-7 vs 5.1e-7 at conv5_1).  What is left in
  the plain "vs fp64" rows (1e-3 for BOTH paths) is not rounding: with 3e7 activations per evaluation a
  few ReLU / max-pool decisions are float64 near-ties that any fp32 evaluation may take either way, and
  each moves the gradient by O(1e-2) of scale inside that unit's receptive field (tests/parity_util.py).
  So the criterion has two parts: (a) on the SAME branch - float64 oracle with the path's own decisions
  imposed - the HIP gradient must be at least as close to float64 as the CPU path's is on ITS branch
  (x2, floor 1e-6), and given the HIP decisions the fp32 CPU oracle must agree with the HIP gradient per
  pixel to 1e-3 of scale (measured ~1e-5); (b) the plain rows stay as a sanity bound: 4x the CPU path's
  own, floor 2.5e-3.
* bf16: losses 2e-3 relative (measured ~1e-4).  The gradient is compared with the oracle that
  rounds to bf16 at the same points, but only as a sanity bound (rms 0.25 of its rms; measured
  0.11-0.13): bf16 storage makes the network chaotic under rounding - see
  tests/test_gpu_bf16_layerwise.py, which holds every stored tensor of the same run to one bf16 ulp
  against the oracle op on identical inputs, and test_oracle_golden.py for the measurement on the
  reference arithmetic itself.
"""
from __future__ import annotations

import time

import numpy as np
import pytest
import torch

from oracle import core_model_ref as ocm
from oracle import optim_ref
from style_transfer_visualizer_amd import _lib, core_model, ops, synthetic
from tests import parity_util as pu
from tests.conftest import record_parity

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")
S_LAYERS, C_LAYERS = [0, 5, 10, 19, 28], [21]
STYLE_W, CONTENT_W = 1e5, 1.0
# Both fp32 paths sit 1.5e-4 .. 3.8e-3 (rms) from the float64 gradient at these sizes, and the CPU path's own
# error swings by 10x from one image to the next (1.5e-4 .. 1.4e-3 at 512^2 within one run): a pure ratio
# bound would hinge on the luckiest CPU sample.  Anything a wrong tile edge, tap or mask would cause is
# far above this floor; rounding-level effects are below it.
GRAD_FLOOR = 2.5e-3
LOCKED_PIXEL_TOL = 2e-5      # HIP vs CPU-fp32 given the HIP decisions, per pixel, of scale (measured 7.6e-7 at 512^2)


def _fp64_cache(size: int, x0: torch.Tensor) -> dict:
    """tests/golden/fullsize_fp64_<size>.npz (oracle/make_fullsize_ref.py), checked to describe THIS start image."""
    import os

    from tests.conftest import GOLDEN_DIR
    d = np.load(os.path.join(GOLDEN_DIR, f"fullsize_fp64_{size}.npz"))
    assert float(x0.double().abs().sum()) == pytest.approx(float(d["x0_abs_sum"]), rel=1e-12)
    assert float(x0.double().sum()) == pytest.approx(float(d["x0_sum"]), rel=1e-9, abs=1e-6)
    return {k: d[k] for k in d.files}


def _fused_style_taps(model) -> list[int]:
    """Orders of the style taps whose Gram-backward term the plan fused into a dgrad launch."""
    eng = next(iter(model._engines.values()))
    prog = next(p for k, p in eng._programs.items() if k[0] == "fused")
    dual = {(H, W, cout) for (op, H, W, cin, cout, taps, n) in prog.op_meta
            if op == _lib.OP_CONV and taps == 9 and n > 0}
    return [t.order for t in eng.sched.style_taps if (t.buf.H, t.buf.W, t.buf.C) in dual]


def _grad_stats(g: torch.Tensor, g_ref: torch.Tensor, bound: float) -> tuple[float, float, float]:
    scale = float(g_ref.abs().max())
    err = (g - g_ref).abs() / scale
    return float(err.max()), float(err.pow(2).mean().sqrt()), float((err > bound).float().mean())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("size", [512, 1024])
def test_fullsize_closure_and_lbfgs_steps_match_oracle(size, precision, monkeypatch):
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    case = f"vgg19_{size}x{size}_{precision}"
    weights = synthetic.synthetic_conv_weights(0)
    content = synthetic.synthetic_image(0, size, size)
    style = synthetic.synthetic_image(1, size, size)
    x0 = torch.randn(content.shape, generator=torch.Generator().manual_seed(0))      # init_method=random, as bench.py

    # ---- HIP path: pinned tiles (STV_CONV_TUNE=0, the conftest default) and measured tiles --------
    def hip_model(tune: str):
        monkeypatch.setenv("STV_CONV_TUNE", tune)
        model = core_model.StyleContentModel(S_LAYERS, C_LAYERS, precision=precision).to(DEV)
        model.set_targets(style.to(DEV), content.to(DEV))
        x = x0.to(DEV).clone().requires_grad_(True)
        s, c, t = model.loss_and_grad(x, STYLE_W, CONTENT_W)
        return model, x, (float(s), float(c), float(t)), x.grad.detach().cpu().clone()
    model, x, l_pin, g_pin = hip_model("0")
    dec0 = pu.hip_decisions(model) if precision == "fp32" else None
    fused = _fused_style_taps(model)
    model_t, x_t, l_tun, g_tun = hip_model("1")
    tiles = {"pinned": [], "tuned": []}
    monkeypatch.setenv("STV_CONV_TUNE", "0")
    eng = next(iter(model._engines.values()))
    dcode = ops.dtype_code(eng.dtype)
    for nd in eng.sched.nodes:
        if nd.kind == "conv":
            tiles["pinned"].append(int(_lib.load().stv_conv_config(nd.dst.H, nd.dst.W, nd.cin, nd.dst.C, 9, dcode)))
    monkeypatch.setenv("STV_CONV_TUNE", "1")
    for nd in eng.sched.nodes:
        if nd.kind == "conv":
            tiles["tuned"].append(int(_lib.load().stv_conv_config(nd.dst.H, nd.dst.W, nd.cin, nd.dst.C, 9, dcode)))
    print(f"{case}: forward tiles pinned {tiles['pinned']} tuned {tiles['tuned']}; fused style taps {fused}")

    # ---- oracle -------------------------------------------------------------------------------------
    t0 = time.time()
    bf16 = precision == "bf16"
    oracle = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), S_LAYERS, C_LAYERS,
                             bf16_storage=bf16, fused_style_taps=fused if bf16 else None)
    oracle.set_targets(style, content)
    ltol, grms_tol = (1e-5, None) if not bf16 else (2e-3, 0.25)
    oracle64 = None
    if not bf16 and size <= 512:        # LIVE float64 evaluation of the same algorithm (512^2 only; the start image's is cached)
        w64 = [(w.double(), b.double()) for w, b in weights]
        oracle64 = ocm.OracleModel(ocm.vgg_program(w64, synthetic.VGG19_CFG), S_LAYERS, C_LAYERS)
        oracle64.set_targets(style.double(), content.double())
    for i, (tg, to) in enumerate(zip(model.style_targets, oracle.style_targets, strict=True)):
        dev = float((tg.cpu() - to).abs().max() / to.abs().max())
        record_parity(case, f"Gram target {i} (of max)", dev, 2e-4 if not bf16 else 2e-3)
        assert dev <= (2e-4 if not bf16 else 2e-3)
    ct, co = model.content_targets[0].float().cpu(), oracle.content_targets[0]
    assert ct.shape == co.shape
    dev = float((ct - co).abs().max() / co.abs().max())
    record_parity(case, "content target (of max)", dev, 2e-5 if not bf16 else 8e-3)
    assert dev <= (2e-5 if not bf16 else 8e-3)

    nl = pu.n_program_layers(S_LAYERS, C_LAYERS)

    def check_same_branch(tag: str, g, g_cpu, xc, dec_hip, cpu_branch: bool = True) -> None:
        """(a) of the module docstring: accuracy with the discrete decisions held fixed.  ``cpu_branch=False`` (1024^2)
        skips the float64 evaluation on the CPU path's branch - 25 s that only produce the bound - and holds the HIP path
        to 1e-6, 2.5x what the CPU path measured on its own branch there in round 3 (4.0e-7) and at 512^2 in this run."""
        dec_cpu = pu.oracle_decisions(oracle.program, xc, nl)
        flips = pu.count_flips(dec_hip, dec_cpu)
        x64 = xc.double()
        g64_h = ocm.loss_and_grad(pu.lock(oracle64, dec_hip), x64, STYLE_W, CONTENT_W)[3]
        err_hip = float((g.double() - g64_h).norm() / g64_h.norm())
        if cpu_branch:
            g64_c = ocm.loss_and_grad(pu.lock(oracle64, dec_cpu), x64, STYLE_W, CONTENT_W)[3]
            err_cpu = float((g_cpu.double() - g64_c).norm() / g64_c.norm())
            bound, note = max(2 * err_cpu, 1e-6), f"reference's CPU-fp32 path on its own branch: {err_cpu:.2e}; "
        else:
            bound, note = 1e-6, "fixed bound (the CPU path on its own branch: 4.0e-7 here in round 3); "
        record_parity(case, f"{tag} grad vs fp64 on the same branch (rel rms)", err_hip, bound,
                      note + f"{flips} ReLU/pool decisions differ between the two fp32 paths")
        assert err_hip <= bound, f"{case} {tag}: HIP {err_hip:.2e} on its own branch vs float64 (bound {bound:.1e})"
        if not cpu_branch:
            return                      # (the per-pixel fp32-vs-fp32 row below: at 512^2 only, 8 s of host time at 1024^2)
        g32_h = ocm.loss_and_grad(pu.lock(oracle, dec_hip), xc, STYLE_W, CONTENT_W)[3]
        mx = float((g - g32_h).abs().max() / g32_h.abs().max())
        record_parity(case, f"{tag} grad HIP vs CPU-fp32 given the HIP decisions, per pixel max (of scale)", mx, 1e-3)
        assert mx <= 1e-3

    def locked_pixel_row(tag: str, g, xc, dec_hip) -> None:
        """The fp32 CPU oracle evaluated with the HIP path's ReLU / pool decisions imposed must reproduce the HIP
        gradient per pixel: no near-tie of either path is left in this row."""
        g32_h = ocm.loss_and_grad(pu.lock(oracle, dec_hip), xc, STYLE_W, CONTENT_W)[3]
        mx = float((g - g32_h).abs().max() / g32_h.abs().max())
        record_parity(case, f"{tag} grad HIP vs CPU-fp32 given the HIP decisions, per pixel max (of scale)", mx, LOCKED_PIXEL_TOL)
        assert mx <= LOCKED_PIXEL_TOL, f"{case} {tag}: {mx:.2e} of scale with the decisions held fixed"

    def check(tag: str, losses, g, ref, g64=None, cached=None, dec_hip=None, xc=None) -> None:
        """``ref`` = (style, content, total[, gradient]) of the oracle at this image.  ``g64``: a live float64 gradient;
        ``cached``: the start image's float64 gradient from tests/golden (subsampled); neither: the locked per-pixel row
        (needs ``dec_hip`` - the decisions of THIS evaluation - and the image ``xc``)."""
        s_ref, c_ref, t_ref = ref[:3]
        g_ref = ref[3] if len(ref) > 3 else None
        for nm, got, want in (("style", losses[0], float(s_ref)), ("content", losses[1], float(c_ref)),
                              ("total", losses[2], float(t_ref))):
            rel = abs(got - want) / abs(want)
            record_parity(case, f"{tag} {nm} loss (rel)", rel, ltol)
            assert rel <= ltol, f"{case} {tag}: {nm} loss {got!r} vs oracle {want!r}"
        if not bf16 and cached is not None:
            k = int(cached["sub"])
            g64s = torch.from_numpy(cached["g64_sub"]).double()
            err_hip = float((g.double()[..., ::k, ::k] - g64s).norm() / g64s.norm())
            err_cpu = float(cached["err_cpu_sub"])
            record_parity(case, f"{tag} grad vs fp64 (rel rms, every {k}th row and column)", err_hip, max(4 * err_cpu, GRAD_FLOOR),
                          f"decision near-ties included; float64 gradient and the reference's CPU-fp32 distance from it ({err_cpu:.2e}; "
                          f"whole image {float(cached['err_cpu_full']):.2e}) from tests/golden/fullsize_fp64_{size}.npz")
            assert err_hip <= max(4 * err_cpu, GRAD_FLOOR), f"{case} {tag}: HIP {err_hip:.2e} vs fp64, CPU-fp32 {err_cpu:.2e}"
        elif not bf16 and g64 is not None:
            err_hip = float((g.double() - g64).norm() / g64.norm())
            err_cpu = float((g_ref.double() - g64).norm() / g64.norm())
            mx, rms, _ = _grad_stats(g, g_ref, 2e-4)
            record_parity(case, f"{tag} grad vs fp64 (rel rms)", err_hip, max(4 * err_cpu, GRAD_FLOOR),
                          f"decision near-ties included; reference's CPU-fp32 path vs fp64: {err_cpu:.2e}; HIP vs CPU-fp32 directly: rms {rms:.1e} max {mx:.1e} of scale")
            assert err_hip <= max(4 * err_cpu, GRAD_FLOOR), f"{case} {tag}: HIP {err_hip:.2e} vs fp64, CPU-fp32 {err_cpu:.2e}"
        elif not bf16 and dec_hip is not None:      # no float64 gradient for this evaluation: decisions held fixed, per pixel
            locked_pixel_row(tag, g, xc, dec_hip() if callable(dec_hip) else dec_hip)
        elif not bf16:          # reported, not compared: both paths' own near-tie decisions are in this number (the gates
            # are the cached float64 rows and the locked per-pixel row at step 1, and the same-branch rows at the last step)
            rel = float((g - g_ref).norm() / g_ref.norm())
            record_parity(case, f"{tag} grad HIP vs CPU-fp32 (rel rms)", rel, float("nan"), "reported only: decision near-ties of BOTH paths included")
        else:
            rel = float((g - g_ref).norm() / g_ref.norm())
            record_parity(case, f"{tag} grad rms (of rms)", rel, grms_tol, "sanity bound only: rounding chaos, see test_gpu_bf16_layerwise.py")
            assert rel <= grms_tol, f"{case} {tag}: bf16 gradient differs from the rounding-faithful oracle by {rel:.2e}"

    def g64_at(xc):
        return None if oracle64 is None else ocm.loss_and_grad(oracle64, xc.double(), STYLE_W, CONTENT_W)[3]
    if bf16:
        ref0 = ocm.loss_and_grad(oracle, x0, STYLE_W, CONTENT_W)
        check("step1 pinned-tiles", l_pin, g_pin, ref0)
        check("step1 tuned-tiles", l_tun, g_tun, ref0)
    else:
        # the fp32 oracle's losses and the float64 gradient at the start image: computed once, tests/golden
        cache = _fp64_cache(size, x0)
        check("step1 pinned-tiles", l_pin, g_pin, tuple(cache["losses_fp32"]), cached=cache)
        check("step1 tuned-tiles", l_tun, g_tun, tuple(cache["losses_fp32"]), cached=cache)
        # `model` was evaluated at x0 before the tuned twin was built: dec0 are the decisions of that evaluation
        locked_pixel_row("step1", g_pin, x0, dec0)
    del model_t, x_t

    # ---- three L-BFGS steps: oracle at the same image, oracle optimizer fed the HIP gradients --------
    # The update is built from fp32 dot products over 0.8M / 3.1M elements, and from the second pair on
    # from DIFFERENCES of such products (y.r = H (y.q) - al (y.s)): two correct fp32 evaluations -
    # torch's CPU dot and the device's wave-partial sums - differ by 1e-5..1e-4 of the image range
    # there.  Yardstick as for the gradient: the same update in float64; the device must be as close
    # to it as the reference's fp32 optimizer is (x4, floor 5e-6).
    x_twin = x.detach().cpu().clone()
    twin = optim_ref.LbfgsRef(x_twin.view(-1), lr=1.0)
    x_twin64 = x.detach().cpu().double()
    twin64 = optim_ref.LbfgsRef(x_twin64.view(-1), lr=1.0)
    state, work = ops.lbfgs_alloc(x.numel(), 100, DEV, compact=True)
    losses, g = l_pin, g_pin
    for step in range(1, 4):
        t_dev = torch.tensor(losses[2])
        twin.step(lambda: (t_dev, g))
        twin64.step(lambda: (t_dev.double(), g.double()))
        ops.lbfgs_step(x.detach(), x.grad, state, work, 100, min(step - 1, 100), 1.0, compact=True)
        scale = float(x_twin64.abs().max())
        err_dev = float((x.detach().cpu().double() - x_twin64).abs().max()) / scale
        err_cpu = float((x_twin.double() - x_twin64).abs().max()) / scale
        record_parity(case, f"L-BFGS update {step} vs float64 update", err_dev, max(4 * err_cpu, 5e-6),
                      f"reference's fp32 optimizer vs float64: {err_cpu:.1e} (of the image range)")
        assert err_dev <= max(4 * err_cpu, 5e-6)
        s, c, t = model.loss_and_grad(x, STYLE_W, CONTENT_W)
        losses, g = (float(s), float(c), float(t)), x.grad.detach().cpu().clone()
        # the CPU oracle is re-evaluated at the LAST step (an image three L-BFGS updates away from the start); the L-BFGS
        # update itself is checked at every step (round 5: the 512^2 runs no longer evaluate it at steps 2 and 3 as well -
        # 6-10 s of host time per run for rows the last step and tests/test_gpu_configs.py repeat)
        if step == 3 and not (bf16 and size > 512):     # (1024^2 bf16 at a moved image: tests/test_gpu_configs.py, 500 steps, oracle at the final image)
            xc = x.detach().cpu()
            ref_k = ocm.loss_and_grad(oracle, xc, STYLE_W, CONTENT_W)
            # float64: plain rows at the first and the last evaluation of the 512^2 run; losses against the fp32 oracle everywhere
            check(f"step{step + 1}", losses, g, ref_k, g64_at(xc) if (size <= 512 and step == 3) else None)
            if not bf16 and step == 3 and size <= 512:       # same branch, both paths, at an image L-BFGS has moved
                check_same_branch(f"step{step + 1}", g, ref_k[3], xc, pu.hip_decisions(model))
    print(f"{case}: oracle time {time.time() - t0:.0f} s")
    del model, x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("n_pixels,C", [(2 ** 18, 64), (2 ** 18, 128), (2 ** 18, 512), (2 ** 20, 64), (2 ** 20, 128),
                                        (2 ** 20, 256)])
def test_gram_chain_at_full_pixel_counts(n_pixels, C, precision):
    """stv_gram_partial / stv_gram_finish called directly at the pixel counts of the 512^2 / 1024^2
    nets (split-K over 2^18 and 2^20 pixels), clamp engaged, vs the oracle's gram_matrix / MSE /
    seed arithmetic in float64 (the 137-GFLOP float64 product F^T F itself is evaluated with torch on the
    device - an independent rocBLAS dgemm - to keep the suite's wall time down; everything after it is
    the oracle's formulae)."""
    dtype = torch.float32 if precision == "fp32" else torch.bfloat16
    g = torch.Generator().manual_seed(n_pixels // 1024 + C)
    side = int(n_pixels ** 0.5)
    feat = (torch.randn(n_pixels, C, generator=g) * 0.7 + 0.1).to(dtype)        # NHWC rows = pixels
    f64 = feat.to(DEV).double()
    raw = (f64.t() @ f64).cpu()
    del f64
    clamp = float(raw.diagonal().median())                                       # engages on part of the diagonal
    norm = float(C * n_pixels)
    gram_ref = raw.clamp(max=clamp) / norm
    target = (gram_ref * 0.9).float()
    fd = feat.to(DEV).reshape(side, side, C)
    partials = ops.gram_partial(fd)
    gram = torch.empty(C, C, device=DEV)
    parts = torch.zeros(ops.gram_loss_parts(C), device=DEV)
    seed = torch.empty(C, C, device=DEV, dtype=dtype)
    ops.gram_finish(partials, n_pixels, C, target=target.to(DEV), gram_out=gram, loss_part=parts, sgrad=seed,
                    clamp_max=clamp, coef=STYLE_W, dtype=dtype)
    case = f"gram n={n_pixels} C={C} {precision}"
    tol = 2e-5
    dev = float((gram.cpu().double() - gram_ref).abs().max() / gram_ref.abs().max())
    record_parity(case, "G (of max)", dev, tol)
    assert dev <= tol
    loss = float(parts.double().sum().cpu()) / (C * C)
    loss_ref = float(((gram_ref - target.double()) ** 2).mean())
    record_parity(case, "mse loss (rel)", abs(loss - loss_ref) / loss_ref, 1e-4)
    assert loss == pytest.approx(loss_ref, rel=1e-4)
    seed_ref = STYLE_W * 4.0 / (C * C * norm) * (raw <= clamp) * (gram_ref - target.double())
    # elements whose raw sum is within rounding of the clamp may fall on either side of it
    near = (raw - clamp).abs() <= 1e-5 * clamp
    sdev = ((seed.cpu().double() - seed_ref).abs() / seed_ref.abs().max()).masked_fill(near, 0.0)
    stol = 1e-5 if precision == "fp32" else 2.0 ** -8
    record_parity(case, "seed S (of max)", float(sdev.max()), stol)
    assert float(sdev.max()) <= stol


@pytest.mark.parametrize("size", [512, 1024])
def test_gradient_is_the_derivative_of_the_loss_at_full_size(size, monkeypatch):
    """A property that needs no oracle: the gradient the backward kernels write must be the derivative of the loss
    the forward kernels compute.  Central differences of the fp32 loss along three directions v (step chosen so
    that the loss moves by ~1e-3 of itself: truncation error ~1e-6, rounding noise of the two fp32 losses ~1e-4
    of the difference) against g.v, at the sizes of BASELINE configs[1] / configs[2]; three directions, one of
    them confined to the image border (the zero-padding rows and columns of every layer)."""
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    case = f"vgg19_{size}x{size}_fp32 derivative"
    content = synthetic.synthetic_image(0, size, size).to(DEV)
    style = synthetic.synthetic_image(1, size, size).to(DEV)
    model = core_model.StyleContentModel(S_LAYERS, C_LAYERS, precision="fp32").to(DEV)
    model.set_targets(style, content)
    x = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(3)).to(DEV).requires_grad_(True)
    _, _, t0 = model.loss_and_grad(x, STYLE_W, CONTENT_W)
    f0 = float(t0)
    g = x.grad.detach().clone().double()
    gen = torch.Generator().manual_seed(11)
    for k in range(3):
        # directions with a sizeable component along the gradient (a random direction in 3e6 dimensions has
        # g.v ~ |g| / 1,700: the step that moves the loss by 1e-3 would leave the linear regime): the gradient
        # itself, the gradient plus an equally long random vector, and the gradient on the border pixels only
        r = torch.randn(x.shape, generator=gen).to(DEV).double()
        v = g / g.norm()
        if k == 1:
            v = v + r / r.norm()
        if k == 2:
            mask = torch.zeros_like(v)
            mask[..., :2, :] = 1; mask[..., -2:, :] = 1; mask[..., :, :2] = 1; mask[..., :, -2:] = 1
            v = v * mask
        v = v / v.norm()
        gv = float((g * v).sum())
        # loss moves by ~1e-3 of itself (border direction: 2e-4 - the same loss change concentrated on 0.4 % of
        # the pixels would push each of them across many ReLU kinks; its tolerance allows for the fp32 loss noise)
        move, tol = (1e-3, 2e-3) if k < 2 else (2e-4, 1.5e-2)
        eps = move * abs(f0) / max(abs(gv), 1e-30)
        fs = []
        for sgn in (1.0, -1.0):
            xp = (x.detach().double() + sgn * eps * v).float().requires_grad_(True)
            fs.append(float(model.loss_and_grad(xp, STYLE_W, CONTENT_W)[2]))
        fd = (fs[0] - fs[1]) / (2 * eps)
        rel = abs(fd - gv) / abs(gv)
        record_parity(case, f"central difference vs g.v, direction {k} (rel)", rel, tol,
                      ("along the gradient", "gradient + random", "gradient on the border pixels only")[k])
        assert rel <= tol, f"{case} direction {k}: finite difference {fd!r} vs g.v {gv!r}"
    del model, x
    torch.cuda.empty_cache()

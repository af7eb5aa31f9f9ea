"""Pin the CPU oracle against golden vectors captured from the unmodified reference."""
from __future__ import annotations

import os

import numpy as np
import pytest
import torch

from oracle import core_model_ref as ocm
from oracle import optim_ref
from tests.conftest import GOLDEN_DIR, LARGE_CASES, GoldenCase

# The oracle uses the same torch CPU kernels in the same order as the reference,
# so agreement is expected at rounding level; this is the pin tolerance.
RTOL = 1e-5


def _oracle_model(case: GoldenCase) -> ocm.OracleModel:
    prog = ocm.vgg_program(case.weights(), case.cfg)
    model = ocm.OracleModel(prog, case.meta["style_layers"], case.meta["content_layers"])
    content, style = case.images()
    model.set_targets(style, content)
    return model


def test_gram_known_answers():
    kats = np.load(os.path.join(GOLDEN_DIR, "gram_kats.npz"))
    a = torch.from_numpy(kats["kat1_in"])
    # values observed from the reference's gram_matrix
    assert ocm.gram_matrix(a).tolist() == [[1.75, 4.75], [4.75, 15.75]]
    assert ocm.gram_matrix(a, clamp_max=30).tolist() == [[1.75, 3.75], [3.75, 3.75]]
    assert np.array_equal(ocm.gram_matrix(a).numpy(), kats["kat1_out"])
    b = torch.from_numpy(kats["kat3_in"])
    g3 = ocm.gram_matrix(b)
    assert g3.shape == (4, 4)
    assert g3[0].tolist() == [0.875, 2.375, 3.875, 5.375]
    assert np.array_equal(g3.numpy(), kats["kat3_out"])
    x = a.clone().requires_grad_(True)
    ocm.gram_matrix(x, clamp_max=30).sum().backward()
    assert x.grad.flatten().tolist() == [0, .25, .5, .75, 0, 0, 0, 0]
    f = torch.from_numpy(kats["kat5_in"]).requires_grad_(True)
    g5 = ocm.gram_matrix(f, clamp_max=20.0)
    assert np.array_equal(g5.detach().numpy(), kats["kat5_out_clamp20"])
    loss = torch.nn.functional.mse_loss(g5, torch.from_numpy(kats["kat5_target"]))
    loss.backward()
    assert loss.item() == pytest.approx(float(kats["kat5_loss"]), rel=1e-6)
    np.testing.assert_allclose(f.grad.numpy(), kats["kat5_grad"], rtol=1e-6, atol=1e-7)


def test_gram_symmetric_psd():
    # mirrors /root/reference/tests/test_core_model.py:84-92
    t = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    g = ocm.gram_matrix(t)
    assert g.shape == (3, 3)
    assert torch.allclose(g, g.t())
    assert torch.all(torch.linalg.eigvals(g).real >= -1e-6)


def test_block_split_defaults():
    # 6 blocks, style ids [0,1,2,3,5], content ids [4]
    blocks, content_ids, style_ids = ocm.split_blocks(37, [0, 5, 10, 19, 28], [21])
    assert [b[0] for b in blocks] == [0, 1, 6, 11, 20, 22]
    assert [b[-1] for b in blocks] == [0, 5, 10, 19, 21, 28]
    assert style_ids == [0, 1, 2, 3, 5]
    assert content_ids == [4]


def test_oracle_matches_reference_fixture(golden_case: GoldenCase):
    case = golden_case
    m = case.meta
    model = _oracle_model(case)
    assert len(model.blocks) == m["block_count"]
    assert model.style_ids == m["style_ids"]
    assert model.content_ids == m["content_ids"]

    # targets
    for i, t in enumerate(model.style_targets):
        if f"style_target_{i}" in case.arrays:
            np.testing.assert_allclose(t.numpy(), case.arrays[f"style_target_{i}"], rtol=RTOL, atol=1e-7)
        else:
            np.testing.assert_allclose(t.numpy()[::16, ::16], case.arrays[f"style_target_{i}_sub16"],
                                       rtol=RTOL, atol=1e-7)
            assert float(t.double().sum()) == pytest.approx(float(case.arrays[f"style_target_{i}_sum"]), rel=1e-5)
    for i, t in enumerate(model.content_targets):
        assert float(t.double().abs().sum()) == pytest.approx(
            float(case.arrays[f"content_target_{i}_abs_sum"]), rel=1e-5)

    # step-1 loss triple and gradient at x0
    x0 = case.tensor("x0")
    s, c, tot, g = ocm.loss_and_grad(model, x0, m["style_w"], m["content_w"])
    assert float(s) == pytest.approx(case.arrays["style_loss"][0], rel=RTOL)
    assert float(c) == pytest.approx(case.arrays["content_loss"][0], rel=RTOL)
    assert float(tot) == pytest.approx(case.arrays["total_loss"][0], rel=RTOL)
    g_ref = case.arrays["grad_step1"]
    np.testing.assert_allclose(g.numpy(), g_ref, rtol=1e-4, atol=1e-6 * np.abs(g_ref).max())

    # full trajectory with the restated optimizer
    res = optim_ref.run_loop(
        lambda x: ocm.loss_and_grad(model, x, m["style_w"], m["content_w"]),
        x0, m["steps"], optimizer=m["optimizer"],
        lr=m["adam_lr"] if m["optimizer"] == "adam" else None)
    np.testing.assert_allclose(res["history"]["total"], case.arrays["total_loss"], rtol=1e-4)
    np.testing.assert_allclose(res["history"]["style"], case.arrays["style_loss"], rtol=1e-4)
    np.testing.assert_allclose(res["history"]["content"], case.arrays["content_loss"], rtol=1e-4)
    xf = case.arrays["x_final"]
    # north_star tolerance: 1e-4 relative per pixel (relative to the image scale)
    np.testing.assert_allclose(res["x"].numpy(), xf, rtol=1e-4, atol=1e-4 * np.abs(xf).max())
    assert int(case.arrays["closure_calls"]) == m["steps"]  # 1 closure per step (F5)
    assert case.arrays["logged_steps"].tolist() == [s for s in range(1, m["steps"] + 1) if s % 2 == 0]


@pytest.mark.parametrize("name", LARGE_CASES)
def test_oracle_matches_large_reference_fixture(name):
    """The oracle pinned to the reference ABOVE 64^2: BASELINE configs[0] as the reference itself runs it (256^2, content
    start, 50 L-BFGS steps, /root/reference/src/style_transfer_visualizer/optimization.py:162-202 driving
    torch.optim.LBFGS) and 12 full-width L-BFGS steps at 128^2 from the random start - every step's loss triple, the
    optimizer's integer state, the image after every step / at the end (subsampled + float64 checksums)."""
    case = GoldenCase(name)
    m, k = case.meta, case.meta["compact"]
    model = _oracle_model(case)
    assert len(model.blocks) == m["block_count"] and model.style_ids == m["style_ids"] and model.content_ids == m["content_ids"]
    for i, t in enumerate(model.style_targets):
        if f"style_target_{i}" in case.arrays:
            np.testing.assert_allclose(t.numpy(), case.arrays[f"style_target_{i}"], rtol=RTOL, atol=1e-7)
        else:
            np.testing.assert_allclose(t.numpy()[::16, ::16], case.arrays[f"style_target_{i}_sub16"], rtol=RTOL, atol=1e-7)
            assert float(t.double().sum()) == pytest.approx(float(case.arrays[f"style_target_{i}_sum"]), rel=1e-5)
    for i, t in enumerate(model.content_targets):
        assert float(t.double().abs().sum()) == pytest.approx(float(case.arrays[f"content_target_{i}_abs_sum"]), rel=1e-5)
    x0 = case.start_image()
    assert float(x0.double().abs().sum()) == float(case.arrays["x0_abs_sum"])
    states = []

    def lg(x):
        return ocm.loss_and_grad(model, x, m["style_w"], m["content_w"])
    res = optim_ref.run_loop(lg, x0, m["steps"], optimizer="lbfgs", keep_steps=True,
                             on_step=lambda opt: states.append((opt.n_iter, len(opt.old_dirs))))
    g_ref = case.arrays["grad_step1_sub"]
    np.testing.assert_allclose(res["first_grad"].numpy()[..., ::k, ::k], g_ref, rtol=1e-4, atol=1e-6 * float(case.arrays["grad_step1_absmax"]))
    assert states == [tuple(r) for r in case.arrays["lbfgs_state"].tolist()]
    xtol, ltol = case.step_tolerances()
    worst_l = worst_x = 0.0
    for j, key in enumerate(("style", "content", "total")):
        got, want = np.asarray(res["history"][key]), case.arrays[f"{key}_loss"]
        rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-30)
        rel[(got == 0) & (want == 0)] = 0.0
        worst_l = max(worst_l, float(rel.max()))
        assert (rel <= ltol[:, j]).all(), f"{name}: {key} loss differs from the reference's at steps {np.nonzero(rel > ltol[:, j])[0] + 1}"
    if "x_steps_sub" in case.arrays:
        for s_i, x_s in enumerate(res["x_steps"]):
            want = case.arrays["x_steps_sub"][s_i]
            dev = float(np.abs(x_s.numpy()[..., ::k, ::k] - want).max() / float(case.arrays["x_steps_absmax"][s_i]))
            worst_x = max(worst_x, dev)
            assert dev <= xtol[s_i], f"{name}: image after step {s_i + 1} differs by {dev:.2e} of its range"
            assert float(x_s.double().abs().sum()) == pytest.approx(float(case.arrays["x_steps_abs_sum"][s_i]), rel=xtol[s_i])
    for k_full in m["full_steps"]:          # the full images from inside the trajectory, and the gradient the next step saw there
        want = case.arrays[f"x_after_step_{k_full}"]
        dev = float(np.abs(res["x_steps"][k_full - 1].numpy() - want).max() / float(np.abs(want).max()))
        worst_x = max(worst_x, dev)
        assert dev <= xtol[k_full - 1]
        g_here = ocm.loss_and_grad(model, torch.from_numpy(want), m["style_w"], m["content_w"])
        assert float(g_here[2]) == pytest.approx(float(case.arrays["total_loss"][k_full]), rel=RTOL)
        np.testing.assert_allclose(g_here[3].numpy()[..., ::k, ::k], case.arrays[f"grad_at_step_{k_full + 1}_sub"], rtol=1e-4,
                                   atol=1e-6 * float(case.arrays[f"grad_at_step_{k_full + 1}_absmax"]))
    xf = res["x"].numpy()
    dev = float(np.abs(xf[..., ::k, ::k] - case.arrays["x_final_sub"]).max() / float(case.arrays["x_final_absmax"]))
    worst_x = max(worst_x, dev)
    assert dev <= xtol[-1]
    assert float(np.abs(xf.astype(np.float64)).sum()) == pytest.approx(float(case.arrays["x_final_abs_sum"]), rel=xtol[-1])
    # the restatement runs the reference's torch kernels in the reference's order: in THIS container (where the fixtures
    # were made) the agreement is far inside the per-step tolerances - what is measured is printed with -s / on failure
    print(f"{name}: oracle vs reference: worst loss deviation {worst_l:.2e}, worst image deviation {worst_x:.2e}")
    assert int(case.arrays["closure_calls"]) == m["steps"]
    assert case.arrays["logged_steps"].tolist() == list(range(m["log_every"], m["steps"] + 1, m["log_every"]))


def test_lbfgs_restatement_is_torch_lbfgs():
    """LbfgsRef is bit-identical to torch.optim.LBFGS on CPU (same op order)."""
    torch.manual_seed(0)
    a = torch.randn(40, 40)
    a = a @ a.t() + torch.eye(40)
    b = torch.randn(40)

    def f(x):
        return 0.5 * x @ a @ x - b @ x + 0.1 * (x ** 4).sum()

    x_t = torch.zeros(40, requires_grad=True)
    opt = torch.optim.LBFGS([x_t], lr=0.05, max_iter=1, max_eval=1)
    x_r = torch.zeros(40)
    ref = optim_ref.LbfgsRef(x_r, lr=0.05)
    for _ in range(30):
        def closure():
            opt.zero_grad()
            loss = f(x_t)
            loss.backward()
            return loss
        opt.step(closure)

        def closure_r():
            with torch.enable_grad():
                xr = x_r.detach().clone().requires_grad_(True)
                loss = f(xr)
                loss.backward()
            return loss.detach(), xr.grad
        ref.step(closure_r)
        assert torch.equal(x_t.detach(), x_r)


def test_adam_restatement_is_torch_adam():
    torch.manual_seed(1)
    x_t = torch.randn(64, requires_grad=True)
    x_r = x_t.detach().clone()
    opt = torch.optim.Adam([x_t], lr=1e-2)
    ref = optim_ref.AdamRef(x_r, lr=1e-2)
    for _ in range(10):
        def closure():
            opt.zero_grad()
            loss = (x_t ** 2).sum() + x_t.sin().sum()
            loss.backward()
            return loss
        opt.step(closure)

        def closure_r():
            with torch.enable_grad():
                xr = x_r.detach().clone().requires_grad_(True)
                loss = (xr ** 2).sum() + xr.sin().sum()
                loss.backward()
            return loss.detach(), xr.grad
        ref.step(closure_r)
    torch.testing.assert_close(x_t.detach(), x_r, rtol=1e-6, atol=1e-7)


def test_bf16_storage_is_chaotic_under_summation_order():
    """Why the bf16 mode's END-TO-END gradient cannot be pinned tightly (tests/test_gpu_bf16_layerwise.py
    pins it op by op instead): on the reference arithmetic itself, with activations rounded to bf16,
    a 1e-7 relative perturbation of the conv sums - what another fp32 summation order amounts to -
    moves the image gradient by percents while the loss moves by ~1e-4; in fp32 the same
    perturbation is invisible.  Measured at 128^2 full-width VGG19: 5.5 % / 6e-5; here a smaller case."""
    from oracle import core_model_ref as ref
    from style_transfer_visualizer_amd import synthetic
    weights = synthetic.synthetic_conv_weights(0)
    S, C = [0, 5, 10, 19, 28], [21]
    n = 64
    content, style = synthetic.synthetic_image(0, n, n), synthetic.synthetic_image(1, n, n)
    x0 = torch.randn(content.shape, generator=torch.Generator().manual_seed(0))
    orig = ref.run_layer

    def noisy(scale):
        g = torch.Generator().manual_seed(1)

        def run(layer, x):
            y = orig(layer, x)
            return y * (1 + scale * torch.randn(y.shape, generator=g)) if layer[0] == "conv" else y
        return run
    out = {}
    try:
        for bf16 in (False, True):
            model = ref.OracleModel(ref.vgg_program(weights, synthetic.VGG19_CFG), S, C, bf16_storage=bf16)
            model.set_targets(style, content)
            base = ref.loss_and_grad(model, x0, 1e5, 1.0)
            ref.run_layer = noisy(1e-7)
            pert = ref.loss_and_grad(model, x0, 1e5, 1.0)
            ref.run_layer = orig
            out[bf16] = (float((pert[3] - base[3]).norm() / base[3].norm()), abs(float(pert[2]) - float(base[2])) / float(base[2]))
    finally:
        ref.run_layer = orig
    print(f"1e-7 noise on conv sums: fp32 grad {out[False][0]:.1e} loss {out[False][1]:.1e}; "
          f"bf16-storage grad {out[True][0]:.1e} loss {out[True][1]:.1e}")
    assert out[False][0] < 1e-3 and out[False][1] < 1e-5
    assert out[True][0] > 1e-2              # percents: rounding chaos
    assert out[True][1] < 2e-3              # while the loss stays put

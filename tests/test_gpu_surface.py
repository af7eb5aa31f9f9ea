"""GPU tests of the public surface around the kernels:
``gram_matrix`` for any channel count against the reference's known-answer vectors, the
checkpoint-loading branch of ``initialize_vgg`` end to end, the device-side frame/PNG conversion
bit for bit, and the frame cadence with a sink attached to the HIP model.
"""
from __future__ import annotations

import os

import numpy as np
import pytest
import torch

from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import config as stv_config
from style_transfer_visualizer_amd import core_model, image_io, ops, optimization, synthetic
from style_transfer_visualizer_amd.constants import IMAGENET_MEAN, IMAGENET_STD
from tests.conftest import GOLDEN_DIR, GoldenCase

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


class _Bar:
    def update(self, n=1):
        return None

    def set_postfix(self, *a, **k):
        return None

    def close(self):
        return None


# ------------------------------------------------------------------------------ a1 gram_matrix
def test_gram_matrix_known_answers_any_channel_count():
    """tests/golden/gram_kats.npz was produced by the unmodified reference gram_matrix
    (core_model.py:29-63): C = 2, 4 (batch folded in), 6 - none a multiple of the kernel's 8."""
    k = np.load(os.path.join(GOLDEN_DIR, "gram_kats.npz"))
    a = torch.from_numpy(k["kat1_in"]).to(DEV)
    assert core_model.gram_matrix(a).cpu().tolist() == [[1.75, 4.75], [4.75, 15.75]]
    assert np.array_equal(core_model.gram_matrix(a, clamp_max=30).cpu().numpy(), k["kat2_out_clamp30"])
    b = torch.from_numpy(k["kat3_in"]).to(DEV)
    g3 = core_model.gram_matrix(b)
    assert g3.shape == (4, 4) and np.array_equal(g3.cpu().numpy(), k["kat3_out"])
    # backward through the clamp mask (kat4): d(sum G)/dx with clamp 30
    x = a.clone().requires_grad_(True)
    core_model.gram_matrix(x, clamp_max=30).sum().backward()
    assert np.array_equal(x.grad.cpu().numpy(), k["kat4_grad_clamp30"])
    # kat5: C = 6, 5x7 pixels, clamp engaged, MSE against a target, gradient
    f = torch.from_numpy(k["kat5_in"]).to(DEV).requires_grad_(True)
    g5 = core_model.gram_matrix(f, clamp_max=20.0)
    np.testing.assert_allclose(g5.detach().cpu().numpy(), k["kat5_out_clamp20"], rtol=1e-6, atol=1e-7)
    loss = torch.nn.functional.mse_loss(g5, torch.from_numpy(k["kat5_target"]).to(DEV))
    loss.backward()
    assert float(loss) == pytest.approx(float(k["kat5_loss"]), rel=1e-5)
    np.testing.assert_allclose(f.grad.cpu().numpy(), k["kat5_grad"], rtol=1e-5, atol=1e-6 * np.abs(k["kat5_grad"]).max())


@pytest.mark.parametrize("shape", [(1, 3, 64, 64), (1, 5, 17, 9), (2, 7, 8, 8), (1, 64, 32, 32), (1, 13, 40, 56)])
def test_gram_matrix_matches_oracle_forward_backward(shape):
    """The reference's own property test (tests/test_core_model.py:84-92: CxC, symmetric, PSD) plus
    values and gradient against the oracle for channel counts that need padding."""
    t = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape)))
    clamp = float(ocm.gram_matrix(t).abs().max() * t[0, 0].numel() * shape[0] * shape[1] * 0.5)   # engages
    xo = t.clone().requires_grad_(True)
    go = ocm.gram_matrix(xo, clamp_max=clamp)
    w = torch.randn(go.shape, generator=torch.Generator().manual_seed(1))
    (go * w).sum().backward()
    xh = t.to(DEV).requires_grad_(True)
    gh = core_model.gram_matrix(xh, clamp_max=clamp)
    C = shape[0] * shape[1]
    assert gh.shape == (C, C)
    assert torch.allclose(gh, gh.t())
    assert torch.all(torch.linalg.eigvalsh(core_model.gram_matrix(xh.detach()).double().cpu()) >= -1e-6)
    (gh * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(gh.detach().cpu().numpy(), go.detach().numpy(), rtol=2e-5, atol=2e-6 * float(go.abs().max()))
    np.testing.assert_allclose(xh.grad.cpu().numpy(), xo.grad.numpy(), rtol=0, atol=2e-5 * float(xo.grad.abs().max()))


# ------------------------------------------------------------------------------- a7 checkpoint
def test_checkpoint_branch_runs_on_the_hip_path(tmp_path, monkeypatch):
    """initialize_vgg's cached-.pth branch (reference core_model.py:103-117) feeding the HIP model:
    the model built from a fabricated torchvision-layout checkpoint must give the oracle's losses
    and gradient for the same weights."""
    import sys

    from tests.test_core_model_host import _fake_checkpoint
    monkeypatch.delenv("STV_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setitem(sys.modules, "torchvision", None)
    monkeypatch.setitem(sys.modules, "torchvision.models", None)
    old_hub = torch.hub.get_dir()
    torch.hub.set_dir(str(tmp_path / "hub"))
    try:
        state = _fake_checkpoint(tmp_path / "hub" / "checkpoints" / "vgg19-dcbb9e9d.pth")
        model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21]).to(DEV)
    finally:
        torch.hub.set_dir(old_hub)
    weights = [(state[f"features.{i}.weight"], state[f"features.{i}.bias"])
               for i in sorted({int(k.split(".")[1]) for k in state if k.startswith("features.")})]
    oracle = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), [0, 5, 10, 19, 28], [21])
    content, style, x0 = (synthetic.synthetic_image(s, 64, 64) for s in (0, 1, 2))
    oracle.set_targets(style, content)
    s_ref, c_ref, t_ref, g_ref = ocm.loss_and_grad(oracle, x0, 1e5, 1.0)
    model.set_targets(style.to(DEV), content.to(DEV))
    assert model.content_targets[0].shape == oracle.content_targets[0].shape        # [1, C, H, W] as the reference
    np.testing.assert_allclose(model.content_targets[0].float().cpu().numpy(), oracle.content_targets[0].numpy(),
                               rtol=0, atol=2e-5 * float(oracle.content_targets[0].abs().max()))
    x = x0.to(DEV).requires_grad_(True)
    s, c, t = model.loss_and_grad(x, 1e5, 1.0)
    assert float(t) == pytest.approx(float(t_ref), rel=1e-4)
    assert float(c) == pytest.approx(float(c_ref), rel=1e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g_ref.numpy(), rtol=0, atol=1e-4 * float(g_ref.abs().max()))


# ----------------------------------------------------------------------- f2/f3 device conversion
def _ref_frame(x: torch.Tensor, normalize: bool) -> np.ndarray:
    """reference optimization.py:438-452 on the CPU."""
    img = image_io.prepare_image_for_output(x.cpu(), normalize=normalize)
    return (img.squeeze(0).permute(1, 2, 0).numpy() * 255).astype("uint8")


def _ref_png(x: torch.Tensor, normalize: bool) -> np.ndarray:
    """runtime/output.py:92-101: prepare_image_for_output + torchvision save_image's conversion."""
    img = image_io.prepare_image_for_output(x.cpu(), normalize=normalize)[0]
    return img.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()


@pytest.mark.parametrize("hw", [(64, 64), (75, 101), (17, 3), (512, 512)])
@pytest.mark.parametrize("normalize", [False, True])
def test_image_to_u8_is_bit_exact(hw, normalize):
    g = torch.Generator().manual_seed(hw[0] * 7 + hw[1])
    x = torch.randn(1, 3, *hw, generator=g) * (1.5 if normalize else 0.6) + (0.0 if normalize else 0.5)
    flat = x.view(-1)
    flat[::97] = float("nan")
    flat[1::193] = float("inf")
    flat[2::211] = float("-inf")
    # values that sit exactly on k/255 boundaries (truncation vs rounding differ there)
    flat[3::101] = torch.arange(flat[3::101].numel()).remainder(256).float() / 255
    xd = x.to(DEV)
    mean, std = (IMAGENET_MEAN, IMAGENET_STD) if normalize else (None, None)
    got_f = ops.image_to_u8(xd, mean=mean, std=std, rounding=False).cpu().numpy()
    got_p = ops.image_to_u8(xd, mean=mean, std=std, rounding=True).cpu().numpy()
    assert got_f.shape == (*hw, 3) and got_f.dtype == np.uint8
    assert np.array_equal(got_f, _ref_frame(x, normalize))
    assert np.array_equal(got_p, _ref_png(x, normalize))
    assert np.array_equal(image_io.frame_uint8(xd, normalize=normalize), got_f)


def test_save_image_from_gpu_matches_host_path(tmp_path):
    from PIL import Image
    x = torch.randn(1, 3, 70, 90, generator=torch.Generator().manual_seed(5))
    image_io.save_image(x.to(DEV), tmp_path / "g.png", normalize=True)
    image_io.save_image(x, tmp_path / "c.png", normalize=True)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "g.png")), np.asarray(Image.open(tmp_path / "c.png")))


# ------------------------------------------------------------------------- a14 / f3 frame cadence
class _MemorySink:
    def __init__(self):
        self.frames, self.closed = [], False

    def append_data(self, frame):
        self.frames.append(frame.copy())

    def close(self):
        self.closed = True


@pytest.mark.parametrize("steps,save_every", [(7, 2), (6, 3), (5, 9)])
def test_frames_from_the_hip_model_cadence_and_values(steps, save_every, monkeypatch):
    """A sink attached to the HIP model: floor(steps/save_every) frames, at the steps the reference
    writes them (optimization.py:424-437), each the truncating uint8 image of that step."""
    case = GoldenCase("mini_white_lbfgs")
    weights = case.weights()
    monkeypatch.setattr(core_model, "initialize_vgg", lambda: core_model.build_vgg_features(weights, case.cfg).eval())
    cfg = stv_config.StyleTransferConfig.model_validate({})
    oc = cfg.optimization
    oc.steps, oc.init_method = steps, "random"
    oc.style_layers, oc.content_layers = list(case.meta["style_layers"]), list(case.meta["content_layers"])
    cfg.video.create_video, cfg.video.save_every = True, save_every
    content, style = case.images()
    model, x, opt = core_model.prepare_model_and_input(content.to(DEV), style.to(DEV), DEV, oc)
    sink, gif = _MemorySink(), _MemorySink()
    seen = []

    def on_frame(frame, step):
        seen.append((step, frame.copy(), x.detach().clone()))
    runner = optimization.OptimizationRunner(
        model, x, cfg, optimizer=opt, progress_bar=_Bar(), video_writer=sink, gif_collector=gif,
        callbacks=optimization.OptimizationCallbacks(on_video_frame=on_frame))
    runner.run()
    assert len(sink.frames) == steps // save_every == len(gif.frames)
    assert [s for s, _, _ in seen] == [k for k in range(1, steps + 1) if k % save_every == 0]
    for (step, frame, x_then), stored in zip(seen, sink.frames, strict=True):
        assert frame.dtype == np.uint8 and frame.shape == (x.shape[2], x.shape[3], 3)
        assert np.array_equal(frame, stored)
        assert np.array_equal(frame, _ref_frame(x_then, oc.normalize)), f"frame of step {step}"


# ------------------------------------------------ loss history kept by the combine kernel itself
@pytest.mark.parametrize("bf16", [False, True])
def test_producer_logged_history_changes_nothing(bf16, monkeypatch):
    """The combine kernel appends each step's scores to the LossAccumulator's ring itself
    (stv_loss_combine_log) instead of one copy kernel per step: image and exported loss history are
    BIT-identical to the run with the feature off, also once the ring has wrapped around."""
    from style_transfer_visualizer_amd import loss_accumulator
    from style_transfer_visualizer_amd.loss_accumulator import LossAccumulator
    case = GoldenCase("mini_white_lbfgs")
    weights = case.weights()
    steps = 9
    results = {}
    for mode in ("copy", "producer", "producer_device_ring"):
        with monkeypatch.context() as mp:
            # "producer": the ring is host memory the combine kernel writes into, a flush waits for that kernel only
            # (no copy, no stream synchronisation); "producer_device_ring": STV_HOST_LOG=0, the ring stays on the device
            mp.setenv("STV_HOST_LOG", "0" if mode == "producer_device_ring" else "1")
            mp.setattr(core_model, "initialize_vgg", lambda: core_model.build_vgg_features(weights, case.cfg).eval())
            mp.setattr(loss_accumulator, "DEFAULT_HISTORY_CAPACITY", 6)       # the ring wraps after six records
            mp.setattr(optimization, "DEFAULT_HISTORY_CAPACITY", 6)
            handed = []
            if mode == "copy":
                mp.setattr(LossAccumulator, "device_log", lambda self: None)
            else:
                orig = LossAccumulator.device_log

                def spy(self, orig=orig, handed=handed):
                    log = orig(self)
                    handed.append(log is not None)
                    return log
                mp.setattr(LossAccumulator, "device_log", spy)
            cfg = stv_config.StyleTransferConfig.model_validate({})
            oc = cfg.optimization
            oc.steps, oc.init_method, oc.seed = steps, "content", 0
            oc.style_layers, oc.content_layers = list(case.meta["style_layers"]), list(case.meta["content_layers"])
            cfg.video.create_video = False
            content, style = case.images()
            model, x, opt = core_model.prepare_model_and_input(content.to(DEV), style.to(DEV), DEV, oc,
                                                               precision="bf16" if bf16 else "fp32")
            runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=_Bar())
            _, history, _ = runner.run()
            results[mode] = (x.detach().clone(), history, handed, runner._loss_accumulator._box is not None)
    assert [results[m][3] for m in ("copy", "producer", "producer_device_ring")] == [False, True, False]
    assert results["producer"][2] == [True] * steps and results["producer_device_ring"][2] == [True] * steps
    assert torch.equal(results["copy"][0], results["producer"][0])
    assert results["copy"][1] == results["producer"][1]
    assert torch.equal(results["copy"][0], results["producer_device_ring"][0])
    assert results["copy"][1] == results["producer_device_ring"][1]
    assert len(results["producer"][1]["total_loss"]) == 6       # the last six of nine steps


# ---------------------------------------------- optimizer update at the end of the closure's launch
@pytest.mark.parametrize("bf16", [False, True])
def test_update_fused_into_the_closure_launch_changes_nothing(bf16, monkeypatch):
    """``HipLBFGS.step(closure)`` offers its update to the closure (optimizers.StepRequest); the model's fused path
    appends it to its schedule (stv_op_t LBFGS_STEP), so a step is ONE replayed hipGraph.  Same kernels in the same
    order: image, loss history and the optimizer's integer state are BIT-identical to STV_FUSE_STEP=0 (four eager
    launches behind the graph), and every step of the fused run really took the offer."""
    from style_transfer_visualizer_amd import optimizers
    case = GoldenCase("mini_white_lbfgs")
    weights = case.weights()
    steps = 14
    results = {}
    for mode in ("0", "1"):
        with monkeypatch.context() as mp:
            mp.setenv("STV_FUSE_STEP", mode)
            mp.setattr(core_model, "initialize_vgg", lambda: core_model.build_vgg_features(weights, case.cfg).eval())
            taken = []
            orig = optimizers.claim_step

            def spy(x, orig=orig, taken=taken):
                req = orig(x)
                taken.append(req is not None)
                return req
            mp.setattr(optimizers, "claim_step", spy)
            cfg = stv_config.StyleTransferConfig.model_validate({})
            oc = cfg.optimization
            oc.steps, oc.init_method, oc.seed = steps, "content", 0
            oc.style_layers, oc.content_layers = list(case.meta["style_layers"]), list(case.meta["content_layers"])
            cfg.video.create_video = False
            content, style = case.images()
            model, x, opt = core_model.prepare_model_and_input(content.to(DEV), style.to(DEV), DEV, oc,
                                                               precision="bf16" if bf16 else "fp32")
            assert isinstance(opt, optimizers.HipLBFGS)
            runner = optimization.OptimizationRunner(model, x, cfg, optimizer=opt, progress_bar=_Bar())
            _, history, _ = runner.run()
            # a closure evaluated OUTSIDE optimizer.step must not find a request to take, and neither must a closure
            # that does not promise to evaluate the model once per call (optimizers.single_evaluation)
            model.loss_and_grad(x, oc.style_w, oc.content_w)
            x_keep = x.detach().clone()
            opt.step(lambda: model.loss_and_grad(x, oc.style_w, oc.content_w)[2])
            with torch.no_grad():
                x.copy_(x_keep)
            results[mode] = (x_keep, history, opt.device_state(), list(taken))
    assert results["0"][3] == [False] * (steps + 2)
    assert results["1"][3] == [True] * steps + [False, False]
    assert torch.equal(results["0"][0], results["1"][0])
    assert results["0"][1] == results["1"][1]
    assert results["0"][2] == results["1"][2] and results["1"][2]["n_iter"] == steps + 1


# ------------------------------------------------------------------ input checks of the HIP model
def test_input_channel_count_and_stale_backward_are_rejected(monkeypatch):
    case = GoldenCase("mini_white_lbfgs")
    weights = case.weights()
    monkeypatch.setattr(core_model, "initialize_vgg", lambda: core_model.build_vgg_features(weights, case.cfg).eval())
    S, C = case.meta["style_layers"], case.meta["content_layers"]
    model = core_model.StyleContentModel(S, C).to(DEV)
    img = synthetic.synthetic_image(0, 64, 64).to(DEV)
    for bad in (torch.rand(1, 1, 64, 64, device=DEV), torch.rand(1, 4, 64, 64, device=DEV)):
        with pytest.raises(RuntimeError, match="expected input with 3 channels"):
            model.set_targets(img, bad)
        with pytest.raises(RuntimeError, match="expected input with 3 channels"):
            model.set_targets(bad, img)
    model.set_targets(img, img)
    with pytest.raises(RuntimeError, match="expected input with 3 channels"):
        model(torch.rand(1, 4, 64, 64, device=DEV))
    # backward of a forward whose activations a later evaluation overwrote: loud, not silently wrong
    x1 = synthetic.synthetic_image(1, 64, 64).to(DEV).requires_grad_(True)
    x2 = synthetic.synthetic_image(2, 64, 64).to(DEV).requires_grad_(True)
    s1, c1 = model(x1)
    model(x2)
    with pytest.raises(RuntimeError, match="overwritten by a later evaluation"):
        (torch.stack(s1).sum() + torch.stack(c1).sum()).backward()
    # the same sequence in the right order gives the oracle's gradient
    s1, c1 = model(x1)
    (1e5 * torch.stack(s1).sum() + torch.stack(c1).sum()).backward()
    oracle = ocm.OracleModel(ocm.vgg_program(weights, case.cfg), S, C)
    oracle.set_targets(img.cpu(), img.cpu())
    _, _, _, g_ref = ocm.loss_and_grad(oracle, x1.detach().cpu(), 1e5, 1.0)
    np.testing.assert_allclose(x1.grad.cpu().numpy(), g_ref.numpy(), rtol=0, atol=2e-4 * float(g_ref.abs().max()))
    # a non-contiguous / non-fp32 input goes through ONE persistent staging buffer: no program growth
    eng = next(iter(model._engines.values()))
    xt = synthetic.synthetic_image(3, 64, 64).to(DEV).permute(0, 1, 3, 2)       # non-contiguous view
    model(xt)
    n_prog = len(eng._programs)
    for _ in range(4):
        model(xt.clone())                      # fresh non-contiguous temporary each time
        model(xt.double())
    assert len(eng._programs) == n_prog


def test_style_transfer_batch_runs_independent_pairs(tmp_path, monkeypatch):
    """main.style_transfer_batch (BASELINE configs[3] as a library call; one process here, ranks under
    torchrun): every pair gets its own targets and optimizer state and its own PNG; results come back in order."""
    from PIL import Image

    from style_transfer_visualizer_amd import main as stv_main
    from style_transfer_visualizer_amd.type_defs import InputPaths
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    for name, seed in (("c0", 0), ("c1", 3), ("s", 1)):
        img = synthetic.synthetic_image(seed, 64, 64, normalize=False)[0].permute(1, 2, 0).mul(255).byte().numpy()
        Image.fromarray(img).save(tmp_path / f"{name}.png")
    cfg = stv_config.StyleTransferConfig.model_validate({})
    cfg.optimization.steps, cfg.optimization.init_method = 4, "random"
    cfg.video.create_video, cfg.video.final_only = False, True
    cfg.output.output = str(tmp_path / "out")
    cfg.hardware.device = "cuda"
    pairs = [InputPaths(content_path=str(tmp_path / "c0.png"), style_path=str(tmp_path / "s.png")),
             InputPaths(content_path=str(tmp_path / "c1.png"), style_path=str(tmp_path / "s.png"))]
    out = stv_main.style_transfer_batch(pairs, cfg, images_per_gpu=1)
    assert len(out) == 2 and all(t.shape == (1, 3, 64, 64) for t in out)
    assert all(float(t.min()) >= 0.0 and float(t.max()) <= 1.0 for t in out)
    assert not torch.equal(out[0], out[1])
    assert (tmp_path / "out" / "stylized_c0_x_s.png").is_file() and (tmp_path / "out" / "stylized_c1_x_s.png").is_file()
    # several images of a rank in flight at once (each on its own host thread and stream; the default is 3): the same
    # images, bit for bit - the seeded start image is drawn under a lock, and nothing else is shared
    from style_transfer_visualizer_amd import parallel
    assert parallel.images_in_flight(5) == 1 and parallel.images_in_flight(1) == 1 and parallel.images_in_flight(5, 2) == 2   # opt-in
    three = pairs + pairs[:1]
    for k in (2, 3):
        cfg.output.output = str(tmp_path / f"out{k}")
        cfg.output.log_loss = str(tmp_path / f"out{k}" / "loss.csv")
        par = stv_main.style_transfer_batch(three, cfg, images_per_gpu=k)
        assert len(par) == 3 and all(torch.equal(a, b) for a, b in zip((out[0], out[1], out[0]), par, strict=True)), f"{k} images in flight"
        # no two pairs write one file: the pair that occurs twice has its index in its PNG name, every pair its own CSV
        names = sorted(p.name for p in (tmp_path / f"out{k}").iterdir() if p.suffix in (".png", ".csv") and not p.name.startswith("loss_plot"))
        assert names == ["loss_000.csv", "loss_001.csv", "loss_002.csv", "stylized_c0_x_s_000.png", "stylized_c0_x_s_002.png",
                         "stylized_c1_x_s.png"], names
        assert (tmp_path / f"out{k}" / "loss_000.csv").read_text() == (tmp_path / f"out{k}" / "loss_002.csv").read_text()
    cfg.output.log_loss = None


# ------------------------------------------------------------------- a2/a5 replaced content targets
def test_fresh_content_targets_are_never_served_from_a_stale_cache(monkeypatch):
    """The reference reads ``model.content_targets`` at every forward (core_model.py:266-295), so a user may
    assign new [1,C,H,W] tensors at any time.  The engine caches their NHWC conversion; the caching allocator
    hands a freed target's address (and version 0) to the next fresh tensor, so the cache must recognise the
    tensor object, not its address: three fresh NCHW-contiguous targets in a row, loss checked each time."""
    cfg = (8, 8, "M", 16, 16)
    weights = synthetic.synthetic_conv_weights(3, cfg)
    monkeypatch.setattr(core_model, "initialize_vgg", lambda: core_model.build_vgg_features(weights, cfg).eval())
    model = core_model.StyleContentModel([0], [5]).to(DEV)
    content, style, x0 = (synthetic.synthetic_image(s, 32, 32) for s in (0, 1, 2))
    model.set_targets(style.to(DEV), content.to(DEV))
    oracle = ocm.OracleModel(ocm.vgg_program(weights, cfg), [0], [5])
    oracle.set_targets(style, content)
    x = x0.to(DEV).requires_grad_(True)
    shape = tuple(model.content_targets[0].shape)
    seen = set()
    for k in range(4):
        tgt = torch.randn(shape, generator=torch.Generator().manual_seed(100 + k)) * (k + 1)
        fresh = tgt.to(DEV)                                   # NCHW-contiguous: goes through the conversion cache
        seen.add(fresh.data_ptr())
        model.content_targets = [fresh]
        oracle.content_targets = [tgt]
        _, c, _ = model.loss_and_grad(x, 1e5, 1.0)
        _, c_ref, _, _ = ocm.loss_and_grad(oracle, x0, 1e5, 1.0)
        assert float(c) == pytest.approx(float(c_ref), rel=1e-5), f"assignment {k}: content loss is that of another target"
        del fresh
    # (on the caching allocator the later targets reuse the first one's address: the case the cache key alone missed)
    print(f"distinct target addresses over 4 assignments: {len(seen)}")

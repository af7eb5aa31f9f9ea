"""OptimizationRunner host logic on CPU with stand-in models (the HIP model itself needs a GPU).

Pins the reference's loop semantics (optimization.py:162-202, 274-348, 424-489): 1-based steps,
one accepted step per optimizer.step however often the closure runs, logging cadence, frame
cadence, error texts.
"""
from __future__ import annotations

import logging

import numpy as np
import pytest
import torch
from torch import nn

from style_transfer_visualizer_amd import config as stv_config
from style_transfer_visualizer_amd import optimization as opt_mod
from style_transfer_visualizer_amd.optimization import OptimizationCallbacks, OptimizationRunner


class TinyModel(nn.Module):
    """Two 'style' terms and one 'content' term of a quadratic in x."""

    def forward(self, x):
        return [(x ** 2).mean(), ((x - 1) ** 2).mean()], [((x + 0.5) ** 2).mean()]


class Bar:
    def __init__(self):
        self.updates, self.postfixes, self.closed = 0, [], False

    def update(self, n=1):
        self.updates += n

    def set_postfix(self, d=None, refresh=True, **kw):
        self.postfixes.append(d)

    def close(self):
        self.closed = True


class Sink:
    def __init__(self):
        self.frames = []

    def append_data(self, frame):
        self.frames.append(frame)

    def close(self):
        pass


class MultiProbeSGD(torch.optim.SGD):
    """Calls the closure three times per step (like a line search would)."""

    def step(self, closure=None):
        for _ in range(3):
            with torch.enable_grad():
                loss = closure()
        super().step()
        return loss


def _cfg(steps=4, log_every=2, save_every=2, log_loss=None):
    cfg = stv_config.StyleTransferConfig.model_validate({})
    cfg.optimization.steps = steps
    cfg.optimization.style_w = 2.0
    cfg.output.log_every = log_every
    cfg.output.log_loss = log_loss
    cfg.video.save_every = save_every
    cfg.optimization.normalize = False
    return cfg


def _img():
    return torch.full((1, 3, 8, 8), 0.25, requires_grad=True)


def test_history_length_cadence_and_return_triple():
    x = _img()
    seen = []
    runner = OptimizationRunner(TinyModel(), x, _cfg(), optimizer=torch.optim.Adam([x], lr=0.1), progress_bar=Bar(),
                                callbacks=OptimizationCallbacks(on_step_end=lambda m: seen.append((m.step, m.has_values))))
    out, history, elapsed = runner.run()
    assert out is x and elapsed >= 0
    assert seen == [(1, False), (2, True), (3, False), (4, True)]        # reference tests :686-723
    assert {k: len(v) for k, v in history.items()} == {"style_loss": 4, "content_loss": 4, "total_loss": 4}
    assert history["total_loss"][0] > history["total_loss"][-1]
    s0, c0 = (0.25 ** 2 + 0.75 ** 2), 0.75 ** 2
    assert history["style_loss"][0] == pytest.approx(s0) and history["content_loss"][0] == pytest.approx(c0)
    assert history["total_loss"][0] == pytest.approx(2.0 * s0 + c0)
    assert runner.progress_bar.updates == 4


def test_default_optimizer_is_lbfgs_with_config_bounds():
    x = _img()
    cfg = _cfg(steps=3)
    cfg.optimization.lr, cfg.optimization.lbfgs_max_iter, cfg.optimization.lbfgs_max_eval = 0.5, 2, 3
    runner = OptimizationRunner(TinyModel(), x, cfg, progress_bar=Bar())
    assert isinstance(runner.optimizer, torch.optim.LBFGS)      # CPU image -> torch's optimizer
    g = runner.optimizer.param_groups[0]
    assert (g["lr"], g["max_iter"], g["max_eval"]) == (0.5, 2, 3)
    _, history, _ = runner.run()
    assert len(history["total_loss"]) == 3


def test_one_accepted_step_per_optimizer_step_even_with_many_closures():
    x = _img()
    video, gif, frames = Sink(), Sink(), []
    runner = OptimizationRunner(TinyModel(), x, _cfg(steps=4, save_every=2), optimizer=MultiProbeSGD([x], lr=0.1),
                                progress_bar=Bar(), video_writer=video, gif_collector=gif,
                                callbacks=OptimizationCallbacks(on_video_frame=lambda f, s: frames.append(s)))
    _, history, _ = runner.run()
    assert runner._closure_calls == 12 and len(history["total_loss"]) == 4
    assert frames == [2, 4] and len(video.frames) == 2 and len(gif.frames) == 2     # floor(steps/save_every)
    f = video.frames[0]
    assert f.dtype == np.uint8 and f.shape == (8, 8, 3)
    expected = (x.detach().clamp(0, 1)[0].permute(1, 2, 0).numpy() * 255).astype("uint8")   # truncation
    assert np.array_equal(video.frames[-1], expected)


def test_intro_crossfade_once_before_first_frame():
    x = _img()
    video = Sink()
    intro = np.zeros((8, 8, 3), dtype=np.uint8)
    runner = OptimizationRunner(TinyModel(), x, _cfg(steps=4, save_every=2), optimizer=torch.optim.SGD([x], lr=0.1),
                                progress_bar=Bar(), video_writer=video, intro_last_frame=intro, intro_crossfade_frames=3)
    runner.run()
    assert len(video.frames) == 3 + 2 and runner.intro_transition_done


def test_csv_mode_returns_empty_history_and_closes(tmp_path):
    x = _img()
    path = tmp_path / "loss.csv"
    runner = OptimizationRunner(TinyModel(), x, _cfg(steps=4, log_loss=str(path)), optimizer=torch.optim.SGD([x], lr=0.1),
                                progress_bar=Bar())
    _, history, _ = runner.run()
    assert history == {} and runner.loss_logger.file.closed
    rows = path.read_text().strip().splitlines()
    assert rows[0] == "step,style_loss,content_loss,total_loss" and [r.split(",")[0] for r in rows[1:]] == ["2", "4"]


def test_csv_open_failure_falls_back_to_history(tmp_path):
    x = _img()
    errors = []
    blocker = tmp_path / "file"
    blocker.write_text("x")
    cfg = _cfg(steps=2, log_loss=str(blocker / "loss.csv"))     # parent is a file -> OSError
    runner = OptimizationRunner(TinyModel(), x, cfg, optimizer=torch.optim.SGD([x], lr=0.1), progress_bar=Bar(),
                                callbacks=OptimizationCallbacks(on_logging_error=errors.append))
    _, history, _ = runner.run()
    assert len(errors) == 1 and isinstance(errors[0], OSError) and len(history["total_loss"]) == 2


def test_errors_and_guards():
    x = _img()
    with pytest.raises(ValueError, match="Provide either optimizer or optimizer_factory, not both."):
        OptimizationRunner(TinyModel(), x, _cfg(), optimizer=torch.optim.SGD([x], lr=0.1),
                           optimizer_factory=lambda t: torch.optim.SGD([t], lr=0.1))
    made = []
    runner = OptimizationRunner(TinyModel(), x, _cfg(), optimizer_factory=lambda t: made.append(t) or torch.optim.SGD([t], lr=0.1))
    assert made == [x]
    with pytest.raises(RuntimeError, match="Progress bar not initialized"):
        _ = runner.progress_bar

    class NoClosure(torch.optim.SGD):
        def step(self, closure=None):
            return None
    bad = OptimizationRunner(TinyModel(), x, _cfg(), optimizer=NoClosure([x], lr=0.1), progress_bar=Bar())
    with pytest.raises(RuntimeError, match="Optimizer closure did not record metrics for step 1"):
        bad.run()


def test_closure_after_completion_returns_last_loss():
    x = _img()
    runner = OptimizationRunner(TinyModel(), x, _cfg(steps=2), optimizer=torch.optim.SGD([x], lr=0.1), progress_bar=Bar())
    assert float(runner._closure.__self__._final_loss_tensor()) == 0.0
    _, history, _ = runner.run()
    calls = runner._closure_calls
    late = runner._closure()
    assert float(late) == pytest.approx(history["total_loss"][-1]) and runner._closure_calls == calls + 1


def test_history_is_capped_and_warned(caplog):
    x = torch.zeros(1, 1, 2, 2, requires_grad=True)
    cfg = _cfg(steps=2050, log_every=1000, save_every=5000)
    logger = logging.getLogger("style_transfer")
    logger.propagate = True
    try:
        with caplog.at_level(logging.WARNING, logger="style_transfer"):
            runner = OptimizationRunner(TinyModel(), x, cfg, optimizer=torch.optim.SGD([x], lr=1e-3), progress_bar=Bar())
            _, history, _ = runner.run()
    finally:
        logger.propagate = False
    assert len(history["total_loss"]) == 2048
    assert any("capped at 2048" in r.getMessage() for r in caplog.records)


def test_nonfinite_losses_warn(caplog):
    class Inf(nn.Module):
        def forward(self, x):
            return [x.sum() * float("inf")], [x.sum() * 0]
    x = _img()
    logger = logging.getLogger("style_transfer")
    logger.propagate = True
    try:
        with caplog.at_level(logging.WARNING, logger="style_transfer"):
            OptimizationRunner(Inf(), x, _cfg(steps=1), optimizer=torch.optim.SGD([x], lr=0.0), progress_bar=Bar()).run()
    finally:
        logger.propagate = False
    msgs = [r.getMessage() for r in caplog.records]
    assert "Non-finite style score at step 1" in msgs
    assert any(m.startswith("Non-finite total loss at step 1") for m in msgs)


def test_fused_model_path_is_used_when_offered():
    """A model exposing loss_and_grad is driven without autograd and x.grad is what it wrote."""
    class Fused(nn.Module):
        calls = 0

        def loss_and_grad(self, x, style_w, content_w):
            Fused.calls += 1
            x.grad = torch.ones_like(x)
            s, c = (x.detach() ** 2).mean(), x.detach().mean() * 0
            return s, c, style_w * s + content_w * c

        def forward(self, x):
            raise AssertionError("autograd path must not run")
    x = _img()
    runner = OptimizationRunner(Fused(), x, _cfg(steps=3), optimizer=torch.optim.SGD([x], lr=0.1), progress_bar=Bar())
    _, history, _ = runner.run()
    assert Fused.calls == 3 and len(history["total_loss"]) == 3
    assert torch.allclose(x.detach(), torch.full_like(x, 0.25 - 0.3))
    assert opt_mod.append_crossfade is not None


@pytest.mark.parametrize("csv_mode", [False, True])
def test_fused_path_warns_for_nonfinite_steps_between_logging_points(caplog, tmp_path, csv_mode):
    """Reference optimization.py:375-391 checks every step's three scores.  The fused path checks them at the
    flush from the device ring - every step since the last flush, with its own step id, also in CSV mode
    (where the history itself is off) and for steps after the last logging point."""
    class Fused(nn.Module):
        calls = 0

        def loss_and_grad(self, x, style_w, content_w):
            Fused.calls += 1
            x.grad = torch.zeros_like(x)
            bad = {3: (float("inf"), 0.0), 5: (1.0, float("nan")), 7: (float("nan"), 1.0)}.get(Fused.calls, (1.0, 2.0))
            s, c = torch.tensor(bad[0]), torch.tensor(bad[1])
            return s, c, style_w * s + content_w * c
    x = _img()
    cfg = _cfg(steps=7, log_every=4, log_loss=str(tmp_path / "loss.csv") if csv_mode else None)
    logger = logging.getLogger("style_transfer")
    logger.propagate = True
    try:
        with caplog.at_level(logging.WARNING, logger="style_transfer"):
            runner = OptimizationRunner(Fused(), x, cfg, optimizer=torch.optim.SGD([x], lr=0.0), progress_bar=Bar())
            _, history, _ = runner.run()
    finally:
        logger.propagate = False
    assert (history == {}) if csv_mode else (len(history["total_loss"]) == 7)
    msgs = [r.getMessage() for r in caplog.records if "Non-finite" in r.getMessage()]
    assert msgs == ["Non-finite style score at step 3", "Non-finite total loss at step 3, using previous loss",
                    "Non-finite content score at step 5", "Non-finite total loss at step 5, using previous loss",
                    "Non-finite style score at step 7", "Non-finite total loss at step 7, using previous loss"]


@pytest.mark.parametrize("fused", [False, True])
def test_no_python_scalar_conversion_between_flushes(monkeypatch, fused):
    """Reference tests/test_optimization.py:943-970: with log_every beyond the run length no closure may turn a
    tensor into a Python scalar (`Tensor.item` patched to fail).  Here also `Tensor.tolist` - what this build's
    flush uses - and for the fused path too (whose per-step finite checks happen at the flush, from the ring)."""
    class Fused(nn.Module):
        def loss_and_grad(self, x, style_w, content_w):
            x.grad = 2 * x.detach()
            s, c = (x.detach() ** 2).mean(), x.detach().mean() * 0
            return s, c, style_w * s + content_w * c
    x = _img()
    cfg = _cfg(steps=3, log_every=50, save_every=100)
    runner = OptimizationRunner(Fused() if fused else TinyModel(), x, cfg, optimizer=torch.optim.SGD([x], lr=0.5), progress_bar=Bar())
    calls = []
    real_tolist = torch.Tensor.tolist

    def fail_item(_t):
        raise AssertionError("tensor.item used during closure")

    def counting_tolist(t):
        calls.append(tuple(t.shape))
        return real_tolist(t)
    monkeypatch.setattr(torch.Tensor, "item", fail_item)
    monkeypatch.setattr(torch.Tensor, "tolist", counting_tolist)
    _, history, _ = runner.run()
    monkeypatch.undo()
    # nothing inside the loop; afterwards one transfer for the unchecked steps (fused path) and one for the history
    assert len(history["total_loss"]) == 3
    assert calls == ([(3, 3), (3, 3)] if fused else [(3, 3)]), calls

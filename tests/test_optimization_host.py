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


# ---- behaviours the reference's own tests pin on private helpers and callbacks (tests/test_optimization.py) ---------
class MemorySink:
    def __init__(self):
        self.frames = []

    def append_data(self, frame):
        self.frames.append(np.asarray(frame, dtype=np.uint8))

    def close(self):
        return None


class CountingBar(Bar):
    def __init__(self):
        super().__init__()
        self.postfix_calls = []

    def set_postfix(self, d=None, refresh=True, **kw):
        self.postfix_calls.append(d)
        super().set_postfix(d, refresh, **kw)


def _runner(cfg=None, **kw):
    x = _img()
    kw.setdefault("optimizer", torch.optim.Adam([x]))
    kw.setdefault("progress_bar", Bar())
    return OptimizationRunner(TinyModel(), x, cfg or _cfg(steps=1, log_every=1, save_every=1), **kw), x


def test_prepare_image_for_output_without_normalisation_is_a_clamp():
    """reference tests/test_optimization.py:133-139."""
    from style_transfer_visualizer_amd import image_io
    t = torch.rand(1, 3, 16, 16) * 3 - 1
    out = image_io.prepare_image_for_output(t, normalize=False)
    assert out.shape == t.shape and bool((out >= 0).all()) and bool((out <= 1).all())
    inside = (t >= 0) & (t <= 1)
    assert torch.equal(out[inside], t[inside])


@pytest.mark.parametrize("opt_class", [torch.optim.Adam, torch.optim.LBFGS])
def test_one_frame_and_one_postfix_for_one_saved_step(opt_class):
    """reference :200-236: steps=1, save_every=1 -> exactly one appended frame, one progress postfix."""
    x = _img()
    bar, sink = CountingBar(), MemorySink()
    r = OptimizationRunner(TinyModel(), x, _cfg(steps=1, log_every=1, save_every=1),
                           optimizer=opt_class([x]), progress_bar=bar, video_writer=sink)
    r.run()
    assert len(sink.frames) == 1 and sink.frames[0].shape == (8, 8, 3) and sink.frames[0].dtype == np.uint8
    assert len(bar.postfix_calls) == 1


def test_gif_collector_gets_the_frames_and_the_intro_crossfade():
    """reference :394-441: a gif collector alone (no video writer) receives frames; with an intro frame and
    crossfade frames configured it gets those first."""
    cfg = _cfg(steps=2, log_every=1, save_every=1)
    cfg.video.create_video = False
    cfg.video.create_gif = True
    cfg.video.gif_include_intro = True
    x = _img()
    gif = MemorySink()
    intro = np.zeros((8, 8, 3), dtype=np.uint8)
    OptimizationRunner(TinyModel(), x, cfg, optimizer=torch.optim.Adam([x]), progress_bar=Bar(), gif_collector=gif,
                       intro_last_frame=intro, intro_crossfade_frames=2).run()
    assert len(gif.frames) == 2 + 2                     # two crossfade frames, then one frame per step


def test_callbacks_fire_once_per_step():
    """reference :653-684."""
    started, ended = [], []
    cb = OptimizationCallbacks(on_step_start=started.append, on_step_end=lambda m: ended.append(m.total_loss))
    r, _ = _runner(callbacks=cb)
    r.run()
    assert started == [1] and len(ended) == 1 and ended[0] is not None


def test_step_metrics_carry_floats_only_at_the_logging_interval():
    """reference :686-723: (step, has values) = (1, F), (2, T), (3, F), (4, T) for log_every = 2."""
    seen = []
    cb = OptimizationCallbacks(on_step_end=lambda m: seen.append((m.step, m.total_loss is not None)))
    r, _ = _runner(_cfg(steps=4, log_every=2, save_every=100), callbacks=cb)
    r.run()
    assert seen == [(1, False), (2, True), (3, False), (4, True)]


def test_summary_is_silent_before_any_step(caplog):
    """reference :725-742."""
    r, _ = _runner()
    caplog.set_level("INFO")
    r._log_optimization_summary()
    assert "Optimization finished" not in caplog.text


def test_check_finite_debug_line_and_warning_texts(caplog):
    """reference :972-1031: a DEBUG line naming the step; three WARNING texts for non-finite values."""
    r, _ = _runner()
    caplog.set_level("DEBUG", logger="style_transfer")
    one = torch.ones(())
    r._check_finite(one, one, one, step_idx=5)
    assert "Step 5" in caplog.text
    caplog.clear()
    caplog.set_level("WARNING")
    nan = torch.tensor(float("nan"))
    r._check_finite(nan, nan, nan, step_idx=5)
    for what in ("Non-finite style score", "Non-finite content score", "Non-finite total loss"):
        assert what in caplog.text


def test_no_frame_when_the_image_cannot_be_prepared(monkeypatch):
    """reference :1033-1073: prepare_image_for_output -> None: nothing appended, no postfix."""
    from style_transfer_visualizer_amd import image_io
    monkeypatch.setattr(image_io, "prepare_image_for_output", lambda *a, **k: None)
    bar, sink = CountingBar(), MemorySink()
    r, _ = _runner(progress_bar=bar, video_writer=sink)
    r._maybe_write_video_frame(opt_mod.StepMetrics(step=1, style_loss=1.0, content_loss=1.0, total_loss=1.0))
    assert sink.frames == [] and bar.postfix_calls == []


def test_video_frame_hook_and_postfix_on_a_saved_frame():
    """reference :1075-1137."""
    steps_seen = []
    bar, sink = CountingBar(), MemorySink()
    cb = OptimizationCallbacks(on_video_frame=lambda _frame, step: steps_seen.append(step))
    r, _ = _runner(progress_bar=bar, video_writer=sink, callbacks=cb)
    r._maybe_write_video_frame(opt_mod.StepMetrics(step=1, style_loss=1.0, content_loss=1.0, total_loss=1.0))
    assert steps_seen == [1] and len(sink.frames) == 1 and len(bar.postfix_calls) == 1


def test_record_losses_without_an_accumulator_returns_none():
    """reference :1139-1163."""
    r, _ = _runner()
    r._loss_accumulator = None
    t = opt_mod.StepTensors(step=1, style_score=torch.tensor(1.0), content_score=torch.tensor(1.0), total_loss=torch.tensor(1.0))
    assert r._record_losses(t) is None


def test_progress_postfix_falls_back_to_the_last_logged_values():
    """reference :1165-1223: empty metrics + a previous logged loss -> that loss, formatted %.4f; nothing at all -> no call."""
    from style_transfer_visualizer_amd.loss_accumulator import LoggedLoss
    bar = CountingBar()
    r, _ = _runner(progress_bar=bar)
    r._update_progress_postfix(opt_mod.StepMetrics(step=1))
    assert bar.postfix_calls == []
    r._latest_logged = LoggedLoss(step=1, style_loss=1.5, content_loss=2.5, total_loss=3.5)
    r._update_progress_postfix(opt_mod.StepMetrics(step=1))
    assert bar.postfix_calls == [{"style": "1.5000", "content": "2.5000", "loss": "3.5000"}]


def test_logging_error_callback_gets_the_csv_open_failure(tmp_path):
    """reference :882-913: a CSV path that cannot be opened -> on_logging_error(exc), the run goes on with history."""
    errors = []
    cfg = _cfg(steps=2, log_every=1, save_every=100, log_loss=str(tmp_path / "no_such_dir" / "x" / "loss.csv"))
    (tmp_path / "no_such_dir").write_text("a file where a directory is needed")
    x = _img()
    r = OptimizationRunner(TinyModel(), x, cfg, optimizer=torch.optim.Adam([x]), progress_bar=Bar(),
                           callbacks=OptimizationCallbacks(on_logging_error=errors.append))
    _, history, _ = r.run()
    assert len(errors) == 1 and isinstance(errors[0], OSError)
    assert len(history["total_loss"]) == 2

"""Optimisation loop: closure, step accounting, loss logging, frame cadence.

Mirrors reference optimization.py:89-529 (``OptimizationRunner`` kwargs,
``run()`` return triple, 1-based step ids, one accepted step per
``optimizer.step`` no matter how often the closure runs, frame/CSV cadence,
error texts).  Differences, all on the device side:

* when the model offers ``loss_and_grad`` (the HIP ``StyleContentModel``), a
  step is one fused command-buffer launch that leaves ``input_img.grad``
  populated - no autograd graph, no tiny reduction kernels;
* the default optimizer for a GPU image is the device-resident ``HipLBFGS``;
* finite-ness of the losses is checked for EVERY step, like the reference does
  (optimization.py:375-391), but at the ``log_every`` cadence: each flush copies the
  scores of all steps since the previous flush out of the device ring in one
  transfer (``LossAccumulator.drain_unchecked``) instead of three blocking
  ``if not torch.isfinite(t)`` reads per step; same warning texts and step ids, also
  in CSV mode (where the ring is kept for this purpose only).
"""
from __future__ import annotations

import logging
import math
import time
from collections.abc import Callable, Mapping
from dataclasses import dataclass
from typing import Protocol

import numpy as np
import torch
from torch import nn
from torch.optim import Optimizer

from . import image_io
from .constants import CSV_LOGGING_RECOMMENDED_STEPS
from .logging_utils import logger
from .loss_accumulator import DEFAULT_HISTORY_CAPACITY, LoggedLoss, LossAccumulator
from .loss_logger import LossCSVLogger
from .optimizers import HipAdam, HipLBFGS, make_lbfgs, single_evaluation
from .type_defs import LossHistory


class ProgressReporter(Protocol):
    """The slice of tqdm's interface the runner uses."""

    def update(self, n: float | None = 1) -> bool | None: ...

    def set_postfix(self, ordered_dict: Mapping[str, object] | None = None,
                    refresh: bool | None = True, **kwargs: object) -> None: ...  # noqa: FBT001, FBT002

    def close(self) -> None: ...


class FrameSink(Protocol):
    """Anything frames can be appended to (video writer, GIF collector, test double)."""

    def append_data(self, frame: np.ndarray) -> None: ...

    def close(self) -> None: ...


@dataclass(slots=True)
class StepMetrics:
    """Host-side view of one step; losses are present only on logging steps."""

    step: int
    style_loss: float | None = None
    content_loss: float | None = None
    total_loss: float | None = None

    @property
    def has_values(self) -> bool:
        return None not in (self.style_loss, self.content_loss, self.total_loss)


@dataclass(slots=True)
class StepTensors:
    """Device-side loss scalars of one closure evaluation."""

    step: int
    style_score: torch.Tensor
    content_score: torch.Tensor
    total_loss: torch.Tensor


@dataclass(slots=True)
class OptimizationCallbacks:
    """Optional hooks."""

    on_step_start: Callable[[int], None] | None = None
    on_step_end: Callable[[StepMetrics], None] | None = None
    on_video_frame: Callable[[np.ndarray, int], None] | None = None
    on_logging_error: Callable[[Exception], None] | None = None


def append_crossfade(sink: FrameSink, start: np.ndarray, end: np.ndarray, frames: int) -> None:
    """Linear blend from ``start`` to ``end`` over ``frames`` frames (endpoints excluded)."""
    if frames <= 0 or start.shape != end.shape:
        return
    a, b = start.astype(np.float32), end.astype(np.float32)
    for k in range(1, frames + 1):
        w = k / (frames + 1)
        sink.append_data(((1.0 - w) * a + w * b).round().clip(0, 255).astype(np.uint8))


class OptimizationRunner:
    """Drive ``optimizer.step(closure)`` for ``config.optimization.steps`` accepted steps."""

    def __init__(  # noqa: PLR0913
        self,
        model: nn.Module,
        input_img: torch.Tensor,
        config,
        *,
        optimizer: Optimizer | None = None,
        optimizer_factory: Callable[[torch.Tensor], Optimizer] | None = None,
        progress_bar: ProgressReporter | None = None,
        callbacks: OptimizationCallbacks | None = None,
        video_writer: FrameSink | None = None,
        gif_collector: FrameSink | None = None,
        intro_last_frame: np.ndarray | None = None,
        intro_crossfade_frames: int = 0,
    ) -> None:
        if optimizer is not None and optimizer_factory is not None:
            msg = "Provide either optimizer or optimizer_factory, not both."
            raise ValueError(msg)
        self.model = model
        self.input_img = input_img
        self.config = config
        self.optimizer = optimizer if optimizer is not None else self._build_optimizer(optimizer_factory)
        self._progress_bar = progress_bar
        self._owns_progress_bar = False
        self.callbacks = callbacks or OptimizationCallbacks()
        self.video_writer = video_writer
        self.gif_collector = gif_collector
        self.intro_last_frame = intro_last_frame
        self.intro_crossfade_frames = intro_crossfade_frames
        self.intro_transition_done = intro_last_frame is None

        self.loss_logger: LossCSVLogger | None = None
        self._loss_accumulator: LossAccumulator | None = None
        self._latest_logged: LoggedLoss | None = None
        self._last_loss_tensor: torch.Tensor | None = None
        self._configure_logging()

        self._step_index = 0
        self._active_step_idx: int | None = None
        self._pending_step_tensors: StepTensors | None = None
        self._closure_calls = 0
        self._fused = callable(getattr(model, "loss_and_grad", None))
        self._live_scores: bool | None = None
        self._model_kwargs: frozenset[str] = frozenset()
        # optimizers that evaluate the closure exactly once per step: the model may then keep the loss history
        # itself (one ring slot per evaluation, written by the kernel that combines the scores)
        self._single_eval = isinstance(self.optimizer, (HipLBFGS, HipAdam))
        self._producer_logged = False

    # ------------------------------------------------------------------ properties
    @property
    def progress_bar(self) -> ProgressReporter:
        if self._progress_bar is None:
            msg = "Progress bar not initialized. Call run() before use."
            raise RuntimeError(msg)
        return self._progress_bar

    @property
    def total_steps(self) -> int:
        return self.config.optimization.steps

    # ------------------------------------------------------------------------- run
    def run(self) -> tuple[torch.Tensor, LossHistory, float]:
        """Run to completion; returns (image, loss history or {}, elapsed seconds)."""
        if self._progress_bar is None:
            from tqdm import tqdm  # noqa: PLC0415
            self._progress_bar = tqdm(total=self.total_steps, desc="Style Transfer")
            self._owns_progress_bar = True
        started = time.time()
        side = self._enter_side_stream()
        try:
            while self._step_index < self.total_steps:
                step_idx = self._step_index + 1
                if self.callbacks.on_step_start is not None:
                    self.callbacks.on_step_start(step_idx)
                self._active_step_idx = step_idx
                self._pending_step_tensors = None
                try:
                    self.optimizer.step(self._closure)  # type: ignore[arg-type]
                finally:
                    self._active_step_idx = None
                recorded = self._pending_step_tensors
                if recorded is None:
                    msg = f"Optimizer closure did not record metrics for step {step_idx}"
                    raise RuntimeError(msg)
                self._finalize_step(recorded)
                self._pending_step_tensors = None
        finally:
            self._leave_side_stream(side)
            self._cleanup()
        elapsed = time.time() - started
        self._log_optimization_summary()
        acc = self._loss_accumulator
        self._audit_unchecked_steps()            # steps after the last logging point
        history: LossHistory = acc.export_history() if (acc is not None and acc.tracks_history) else {}
        return self.input_img, history, elapsed

    # ------------------------------------------------------------------- set-up
    def _enter_side_stream(self):
        """GPU runs use a non-default HIP stream so the fused step can replay as a hipGraph
        (the legacy default stream cannot be captured)."""
        if not (self._fused and self.input_img.is_cuda):
            return None
        side = torch.cuda.Stream(device=self.input_img.device)
        side.wait_stream(torch.cuda.current_stream(self.input_img.device))
        ctx = torch.cuda.stream(side)
        ctx.__enter__()
        return side, ctx

    def _leave_side_stream(self, entered) -> None:
        if entered is None:
            return
        side, ctx = entered
        ctx.__exit__(None, None, None)
        torch.cuda.current_stream(self.input_img.device).wait_stream(side)

    def _build_optimizer(self, optimizer_factory: Callable[[torch.Tensor], Optimizer] | None) -> Optimizer:
        if optimizer_factory is not None:
            return optimizer_factory(self.input_img)
        oc = self.config.optimization
        return make_lbfgs(self.input_img, lr=oc.lr, max_iter=oc.lbfgs_max_iter, max_eval=oc.lbfgs_max_eval)

    def _configure_logging(self) -> None:
        """CSV logger when requested (history then stays off), else a capped in-memory history."""
        log_path = self.config.output.log_loss
        log_every = self.config.output.log_every
        steps = self.total_steps
        track_history = True
        if log_path:
            try:
                self.loss_logger = LossCSVLogger(log_path, log_every)
                logger.info("Loss CSV logging enabled: %s", log_path)
                track_history = False
            except OSError as exc:
                logger.error("Failed to initialize CSV logging: %s", exc)
                if self.callbacks.on_logging_error is not None:
                    self.callbacks.on_logging_error(exc)
        capacity = min(steps, DEFAULT_HISTORY_CAPACITY)
        self._loss_accumulator = LossAccumulator(
            log_every=log_every, history_capacity=capacity, track_history=track_history,
            device=self.input_img.device, dtype=self.input_img.dtype,
            audit_ring=True)      # CSV mode: the ring is still kept, for the per-step finite checks at the flush
        if track_history and steps > capacity:
            logger.warning(
                "Long run detected (%d steps). In-memory loss history is "
                "capped at %d entries; enable --log-loss for a full CSV.", steps, capacity)
        elif track_history and steps > CSV_LOGGING_RECOMMENDED_STEPS:
            logger.warning(
                "Long run detected (%d steps). Consider enabling "
                "--log-loss to capture every step.", steps)

    # ------------------------------------------------------------------- closure
    @single_evaluation
    def _closure(self) -> torch.Tensor:
        """Closure handed to the optimizer; may run several times per step (each call evaluates the model once:
        ``HipLBFGS`` may append its update to that evaluation's launch)."""
        self._closure_calls += 1
        if self._step_index >= self.total_steps:
            return self._final_loss_tensor()
        step_idx = self._active_step_idx or (self._step_index + 1)
        tensors = self._run_single_step(step_idx)
        self._pending_step_tensors = tensors
        return tensors.total_loss

    def _run_single_step(self, step_idx: int) -> StepTensors:
        """One forward + backward; leaves d(total)/d(image) in ``input_img.grad``."""
        oc = self.config.optimization
        self.optimizer.zero_grad()
        if self._fused:
            style_score, content_score, loss = self._fused_eval(oc.style_w, oc.content_w)
        else:
            style_losses, content_losses = self.model(self.input_img)
            zero = torch.zeros((), device=self.input_img.device, dtype=self.input_img.dtype)
            style_score = torch.stack(style_losses).sum() if style_losses else zero
            content_score = torch.stack(content_losses).sum() if content_losses else zero
            loss = oc.style_w * style_score + oc.content_w * content_score
            loss.backward()
            self._check_finite(style_score, content_score, loss, step_idx)
        return StepTensors(step=step_idx, style_score=style_score, content_score=content_score, total_loss=loss)

    def _fused_eval(self, style_w: float, content_w: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """The model's fused step; its scores are recorded before the next evaluation, so views of the
        live score buffer are enough where the model offers them (models without the keyword: plain call)."""
        if self._live_scores is None:
            import inspect
            try:
                self._model_kwargs = frozenset(inspect.signature(self.model.loss_and_grad).parameters)
            except (TypeError, ValueError):
                self._model_kwargs = frozenset()
            self._live_scores = "live_scores" in self._model_kwargs
        kwargs = {}
        self._producer_logged = False
        if self._live_scores:
            kwargs["live_scores"] = True
        if self._single_eval and "score_log" in self._model_kwargs and self._loss_accumulator is not None:
            log = self._loss_accumulator.device_log()
            if log is not None:
                kwargs["score_log"] = log
                self._producer_logged = True
        return self.model.loss_and_grad(self.input_img, style_w, content_w, **kwargs)

    def _final_loss_tensor(self) -> torch.Tensor:
        if self._last_loss_tensor is not None:
            return self._last_loss_tensor.detach()
        return torch.zeros((), device=self.input_img.device, dtype=self.input_img.dtype)

    # ----------------------------------------------------------------- per step
    def _finalize_step(self, tensors: StepTensors) -> None:
        self._step_index = tensors.step
        self._last_loss_tensor = tensors.total_loss.detach()      # a view of the live scores on the fused path: "last" by construction
        logged = self._record_losses(tensors)
        if logged is not None:
            self._latest_logged = logged
            self._audit_unchecked_steps()
            metrics = StepMetrics(logged.step, logged.style_loss, logged.content_loss, logged.total_loss)
        else:
            metrics = StepMetrics(step=tensors.step)
        self._maybe_write_video_frame(metrics)
        self.progress_bar.update(1)
        if self.callbacks.on_step_end is not None:
            self.callbacks.on_step_end(metrics)

    def _record_losses(self, tensors: StepTensors) -> LoggedLoss | None:
        if self._loss_accumulator is None:
            return None
        by_producer, self._producer_logged = self._producer_logged, False
        logged = self._loss_accumulator.accumulate(
            tensors.step, tensors.style_score, tensors.content_score, tensors.total_loss, logged_by_producer=by_producer)
        if logged is not None and self.loss_logger is not None:
            self.loss_logger.log(logged.step, logged.style_loss, logged.content_loss, logged.total_loss)
        return logged

    def _check_finite(self, style_score: torch.Tensor, content_score: torch.Tensor,
                      total_loss: torch.Tensor, step_idx: int) -> None:
        """Autograd path: warn like the reference does (blocking reads)."""
        self._warn_nonfinite(float(style_score.detach()), float(content_score.detach()),
                             float(total_loss.detach()), step_idx)

    def _warn_nonfinite(self, style: float, content: float, total: float, step_idx: int) -> None:
        if not math.isfinite(style):
            logger.warning("Non-finite style score at step %d", step_idx)
        if not math.isfinite(content):
            logger.warning("Non-finite content score at step %d", step_idx)
        if not math.isfinite(total):
            logger.warning("Non-finite total loss at step %d, using previous loss", step_idx)
        if logger.isEnabledFor(logging.DEBUG):
            logger.debug("Step %d: Style %.4e, Content %.4e, Total %.4e", step_idx, style, content, total)

    def _audit_unchecked_steps(self) -> None:
        """Fused path: the reference's per-step finite checks (optimization.py:375-391), run for every step
        since the previous logging point from one copy of the device ring.  The autograd path has checked
        each step as it went (`_check_finite`)."""
        acc = self._loss_accumulator
        if not self._fused or acc is None:
            return
        for step, style, content, total in acc.drain_unchecked():
            self._warn_nonfinite(style, content, total, step)

    def _maybe_write_video_frame(self, metrics: StepMetrics) -> None:
        """Every ``save_every`` accepted steps, hand a uint8 HWC frame to the sinks."""
        vc = self.config.video
        step_idx = metrics.step
        if (not vc.save_every or step_idx % vc.save_every != 0
                or (self.video_writer is None and self.gif_collector is None)):
            return
        with torch.no_grad():
            frame = image_io.frame_uint8(self.input_img, normalize=self.config.optimization.normalize)
            if frame is None:
                return
        if self.intro_last_frame is not None and not self.intro_transition_done:
            if self.video_writer is not None and vc.intro_enabled:
                append_crossfade(self.video_writer, self.intro_last_frame, frame, self.intro_crossfade_frames)
            if self.gif_collector is not None and vc.gif_include_intro:
                append_crossfade(self.gif_collector, self.intro_last_frame, frame, self.intro_crossfade_frames)
            self.intro_transition_done = True
            self.intro_last_frame = None
        for sink in (self.video_writer, self.gif_collector):
            if sink is not None:
                sink.append_data(frame)
        self._update_progress_postfix(metrics)
        if self.callbacks.on_video_frame is not None:
            self.callbacks.on_video_frame(frame, step_idx)

    def _update_progress_postfix(self, metrics: StepMetrics) -> None:
        src = metrics if metrics.has_values else self._latest_logged
        if src is None:
            return
        shown = {"style": src.style_loss, "content": src.content_loss, "loss": src.total_loss}
        postfix = {k: f"{v:.4f}" for k, v in shown.items() if v is not None}
        if postfix:
            self.progress_bar.set_postfix(postfix)

    # ------------------------------------------------------------------ wrap-up
    def _log_optimization_summary(self) -> None:
        if self._step_index <= 0:
            return
        logger.info(
            "Optimization finished with %d accepted steps and %d closure "
            "evaluations (%.2f closures/step).",
            self._step_index, self._closure_calls, self._closure_calls / self._step_index)

    def _cleanup(self) -> None:
        if self.loss_logger is not None:
            self.loss_logger.close()
        if self._owns_progress_bar and self._progress_bar is not None:
            self._progress_bar.close()

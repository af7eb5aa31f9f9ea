"""Device-side loss bookkeeping with batched host synchronisation.

Behaviour follows reference loss_accumulator.py:26-213: every step's
(style, content, total) scalars are appended to a device ring buffer of
``history_capacity`` entries; Python floats are produced only when
``step % log_every == 0`` (or ``force``), so the optimisation loop never waits
for the GPU in between.  Here the three series share one ``[3, capacity]``
tensor and a flush is a single device-to-host copy.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

DEFAULT_HISTORY_CAPACITY = 2048
_KEYS = ("style_loss", "content_loss", "total_loss")


@dataclass(slots=True)
class LoggedLoss:
    """Loss scalars that have been copied to the host."""

    step: int
    style_loss: float
    content_loss: float
    total_loss: float


class LossAccumulator:
    """Ring-buffered loss history on the device, floats on demand."""

    def __init__(self, *, log_every: int, history_capacity: int | None, track_history: bool,
                 device: torch.device, dtype: torch.dtype, audit_ring: bool = False) -> None:
        """``audit_ring`` (extension): keep the device ring even when ``track_history`` is off (CSV mode), so
        that every step's scores can still be inspected at the next flush (:meth:`drain_unchecked`) - the
        reference checks each step's losses for non-finite values as it goes (optimization.py:375-391);
        ``export_history`` stays empty."""
        self._log_every = max(1, log_every)
        self._capacity = max(1, history_capacity or DEFAULT_HISTORY_CAPACITY)
        self._track = track_history
        self._device = device
        # fp16 images keep fp16 buffers; everything else (incl. bf16) logs in fp32
        self._buffer_dtype = torch.float16 if dtype == torch.float16 else torch.float32
        self._ring: torch.Tensor | None = None
        self._ringed = track_history or audit_ring
        if self._ringed:
            self._ring = torch.empty(3, self._capacity, dtype=self._buffer_dtype, device=device)
        self._unchecked: list[int] = []   # step ids recorded since the last drain_unchecked()
        self._next = 0            # slot the next record goes to
        self._count = 0           # valid records in the ring
        self._records = 0         # records ever written
        self._pending: tuple[int, torch.Tensor] | None = None
        self._last: LoggedLoss | None = None
        self._counter: torch.Tensor | None = None      # device-side record count, once a producer logs for us

    def device_log(self) -> tuple[torch.Tensor, torch.Tensor] | None:
        """(ring [3, capacity] fp32, counter [1] int32) for a producer that appends each step's scores to the
        history itself (``stv_loss_combine_log``: slot = counter % capacity, counter += 1) - the per-step copy
        kernel then disappears from the step.  None when there is no fp32 device ring to share.  Records that
        still arrive through :meth:`accumulate` without ``logged_by_producer`` keep the counter in step."""
        if not self._ringed or self._ring is None or self._ring.dtype != torch.float32 or not self._ring.is_cuda:
            return None
        if self._counter is None:
            self._counter = torch.full((1,), self._records, dtype=torch.int32, device=self._ring.device)
        return self._ring, self._counter

    @property
    def capacity(self) -> int:
        """Maximum number of history entries kept in memory."""
        return self._capacity

    @property
    def tracks_history(self) -> bool:
        """True when per-step history is being recorded."""
        return self._track

    @property
    def history_truncated(self) -> bool:
        """True once the ring has overwritten its oldest entries."""
        return self._track and self._records > self._capacity

    def accumulate(self, step_idx: int, style_loss: torch.Tensor, content_loss: torch.Tensor,
                   total_loss: torch.Tensor, *, force: bool = False, logged_by_producer: bool = False,
                   ) -> LoggedLoss | None:
        """Record one step; return host scalars only at the logging cadence.  ``logged_by_producer``: the
        evaluation that produced these scores already appended them to the ring (:meth:`device_log`)."""
        triple = self._adjacent(style_loss, content_loss, total_loss)
        if triple is None:
            triple = torch.stack((style_loss.detach().reshape(()), content_loss.detach().reshape(()),
                                  total_loss.detach().reshape(())))
        self._pending = (step_idx, triple)
        if self._ringed:
            if self._ring is None:
                msg = "History buffers are uninitialized."
                raise RuntimeError(msg)
            self._unchecked.append(step_idx)
            if len(self._unchecked) > 2 * self._capacity:       # nobody drains (autograd path): stay bounded
                del self._unchecked[:-self._capacity]
            if not (logged_by_producer and self._counter is not None):
                self._ring[:, self._next] = triple.to(dtype=self._buffer_dtype, device=self._device)
                if self._counter is not None:
                    self._counter += 1
            self._next = (self._next + 1) % self._capacity
            self._count = min(self._count + 1, self._capacity)
            self._records += 1
        if force or step_idx % self._log_every == 0:
            return self._sync_pending()
        return None

    @staticmethod
    def _adjacent(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor | None:
        """The three scalars as one [3] view when they already sit side by side in one buffer (the
        fused step returns views of its score vector): saves the stack kernel."""
        try:
            if not (a.numel() == b.numel() == c.numel() == 1 and a.dtype == b.dtype == c.dtype
                    and a.device == b.device == c.device):
                return None
            base = a.untyped_storage().data_ptr()
            if b.untyped_storage().data_ptr() != base or c.untyped_storage().data_ptr() != base:
                return None
            o = a.storage_offset()
            if b.storage_offset() != o + 1 or c.storage_offset() != o + 2:
                return None
            return torch.as_strided(a.detach(), (3,), (1,), o)
        except RuntimeError:
            return None

    def latest(self) -> LoggedLoss | None:
        """Most recent host-synced scalars."""
        return self._last

    def export_history(self) -> dict[str, list[float]]:
        """Chronological bounded history as Python lists."""
        if not self._track or self._count == 0 or self._ring is None:
            return {k: [] for k in _KEYS}
        start = (self._next - self._count) % self._capacity
        if start + self._count <= self._capacity:
            window = self._ring[:, start:start + self._count]
        else:
            window = torch.cat((self._ring[:, start:], self._ring[:, :self._count - (self._capacity - start)]), dim=1)
        rows = window.cpu().tolist()
        return dict(zip(_KEYS, rows, strict=True))

    def drain_unchecked(self) -> list[tuple[int, float, float, float]]:
        """(step, style, content, total) of every step recorded since the previous call, oldest first (at most
        ``capacity`` of them) - one device-to-host copy.  Lets the caller examine EVERY step's scores at the
        logging cadence without a host synchronisation per step."""
        steps, self._unchecked = self._unchecked, []
        if not steps or self._ring is None:
            return []
        steps = steps[-min(self._count, self._capacity):]
        k = len(steps)
        start = (self._next - k) % self._capacity
        if start + k <= self._capacity:
            window = self._ring[:, start:start + k]
        else:
            window = torch.cat((self._ring[:, start:], self._ring[:, :k - (self._capacity - start)]), dim=1)
        s, c, t = window.cpu().tolist()
        return [(step, float(a), float(b), float(d)) for step, a, b, d in zip(steps, s, c, t, strict=True)]

    def _sync_pending(self) -> LoggedLoss | None:
        if self._pending is None:
            return None
        step, triple = self._pending
        s, c, t = self._to_floats(triple)
        self._last = LoggedLoss(step=step, style_loss=s, content_loss=c, total_loss=t)
        return self._last

    def _to_floats(self, triple: torch.Tensor) -> tuple[float, float, float]:
        s, c, t = triple.tolist()          # the only host synchronisation
        return float(s), float(c), float(t)

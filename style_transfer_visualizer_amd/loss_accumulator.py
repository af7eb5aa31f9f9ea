"""Device-side loss bookkeeping with batched host synchronisation.

Behaviour follows reference loss_accumulator.py:26-213: every step's
(style, content, total) scalars are appended to a device ring buffer of
``history_capacity`` entries; Python floats are produced only when
``step % log_every == 0`` (or ``force``), so the optimisation loop never waits
for the GPU in between.  Here the three series share one ``[3, capacity]``
tensor and a flush is a single device-to-host copy.
"""
from __future__ import annotations

import os
import time
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
        # once a producer logs for us on a GPU: the ring moves to host memory the GPU writes into (ops.HostMailbox) and
        # ``_seq`` is the record count the producer publishes behind each record - a flush then reads host memory as
        # soon as the step's combine kernel has run, instead of copying from the device behind the whole step
        self._box = None
        self._seq: torch.Tensor | None = None
        self._seq_np = None
        self._pending_record: int | None = None        # the pending step's record number when the producer logged it

    def device_log(self) -> tuple[torch.Tensor, ...] | None:
        """(ring [3, capacity] fp32, counter [1] int32[, seq [1] int32]) for a producer that appends each step's
        scores to the history itself (``stv_loss_combine_log``: slot = counter % capacity, counter += 1) - the
        per-step copy kernel then disappears from the step.  With ``seq`` the ring is HOST memory mapped into the
        device (the producer publishes the record count there behind each record).  None when there is no fp32
        ring to share.  Records that still arrive through :meth:`accumulate` without ``logged_by_producer`` keep
        the counters in step."""
        if not self._ringed or self._ring is None or self._ring.dtype != torch.float32:
            return None
        if self._box is None and not self._ring.is_cuda:
            return None
        if self._counter is None:
            self._counter = torch.full((1,), self._records, dtype=torch.int32, device=self._device)
            self._adopt_host_ring()
        if self._seq is not None:
            return self._ring, self._counter, self._seq
        return self._ring, self._counter

    def _adopt_host_ring(self) -> None:
        """Move the ring into pinned, device-mapped host memory (STV_HOST_LOG=0: keep it on the device)."""
        if os.environ.get("STV_HOST_LOG", "1") == "0" or self._ring is None:
            return
        from . import ops  # noqa: PLC0415
        import numpy as np  # noqa: PLC0415
        try:
            box = ops.HostMailbox(64 + 3 * self._capacity * 4)
        except RuntimeError:          # no pinned memory to be had: the ring stays on the device (copied at the logging point)
            return
        ring = box.tensor(torch.float32, (3, self._capacity), offset=64)
        if self._records:
            ring.copy_(self._ring.cpu())
        seq = box.tensor(torch.int32, (1,), offset=0)
        seq[0] = self._records
        self._box, self._ring, self._seq = box, ring, seq
        self._seq_np = box.array(np.uint32, 1, 0)

    def _wait_published(self, records: int) -> None:
        """Until the producer has published ``records`` records (the combine kernel of that step has run)."""
        if self._seq_np is None or int(self._seq_np[0]) >= records:
            return
        t0 = time.perf_counter()
        spins = 0
        while int(self._seq_np[0]) < records:
            spins += 1
            if spins > 300:            # (then in short sleeps: several runners may share this interpreter - style_transfer_batch)
                time.sleep(2e-5)
                if time.perf_counter() - t0 > 120.0:
                    torch.cuda.synchronize(self._device)        # surfaces a device fault, if that is the reason
                    if int(self._seq_np[0]) >= records:
                        return
                    msg = f"loss history: record {records} was never published by the device (seen {int(self._seq_np[0])})"
                    raise RuntimeError(msg)

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
        self._pending_record = None
        if self._ringed:
            if self._ring is None:
                msg = "History buffers are uninitialized."
                raise RuntimeError(msg)
            self._unchecked.append(step_idx)
            if len(self._unchecked) > 2 * self._capacity:       # nobody drains (autograd path): stay bounded
                del self._unchecked[:-self._capacity]
            if not (logged_by_producer and self._counter is not None):
                if self._seq is not None:
                    self._wait_published(self._records)          # (never overtake the producer's own records)
                self._ring[:, self._next] = triple.to(dtype=self._buffer_dtype, device=self._ring.device)
                if self._counter is not None:
                    self._counter += 1
                if self._seq is not None:
                    self._seq[0] = self._records + 1
            elif self._seq is not None:
                self._pending_record = self._records + 1
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
        self._wait_published(self._records)
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
        self._wait_published(self._records)
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
        if self._pending_record is not None and self._pending_record == self._records and self._ring is not None:
            # the producer wrote this record into the host ring: wait for ITS combine kernel only
            self._wait_published(self._pending_record)
            slot = (self._next - 1) % self._capacity
            s, c, t = (float(v) for v in self._ring[:, slot].tolist())
        else:
            s, c, t = self._to_floats(triple)
        self._last = LoggedLoss(step=step, style_loss=s, content_loss=c, total_loss=t)
        return self._last

    def _to_floats(self, triple: torch.Tensor) -> tuple[float, float, float]:
        s, c, t = triple.tolist()          # the only host synchronisation
        return float(s), float(c), float(t)

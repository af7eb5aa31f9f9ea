"""LossAccumulator: flush cadence, ring-buffer contents, chronological export (CPU tensors)."""
from __future__ import annotations

import torch

from style_transfer_visualizer_amd.loss_accumulator import DEFAULT_HISTORY_CAPACITY, LossAccumulator

CPU = torch.device("cpu")


def _acc(**kw):
    base = dict(log_every=2, history_capacity=4, track_history=True, device=CPU, dtype=torch.float32)
    base.update(kw)
    return LossAccumulator(**base)


def _push(acc, step, force=False):
    t = torch.tensor(float(step))
    return acc.accumulate(step, t, t * 0.5, t * 1.5, force=force)


def test_scalars_only_on_logging_steps():
    acc = _acc(log_every=3, history_capacity=16)
    out = [_push(acc, s) for s in range(1, 8)]
    assert [o is not None for o in out] == [False, False, True, False, False, True, False]
    assert out[2].step == 3 and out[2].style_loss == 3.0 and out[2].content_loss == 1.5 and out[2].total_loss == 4.5
    assert acc.latest().step == 6


def test_force_flushes_off_cadence():
    acc = _acc(log_every=10)
    assert _push(acc, 1) is None
    logged = _push(acc, 2, force=True)
    assert logged is not None and logged.step == 2


def test_ring_keeps_the_newest_entries():
    # reference semantics (tests/test_loss_accumulator.py:48-70): 6 pushes, capacity 4 -> [3,4,5,6]
    acc = _acc()
    for s in range(1, 7):
        _push(acc, s)
    hist = acc.export_history()
    assert hist["style_loss"] == [3.0, 4.0, 5.0, 6.0]
    assert hist["content_loss"] == [1.5, 2.0, 2.5, 3.0]
    assert hist["total_loss"] == [4.5, 6.0, 7.5, 9.0]
    assert acc.history_truncated and acc.capacity == 4


def test_wrapped_window_is_chronological():
    acc = _acc(history_capacity=3)
    for s in range(1, 5):
        _push(acc, s)
    assert acc.export_history()["style_loss"] == [2.0, 3.0, 4.0]


def test_not_truncated_until_overwrite():
    acc = _acc(history_capacity=4)
    for s in range(1, 5):
        _push(acc, s)
    assert not acc.history_truncated
    assert acc.export_history()["style_loss"] == [1.0, 2.0, 3.0, 4.0]


def test_untracked_history_is_empty():
    acc = _acc(track_history=False)
    _push(acc, 2)
    assert not acc.tracks_history
    assert acc.export_history() == {"style_loss": [], "content_loss": [], "total_loss": []}
    assert acc.latest().step == 2


def test_defaults_and_edge_values():
    acc = LossAccumulator(log_every=0, history_capacity=None, track_history=True, device=CPU, dtype=torch.bfloat16)
    assert acc.capacity == DEFAULT_HISTORY_CAPACITY
    assert _push(acc, 1) is not None          # log_every clamps to 1
    assert acc.export_history()["total_loss"] == [1.5]
    half = LossAccumulator(log_every=1, history_capacity=2, track_history=True, device=CPU, dtype=torch.float16)
    assert half._ring.dtype == torch.float16  # fp16 images keep fp16 buffers, everything else fp32
    assert acc._ring.dtype == torch.float32


def test_audit_ring_keeps_scores_without_tracking_history():
    """audit_ring: per-step scores stay inspectable at the flush although the history itself is off (CSV mode);
    the reference-visible surface (tracks_history, export_history, history_truncated) is unchanged."""
    acc = LossAccumulator(log_every=3, history_capacity=4, track_history=False, device=torch.device("cpu"),
                          dtype=torch.float32, audit_ring=True)
    for step in range(1, 8):
        acc.accumulate(step, torch.tensor(float(step)), torch.tensor(0.5), torch.tensor(2.0 * step))
        if step == 3:
            assert acc.drain_unchecked() == [(1, 1.0, 0.5, 2.0), (2, 2.0, 0.5, 4.0), (3, 3.0, 0.5, 6.0)]
    # four more records since the drain: the ring (capacity 4) wrapped, all four are still there
    assert acc.drain_unchecked() == [(4, 4.0, 0.5, 8.0), (5, 5.0, 0.5, 10.0), (6, 6.0, 0.5, 12.0), (7, 7.0, 0.5, 14.0)]
    assert acc.drain_unchecked() == []
    assert not acc.tracks_history and not acc.history_truncated
    assert acc.export_history() == {"style_loss": [], "content_loss": [], "total_loss": []}


class _FakeMailbox:
    """Stands in for ops.HostMailbox on a machine without a GPU: plain host memory, same views."""

    def __init__(self, nbytes: int) -> None:
        import ctypes
        self._buf = (ctypes.c_char * nbytes)()
        self.ptr, self.nbytes = ctypes.addressof(self._buf), nbytes

    def tensor(self, dtype, shape, offset=0):
        count = 1
        for d in shape:
            count *= int(d)
        return torch.frombuffer(self._buf, dtype=dtype, count=count, offset=offset).view(*shape)

    def array(self, dtype, count, offset=0):
        import numpy as np
        return np.frombuffer(self._buf, dtype=dtype, count=count, offset=offset)


def _host_ring_accumulator(monkeypatch, *, log_every=4, capacity=8, prior=0):
    from style_transfer_visualizer_amd import ops
    monkeypatch.setattr(ops, "HostMailbox", _FakeMailbox)
    acc = LossAccumulator(log_every=log_every, history_capacity=capacity, track_history=True,
                          device=torch.device("cpu"), dtype=torch.float32)
    for k in range(prior):                       # records that arrived before any producer offered to log
        acc.accumulate(k + 1, *(torch.tensor(float(10 * (k + 1) + j)) for j in range(3)))
    acc._counter = torch.full((1,), acc._records, dtype=torch.int32)      # what device_log() does on a GPU
    acc._adopt_host_ring()
    assert acc._box is not None and acc.device_log() is not None and len(acc.device_log()) == 3
    return acc


def _produce(acc, record, values):
    """What stv_loss_combine_log does: the record into its slot, then the count behind it."""
    ring, counter, seq = acc.device_log()
    slot = (record - 1) % ring.shape[1]
    for j, v in enumerate(values):
        ring[j, slot] = v
    counter += 1
    seq[0] = record


def test_host_ring_flush_reads_what_the_producer_published(monkeypatch):
    """The loss history ring in host memory written by the producer (stv_loss_combine_log with log_seq): a logging
    point returns the producer's record without touching the live score tensors, history and audit read the same
    ring, and records that arrived before the ring moved are kept."""
    acc = _host_ring_accumulator(monkeypatch, log_every=4, capacity=8, prior=2)
    live = torch.zeros(3)                        # stands for the engine's score buffer: already overwritten by a later step
    for step in range(3, 9):
        _produce(acc, step, (step + 0.25, step + 0.5, step + 0.75))
        out = acc.accumulate(step, live[0], live[1], live[2], logged_by_producer=True)
        if step % 4 == 0:
            assert out is not None and (out.step, out.style_loss, out.content_loss, out.total_loss) == (
                step, step + 0.25, step + 0.5, step + 0.75)
        else:
            assert out is None
    hist = acc.export_history()
    assert hist["style_loss"] == [10.0, 20.0, 3.25, 4.25, 5.25, 6.25, 7.25, 8.25]
    assert [r[0] for r in acc.drain_unchecked()] == list(range(1, 9))


def test_host_ring_flush_waits_for_the_producer(monkeypatch):
    """The host may be ahead of the device: a logging point blocks until the record count says its step's combine
    kernel has run (here a thread publishes 30 ms late), and a record that arrives WITHOUT the producer keeps ring,
    device counter and published count in step."""
    import threading
    import time
    acc = _host_ring_accumulator(monkeypatch, log_every=1, capacity=4)
    t = threading.Timer(0.03, _produce, args=(acc, 1, (1.0, 2.0, 3.0)))
    t0 = time.perf_counter()
    t.start()
    out = acc.accumulate(1, torch.tensor(9.0), torch.tensor(9.0), torch.tensor(9.0), logged_by_producer=True)
    t.join()
    assert time.perf_counter() - t0 >= 0.025
    assert (out.style_loss, out.content_loss, out.total_loss) == (1.0, 2.0, 3.0)
    out = acc.accumulate(2, torch.tensor(4.0), torch.tensor(5.0), torch.tensor(6.0))       # e.g. an autograd-path step
    assert (out.style_loss, out.content_loss, out.total_loss) == (4.0, 5.0, 6.0)
    ring, counter, seq = acc.device_log()
    assert int(counter) == 2 and int(seq) == 2 and ring[:, 1].tolist() == [4.0, 5.0, 6.0]
    assert acc.export_history()["total_loss"] == [3.0, 6.0]

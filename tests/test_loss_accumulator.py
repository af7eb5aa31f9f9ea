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

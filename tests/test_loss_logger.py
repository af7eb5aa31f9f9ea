"""LossCSVLogger: on-disk format and cadence."""
from __future__ import annotations

import csv

from style_transfer_visualizer_amd.loss_logger import LossCSVLogger


def _rows(path):
    with open(path, newline="", encoding="utf-8") as f:
        return list(csv.reader(f))


def test_header_is_flushed_on_open(tmp_path):
    p = tmp_path / "sub" / "loss.csv"
    logger = LossCSVLogger(p, log_every=2)
    assert _rows(p) == [["step", "style_loss", "content_loss", "total_loss"]]
    logger.close()


def test_rows_only_on_cadence(tmp_path):
    p = tmp_path / "loss.csv"
    with LossCSVLogger(p, log_every=2) as logger:
        for step in range(1, 6):
            logger.log(step, 1.0, 0.5, 1.5)
        assert _rows(p)[1:] == [["2", "1.0", "0.5", "1.5"], ["4", "1.0", "0.5", "1.5"]]   # flushed per row
    assert logger.file.closed
    logger.close()   # idempotent


# ---- the remaining behaviours the reference's tests/test_loss_logger.py pins ------------------------------
def test_close_closes_the_file_and_is_idempotent(tmp_path):
    logger = LossCSVLogger(tmp_path / "loss.csv", log_every=1)
    assert not logger.file.closed
    logger.close()
    assert logger.file.closed
    logger.close()


def test_directory_creation_failure_propagates(tmp_path, monkeypatch):
    """reference :70-76: an OSError from creating the parent directory reaches the caller."""
    import pathlib

    import pytest

    def boom(self, *a, **k):
        raise OSError("Mocked error")
    monkeypatch.setattr(pathlib.Path, "mkdir", boom)
    with pytest.raises(OSError, match="Mocked error"):
        LossCSVLogger(tmp_path / "loss.csv", log_every=1)


def test_every_logged_row_is_flushed(tmp_path):
    """reference :79-86: flush() after each row."""
    logger = LossCSVLogger(tmp_path / "loss.csv", log_every=1)
    calls = []

    class Spy:
        def __init__(self, f):
            self._f = f
            self.closed = False

        def write(self, s):
            return self._f.write(s)

        def flush(self):
            calls.append("flush")
            self._f.flush()

        def close(self):
            self.closed = True
            self._f.close()
    real = logger.file
    logger.file = Spy(real)
    logger.writer = csv.writer(logger.file)
    logger.log(1, 1.0, 0.5, 1.5)
    assert calls == ["flush"]
    logger.close()


def test_close_without_a_file_or_with_a_closed_one_does_nothing(tmp_path):
    """reference :89-110."""
    logger = LossCSVLogger(tmp_path / "a.csv", log_every=1)
    real = logger.file
    logger.file = None
    logger.close()

    class Closed:
        closed = True

        def close(self):
            raise AssertionError("close() on a closed file")
    logger.file = Closed()
    logger.close()
    real.close()

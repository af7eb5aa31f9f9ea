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

"""CSV sink for loss scalars: ``step,style_loss,content_loss,total_loss``.

Same on-disk format and cadence as reference loss_logger.py:14-126: header
written (and flushed) on open, one row per call whose ``step`` is a multiple of
``log_every``, flushed per row.
"""
from __future__ import annotations

import csv
from pathlib import Path

HEADER = ["step", "style_loss", "content_loss", "total_loss"]


class LossCSVLogger:
    """Append loss rows to a CSV file; usable as a context manager."""

    def __init__(self, path: str | Path, log_every: int) -> None:
        self.path = Path(path)
        self.log_every = log_every
        self.path.parent.mkdir(parents=True, exist_ok=True)
        self.file = self.path.open("w", newline="", encoding="utf-8")   # OSError propagates
        self.writer = csv.writer(self.file)
        self.writer.writerow(HEADER)
        self.file.flush()

    def log(self, step: int, style_loss: float, content_loss: float, total_loss: float) -> None:
        """Write one row when ``step`` falls on the logging cadence."""
        if self.writer is None or step % self.log_every != 0:
            return
        self.writer.writerow([step, style_loss, content_loss, total_loss])
        self.file.flush()

    def close(self) -> None:
        """Close the file (idempotent)."""
        if self.file and not self.file.closed:
            self.file.close()

    def __enter__(self) -> "LossCSVLogger":
        return self

    def __exit__(self, exc_type, exc_value, traceback) -> None:
        self.close()

"""Package logger: name ``style_transfer`` (as the reference), no propagation."""
from __future__ import annotations

import logging
import sys

_FORMAT = "%(asctime)s [%(levelname)s] %(message)s"


def setup_logger(name: str = "style_transfer", level: int = logging.INFO,
                 formatter: logging.Formatter | None = None,
                 handler: logging.Handler | None = None) -> logging.Logger:
    """Return the named logger with exactly one stream handler attached."""
    log = logging.getLogger(name)
    log.setLevel(level)
    log.propagate = False
    if not log.handlers:
        h = handler or logging.StreamHandler(sys.stderr)
        h.setFormatter(formatter or logging.Formatter(_FORMAT))
        log.addHandler(h)
    return log


logger = setup_logger()

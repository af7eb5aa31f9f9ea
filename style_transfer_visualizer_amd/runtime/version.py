"""Which version string ``--version`` prints (reference runtime/version.py:12-44).

Order: an installed distribution's metadata, else the nearest ``pyproject.toml`` above this file that carries a
non-empty ``project.version``, else ``"0.0.0"``.  An unreadable pyproject ends the search with a warning.
"""
from __future__ import annotations

from importlib import metadata as importlib_metadata
from pathlib import Path

try:
    import tomllib
except ModuleNotFoundError:             # Python 3.10 (this image): the same parser under its pre-stdlib name
    import tomli as tomllib

from ..logging_utils import logger

DISTRIBUTION_NAMES = ("style-transfer-visualizer-amd", "style_transfer_visualizer_amd",
                      "style-transfer-visualizer", "style_transfer_visualizer")
FALLBACK_VERSION = "0.0.0"


def _version_in(pyproject: Path) -> str | None:
    with pyproject.open("rb") as fh:
        found = tomllib.load(fh).get("project", {}).get("version")
    return found.strip() if isinstance(found, str) and found.strip() else None


def resolve_project_version() -> str:
    for name in DISTRIBUTION_NAMES:
        try:
            return importlib_metadata.version(name)
        except importlib_metadata.PackageNotFoundError:
            pass
    for folder in Path(__file__).resolve().parents:
        candidate = folder / "pyproject.toml"
        if not candidate.is_file():
            continue
        try:
            found = _version_in(candidate)
        except OSError as exc:
            logger.warning("Error reading %s: %s", candidate, exc)
            break
        if found:
            return found
    return FALLBACK_VERSION

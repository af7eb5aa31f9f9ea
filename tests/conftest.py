"""Shared pytest configuration: markers, paths, golden-fixture helpers."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# Tests assert tolerances near fp32 rounding: keep ONE summation order per conv shape (the analytic
# tile choice) unless a test asks for the measured one (monkeypatch.setenv("STV_CONV_TUNE", "1")).
os.environ.setdefault("STV_CONV_TUNE", "0")

# Parity table: tests append (case, quantity, deviation, tolerance, note); printed once at the end
# of the run so the GPU test log shows what was measured, not only that it passed.
PARITY_ROWS: list[tuple[str, str, float, float, str]] = []


def record_parity(case: str, quantity: str, deviation: float, tolerance: float, note: str = "") -> None:
    PARITY_ROWS.append((case, quantity, float(deviation), float(tolerance), note))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not PARITY_ROWS:
        return
    tr = terminalreporter
    tr.section("parity table (HIP path vs oracle / golden fixtures)")
    tr.write_line(f"{'case':44s} {'quantity':30s} {'deviation':>11s} {'tolerance':>11s}  note")
    for case, q, dev, tol, note in PARITY_ROWS:
        # a row whose tolerance is NaN is "reported, not compared" (its note says why); a NaN deviation against
        # a real tolerance is a failure
        flag = "" if (tol != tol or dev <= tol) else "  <-- OVER"
        tr.write_line(f"{case:44s} {q:30s} {dev:11.3e} {tol:11.3e}  {note}{flag}")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: longer-running CPU test")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class GoldenCase:
    """One ``tests/golden/<name>.npz`` produced by ``oracle/make_golden.py``."""

    def __init__(self, name: str) -> None:
        data = np.load(os.path.join(GOLDEN_DIR, f"{name}.npz"))
        self.arrays = {k: data[k] for k in data.files}
        self.meta = json.loads(bytes(self.arrays.pop("meta_json")).decode())
        self.cfg = tuple(v if v == "M" else int(v) for v in self.meta["cfg"])

    def weights(self):
        from style_transfer_visualizer_amd import synthetic
        m = self.meta
        ws = synthetic.synthetic_conv_weights(m["wseed"], self.cfg)
        out = []
        for li, (w, b) in enumerate(ws):
            if li == 0:
                w = w * m["gain_first"]
            if m["bias_scale"]:
                b = synthetic.synthetic_bias(m["wseed"], li, w.shape[0], m["bias_scale"])
            out.append((w, b))
        return out

    def images(self):
        from style_transfer_visualizer_amd import synthetic
        m = self.meta
        if m.get("png_inputs"):
            # what the image loader returns for the 8-bit PNG of the synthetic image (oracle/make_golden.py::png_roundtrip;
            # reference image_io.py:72-84): uint8 quantisation, /255, Normalize
            out = []
            for seed, hw in ((0, m["hw_content"]), (1, m["hw_style"])):
                t = synthetic.synthetic_image(seed, *hw, normalize=False)[0].mul(255).byte().to(torch.float32).div(255)
                if m["normalize"]:
                    mean = torch.tensor([0.485, 0.456, 0.406], dtype=torch.float32).view(3, 1, 1)
                    std = torch.tensor([0.229, 0.224, 0.225], dtype=torch.float32).view(3, 1, 1)
                    t = (t - mean) / std
                out.append(t.unsqueeze(0))
            return out[0], out[1]
        content = synthetic.synthetic_image(0, *m["hw_content"], normalize=m["normalize"])
        style = synthetic.synthetic_image(1, *m["hw_style"], normalize=m["normalize"])
        return content, style

    def start_image(self) -> torch.Tensor:
        """x0 of the run (the large fixtures do not store a start image that IS the content image)."""
        if "x0" in self.arrays:
            return self.tensor("x0")
        assert bool(self.arrays["x0_is_content"])
        return self.images()[0].clone()

    def spread_level(self, grad_dev_rms: float = 0.0) -> int:
        """Index into the fixture's perturbation levels (``sens_eps``: 3e-7 = two fp32 ulps, 3e-6, 3e-5): the smallest
        level that is at least ``grad_dev_rms`` - the measured relative rms distance of a path's step-1 gradient from the
        reference's.  An implementation that sums a convolution's 9*Cin products in another order is not the reference's
        kernels re-run with another thread count: its trajectory may spread like the reference's own under gradient
        noise of ITS size, not of 2 ulps."""
        eps = np.asarray(self.arrays["sens_eps"], dtype=np.float64)
        idx = int(np.searchsorted(eps, grad_dev_rms, side="left"))
        return min(idx, len(eps) - 1)

    def step_tolerances(self, level: int = 0) -> tuple[np.ndarray, np.ndarray]:
        """Large fixtures: per-step tolerance of the image (relative to its max) and of the three losses
        (relative): north_star's 1e-4 OUTRIGHT wherever the reference's own trajectory reproduces to a quarter of
        that under gradient noise of level ``level`` (measured per step by oracle/make_golden.py::trajectory_spread),
        otherwise 4x the measured spread."""
        xs = np.asarray(self.arrays["x_steps_sensitivity"], dtype=np.float64)[level]
        ls = np.asarray(self.arrays["loss_sensitivity"], dtype=np.float64)[level]
        return np.maximum(1e-4, 4.0 * xs), np.maximum(1e-4, 4.0 * ls)

    def pixel_tolerance(self) -> float:
        """Per-pixel tolerance for x_final, relative to max|x_final|.

        north_star asks for 1e-4; a fixture whose trajectory amplifies 2-ulp gradient
        differences beyond that (``x_final_sensitivity``, see oracle/make_golden.py)
        gets 4x its measured sensitivity instead.
        """
        return max(1e-4, 4.0 * float(self.arrays["x_final_sensitivity"]))

    def tensor(self, key: str) -> torch.Tensor:
        return torch.from_numpy(np.asarray(self.arrays[key]))


GOLDEN_CASES = [
    "mini_white_lbfgs", "mini_content_lbfgs", "mini_random_lbfgs_nonorm",
    "mini_white_adam", "mini_clamp_lbfgs", "mini_clamp_adam", "tiny_taps_lbfgs", "vgg19_white_lbfgs",
    "vgg19_random_adam", "vgg19_content_lbfgs",
]

# the two LARGE reference runs (configs[0] literally; 12 full-width L-BFGS steps at 128^2): images stored subsampled
LARGE_CASES = ["cfg0_256_content_lbfgs50", "vgg19_128_random_lbfgs12"]


@pytest.fixture(params=GOLDEN_CASES)
def golden_case(request) -> GoldenCase:
    return GoldenCase(request.param)


@pytest.fixture(autouse=True)
def _package_logger_propagates(monkeypatch):
    """The package logger does not propagate (as the reference's); ``caplog`` listens on the root logger."""
    from style_transfer_visualizer_amd.logging_utils import logger
    monkeypatch.setattr(logger, "propagate", True)

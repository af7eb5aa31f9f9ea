"""Cache what the ORACLE ALONE says about the full-size start images (test infrastructure, see ``oracle/__init__.py``).

    python -m oracle.make_fullsize_ref            # build container or any host: only the oracle runs, ~3 minutes

``tests/test_gpu_fullsize.py`` measures the HIP path's fp32 gradient against a float64 evaluation of the same
algorithm.  At the start image that evaluation is a property of the oracle and of the seeded inputs alone - it does not
depend on anything the HIP path does - and costs 25 s (1024^2) of the GPU suite's host time.  This script computes it
once and stores, per size, as ``tests/golden/fullsize_fp64_<size>.npz``:

* ``g64_sub``   the float64 gradient at x0, every ``sub``-th row and column, rounded to float32 (6e-8 relative:
                four orders of magnitude below the differences it is compared with);
* ``err_cpu_full`` / ``err_cpu_sub``  relative rms distance of the oracle's OWN fp32 gradient from the float64 one
                (over the whole image / over the subsample): the yardstick "the HIP path may be as far from float64
                as the reference arithmetic is";
* losses of both evaluations and float64 checksums of x0 (so a test can tell it evaluates the same image).

Nothing here comes from /root/reference: the oracle is pinned to it by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import core_model_ref as ocm  # noqa: E402
from style_transfer_visualizer_amd import synthetic  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
S_LAYERS, C_LAYERS = [0, 5, 10, 19, 28], [21]
STYLE_W, CONTENT_W = 1e5, 1.0
SUB = {512: 2, 1024: 4}


def make(size: int) -> None:
    t0 = time.time()
    weights = synthetic.synthetic_conv_weights(0)
    content = synthetic.synthetic_image(0, size, size)
    style = synthetic.synthetic_image(1, size, size)
    x0 = torch.randn(content.shape, generator=torch.Generator().manual_seed(0))       # init_method=random, as bench.py
    oracle = ocm.OracleModel(ocm.vgg_program(weights, synthetic.VGG19_CFG), S_LAYERS, C_LAYERS)
    oracle.set_targets(style, content)
    s32, c32, t32, g32 = ocm.loss_and_grad(oracle, x0, STYLE_W, CONTENT_W)
    w64 = [(w.double(), b.double()) for w, b in weights]
    oracle64 = ocm.OracleModel(ocm.vgg_program(w64, synthetic.VGG19_CFG), S_LAYERS, C_LAYERS)
    oracle64.set_targets(style.double(), content.double())
    s64, c64, t64, g64 = ocm.loss_and_grad(oracle64, x0.double(), STYLE_W, CONTENT_W)
    k = SUB[size]
    g64s, g32s = g64[..., ::k, ::k], g32.double()[..., ::k, ::k]
    out = {
        "size": np.asarray(size), "sub": np.asarray(k),
        "g64_sub": g64s.float().numpy(),
        "g64_norm_full": np.asarray(float(g64.norm())), "g64_absmax": np.asarray(float(g64.abs().max())),
        "err_cpu_full": np.asarray(float((g32.double() - g64).norm() / g64.norm())),
        "err_cpu_sub": np.asarray(float((g32s - g64s).norm() / g64s.norm())),
        "losses_fp32": np.asarray([float(s32), float(c32), float(t32)]),
        "losses_fp64": np.asarray([float(s64), float(c64), float(t64)]),
        "x0_sum": np.asarray(float(x0.double().sum())), "x0_abs_sum": np.asarray(float(x0.double().abs().sum())),
        "torch_version": np.frombuffer(torch.__version__.encode(), dtype=np.uint8),
    }
    path = os.path.join(GOLDEN_DIR, f"fullsize_fp64_{size}.npz")
    np.savez_compressed(path, **out)
    print(f"{size}: fp32 oracle vs float64: rel rms {float(out['err_cpu_full']):.3e} (subsample {float(out['err_cpu_sub']):.3e}); "
          f"total {float(t32)!r} / {float(t64)!r}; {os.path.getsize(path) / 1024:.0f} KiB; {time.time() - t0:.0f} s")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for size in ([int(a) for a in sys.argv[1:]] or [512, 1024]):
        make(size)

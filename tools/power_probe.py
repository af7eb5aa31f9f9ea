#!/usr/bin/env python3
"""Samples GPU power / clocks / temperature (rocm-smi) while the optimisation loop runs at steady state.

    python tools/power_probe.py [--size 1024] [--seconds 6] [--json gpurun_out/power.json]

Evidence for DESIGN.md: whether the step runs power-limited (sclk below the 2.4 GHz peak at the board's
power cap).  Read-only: no setting is changed.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sample() -> dict:
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showuse", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        keep = {}
        for k, v in card.items():
            lk = k.lower()
            if any(w in lk for w in ("power", "sclk", "mclk", "fclk", "temperature (sensor junction)", "temperature (sensor edge)", "gpu use")):
                keep[k] = v
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    import torch

    from style_transfer_visualizer_amd import core_model, optimizers, synthetic
    os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
    dev = torch.device("cuda")
    idle = sample()
    content = synthetic.synthetic_image(0, a.size, a.size).to(dev)
    style = synthetic.synthetic_image(1, a.size, a.size).to(dev)
    model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21], precision="bf16").to(dev)
    model.set_targets(style, content)
    x = torch.randn(1, 3, a.size, a.size, device=dev).requires_grad_(True)
    opt = optimizers.HipLBFGS([x], lr=1.0)
    side = torch.cuda.Stream()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            s = sample()
            s["t"] = time.time()
            samples.append(s)
            time.sleep(0.2)
    with torch.cuda.stream(side):
        for _ in range(110):                      # fill the history
            opt.step(lambda: model.loss_and_grad(x, 1e5, 1.0, live_scores=True)[2])
        torch.cuda.synchronize()
        th = threading.Thread(target=sampler)
        th.start()
        t0, n = time.time(), 0
        while time.time() - t0 < a.seconds:
            for _ in range(50):
                opt.step(lambda: model.loss_and_grad(x, 1e5, 1.0, live_scores=True)[2])
            torch.cuda.synchronize()
            n += 50
        dt = time.time() - t0
        stop.set()
        th.join()
    res = {"size": a.size, "steps_per_s": n / dt, "idle": idle, "samples": samples[1:]}
    print(json.dumps(res, indent=1)[:4000])
    if a.json:
        json.dump(res, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

"""Discrete decisions of the network (ReLU on/off, max-pool arg-max) as data: extract them from the HIP
engine or from a CPU oracle forward pass, impose them on an oracle, count where two evaluations differ.

Why: between two CORRECT fp32 evaluations of VGG19 a pre-activation whose float64 value is within ~1e-6 of
zero (or a pooling window whose two largest entries are that close) may fall on either side - with 1e6..3e7
activations per evaluation such near-ties exist in nearly every image (tests/diag).  The decision
is discrete, so the two gradients then differ by O(1e-2) of scale at the pixels of that unit's receptive
field although every sum in both paths is accurate to 1e-7.  Comparing on the SAME piecewise-linear branch
(``lock``) separates rounding accuracy - which has a bound - from which side of a tie a path happened to
take - which has none.  Test infrastructure only.
"""
from __future__ import annotations

import copy

import torch
import torch.nn.functional as F

from oracle import core_model_ref as ocm
from style_transfer_visualizer_amd import ops

Decisions = dict  # layer index -> ("relu_mask", bool tensor [1,C,H,W]) | ("pool_idx", int64 tensor [1,C,H/2,W/2])


def hip_decisions(model) -> Decisions:
    """The decisions the HIP path took in its LAST evaluation (read from the engine's stored activations)."""
    eng = next(iter(model._engines.values()))
    n_layers = max(n.layer for n in eng.sched.nodes) + 2
    relu_at = {nd.layer for nd in eng.sched.nodes if nd.kind == "relu"}
    dec: Decisions = {}
    for nd in eng.sched.nodes:
        if nd.kind in ("conv", "conv_first"):
            # the ReLU behind this conv, fused into its epilogue or applied by the consumer: act > 0 <=> z > 0
            followed = nd.dst.relu_fused or any(n.src is nd.dst and (n.relu_in or n.kind == "relu") for n in eng.sched.nodes)
            if followed and nd.layer + 1 < n_layers and nd.layer + 1 not in relu_at:
                dec[nd.layer + 1] = ("relu_mask", ops.from_nhwc(eng.sched.interior(nd.dst.act)).cpu() > 0)
        elif nd.kind == "relu":
            dec[nd.layer] = ("relu_mask", ops.from_nhwc(eng.sched.interior(nd.dst.act)).cpu() > 0)
        elif nd.kind == "pool":
            _, idx = F.max_pool2d(ops.from_nhwc(eng.sched.interior(nd.src.act)).cpu().double(), 2, 2, return_indices=True)
            dec[nd.layer] = ("pool_idx", idx)
    return dec


def oracle_decisions(program, x: torch.Tensor, n_layers: int) -> Decisions:
    """The decisions a CPU forward pass of ``program`` takes at ``x`` (its dtype decides the arithmetic)."""
    dec: Decisions = {}
    h = x
    with torch.no_grad():
        for li in range(n_layers):
            layer = program[li]
            if layer[0] == "relu":
                dec[li] = ("relu_mask", h > 0)
            elif layer[0] == "pool":
                _, idx = F.max_pool2d(h, 2, 2, return_indices=True)
                dec[li] = ("pool_idx", idx)
            h = ocm.run_layer(layer, h)
    return dec


def lock(oracle: ocm.OracleModel, dec: Decisions) -> ocm.OracleModel:
    """Copy of ``oracle`` whose ReLU masks and pooling arg-maxes are ``dec`` instead of its own."""
    prog = list(oracle.program)
    for li, d in dec.items():
        if li < len(prog) and prog[li][0] in ("relu", "pool", "relu_mask", "pool_idx"):
            prog[li] = d
    locked = copy.copy(oracle)
    locked.program = prog
    return locked


def count_flips(a: Decisions, b: Decisions) -> int:
    """Number of individual decisions on which two evaluations differ.  A pooling arg-max that differs only
    inside a window of equal values below a closed ReLU cannot be told from the masks here and is counted."""
    n = 0
    for li in a.keys() & b.keys():
        n += int((a[li][1] != b[li][1]).sum())
    return n


def flip_gaps(a: Decisions, b: Decisions, program64, x64: torch.Tensor, n_layers: int) -> float:
    """Largest float64 gap - relative to the rms of its layer - among the decisions on which ``a`` and ``b``
    differ: |z| for a ReLU, (value picked by a) - (value picked by b) for a pooling window.  Genuine near-ties
    have gaps at fp32 rounding level (<= ~1e-5); a wrong kernel flips decisions with large gaps."""
    worst = 0.0
    h = x64
    with torch.no_grad():
        for li in range(n_layers):
            layer = program64[li]
            if li in a and li in b:
                rms = float(h.pow(2).mean().sqrt()) + 1e-300
                diff = a[li][1] != b[li][1]
                if diff.any():
                    if a[li][0] == "relu_mask":
                        worst = max(worst, float(h[diff].abs().max()) / rms)
                    else:
                        va = h.flatten(2).gather(2, a[li][1].flatten(2)).reshape(diff.shape)
                        vb = h.flatten(2).gather(2, b[li][1].flatten(2)).reshape(diff.shape)
                        worst = max(worst, float((va - vb)[diff].abs().max()) / rms)
            h = ocm.run_layer(layer, h)
    return worst


def n_program_layers(style_layers, content_layers) -> int:
    return max(list(style_layers) + list(content_layers)) + 1

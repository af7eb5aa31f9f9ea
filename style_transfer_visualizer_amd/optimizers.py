"""Device-resident optimizers with the ``torch.optim`` ``step(closure)`` protocol.

``HipLBFGS`` replaces ``torch.optim.LBFGS`` as constructed by the reference
(core_model.py:344-349, optimization.py:212-217) for its default
``max_iter = max_eval = 1`` configuration: same state machine, but every scalar
and branch stays on the GPU (``stv_lbfgs_step``), so ``step`` never syncs.
``HipAdam`` is the injected-Adam counterpart (reference tests/test_optimization.py:178).
"""
from __future__ import annotations

import os
import threading
from collections.abc import Callable
from dataclasses import dataclass

import torch

from . import ops


@dataclass
class StepRequest:
    """What ``HipLBFGS.step`` offers the closure: "if your evaluation is one command-buffer launch that ends
    with the gradient of exactly this tensor, append my update to it" (``stv_op_t`` LBFGS_STEP, include/stv.h).
    The model's fused path (``StyleContentModel.loss_and_grad``) takes it with :func:`claim_step`; closure and
    update then replay as ONE hipGraph instead of a graph followed by four eager launches."""

    x: torch.Tensor
    state: torch.Tensor
    work: torch.Tensor
    history: int
    lr: float
    tol_grad: float
    tol_change: float
    taken: bool = False

    def key(self) -> tuple:
        return (self.state.data_ptr(), self.work.data_ptr(), self.history, self.lr, self.tol_grad, self.tol_change)


_tls = threading.local()      # (style_transfer_batch runs images on several host threads)


def single_evaluation(closure):
    """Mark a closure that evaluates the model exactly ONCE per call (and does nothing with the image afterwards):
    only then may the update ride at the end of that evaluation's launch - a closure that evaluated twice would
    compute its second gradient at the already updated image.  ``OptimizationRunner``'s closure is marked."""
    closure._stv_single_eval = True
    return closure


def claim_step(x: torch.Tensor) -> StepRequest | None:
    """The pending request of the optimizer step this thread is inside of, if it is for ``x`` (same storage)
    and nobody has taken it yet.  The taker MUST enqueue the update behind the gradient it computes."""
    req = getattr(_tls, "pending", None)
    if req is None or req.taken or req.x.data_ptr() != x.data_ptr() or req.x.numel() != x.numel():
        return None
    req.taken = True
    return req


def _single_param(params) -> torch.Tensor:
    plist = list(params)
    if len(plist) != 1 or not isinstance(plist[0], torch.Tensor):
        msg = "HIP optimizers take exactly one tensor (the image being optimised)"
        raise ValueError(msg)
    p = plist[0]
    if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
        msg = "HIP optimizers need a contiguous float32 GPU tensor (no CPU fallback on this path)"
        raise RuntimeError(msg)
    return p


class HipLBFGS(torch.optim.Optimizer):
    """L-BFGS without line search, one iteration per ``step`` (torch semantics)."""

    def __init__(self, params, lr: float = 1.0, max_iter: int = 1, max_eval: int | None = None,
                 tolerance_grad: float = 1e-7, tolerance_change: float = 1e-9,
                 history_size: int = 100, *, shard_group=None) -> None:
        """``shard_group`` (a ``torch.distributed`` group, or ``True`` for the default one): the tensor
        is ONE SHARD of the optimised image (row strips of one large image).  L-BFGS is the same
        update on every shard once its inner products are summed over the shards, so each step
        all-reduces the ~650 doubles of the dot-product table (SUM; max|g| with MAX) between the two
        history sweeps - no vector ever crosses a link."""
        if lr < 0.0:
            msg = f"Invalid learning rate: {lr}"
            raise ValueError(msg)
        if max_iter != 1:
            msg = "HipLBFGS implements max_iter=1 (the reference default); use make_lbfgs for other settings"
            raise ValueError(msg)
        if not 1 <= history_size <= 128:
            msg = "history_size must be in [1, 128]"
            raise ValueError(msg)
        if max_eval is None:
            max_eval = max_iter * 5 // 4
        p = _single_param(params)
        defaults = dict(lr=lr, max_iter=max_iter, max_eval=max_eval, tolerance_grad=tolerance_grad,
                        tolerance_change=tolerance_change, history_size=history_size, line_search_fn=None)
        super().__init__([p], defaults)
        self._p = p
        # "compact": history read twice per step (inner-product tables); "twoloop": torch's
        # operation order, 2m dependent passes.  Same state machine, rounding-level differences.
        self._compact = os.environ.get("STV_LBFGS", "compact") != "twoloop"
        self._shard_group = shard_group
        if shard_group is not None and not self._compact:
            msg = "a sharded image needs the inner-product form of L-BFGS (STV_LBFGS=compact)"
            raise ValueError(msg)
        self._dev_state, self._work = ops.lbfgs_alloc(p.numel(), history_size, p.device, compact=self._compact)
        self._steps = 0
        # STV_FUSE_STEP=0: the update always as its own launches behind the closure
        self._fuse = self._compact and shard_group is None and os.environ.get("STV_FUSE_STEP", "1") != "0"

    @torch.no_grad()
    def step(self, closure: Callable[[], torch.Tensor]) -> torch.Tensor:  # type: ignore[override]
        g = self.param_groups[0]
        req = None
        if self._fuse and getattr(closure, "_stv_single_eval", False):
            req = StepRequest(self._p, self._dev_state, self._work, int(g["history_size"]), float(g["lr"]),
                              float(g["tolerance_grad"]), float(g["tolerance_change"]))
            _tls.pending = req
        try:
            with torch.enable_grad():
                loss = closure()
        finally:
            _tls.pending = None
        if req is not None and req.taken:      # the closure's launch ended with this step's update
            self._steps += 1
            return loss
        grad = self._p.grad
        if grad is None:
            grad = torch.zeros_like(self._p)
        if not grad.is_contiguous():
            grad = grad.contiguous()
        m_max = min(self._steps, g["history_size"])
        if self._shard_group is None:
            ops.lbfgs_step(self._p, grad, self._dev_state, self._work, g["history_size"], m_max, float(g["lr"]),
                           g["tolerance_grad"], g["tolerance_change"], compact=self._compact)
        else:
            import torch.distributed as dist  # noqa: PLC0415
            group = None if self._shard_group is True else self._shard_group
            dots = ops.lbfgs_dots(grad, self._dev_state, self._work, g["history_size"], m_max)
            imax = ops.lbfgs_dots_view(self._work, self._p.numel(), g["history_size"])[1]
            gmax = dots[imax:imax + 1].clone()
            dist.all_reduce(gmax, op=dist.ReduceOp.MAX, group=group)
            dist.all_reduce(dots, op=dist.ReduceOp.SUM, group=group)
            dots[imax:imax + 1].copy_(gmax)
            ops.lbfgs_apply(self._p, grad, self._dev_state, self._work, g["history_size"], float(g["lr"]),
                            g["tolerance_grad"], g["tolerance_change"])
        self._steps += 1
        return loss

    def device_state(self) -> dict:
        """Debug/test view of the device state block (this call synchronises)."""
        raw = self._dev_state.cpu()
        ints, flts = raw.view(torch.int32), raw.view(torch.float32)
        f0 = 8 if self._compact else 6      # first float field of the state struct
        return {"n_iter": int(ints[0]), "hist_len": int(ints[1]), "head": int(ints[2]), "skip": int(ints[3]),
                "no_update": int(ints[4]), "t": float(flts[f0]), "H_diag": float(flts[f0 + 1]),
                "gtd": float(flts[f0 + 2]), "gmax": float(flts[f0 + 3])}


class HipAdam(torch.optim.Optimizer):
    """Adam (no amsgrad / weight decay) as one fused pass per step."""

    def __init__(self, params, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8) -> None:
        p = _single_param(params)
        super().__init__([p], dict(lr=lr, betas=betas, eps=eps))
        self._p = p
        self._m = torch.zeros_like(p, requires_grad=False)
        self._v = torch.zeros_like(p, requires_grad=False)
        self._t = 0

    @torch.no_grad()
    def step(self, closure: Callable[[], torch.Tensor] | None = None):  # type: ignore[override]
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._p.grad is None:
            return loss
        g = self.param_groups[0]
        self._t += 1
        ops.adam_step(self._p, self._p.grad.contiguous(), self._m, self._v, self._t, lr=g["lr"],
                      betas=g["betas"], eps=g["eps"])
        return loss


def make_lbfgs(input_img: torch.Tensor, lr: float, max_iter: int, max_eval: int) -> torch.optim.Optimizer:
    """L-BFGS for the image: device-resident when it can be, ``torch.optim.LBFGS`` otherwise.

    The device-resident form covers the reference default (one iteration per
    step on a float32 GPU image).  Other settings run torch's optimizer on the
    same GPU tensors; the closure (all the FLOPs) is the HIP path either way.
    """
    if max_iter == 1 and input_img.is_cuda and input_img.dtype == torch.float32:
        return HipLBFGS([input_img], lr=lr, max_iter=max_iter, max_eval=max_eval)
    return torch.optim.LBFGS([input_img], lr=lr, max_iter=max_iter, max_eval=max_eval)

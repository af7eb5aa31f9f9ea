"""Oracle restatement of the optimizer step and the step loop (torch CPU fp32).

Test infrastructure (see ``oracle/__init__.py``).

The reference drives ``torch.optim.LBFGS`` (constructed at
/root/reference/src/style_transfer_visualizer/core_model.py:344-349 and
optimization.py:212-217; the implementation is third-party:
torch 2.10 ``torch/optim/lbfgs.py`` ``LBFGS.step``, no line search) and, in its
tests, an injected ``torch.optim.Adam`` (tests/test_optimization.py:178).
``LbfgsRef`` restates the no-line-search branch of that published algorithm
with the same fp32 operation order, so on CPU it is bit-identical to
``torch.optim.LBFGS`` (checked in tests/test_oracle_golden.py); it exists so
the device-resident L-BFGS can be compared state-by-state.
"""
from __future__ import annotations

from collections.abc import Callable

import torch


class LbfgsRef:
    """``torch.optim.LBFGS.step`` restated for one flat fp32 parameter."""

    def __init__(
        self,
        x: torch.Tensor,
        lr: float = 1.0,
        max_iter: int = 1,
        max_eval: int | None = None,
        tolerance_grad: float = 1e-7,
        tolerance_change: float = 1e-9,
        history_size: int = 100,
    ) -> None:
        self.x = x
        self.lr = lr
        self.max_iter = max_iter
        self.max_eval = max_eval if max_eval is not None else max_iter * 5 // 4
        self.tolerance_grad = tolerance_grad
        self.tolerance_change = tolerance_change
        self.history_size = history_size
        self.n_iter = 0
        self.func_evals = 0
        self.d: torch.Tensor | None = None
        self.t: float | torch.Tensor | None = None
        self.old_dirs: list[torch.Tensor] = []
        self.old_stps: list[torch.Tensor] = []
        self.ro: list[torch.Tensor] = []
        self.H_diag: float | torch.Tensor = 1
        self.prev_flat_grad: torch.Tensor | None = None
        self.prev_loss: float | None = None
        self.al: list = [None] * history_size

    @torch.no_grad()
    def step(self, closure: Callable[[], tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
        """``closure() -> (loss, grad)`` evaluated at the current ``self.x``."""
        orig_loss, grad = closure()
        loss = float(orig_loss)
        current_evals = 1
        self.func_evals += 1
        flat_grad = grad.reshape(-1)
        if flat_grad.abs().max() <= self.tolerance_grad:
            return orig_loss

        d, t = self.d, self.t
        n_iter = 0
        while n_iter < self.max_iter:
            n_iter += 1
            self.n_iter += 1
            if self.n_iter == 1:
                d = flat_grad.neg()
                self.old_dirs, self.old_stps, self.ro = [], [], []
                self.H_diag = 1
            else:
                y = flat_grad.sub(self.prev_flat_grad)
                s = d.mul(t)
                ys = y.dot(s)
                if ys > 1e-10:
                    if len(self.old_dirs) == self.history_size:
                        self.old_dirs.pop(0)
                        self.old_stps.pop(0)
                        self.ro.pop(0)
                    self.old_dirs.append(y)
                    self.old_stps.append(s)
                    self.ro.append(1.0 / ys)
                    self.H_diag = ys / y.dot(y)
                num_old = len(self.old_dirs)
                al = self.al
                q = flat_grad.neg()
                for i in range(num_old - 1, -1, -1):
                    al[i] = self.old_stps[i].dot(q) * self.ro[i]
                    q.add_(self.old_dirs[i], alpha=-al[i])
                d = r = torch.mul(q, self.H_diag)
                for i in range(num_old):
                    be_i = self.old_dirs[i].dot(r) * self.ro[i]
                    r.add_(self.old_stps[i], alpha=al[i] - be_i)
            if self.prev_flat_grad is None:
                self.prev_flat_grad = flat_grad.clone()
            else:
                self.prev_flat_grad.copy_(flat_grad)
            self.prev_loss = loss
            if self.n_iter == 1:
                t = min(1.0, 1.0 / flat_grad.abs().sum()) * self.lr
            else:
                t = self.lr
            gtd = flat_grad.dot(d)
            if gtd > -self.tolerance_change:
                break
            self.x.view(-1).add_(d, alpha=t)
            ls_func_evals = 0
            if n_iter != self.max_iter:
                new_loss, grad = closure()
                loss = float(new_loss)
                flat_grad = grad.reshape(-1)
                ls_func_evals = 1
            current_evals += ls_func_evals
            self.func_evals += ls_func_evals
            if n_iter == self.max_iter:
                break
            if current_evals >= self.max_eval:
                break
            if flat_grad.abs().max() <= self.tolerance_grad:
                break
            if d.mul(t).abs().max() <= self.tolerance_change:
                break
            if abs(loss - self.prev_loss) < self.tolerance_change:
                break
        self.d, self.t = d, t
        return orig_loss


class AdamRef:
    """torch 2.10 ``_single_tensor_adam`` (no amsgrad / weight decay) restated."""

    def __init__(self, x: torch.Tensor, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        self.x = x
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = torch.zeros_like(x)
        self.v = torch.zeros_like(x)
        self.t = 0

    @torch.no_grad()
    def step(self, closure: Callable[[], tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
        loss, grad = closure()
        b1, b2 = self.betas
        self.t += 1
        self.m.lerp_(grad, 1 - b1)
        self.v.mul_(b2).addcmul_(grad, grad, value=1 - b2)
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        step_size = self.lr / bc1
        denom = (self.v.sqrt() / (bc2 ** 0.5)).add_(self.eps)
        self.x.addcdiv_(self.m, denom, value=-step_size)
        return loss


def run_loop(
    loss_and_grad: Callable[[torch.Tensor], tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]],
    x0: torch.Tensor,
    steps: int,
    *,
    optimizer: str = "lbfgs",
    lr: float | None = None,
    keep_steps: bool = False,
    on_step: Callable | None = None,
) -> dict:
    """Step loop with the reference's accounting (optimization.py:162-202).

    One ``optimizer.step(closure)`` per step, 1-based step ids, the loss triple
    recorded is the one from the (last) closure call of that step.
    """
    x = x0.detach().clone()
    if optimizer == "lbfgs":
        opt = LbfgsRef(x, lr=1.0 if lr is None else lr)
    elif optimizer == "adam":
        opt = AdamRef(x, lr=1e-3 if lr is None else lr)
    else:
        raise ValueError(optimizer)
    hist = {"style": [], "content": [], "total": []}
    first_grad = None
    x_steps: list[torch.Tensor] = []          # image after each step (keep_steps)

    for _ in range(steps):
        rec = {}

        def closure():
            s, c, tot, g = loss_and_grad(x)
            rec["s"], rec["c"], rec["t"] = float(s), float(c), float(tot)
            rec["g"] = g
            return tot, g

        opt.step(closure)
        if first_grad is None:
            first_grad = rec["g"].clone()
        hist["style"].append(rec["s"])
        hist["content"].append(rec["c"])
        hist["total"].append(rec["t"])
        if keep_steps:
            x_steps.append(x.detach().clone())
        if on_step is not None:
            on_step(opt)
    return {"x": x, "history": hist, "first_grad": first_grad, "optimizer": opt, "x_steps": x_steps}

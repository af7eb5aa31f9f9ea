"""Device L-BFGS over a FULL history: the regime every step of BASELINE configs[1]/[2] past step 100 runs in.

The reference's optimizer is ``torch.optim.LBFGS`` as constructed at reference core_model.py:344-349 and
driven at optimization.py:175 (history_size 100, max_iter 1, no line search): it keeps at most 100 (y, s)
pairs and pops the oldest.  Here both device forms - ``stv_lbfgsc_step`` (inner-product space,
csrc/lbfgs_compact.hip: the ``hi_t`` halves of the three coefficient walks serve history indices >= 64, the
S.Y / Y.Y tables are indexed through a ring of history+1 slots) and ``stv_lbfgs_step`` (operation-ordered,
csrc/optim.hip) - run for history + 40 steps (>= 130 at the default history) beside two twins of
``oracle.optim_ref.LbfgsRef`` (bit-identical to ``torch.optim.LBFGS`` on CPU, tests/test_oracle_golden.py):
one in fp32 (the reference's arithmetic) and one in float64 (the yardstick).

All three are fed THE SAME gradient sequence - the gradient of an ill-conditioned quartic at the DEVICE's
iterate - so they build their histories from identical ``y`` vectors and nothing chaotic separates them:
what is compared at every step is the update ``x_after - x_before`` each of them applies.  Criterion
(VERDICT r2 item 1): the device's deviation from the float64 update is at most 4x the fp32 reference's own
(measured over the same and the few preceding steps: a single step's fp32 error is one draw of a noisy
quantity), with a floor of 2e-6 of the update.  Integer state (n_iter, history length, skip / no-update
flags) must equal the fp32 twin's at every step.

Scripted events, all with more than 64 pairs stored (history 64: with the ring full):
* a repeated gradient (``y = 0`` -> ``ys = 0 <= 1e-10``): no pair is pushed, the direction is recomputed;
* ``max|g| <= 1e-7``: early return, nothing changes, the step does not count as an iteration;
* a barely-above-tolerance gradient: ``g.d > -1e-9`` -> state is saved but the image does not move.
"""
from __future__ import annotations

import math

import pytest
import torch

from oracle import optim_ref
from style_transfer_visualizer_amd import ops
from tests.conftest import record_parity

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _objective(n: int, cond: float = 1e4, quart: float = 0.05, seed: int = 5):
    """Gradient of  sum 0.5 a x^2 - b x + quart x^4 + 0.1 x_i x_{i+1}  with a spectrum 1..cond: L-BFGS
    (no line search) is still descending after 170 steps, every pair has y.s > 0 by a wide margin."""
    gen = torch.Generator().manual_seed(seed)
    a = torch.exp(torch.rand(n, generator=gen, dtype=torch.float64) * math.log(cond))
    b = torch.randn(n, generator=gen, dtype=torch.float64)

    def grad(x: torch.Tensor) -> torch.Tensor:
        x = x.double()
        g = a * x - b + 4.0 * quart * x ** 3
        g[1:] += 0.1 * x[:-1]
        g[:-1] += 0.1 * x[1:]
        return g.float()
    return grad


class _Device:
    """One device L-BFGS instance over ``n`` elements, optionally as row shards whose inner products are
    summed on the host the way ``HipLBFGS(shard_group=)`` all-reduces them (SUM; max|g| with MAX)."""

    def __init__(self, n: int, history: int, compact: bool, shards: tuple[int, ...] | None = None) -> None:
        self.history, self.compact = history, compact
        self.bounds = [0, n] if shards is None else [0, *torch.tensor(shards).cumsum(0).tolist()]
        assert self.bounds[-1] == n
        self.x = [torch.zeros(e - b, device=DEV) for b, e in zip(self.bounds[:-1], self.bounds[1:], strict=True)]
        self.st = [ops.lbfgs_alloc(t.numel(), history, DEV, compact=compact) for t in self.x]
        self.calls = 0

    def step(self, g: torch.Tensor) -> None:
        m_max = min(self.calls, self.history)
        gs = [g[b:e].to(DEV) for b, e in zip(self.bounds[:-1], self.bounds[1:], strict=True)]
        if len(self.x) == 1:
            state, work = self.st[0]
            ops.lbfgs_step(self.x[0], gs[0], state, work, self.history, m_max, 1.0, compact=self.compact)
        else:
            views = [ops.lbfgs_dots(gi, state, work, self.history, m_max) for gi, (state, work) in zip(gs, self.st, strict=True)]
            imax = ops.lbfgs_dots_view(self.st[0][1], self.x[0].numel(), self.history)[1]
            total = torch.stack(views).sum(0)
            total[imax] = torch.stack([v[imax] for v in views]).max()
            for v in views:
                v.copy_(total)
            for xi, gi, (state, work) in zip(self.x, gs, self.st, strict=True):
                ops.lbfgs_apply(xi, gi, state, work, self.history, 1.0)
        self.calls += 1

    def image(self) -> torch.Tensor:
        return torch.cat([t.cpu() for t in self.x])

    def ints(self, shard: int = 0) -> dict:
        raw = self.st[shard][0].cpu().view(torch.int32)
        return {"n_iter": int(raw[0]), "hist_len": int(raw[1]), "head": int(raw[2]), "skip": int(raw[3]),
                "no_update": int(raw[4])}


def _run(case: str, n: int, history: int, compact: bool, shards: tuple[int, ...] | None = None,
         extra_steps: int = 40) -> None:
    grad = _objective(n)
    dev = _Device(n, history, compact, shards)
    # For large n both twins run torch's own vector ops on the GPU (same algorithm, same arithmetic width; on the host
    # cores they were 50 + 25 of this test's 80 s at n = 3 x 512^2).  At n = 20,000 the fp32 twin - the reference's
    # arithmetic, bit-identical to torch.optim.LBFGS on CPU - stays on the CPU.
    dev64 = DEV if n >= 500_000 else torch.device("cpu")
    x32 = torch.zeros(n, device=dev64)
    x64 = torch.zeros(n, dtype=torch.float64, device=dev64)
    twin32 = optim_ref.LbfgsRef(x32, lr=1.0, history_size=history)
    twin64 = optim_ref.LbfgsRef(x64, lr=1.0, history_size=history)
    steps = history + extra_steps
    ev_repeat, ev_tiny, ev_flat = history + 10, history + 15, history + 20      # scripted events (1-based steps)
    g_prev = None
    e32_hist: list[float] = []
    worst_dev = worst_32 = worst_ratio = 0.0
    pushes_after_full = 0
    zero = torch.tensor(0.0)
    for step in range(1, steps + 1):
        x_before = dev.image()
        if step == ev_repeat:
            g = g_prev.clone()                                  # y = 0 exactly: ys <= 1e-10, no push
        elif step == ev_tiny:
            g = torch.full((n,), 5e-8)                          # max|g| <= tolerance_grad: early return
        elif step == ev_flat:
            g = 2e-7 * torch.sign(grad(x_before))               # passes the gradient test, g.d > -1e-9
        else:
            g = grad(x_before)
        b32, b64 = x32.clone(), x64.clone()
        g64 = g.to(dev64).double()
        len_before = len(twin32.old_dirs)
        newest = twin32.old_dirs[-1] if twin32.old_dirs else None
        g32 = g.to(dev64)
        twin32.step(lambda: (zero, g32.clone()))
        twin64.step(lambda: (zero.double(), g64))
        dev.step(g)
        x_after = dev.image()
        assert torch.isfinite(x_after).all(), f"{case}: non-finite image at step {step}"
        # ---- integer state: every shard equal to the fp32 twin --------------------------------------
        for k in range(len(dev.x)):
            st = dev.ints(k)
            assert st["n_iter"] == twin32.n_iter, f"{case} step {step}: n_iter {st['n_iter']} vs {twin32.n_iter}"
            assert st["hist_len"] == len(twin32.old_dirs) == len(twin64.old_dirs), \
                f"{case} step {step}: history {st['hist_len']} vs {len(twin32.old_dirs)}/{len(twin64.old_dirs)}"
            assert st["skip"] == (1 if step == ev_tiny else 0), f"{case} step {step}: skip flag {st['skip']}"
            assert st["no_update"] == (1 if step == ev_flat else 0), f"{case} step {step}: no_update {st['no_update']}"
        pushed = bool(twin32.old_dirs) and twin32.old_dirs[-1] is not newest
        if pushed and len_before == history:
            pushes_after_full += 1                              # the oldest pair was evicted (ring head moved)
        assert not (pushed and step in (ev_repeat, ev_tiny))
        # ---- the update each optimizer applied ----------------------------------------------------------
        u_dev = (x_after.double() - x_before.double())
        u_32 = (x32 - b32).double().cpu()
        u_64 = (x64 - b64).cpu()
        scale = float(u_64.abs().max())
        if step in (ev_tiny, ev_flat):
            assert scale == 0.0 and float(u_dev.abs().max()) == 0.0 and float(u_32.abs().max()) == 0.0, \
                f"{case} step {step}: the image must not move"
        else:
            assert scale > 0.0
            e_dev = float((u_dev - u_64).abs().max()) / scale
            e_32 = float((u_32 - u_64).abs().max()) / scale
            e32_hist.append(e_32)
            ref = max(e32_hist[-8:])                            # the reference's own error, this and the last steps
            tol = max(4.0 * ref, 2e-6)
            worst_dev, worst_32 = max(worst_dev, e_dev), max(worst_32, e_32)
            worst_ratio = max(worst_ratio, e_dev / max(ref, 5e-7))
            assert e_dev <= tol, (f"{case} step {step} (history {len(twin32.old_dirs)}): device update is {e_dev:.2e} "
                                  f"from the float64 update, the fp32 reference {e_32:.2e} (window max {ref:.2e})")
        g_prev = g
    st = dev.ints()
    assert st["hist_len"] == history and len(twin32.old_dirs) == history
    assert pushes_after_full >= extra_steps - 8
    if compact:   # ring of history+1 slots: the head has advanced once per eviction
        assert st["head"] == pushes_after_full % (history + 1), f"{case}: head {st['head']} after {pushes_after_full} evictions"
    x_end = dev.image().double()
    x64 = x64.cpu()
    dx = float((x_end - x64).abs().max() / x64.abs().max())
    dx32 = float((x32.cpu().double() - x64).abs().max() / x64.abs().max())
    note = (f"fp32 reference's own worst {worst_32:.1e}; worst device/reference ratio {worst_ratio:.2f}; {steps} steps, "
            f"{pushes_after_full} evictions; final x vs float64 twin {dx:.1e} (reference {dx32:.1e})")
    record_parity(case, "L-BFGS update vs float64, worst step", worst_dev, max(4.0 * worst_32, 2e-6), note)
    assert dx <= max(4.0 * dx32, 2e-6)


@pytest.mark.parametrize("compact", [True, False], ids=["compact", "twoloop"])
@pytest.mark.parametrize("history", [100, 64, 65, 128])
def test_lbfgs_full_history_matches_oracle(history, compact):
    _run(f"lbfgs m={history} n=20000 {'compact' if compact else 'twoloop'}", 20000, history, compact)


def test_lbfgs_full_history_at_512_image_size():
    """n = 3 x 512 x 512 (the image of configs[1]): 4096-float tiles, 768 per-wave partial sums per dot."""
    _run("lbfgs m=100 n=3x512x512 compact", 3 * 512 * 512, 100, True, extra_steps=30)


def test_lbfgs_full_history_sharded_inner_products():
    """The row-strip form (stv_lbfgsc_dots + host-side sum as the all-reduce + stv_lbfgsc_apply) with
    m > 64: two uneven shards (neither a multiple of the 4096-float tile) vs the same twins."""
    _run("lbfgs m=100 n=11000+9000 sharded", 20000, 100, True, shards=(11000, 9000))

"""One large image across several GPUs: row-strip partition with recomputed halos.

BASELINE configs[4] (3840x2160, Adam, 4 GPUs; not in the reference - SURVEY.md §8(e)).  Rank r
owns the image rows [c0, c1) (multiples of 16, so every max-pool window lies in one strip) and
runs the ordinary single-GPU schedule on the *extended* strip [c0 - HALO, c1 + HALO).  HALO = 160
rows = 2 x the receptive-field radius of conv5_1 (78 px, rounded up to a multiple of 16):

* forward losses are accumulated over core rows only and reduced across ranks:
  the raw Gram sums R = F^T F are all-reduced BEFORE the clamp (the clamp is non-linear), the
  content squared error as one scalar;
* the backward seeds (dF = F.S with the global S, content difference) are applied over the whole
  extended strip: every loss term within one receptive field of a core pixel is present and was
  computed from correct features (they are a further receptive field away from the artificial
  strip boundary), so the gradient of the core rows is exact; halo rows of the gradient are dropped;
* per step the ranks exchange only: 5 raw Gram matrices (2.4 MB), one scalar, and the updated core
  strips of the image (all-gather).  No per-layer halo exchange: 26 latency-bound exchanges per
  closure are traded for recomputing the halo rows (efficiency = core / extended rows, 62 % at 4K
  on 4 GPUs).

This class (``SpatialShard``) is the recompute variant, kept for comparison; ``HaloShard`` below is the
partition BASELINE configs[4] describes - a 1-row halo exchanged before every 3x3 convolution, nothing
recomputed, Adam and L-BFGS (inner products all-reduced).
"""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from . import ops, plan
from .optimizers import HipAdam

HALO_ROWS = 160


def strip_rows(H: int, rank: int, world: int) -> tuple[int, int, int, int]:
    """(core_begin, core_end, ext_begin, ext_end) for `rank`; strip heights are multiples of 16."""
    base = -(-H // world)
    base = -(-base // 16) * 16
    c0 = min(H, rank * base)
    c1 = min(H, c0 + base)
    if c1 <= c0:
        msg = f"image of {H} rows is too small for {world} strips of at least 16 rows"
        raise ValueError(msg)
    return c0, c1, max(0, c0 - HALO_ROWS), min(H, c1 + HALO_ROWS)


class SpatialShard:
    """This rank's share of one image: buffers, the two programs around the all-reduce, Adam state."""

    def __init__(self, layers: list[nn.Module], style_at: list[int], content_at: list[int],
                 content_img: torch.Tensor, style_targets: list[torch.Tensor], *, dtype: torch.dtype,
                 style_w: float, content_w: float) -> None:
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        dev = content_img.device
        _, _, H, W = content_img.shape
        self.H, self.W = H, W
        self.c0, self.c1, self.e0, self.e1 = strip_rows(H, self.rank, self.world)
        self.style_w, self.content_w = float(style_w), float(content_w)
        He = self.e1 - self.e0
        self.sched = s = plan.Schedule(layers, style_at, content_at, He, W, dtype, dev, with_grad=True)
        s.fuse_first_gram = False        # the Gram sums run over the strip's core rows only (below)
        self.x_ext = torch.zeros(1, 3, He, W, device=dev)
        self.g_ext = torch.zeros(1, 3, He, W, device=dev)

        # down-sampling factor of every activation (doubles at each max-pool)
        factor: dict[int, int] = {}
        for nd in s.nodes:
            k_src = 1 if nd.src is None else factor[id(nd.src)]
            factor[id(nd.dst)] = k_src * 2 if nd.kind == "pool" else k_src
        self._factor = factor

        def core_rows(buf: plan.Buf) -> tuple[int, int]:
            k = factor[id(buf)]
            r0 = (self.c0 - self.e0) // k
            r1 = buf.H if self.c1 >= H else (self.c1 - self.e0) // k
            return r0, r1

        # --- content targets: features of the content image on this extended strip ---
        plan.Program(s.forward_ops(content_img[:, :, self.e0:self.e1].contiguous()), s._keep).run()
        for tap in s.content_taps:
            tap.target = tap.buf.act.clone()
        n_terms = len(s.style_taps) + len(s.content_taps)
        self.r_local = [torch.zeros(t.buf.C, t.buf.C, device=dev) for t in s.style_taps]
        self.flat = torch.zeros(sum(r.numel() for r in self.r_local) + len(s.content_taps), device=dev)
        self.parts_c = torch.zeros(ops._lib.CONTENT_LOSS_PARTS * max(1, len(s.content_taps)), device=dev)
        self.losses = torch.zeros(max(n_terms, 1), device=dev)
        self.scores = torch.zeros(4, device=dev)

        # --- program 1: forward, raw Gram sums and content squared error over CORE rows ---
        p1 = s.forward_ops(self.x_ext)
        self._views = []
        for tap, r in zip(s.style_taps, self.r_local, strict=True):
            r0, r1 = core_rows(tap.buf)
            core = tap.buf.act[r0:r1]
            n = (r1 - r0) * tap.buf.W
            tap.partials = torch.empty(ops.gram_ksplit(n, tap.buf.C), tap.buf.C, tap.buf.C, device=dev)
            p1.append(s._op(op=plan.OP_GRAM_PARTIAL, p0=core, q0=tap.partials, n=n, cin=tap.buf.C))
            # finish in "raw" mode: no clamp, norm 1 -> the mirrored full matrix R
            p1.append(s._op(op=plan.OP_GRAM_FINISH, p0=tap.partials, q0=r, n=n, cin=tap.buf.C, f0=float("inf"),
                            f1=1.0, f2=0.0))
            tap.target = style_targets[tap.order]
            tap.sgrad = torch.zeros(1, tap.buf.C, tap.buf.C, device=dev, dtype=dtype)
        for i, tap in enumerate(s.content_taps):
            r0, r1 = core_rows(tap.buf)
            f, t = tap.buf.act[r0:r1], tap.target[r0:r1]
            p1.append(s._op(op=plan.OP_CONTENT_LOSS, p0=f, p1=t,
                            q0=self.parts_c[i * ops._lib.CONTENT_LOSS_PARTS:], n=f.numel()))
        self.p1 = plan.Program(p1, s._keep)

        # --- program 2: clamp/loss/seed from the GLOBAL R, score combine, backward over the strip ---
        rows, scale = [], []
        off = 0
        self.r_global = [torch.zeros_like(r) for r in self.r_local]
        self.parts2 = torch.zeros(sum(ops.gram_loss_parts(t.buf.C) for t in s.style_taps)
                                  + max(1, len(s.content_taps)), device=dev)
        p2 = []
        for tap, rg in zip(s.style_taps, self.r_global, strict=True):
            k = factor[id(tap.buf)]
            n_global = (H // k) * tap.buf.W
            cnt = ops.gram_loss_parts(tap.buf.C)
            # n = 1 selects a single slab (ksplit = 1): `rg` already is the complete raw Gram
            p2.append(s._op(op=plan.OP_GRAM_FINISH, p0=rg, p1=tap.target, q1=self.parts2[off:], q2=tap.sgrad, n=1,
                            cin=tap.buf.C, f0=plan.GRAM_CLAMP_MAX, f1=float(tap.buf.C * n_global), f2=self.style_w))
            rows.append([off, cnt, 0])
            scale.append(1.0 / float(tap.buf.C * tap.buf.C))
            off += cnt
        self._content_slot = off
        self._content_scale = []
        for tap in s.content_taps:
            k = factor[id(tap.buf)]
            n_global = (H // k) * tap.buf.W * tap.buf.C
            rows.append([off, 1, 1])
            scale.append(1.0 / float(n_global))
            self._content_scale.append(float(tap.buf.act.numel()) / float(n_global))
            off += 1
        self.table = torch.tensor(rows, dtype=torch.int32, device=dev).reshape(-1, 3)
        self.scale = torch.tensor(scale, dtype=torch.float32, device=dev)
        p2.append(s._op(op=plan.OP_LOSS_COMBINE, p0=self.parts2, p1=self.table, p2=self.scale, q0=self.losses,
                        q1=self.scores, cin=n_terms, f0=self.style_w, f1=self.content_w))
        bwd = s.backward_ops(self.g_ext, style_coef=self.style_w, content_coef=self.content_w, coef_dev=None)
        ci = 0
        for o in bwd:     # content gradient is normalised by the GLOBAL element count
            if o.op == plan.OP_CONTENT_GRAD:
                o.f0 = self.content_w * self._content_scale[min(ci, len(self._content_scale) - 1)]
                ci += 1
        self.p2 = plan.Program(p2 + bwd, s._keep)

        # --- this rank's shard of the image and its Adam state ---
        self.x_core = torch.zeros(1, 3, self.c1 - self.c0, W, device=dev, requires_grad=True)
        self.g_core = torch.zeros(1, 3, self.c1 - self.c0, W, device=dev)
        self.adam: HipAdam | None = None

    # ------------------------------------------------------------------------------------------
    def loss_and_grad(self, x_full: torch.Tensor) -> torch.Tensor:
        """Scores [style, content, total] for the whole image; d(total)/dx of the core rows in g_core."""
        self.x_ext.copy_(x_full[:, :, self.e0:self.e1])
        self.p1.run()
        n_r = 0
        for r in self.r_local:
            self.flat[n_r:n_r + r.numel()].copy_(r.reshape(-1))
            n_r += r.numel()
        n_c = len(self.sched.content_taps)
        if n_c:
            self.flat[n_r:n_r + n_c].copy_(self.parts_c.reshape(n_c, -1).double().sum(1).float())
        if self.world > 1:
            dist.all_reduce(self.flat)                   # RCCL over xGMI: 2.4 MB + a scalar per step
        n_r = 0
        for rg in self.r_global:
            rg.copy_(self.flat[n_r:n_r + rg.numel()].reshape(rg.shape))
            n_r += rg.numel()
        if n_c:
            self.parts2[self._content_slot:self._content_slot + n_c].copy_(self.flat[n_r:n_r + n_c])
        self.p2.run()
        self.g_core.copy_(self.g_ext[:, :, self.c0 - self.e0:self.c1 - self.e0])
        return self.scores[:3].clone()

    def adam_step(self, x_full: torch.Tensor, lr: float = 1e-3) -> torch.Tensor:
        """One Adam step on this rank's rows; returns the updated full image (all-gathered)."""
        if self.adam is None:
            with torch.no_grad():
                self.x_core.copy_(x_full[:, :, self.c0:self.c1])
            self.adam = HipAdam([self.x_core], lr=lr)
        scores = self.loss_and_grad(x_full)
        self.x_core.grad = self.g_core
        self.adam.step()
        # strip bounds are a pure function of (H, rank, world): nothing to ask the other ranks.  The
        # updated strips are written into the caller's image in place (no per-step copy of it).
        out = x_full
        if self.world > 1:
            parts = [strip_rows(self.H, r, self.world)[:2] for r in range(self.world)]
            pieces = [torch.zeros(1, 3, b - a, self.W, device=x_full.device) for a, b in parts]
            shapes_equal = len({p.shape for p in pieces}) == 1
            if shapes_equal:
                dist.all_gather(pieces, self.x_core.detach())
            else:                                                   # ragged last strip: per-rank broadcast
                for r, p in enumerate(pieces):
                    if r == self.rank:
                        p.copy_(self.x_core.detach())
                    dist.broadcast(p, src=r)
            with torch.no_grad():
                for (a, b), p in zip(parts, pieces, strict=True):
                    out[:, :, a:b] = p
        else:
            with torch.no_grad():
                out[:, :, self.c0:self.c1] = self.x_core.detach()
        self.last_scores = scores
        return out


# =================================================================================================
# Per-layer halo exchange (BASELINE configs[4] as specified: "spatial-tile partition with halo
# exchange")
# =================================================================================================
class HaloShard:
    """This rank's row strip of ONE image, with a 1-row halo exchanged before every 3x3 convolution.

    Rank r owns image rows ``[c0, c1)`` (multiples of 16: every max-pool window of all four levels
    lies in one strip) and holds, for the image and for every activation and activation gradient,
    its own rows plus one row above and below.  Nothing is recomputed:

    * forward: before a conv reads a buffer, each rank sends its first / last own row of that buffer
      to the rank above / below and receives theirs into its halo rows (zeros at the image border) -
      13 exchanges of ``W x C`` elements per neighbour (e.g. conv1_2: 3840 x 64 bf16 = 0.49 MB);
    * losses: raw Gram sums ``F^T F`` and the content squared error over the rank's own rows, one
      all-reduce of 2.4 MB + a scalar BEFORE the clamp (non-linear); every rank then forms the same
      seeds ``S`` and scores;
    * backward: the gradient buffer a dgrad reads gets the same 1-row exchange first (13 more);
    * update: Adam is element-wise; L-BFGS all-reduces its inner-product table (``HipLBFGS(shard_group=)``).
      The image is never gathered during the run - each step moves two image rows per neighbour.

    Point-to-point goes through ``torch.distributed`` (RCCL send/recv over xGMI with the "nccl" backend;
    with "gloo" - the one-GPU rehearsal - rows are staged through host memory because gloo has no GPU
    point-to-point).  The strips of the other class in this file (recomputed 160-row halos) remain for
    comparison: no per-layer messages, but 38 % redundant compute at 4K on 4 GPUs.
    """

    def __init__(self, layers: list[nn.Module], style_at: list[int], content_at: list[int],
                 content_img: torch.Tensor, style_targets: list[torch.Tensor], *, dtype: torch.dtype,
                 style_w: float, content_w: float, group=None) -> None:
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._host_p2p = dist.is_initialized() and dist.get_backend(group) == "gloo"
        # Whole closure as ONE captured graph (the 27 program segments, the 26 halo exchanges and the Gram
        # all-reduce in between): nothing is issued from Python while a step runs.  With ONE strip the graph holds
        # kernels only and is the default (verified: replay bit-identical to the eager form).  With several strips
        # it would hold RCCL point-to-point and collective operations, a route that has never run on two devices:
        # there the eager segment-by-segment form is the default and the capture is opt-in (STV_SPATIAL_GRAPH=1);
        # an opted-in capture that fails on ANY rank raises on EVERY rank (the outcome is agreed by an all-reduce
        # on the eager path) instead of leaving some ranks replaying and others eager.  The gloo rehearsal stages
        # rows through host memory and cannot be captured.  STV_SPATIAL_GRAPH=0: always eager.
        import os  # noqa: PLC0415
        want = os.environ.get("STV_SPATIAL_GRAPH", "")
        self._graph_forced = want == "1" and self.world > 1
        self._graph_ok = (not self._host_p2p) and want != "0" and (self.world == 1 or want == "1")
        self._graph: torch.cuda.CUDAGraph | None = None
        self._graph_stream: torch.cuda.Stream | None = None
        dev = content_img.device
        _, _, H, W = content_img.shape
        self.H, self.W, self.dtype = H, W, dtype
        self.c0, self.c1, _, _ = strip_rows(H, self.rank, self.world)
        rows = self.c1 - self.c0
        self.style_w, self.content_w = float(style_w), float(content_w)
        self.sched = s = plan.Schedule(layers, style_at, content_at, rows, W, dtype, dev, with_grad=True, halo=1)
        self.x_ext = torch.zeros(1, 3, rows + 2, W, device=dev)
        self.g_ext = torch.zeros(1, 3, rows + 2, W, device=dev)
        self._row_stage = torch.zeros(2, 3, W, device=dev)
        factor: dict[int, int] = {}
        for nd in s.nodes:
            k_src = 1 if nd.src is None else factor[id(nd.src)]
            factor[id(nd.dst)] = k_src * 2 if nd.kind == "pool" else k_src
        # buffers by address: an op names its operands by raw pointer
        self._by_ptr = {self.x_ext.data_ptr(): self.x_ext}
        for nd in s.nodes:
            self._by_ptr[nd.dst.act.data_ptr()] = nd.dst.act
        s.alloc_grads()
        for nd in s.nodes:
            self._by_ptr[nd.dst.grad.data_ptr()] = nd.dst.grad

        # --- content targets: this strip's features of the content image -----------------------
        self.x_ext[:, :, 1:-1].copy_(content_img[:, :, self.c0:self.c1])
        self._fwd = self._segments(s.forward_ops(self.x_ext), forward=True)
        self._run(self._fwd)
        for tap in s.content_taps:
            tap.target = tap.buf.act.clone()

        # --- program 1b: raw Gram sums and content squared error over this rank's own rows --------
        n_terms = len(s.style_taps) + len(s.content_taps)
        self.r_local = [torch.zeros(t.buf.C, t.buf.C, device=dev) for t in s.style_taps]
        self.flat = torch.zeros(sum(r.numel() for r in self.r_local) + len(s.content_taps), device=dev)
        self.parts_c = torch.zeros(ops._lib.CONTENT_LOSS_PARTS * max(1, len(s.content_taps)), device=dev)
        self.losses = torch.zeros(max(n_terms, 1), device=dev)
        self.scores = torch.zeros(4, device=dev)
        p1 = []
        for tap, r in zip(s.style_taps, self.r_local, strict=True):
            own = s.interior(tap.buf.act)
            n = own.shape[0] * tap.buf.W
            tap.partials = torch.empty(ops.gram_ksplit(n, tap.buf.C), tap.buf.C, tap.buf.C, device=dev)
            p1.append(s._op(op=plan.OP_GRAM_PARTIAL, p0=own, q0=tap.partials, n=n, cin=tap.buf.C))
            p1.append(s._op(op=plan.OP_GRAM_FINISH, p0=tap.partials, q0=r, n=n, cin=tap.buf.C, f0=float("inf"),
                            f1=1.0, f2=0.0))            # "raw" finish: no clamp, norm 1 -> the mirrored R
            tap.target = style_targets[tap.order]
            tap.sgrad = torch.zeros(1, tap.buf.C, tap.buf.C, device=dev, dtype=dtype)
        for i, tap in enumerate(s.content_taps):
            f, t = s.interior(tap.buf.act), s.interior(tap.target)
            p1.append(s._op(op=plan.OP_CONTENT_LOSS, p0=f, p1=t,
                            q0=self.parts_c[i * ops._lib.CONTENT_LOSS_PARTS:], n=f.numel()))
        self.p1 = plan.Program(p1, s._keep)

        # --- program 2: clamp / loss / seeds from the GLOBAL sums, score combine ---------------------
        rows_tab, scale = [], []
        off = 0
        self.r_global = [torch.zeros_like(r) for r in self.r_local]
        self.parts2 = torch.zeros(sum(ops.gram_loss_parts(t.buf.C) for t in s.style_taps)
                                  + max(1, len(s.content_taps)), device=dev)
        p2 = []
        for tap, rg in zip(s.style_taps, self.r_global, strict=True):
            k = factor[id(tap.buf)]
            n_global = (H // k) * tap.buf.W
            cnt = ops.gram_loss_parts(tap.buf.C)
            p2.append(s._op(op=plan.OP_GRAM_FINISH, p0=rg, p1=tap.target, q1=self.parts2[off:], q2=tap.sgrad, n=1,
                            cin=tap.buf.C, f0=plan.GRAM_CLAMP_MAX, f1=float(tap.buf.C * n_global), f2=self.style_w))
            rows_tab.append([off, cnt, 0])
            scale.append(1.0 / float(tap.buf.C * tap.buf.C))
            off += cnt
        self._content_slot = off
        content_scale = []
        for tap in s.content_taps:
            k = factor[id(tap.buf)]
            n_global = (H // k) * tap.buf.W * tap.buf.C
            rows_tab.append([off, 1, 1])
            scale.append(1.0 / float(n_global))
            content_scale.append(float(tap.buf.act.numel()) / float(n_global))   # the op divides by ITS element count
            off += 1
        self.table = torch.tensor(rows_tab, dtype=torch.int32, device=dev).reshape(-1, 3)
        self.scale = torch.tensor(scale, dtype=torch.float32, device=dev)
        p2.append(s._op(op=plan.OP_LOSS_COMBINE, p0=self.parts2, p1=self.table, p2=self.scale, q0=self.losses,
                        q1=self.scores, cin=n_terms, f0=self.style_w, f1=self.content_w))
        self.p2 = plan.Program(p2, s._keep)
        bwd = s.backward_ops(self.g_ext, style_coef=self.style_w, content_coef=self.content_w, coef_dev=None)
        ci = 0
        for o in bwd:
            if o.op == plan.OP_CONTENT_GRAD:
                o.f0 = self.content_w * content_scale[min(ci, len(content_scale) - 1)]
                ci += 1
        self._bwd = self._segments(bwd, forward=False)

        self.x_core = torch.zeros(1, 3, rows, W, device=dev, requires_grad=True)
        self.g_core = torch.zeros(1, 3, rows, W, device=dev)
        self._opt = None
        self.exchanges_per_closure = sum(1 for e, _ in self._fwd + self._bwd if e is not None)

    # -- op list -> [(buffer to exchange first | None, program)] ----------------------------------------
    def _segments(self, op_list: list, *, forward: bool) -> list:
        """Cut before every op that reads a 3x3 neighbourhood (a conv / first-layer op): its input
        buffer ``p0`` needs valid halo rows."""
        conv_ops = (plan.OP_CONV_FIRST_FWD,) if forward else (plan.OP_CONV_FIRST_DGRAD,)
        segs, cur, cur_ex = [], [], None
        for o in op_list:
            needs = (o.op == plan.OP_CONV and o.taps == 9) or o.op in conv_ops
            if needs:
                if cur:
                    segs.append((cur_ex, plan.Program(cur, self.sched._keep)))
                cur, cur_ex = [], self._by_ptr[int(o.p0)]
            cur.append(o)
        if cur:
            segs.append((cur_ex, plan.Program(cur, self.sched._keep)))
        return segs

    def _run(self, segs: list) -> None:
        for buf, prog in segs:
            if buf is not None:
                self._exchange(buf)
            prog.run()

    # -- halo exchange ------------------------------------------------------------------------------
    def _exchange(self, buf: torch.Tensor) -> None:
        """Fill the halo rows of ``buf`` ([rows+2, W, C] NHWC, or the NCHW image [1,3,rows+2,W])."""
        up, down = self.rank - 1, self.rank + 1
        if buf.dim() == 4:               # image: a row is three W-float pieces -> stage it
            st = self._row_stage
            st[0].copy_(buf[0, :, 1]); st[1].copy_(buf[0, :, -2])
            recv_top, recv_bot = torch.empty_like(st[0]), torch.empty_like(st[0])
            self._p2p(st[0], st[1], recv_top, recv_bot)
            buf[0, :, 0].copy_(recv_top) if up >= 0 else buf[0, :, 0].zero_()
            buf[0, :, -1].copy_(recv_bot) if down < self.world else buf[0, :, -1].zero_()
            return
        self._p2p(buf[1], buf[-2], buf[0], buf[-1])
        if up < 0:
            buf[0].zero_()
        if down >= self.world:
            buf[-1].zero_()

    def _peer(self, group_rank: int) -> int:
        """Global rank of a rank of ``self.group``: isend / irecv / P2POp name their peers by GLOBAL rank."""
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)

    def _p2p(self, send_top: torch.Tensor, send_bot: torch.Tensor, recv_top: torch.Tensor, recv_bot: torch.Tensor) -> None:
        up, down = self.rank - 1, self.rank + 1
        if self.world == 1:
            return
        if self._host_p2p:
            # gloo (rehearsal on one GPU): no GPU point-to-point -> through host memory
            reqs, back = [], []
            for peer, dst in ((up, recv_top), (down, recv_bot)):
                if 0 <= peer < self.world:
                    h = torch.empty(dst.shape, dtype=dst.dtype).view(torch.uint8)
                    reqs.append(dist.irecv(h, src=self._peer(peer), group=self.group))
                    back.append((dst, h))
            for peer, src in ((up, send_top), (down, send_bot)):
                if 0 <= peer < self.world:
                    reqs.append(dist.isend(src.detach().cpu().contiguous().view(torch.uint8), dst=self._peer(peer), group=self.group))
            for r in reqs:
                r.wait()
            for dst, h in back:
                dst.copy_(h.view(dst.dtype).reshape(dst.shape))
            return
        # RCCL send/recv wants contiguous tensors: a row of an NHWC activation is, a row of the NCHW image
        # ([1, 3, W] out of [1, 3, H, W]) is not - those go through a small contiguous staging row
        p2p, back = [], []
        for peer, src, dst in ((up, send_top, recv_top), (down, send_bot, recv_bot)):
            if 0 <= peer < self.world:
                rbuf = dst if dst.is_contiguous() else torch.empty(dst.shape, dtype=dst.dtype, device=dst.device)
                if rbuf is not dst:
                    back.append((dst, rbuf))
                p2p.append(dist.P2POp(dist.irecv, rbuf.view(torch.uint8), self._peer(peer), self.group))
                p2p.append(dist.P2POp(dist.isend, src.detach().contiguous().view(torch.uint8), self._peer(peer), self.group))
        for r in dist.batch_isend_irecv(p2p):
            r.wait()
        for dst, rbuf in back:
            dst.copy_(rbuf)

    # -- one evaluation -------------------------------------------------------------------------------
    def loss_and_grad(self) -> torch.Tensor:
        """Scores [style, content, total] of the whole image at the current ``x_core``; d(total)/dx of
        this rank's rows lands in ``g_core``.  After a first eager evaluation (one-time kernel attributes,
        allocations) the whole closure is captured once and replayed as one graph."""
        if not self._graph_ok:
            return self._closure_eager()
        if self._graph is None:
            out = self._closure_eager()                      # warm-up, and this call's result
            err: Exception | None = None
            try:
                self._capture_closure()
            except RuntimeError as exc:   # a runtime that cannot capture this stream work (torch.AcceleratorError is one)
                err = exc
            ok = torch.tensor([0.0 if err is not None else 1.0], device=self.x_core.device)
            if self.world > 1:                               # every rank replays, or none does
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if float(ok) < 1.0:
                self._graph_ok, self._graph = False, None
                if self._graph_forced:
                    msg = f"STV_SPATIAL_GRAPH=1: the row-strip closure could not be captured on every rank ({err})"
                    raise RuntimeError(msg) from err
                from .logging_utils import logger  # noqa: PLC0415
                logger.warning("row-strip closure could not be captured as a graph (%s): running it segment by segment", err)
            return out
        self._graph.replay()
        return self.scores[:3].clone()

    @property
    def route(self) -> str:
        """``graph`` once the closure replays as one captured graph, ``eager`` while / when it runs segment by segment."""
        return "graph" if self._graph is not None else "eager"

    def _capture_closure(self) -> None:
        torch.cuda.synchronize(self.x_core.device)
        self._graph_stream = torch.cuda.Stream(device=self.x_core.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=self._graph_stream):
            self._closure_body()
        self._graph = graph

    def _closure_eager(self) -> torch.Tensor:
        self._closure_body()
        return self.scores[:3].clone()

    def _closure_body(self) -> None:
        self.x_ext[:, :, 1:-1].copy_(self.x_core.detach())
        self._run(self._fwd)
        self.p1.run()
        n_r = 0
        for r in self.r_local:
            self.flat[n_r:n_r + r.numel()].copy_(r.reshape(-1))
            n_r += r.numel()
        n_c = len(self.sched.content_taps)
        if n_c:
            self.flat[n_r:n_r + n_c].copy_(self.parts_c.reshape(n_c, -1).double().sum(1).float())
        if self.world > 1:
            dist.all_reduce(self.flat, group=self.group)      # 2.4 MB + a scalar per step, before the clamp
        n_r = 0
        for rg in self.r_global:
            rg.copy_(self.flat[n_r:n_r + rg.numel()].reshape(rg.shape))
            n_r += rg.numel()
        if n_c:
            self.parts2[self._content_slot:self._content_slot + n_c].copy_(self.flat[n_r:n_r + n_c])
        self.p2.run()
        self._run(self._bwd)
        self.g_core.copy_(self.g_ext[:, :, 1:-1])

    def set_image(self, x_full: torch.Tensor) -> None:
        with torch.no_grad():
            self.x_core.copy_(x_full[:, :, self.c0:self.c1])

    def step(self, optimizer: str = "adam", lr: float | None = None) -> torch.Tensor:
        """One optimisation step on this rank's rows (``adam`` | ``lbfgs``); returns the scores."""
        if self._opt is None:
            if optimizer == "adam":
                self._opt = HipAdam([self.x_core], lr=1e-3 if lr is None else lr)
            elif optimizer == "lbfgs":
                from .optimizers import HipLBFGS  # noqa: PLC0415
                self._opt = HipLBFGS([self.x_core], lr=1.0 if lr is None else lr,
                                     shard_group=(self.group if self.group is not None else True) if self.world > 1 else None)
            else:
                msg = f"unknown optimizer {optimizer!r}"
                raise ValueError(msg)
        scores = {}

        def closure():
            scores["v"] = self.loss_and_grad()
            self.x_core.grad = self.g_core
            return scores["v"][2]
        self._opt.step(closure)
        self.last_scores = scores["v"]
        return scores["v"]

    def gather_image(self) -> torch.Tensor:
        """The whole image on every rank (end of a run; not on the step path)."""
        if self.world == 1:
            return self.x_core.detach().clone()
        bounds = [strip_rows(self.H, r, self.world)[:2] for r in range(self.world)]
        pieces = [torch.zeros(1, 3, b - a, self.W, device=self.x_core.device) for a, b in bounds]
        for r, p in enumerate(pieces):          # strips may differ in height: one broadcast per strip
            if r == self.rank:
                p.copy_(self.x_core.detach())
            dist.broadcast(p, src=r if self.group is None else dist.get_global_rank(self.group, r), group=self.group)
        return torch.cat(pieces, dim=2)

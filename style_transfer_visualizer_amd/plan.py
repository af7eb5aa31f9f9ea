"""Lower a sliced VGG feature stack + loss taps to ``stv_op_t`` command buffers.

This is the host half of the hot path: it runs once per (model, image size),
allocates every activation / gradient / workspace buffer through PyTorch, and
emits flat op arrays that ``libstv_hip.so`` executes each optimisation step
(``stv_program_run``).  It restates, as a schedule, what autograd does for the
reference's ``StyleContentModel.forward`` + ``loss.backward()``
(/root/reference/src/style_transfer_visualizer/core_model.py:297-328,
optimization.py:292-313):

* conv -> ReLU pairs are fused (ReLU in the producer's epilogue) unless the
  conv output itself is tapped; then the consumer applies ReLU while staging;
* every gradient w.r.t. a stored activation is written exactly once by its
  consumer (with the ReLU mask fused in the epilogue) and tap gradients
  (Gram product, content difference) accumulate on top;
* Gram backward is ``dF = F . S`` with the symmetric seed ``S`` produced by
  ``stv_gram_finish``; it runs as a 1x1 conv on the matrix cores.
"""
from __future__ import annotations

import os

import ctypes
from dataclasses import dataclass, field

import torch
from torch import nn

from . import _lib, ops
from ._lib import (ACCUM, MASK, POOL_IDX, POOL_ONLY, POOL_ROUTE, W_BLOCKED, OP_GRAM_MULTI, OP_CONTENT_GRAD, OP_CONTENT_LOSS, OP_CONV, OP_CONV_FIRST_DGRAD,
                   OP_CONV_FIRST_FWD, OP_GRAM_FINISH, OP_GRAM_PARTIAL, OP_LOSS_COMBINE, OP_POOL_BWD,
                   OP_POOL_FWD, OP_RELU_BWD, OP_RELU_FWD, RELU_IN, RELU_OUT, StvOp)

GRAM_CLAMP_MAX = 5e5  # reference constants.py:15


@dataclass
class Buf:
    """One materialised activation (NHWC) and, for the backward pass, its gradient."""

    H: int
    W: int
    C: int
    act: torch.Tensor
    relu_fused: bool = False       # stored value is relu(z)
    taps: list = field(default_factory=list)
    grad: torch.Tensor | None = None
    stored: bool = True            # False: the forward program does not write `act` (a pre-pool map nobody reads: STV_POOL_ONLY)


@dataclass
class Node:
    kind: str                      # conv_first | conv | pool | relu
    src: Buf | None
    dst: Buf
    relu_in: bool = False
    layer: int = -1
    wf: torch.Tensor | None = None
    wb: torch.Tensor | None = None
    bias: torch.Tensor | None = None
    cin: int = 0
    idx: torch.Tensor | None = None     # pool nodes fused into a conv: arg-max byte map for the backward


@dataclass
class Tap:
    kind: str                      # style | content
    order: int                     # index inside its kind (block order)
    buf: Buf
    target: torch.Tensor | None = None
    # style
    partials: torch.Tensor | None = None
    partials_fused: bool = False      # the producing conv fills `partials` itself (stv_conv_first_fwd_gram)
    sgrad: torch.Tensor | None = None
    parts_off: int = 0
    parts_cnt: int = 0


def _check_layer(layer: nn.Module, idx: int) -> str:
    if isinstance(layer, nn.Conv2d):
        ok = (layer.kernel_size == (3, 3) and layer.stride == (1, 1) and layer.padding == (1, 1)
              and layer.dilation == (1, 1) and layer.groups == 1 and layer.padding_mode == "zeros")
        if not ok:
            msg = f"layer {idx}: only Conv2d(k=3, stride=1, padding=1) has a HIP kernel"
            raise RuntimeError(msg)
        return "conv"
    if isinstance(layer, nn.ReLU):
        return "relu"
    if isinstance(layer, nn.MaxPool2d):
        ks = layer.kernel_size if isinstance(layer.kernel_size, tuple) else (layer.kernel_size,) * 2
        stv = layer.stride if isinstance(layer.stride, tuple) else (layer.stride,) * 2
        pad = layer.padding if isinstance(layer.padding, tuple) else (layer.padding,) * 2
        if ks != (2, 2) or stv != (2, 2) or pad != (0, 0) or layer.ceil_mode:
            msg = f"layer {idx}: only MaxPool2d(2, 2) has a HIP kernel"
            raise RuntimeError(msg)
        return "pool"
    msg = f"layer {idx}: {type(layer).__name__} has no HIP kernel on this path"
    raise RuntimeError(msg)


class Schedule:
    """Buffers + forward/backward op lists for one image size."""

    def __init__(self, layers: list[nn.Module], style_at: list[int], content_at: list[int],
                 H: int, W: int, dtype: torch.dtype, device: torch.device, *, with_grad: bool, halo: int = 0) -> None:
        """``halo`` = 1: the schedule of one ROW STRIP of a larger image (spatial.py).  ``H`` is the
        strip's own row count; every activation (and the image) carries ``halo`` extra rows above and
        below that the owner fills before each 3x3 convolution reads them (neighbour's rows, or zeros
        at the image border).  Convolutions run over the whole buffer - their output in the halo
        rows is meaningless and is replaced by the next exchange - while pooling works on the
        strip's own rows only (strip heights are multiples of 16, so no window straddles two strips)."""
        self.H, self.W, self.dtype, self.device = H, W, dtype, device
        self.halo = halo
        self.fuse_first_gram = True      # off where the Gram runs over a sub-range of the buffer (spatial.SpatialShard)
        self.nodes: list[Node] = []
        self.style_taps: list[Tap] = []
        self.content_taps: list[Tap] = []
        self.with_grad = with_grad
        self._keep: list = []      # tensors referenced by raw pointer from op arrays
        self._lower_forward(layers, style_at, content_at)

    # ------------------------------------------------------------------ forward walk
    def _new_buf(self, H: int, W: int, C: int) -> Buf:
        if self.halo:
            act = torch.zeros(H + 2 * self.halo, W, C, device=self.device, dtype=self.dtype)
            return Buf(H + 2 * self.halo, W, C, act)
        act = torch.empty(H, W, C, device=self.device, dtype=self.dtype)
        return Buf(H, W, C, act)

    def interior(self, t: torch.Tensor) -> torch.Tensor:
        """The strip's own rows of an NHWC buffer (the whole buffer without halos)."""
        return t[self.halo:t.shape[0] - self.halo] if self.halo else t

    def _lower_forward(self, layers: list[nn.Module], style_at: list[int], content_at: list[int]) -> None:
        tapped = set(style_at) | set(content_at)
        last = max(tapped)
        kinds = [_check_layer(layer, i) for i, layer in enumerate(layers[:last + 1])]
        cur: Buf | None = None           # None = the NCHW fp32 image
        pending_relu = False
        H, W = self.H, self.W
        i = 0
        while i <= last:
            kind = kinds[i]
            out_idx = i
            if kind == "conv":
                conv: nn.Conv2d = layers[i]
                w = conv.weight.detach().to(self.device, torch.float32)
                bias = (conv.bias.detach().to(self.device, torch.float32).contiguous()
                        if conv.bias is not None else None)
                cout, cin = w.shape[:2]
                dst = self._new_buf(H, W, cout)
                if cur is None:
                    if pending_relu:
                        msg = "a ReLU in front of the first convolution is not supported"
                        raise RuntimeError(msg)
                    node = Node("conv_first", None, dst, layer=i, wf=ops.pack_weights_fwd(w), bias=bias, cin=cin)
                    if w.is_cuda:
                        node.wb = ops.conv_first_pack(node.wf)   # frozen weights: kernel-side packing, once
                else:
                    wf = ops.pack_weights_fwd(w).to(self.dtype)
                    wb = ops.pack_weights_bwd(w).to(self.dtype) if self.with_grad else None
                    # matrix-core shapes take K-blocked weights (W_BLOCKED is derived from w.dim() == 4)
                    blocked = os.environ.get("STV_W_BLOCKED", "1") != "0"     # A/B knob
                    if blocked and ops.conv_uses_mfma(H, W, cin, cout, self.dtype):
                        wf = ops.block_weights(wf)
                    if blocked and wb is not None and ops.conv_uses_mfma(H, W, cout, cin, self.dtype):
                        wb = ops.block_weights(wb)
                    if w.is_cuda:      # measure the tile configurations of this layer's shapes once
                        ops.conv_tune(H, W, cin, cout, 9, self.dtype)
                        if self.with_grad:
                            ops.conv_tune(H, W, cout, cin, 9, self.dtype)
                    node = Node("conv", cur, dst, relu_in=pending_relu, layer=i, wf=wf, wb=wb, bias=bias, cin=cin)
                pending_relu = False
                if i + 1 <= last and kinds[i + 1] == "relu" and i not in tapped:
                    dst.relu_fused = True      # ReLU runs in this conv's epilogue
                    out_idx = i + 1
                self.nodes.append(node)
                cur = dst
            elif kind == "relu":
                if cur is None:
                    msg = "ReLU directly on the input image is not supported"
                    raise RuntimeError(msg)
                if i in tapped:
                    dst = self._new_buf(cur.H, cur.W, cur.C)
                    self.nodes.append(Node("relu", cur, dst, layer=i))
                    cur = dst
                    pending_relu = False
                else:
                    pending_relu = True        # applied by the consumer while staging
            else:  # pool
                if cur is None:
                    msg = "MaxPool directly on the input image is not supported"
                    raise RuntimeError(msg)
                H, W = H // 2, W // 2
                if H < 1 or W < 1:
                    msg = "image too small for this many pooling layers"
                    raise RuntimeError(msg)
                dst = self._new_buf(H, W, cur.C)
                self.nodes.append(Node("pool", cur, dst, layer=i))
                cur = dst
                if pending_relu and i in tapped:   # relu(pool(z)) must exist as data
                    r = self._new_buf(H, W, cur.C)
                    self.nodes.append(Node("relu", cur, r, layer=i))
                    cur = r
                    pending_relu = False
            for idx in range(i, out_idx + 1):
                if idx in style_at:
                    assert cur is not None
                    tap = Tap("style", len(self.style_taps), cur)
                    cur.taps.append(tap)
                    self.style_taps.append(tap)
                if idx in content_at:
                    assert cur is not None
                    tap = Tap("content", len(self.content_taps), cur)
                    cur.taps.append(tap)
                    self.content_taps.append(tap)
            i = out_idx + 1

    # ------------------------------------------------------------------ op emission
    def _op(self, **kw) -> StvOp:
        op = StvOp()
        op.dtype = ops.dtype_code(self.dtype)
        for k, v in kw.items():
            if isinstance(v, torch.Tensor):
                self._keep.append(v)
                v = v.data_ptr()
            setattr(op, k, v)
        return op

    def forward_ops(self, x: torch.Tensor, after_node=None) -> list[StvOp]:
        """Forward schedule; ``after_node(node)`` may return extra ops to splice in right after a
        node's op (loss-side work that only needs that node's output)."""
        out = []
        fuse_pool = os.environ.get("STV_FUSE_POOL", "1") != "0" and not self.halo     # A/B knob; strips pool their own rows
        fused: set[int] = set()          # pool nodes whose work rides in the preceding conv's epilogue
        for k, nd in enumerate(self.nodes):
            d = nd.dst
            # conv (ReLU in its epilogue) -> pool: one launch writes both maps.  Only where the conv
            # runs on the matrix cores, and not when the conv output itself is tapped pre-ReLU.
            nxt = self.nodes[k + 1] if k + 1 < len(self.nodes) else None
            pool_dst = None
            if (fuse_pool and nd.kind == "conv" and nxt is not None and nxt.kind == "pool" and nxt.src is d
                    and d.relu_fused and nd.wf.dim() == 4 and d.act.is_cuda):
                pool_dst = nxt.dst.act
                fused.add(id(nxt))
                # the fused epilogue also leaves the arg-max map the pooling backward needs: one byte
                # per pooled element instead of re-reading the full-resolution activation
                if self.with_grad and nxt.idx is None and os.environ.get("STV_POOL_IDX", "1") != "0":
                    nxt.idx = torch.empty(nxt.dst.H, nxt.dst.W, nxt.dst.C, device=self.device, dtype=torch.uint8)
            if id(nd) in fused:
                if after_node is not None:
                    out += after_node(nd)
                continue
            if nd.kind == "conv_first":
                # a tapped first layer leaves the Gram slabs of its own output (no second pass over the map)
                slabs = None
                tap = next((t for t in d.taps if t.kind == "style"), None)
                if (tap is not None and nd.wb is not None and d.act.is_cuda and not self.halo and self.fuse_first_gram
                        and os.environ.get("STV_FUSE_GRAM_FIRST", "1") != "0"
                        and ops.conv_first_gram_supported(d.H, d.W, nd.cin, d.C, self.dtype)
                        # one slab per workgroup: pays once a workgroup walks >= 4 tiles of 8 x 32 pixels (1024^2: 8,
                        # -11 us; at 512^2, 2 tiles each, the slab reduction costs what the separate pass did)
                        and (os.environ.get("STV_FUSE_GRAM_FIRST") == "2"
                             or d.H * d.W >= 4 * 256 * ops.gram_ksplit(d.H * d.W, d.C))):
                    if tap.partials is None:
                        tap.partials = torch.empty(ops.gram_ksplit(d.H * d.W, d.C), d.C, d.C, device=self.device,
                                                   dtype=torch.float32)
                    slabs = tap.partials
                if tap is not None:
                    tap.partials_fused = slabs is not None      # (re-decided per build: the switch may have changed)
                out.append(self._op(op=OP_CONV_FIRST_FWD, p0=x, p1=nd.wf, p2=nd.bias, p3=nd.wb, q0=d.act, q1=slabs,
                                    H=d.H, W=d.W, cin=nd.cin, cout=d.C))
            elif nd.kind == "conv":
                flags = ((RELU_IN if nd.relu_in else 0) | (RELU_OUT if d.relu_fused else 0)
                         | (W_BLOCKED if nd.wf.dim() == 4 else 0))
                # The full-resolution map of a conv with the pool in its epilogue is dead in the bf16 step: the forward
                # pass continues from the pooled map, the backward pass routes through the arg-max byte map (its bit 2 is
                # the ReLU mask), and no tap sits on it - so it is not stored (conv1_2 at 1024^2: 134 MB of the
                # kernel's 312; STV_SKIP_PREPOOL=0 stores it, e.g. for the tests that look at every stored tensor).
                # fp32 (parity mode) keeps it: the parity tests read ReLU / arg-max decisions off the stored maps.
                d.stored = True
                if (pool_dst is not None and not d.taps and self.dtype == torch.bfloat16
                        and (not self.with_grad or nxt.idx is not None)
                        and os.environ.get("STV_SKIP_PREPOOL", "1") != "0"):
                    flags |= POOL_ONLY
                    d.stored = False
                out.append(self._op(op=OP_CONV, p0=nd.src.act, p1=nd.wf, p2=nd.bias, q0=d.act, q1=pool_dst,
                                    q2=nxt.idx if pool_dst is not None else None, H=d.H,
                                    W=d.W, cin=nd.cin, cout=d.C, taps=9, flags=flags))
            elif nd.kind == "pool":
                src_i = self.interior(nd.src.act)
                out.append(self._op(op=OP_POOL_FWD, p0=src_i, q0=self.interior(d.act), H=src_i.shape[0], W=nd.src.W, cin=d.C))
            else:
                out.append(self._op(op=OP_RELU_FWD, p0=nd.src.act, q0=d.act, n=d.act.numel()))
            if after_node is not None:
                out += after_node(nd)
        return out

    def gram_ops(self, tap: Tap, *, gram_out: torch.Tensor | None, target: torch.Tensor | None,
                 loss_part: torch.Tensor | None, sgrad: torch.Tensor | None, coef: float,
                 coef_dev: torch.Tensor | None, partial: bool = True, finish: bool = True) -> list[StvOp]:
        b = tap.buf
        n = b.H * b.W
        if tap.partials is None:
            tap.partials = torch.empty(ops.gram_ksplit(n, b.C), b.C, b.C, device=self.device, dtype=torch.float32)
        out = []
        if partial and not tap.partials_fused:
            out.append(self._op(op=OP_GRAM_PARTIAL, p0=b.act, q0=tap.partials, n=n, cin=b.C))
        if not finish:          # (the finish pass runs later, in a batched launch: gram_multi_op with partials_ready)
            return out
        out.append(self._op(op=OP_GRAM_FINISH, p0=tap.partials, p1=target, p2=coef_dev, q0=gram_out, q1=loss_part,
                            q2=sgrad, n=n, cin=b.C, f0=GRAM_CLAMP_MAX, f1=float(b.C * n), f2=coef))
        return out

    def gram_multi_op(self, specs: list[dict]) -> StvOp:
        """One batched Gram chain (stv_gram_multi) for several taps.  ``specs``: per tap the keyword
        arguments of :meth:`gram_ops` plus ``tap``.  The tap table is a host array the program copies."""
        table = (_lib.StvGramTap * len(specs))()

        def ptr(t: torch.Tensor | None) -> int | None:
            if t is None:
                return None
            self._keep.append(t)
            return t.data_ptr()
        for e, sp in zip(table, specs, strict=True):
            tap = sp["tap"]
            b = tap.buf
            n = b.H * b.W
            if tap.partials is None:
                tap.partials = torch.empty(ops.gram_ksplit(n, b.C), b.C, b.C, device=self.device, dtype=torch.float32)
            e.F, e.partials = (None if (tap.partials_fused or sp.get("partials_ready")) else ptr(b.act)), ptr(tap.partials)
            e.target, e.gram_out, e.loss_part = ptr(sp.get("target")), ptr(sp.get("gram_out")), ptr(sp.get("loss_part"))
            e.sgrad, e.coef_dev = ptr(sp.get("sgrad")), ptr(sp.get("coef_dev"))
            e.n_pixels, e.channels = n, b.C
            e.clamp_max, e.norm, e.coef = GRAM_CLAMP_MAX, float(b.C * n), float(sp.get("coef", 0.0))
        self._keep.append(table)                    # the host array must outlive stv_program_create
        return self._op(op=OP_GRAM_MULTI, p0=ctypes.addressof(table), n=len(specs))

    def _route_candidate(self, nd: Node, producer: dict) -> Node | None:
        """The pool node whose backward can ride in the dgrad of conv `nd` (everything about that decision that does
        not depend on what has been written so far): `nd`'s input is a pooled map with an arg-max byte map, no tap on
        it, no mask, no Gram term on this launch (`backward_ops` adds: nothing wrote the pre-pool gradient yet)."""
        s, d = nd.src, nd.dst
        if nd.kind != "conv" or s is None or os.environ.get("STV_FUSE_POOL_BWD", "1") == "0":
            return None
        mask_src = nd.relu_in or (s.relu_fused and not s.taps)
        pool_nd = producer.get(id(s))
        if (pool_nd is None or pool_nd.kind != "pool" or pool_nd.idx is None or self.dtype != torch.bfloat16
                or nd.wb is None or nd.wb.dim() != 4 or s.taps or mask_src
                or pool_nd.src.H != 2 * s.H or pool_nd.src.W != 2 * s.W
                or 4 * s.act.numel() * s.act.element_size() >= 2 ** 31
                or d.act.numel() * d.act.element_size() >= 2 ** 31):
            return None
        return pool_nd

    def alloc_grads(self) -> None:
        """Gradient storage of every activation.  The reverse schedule is a chain - the op of node i reads the gradient
        of node i's output and writes that of node i - 1's (i - 2's when the pooling backward rides in a dgrad's
        epilogue) - so on one GPU the gradients ROTATE through a few slabs instead of each owning memory that is touched
        once per step: the walk below follows the reverse schedule and hands every gradient the slab that was released
        LAST (the one the previous launch read), so a dgrad writes into lines that were in use one launch ago and are
        still on chip (Infinity Cache), not into lines last seen a step ago (`tools/cold_probe2.py`: a conv whose
        output range was just written runs 10-15 % faster than one storing to cold memory), and the working set of the
        backward pass shrinks from the sum of all gradients to two or three times the largest.  A buffer with a content
        tap keeps a tensor of its own (its gradient is written during the forward half).  Row strips keep one tensor
        per node (their halo rows are exchanged by address).  `STV_GRAD_ARENA=0`: one tensor per node everywhere;
        `backward_ops` checks, op by op, that no slab is read after somebody else wrote it."""
        todo = [nd for nd in self.nodes if nd.dst.grad is None]
        if not todo:
            return
        chain = all(nd.src is self.nodes[i - 1].dst for i, nd in enumerate(self.nodes) if i > 0)
        mode = os.environ.get("STV_GRAD_ARENA", "1")          # "2": also for host tensors (the host tests walk the allocator)
        arena = (mode != "0" and not self.halo and chain and len(todo) == len(self.nodes)
                 and (mode == "2" or all(nd.dst.act.is_cuda for nd in self.nodes)))
        if not arena:
            for nd in todo:              # strips: halo rows are read before anything wrote them -> start finite
                nd.dst.grad = torch.zeros_like(nd.dst.act) if self.halo else torch.empty_like(nd.dst.act)
            return
        own = {id(t.buf) for t in self.content_taps}
        producer = {id(n.dst): n for n in self.nodes}
        slab_of: dict[int, int] = {}       # id(buf) -> slab number
        free: list[int] = []               # released slabs, the most recently released last
        n_slabs = 0

        def take(buf: Buf) -> None:
            nonlocal n_slabs
            if id(buf) in own or id(buf) in slab_of:
                return
            if free:
                slab_of[id(buf)] = free.pop()
            else:
                slab_of[id(buf)] = n_slabs
                n_slabs += 1

        def release(buf: Buf) -> None:
            if id(buf) in slab_of:
                free.append(slab_of[id(buf)])

        routed: set[int] = set()
        unused: set[int] = set()
        for nd in reversed(self.nodes):
            d = nd.dst
            if id(nd) in routed:           # the pooled map's gradient is never formed
                unused.add(id(d))
                continue
            take(d)                        # (the deepest activation: first written by its own taps)
            if nd.kind != "conv_first":
                pool_nd = self._route_candidate(nd, producer)
                if pool_nd is not None and not any(t.kind == "content" for t in pool_nd.src.taps):
                    take(pool_nd.src)
                    routed.add(id(pool_nd))
                else:
                    take(nd.src)
            release(d)
        rot = [nd.dst for nd in self.nodes if id(nd.dst) in slab_of]
        nbytes = max((b.act.numel() * b.act.element_size() for b in rot), default=0)
        nbytes = (nbytes + 4095) // 4096 * 4096
        self._grad_slabs = torch.empty(max(n_slabs, 1), nbytes, device=self.device, dtype=torch.uint8)
        self._grad_slab_of = slab_of
        for nd in self.nodes:
            b = nd.dst
            if id(b) in slab_of:
                n = b.act.numel() * b.act.element_size()
                b.grad = self._grad_slabs[slab_of[id(b)], :n].view(b.act.dtype).view(b.act.shape)
            elif id(b) in unused:
                b.grad = self._grad_slabs[0, :0].view(b.act.dtype)       # never read or written
            else:
                b.grad = torch.empty_like(b.act)

    def backward_ops(self, x_grad: torch.Tensor, *, style_coef: float, content_coef: float,
                     coef_dev: torch.Tensor | None, prewritten: tuple = ()) -> list[StvOp]:
        """Reverse schedule.  ``coef_dev`` (optional fp32 device vector, style terms
        first) holds upstream d(total)/d(loss_k) for the autograd path.  ``prewritten``: content taps whose
        gradient the forward half already WROTE into their buffer's ``grad`` (stv_content_loss_grad): no
        content-gradient pass for them, and whatever produces that gradient next accumulates onto it."""
        self.alloc_grads()
        out: list[StvOp] = []
        pre = {id(t) for t in prewritten}
        written: set[int] = {id(t.buf) for t in prewritten}

        def acc_flag(buf: Buf) -> int:
            return ACCUM if id(buf) in written else 0

        n_style = len(self.style_taps)
        fuse_gram = os.environ.get("STV_FUSE_GRAM", "1") != "0"      # A/B knob
        fused_taps: set[int] = set()     # style taps whose dF = F.S rides in the dgrad that shares their buffer
        routed: set[int] = set()         # pool nodes whose backward rides in the dgrad of the conv behind them
        producer = {id(n.dst): n for n in self.nodes}
        # gradients that share a slab (alloc_grads): whoever reads one must find its own writer's data there
        slab_of = getattr(self, "_grad_slab_of", {})
        holder: dict[int, int] = {}      # slab -> id(buf) of its last writer

        def wr(buf: Buf) -> torch.Tensor:
            if id(buf) in slab_of:
                if id(buf) in written and holder.get(slab_of[id(buf)]) != id(buf):
                    raise RuntimeError("internal: gradient slab overwritten before its accumulation")
                holder[slab_of[id(buf)]] = id(buf)
            return buf.grad

        def rd(buf: Buf) -> torch.Tensor:
            if id(buf) in slab_of and holder.get(slab_of[id(buf)]) != id(buf):
                raise RuntimeError("internal: gradient slab overwritten before its reader ran")
            return buf.grad

        for nd in reversed(self.nodes):
            d = nd.dst
            if id(nd) in routed:         # its consumer's dgrad already wrote nd.src.grad
                continue
            for tap in d.taps:
                if id(tap) in fused_taps or id(tap) in pre:
                    continue
                if tap.kind == "style":
                    if d.act.is_cuda:
                        ops.conv_tune(d.H, d.W, d.C, d.C, 1, self.dtype)     # cached per shape
                    out.append(self._op(op=OP_CONV, p0=d.act, p1=tap.sgrad, q0=wr(d), H=d.H, W=d.W,
                                        cin=d.C, cout=d.C, taps=1, flags=acc_flag(d)))
                else:
                    cd = coef_dev[n_style + tap.order:] if coef_dev is not None else None
                    out.append(self._op(op=OP_CONTENT_GRAD, p0=d.act, p1=tap.target, p2=cd, q0=wr(d),
                                        n=d.act.numel(), f0=content_coef, flags=acc_flag(d)))
                written.add(id(d))
            if id(d) not in written:
                msg = "internal: activation without any gradient contribution"
                raise RuntimeError(msg)
            if d.relu_fused and d.taps:
                # taps see relu(z): mask the summed gradient once, in place
                out.append(self._op(op=OP_RELU_BWD, p0=d.act, p1=rd(d), q0=d.grad, n=d.act.numel()))
            s = nd.src
            if nd.kind == "conv_first":
                out.append(self._op(op=OP_CONV_FIRST_DGRAD, p0=rd(d), p1=nd.wf, p2=nd.wb, q0=x_grad, H=d.H, W=d.W,
                                    cin=nd.cin, cout=d.C))
                continue
            mask_src = nd.relu_in or (s.relu_fused and not s.taps)
            if nd.kind == "conv":
                flags = (MASK if mask_src else 0) | acc_flag(s) | (W_BLOCKED if nd.wb.dim() == 4 else 0)
                # A Gram tap on the pre-ReLU output s: its gradient term F.S lands on the same buffer
                # as this dgrad.  One launch computes mask * dgrad + F.S (stv_conv_igemm_dual) instead
                # of a second launch that re-reads and re-writes s.grad.
                gram = None
                if (fuse_gram and nd.wb.dim() == 4 and s.act.is_cuda and not (s.relu_fused and s.taps)
                        and s.C % (32 // s.act.element_size()) == 0):
                    gram = next((t for t in s.taps if t.kind == "style" and t.sgrad is not None), None)
                # s is a pooled map (arg-max byte map available, nothing else contributes to its gradient):
                # the dgrad's epilogue routes straight into the pre-pool gradient - no pooled-resolution
                # gradient, no pooling-backward pass
                pool_nd = self._route_candidate(nd, producer)
                if (gram is None and pool_nd is not None and id(s) not in written and id(pool_nd.src) not in written):
                    ps = pool_nd.src
                    if d.act.is_cuda:
                        ops.conv_tune(s.H, s.W, d.C, s.C, ops.TUNE_ROUTE, self.dtype)     # cached per shape
                    rflags = (MASK if (ps.relu_fused and not ps.taps) else 0) | W_BLOCKED | POOL_ROUTE
                    out.append(self._op(op=OP_CONV, p0=rd(d), p1=nd.wb, p2=pool_nd.idx, q1=wr(ps), H=s.H, W=s.W,
                                        cin=d.C, cout=s.C, taps=9, flags=rflags))
                    routed.add(id(pool_nd))
                    written.add(id(s))
                    written.add(id(ps))
                    continue
                if gram is not None:
                    fused_taps.add(id(gram))
                    out.append(self._op(op=OP_CONV, p0=rd(d), p1=nd.wb, p3=s.act if mask_src else None, q0=wr(s),
                                        q2=s.act, q3=gram.sgrad, n=s.C, H=s.H, W=s.W, cin=d.C, cout=s.C, taps=9,
                                        flags=flags))
                else:
                    out.append(self._op(op=OP_CONV, p0=rd(d), p1=nd.wb, p3=s.act if mask_src else None, q0=wr(s),
                                        H=s.H, W=s.W, cin=d.C, cout=s.C, taps=9, flags=flags))
            elif nd.kind == "pool":
                flags = (MASK if (s.relu_fused and not s.taps) else 0) | acc_flag(s)
                if nd.idx is not None:      # written by the forward conv that carried this pool
                    out.append(self._op(op=OP_POOL_BWD, p0=nd.idx, p1=rd(d), q0=wr(s), H=s.H, W=s.W, cin=s.C,
                                        flags=flags | POOL_IDX))
                else:
                    s_i = self.interior(s.act)
                    out.append(self._op(op=OP_POOL_BWD, p0=s_i, p1=self.interior(rd(d)), q0=self.interior(wr(s)),
                                        H=s_i.shape[0], W=s.W, cin=s.C, flags=flags))
            else:  # materialised relu
                out.append(self._op(op=OP_RELU_BWD, p0=s.act, p1=rd(d), q0=wr(s), n=s.act.numel(),
                                    flags=acc_flag(s)))
            written.add(id(s))
        _ = style_coef
        return out


class Program:
    """Owns a ``stv_program`` handle (host object inside libstv_hip.so)."""

    def __init__(self, op_list: list[StvOp], keep: list) -> None:
        self._keep = list(keep)
        arr = (StvOp * len(op_list))(*op_list)
        handle = ctypes.c_void_p()
        lib = _lib.load()
        _lib.check(lib.stv_program_create(arr, len(op_list), ctypes.byref(handle)), "stv_program_create")
        self._handle = handle
        self.n_ops = len(op_list)
        self.op_meta = [(int(o.op), int(o.H), int(o.W), int(o.cin), int(o.cout), int(o.taps), int(o.n))
                        for o in op_list]
        self.op_flags = [int(o.flags) for o in op_list]
        # scratch of the convolutions that split K across workgroups (stv_conv_workspace, include/stv.h): owned here,
        # zeroed once, handed to the library around every run of THIS program (its launches are one chain on one stream;
        # the pointer is baked into the captured graph, which this object outlives)
        self._conv_ws = None
        if any(int(o.op) == _lib.OP_CONV and int(o.dtype) == _lib.STV_BF16 and int(o.taps) == 9 for o in op_list) and torch.cuda.is_available():
            self._conv_ws = torch.zeros(int(lib.stv_conv_workspace_bytes()), dtype=torch.uint8, device=torch.cuda.current_device())

    def _with_workspace(self, call):
        lib = _lib.load()
        if self._conv_ws is None:
            return call()
        lib.stv_conv_workspace(self._conv_ws.data_ptr(), self._conv_ws.numel())
        try:
            return call()
        finally:
            lib.stv_conv_workspace(None, 0)

    def run(self, use_graph: bool = False) -> None:
        lib = _lib.load()
        self._with_workspace(lambda: _lib.check(lib.stv_program_run(
            self._handle, 1 if use_graph else 0, torch.cuda.current_stream().cuda_stream), "stv_program_run"))

    def profile(self, reps: int = 1) -> list[float]:
        """Per-op device milliseconds (HIP events on the current stream); synchronises.  ``reps`` > 1
        launches every op that many times inside its event pair (timing only: buffers are then garbage)."""
        lib = _lib.load()
        out = (ctypes.c_float * self.n_ops)()
        self._with_workspace(lambda: _lib.check(lib.stv_program_profile_reps(
            self._handle, torch.cuda.current_stream().cuda_stream, int(reps), out, self.n_ops), "stv_program_profile_reps"))
        return list(out)

    def __del__(self) -> None:
        try:
            if self._handle:
                _lib.load().stv_program_destroy(self._handle)
                self._handle = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

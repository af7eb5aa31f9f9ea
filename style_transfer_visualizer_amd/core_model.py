"""MI355X-native model + losses behind the reference's ``core_model`` API.

Public surface mirrors /root/reference/src/style_transfer_visualizer/core_model.py
(``gram_matrix`` :29, ``initialize_input`` :66, ``initialize_vgg`` :103,
``create_feature_blocks`` :120, ``StyleContentModel`` :149,
``prepare_model_and_input`` :331) with the same argument meaning and error
texts, but every forward/backward FLOP runs in hand-written HIP kernels from
``libstv_hip.so``; PyTorch only owns memory and streams.  There is no eager or
CPU fallback: tensors must live on the GPU and the library must be built.
"""
from __future__ import annotations

import copy
import hashlib
import os
import threading
from pathlib import Path
from urllib.parse import urlparse

import torch
from torch import nn

from . import _lib, ops, plan, synthetic
from .constants import GRAM_MATRIX_CLAMP_MAX
from .logging_utils import logger

VGG19_WEIGHTS_URL = "https://download.pytorch.org/models/vgg19-dcbb9e9d.pth"
_PRECISIONS = {"fp32": torch.float32, "bf16": torch.bfloat16}

# The two torchvision names the reference imports at module level (core_model.py:12) and its tests patch.  Without
# torchvision (this image) ``vgg19`` is None - ``initialize_vgg`` then builds the same stack itself and loads the
# cached checkpoint - and ``VGG19_Weights`` only carries the URL the cache file name is derived from.
try:
    from torchvision.models import VGG19_Weights, vgg19
except ImportError:
    vgg19 = None

    class _WeightsEntry:
        url = VGG19_WEIGHTS_URL

    class VGG19_Weights:  # noqa: N801
        IMAGENET1K_V1 = _WeightsEntry()
        DEFAULT = IMAGENET1K_V1


def resolve_precision(precision: str | None = None) -> torch.dtype:
    """Activation storage dtype: fp32 = parity mode (default), bf16 = performance mode."""
    name = precision or os.environ.get("STV_PRECISION", "fp32")
    if name not in _PRECISIONS:
        msg = f"Unsupported precision: {name} (expected one of {sorted(_PRECISIONS)})"
        raise ValueError(msg)
    return _PRECISIONS[name]


# --------------------------------------------------------------------------- gram
class _GramFn(torch.autograd.Function):
    """clamp(F F^T, max)/(b*c*h*w) for an NCHW tensor, via the split-K MFMA Gram kernels.

    Any ``b*c`` is accepted, like the reference: the channel axis is zero-padded to the kernels'
    granularity (8) - padded channels only add zero rows/columns, which are sliced away again.
    """

    _PAD = 8

    @staticmethod
    def forward(ctx, tensor: torch.Tensor, clamp_max: float) -> torch.Tensor:
        if tensor.dim() != 4:
            msg = f"gram_matrix expects a [batch, channels, h, w] tensor, got shape {tuple(tensor.shape)}"
            raise ValueError(msg)
        b, c, h, w = tensor.shape
        C, n = b * c, h * w
        Cp = -(-C // _GramFn._PAD) * _GramFn._PAD
        feat = tensor.detach().reshape(C, n).t().float()                 # [N, C] (NHWC view)
        if Cp != C:
            feat = torch.nn.functional.pad(feat, (0, Cp - C))
        feat = feat.contiguous()
        norm = float(C * n)
        partials = ops.gram_partial(feat)
        gram = torch.empty(Cp, Cp, device=tensor.device, dtype=torch.float32)
        raw = torch.empty(Cp, Cp, device=tensor.device, dtype=torch.float32)
        ops.gram_finish(partials, n, Cp, gram_out=gram, clamp_max=clamp_max, norm=norm)
        ops.gram_finish(partials, n, Cp, gram_out=raw, clamp_max=float("inf"), norm=1.0)    # R itself
        ctx.save_for_backward(feat, raw)
        ctx.meta = (b, c, h, w, clamp_max)
        return gram[:C, :C].to(tensor.dtype)

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        feat, raw = ctx.saved_tensors
        b, c, h, w, clamp_max = ctx.meta
        C, n = b * c, h * w
        Cp = raw.shape[0]
        g = torch.zeros(Cp, Cp, device=raw.device, dtype=torch.float32)
        g[:C, :C] = gout.float()
        seed = g * (raw <= clamp_max).float() / float(C * n)           # clamp passes gradient where R <= max
        seed = (seed + seed.t()).contiguous()                           # dF^T = F^T (dR + dR^T)
        d_feat = ops.conv_igemm(feat.reshape(1, n, Cp), seed.reshape(1, Cp, Cp))
        return d_feat.reshape(n, Cp)[:, :C].t().reshape(b, c, h, w).to(gout.dtype), None


def gram_matrix(tensor: torch.Tensor, clamp_max: float = GRAM_MATRIX_CLAMP_MAX) -> torch.Tensor:
    """Gram matrix of ``[batch, channels, h, w]`` features (batch folded into channels).

    Same arithmetic as reference core_model.py:56-63: ``mm(F, F.t()).clamp(max=clamp_max)``
    then ``div(b*c*h*w)``; returns ``[b*c, b*c]``.
    """
    return _GramFn.apply(tensor, float(clamp_max))


# ------------------------------------------------------------------- initialisation
def initialize_input(content_img: torch.Tensor, method: str) -> torch.Tensor:
    """Start image for the optimisation (reference core_model.py:66-100)."""
    if not isinstance(content_img, torch.Tensor):
        msg = f"Expected content_img to be a Tensor, got {type(content_img)}"
        raise TypeError(msg)
    if method == "content":
        start = content_img.clone()
    elif method == "random":
        if content_img.is_cuda:
            # The reference's --device cpu path (the one north_star compares with) draws randn_like on the
            # CPU generator; a draw on the GPU generator could never match it on the same --seed.  Same
            # generator, same shape/dtype -> the same values, then one copy to the device.
            start = torch.randn(content_img.shape, dtype=content_img.dtype).to(content_img.device)
        else:
            start = torch.randn_like(content_img)
    elif method == "white":
        start = torch.ones_like(content_img)
    else:
        msg = f"Unsupported initialization method: {method}"
        raise ValueError(msg)
    return start.requires_grad_(True)  # noqa: FBT003


def build_vgg_features(weights: list[tuple[torch.Tensor, torch.Tensor]] | None = None,
                       cfg: tuple = synthetic.VGG19_CFG) -> nn.Sequential:
    """torchvision-free ``vgg19().features`` topology (conv3x3 pad 1 / ReLU / MaxPool 2x2)."""
    layers: list[nn.Module] = []
    cin = 3
    it = iter(weights) if weights is not None else None
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        conv = nn.Conv2d(cin, int(v), kernel_size=3, padding=1)
        if it is not None:
            w, b = next(it)
            with torch.no_grad():
                conv.weight.copy_(w)
                conv.bias.copy_(b)
        layers += [conv, nn.ReLU(inplace=True)]
        cin = int(v)
    return nn.Sequential(*layers)


def consume_classifier_init() -> None:
    """Advance the CPU generator as constructing the rest of torchvision's VGG19 would.

    The reference builds ``vgg19(weights=IMAGENET1K_V1)`` (core_model.py:114) AFTER seeding
    (main.py:36) and BEFORE drawing the ``random`` start image (core_model.py:92): torchvision's
    ``_vgg`` runs ``VGG(make_layers(cfg), init_weights=False)``, i.e. the default ``reset_parameters`` of
    16 ``Conv2d`` (kaiming_uniform_ weight + uniform_ bias each, in layer order) and then of the classifier's
    ``Linear(25088, 4096)``, ``Linear(4096, 4096)``, ``Linear(4096, 1000)``; the checkpoint is loaded over them.
    ``build_vgg_features`` constructs the same 16 ``Conv2d`` in the same order; this function does the three
    ``Linear`` (by constructing and dropping them: identical consumption by construction, ~0.5 GB / 1 s once
    per run), so that ``randn`` afterwards continues where the reference's would.
    """
    for fan_in, fan_out in ((512 * 7 * 7, 4096), (4096, 4096), (4096, 1000)):
        nn.Linear(fan_in, fan_out)


# initialize_vgg() of the torchvision-free branches costs 0.75 s per call (tools/setup_profile.py: 143 M uniform draws that
# only exist to leave the CPU generator where torchvision's constructor would, 16 weight copies, the checkpoint load) - four
# times the 0.17 s a 200-step run at 512^2 takes, once per image of a batch.  Both results are functions of their inputs: the
# stack of (weight source), the generator state after of (generator state before).  So the process keeps the stack it built
# (a CPU master, handed out as a deep copy) and, per generator state it has seen, the state construction left behind; a
# second call from the same state (style_transfer reseeds per image) restores that state instead of drawing again.
# STV_VGG_CACHE=0: always construct.
_VGG_CACHE: dict = {}
_VGG_CACHE_LOCK = threading.Lock()


def clear_vgg_cache() -> None:
    with _VGG_CACHE_LOCK:
        _VGG_CACHE.clear()


def _vgg_cache_get(key: tuple) -> nn.Module | None:
    if os.environ.get("STV_VGG_CACHE", "1") == "0":
        return None
    digest = hashlib.sha1(torch.get_rng_state().numpy().tobytes()).digest()
    with _VGG_CACHE_LOCK:
        hit = _VGG_CACHE.get(key)
        after = hit["rng"].get(digest) if hit is not None else None
    if after is None:
        return None
    torch.set_rng_state(after.clone())
    return copy.deepcopy(hit["master"])


def _vgg_cache_put(key: tuple, before: torch.Tensor, vgg: nn.Module) -> None:
    if os.environ.get("STV_VGG_CACHE", "1") == "0":
        return
    digest = hashlib.sha1(before.numpy().tobytes()).digest()
    with _VGG_CACHE_LOCK:
        hit = _VGG_CACHE.get(key)
        if hit is None:
            if len(_VGG_CACHE) >= 2:                      # (a stack is 80 MB of host memory)
                _VGG_CACHE.pop(next(iter(_VGG_CACHE)))
            hit = _VGG_CACHE[key] = {"master": copy.deepcopy(vgg), "rng": {}}
        if len(hit["rng"]) >= 16:
            hit["rng"].pop(next(iter(hit["rng"])))
        hit["rng"][digest] = torch.get_rng_state()


def initialize_vgg() -> nn.Module:
    """Frozen VGG19 feature stack (reference core_model.py:103-117).

    Order of preference: torchvision with IMAGENET1K_V1 weights (as the
    reference); the cached ``vgg19-dcbb9e9d.pth`` loaded into the
    torchvision-free topology; deterministic synthetic weights when
    ``STV_SYNTHETIC_WEIGHTS=<seed>`` is set (benchmarks / offline boxes).
    The two torchvision-free branches are served from a per-process cache when weight source AND generator state
    have been seen before (same stack, same generator state afterwards as constructing again).
    """
    cache_dir = Path(torch.hub.get_dir()) / "checkpoints"
    cache_path = cache_dir / Path(urlparse(VGG19_Weights.IMAGENET1K_V1.url).path).name
    synth = os.environ.get("STV_SYNTHETIC_WEIGHTS")
    key = None
    if synth is not None:
        key = ("synthetic", synth)
    elif vgg19 is None and cache_path.exists():
        st = cache_path.stat()
        key = ("checkpoint", str(cache_path), st.st_mtime_ns, st.st_size)
    if key is not None:
        cached = _vgg_cache_get(key)
        if cached is not None:
            logger.info("Using synthetic VGG19 weights (seed %s)", synth) if synth is not None else logger.info(
                "Using cached VGG19 weights at %s", cache_path)
            return cached
    rng_before = torch.get_rng_state()
    if synth is not None:
        logger.info("Using synthetic VGG19 weights (seed %s)", synth)
        vgg = build_vgg_features(synthetic.synthetic_conv_weights(int(synth)))
        consume_classifier_init()
    else:
        if cache_path.exists():
            logger.info("Using cached VGG19 weights at %s", cache_path)
        else:
            logger.info("Downloading VGG19 weights to %s", cache_path)
        if vgg19 is not None:       # torchvision's constructor - or whatever a caller put under that name
            vgg = vgg19(weights=VGG19_Weights.IMAGENET1K_V1).features
        else:
            if not cache_path.exists():
                msg = (f"torchvision is not installed and {cache_path} is absent: place the IMAGENET1K_V1 "
                       "checkpoint there, or set STV_SYNTHETIC_WEIGHTS=<seed> for synthetic weights.")
                raise RuntimeError(msg) from None
            vgg = build_vgg_features()
            consume_classifier_init()
            state = torch.load(cache_path, map_location="cpu", weights_only=True)
            feats = {k[len("features."):]: v for k, v in state.items() if k.startswith("features.")}
            vgg.load_state_dict(feats)
    vgg = vgg.eval()
    for p in vgg.parameters():
        p.requires_grad_(False)  # noqa: FBT003
    if key is not None:
        _vgg_cache_put(key, rng_before, vgg)
    return vgg


def create_feature_blocks(
    vgg: nn.Module,
    style_layers: list[int],
    content_layers: list[int],
) -> tuple[nn.ModuleList, list[int], list[int]]:
    """Slice the stack after every tapped index (reference core_model.py:120-146).

    Returns ``(blocks, content_ids, style_ids)``; trailing untapped layers are
    dropped and every ReLU is made out-of-place.
    """
    blocks = nn.ModuleList()
    content_ids: list[int] = []
    style_ids: list[int] = []
    pending = nn.Sequential()
    for idx, layer in enumerate(vgg.children()):
        pending.add_module(str(idx), nn.ReLU(inplace=False) if isinstance(layer, nn.ReLU) else layer)
        hit_style, hit_content = idx in style_layers, idx in content_layers
        if hit_style or hit_content:
            blocks.append(pending)
            pending = nn.Sequential()
        if hit_style:
            style_ids.append(len(blocks) - 1)
        if hit_content:
            content_ids.append(len(blocks) - 1)
    return blocks, content_ids, style_ids


# --------------------------------------------------------------------------- engine
class _Engine:
    """Buffers + command buffers for one (model, image size, device)."""

    def __init__(self, layers: list[nn.Module], style_at: list[int], content_at: list[int],
                 H: int, W: int, dtype: torch.dtype, device: torch.device) -> None:
        self.layers, self.style_at, self.content_at = layers, style_at, content_at
        self.H, self.W, self.dtype, self.device = H, W, dtype, device
        self.sched = plan.Schedule(layers, style_at, content_at, H, W, dtype, device, with_grad=True)
        s = self.sched
        self.n_style, self.n_content = len(s.style_taps), len(s.content_taps)
        n_terms = self.n_style + self.n_content
        off = 0
        rows, scale = [], []
        for tap in s.style_taps:
            tap.parts_off, tap.parts_cnt = off, ops.gram_loss_parts(tap.buf.C)
            off += tap.parts_cnt
            rows.append([tap.parts_off, tap.parts_cnt, 0])
            scale.append(1.0 / float(tap.buf.C * tap.buf.C))
            tap.sgrad = torch.zeros(1, tap.buf.C, tap.buf.C, device=device, dtype=dtype)
        for tap in s.content_taps:
            tap.parts_off, tap.parts_cnt = off, ops._lib.CONTENT_LOSS_PARTS
            off += tap.parts_cnt
            rows.append([tap.parts_off, tap.parts_cnt, 1])
            scale.append(1.0 / float(tap.buf.act.numel()))
        self.parts = torch.zeros(max(off, 1), device=device, dtype=torch.float32)
        self.table = torch.tensor(rows, dtype=torch.int32, device=device).reshape(-1, 3)
        self.scale = torch.tensor(scale, dtype=torch.float32, device=device)
        self.losses = torch.zeros(max(n_terms, 1), device=device, dtype=torch.float32)
        self.scores = torch.zeros(4, device=device, dtype=torch.float32)
        self.coef_buf = torch.ones(max(n_terms, 1), device=device, dtype=torch.float32)
        self._programs: dict = {}
        self.use_graph = os.environ.get("STV_HIP_GRAPH", "1") != "0"
        # Every evaluation overwrites the shared activation buffers; an autograd backward is only
        # valid against the forward that filled them last.  `generation` counts evaluations.
        self.generation = 0
        self._stage: torch.Tensor | None = None      # fp32 contiguous copy of a non-conforming input

    # -- op list pieces -------------------------------------------------------
    def _tap_loss_ops(self, tap, *, style_coef: float, coef_dev: torch.Tensor | None, with_seed: bool,
                      content_coef: float | None = None) -> list:
        """Loss-side ops of one tap.  They are spliced in right after the op that produces the tapped
        activation, while it is still L2 / Infinity-Cache resident.  STV_SIDE_LANE=1 flags them for
        the executor's second stream (fork after the tapped conv, join at the score combine); measured
        twice on MI355X, it loses 2-5 % inside the captured graph (6 forks + 1 join cost more than the
        ~100 us of overlap they buy), so it stays off by default."""
        s = self.sched
        if tap.kind == "style":
            cd = coef_dev[tap.order:] if coef_dev is not None else None
            ops_ = s.gram_ops(tap, gram_out=None, target=tap.target, loss_part=self.parts[tap.parts_off:],
                              sgrad=tap.sgrad if with_seed else None, coef=style_coef, coef_dev=cd)
        elif content_coef is not None and self._content_fused(tap):
            # loss and gradient of the content term in one pass (the coefficient is known: the fused step)
            ops_ = [s._op(op=plan.OP_CONTENT_LOSS, p0=tap.buf.act, p1=tap.target, q0=self.parts[tap.parts_off:],
                          q1=tap.buf.grad, n=tap.buf.act.numel(), f0=content_coef)]
        else:
            ops_ = [s._op(op=plan.OP_CONTENT_LOSS, p0=tap.buf.act, p1=tap.target,
                          q0=self.parts[tap.parts_off:], n=tap.buf.act.numel())]
        if os.environ.get("STV_SIDE_LANE", "0") == "1":
            # loss-side chains only read the tapped activation and write buffers that nothing reads
            # before the score combine: they may overlap the main chain (executor's second stream)
            for o in ops_:
                o.flags |= _lib.LANE_SIDE
        return ops_

    def _content_fused(self, tap) -> bool:
        """One content tap per buffer, nothing else writing that buffer's gradient first (A/B: STV_FUSE_CONTENT=0)."""
        return (os.environ.get("STV_FUSE_CONTENT", "1") != "0" and tap.buf.grad is not None
                and sum(1 for t in tap.buf.taps if t.kind == "content") == 1)

    def _forward_with_losses(self, x: torch.Tensor, *, style_coef: float, with_seed: bool,
                             content_coef: float | None = None) -> list:
        # Batched loss side: the Gram chain of a tap is a handful of latency-bound launches (partial
        # sums, finish), and side by side in one grid (stv_gram_multi) several taps cost the slowest
        # instead of the sum.  Deferring a tap to the end of the forward pass only pays while its
        # activation is small enough to still sit in the Infinity Cache by then, so the decision is
        # per tap: taps up to 48 MiB are batched behind the last conv, larger ones keep their place
        # right behind their producer.  512^2: all five taps batched; 1024^2: the three deep ones.
        # Same arithmetic and summation order either way (the batched kernels run the per-tap bodies).
        s = self.sched
        mode = os.environ.get("STV_LOSS_BATCH", "auto")
        limit = 48 * 2 ** 20

        def small(tap) -> bool:
            return tap.buf.act.numel() * tap.buf.act.element_size() <= limit
        deferred = [tap for tap in s.style_taps if mode == "1" or (mode == "auto" and small(tap))]
        if len(deferred) < 2 or len(deferred) > 8 or not x.is_cuda:
            deferred = []
        held = {id(tap) for tap in deferred}
        # Round 4: a LARGE tap keeps only its partial-sum pass behind its producer (that pass reads the activation:
        # 67-134 MB at 1024^2); its FINISH pass - a reduction of a few MB of fp32 slabs - joins the batched finish
        # launch at the end of the forward pass instead of being a 6-7 us launch of its own (two launches fewer at
        # 1024^2; STV_GRAM_FIN_LATE=0: finish right behind the partial sums, as before).  Same kernels' bodies, same
        # summation order per tap up to the grouping of a many-slab tap's slabs (8 instead of 32 per partial sum).
        late: list = []
        fin_late = bool(deferred) and os.environ.get("STV_GRAM_FIN_LATE", "1") != "0" and len(s.style_taps) <= 8

        def batched_tail() -> list:
            # built AFTER the forward ops: whether the first layer leaves its own Gram slabs
            # (tap.partials_fused) is decided while those are emitted
            if not deferred:
                return []
            members = sorted(deferred + late, key=lambda t: t.order)
            specs = [dict(tap=tap, target=tap.target, loss_part=self.parts[tap.parts_off:],
                          sgrad=tap.sgrad if with_seed else None, coef=style_coef,
                          partials_ready=any(tap is t for t in late)) for tap in members]
            return [s.gram_multi_op(specs)]
        if deferred and len(deferred) == len(s.style_taps):
            fwd = s.forward_ops(x)
            tail = batched_tail()
            for tap in s.content_taps:
                tail += self._tap_loss_ops(tap, style_coef=style_coef, coef_dev=None, with_seed=with_seed,
                                           content_coef=content_coef)
            return fwd + tail

        def after(node):
            out = []
            for tap in node.dst.taps:
                if id(tap) in held:
                    continue
                if fin_late and tap.kind == "style":
                    if not tap.partials_fused:      # (a first layer that left its own slabs has nothing to do here)
                        out += s.gram_ops(tap, gram_out=None, target=None, loss_part=None, sgrad=None, coef=0.0,
                                          coef_dev=None, finish=False)
                    late.append(tap)
                    continue
                out += self._tap_loss_ops(tap, style_coef=style_coef, coef_dev=None, with_seed=with_seed,
                                          content_coef=content_coef)
            return out
        if os.environ.get("STV_LOSS_INTERLEAVE", "1") == "1":
            fwd = self.sched.forward_ops(x, after_node=after)
            return fwd + batched_tail()
        fwd = self.sched.forward_ops(x)
        inline = []
        for node in self.sched.nodes:
            inline += after(node)
        return fwd + inline + batched_tail()

    def _combine_op(self, style_w: float, content_w: float, score_log: tuple | None = None):
        """``score_log`` = (ring fp32 [3, capacity], device counter int32 [1][, host-visible record count int32 [1]]):
        the combine kernel also appends the three scores to the caller's history ring (stv_loss_combine_log)."""
        ring, count, seq = (tuple(score_log) + (None,))[:3] if score_log is not None else (None, None, None)
        op = self.sched._op(op=plan.OP_LOSS_COMBINE, p0=self.parts, p1=self.table, p2=self.scale, p3=seq,
                            q0=self.losses, q1=self.scores, q2=ring, q3=count, n=ring.shape[1] if ring is not None else 0,
                            cin=self.n_style + self.n_content, f0=style_w, f1=content_w)
        op.flags |= _lib.LANE_JOIN          # first reader of what the loss-side ops wrote
        return op

    def _program(self, key: tuple, builder) -> plan.Program:
        prog = self._programs.get(key)
        if prog is None:
            if len(self._programs) >= 16:     # programs are keyed by buffer addresses: bound the cache (oldest out)
                self._programs.pop(next(iter(self._programs)))
            prog = self._build(self.sched, builder)
            self._programs[key] = prog
        return prog

    @staticmethod
    def _build(sched: plan.Schedule, builder, extra: tuple = ()) -> plan.Program:
        """Program from ``builder()``'s ops, owning exactly the tensors those ops reference by raw
        pointer (the schedule's keep-alive list is restored: it must not grow with every program,
        or every image ever evaluated would stay allocated)."""
        base = len(sched._keep)
        op_list = builder()
        keep = sched._keep[base:] + list(extra)
        del sched._keep[base:]
        return plan.Program(op_list, keep)

    # -- targets ---------------------------------------------------------------
    def stage_input(self, x: torch.Tensor) -> torch.Tensor:
        """A contiguous fp32 view of ``x``: ``x`` itself when it already is one, else a copy into one
        persistent staging buffer (a fresh temporary per call would key a new program + hipGraph by
        its address every time)."""
        xd = x.detach()
        if xd.is_contiguous() and xd.dtype == torch.float32:
            return xd
        if self._stage is None or self._stage.shape != xd.shape:
            self._stage = torch.empty(xd.shape, device=self.device, dtype=torch.float32)
        self._stage.copy_(xd)
        return self._stage

    def capture_content(self, content_img: torch.Tensor) -> list[torch.Tensor]:
        self.generation += 1
        x = content_img.detach().contiguous().float()
        self._build(self.sched, lambda: self.sched.forward_ops(x), (x,)).run()
        targets = [tap.buf.act.clone() for tap in self.sched.content_taps]
        for tap, t in zip(self.sched.content_taps, targets, strict=True):
            tap.target = t
        return targets

    def capture_style(self, style_img: torch.Tensor) -> list[torch.Tensor]:
        self.generation += 1
        x = style_img.detach().contiguous().float()
        Hs, Ws = x.shape[-2:]
        sched = (self.sched if (Hs, Ws) == (self.H, self.W) else
                 plan.Schedule(self.layers, self.style_at, self.content_at, Hs, Ws, self.dtype, self.device,
                               with_grad=False))
        grams = []

        def build():
            op_list = sched.forward_ops(x)
            for tap in sched.style_taps:
                g = torch.empty(tap.buf.C, tap.buf.C, device=self.device, dtype=torch.float32)
                grams.append(g)
                op_list += sched.gram_ops(tap, gram_out=g, target=None, loss_part=None, sgrad=None, coef=0.0,
                                          coef_dev=None)
            return op_list
        self._build(sched, build, (x, sched)).run()
        for tap, g in zip(self.sched.style_taps, grams, strict=True):
            tap.target = g
        return grams

    def bind_targets(self, style_targets: list[torch.Tensor], content_targets: list[torch.Tensor]) -> None:
        """Re-point taps at the model's public target lists (they may have been replaced)."""
        if len(style_targets) != self.n_style or len(content_targets) != self.n_content:
            msg = "target lists do not match the model's style/content layers"
            raise RuntimeError(msg)
        changed = False
        for tap, t in zip(self.sched.style_taps, style_targets, strict=True):
            t = self._as_target(t, (tap.buf.C, tap.buf.C), torch.float32)
            changed |= tap.target is None or tap.target.data_ptr() != t.data_ptr()
            tap.target = t
        for tap, t in zip(self.sched.content_taps, content_targets, strict=True):
            if t.dim() == 4:    # [1,C,H,W] as the reference stores it -> NHWC storage
                t = self._nhwc_of(t)
            t = self._as_target(t, tuple(tap.buf.act.shape), self.dtype)
            changed |= tap.target is None or tap.target.data_ptr() != t.data_ptr()
            tap.target = t
        if changed:
            self._programs.clear()

    def _nhwc_of(self, t: torch.Tensor) -> torch.Tensor:
        """NHWC ``[H,W,C]`` storage of a ``[1,C,H,W]`` target.  The views ``set_targets`` publishes
        are recognised (channels-last strides, storage dtype): no copy.  Anything else is converted
        once and cached per (tensor, version) - this runs on every evaluation."""
        if t.shape[0] != 1:
            msg = f"content target must have batch size 1, got shape {tuple(t.shape)}"
            raise RuntimeError(msg)
        v = t[0].permute(1, 2, 0)
        if v.is_contiguous() and v.dtype == self.dtype and v.device == self.device:
            return v
        # The entry keeps the source tensor alive and is only reused for that very tensor object: a key of
        # (address, version) alone can be recycled by the caching allocator for a NEW target tensor once the
        # old one is freed, and the stale converted copy would be returned.
        cache = self.__dict__.setdefault("_target_cache", {})
        key = (t.data_ptr(), t._version, tuple(t.shape), t.dtype)
        hit = cache.get(key)
        if hit is None or hit[0] is not t:
            if len(cache) > 8:
                cache.clear()
            hit = (t, v.to(self.device, self.dtype).contiguous())
            cache[key] = hit
        return hit[1]

    def _as_target(self, t: torch.Tensor, shape: tuple, dtype: torch.dtype) -> torch.Tensor:
        if tuple(t.shape) != shape:
            msg = f"target shape {tuple(t.shape)} does not match features {shape}"
            raise RuntimeError(msg)
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous():
            t = t.to(self.device, dtype).contiguous()
        return t

    # -- execution -------------------------------------------------------------
    def loss_and_grad(self, x: torch.Tensor, grad: torch.Tensor, style_w: float, content_w: float, *,
                      score_log: tuple | None = None, then_step=None) -> None:
        """``then_step`` (an ``optimizers.StepRequest`` for ``x``): the L-BFGS update of ``x`` from ``grad`` is the
        schedule's last op - closure and update are one launch (one hipGraph)."""
        self.generation += 1
        key = ("fused", x.data_ptr(), grad.data_ptr(), style_w, content_w,
               None if score_log is None else (tuple(t.data_ptr() for t in score_log), tuple(score_log[0].shape)),
               None if then_step is None else then_step.key())

        def build():
            s = self.sched
            s.alloc_grads()          # the content term's gradient is written during the forward half
            fused_content = tuple(t for t in s.content_taps if self._content_fused(t))
            tail = []
            if then_step is not None:
                # m_max = history: the reduction's grid covers a full history from the first step on (blocks past the
                # live pairs return at once), so ONE captured graph serves every step
                op = s._op(op=_lib.OP_LBFGS_STEP, p0=grad, q0=x, q1=then_step.state, q2=then_step.work, n=x.numel(),
                           cin=then_step.history, cout=then_step.history, f0=then_step.lr, f1=then_step.tol_grad,
                           f2=then_step.tol_change)
                op.flags |= _lib.LANE_JOIN
                tail.append(op)
            return (self._forward_with_losses(x, style_coef=style_w, with_seed=True, content_coef=content_w)
                    + [self._combine_op(style_w, content_w, score_log)]
                    + s.backward_ops(grad, style_coef=style_w, content_coef=content_w, coef_dev=None,
                                     prewritten=fused_content)
                    + tail)
        self._program(key, build).run(self.use_graph)

    def forward_losses(self, x: torch.Tensor) -> None:
        self.generation += 1
        key = ("fwd", x.data_ptr())

        def build():
            return (self._forward_with_losses(x, style_coef=0.0, with_seed=False)
                    + [self._combine_op(1.0, 1.0)])
        self._program(key, build).run(self.use_graph)

    def backward_from(self, gout: torch.Tensor, grad: torch.Tensor) -> None:
        self.coef_buf.copy_(gout)
        key = ("bwd", grad.data_ptr())

        def build():
            s = self.sched
            seeds = []
            for tap in s.style_taps:   # recompute the seeds with the upstream coefficients
                seeds += s.gram_ops(tap, gram_out=None, target=tap.target, loss_part=None, sgrad=tap.sgrad,
                                    coef=1.0, coef_dev=self.coef_buf[tap.order:], partial=False)
            return seeds + s.backward_ops(grad, style_coef=1.0, content_coef=1.0, coef_dev=self.coef_buf)
        self._program(key, build).run(self.use_graph)


class _LossesFn(torch.autograd.Function):
    """``model(x)`` for autograd users: HIP forward now, HIP backward on ``.backward()``."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, engine: _Engine) -> torch.Tensor:
        engine.forward_losses(engine.stage_input(x))
        ctx.engine = engine
        ctx.generation = engine.generation
        ctx.x_meta = (x.shape, x.dtype)
        return engine.losses.clone()

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        engine: _Engine = ctx.engine
        if engine.generation != ctx.generation:
            # the engine keeps ONE set of activations per image size: a later model(x'),
            # loss_and_grad or set_targets has overwritten what this backward would differentiate
            msg = ("StyleContentModel: backward() of a forward pass whose activations were overwritten by a later "
                   "evaluation of the same model (call backward() before evaluating the model again).")
            raise RuntimeError(msg)
        shape, dtype = ctx.x_meta
        grad = torch.empty(shape, device=engine.device, dtype=torch.float32)
        engine.backward_from(gout.contiguous().float(), grad)
        return grad.to(dtype), None


# ---------------------------------------------------------------------------- model
class StyleContentModel(nn.Module):
    """VGG19 feature extractor with Gram style losses and content MSE losses.

    Same attributes as reference core_model.py:149-328 (``vgg_blocks``,
    ``style_ids``, ``content_ids``, ``style_targets``, ``content_targets``);
    ``forward`` returns two lists of 0-d loss tensors differentiable w.r.t. the
    input image.  ``loss_and_grad`` is the fused fast path used by the runner.
    """

    def __init__(self, style_layers: list[int], content_layers: list[int], *,
                 precision: str | None = None) -> None:
        super().__init__()
        vgg = initialize_vgg()
        self.vgg_blocks, self.content_ids, self.style_ids = create_feature_blocks(
            vgg, style_layers, content_layers)
        self.style_targets: list[torch.Tensor] | None = None
        self.content_targets: list[torch.Tensor] | None = None
        self._style_at = sorted(set(style_layers))
        self._content_at = sorted(set(content_layers))
        self._dtype = resolve_precision(precision)
        self._engines: dict = {}

    # -- engine plumbing ---------------------------------------------------------
    def _layers(self) -> list[nn.Module]:
        return [layer for block in self.vgg_blocks for layer in block]

    def _check_image(self, x: torch.Tensor) -> None:
        if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dim() != 4 or x.shape[0] != 1:
            msg = f"expected a GPU image of shape [1, C, H, W], got {tuple(x.shape) if isinstance(x, torch.Tensor) else type(x)}"
            raise RuntimeError(msg)
        # the condition torch's conv2d rejects in the reference; here the first-layer kernel takes
        # cin from the weights and would otherwise read out of bounds
        first = next((layer for layer in self._layers() if isinstance(layer, nn.Conv2d)), None)
        if first is not None and int(x.shape[1]) != first.in_channels:
            msg = (f"expected input with {first.in_channels} channels (the first convolution's in_channels), "
                   f"got {int(x.shape[1])} channels in shape {tuple(x.shape)}")
            raise RuntimeError(msg)

    def _engine_for(self, x: torch.Tensor) -> _Engine:
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            msg = ("StyleContentModel runs on the MI355X HIP kernels only: the image must be a GPU tensor "
                   "(there is no CPU fallback on this path).")
            raise RuntimeError(msg)
        if x.dim() != 4 or x.shape[0] != 1:
            msg = f"expected an image of shape [1, C, H, W], got {tuple(x.shape)}"
            raise RuntimeError(msg)
        if not self._style_at and not self._content_at:
            msg = "no style or content layers configured"
            raise RuntimeError(msg)
        self._check_image(x)
        H, W = int(x.shape[-2]), int(x.shape[-1])
        key = (H, W, x.device.index)
        eng = self._engines.get(key)
        if eng is None:
            eng = _Engine(self._layers(), self._style_at, self._content_at, H, W, self._dtype, x.device)
            self._engines[key] = eng
        return eng

    def set_targets(self, style_img: torch.Tensor, content_img: torch.Tensor) -> None:
        """Capture Gram targets from the style image and activations from the content image."""
        eng = self._engine_for(content_img)
        if not style_img.is_cuda:
            msg = "style image must be a GPU tensor (no CPU fallback on this path)"
            raise RuntimeError(msg)
        self._check_image(style_img)
        self.style_targets = eng.capture_style(style_img)
        # public form as in the reference (core_model.py:230-232): [1, C, H, W].  They are zero-copy
        # NCHW views of the engine's NHWC buffers (channels-last strides, storage dtype).
        self.content_targets = [t.permute(2, 0, 1).unsqueeze(0) for t in eng.capture_content(content_img)]

    def _require_targets(self) -> None:
        # same order and texts as reference core_model.py:255-258, 287-290
        if self.style_targets is None:
            msg = "style_targets must be set before computing losses."
            raise RuntimeError(msg)
        if self.content_targets is None:
            msg = "content_targets must be set before computing losses."
            raise RuntimeError(msg)

    def forward(self, x: torch.Tensor) -> tuple[list[torch.Tensor], list[torch.Tensor]]:
        """Per-layer style and content losses for ``x`` (``[1, C, H, W]``)."""
        self._require_targets()
        eng = self._engine_for(x)
        eng.bind_targets(self.style_targets, self.content_targets)
        losses = _LossesFn.apply(x, eng)
        style = [losses[i] for i in range(eng.n_style)]
        content = [losses[eng.n_style + i] for i in range(eng.n_content)]
        return style, content

    def loss_and_grad(self, x: torch.Tensor, style_w: float, content_w: float, *, live_scores: bool = False,
                      score_log: tuple | None = None,
                      ) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Fused step: writes d(style_w*S + content_w*C)/dx into ``x.grad``.

        Equivalent to reference optimization.py:292-313 (forward, weighted sum,
        ``loss.backward()``) in one command-buffer launch with no host sync.
        Returns 0-d device tensors (style_score, content_score, total).  With
        ``live_scores`` they are views of the engine's score buffer - valid until the
        next evaluation, one copy kernel less per step (the runner consumes them at once).
        ``score_log`` = (ring [3, capacity] fp32, counter [1] int32) on this device: the three scores are also
        appended to that history ring by the combine kernel itself (LossAccumulator.device_log()).
        """
        self._require_targets()
        eng = self._engine_for(x)
        eng.bind_targets(self.style_targets, self.content_targets)
        if x.dtype != torch.float32 or not x.is_contiguous():
            msg = "loss_and_grad needs a contiguous float32 image"
            raise RuntimeError(msg)
        grad = getattr(self, "_grad_buf", None)
        if grad is None or grad.shape != x.shape or grad.device != x.device:
            grad = torch.zeros_like(x, requires_grad=False)
            self._grad_buf = grad
        from . import optimizers  # noqa: PLC0415
        # inside HipLBFGS.step(closure) for this very tensor: the update rides at the end of the same launch
        eng.loss_and_grad(x.detach(), grad, float(style_w), float(content_w), score_log=score_log,
                          then_step=optimizers.claim_step(x))
        x.grad = grad
        scores = eng.scores if live_scores else eng.scores.clone()
        return scores[0], scores[1], scores[2]


def prepare_model_and_input(
    content_img: torch.Tensor,
    style_img: torch.Tensor,
    device: torch.device,
    optimization,
    *,
    precision: str | None = None,
) -> tuple[nn.Module, torch.Tensor, torch.optim.Optimizer]:
    """Build model, start image and optimizer (reference core_model.py:331-350).

    ``precision`` (keyword-only extension): "fp32" | "bf16" activation storage;
    default from ``STV_PRECISION`` or fp32.
    """
    from .optimizers import make_lbfgs  # noqa: PLC0415

    model = StyleContentModel(
        style_layers=optimization.style_layers,
        content_layers=optimization.content_layers,
        precision=precision,
    ).to(device)
    model.set_targets(style_img, content_img)
    input_img = initialize_input(content_img, optimization.init_method)
    optimizer = make_lbfgs(
        input_img,
        lr=optimization.lr,
        max_iter=optimization.lbfgs_max_iter,
        max_eval=optimization.lbfgs_max_eval,
    )
    return model, input_img, optimizer

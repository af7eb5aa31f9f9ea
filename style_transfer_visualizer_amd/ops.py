"""Tensor-level wrappers over the C ABI (``include/stv.h``).

PyTorch is used only to own device memory and the current HIP stream; every
function here enqueues hand-written HIP kernels from ``libstv_hip.so`` and
raises ``RuntimeError`` if the library is missing or a call fails.

Layouts: activations are NHWC ``[H, W, C]`` tensors (batch 1) of dtype
float32 or bfloat16; the image / its gradient are ``[1, 3, H, W]`` float32.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import ACCUM, MASK, POOL_ONLY, RELU_IN, RELU_OUT, STV_BF16, STV_F32, W_BLOCKED  # noqa: F401


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return STV_F32
    if dtype == torch.bfloat16:
        return STV_BF16
    msg = f"unsupported activation dtype {dtype}; use torch.float32 or torch.bfloat16"
    raise RuntimeError(msg)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        msg = "libstv_hip operates on device memory only (tensor is on CPU); there is no CPU fallback"
        raise RuntimeError(msg)
    if not t.is_contiguous():
        msg = "libstv_hip expects contiguous tensors"
        raise RuntimeError(msg)
    return t.data_ptr()


# ---- weight packing (once, at model build; not on the per-step path) ---------

def pack_weights_fwd(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [9,Cout,Cin], tap = ky*3+kx (K-contiguous rows)."""
    cout, cin = w.shape[:2]
    return w.permute(2, 3, 0, 1).reshape(9, cout, cin).contiguous()


def pack_weights_bwd(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [9,Cin,Cout] with flipped taps: dgrad as a forward conv."""
    cout, cin = w.shape[:2]
    return w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, cin, cout).contiguous()


def block_weights(w: torch.Tensor) -> torch.Tensor:
    """[taps,Cout,Cin] (already in the compute dtype) -> K-blocked [taps,Cin/CK,Cout,CK], CK = 32 bytes.

    The layout ``STV_W_BLOCKED`` names (include/stv.h): the 32-byte K slices that one workgroup
    stages for its output channels become contiguous in memory.
    """
    taps, cout, cin = w.shape
    ck = 32 // w.element_size()
    if cin % ck:
        msg = f"cin={cin} is not a multiple of {ck}: this layer cannot use the K-blocked layout"
        raise RuntimeError(msg)
    return w.reshape(taps, cout, cin // ck, ck).permute(0, 2, 1, 3).contiguous()


def conv_uses_mfma(H: int, W: int, cin: int, cout: int, dtype: torch.dtype) -> bool:
    """True when stv_conv_igemm runs this shape on the matrix cores (else: direct kernel, plain weights)."""
    return int(_lib.load().stv_conv_config(H, W, cin, cout, 9, dtype_code(dtype))) >= 0


def conv_uses_ws(H: int, W: int, cin: int, cout: int, dtype: torch.dtype, *, flags: int = 0, has_ref: bool = False,
                 has_pool: bool = False) -> bool:
    """True when this 3x3 launch runs on the weight-stationary persistent kernel (csrc/conv_ws.hip)."""
    return bool(_lib.load().stv_conv_uses_ws(H, W, cin, cout, 9, dtype_code(dtype), flags, int(has_ref), int(has_pool)))


TUNE_ROUTE = 109      # stv.h STV_TUNE_ROUTE: `taps` value that tunes the shape as conv_igemm_route runs it


def conv_tune(H: int, W: int, cin: int, cout: int, taps: int, dtype: torch.dtype) -> int:
    """Measure the tile configurations for one conv shape once (stv_conv_tune); returns the choice (-1: direct kernel).
    ``taps=TUNE_ROUTE``: the dgrad with the pooling backward in its epilogue (its own table entry)."""
    r = int(_lib.load().stv_conv_tune(H, W, cin, cout, taps, dtype_code(dtype), _stream()))
    if r < -1:
        _lib.check(-r - 100, "stv_conv_tune")
    return r


def to_nhwc(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[1,C,H,W] -> [H,W,C] (test/fixture helper; the product never converts activations)."""
    return x[0].permute(1, 2, 0).contiguous().to(dtype)


def from_nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(2, 0, 1).unsqueeze(0).float().contiguous()


# ---- conv ---------------------------------------------------------------------

def conv_workspace_bytes() -> int:
    return int(_lib.load().stv_conv_workspace_bytes())


def set_conv_workspace(ws: torch.Tensor | None) -> None:
    """Scratch for convolutions that split K across workgroups (``stv_conv_workspace``): a ZEROED uint8 device tensor of
    ``conv_workspace_bytes()`` bytes the caller keeps alive, for the conv launches of this host thread; ``None`` clears it."""
    lib = _lib.load()
    if ws is None:
        lib.stv_conv_workspace(None, 0)
    else:
        lib.stv_conv_workspace(_ptr(ws), ws.numel() * ws.element_size())


def conv_first_pack(wf: torch.Tensor) -> torch.Tensor:
    """Kernel-side packing of a frozen first-layer weight [9,Cout,Cin] (fp32), done once (stv_conv_first_pack)."""
    _, cout, cin = wf.shape
    lib = _lib.load()
    packed = torch.empty(int(lib.stv_conv_first_packed_bytes(cin, cout)) // 4, device=wf.device, dtype=torch.float32)
    _lib.check(lib.stv_conv_first_pack(_ptr(wf), _ptr(packed), cin, cout, _stream()), "stv_conv_first_pack")
    return packed


def conv_first_gram_supported(H: int, W: int, cin: int, cout: int, dtype: torch.dtype) -> bool:
    """True when the first layer can leave the Gram slabs of its own output (stv_conv_first_fwd_gram)."""
    return bool(_lib.load().stv_conv_first_gram_supported(H, W, cin, cout, dtype_code(dtype)))


def conv_first_fwd(x_nchw: torch.Tensor, wf: torch.Tensor, bias: torch.Tensor | None,
                   dtype: torch.dtype, out: torch.Tensor | None = None, *,
                   packed: torch.Tensor | None = None, gram_partials: torch.Tensor | None = None) -> torch.Tensor:
    """``packed`` (from :func:`conv_first_pack`) skips the per-call weight repack.  ``gram_partials``
    ([gram_ksplit(H*W, cout), cout, cout] fp32; needs ``packed``): also filled, as :func:`gram_partial` of the result would."""
    _, cin, H, W = x_nchw.shape
    cout = wf.shape[1]
    if packed is None and gram_partials is None and (cin, cout) == (3, 64):
        packed = conv_first_pack(wf)      # the 3 -> 64 kernels read the kernel-side order (a torch-owned buffer, this stream)
    if out is None:
        out = torch.empty(H, W, cout, device=x_nchw.device, dtype=dtype)
    lib = _lib.load()
    if gram_partials is not None:
        if packed is None or tuple(gram_partials.shape) != (gram_ksplit(H * W, cout), cout, cout):
            raise ValueError("gram_partials needs packed weights and [gram_ksplit(H*W, cout), cout, cout] slabs")
        _lib.check(lib.stv_conv_first_fwd_gram(_ptr(x_nchw), _ptr(packed), _ptr(bias), _ptr(out), _ptr(gram_partials),
                                               H, W, cin, cout, dtype_code(dtype), _stream()), "stv_conv_first_fwd_gram")
    elif packed is not None:
        _lib.check(lib.stv_conv_first_fwd_packed(_ptr(x_nchw), _ptr(packed), _ptr(bias), _ptr(out), H, W, cin, cout,
                                                 dtype_code(dtype), _stream()), "stv_conv_first_fwd_packed")
    else:
        _lib.check(lib.stv_conv_first_fwd(_ptr(x_nchw), _ptr(wf), _ptr(bias), _ptr(out), H, W, cin, cout,
                                          dtype_code(dtype), _stream()), "stv_conv_first_fwd")
    return out


def conv_first_dgrad(dy: torch.Tensor, wf: torch.Tensor, cin: int,
                     out: torch.Tensor | None = None, *, packed: torch.Tensor | None = None) -> torch.Tensor:
    H, W, cout = dy.shape
    if out is None:
        out = torch.empty(1, cin, H, W, device=dy.device, dtype=torch.float32)
    lib = _lib.load()
    if packed is None and (cin, cout) == (3, 64):
        packed = conv_first_pack(wf)
    if packed is not None:
        _lib.check(lib.stv_conv_first_dgrad_packed(_ptr(dy), _ptr(packed), _ptr(out), H, W, cin, cout,
                                                   dtype_code(dy.dtype), _stream()), "stv_conv_first_dgrad_packed")
    else:
        _lib.check(lib.stv_conv_first_dgrad(_ptr(dy), _ptr(wf), _ptr(out), H, W, cin, cout,
                                            dtype_code(dy.dtype), _stream()), "stv_conv_first_dgrad")
    return out


# ---- conv ---------------------------------------------------------------------

def conv_igemm(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None,
               ref: torch.Tensor | None = None, out: torch.Tensor | None = None,
               flags: int = 0) -> torch.Tensor:
    """x [H,W,Cin], w [taps,Cout,Cin] or K-blocked [taps,Cin/CK,Cout,CK] (same dtype) -> [H,W,Cout]."""
    H, W, cin = x.shape
    if w.dim() == 4:
        taps, nck, cout, ck = w.shape
        cin_w = nck * ck
        flags |= W_BLOCKED
    else:
        taps, cout, cin_w = w.shape
    if cin_w != cin or w.dtype != x.dtype:
        msg = f"weight {tuple(w.shape)}/{w.dtype} does not match input {tuple(x.shape)}/{x.dtype}"
        raise RuntimeError(msg)
    if out is None:
        if flags & ACCUM:
            msg = "ACCUM needs an existing output tensor"
            raise RuntimeError(msg)
        out = torch.empty(H, W, cout, device=x.device, dtype=x.dtype)
    lib = _lib.load()
    _lib.check(lib.stv_conv_igemm(_ptr(x), _ptr(w), _ptr(bias), _ptr(ref), _ptr(out), H, W, cin, cout,
                                  taps, flags, dtype_code(x.dtype), _stream()), "stv_conv_igemm")
    return out


def conv_igemm_pool(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, flags: int = 0,
                    out: torch.Tensor | None = None, pool_out: torch.Tensor | None = None,
                    pool_idx: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """3x3 conv (+bias, +ReLU by flag) and MaxPool2d(2,2) of its output in one launch (stv_conv_igemm_pool).
    ``pool_idx`` (uint8 [H/2][W/2][cout], optional) receives the arg-max map ``maxpool_bwd_idx`` consumes."""
    H, W, cin = x.shape
    if w.dim() == 4:
        taps, nck, cout, ck = w.shape
        flags |= W_BLOCKED
    else:
        taps, cout, _ = w.shape
    if taps != 9:
        msg = "the fused pool needs a 3x3 convolution"
        raise RuntimeError(msg)
    if out is None:
        out = torch.empty(H, W, cout, device=x.device, dtype=x.dtype)
    if pool_out is None:
        pool_out = torch.empty(H // 2, W // 2, cout, device=x.device, dtype=x.dtype)
    lib = _lib.load()
    _lib.check(lib.stv_conv_igemm_pool(_ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(pool_out), _ptr(pool_idx), H, W,
                                       cin, cout, flags, dtype_code(x.dtype), _stream()), "stv_conv_igemm_pool")
    return out, pool_out


def conv_igemm_dual(x: torch.Tensor, w: torch.Tensor, x2: torch.Tensor, w2: torch.Tensor,
                    ref: torch.Tensor | None = None, out: torch.Tensor | None = None, flags: int = 0) -> torch.Tensor:
    """out = [out +] mask(ref>0) * conv3x3(x, w) + x2 . w2^T in one launch (stv_conv_igemm_dual)."""
    H, W, cin = x.shape
    if w.dim() == 4:
        _, nck, cout, ck = w.shape
        flags |= W_BLOCKED
    else:
        _, cout, _ = w.shape
    cin2 = x2.shape[2]
    if out is None:
        if flags & ACCUM:
            msg = "ACCUM needs an existing output tensor"
            raise RuntimeError(msg)
        out = torch.empty(H, W, cout, device=x.device, dtype=x.dtype)
    lib = _lib.load()
    _lib.check(lib.stv_conv_igemm_dual(_ptr(x), _ptr(w), _ptr(x2), _ptr(w2), _ptr(ref), _ptr(out), H, W, cin, cin2,
                                       cout, flags, dtype_code(x.dtype), _stream()), "stv_conv_igemm_dual")
    return out


def conv_igemm_route(dy: torch.Tensor, w: torch.Tensor, pool_idx: torch.Tensor, out: torch.Tensor | None = None,
                     flags: int = 0) -> torch.Tensor:
    """dgrad of the conv behind a max-pool with the pooling backward in its epilogue (stv_conv_igemm_route):
    dy [H,W,Cin], w backward-packed, pool_idx [H,W,Cout] uint8 -> routed gradient [2H,2W,Cout]."""
    H, W, cin = dy.shape
    if w.dim() == 4:
        _, nck, cout, ck = w.shape
        flags |= W_BLOCKED
    else:
        _, cout, _ = w.shape
    if out is None:
        out = torch.empty(2 * H, 2 * W, cout, device=dy.device, dtype=dy.dtype)
    lib = _lib.load()
    _lib.check(lib.stv_conv_igemm_route(_ptr(dy), _ptr(w), _ptr(pool_idx), _ptr(out), H, W, cin, cout, flags,
                                        dtype_code(dy.dtype), _stream()), "stv_conv_igemm_route")
    return out


def gram_multi(feats: list[torch.Tensor], targets: list[torch.Tensor], *, coef: float = 1.0,
               clamp_max: float = 5e5) -> tuple[list[torch.Tensor], list[torch.Tensor], list[torch.Tensor]]:
    """Batched Gram chain (stv_gram_multi) over NHWC feature maps: returns (grams, loss partials, seeds) per tap."""
    lib = _lib.load()
    table = (_lib.StvGramTap * len(feats))()
    keep, grams, parts, seeds = [], [], [], []
    for e, f, t in zip(table, feats, targets, strict=True):
        H, W, C = f.shape
        n = H * W
        partials = torch.empty(gram_ksplit(n, C), C, C, device=f.device, dtype=torch.float32)
        g = torch.empty(C, C, device=f.device, dtype=torch.float32)
        lp = torch.empty(gram_loss_parts(C), device=f.device, dtype=torch.float32)
        sg = torch.empty(C, C, device=f.device, dtype=f.dtype)
        keep += [partials]
        grams.append(g); parts.append(lp); seeds.append(sg)
        e.F, e.partials, e.target, e.gram_out, e.loss_part, e.sgrad = (_ptr(f), _ptr(partials), _ptr(t), _ptr(g), _ptr(lp),
                                                                      _ptr(sg))
        e.coef_dev = None
        e.n_pixels, e.channels = n, C
        e.clamp_max, e.norm, e.coef = clamp_max, float(C * n), coef
    _lib.check(lib.stv_gram_multi(ctypes.addressof(table), len(feats), dtype_code(feats[0].dtype), _stream()),
               "stv_gram_multi")
    torch.cuda.current_stream().synchronize()       # `partials` may go out of scope now
    return grams, parts, seeds


# ---- pool / relu --------------------------------------------------------------

def maxpool_fwd(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    H, W, C = x.shape
    if out is None:
        out = torch.empty(H // 2, W // 2, C, device=x.device, dtype=x.dtype)
    lib = _lib.load()
    _lib.check(lib.stv_maxpool_fwd(_ptr(x), _ptr(out), H, W, C, dtype_code(x.dtype), _stream()),
               "stv_maxpool_fwd")
    return out


def maxpool_bwd_idx(idx: torch.Tensor, dy: torch.Tensor, H: int, W: int, out: torch.Tensor | None = None,
                    flags: int = 0) -> torch.Tensor:
    """Pooling backward from the arg-max byte map of ``conv_igemm_pool`` (H, W: the un-pooled size)."""
    C = dy.shape[-1]
    if out is None:
        out = torch.empty(H, W, C, device=dy.device, dtype=dy.dtype)
    lib = _lib.load()
    _lib.check(lib.stv_maxpool_bwd(_ptr(idx), _ptr(dy), _ptr(out), H, W, C, flags | _lib.POOL_IDX,
                                   dtype_code(dy.dtype), _stream()), "stv_maxpool_bwd")
    return out


def maxpool_bwd(x: torch.Tensor, dy: torch.Tensor, out: torch.Tensor | None = None,
                flags: int = 0) -> torch.Tensor:
    H, W, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.stv_maxpool_bwd(_ptr(x), _ptr(dy), _ptr(out), H, W, C, flags, dtype_code(x.dtype),
                                   _stream()), "stv_maxpool_bwd")
    return out


def relu_fwd(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.stv_relu_fwd(_ptr(x), _ptr(out), x.numel(), dtype_code(x.dtype), _stream()), "stv_relu_fwd")
    return out


def relu_bwd(x: torch.Tensor, dy: torch.Tensor, out: torch.Tensor | None = None, flags: int = 0) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    _lib.check(lib.stv_relu_bwd(_ptr(x), _ptr(dy), _ptr(out), x.numel(), flags, dtype_code(x.dtype), _stream()),
               "stv_relu_bwd")
    return out


# ---- gram / content -------------------------------------------------------------

def gram_ksplit(n_pixels: int, C: int) -> int:
    return _lib.load().stv_gram_ksplit(n_pixels, C)


def gram_loss_parts(C: int) -> int:
    return _lib.load().stv_gram_loss_parts(C)


def gram_partial(F: torch.Tensor, partials: torch.Tensor | None = None) -> torch.Tensor:
    """F [H,W,C] (or [N,C]) -> fp32 partial slabs [ksplit,C,C]."""
    C = F.shape[-1]
    n = F.numel() // C
    if partials is None:
        partials = torch.empty(gram_ksplit(n, C), C, C, device=F.device, dtype=torch.float32)
    lib = _lib.load()
    _lib.check(lib.stv_gram_partial(_ptr(F), _ptr(partials), n, C, dtype_code(F.dtype), _stream()),
               "stv_gram_partial")
    return partials


def gram_finish(partials: torch.Tensor, n_pixels: int, C: int, *, target: torch.Tensor | None = None,
                gram_out: torch.Tensor | None = None, loss_part: torch.Tensor | None = None,
                sgrad: torch.Tensor | None = None, clamp_max: float = 5e5, coef: float = 1.0,
                coef_dev: torch.Tensor | None = None, dtype: torch.dtype = torch.float32,
                norm: float | None = None) -> None:
    """``norm`` defaults to ``C * n_pixels`` (the reference's ``b*c*h*w``); callers that padded the
    channel axis pass the un-padded product."""
    lib = _lib.load()
    _lib.check(lib.stv_gram_finish(_ptr(partials), _ptr(target), _ptr(gram_out), _ptr(loss_part), _ptr(sgrad),
                                   n_pixels, C, clamp_max, float(C * n_pixels) if norm is None else float(norm),
                                   coef, _ptr(coef_dev),
                                   dtype_code(dtype), _stream()), "stv_gram_finish")


def content_loss(F: torch.Tensor, target: torch.Tensor, loss_part: torch.Tensor) -> None:
    lib = _lib.load()
    _lib.check(lib.stv_content_loss(_ptr(F), _ptr(target), _ptr(loss_part), F.numel(), dtype_code(F.dtype),
                                    _stream()), "stv_content_loss")


def content_loss_grad(F: torch.Tensor, target: torch.Tensor, loss_part: torch.Tensor, dF: torch.Tensor, coef: float) -> None:
    """Loss partials (as :func:`content_loss`) and ``dF = coef * 2/n * (F - target)`` (written) in one pass."""
    lib = _lib.load()
    _lib.check(lib.stv_content_loss_grad(_ptr(F), _ptr(target), _ptr(loss_part), _ptr(dF), F.numel(), coef,
                                         dtype_code(F.dtype), _stream()), "stv_content_loss_grad")


def content_grad(F: torch.Tensor, target: torch.Tensor, dF: torch.Tensor, coef: float,
                 coef_dev: torch.Tensor | None = None, flags: int = 0) -> None:
    lib = _lib.load()
    _lib.check(lib.stv_content_grad(_ptr(F), _ptr(target), _ptr(dF), F.numel(), coef, _ptr(coef_dev), flags,
                                    dtype_code(F.dtype), _stream()), "stv_content_grad")


class HostMailbox:
    """Pinned host memory the GPU writes while the CPU reads (``stv_host_mailbox_alloc``: mapped under the same
    pointer on the device, fine-grained coherent, zeroed).  ``tensor`` / ``array`` are views; every view keeps this
    object - and with it the allocation - alive."""

    def __init__(self, nbytes: int) -> None:
        out = ctypes.c_void_p()
        _lib.check(_lib.load().stv_host_mailbox_alloc(int(nbytes), ctypes.byref(out)), "stv_host_mailbox_alloc")
        self.ptr, self.nbytes = int(out.value), int(nbytes)
        self._buf = (ctypes.c_char * self.nbytes).from_address(self.ptr)

    def tensor(self, dtype: torch.dtype, shape: tuple, offset: int = 0) -> torch.Tensor:
        count = 1
        for d in shape:
            count *= int(d)
        t = torch.frombuffer(self._buf, dtype=dtype, count=count, offset=offset).view(*shape)
        t._stv_owner = self          # (a program that baked this pointer into a graph keeps the tensor, hence the memory)
        return t

    def array(self, dtype, count: int, offset: int = 0):
        import numpy as np  # noqa: PLC0415
        a = np.frombuffer(self._buf, dtype=dtype, count=count, offset=offset)
        return a

    def __del__(self) -> None:
        try:
            if self.ptr:
                _lib.load().stv_host_mailbox_free(self.ptr)
                self.ptr = 0
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


def loss_combine(parts: torch.Tensor, table: torch.Tensor, scale: torch.Tensor, style_w: float,
                 content_w: float, losses: torch.Tensor, scores: torch.Tensor) -> None:
    lib = _lib.load()
    _lib.check(lib.stv_loss_combine(_ptr(parts), _ptr(table), _ptr(scale), table.shape[0], style_w, content_w,
                                    _ptr(losses), _ptr(scores), _stream()), "stv_loss_combine")


# ---- frame / PNG export ---------------------------------------------------------

def image_to_u8(x: torch.Tensor, *, mean: tuple | list | None, std: tuple | list | None, rounding: bool,
                out: torch.Tensor | None = None) -> torch.Tensor:
    """[1,3,H,W] (or [3,H,W]) fp32 GPU image -> [H,W,3] uint8 GPU tensor (stv_image_to_u8).

    ``mean``/``std`` given: the image is ImageNet-normalised and is denormalised first; ``rounding``
    False = truncating ``*255`` (frames), True = ``+0.5`` rounding (final PNG)."""
    img = x.detach()
    if img.dim() == 4:
        if img.shape[0] != 1:
            msg = f"expected one image, got a batch of {img.shape[0]}"
            raise RuntimeError(msg)
        img = img[0]
    if img.dim() != 3 or img.shape[0] != 3 or img.dtype != torch.float32:
        msg = f"expected a float32 [3,H,W] image, got {tuple(img.shape)} {img.dtype}"
        raise RuntimeError(msg)
    img = img.contiguous()
    _, H, W = img.shape
    if out is None:
        out = torch.empty(H, W, 3, device=img.device, dtype=torch.uint8)
    m3 = (ctypes.c_float * 3)(*mean) if mean is not None else None
    s3 = (ctypes.c_float * 3)(*std) if std is not None else None
    lib = _lib.load()
    _lib.check(lib.stv_image_to_u8(_ptr(img), _ptr(out), H, W, m3, s3, 1 if rounding else 0, _stream()),
               "stv_image_to_u8")
    return out


# ---- optimizers ---------------------------------------------------------------

def lbfgs_alloc(n: int, history: int, device: torch.device, *, compact: bool = False,
                ) -> tuple[torch.Tensor, torch.Tensor]:
    """Zeroed device state block + workspace for ``lbfgs_step`` (``compact`` selects the layout)."""
    lib = _lib.load()
    if compact:
        st_bytes, ws_bytes = lib.stv_lbfgsc_state_bytes(history), lib.stv_lbfgsc_workspace_bytes(n, history)
    else:
        st_bytes, ws_bytes = lib.stv_lbfgs_state_bytes(history), lib.stv_lbfgs_workspace_bytes(n, history)
    state = torch.zeros(st_bytes, dtype=torch.uint8, device=device)
    work = torch.zeros((ws_bytes + 3) // 4, dtype=torch.float32, device=device)
    return state, work


def lbfgs_step(x: torch.Tensor, grad: torch.Tensor, state: torch.Tensor, work: torch.Tensor, history: int,
               m_max: int, lr: float, tol_grad: float = 1e-7, tol_change: float = 1e-9, *,
               compact: bool = False) -> None:
    lib = _lib.load()
    fn = lib.stv_lbfgsc_step if compact else lib.stv_lbfgs_step
    _lib.check(fn(_ptr(x), _ptr(grad), _ptr(state), _ptr(work), x.numel(), history, m_max, lr,
                  tol_grad, tol_change, _stream()), "stv_lbfgsc_step" if compact else "stv_lbfgs_step")


def lbfgs_dots(grad: torch.Tensor, state: torch.Tensor, work: torch.Tensor, history: int, m_max: int) -> torch.Tensor:
    """First half of a compact L-BFGS step (stv_lbfgsc_dots): this shard's partial inner products.
    Returns a float64 VIEW into ``work`` (5*128 + 8 entries) for the caller to all-reduce."""
    lib = _lib.load()
    n = grad.numel()
    _lib.check(lib.stv_lbfgsc_dots(_ptr(grad), _ptr(state), _ptr(work), n, history, m_max, _stream()), "stv_lbfgsc_dots")
    return lbfgs_dots_view(work, n, history)[0]


def lbfgs_dots_view(work: torch.Tensor, n: int, history: int) -> tuple[torch.Tensor, int]:
    """(float64 view of the step's inner products inside ``work``, index of the max|g| entry)."""
    count, imax = ctypes.c_int(), ctypes.c_int()
    off = int(_lib.load().stv_lbfgsc_dots_offset(n, history, ctypes.byref(count), ctypes.byref(imax)))
    if off % 8:
        msg = "internal: unaligned inner-product block"
        raise RuntimeError(msg)
    return work[off // 4: off // 4 + 2 * count.value].view(torch.float64), imax.value


def lbfgs_apply(x: torch.Tensor, grad: torch.Tensor, state: torch.Tensor, work: torch.Tensor, history: int, lr: float,
                tol_grad: float = 1e-7, tol_change: float = 1e-9) -> None:
    """Second half (stv_lbfgsc_apply): scalar recursion from the (all-reduced) inner products, then the update."""
    lib = _lib.load()
    _lib.check(lib.stv_lbfgsc_apply(_ptr(x), _ptr(grad), _ptr(state), _ptr(work), x.numel(), history, lr, tol_grad,
                                    tol_change, _stream()), "stv_lbfgsc_apply")


def adam_step(x: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int,
              lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8) -> None:
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2_sqrt = (1 - b2 ** step) ** 0.5
    lib = _lib.load()
    _lib.check(lib.stv_adam_step(_ptr(x), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), x.numel(), lr, 1 - b1, b2,
                                 1 - b2, eps, bc1, bc2_sqrt, _stream()), "stv_adam_step")

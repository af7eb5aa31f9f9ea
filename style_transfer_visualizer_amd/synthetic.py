"""Deterministic synthetic inputs: counter-hash PRNG, images and VGG19 weights.

There is no network in the build/bench environment, so the pretrained
``vgg19-dcbb9e9d.pth`` the reference downloads in ``initialize_vgg``
(/root/reference/src/style_transfer_visualizer/core_model.py:103-117) is not
available.  Parity and benchmarks therefore use *identical synthetic weights*
on both sides: He-scaled uniform weights and zero bias
generated from a counter-based hash, so no torch RNG state is involved and the
52 MB of weights never need to be committed.

Everything here is plain numpy on the host; it is input generation, not part
of the per-step path.
"""
from __future__ import annotations

import numpy as np
import torch

# torchvision's VGG19 "E" configuration (features only).  Numbers are conv
# output channels, "M" is MaxPool2d(2, 2).  Module indices in
# ``vgg19().features`` follow from this list: conv, relu, [conv, relu,] pool ...
VGG19_CFG: tuple[int | str, ...] = (
    64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M",
    512, 512, 512, 512, "M", 512, 512, 512, 512, "M",
)

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_GOLD = np.uint32(0x9E3779B9)


def _mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32 integer finaliser (bijective on uint32)."""
    x = x.astype(np.uint32, copy=True)
    x ^= x >> np.uint32(16)
    x *= _M1
    x ^= x >> np.uint32(15)
    x *= _M2
    x ^= x >> np.uint32(16)
    return x


def hash_uniform(seed: int, stream: int, count: int) -> np.ndarray:
    """Return ``count`` float32 values in [0, 1) for (seed, stream).

    value[i] = (mix32(i ^ mix32(stream*GOLD + mix32(seed))) >> 8) * 2**-24
    """
    with np.errstate(over="ignore"):
        key = _mix32(np.array([seed], dtype=np.uint32))
        key = _mix32(np.uint32(stream) * _GOLD + key)
        idx = np.arange(count, dtype=np.uint32)
        h = _mix32(idx ^ key)
    return (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def synthetic_image(
    seed: int,
    height: int,
    width: int,
    *,
    normalize: bool = True,
) -> torch.Tensor:
    """U[0,1) RGB image ``[1,3,H,W]`` fp32, optionally ImageNet-normalised.

    Normalisation restates ``transforms.Normalize(IMAGENET_MEAN, IMAGENET_STD)``
    (/root/reference/src/style_transfer_visualizer/image_io.py:72-84,
    constants.py:11-12).
    """
    vals = hash_uniform(seed, 0x1A6E, 3 * height * width)
    img = torch.from_numpy(vals.reshape(1, 3, height, width).copy())
    if normalize:
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        img = (img - mean) / std
    return img


def vgg_conv_shapes(cfg: tuple[int | str, ...] = VGG19_CFG) -> list[tuple[int, int]]:
    """(Cin, Cout) of every conv in ``cfg`` order."""
    shapes = []
    cin = 3
    for v in cfg:
        if v == "M":
            continue
        shapes.append((cin, int(v)))
        cin = int(v)
    return shapes


def synthetic_conv_weights(
    seed: int = 0,
    cfg: tuple[int | str, ...] = VGG19_CFG,
    *,
    gain: float = 1.0,
) -> list[tuple[torch.Tensor, torch.Tensor]]:
    """He-scaled uniform weights ``[Cout,Cin,3,3]`` and zero bias per conv.

    Uniform on [-a, a] with a = gain*sqrt(6/(9*Cin)) has std sqrt(2/(9*Cin)).
    """
    out = []
    for li, (cin, cout) in enumerate(vgg_conv_shapes(cfg)):
        n = cout * cin * 9
        u = hash_uniform(seed, 0xC0 + li, n)
        a = np.float32(gain * np.sqrt(6.0 / (9.0 * cin)))
        w = ((u * np.float32(2.0) - np.float32(1.0)) * a).astype(np.float32)
        weight = torch.from_numpy(w.reshape(cout, cin, 3, 3).copy())
        bias = torch.zeros(cout, dtype=torch.float32)
        out.append((weight, bias))
    return out


def synthetic_bias(seed: int, layer: int, cout: int, scale: float = 0.05) -> torch.Tensor:
    """Small non-zero bias for tests that must exercise the bias path."""
    u = hash_uniform(seed, 0xB1A5 + layer, cout)
    return torch.from_numpy(((u * 2.0 - 1.0) * scale).astype(np.float32))

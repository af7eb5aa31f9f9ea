"""Image loading / saving at the edges of the hot path, torchvision-free.

Restates reference image_io.py:24-152 and the ``torchvision.utils.save_image``
call of runtime/output.py:101: PIL RGB -> float32 ``[1,3,H,W]`` in [0,1]
(``ToTensor``), optional ImageNet ``Normalize``, and the inverse for output.
These run once per run / per saved frame, not per step.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch
from PIL import Image

from .constants import (COLOR_MODE_RGB, DENORM_VIEW_SHAPE, IMAGENET_MEAN, IMAGENET_STD, MAX_DIMENSION,
                        MIN_DIMENSION)
from .logging_utils import logger


def load_image(path: str) -> Image.Image:
    """Open ``path`` as RGB; error texts as reference image_io.py:38-45."""
    try:
        return Image.open(path).convert(COLOR_MODE_RGB)
    except FileNotFoundError as e:
        msg = f"Image file not found: '{path}'"
        raise FileNotFoundError(msg) from e
    except OSError as e:
        msg = f"Error loading image '{path}': {e!s}"
        raise OSError(msg) from e


def validate_image_dimensions(img: Image.Image) -> None:
    """< MIN_DIMENSION raises, > MAX_DIMENSION only warns."""
    if img.width < MIN_DIMENSION or img.height < MIN_DIMENSION:
        msg = (f"Image too small: {img.width}x{img.height}. "
               f"Minimum dimension is {MIN_DIMENSION}px.")
        raise ValueError(msg)
    if img.width > MAX_DIMENSION or img.height > MAX_DIMENSION:
        logger.warning("Image is large: %dx%d. This may slow processing.", img.width, img.height)


def apply_transforms(img: Image.Image, device: torch.device, *, normalize: bool) -> torch.Tensor:
    """``ToTensor`` (+ ``Normalize``) -> ``[1,3,H,W]`` float32 on ``device``."""
    arr = np.asarray(img, dtype=np.uint8)
    tensor = torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    if normalize:
        mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(3, 1, 1)
        std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(3, 1, 1)
        tensor = (tensor - mean) / std
    return tensor.unsqueeze(0).to(device)


def load_image_to_tensor(path: str, device: torch.device, *, normalize: bool = False) -> torch.Tensor:
    """Load without resizing, validate size, convert."""
    img = load_image(path)
    validate_image_dimensions(img)
    return apply_transforms(img, device, normalize=normalize)


def denormalize(tensor: torch.Tensor) -> torch.Tensor:
    """Undo the ImageNet normalisation."""
    mean = torch.tensor(IMAGENET_MEAN).view(*DENORM_VIEW_SHAPE).to(tensor.device)
    std = torch.tensor(IMAGENET_STD).view(*DENORM_VIEW_SHAPE).to(tensor.device)
    return tensor * std + mean


def prepare_image_for_output(tensor: torch.Tensor, *, normalize: bool) -> torch.Tensor:
    """Denormalise if needed, replace NaN/inf, clamp to [0, 1]."""
    img = denormalize(tensor) if normalize else tensor
    img = torch.nan_to_num(img, nan=0.0, posinf=1.0, neginf=0.0)
    return img.clamp(0, 1)


def _on_gpu_path(tensor: torch.Tensor) -> bool:
    t = tensor
    return (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32
            and t.dim() in (3, 4) and t.shape[-3] == 3 and (t.dim() == 3 or t.shape[0] == 1))


def frame_uint8(tensor: torch.Tensor, *, normalize: bool) -> np.ndarray | None:
    """``[H,W,3]`` uint8 frame of the image being optimised (reference optimization.py:438-452):
    ``prepare_image_for_output`` then the TRUNCATING ``(x * 255).astype("uint8")``.

    A GPU image is converted on the device by ``stv_image_to_u8`` (one kernel, H*W*3 bytes to the
    host); a host tensor - the runner also drives CPU stand-in models in tests - takes the torch ops
    the reference uses."""
    if _on_gpu_path(tensor):
        from . import ops  # noqa: PLC0415
        u8 = ops.image_to_u8(tensor, mean=IMAGENET_MEAN if normalize else None,
                             std=IMAGENET_STD if normalize else None, rounding=False)
        return u8.cpu().numpy()
    image = prepare_image_for_output(tensor, normalize=normalize)
    if image is None:
        return None
    return (image.squeeze(0).permute(1, 2, 0).cpu().numpy() * 255).astype("uint8")


def save_image(tensor: torch.Tensor, path: str | Path, *, normalize: bool | None = None) -> None:
    """``torchvision.utils.save_image`` for one image: ``mul(255).add(0.5).clamp(0,255)`` -> uint8 PNG.

    ``normalize`` given: ``tensor`` is the raw optimised image and ``prepare_image_for_output`` is
    applied first (reference runtime/output.py:92-101) - on the device for a GPU tensor."""
    if normalize is not None and _on_gpu_path(tensor):
        from . import ops  # noqa: PLC0415
        arr = ops.image_to_u8(tensor, mean=IMAGENET_MEAN if normalize else None,
                              std=IMAGENET_STD if normalize else None, rounding=True).cpu().numpy()
        Image.fromarray(arr).save(str(path))
        return
    img = tensor.detach()
    if normalize is not None:
        img = prepare_image_for_output(img, normalize=normalize)
    if img.dim() == 4:
        img = img[0]
    arr = img.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    Image.fromarray(arr).save(str(path))

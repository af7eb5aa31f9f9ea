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


def save_image(tensor: torch.Tensor, path: str | Path) -> None:
    """``torchvision.utils.save_image`` for one image: ``mul(255).add(0.5).clamp(0,255)`` -> uint8 PNG."""
    img = tensor.detach()
    if img.dim() == 4:
        img = img[0]
    arr = img.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
    Image.fromarray(arr).save(str(path))

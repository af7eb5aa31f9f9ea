"""Shared type aliases and small records."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Literal

import torch

InitMethod = Literal["content", "random", "white"]
VideoMode = Literal["realtime", "postprocess"]
Precision = Literal["fp32", "bf16"]
LossHistory = dict[str, list[float]]
TensorList = list[torch.Tensor]


@dataclass(slots=True)
class InputPaths:
    """Content and style image paths."""

    content_path: str
    style_path: str


@dataclass(slots=True)
class SaveOptions:
    """What the final save step should write and how to name it."""

    content_name: str
    style_name: str
    video_name: str | None = None
    gif_name: str | None = None
    normalize: bool = True
    video_created: bool = True
    gif_created: bool = False
    plot_losses: bool = True

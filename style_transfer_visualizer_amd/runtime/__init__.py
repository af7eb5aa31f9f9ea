"""Host-side runtime glue around the hot path: device/seed set-up, input validation, output files.

Behaviour follows reference runtime/{device,validation,output}.py; nothing here runs per step.
"""
from __future__ import annotations

import random
from collections.abc import Callable
from pathlib import Path

import torch

from .. import image_io
from ..constants import VIDEO_QUALITY_MAX, VIDEO_QUALITY_MIN
from ..logging_utils import logger
from ..type_defs import LossHistory, SaveOptions


def setup_device(device_name: str) -> torch.device:
    """``"cuda"`` is the MI355X under ROCm; falls back to CPU with a warning like the reference.

    (On CPU the HIP model refuses to run - the fallback only keeps non-GPU tooling usable.)
    """
    if device_name == "cuda" and not torch.cuda.is_available():
        logger.warning("CUDA requested but not available. Falling back to CPU.")
        device = torch.device("cpu")
    else:
        device = torch.device(device_name)
    logger.info("Using device: %s", device)
    return device


def setup_random_seed(seed: int) -> None:
    """Seed torch (CPU + GPU) and Python's ``random``."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    random.seed(seed)


def validate_input_paths(content_path: str, style_path: str) -> None:
    if not Path(content_path).is_file():
        msg = f"Content image not found: {content_path}"
        raise FileNotFoundError(msg)
    if not Path(style_path).is_file():
        msg = f"Style image not found: {style_path}"
        raise FileNotFoundError(msg)


def validate_parameters(video_quality: int) -> None:
    if video_quality < VIDEO_QUALITY_MIN or video_quality > VIDEO_QUALITY_MAX:
        msg = f"Video quality must be between 1 and 10, got {video_quality}"
        raise ValueError(msg)


def setup_output_directory(output_path: str, path_factory: Callable[[str], Path] = Path) -> Path:
    """Create the output directory; fall back to ``style_transfer_output`` if that fails."""
    target = path_factory(output_path)
    try:
        target.mkdir(parents=True, exist_ok=True)
    except OSError:
        fallback = path_factory("style_transfer_output")
        fallback.mkdir(parents=True, exist_ok=True)
        return fallback
    return target


def stylized_image_path_from_names(output_dir: Path, content_name: str, style_name: str) -> Path:
    return output_dir / f"stylized_{content_name}_x_{style_name}.png"


def stylized_image_path_from_paths(output_dir: Path, content_path: Path, style_path: Path) -> Path:
    def stem(p: Path) -> str:
        return p.stem.replace(" ", "_")
    return stylized_image_path_from_names(output_dir, stem(content_path), stem(style_path))


def save_outputs(input_img: torch.Tensor, loss_metrics: LossHistory, output_dir: Path, elapsed: float,
                 opts: SaveOptions, *, plot_name: str = "loss_plot.png") -> None:
    """Write the final PNG (and the loss plot when matplotlib is present and plotting is on)."""
    try:
        if not output_dir.exists():
            output_dir.mkdir(parents=True, exist_ok=True)
            logger.info("Created output directory: %s", output_dir)
    except OSError as exc:
        logger.error("Failed to create output directory: %s", exc)
        output_dir = Path("style_transfer_output")
        output_dir.mkdir(exist_ok=True)
        logger.info("Using fallback directory: %s", output_dir)
    final_path = stylized_image_path_from_names(output_dir, opts.content_name, opts.style_name)
    image_io.save_image(input_img, final_path, normalize=opts.normalize)
    if opts.video_created and opts.video_name:
        logger.info("Video saved to: %s", output_dir / opts.video_name)
    if opts.plot_losses and loss_metrics:
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, ax = plt.subplots(figsize=(10, 6))
            for key, values in loss_metrics.items():
                if values:
                    ax.plot(values, label=key)
            ax.set_xlabel("Step")
            ax.set_ylabel("Loss")
            ax.set_yscale("log")
            ax.legend()
            fig.savefig(output_dir / plot_name)
            plt.close(fig)
        except ImportError:
            logger.warning("matplotlib not found: skipping loss plot.")
    logger.info("Style transfer completed in %.2f seconds", elapsed)
    logger.info("Final stylized image saved to: %s", final_path)

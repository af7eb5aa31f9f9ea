"""Configuration schema, TOML loader and CLI override rules.

Field names, bounds and override precedence follow reference config.py:53-309
so existing ``config.toml`` files and CLI invocations keep working.  The one
addition is ``hardware.precision`` ("fp32" parity mode | "bf16" storage).
"""
from __future__ import annotations

from collections.abc import Callable, Mapping
from pathlib import Path
from typing import Any

from pydantic import BaseModel, Field

from . import config_defaults as d
from .constants import VIDEO_QUALITY_MAX, VIDEO_QUALITY_MIN
from .logging_utils import logger
from .type_defs import InitMethod, Precision, VideoMode


class OptimizationConfig(BaseModel):
    """Optimisation settings."""

    steps: int = Field(d.DEFAULT_STEPS, ge=1)
    style_w: float = Field(d.DEFAULT_STYLE_WEIGHT, ge=0)
    content_w: float = Field(d.DEFAULT_CONTENT_WEIGHT, ge=0)
    lr: float = Field(d.DEFAULT_LEARNING_RATE, gt=0)
    init_method: InitMethod = Field(d.DEFAULT_INIT_METHOD)
    seed: int = Field(d.DEFAULT_SEED, ge=0)
    normalize: bool = d.DEFAULT_NORMALIZE
    lbfgs_max_iter: int = Field(d.DEFAULT_LBFGS_MAX_ITER, ge=1)
    lbfgs_max_eval: int = Field(d.DEFAULT_LBFGS_MAX_EVAL, ge=1)
    style_layers: list[int] = Field(default_factory=lambda: list(d.DEFAULT_STYLE_LAYERS))
    content_layers: list[int] = Field(default_factory=lambda: list(d.DEFAULT_CONTENT_LAYERS))


class VideoConfig(BaseModel):
    """Timelapse / GIF output settings."""

    save_every: int = Field(d.DEFAULT_SAVE_EVERY, ge=1)
    fps: int = Field(d.DEFAULT_FPS, ge=1, le=60)
    quality: int = Field(d.DEFAULT_VIDEO_QUALITY, ge=VIDEO_QUALITY_MIN, le=VIDEO_QUALITY_MAX)
    create_video: bool = d.DEFAULT_CREATE_VIDEO
    final_only: bool = d.DEFAULT_FINAL_ONLY
    intro_enabled: bool = d.DEFAULT_VIDEO_INTRO_ENABLED
    intro_duration_seconds: float = Field(d.DEFAULT_VIDEO_INTRO_DURATION, ge=0.0)
    metadata_title: str | None = None
    metadata_artist: str | None = None
    final_frame_compare: bool = d.DEFAULT_VIDEO_FINAL_FRAME_COMPARE
    outro_duration_seconds: float = Field(d.DEFAULT_VIDEO_OUTRO_DURATION, ge=0.0)
    mode: VideoMode = Field(d.DEFAULT_VIDEO_MODE)
    create_gif: bool = d.DEFAULT_CREATE_GIF
    gif_include_intro: bool = d.DEFAULT_GIF_INCLUDE_INTRO
    gif_include_outro: bool = d.DEFAULT_GIF_INCLUDE_OUTRO
    mode_override: bool = Field(default=False, exclude=True, repr=False)


class HardwareConfig(BaseModel):
    """Device selection (+ activation storage precision on MI355X)."""

    device: str = Field(d.DEFAULT_DEVICE)
    precision: Precision = Field(d.DEFAULT_PRECISION)


class OutputConfig(BaseModel):
    """Output directory and loss logging."""

    output: str = Field(d.DEFAULT_OUTPUT_DIR)
    log_every: int = Field(d.DEFAULT_LOG_EVERY, ge=1)
    log_loss: str | None = None
    plot_losses: bool = True


class StyleTransferConfig(BaseModel):
    """Root object mirroring the sections of ``config.toml``."""

    output: OutputConfig = Field(default_factory=lambda: OutputConfig.model_validate({}))
    optimization: OptimizationConfig = Field(default_factory=lambda: OptimizationConfig.model_validate({}))
    video: VideoConfig = Field(default_factory=lambda: VideoConfig.model_validate({}))
    hardware: HardwareConfig = Field(default_factory=lambda: HardwareConfig.model_validate({}))


def _read_toml(path: Path) -> dict:
    try:
        import tomllib as toml_reader  # py >= 3.11
    except ModuleNotFoundError:
        try:
            import tomli as toml_reader
        except ModuleNotFoundError:
            import tomlkit
            with path.open("r", encoding="utf-8") as f:
                return dict(tomlkit.load(f))
    with path.open("rb") as f:
        return toml_reader.load(f)


class ConfigLoader:
    """TOML -> validated ``StyleTransferConfig``; missing sections fall back to defaults."""

    @staticmethod
    def load(path: str) -> StyleTransferConfig:
        config_path = Path(path)
        if not config_path.is_file():
            msg = f"Config file not found: {path}"
            raise FileNotFoundError(msg)
        return StyleTransferConfig.model_validate(_read_toml(config_path))


def parse_int_list(value: str | list[int]) -> list[int]:
    """``"0,5,10"`` -> ``[0, 5, 10]`` (lists pass through)."""
    if isinstance(value, list):
        return value
    return [int(v) for v in value.split(",")]


# CLI key -> (section, attribute) for plain "present in args => overwrite" options
_DIRECT = {
    "output": ("output", "output"), "log_every": ("output", "log_every"), "log_loss": ("output", "log_loss"),
    "steps": ("optimization", "steps"), "style_w": ("optimization", "style_w"),
    "content_w": ("optimization", "content_w"), "lr": ("optimization", "lr"),
    "init_method": ("optimization", "init_method"), "seed": ("optimization", "seed"),
    "save_every": ("video", "save_every"), "fps": ("video", "fps"), "quality": ("video", "quality"),
    "metadata_title": ("video", "metadata_title"), "metadata_artist": ("video", "metadata_artist"),
    "create_gif": ("video", "create_gif"), "gif_include_intro": ("video", "gif_include_intro"),
    "gif_include_outro": ("video", "gif_include_outro"),
    "final_frame_compare": ("video", "final_frame_compare"),
    "device": ("hardware", "device"), "precision": ("hardware", "precision"),
}
# truthy flag -> (section, attribute, value)
_FLAGS = {
    "no_plot": ("output", "plot_losses", False), "no_normalize": ("optimization", "normalize", False),
    "no_video": ("video", "create_video", False), "no_intro": ("video", "intro_enabled", False),
    "final_only": ("video", "final_only", True),
}


def build_config_from_cli(
    cli_args: Mapping[str, Any],
    *,
    loader: Callable[[str], StyleTransferConfig] | None = None,
    base_config: StyleTransferConfig | None = None,
) -> StyleTransferConfig:
    """Apply CLI overrides on top of a TOML/base/default config (reference config.py:181-207)."""
    args = dict(cli_args)
    if base_config is not None:
        cfg = base_config.model_copy(deep=True)
    elif args.get("config"):
        cfg = (loader or ConfigLoader.load)(args["config"])
    else:
        cfg = StyleTransferConfig.model_validate({})

    _apply_output_overrides(cfg, args)
    _apply_optimization_overrides(cfg, args)
    _apply_video_overrides(cfg, args)
    _apply_hardware_overrides(cfg, args)
    _enforce_csv_plot_rule(cfg)
    return cfg


def _apply_section(cfg: StyleTransferConfig, args: Mapping[str, Any], section: str) -> None:
    """The table-driven part of one section: keys present in ``args`` overwrite, truthy flags set their value."""
    for key, (sec, attr) in _DIRECT.items():
        if sec == section and key in args:
            setattr(getattr(cfg, sec), attr, args[key])
    for key, (sec, attr, value) in _FLAGS.items():
        if sec == section and args.get(key):
            setattr(getattr(cfg, sec), attr, value)


# Per-section entry points under the names the reference's own tests reach for (config.py:210-299 there).
def _apply_output_overrides(cfg: StyleTransferConfig, args: Mapping[str, Any]) -> None:
    _apply_section(cfg, args, "output")


def _apply_optimization_overrides(cfg: StyleTransferConfig, args: Mapping[str, Any]) -> None:
    _apply_section(cfg, args, "optimization")
    if args.get("style_layers"):
        cfg.optimization.style_layers = parse_int_list(args["style_layers"])
    if args.get("content_layers"):
        cfg.optimization.content_layers = parse_int_list(args["content_layers"])


def _apply_video_overrides(cfg: StyleTransferConfig, args: Mapping[str, Any]) -> None:
    _apply_section(cfg, args, "video")
    for key, attr in (("intro_duration", "intro_duration_seconds"), ("outro_duration", "outro_duration_seconds")):
        if key in args:
            setattr(cfg.video, attr, max(args[key], 0.0))       # negative durations mean "none"
    if "video_mode" in args:
        cfg.video.mode = args["video_mode"]
        cfg.video.mode_override = True
    # a non-default mode that came from a TOML file / base config counts as an explicit choice too
    if not cfg.video.mode_override and cfg.video.mode != d.DEFAULT_VIDEO_MODE:
        cfg.video.mode_override = True


def _apply_hardware_overrides(cfg: StyleTransferConfig, args: Mapping[str, Any]) -> None:
    _apply_section(cfg, args, "hardware")


def _enforce_csv_plot_rule(cfg: StyleTransferConfig) -> None:
    """CSV logging replaces the loss plot (reference config.py:302-309)."""
    if getattr(cfg.output, "log_loss", None) and cfg.output.plot_losses:
        logger.warning(
            "Loss plotting is disabled because CSV logging is enabled. "
            "Only loss CSV will be created.",
        )
        cfg.output.plot_losses = False

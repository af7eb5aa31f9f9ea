"""``style-visualizer`` command line: same flags and override rules as reference cli.py:26-354.

Options that only have meaning for presentation features outside this build (comparison grids)
are accepted and reported as unavailable.  One addition: ``--precision {fp32,bf16}``.
"""
from __future__ import annotations

import argparse
import sys

from . import config as stv_config
from . import main as stv_main
from .config_defaults import DEFAULT_LOG_EVERY
from .constants import VIDEO_QUALITY_MAX, VIDEO_QUALITY_MIN
from .logging_utils import logger
from .type_defs import InputPaths

S = argparse.SUPPRESS


def build_arg_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(
        description="Neural Style Transfer on AMD MI355X (hand-written HIP kernels)",
        formatter_class=argparse.RawDescriptionHelpFormatter,
        epilog="Normalization is enabled by default. Use --no-normalize to disable it")
    req = p.add_argument_group("required arguments")
    req.add_argument("--content", type=str, help="Path to content image")
    req.add_argument("--style", type=str, help="Path to style image")

    out = p.add_argument_group("output")
    out.add_argument("--output", type=str, default=S, help="Output directory")
    out.add_argument("--no-plot", action="store_true", help="Disable loss plotting")
    out.add_argument("--log-loss", type=str, help="CSV file for loss metrics (disables the loss plot)")
    out.add_argument("--log-every", type=int, default=DEFAULT_LOG_EVERY, help="Log losses every N steps")
    out.add_argument("--compare-inputs", action="store_true", help="(not available in this build)")
    out.add_argument("--compare-result", action="store_true", help="(not available in this build)")

    opt = p.add_argument_group("optimization")
    opt.add_argument("--steps", type=int, default=S, help="Number of optimization steps")
    opt.add_argument("--style-w", type=float, default=S, help="Style weight")
    opt.add_argument("--content-w", type=float, default=S, help="Content weight")
    opt.add_argument("--lr", type=float, default=S, help="Learning rate")
    opt.add_argument("--init-method", choices=["random", "white", "content"], default=S, help="Initialization method")
    opt.add_argument("--seed", type=int, default=S, help="Random seed")
    opt.add_argument("--no-normalize", action="store_true", help="Disable VGG19 normalization")
    opt.add_argument("--style-layers", type=str, help="Comma-separated VGG19 layer indices for style loss")
    opt.add_argument("--content-layers", type=str, help="Comma-separated VGG19 layer indices for content loss")

    vid = p.add_argument_group("video")
    vid.add_argument("--save-every", type=int, default=S, help="Save a frame every N steps")
    vid.add_argument("--fps", type=int, default=S, help="Frames per second for video")
    vid.add_argument("--quality", type=int, default=S, help="Video quality 1-10")
    vid.add_argument("--no-video", action="store_true", help="Disable video creation")
    vid.add_argument("--final-only", action="store_true", help="Only save the final image")
    vid.add_argument("--no-intro", action="store_true", help="Disable the intro segment")
    vid.add_argument("--intro-duration", type=float, default=S)
    vid.add_argument("--no-final-frame-compare", dest="final_frame_compare", action="store_false", default=S)
    vid.add_argument("--outro-duration", type=float, default=S)
    vid.add_argument("--metadata-title", type=str, default=S)
    vid.add_argument("--metadata-artist", type=str, default=S)
    vid.add_argument("--gif", dest="create_gif", action="store_true", default=S)
    vid.add_argument("--no-gif", dest="create_gif", action="store_false", default=S)
    vid.add_argument("--gif-include-intro", dest="gif_include_intro", action="store_true", default=S)
    vid.add_argument("--gif-include-outro", dest="gif_include_outro", action="store_true", default=S)
    vid.add_argument("--video-mode", choices=["realtime", "postprocess"], default=S)

    hw = p.add_argument_group("hardware")
    hw.add_argument("--device", type=str, default=S, help="Device to run on (cuda = MI355X under ROCm)")
    hw.add_argument("--precision", choices=["fp32", "bf16"], default=S,
                    help="Activation storage: fp32 (parity mode) or bf16 (fp32 accumulate)")

    cfg = p.add_argument_group("configuration")
    cfg.add_argument("--config", type=str, help="Path to config.toml")
    cfg.add_argument("--validate-config-only", action="store_true", help="Validate the config file and exit")
    return p


def log_parameters(paths: InputPaths, cfg: stv_config.StyleTransferConfig,
                   args: argparse.Namespace | None = None) -> None:
    """One INFO line per setting, "Label: value" (the labels of reference cli.py:247-300, which its tests grep for;
    plus the two hardware settings this build adds)."""
    def yes(flag: bool) -> str:
        return "Yes" if flag else "No"

    def on(flag: bool) -> str:
        return "Enabled" if flag else "Disabled"
    o, v = cfg.optimization, cfg.video
    rows: list[tuple[str, object]] = [("Content image loaded", paths.content_path), ("Style image loaded", paths.style_path)]
    if getattr(args, "config", None):
        rows.append(("Loaded config from", args.config))
    rows += [
        ("Output Directory", cfg.output.output), ("Steps", o.steps), ("Save Every", v.save_every),
        ("Style Weight", f"{o.style_w:g}"), ("Content Weight", f"{o.content_w:g}"), ("Learning Rate", f"{o.lr:g}"),
        ("Style Layers", o.style_layers), ("Content Layers", o.content_layers),
        ("FPS for Timelapse Video", v.fps), ("Video Quality", f"{v.quality} ({VIDEO_QUALITY_MIN}-{VIDEO_QUALITY_MAX} scale)"),
        ("Initialization Method", o.init_method), ("Normalization", on(o.normalize)),
        ("Video Creation", on(v.create_video)), ("Intro Duration (s)", f"{v.intro_duration_seconds:.2f}"),
        ("Outro Duration (s)", f"{v.outro_duration_seconds:.2f}"),
        ("GIF Export", on(v.create_gif)), ("GIF Intro Included", yes(v.gif_include_intro)),
        ("GIF Outro Included", yes(v.gif_include_outro)), ("Video Mode", v.mode),
        ("Loss Plotting", on(cfg.output.plot_losses)), ("Random Seed", o.seed),
        ("Device", cfg.hardware.device), ("Precision", cfg.hardware.precision),
    ]
    for label, value in rows:
        logger.info("%s: %s", label, value)


def parse_int_list(s: str | list[int]) -> list[int]:
    """``"0,1,2"`` or a list of ints -> list of ints (reference cli.py:303-314 re-exports config's helper)."""
    return stv_config.parse_int_list(s)


def run_from_args(args: argparse.Namespace) -> None:
    base_cfg = None
    if args.config:
        base_cfg = stv_config.ConfigLoader.load(args.config)
        if args.validate_config_only:
            logger.info("Config %s validated successfully.", args.config)
            sys.exit(0)
    cfg = stv_config.build_config_from_cli(vars(args), base_config=base_cfg)
    paths = InputPaths(content_path=args.content, style_path=args.style)
    log_parameters(paths, cfg, args)
    stv_main.style_transfer(paths, cfg)
    if args.compare_inputs or args.compare_result:
        logger.warning("Comparison grids are a presentation feature outside this build; skipped.")


def main(argv: list[str] | None = None) -> None:
    parser = build_arg_parser()
    args = parser.parse_args(argv)
    if not args.validate_config_only and (not args.content or not args.style):
        parser.error("the following arguments are required: --content, --style")
    run_from_args(args)


if __name__ == "__main__":  # pragma: no cover
    main()

"""CLI surface: flags, prefix matching, override quirks (reference cli.py / config.py)."""
from __future__ import annotations

import pytest

from style_transfer_visualizer_amd import cli, config as stv_config


def _cfg(argv):
    args = cli.build_arg_parser().parse_args(argv)
    return args, stv_config.build_config_from_cli(vars(args))


def test_baseline_plumbing_command_line_parses():
    # BASELINE.json configs[0]; "--init" reaches --init-method through argparse prefix matching
    args, cfg = _cfg("--content c.png --style s.png --steps 50 --init content --device cpu --no-video "
                     "--final-only --seed 0".split())
    assert cfg.optimization.steps == 50 and cfg.optimization.init_method == "content"
    assert cfg.hardware.device == "cpu" and cfg.video.create_video is False and cfg.video.final_only is True
    assert args.content == "c.png" and args.style == "s.png"


def test_suppressed_defaults_leave_config_untouched():
    args, cfg = _cfg(["--content", "a", "--style", "b"])
    present = vars(args)
    assert "steps" not in present and "device" not in present and "fps" not in present
    assert present["log_every"] == 10 and present["log_loss"] is None     # real defaults -> always override
    assert cfg.optimization.steps == 1500 and cfg.hardware.precision == "fp32"


def test_flags_map_to_config():
    _, cfg = _cfg("--content a --style b --style-layers 0,2 --content-layers 1 --no-normalize --gif --no-intro "
                  "--precision bf16 --log-loss x.csv --video-mode postprocess --no-final-frame-compare".split())
    assert cfg.optimization.style_layers == [0, 2] and cfg.optimization.content_layers == [1]
    assert cfg.optimization.normalize is False and cfg.video.create_gif is True and cfg.video.intro_enabled is False
    assert cfg.hardware.precision == "bf16" and cfg.output.plot_losses is False
    assert cfg.video.mode == "postprocess" and cfg.video.mode_override and cfg.video.final_frame_compare is False


def test_required_arguments_enforced(capsys):
    with pytest.raises(SystemExit):
        cli.main([])
    assert "--content, --style" in capsys.readouterr().err


def test_validate_config_only(tmp_path):
    p = tmp_path / "c.toml"
    p.write_text("[optimization]\nsteps = 3\n")
    with pytest.raises(SystemExit) as e:
        cli.main(["--config", str(p), "--validate-config-only"])
    assert e.value.code == 0

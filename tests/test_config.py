"""Config schema, TOML loading and CLI override rules."""
from __future__ import annotations

import pytest
from pydantic import ValidationError

from style_transfer_visualizer_amd import config as cfg
from style_transfer_visualizer_amd import config_defaults as d


def test_defaults_match_the_reference_surface():
    c = cfg.StyleTransferConfig.model_validate({})
    o = c.optimization
    assert (o.steps, o.style_w, o.content_w, o.lr) == (1500, 1e5, 1.0, 1.0)
    assert o.init_method == "random" and o.seed == 0 and o.normalize is True
    assert (o.lbfgs_max_iter, o.lbfgs_max_eval) == (1, 1)
    assert o.style_layers == [0, 5, 10, 19, 28] and o.content_layers == [21]
    assert c.video.save_every == 20 and c.video.fps == 10 and c.video.mode == "realtime"
    assert c.hardware.device == "cuda" and c.hardware.precision == d.DEFAULT_PRECISION
    assert c.output.output == "out" and c.output.log_every == 10 and c.output.log_loss is None


@pytest.mark.parametrize("section,field,value", [
    ("optimization", "steps", 0), ("optimization", "lr", 0.0), ("optimization", "style_w", -1.0),
    ("optimization", "init_method", "blue"), ("video", "fps", 61), ("video", "quality", 11),
    ("output", "log_every", 0), ("hardware", "precision", "fp8"),
])
def test_bounds_are_validated(section, field, value):
    with pytest.raises(ValidationError):
        cfg.StyleTransferConfig.model_validate({section: {field: value}})


def test_toml_round_trip(tmp_path):
    p = tmp_path / "config.toml"
    p.write_text('[optimization]\nsteps = 7\nstyle_w = 1e6\nstyle_layers = [0, 2]\n'
                 '[hardware]\ndevice = "cpu"\nprecision = "bf16"\n[output]\nlog_every = 3\n')
    c = cfg.ConfigLoader.load(str(p))
    assert c.optimization.steps == 7 and c.optimization.style_w == 1e6 and c.optimization.style_layers == [0, 2]
    assert c.hardware.device == "cpu" and c.hardware.precision == "bf16" and c.output.log_every == 3
    assert c.video.fps == 10          # untouched sections keep defaults
    with pytest.raises(FileNotFoundError, match="Config file not found"):
        cfg.ConfigLoader.load(str(tmp_path / "missing.toml"))


def test_cli_overrides_only_for_present_keys():
    base = cfg.StyleTransferConfig.model_validate({"optimization": {"steps": 9}, "output": {"log_every": 4}})
    c = cfg.build_config_from_cli({"style_w": 5.0, "no_video": True, "final_only": True}, base_config=base)
    assert c.optimization.steps == 9 and c.optimization.style_w == 5.0
    assert c.video.create_video is False and c.video.final_only is True
    assert c.output.log_every == 4
    assert base.optimization.style_w == 1e5                 # base is not mutated
    # keys that argparse always supplies overwrite the file (reference quirk, config.py:216-219)
    c2 = cfg.build_config_from_cli({"log_every": 10, "log_loss": None}, base_config=base)
    assert c2.output.log_every == 10


def test_layer_lists_and_durations():
    c = cfg.build_config_from_cli({"style_layers": "0,2,4", "content_layers": [1, 3], "intro_duration": -2.0,
                                   "outro_duration": 3.5, "no_normalize": True, "no_plot": True})
    assert c.optimization.style_layers == [0, 2, 4] and c.optimization.content_layers == [1, 3]
    assert c.video.intro_duration_seconds == 0.0 and c.video.outro_duration_seconds == 3.5
    assert c.optimization.normalize is False and c.output.plot_losses is False
    assert cfg.parse_int_list("3, 4") == [3, 4]


def test_csv_logging_disables_plot():
    c = cfg.build_config_from_cli({"log_loss": "loss.csv"})
    assert c.output.log_loss == "loss.csv" and c.output.plot_losses is False


def test_video_mode_override_flag():
    assert cfg.build_config_from_cli({}).video.mode_override is False
    assert cfg.build_config_from_cli({"video_mode": "postprocess"}).video.mode_override is True
    base = cfg.StyleTransferConfig.model_validate({"video": {"mode": "postprocess"}})
    assert cfg.build_config_from_cli({}, base_config=base).video.mode_override is True


def test_config_file_through_loader_hook():
    seen = []

    def loader(path):
        seen.append(path)
        return cfg.StyleTransferConfig.model_validate({"optimization": {"steps": 3}})
    c = cfg.build_config_from_cli({"config": "x.toml", "steps": 5}, loader=loader)
    assert seen == ["x.toml"] and c.optimization.steps == 5


# ---- loader and section-model behaviours the reference's tests/test_config.py pins ----------------------------
def _toml(tmp_path, text):
    p = tmp_path / "cfg.toml"
    p.write_text(text)
    return str(p)


def test_loader_missing_file_partial_and_empty_files(tmp_path):
    """reference :86-104, :273-289: a missing file raises; missing sections / an empty file fall back to the defaults."""
    from style_transfer_visualizer_amd import config_defaults as d
    with pytest.raises(FileNotFoundError):
        cfg.ConfigLoader.load(str(tmp_path / "nonexistent_file.toml"))
    part = cfg.ConfigLoader.load(_toml(tmp_path, "[optimization]\nsteps = 42\n"))
    assert part.optimization.steps == 42 and part.optimization.lr == d.DEFAULT_LEARNING_RATE
    assert part.video.fps == d.DEFAULT_FPS and part.hardware.device == d.DEFAULT_DEVICE
    assert part.video.outro_duration_seconds == d.DEFAULT_VIDEO_OUTRO_DURATION
    empty = cfg.ConfigLoader.load(_toml(tmp_path, ""))
    assert empty.optimization.steps == d.DEFAULT_STEPS and empty.video.quality == d.DEFAULT_VIDEO_QUALITY
    assert empty.video.create_gif == d.DEFAULT_CREATE_GIF and empty.video.gif_include_intro == d.DEFAULT_GIF_INCLUDE_INTRO
    assert empty.video.gif_include_outro == d.DEFAULT_GIF_INCLUDE_OUTRO
    assert empty.output.output == d.DEFAULT_OUTPUT_DIR and empty.output.log_every == d.DEFAULT_LOG_EVERY


def test_loader_rejects_wrong_types(tmp_path):
    """reference :291-297."""
    with pytest.raises(ValidationError):
        cfg.ConfigLoader.load(_toml(tmp_path, '[optimization]\nsteps = "not_an_int"\n'))


@pytest.mark.parametrize("model,field,value", [
    ("OptimizationConfig", "seed", -42), ("OptimizationConfig", "steps", -5), ("OptimizationConfig", "content_w", -1.0),
    ("VideoConfig", "fps", 100), ("VideoConfig", "fps", 0), ("VideoConfig", "quality", 0),
])
def test_section_models_name_the_offending_field(model, field, value):
    """reference :106-226: the section models can be built on their own and their ValidationError names the field."""
    with pytest.raises(ValidationError) as err:
        getattr(cfg, model)(**{field: value})
    assert field in str(err.value)


def test_output_section_defaults_and_default_layers():
    """reference :264-271, :300-305."""
    from style_transfer_visualizer_amd import config_defaults as d
    out = cfg.OutputConfig.model_validate({})
    assert out.output == d.DEFAULT_OUTPUT_DIR and out.log_every == d.DEFAULT_LOG_EVERY
    assert out.log_loss is None and out.plot_losses is True
    c = cfg.StyleTransferConfig.model_validate({})
    assert c.optimization.style_layers == list(d.DEFAULT_STYLE_LAYERS)
    assert c.optimization.content_layers == list(d.DEFAULT_CONTENT_LAYERS)


def test_base_config_with_a_non_default_mode_counts_as_an_explicit_choice():
    """reference :228-238."""
    from style_transfer_visualizer_amd import config_defaults as d
    base = cfg.StyleTransferConfig.model_validate({})
    base.video.mode = next(m for m in ("postprocess", "realtime") if m != d.DEFAULT_VIDEO_MODE)
    base.video.mode_override = False
    built = cfg.build_config_from_cli({}, base_config=base)
    assert built.video.mode == base.video.mode and built.video.mode_override is True
    assert base.video.mode_override is False                      # the base object is not touched

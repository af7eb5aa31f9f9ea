"""The ``runtime`` sub-modules and config helpers under the names the reference's tests import
(reference tests/runtime/test_{device,output,validation,version}.py, tests/test_config.py)."""
from __future__ import annotations

import pytest

from style_transfer_visualizer_amd import config as stv_config
from style_transfer_visualizer_amd import config_defaults as d
from style_transfer_visualizer_amd import constants, runtime
from style_transfer_visualizer_amd.runtime import device, output, validation
from style_transfer_visualizer_amd.runtime import version as runtime_version


def test_submodules_expose_the_package_functions():
    assert device.setup_device is runtime.setup_device and device.setup_random_seed is runtime.setup_random_seed
    assert validation.validate_input_paths is runtime.validate_input_paths
    assert validation.validate_parameters is runtime.validate_parameters
    for name in ("save_outputs", "setup_output_directory", "stylized_image_path_from_names", "stylized_image_path_from_paths"):
        assert getattr(output, name) is getattr(runtime, name)


def test_version_prefers_an_installed_distribution(monkeypatch):
    monkeypatch.setattr(runtime_version.importlib_metadata, "version", lambda name: "9.9.9")
    assert runtime_version.resolve_project_version() == "9.9.9"


def _no_distribution(monkeypatch):
    def missing(name):
        raise runtime_version.importlib_metadata.PackageNotFoundError
    monkeypatch.setattr(runtime_version.importlib_metadata, "version", missing)


def test_version_from_the_nearest_pyproject(monkeypatch, tmp_path):
    _no_distribution(monkeypatch)
    (tmp_path / "pyproject.toml").write_text("[project]\nversion = '1.2.3'\n")
    (tmp_path / "pkg").mkdir()
    monkeypatch.setattr(runtime_version, "__file__", str(tmp_path / "pkg" / "version.py"))
    assert runtime_version.resolve_project_version() == "1.2.3"


def test_version_falls_back_when_the_pyproject_cannot_be_read(monkeypatch, tmp_path, caplog):
    _no_distribution(monkeypatch)
    (tmp_path / "pyproject.toml").write_text("[project]\nversion = '1.2.3'\n")
    (tmp_path / "pkg").mkdir()

    def boom(fh):
        raise OSError("unreadable")
    monkeypatch.setattr(runtime_version.tomllib, "load", boom)
    monkeypatch.setattr(runtime_version, "__file__", str(tmp_path / "pkg" / "version.py"))
    assert runtime_version.resolve_project_version() == "0.0.0"


def test_video_overrides_mark_a_non_default_mode_as_explicit():
    """Reference tests/test_config.py (``_apply_video_overrides``): a mode that differs from the default - wherever
    it came from - and a mode given on the command line both set ``mode_override``; an untouched default does not."""
    cfg = stv_config.StyleTransferConfig.model_validate({})
    stv_config._apply_video_overrides(cfg, {})
    assert cfg.video.mode == d.DEFAULT_VIDEO_MODE and cfg.video.mode_override is False

    other = next(m for m in ("postprocess", "realtime") if m != d.DEFAULT_VIDEO_MODE)
    cfg = stv_config.StyleTransferConfig.model_validate({"video": {"mode": other}})
    stv_config._apply_video_overrides(cfg, {})
    assert cfg.video.mode_override is True

    cfg = stv_config.StyleTransferConfig.model_validate({})
    stv_config._apply_video_overrides(cfg, {"video_mode": d.DEFAULT_VIDEO_MODE, "intro_duration": -3.0, "fps": 24})
    assert cfg.video.mode_override is True and cfg.video.intro_duration_seconds == 0.0 and cfg.video.fps == 24


def test_section_helpers_touch_only_their_section():
    cfg = stv_config.StyleTransferConfig.model_validate({})
    before = cfg.model_dump()
    stv_config._apply_hardware_overrides(cfg, {"device": "cpu", "fps": 5, "steps": 7})
    after = cfg.model_dump()
    assert after["hardware"]["device"] == "cpu"
    assert after["video"] == before["video"] and after["optimization"] == before["optimization"]
    stv_config._apply_optimization_overrides(cfg, {"steps": 7, "style_layers": "1, 2,3"})
    assert cfg.optimization.steps == 7 and cfg.optimization.style_layers == [1, 2, 3]


def test_presentation_constants_exist():
    assert constants.COLOR_BLACK == (0, 0, 0) and constants.COLOR_WHITE == (255, 255, 255)
    assert len(constants.COLOR_GREY) == 3 and len(constants.COLOR_BEIGE) == 3
    assert constants.RESOLUTION_FULL_HD == (1920, 1080)
    assert constants.VIDEO_CODEC == "libx264" and constants.ENCODING_BLOCK_SIZE == 16

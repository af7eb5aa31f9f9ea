"""Edge I/O either side of the hot path: values, not just file existence.

Mirrors what the reference pins in tests/test_image_io.py (tensor shape for landscape/portrait
inputs, black < -2 / white > 2.2 after Normalize, prepare_image_for_output clamps and is NaN-safe,
MIN_DIMENSION raises) and the output conversions of image_io.py:64-152, runtime/output.py:92-101
(``save_image`` = mul(255).add(0.5).clamp(0,255) -> uint8) and optimization.py:445-451
(frames: truncating ``*255``).
"""
from __future__ import annotations

import numpy as np
import pytest
import torch
from PIL import Image

from style_transfer_visualizer_amd import image_io
from style_transfer_visualizer_amd.constants import IMAGENET_MEAN, IMAGENET_STD

CPU = torch.device("cpu")


def _png(tmp_path, name, arr):
    p = tmp_path / name
    Image.fromarray(arr).save(p)
    return str(p)


@pytest.mark.parametrize("hw", [(72, 128), (128, 72), (64, 64)])
def test_load_shape_and_to_tensor_values(tmp_path, hw):
    h, w = hw
    rng = np.random.default_rng(0)
    arr = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
    t = image_io.load_image_to_tensor(_png(tmp_path, "a.png", arr), CPU, normalize=False)
    assert t.shape == (1, 3, h, w) and t.dtype == torch.float32
    # ToTensor: uint8 / 255, HWC -> CHW, exactly
    expect = torch.from_numpy(arr).permute(2, 0, 1).float().div(255).unsqueeze(0)
    assert torch.equal(t, expect)


def test_normalize_black_and_white_ranges(tmp_path):
    black = image_io.load_image_to_tensor(_png(tmp_path, "b.png", np.zeros((64, 64, 3), np.uint8)), CPU, normalize=True)
    white = image_io.load_image_to_tensor(_png(tmp_path, "w.png", np.full((64, 64, 3), 255, np.uint8)), CPU,
                                          normalize=True)
    assert float(black.max()) < -1.7 and float(black[:, 0].max()) < -2.0      # reference: "black < -2"
    assert float(white.min()) > 2.2
    for c in range(3):      # (x - mean) / std per channel
        assert float(black[0, c, 0, 0]) == pytest.approx(-IMAGENET_MEAN[c] / IMAGENET_STD[c], rel=1e-6)
        assert float(white[0, c, 0, 0]) == pytest.approx((1 - IMAGENET_MEAN[c]) / IMAGENET_STD[c], rel=1e-6)


def test_too_small_raises_and_missing_file(tmp_path):
    p = _png(tmp_path, "s.png", np.zeros((63, 80, 3), np.uint8))
    with pytest.raises(ValueError, match=r"Image too small: 80x63\. Minimum dimension is 64px\."):
        image_io.load_image_to_tensor(p, CPU)
    with pytest.raises(FileNotFoundError, match="Image file not found"):
        image_io.load_image_to_tensor(str(tmp_path / "nope.png"), CPU)


def test_denormalize_round_trip_and_prepare_for_output():
    x = torch.rand(1, 3, 8, 8)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    z = (x - mean) / std
    np.testing.assert_allclose(image_io.denormalize(z).numpy(), x.numpy(), atol=1e-6)
    bad = torch.tensor([float("nan"), float("inf"), float("-inf"), -3.0, 0.25, 7.0]).view(1, 1, 1, 6).expand(1, 3, 1, 6)
    out = image_io.prepare_image_for_output(bad.clone(), normalize=False)
    assert out[0, 0, 0].tolist() == [0.0, 1.0, 0.0, 0.0, 0.25, 1.0]
    out_n = image_io.prepare_image_for_output(bad.clone(), normalize=True)
    assert torch.isfinite(out_n).all() and float(out_n.min()) >= 0.0 and float(out_n.max()) <= 1.0
    assert out_n[0, 0, 0, 0] == 0.0 and out_n[0, 0, 0, 1] == 1.0 and out_n[0, 0, 0, 2] == 0.0


def test_save_image_rounds_half_up_and_round_trips(tmp_path):
    # 0..255 / 255 must come back exactly; k/255 + 0.4/255 rounds down, + 0.6/255 rounds up
    vals = torch.arange(256, dtype=torch.float32).div(255)
    img = vals.view(1, 1, 16, 16).expand(1, 3, 16, 16).contiguous()
    p = tmp_path / "o.png"
    image_io.save_image(img, p)
    back = np.asarray(Image.open(p))
    assert back.shape == (16, 16, 3) and np.array_equal(back[..., 0].reshape(-1), np.arange(256))
    image_io.save_image(img + 0.4 / 255, p)
    assert np.array_equal(np.asarray(Image.open(p))[..., 1].reshape(-1), np.arange(256))
    image_io.save_image((img + 0.6 / 255).clamp(0, 1), p)
    assert np.array_equal(np.asarray(Image.open(p))[..., 2].reshape(-1), np.minimum(np.arange(256) + 1, 255))
    # normalize given: prepare_image_for_output is applied first (runtime/output.py:92-101)
    mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    image_io.save_image((img - mean) / std, p, normalize=True)
    assert np.abs(np.asarray(Image.open(p)).astype(int)[..., 0].reshape(-1) - np.arange(256)).max() <= 1


def test_frame_uint8_truncates_on_host_tensors():
    # optimization.py:445-451: (x * 255).astype("uint8") truncates - 0.999 -> 254, not 255
    x = torch.tensor([0.0, 0.5, 0.999, 1.0, 1.7, -0.2]).view(1, 1, 1, 6).expand(1, 3, 1, 6).contiguous()
    f = image_io.frame_uint8(x, normalize=False)
    assert f.shape == (1, 6, 3) and f.dtype == np.uint8
    assert f[0, :, 0].tolist() == [0, 127, 254, 255, 255, 0]


# ---- edge behaviours the reference's tests/test_image_io.py pins --------------------------------------------
def test_load_image_gives_rgb_and_reports_bad_files(tmp_path):
    """reference :27-45: RGB mode; a missing file -> FileNotFoundError; bytes that are no image -> OSError
    "Error loading image"."""
    import pytest
    from PIL import Image

    from style_transfer_visualizer_amd import image_io
    p = tmp_path / "grey.png"
    Image.new("L", (80, 70), 128).save(p)
    img = image_io.load_image(str(p))
    assert isinstance(img, Image.Image) and img.mode == "RGB" and img.size == (80, 70)
    with pytest.raises(FileNotFoundError):
        image_io.load_image(str(tmp_path / "nonexistent_image.jpg"))
    bad = tmp_path / "bad.jpg"
    bad.write_bytes(b"not an image data")
    with pytest.raises(OSError, match="Error loading image"):
        image_io.load_image(str(bad))


def test_dimension_checks_raise_for_small_and_warn_for_large(caplog):
    """reference :182-199."""
    import pytest
    from PIL import Image

    from style_transfer_visualizer_amd import image_io
    image_io.validate_image_dimensions(Image.new("RGB", (512, 512)))
    with pytest.raises(ValueError, match="Image too small"):
        image_io.validate_image_dimensions(Image.new("RGB", (32, 100)))
    caplog.set_level("WARNING")
    image_io.validate_image_dimensions(Image.new("RGB", (4000, 4000)))
    assert "may slow processing" in caplog.text


def test_denormalize_changes_values_and_keeps_batches():
    """reference :99-110."""
    import torch

    from style_transfer_visualizer_amd import image_io
    t = torch.randn(2, 3, 20, 20)
    out = image_io.denormalize(t)
    assert out.shape == t.shape and not torch.allclose(out, t)


def test_prepare_for_output_clamps_sanitises_and_keeps_shape():
    """reference :204-262: with and without normalisation the result is in [0, 1], batches keep their shape,
    extreme values are clamped, NaN / +-inf are sanitised."""
    import torch

    from style_transfer_visualizer_amd import image_io
    ramp = torch.tensor([[[-3.0, -2.0, -1.0], [0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]]).view(1, 1, 3, 3).repeat(1, 3, 1, 1)
    out = image_io.prepare_image_for_output(ramp, normalize=True)
    assert out.shape == ramp.shape and float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert not torch.allclose(out, ramp.clamp(0, 1))                      # denormalised first, then clamped
    plain = torch.tensor([[-0.5, 0.2, 0.7], [0.0, 1.0, 1.5]]).view(1, 1, 2, 3).repeat(1, 3, 1, 1)
    assert torch.allclose(image_io.prepare_image_for_output(plain, normalize=False), plain.clamp(0, 1))
    wild = torch.tensor([[-100.0, -50.0, 0.0], [1.0, 50.0, 100.0]]).view(1, 1, 2, 3).repeat(1, 3, 1, 1)
    batch = torch.rand(2, 3, 10, 10)
    for norm in (True, False):
        for t in (wild, batch):
            o = image_io.prepare_image_for_output(t, normalize=norm)
            assert o.shape == t.shape and float(o.min()) >= 0.0 and float(o.max()) <= 1.0
    bad = torch.tensor([[[[float("nan"), float("inf"), -float("inf")]]]]).repeat(1, 3, 1, 1)
    o = image_io.prepare_image_for_output(bad, normalize=False)
    assert bool(torch.isfinite(o).all()) and float(o.min()) >= 0.0 and float(o.max()) <= 1.0

"""Import the UNMODIFIED reference modules from /root/reference (build container only).

Test infrastructure (see ``oracle/__init__.py``).  Used only by
``oracle/make_golden.py`` to produce ``tests/golden/*.npz``; never imported at
run time on the GPU box (where /root/reference does not exist).

Recipe: the reference's arithmetic for this path lives in
``torch`` (present).  ``torchvision``/``tomlkit``/``imageio`` are not installed
and Python is 3.10, so before importing we register *empty name stand-ins* for
those third-party modules and alias ``tomllib``/``datetime.UTC``; they supply
no arithmetic.  The VGG network is injected through the same seam the
reference's own tests patch (``core_model.initialize_vgg``,
/root/reference/tests/test_core_model.py:149-157).
"""
from __future__ import annotations

import datetime
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def import_reference():
    """Return (core_model, optimization, config, loss_accumulator) modules."""
    if not os.path.isdir(REFERENCE_SRC):
        msg = f"{REFERENCE_SRC} not present: golden vectors can only be regenerated in the build container"
        raise RuntimeError(msg)
    sys.dont_write_bytecode = True
    if not hasattr(datetime, "UTC"):
        datetime.UTC = datetime.timezone.utc  # py3.10 shim for video.py:9
    try:
        import tomllib  # noqa: F401
    except ModuleNotFoundError:
        import tomli
        sys.modules["tomllib"] = tomli
    import tomli as _tomli

    if "tomlkit" not in sys.modules:
        _stub("tomlkit", load=lambda f: _tomli.loads(f.read()))
    if "imageio" not in sys.modules:
        _stub("imageio")
        _stub("imageio.v2")
    if "torchvision" not in sys.modules:
        class _W:
            url = "https://download.pytorch.org/models/vgg19-dcbb9e9d.pth"

        class _Weights:
            IMAGENET1K_V1 = _W()

        def _no_vgg(*_a, **_k):
            msg = "torchvision is not installed; inject a network via initialize_vgg"
            raise RuntimeError(msg)

        tv = _stub("torchvision")
        tv.models = _stub("torchvision.models", VGG19_Weights=_Weights, vgg19=_no_vgg)
        tv.transforms = _stub("torchvision.transforms")
        tv.utils = _stub("torchvision.utils")
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import style_transfer_visualizer.config as ref_config
    import style_transfer_visualizer.core_model as ref_core
    import style_transfer_visualizer.loss_accumulator as ref_acc
    import style_transfer_visualizer.optimization as ref_opt
    return ref_core, ref_opt, ref_config, ref_acc


def build_sequential(weights, cfg):
    """VGG-'E'-style ``nn.Sequential`` (Conv3x3 pad1 / ReLU(inplace) / MaxPool2)."""
    import torch
    from torch import nn

    layers: list[nn.Module] = []
    it = iter(weights)
    for v in cfg:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            w, b = next(it)
            conv = nn.Conv2d(w.shape[1], w.shape[0], kernel_size=3, padding=1)
            with torch.no_grad():
                conv.weight.copy_(w)
                conv.bias.copy_(b)
            layers += [conv, nn.ReLU(inplace=True)]
    seq = nn.Sequential(*layers).eval()
    for p in seq.parameters():
        p.requires_grad_(False)
    return seq

"""core_model host logic that runs without a GPU: init modes, block slicing, schedule lowering,
and the loud failure of the HIP path on CPU tensors."""
from __future__ import annotations

import pytest
import torch
from torch import nn

from style_transfer_visualizer_amd import _lib, core_model, plan, synthetic

CPU = torch.device("cpu")


@pytest.fixture
def vgg19(monkeypatch):
    monkeypatch.setattr(core_model, "initialize_vgg", lambda: core_model.build_vgg_features().eval())


def test_initialize_input_modes_and_errors():
    content = torch.rand(1, 3, 8, 8)
    same = core_model.initialize_input(content, "content")
    assert torch.equal(same, content) and same.requires_grad and not content.requires_grad
    assert torch.equal(core_model.initialize_input(content, "white"), torch.ones_like(content))
    rnd = core_model.initialize_input(content, "random")
    assert rnd.shape == content.shape and not torch.allclose(rnd, content)
    with pytest.raises(ValueError, match="Unsupported initialization method: blue"):
        core_model.initialize_input(content, "blue")
    with pytest.raises(TypeError, match=r"Expected content_img to be a Tensor"):
        core_model.initialize_input("nope", "content")


def test_vgg19_topology_and_default_slicing(vgg19):
    model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21])
    assert len(model.vgg_blocks) == 6
    assert model.style_ids == [0, 1, 2, 3, 5] and model.content_ids == [4]
    layers = [l for b in model.vgg_blocks for l in b]
    assert len(layers) == 29                                # layers 29..36 are dropped
    assert sum(isinstance(l, nn.Conv2d) for l in layers) == 13
    assert sum(isinstance(l, nn.MaxPool2d) for l in layers) == 4
    assert all(not l.inplace for l in layers if isinstance(l, nn.ReLU))
    assert model.style_targets is None and model.content_targets is None


def test_synthetic_weights_are_deterministic_and_he_scaled():
    a = synthetic.synthetic_conv_weights(0)
    b = synthetic.synthetic_conv_weights(0)
    assert len(a) == 16 and all(torch.equal(x[0], y[0]) for x, y in zip(a, b))
    w = a[5][0]
    assert w.shape == (256, 256, 3, 3)
    assert float(w.std()) == pytest.approx((2.0 / (9 * 256)) ** 0.5, rel=0.02)
    assert not torch.equal(a[0][0], synthetic.synthetic_conv_weights(1)[0][0])
    img = synthetic.synthetic_image(0, 4, 5, normalize=False)
    assert img.shape == (1, 3, 4, 5) and 0 <= float(img.min()) and float(img.max()) < 1


def test_no_cpu_fallback(vgg19):
    model = core_model.StyleContentModel([0, 5], [2])
    x = torch.rand(1, 3, 16, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.set_targets(x, x)
    model.style_targets, model.content_targets = [x], [x]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.loss_and_grad(x, 1.0, 1.0)
    model.style_targets = None
    with pytest.raises(RuntimeError, match="style_targets must be set before computing losses."):
        model(x)
    model.style_targets, model.content_targets = [x], None
    with pytest.raises(RuntimeError, match="content_targets must be set before computing losses."):
        model(x)


def test_precision_selection(monkeypatch):
    assert core_model.resolve_precision(None) == torch.float32
    assert core_model.resolve_precision("bf16") == torch.bfloat16
    monkeypatch.setenv("STV_PRECISION", "bf16")
    assert core_model.resolve_precision(None) == torch.bfloat16
    with pytest.raises(ValueError, match="Unsupported precision"):
        core_model.resolve_precision("fp8")


def _layers():
    return list(core_model.build_vgg_features().eval().children())


def test_schedule_lowering_for_default_vgg19():
    """Buffers and op flags of the lowered forward/backward schedule (no device work: CPU tensors)."""
    s = plan.Schedule(_layers(), [0, 5, 10, 19, 28], [21], 64, 48, torch.float32, CPU, with_grad=True)
    kinds = [n.kind for n in s.nodes]
    assert kinds.count("conv_first") == 1 and kinds.count("conv") == 12 and kinds.count("pool") == 4
    assert "relu" not in kinds                               # every ReLU is fused or applied on load
    assert len(s.style_taps) == 5 and len(s.content_taps) == 1
    tapped = [n for n in s.nodes if n.dst.taps]
    assert [n.layer for n in tapped] == [0, 5, 10, 19, 21, 28]
    assert all(not n.dst.relu_fused for n in tapped)         # taps are pre-ReLU conv outputs (F6)
    assert all(n.dst.relu_fused for n in s.nodes if n.kind == "conv" and not n.dst.taps)
    # consumers of a tapped conv apply the ReLU while staging
    assert [n.layer for n in s.nodes if n.relu_in] == [2, 7, 12, 21, 23]
    assert (s.nodes[-1].dst.H, s.nodes[-1].dst.W, s.nodes[-1].dst.C) == (4, 3, 512)

    x = torch.zeros(1, 3, 64, 48)
    fwd = s.forward_ops(x)
    assert [o.op for o in fwd].count(_lib.OP_CONV) == 12 and fwd[0].op == _lib.OP_CONV_FIRST_FWD
    for tap in s.style_taps:
        tap.sgrad = torch.zeros(1, tap.buf.C, tap.buf.C)
    for tap in s.content_taps:
        tap.target = torch.zeros_like(tap.buf.act)
    bwd = s.backward_ops(torch.zeros_like(x), style_coef=1.0, content_coef=1.0, coef_dev=None)
    ops_ = [o.op for o in bwd]
    assert ops_.count(_lib.OP_CONV) == 12 + 5               # 12 dgrads + 5 Gram products (1x1)
    assert ops_.count(_lib.OP_POOL_BWD) == 4 and ops_.count(_lib.OP_CONTENT_GRAD) == 1
    assert ops_[-1] == _lib.OP_CONV_FIRST_DGRAD and ops_[0] == _lib.OP_CONV and bwd[0].taps == 1
    dgrads = [o for o in bwd if o.op == _lib.OP_CONV and o.taps == 9]
    # a dgrad masks when its target is a conv output behind a ReLU; the 4 dgrads that write a pooled
    # gradient do not - there the mask is applied by the max-pool backward on the pre-pool buffer
    assert sum(bool(o.flags & _lib.MASK) for o in dgrads) == 8
    assert all(o.flags & _lib.MASK for o in bwd if o.op == _lib.OP_POOL_BWD)
    # a gradient buffer is stored once, then accumulated into
    first_writer = {}
    for o in bwd:
        if o.op in (_lib.OP_CONV, _lib.OP_POOL_BWD, _lib.OP_CONTENT_GRAD):
            acc = bool(o.flags & _lib.ACCUM)
            assert acc == (o.q0 in first_writer), "first write must store, later ones accumulate"
            first_writer[o.q0] = True


def test_schedule_handles_taps_on_relu_and_pool_outputs():
    layers = list(core_model.build_vgg_features(None, (4, 4, "M")).children())   # conv relu conv relu pool
    s = plan.Schedule(layers, [0, 2, 4], [1, 3], 16, 16, torch.float32, CPU, with_grad=True)
    assert [n.kind for n in s.nodes] == ["conv_first", "relu", "conv", "relu", "pool"]
    assert [t.buf is s.nodes[i].dst for t, i in zip(s.style_taps, (0, 2, 4))] == [True] * 3
    assert [t.buf is s.nodes[i].dst for t, i in zip(s.content_taps, (1, 3))] == [True] * 2
    with pytest.raises(RuntimeError, match="has a HIP kernel"):
        plan.Schedule([nn.Conv2d(3, 4, 5, padding=2)], [0], [], 8, 8, torch.float32, CPU, with_grad=False)


# ----------------------------------------------------------------------------------------------
# initialize_vgg: the pretrained-checkpoint branch (reference core_model.py:103-117).  The real
# vgg19-dcbb9e9d.pth cannot be fetched here, so a checkpoint with the same key layout
# (torchvision's `features.<idx>.weight/bias` + `classifier.*`) is fabricated in a temporary
# torch.hub directory; torchvision itself is absent, which is exactly the branch under test.
def _fake_checkpoint(path, seed=3):
    g = torch.Generator().manual_seed(seed)
    state = {}
    idx, cin = 0, 3
    for v in synthetic.VGG19_CFG:
        if v == "M":
            idx += 1
            continue
        state[f"features.{idx}.weight"] = torch.randn(int(v), cin, 3, 3, generator=g) * 0.05
        state[f"features.{idx}.bias"] = torch.randn(int(v), generator=g) * 0.1
        cin = int(v)
        idx += 2
    state["classifier.0.weight"] = torch.zeros(4, 4)          # present in the real file, must be ignored
    state["classifier.0.bias"] = torch.zeros(4)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save(state, path)
    return state


def test_initialize_vgg_loads_cached_checkpoint(tmp_path, monkeypatch, caplog):
    import logging
    import sys
    monkeypatch.delenv("STV_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setitem(sys.modules, "torchvision", None)         # import torchvision -> ImportError
    monkeypatch.setitem(sys.modules, "torchvision.models", None)
    old_hub = torch.hub.get_dir()
    torch.hub.set_dir(str(tmp_path / "hub"))
    try:
        ckpt = tmp_path / "hub" / "checkpoints" / "vgg19-dcbb9e9d.pth"
        # cache miss without torchvision: a clear error, not a download attempt
        with pytest.raises(RuntimeError, match="vgg19-dcbb9e9d.pth is absent"):
            core_model.initialize_vgg()
        state = _fake_checkpoint(ckpt)
        lg = logging.getLogger("style_transfer")
        lg.addHandler(caplog.handler)
        try:
            with caplog.at_level(logging.INFO, logger="style_transfer"):
                vgg = core_model.initialize_vgg()
        finally:
            lg.removeHandler(caplog.handler)
        assert any("Using cached VGG19 weights at" in r.getMessage() for r in caplog.records)
    finally:
        torch.hub.set_dir(old_hub)
    assert not vgg.training and all(not p.requires_grad for p in vgg.parameters())
    convs = [(i, l) for i, l in enumerate(vgg) if isinstance(l, nn.Conv2d)]
    assert len(convs) == 16 and len(vgg) == 37              # torchvision vgg19().features layout
    for i, conv in convs:
        assert torch.equal(conv.weight, state[f"features.{i}.weight"])
        assert torch.equal(conv.bias, state[f"features.{i}.bias"])
    # and it slices like the reference's default layers
    monkeypatch.setattr(core_model, "initialize_vgg", lambda: vgg)
    model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21])
    assert len(model.vgg_blocks) == 6 and model.style_ids == [0, 1, 2, 3, 5] and model.content_ids == [4]
    assert torch.equal(model.vgg_blocks[1][4].weight, state["features.5.weight"])


# ------------------------------------------------------------------ a6: same-seed `random` start image
def test_random_init_is_the_reference_draw_for_the_same_seed():
    """tests/golden/mini_random_lbfgs_nonorm.npz["x0"] is the start image the UNMODIFIED reference drew
    (core_model.py:66-100, ``randn_like`` on the CPU generator) after ``torch.manual_seed(0)`` and after its
    ``initialize_vgg`` seam had constructed the network (oracle/make_golden.py).  Seeding, constructing the same
    network through OUR builder and calling ``initialize_input`` must reproduce it bit for bit - nothing is copied in."""
    from tests.conftest import GoldenCase
    case = GoldenCase("mini_random_lbfgs_nonorm")
    assert case.meta["init_method"] == "random"
    content, _ = case.images()
    torch.manual_seed(0)
    core_model.build_vgg_features(case.weights(), case.cfg)          # the module constructions consume the generator
    x0 = core_model.initialize_input(content, "random")
    assert x0.requires_grad and torch.equal(x0.detach(), case.tensor("x0"))


def test_initialize_vgg_consumes_the_generator_like_torchvision_vgg19(monkeypatch):
    """What ``torchvision.models.vgg19(weights=...)`` constructs before the checkpoint is loaded over it
    (torchvision 0.24 models/vgg.py: ``VGG(make_layers(cfgs["E"]), init_weights=False)``): 16 Conv2d(k=3, pad=1)
    in layer order, then Linear(25088, 4096), Linear(4096, 4096), Linear(4096, 1000) - each with torch's default
    ``reset_parameters``.  The torchvision-free branches of ``initialize_vgg`` must leave the CPU generator in the
    same state, so that the ``random`` start image drawn afterwards is the one the reference's CPU path draws."""
    def torchvision_like_construction() -> None:
        cin = 3
        for v in synthetic.VGG19_CFG:
            if v != "M":
                nn.Conv2d(cin, int(v), kernel_size=3, padding=1)
                cin = int(v)
        nn.Linear(512 * 7 * 7, 4096), nn.Linear(4096, 4096), nn.Linear(4096, 1000)
    torch.manual_seed(1234)
    torchvision_like_construction()
    want = torch.randn(1, 3, 5, 7)
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "0")
    torch.manual_seed(1234)
    core_model.initialize_vgg()
    got = core_model.initialize_input(torch.zeros(1, 3, 5, 7), "random")
    assert torch.equal(got.detach(), want)


def test_initialize_vgg_cache_returns_the_same_stack_and_generator_state(monkeypatch):
    """The per-process cache of the torchvision-free branches: a second call from the same generator state returns
    an independent copy of the same stack and leaves the generator exactly where constructing again would have -
    the ``random`` start image drawn afterwards is the same; another generator state, or STV_VGG_CACHE=0, constructs."""
    monkeypatch.setenv("STV_SYNTHETIC_WEIGHTS", "7")
    core_model.clear_vgg_cache()
    built = []
    real_build = core_model.build_vgg_features
    monkeypatch.setattr(core_model, "build_vgg_features", lambda *a, **k: (built.append(1), real_build(*a, **k))[1])
    torch.manual_seed(99)
    first = core_model.initialize_vgg()
    want = torch.randn(3, 5)
    torch.manual_seed(99)
    second = core_model.initialize_vgg()
    got = torch.randn(3, 5)
    assert built == [1]                                           # the second call did not construct
    assert torch.equal(got, want)
    assert all(torch.equal(a, b) and a.data_ptr() != b.data_ptr() for a, b in zip(first.parameters(), second.parameters()))
    assert not second.training and all(not p.requires_grad for p in second.parameters())
    with torch.no_grad():
        next(second.parameters()).zero_()                         # a caller's copy is its own
    torch.manual_seed(99)
    third = core_model.initialize_vgg()
    assert torch.equal(next(third.parameters()), next(first.parameters()))
    torch.manual_seed(100)                                        # a generator state not seen before: constructs
    core_model.initialize_vgg()
    assert built == [1, 1]
    monkeypatch.setenv("STV_VGG_CACHE", "0")
    torch.manual_seed(99)
    core_model.initialize_vgg()
    assert built == [1, 1, 1] and torch.equal(torch.randn(3, 5), want)
    core_model.clear_vgg_cache()


def test_initialize_vgg_goes_through_the_module_level_constructor(monkeypatch, tmp_path):
    """Reference tests/test_core_model.py:225-245: ``core_model.vgg19`` / ``core_model.VGG19_Weights`` are
    module-level names a caller can replace; the stack comes from ``vgg19(weights=IMAGENET1K_V1).features`` and is
    returned frozen and in eval mode."""
    from urllib.parse import urlparse

    import torch
    from torch import nn

    from style_transfer_visualizer_amd import core_model
    monkeypatch.delenv("STV_SYNTHETIC_WEIGHTS", raising=False)
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path))
    calls = []

    class Fake(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.ReLU())

    def fake_vgg19(weights=None):
        calls.append(weights)
        return Fake()
    monkeypatch.setattr(core_model, "vgg19", fake_vgg19)
    out = core_model.initialize_vgg()
    assert calls == [core_model.VGG19_Weights.IMAGENET1K_V1]
    assert urlparse(core_model.VGG19_Weights.IMAGENET1K_V1.url).path.endswith("vgg19-dcbb9e9d.pth")
    assert not out.training and all(not p.requires_grad for p in out.parameters())


def _fake_vgg19(monkeypatch):
    from torch import nn

    from style_transfer_visualizer_amd import core_model

    class Fake(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 4, 3, padding=1), nn.ReLU())
    monkeypatch.setattr(core_model, "vgg19", lambda weights=None: Fake())
    return core_model


def test_download_notice_when_the_checkpoint_is_absent(monkeypatch, tmp_path, caplog):
    """reference tests/test_core_model.py:242-259: INFO "Downloading VGG19 weights to <hub>/checkpoints/<file>"."""
    import logging
    from pathlib import Path
    from urllib.parse import urlparse

    import torch
    monkeypatch.delenv("STV_SYNTHETIC_WEIGHTS", raising=False)
    core_model = _fake_vgg19(monkeypatch)
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path))
    want = tmp_path / "checkpoints" / Path(urlparse(core_model.VGG19_Weights.IMAGENET1K_V1.url).path).name
    caplog.set_level(logging.INFO, logger="style_transfer")
    core_model.initialize_vgg()
    assert any("Downloading VGG19 weights" in m and str(want) in m for m in caplog.messages)


def test_cache_notice_when_the_checkpoint_is_present(monkeypatch, tmp_path, caplog):
    """reference tests/test_core_model.py:261-281."""
    import logging
    from pathlib import Path
    from urllib.parse import urlparse

    import torch
    monkeypatch.delenv("STV_SYNTHETIC_WEIGHTS", raising=False)
    core_model = _fake_vgg19(monkeypatch)
    monkeypatch.setattr(torch.hub, "get_dir", lambda: str(tmp_path))
    cached = tmp_path / "checkpoints" / Path(urlparse(core_model.VGG19_Weights.IMAGENET1K_V1.url).path).name
    cached.parent.mkdir(parents=True)
    cached.touch()
    caplog.set_level(logging.INFO, logger="style_transfer")
    core_model.initialize_vgg()
    assert any("Using cached VGG19 weights" in m and str(cached) in m for m in caplog.messages)


@pytest.mark.parametrize("style_at,content_at", [([0, 5, 10, 19, 28], [21]), ([0, 5, 10, 19, 28], [7]), ([2, 7], [16, 25]), ([28], [0])])
def test_gradient_slabs_follow_the_reverse_schedule(style_at, content_at, monkeypatch):
    """plan.alloc_grads hands the gradients of the reverse chain a few rotating slabs (STV_GRAD_ARENA=2: also for host
    tensors).  Invariants, for several tap layouts (content tap in the middle, on a layer in front of a pool, on the very first
    conv; taps on conv / ReLU / pool outputs): buffers with a content tap keep a tensor of their own; two gradients one
    op reads and writes never share a slab; `backward_ops` itself re-checks op by op that every reader finds its writer's
    data (it raises otherwise); the per-node form gives the same op list."""
    def build(arena):
        monkeypatch.setenv("STV_GRAD_ARENA", arena)
        s = plan.Schedule(_layers(), style_at, content_at, 32, 32, torch.float32, CPU, with_grad=True)
        for tap in s.style_taps:
            tap.sgrad = torch.zeros(1, tap.buf.C, tap.buf.C)
        for tap in s.content_taps:
            tap.target = torch.zeros_like(tap.buf.act)
        ops_ = s.backward_ops(torch.zeros(1, 3, 32, 32), style_coef=1.0, content_coef=1.0, coef_dev=None)
        return s, ops_
    s, ops_a = build("2")
    s0, ops_0 = build("0")
    assert [(o.op, o.H, o.W, o.cin, o.cout, o.taps, o.flags) for o in ops_a] == [(o.op, o.H, o.W, o.cin, o.cout, o.taps, o.flags) for o in ops_0]
    slabs, slab_of = s._grad_slabs, s._grad_slab_of
    assert 1 <= slabs.shape[0] <= 3
    base = slabs.untyped_storage().data_ptr()
    own = {id(t.buf) for t in s.content_taps}
    for i, nd in enumerate(s.nodes):
        b = nd.dst
        assert b.grad.shape == b.act.shape and b.grad.dtype == b.act.dtype
        if id(b) in own:
            assert b.grad.untyped_storage().data_ptr() != base and id(b) not in slab_of
        else:
            assert b.grad.untyped_storage().data_ptr() == base
            assert b.grad.data_ptr() == slabs[slab_of[id(b)]].data_ptr()
        if i > 0 and id(b) in slab_of and id(nd.src) in slab_of:      # the op of node i reads b's gradient and writes src's
            assert slab_of[id(b)] != slab_of[id(nd.src)]
    per_node = sum(nd.dst.act.numel() * 4 for nd in s0.nodes)
    assert slabs.numel() < per_node
    # a second build of the reverse schedule (the autograd path builds its own program) reuses the assignment
    again = s.backward_ops(torch.zeros(1, 3, 32, 32), style_coef=1.0, content_coef=1.0, coef_dev=None)
    assert len(again) == len(ops_a) and s._grad_slabs is slabs

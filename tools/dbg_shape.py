import sys, torch
sys.path.insert(0, '.')
from tests.test_core_model_host import _fake_checkpoint
from style_transfer_visualizer_amd import core_model, synthetic
import pathlib, tempfile, os
sys.modules["torchvision"] = None; sys.modules["torchvision.models"] = None
tmp = pathlib.Path(tempfile.mkdtemp())
torch.hub.set_dir(str(tmp / "hub"))
state = _fake_checkpoint(tmp / "hub" / "checkpoints" / "vgg19-dcbb9e9d.pth")
DEV = torch.device("cuda")
model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21]).to(DEV)
content, style = (synthetic.synthetic_image(s, 64, 64) for s in (0, 1))
model.set_targets(style.to(DEV), content.to(DEV))
eng = next(iter(model._engines.values()))
print("taps", [(t.kind, t.buf.H, t.buf.W, t.buf.C, tuple(t.buf.act.shape)) for t in eng.sched.style_taps + eng.sched.content_taps])
print("public", [tuple(t.shape) for t in model.content_targets], [tuple(t.target.shape) for t in eng.sched.content_taps])
print("nodes", [(n.kind, n.layer, n.dst.H, n.dst.W, n.dst.C) for n in eng.sched.nodes])

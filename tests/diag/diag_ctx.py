"""Recompute every conv-produced gradient of a program from the program's own buffers, op by op (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from tests.conftest import GoldenCase
from style_transfer_visualizer_amd import core_model, ops
DEV = torch.device("cuda")
name = sys.argv[1]; nsteps = int(sys.argv[2]); compact = sys.argv[3] != "0"
case = GoldenCase(name); m = case.meta
weights = case.weights()
core_model.initialize_vgg = lambda: core_model.build_vgg_features(weights, case.cfg).eval()
model = core_model.StyleContentModel(m["style_layers"], m["content_layers"]).to(DEV)
content, style = case.images()
model.set_targets(style.to(DEV), content.to(DEV))
x = case.tensor("x0").to(DEV).requires_grad_(True)
state, work = ops.lbfgs_alloc(x.numel(), 100, DEV, compact=compact)
for step in range(nsteps):
    model.loss_and_grad(x, m["style_w"], m["content_w"])
    ops.lbfgs_step(x.detach(), x.grad, state, work, 100, min(step, 100), 1.0, compact=compact)
model.loss_and_grad(x, m["style_w"], m["content_w"])
torch.cuda.synchronize()
eng = next(iter(model._engines.values()))
nodes = eng.sched.nodes
wmap = {i: w for i, (w, b) in zip([i for i, c in enumerate(case.cfg)], weights)} if False else None
for k, nd in enumerate(nodes):
    if nd.kind != "conv":
        continue
    d, s = nd.dst, nd.src
    # torch reference of: s.grad = [mask] conv_transpose(d.grad) (+ tap terms on s)
    w = None
    dg = ops.from_nhwc(d.grad).cpu()
    wb = nd.wb
    if wb.dim() == 4:
        t, nck, co, ck = wb.shape
        wb_plain = wb.permute(0, 2, 1, 3).reshape(t, co, nck * ck)
    else:
        wb_plain = wb
    # wb_plain [9, cin_of_fwd(cout here), cout_of_fwd(cin here)] = flipped/transposed: run as forward conv on CPU
    co, ci = wb_plain.shape[1], wb_plain.shape[2]
    wt = wb_plain.float().cpu().reshape(3, 3, co, ci).permute(2, 3, 0, 1).contiguous()
    ref = F.conv2d(dg, wt, None, padding=1)
    mask_src = nd.relu_in or (s.relu_fused and not s.taps)
    if mask_src:
        ref = ref * (ops.from_nhwc(s.act).cpu() > 0).float()
    iso = ops.conv_igemm(d.grad, nd.wb, None, ref=s.act if mask_src else None, flags=ops.MASK if mask_src else 0)
    iso = ops.from_nhwc(iso).cpu()
    got = ops.from_nhwc(s.grad).cpu()
    extra = ""
    for tap in s.taps:
        if tap.kind == "style":
            f = s.act.float().cpu().reshape(-1, s.C)
            sg = tap.sgrad.float().cpu().reshape(s.C, s.C)
            term = (f @ sg.t()).reshape(s.H, s.W, s.C).permute(2, 0, 1).unsqueeze(0)
            iso1 = ops.from_nhwc(ops.conv_igemm(s.act, tap.sgrad, None)).cpu()
            extra += f" gram1x1 iso-err {float((iso1 - term).abs().max() / term.abs().max()):.1e}"
            ref = ref + term
            iso = iso + iso1
        else:
            extra += " (content tap: skipped)"
    sc = float(ref.abs().max())
    print(f"layer {nd.layer:2d} -> grad of layer {nodes[k-1].layer if k else -1}: program-vs-torch {float((got - ref).abs().max()) / sc:.2e}  "
          f"isolated-vs-torch {float((iso - ref).abs().max()) / sc:.2e}{extra}")

print("--- seeds vs fp64 recomputation from the program's own activations ---")
for tap in eng.sched.style_taps:
    b = tap.buf
    f = b.act.float().cpu().double().reshape(-1, b.C)
    n = f.shape[0]
    R = f.t() @ f
    norm = float(b.C * n)
    G = R.clamp(max=5e5) / norm
    T = tap.target.double().cpu().reshape(b.C, b.C)
    S = (R <= 5e5).double() * (G - T)
    sg = tap.sgrad.float().cpu().double().reshape(b.C, b.C)
    k = float((sg * S).sum() / (S * S).sum())            # common scale factor
    diff = (sg - k * S).abs()
    i = int(diff.flatten().argmax()); r, c = divmod(i, b.C)
    near = int(((R - 5e5).abs() < 5.0).sum())
    print(f"tap {tap.order}: C={b.C} n={n} scale {k:.4e} max|dS|/max|S| {float(diff.max() / (k * S).abs().max()):.2e} at ({r},{c}) R={float(R[r, c]):.6f} "
          f"sg={float(sg[r, c]):.4e} expect={float(k * S[r, c]):.4e}; elements with |R-5e5|<5: {near}; clamped: {int((R > 5e5).sum())}")

print("--- program weights vs the model's ---")
convs = [(i, l) for i, l in enumerate(eng.sched.layers if hasattr(eng.sched, "layers") else []) ]
wi = 0
for nd in nodes:
    if nd.kind not in ("conv", "conv_first"):
        continue
    w, b = weights[wi]; wi += 1
    if nd.kind == "conv_first":
        continue
    for label, got, exp in (("wf", nd.wf, ops.pack_weights_fwd(w)), ("wb", nd.wb, ops.pack_weights_bwd(w))):
        if got.dim() == 4:
            t, nck, co, ck = got.shape
            got = got.permute(0, 2, 1, 3).reshape(t, co, nck * ck)
        print(f"layer {nd.layer} {label}: max|diff| {float((got.float().cpu() - exp).abs().max()):.2e}", end="; ")
    print()

print("--- seeds and targets vs the fp64 oracle ---")
from oracle import core_model_ref as ocm
w64 = [(w.double(), b.double()) for w, b in weights]
prog64 = ocm.vgg_program(w64, case.cfg)
oracle64 = ocm.OracleModel(prog64, m["style_layers"], m["content_layers"])
oracle64.set_targets(style.double(), content.double())
h = x.detach().cpu().double()
feats = {}
for li in range(max(m["style_layers"]) + 1):
    h = ocm.run_layer(prog64[li], h)
    feats[li] = h
for tap in eng.sched.style_taps:
    b = tap.buf
    li = sorted(m["style_layers"])[tap.order]
    f = feats[li][0].reshape(b.C, -1).t()
    n = f.shape[0]
    R = f.t() @ f
    T = oracle64.style_targets[tap.order].reshape(b.C, b.C)
    Tp = tap.target.double().cpu().reshape(b.C, b.C)
    G = R.clamp(max=5e5) / float(b.C * n)
    S = (R <= 5e5).double() * (G - T)
    sg = tap.sgrad.float().cpu().double().reshape(b.C, b.C)
    k = float((sg * S).sum() / (S * S).sum())
    diff = (sg - k * S).abs()
    i = int(diff.flatten().argmax()); r, c = divmod(i, b.C)
    fp = b.act.float().cpu().double().reshape(-1, b.C)
    print(f"tap {tap.order} (layer {li}): target err {float((Tp - T).abs().max() / T.abs().max()):.2e}  act err {float((fp - f).abs().max() / f.abs().max()):.2e}  "
          f"seed err {float(diff.max() / (k * S).abs().max()):.2e} at ({r},{c}): R64={float(R[r,c]):.4f} G-T={float((G-T)[r,c]):.4e} max|G-T|={float((G-T).abs().max()):.4e}")

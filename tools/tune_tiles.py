"""Measure the conv tile table on this GPU and write style_transfer_visualizer_amd/conv_tiles_gfx950.json.

    python tools/tune_tiles.py [--rounds 3] [--sizes 256 512 1024] [--out PATH]

Builds the schedules of the VGG19 path at the given square image sizes (bf16 and fp32, forward + backward:
every 3x3 shape, the routed dgrads, the 1x1 Gram-backward products) with STV_CONV_TUNE=1, `rounds` times from an
empty table, and keeps per shape the tile most rounds agreed on (ties: the analytic choice if it is among them,
else the lowest index).  The result is what the library imports at load time (_lib._import_tile_table): every
later process then runs each shape on the same tile.
"""
import argparse, collections, datetime, json, os, sys
os.environ["STV_CONV_TUNE"] = "1"
os.environ.setdefault("STV_SYNTHETIC_WEIGHTS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from style_transfer_visualizer_amd import _lib, core_model, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--sizes", type=int, nargs="*", default=[256, 512, 1024])
ap.add_argument("--wide", type=int, nargs="*", default=[2160, 3840], help="one H W pair (the 4K config), bf16 only; empty to skip")
ap.add_argument("--out", default=os.path.join(os.path.dirname(_lib.LIB_PATH), "conv_tiles_gfx950.json"))
args = ap.parse_args()
dev = torch.device("cuda")
lib = _lib.load()
votes: dict = collections.defaultdict(collections.Counter)
shapes = [(s, s, p) for s in args.sizes for p in ("bf16", "fp32")]
if len(args.wide) == 2:
    shapes.append((args.wide[0], args.wide[1], "bf16"))
for rnd in range(args.rounds):
    lib.stv_conv_tune_import(None, 0)                      # empty table: everything is measured again
    for H, W, precision in shapes:
        model = core_model.StyleContentModel([0, 5, 10, 19, 28], [21], precision=precision).to(dev)
        content = synthetic.synthetic_image(0, H, W).to(dev)
        style = synthetic.synthetic_image(1, min(H, 1024), min(W, 1024)).to(dev)
        model.set_targets(style, content)
        x = torch.randn(1, 3, H, W, device=dev).requires_grad_(True)
        model.loss_and_grad(x, 1e5, 1.0)                   # builds the fused program: tunes the backward shapes too
        torch.cuda.synchronize()
        del model, x, content, style
        torch.cuda.empty_cache()
    for e in _lib.export_tile_table():
        votes[(e["H"], e["W"], e["cin"], e["cout"], e["taps"], e["elem_bytes"])][e["cfg"]] += 1
    print(f"round {rnd + 1}: {len(votes)} shapes", flush=True)
os.environ["STV_CONV_TUNE"] = "0"
entries = []
for key, cnt in sorted(votes.items()):
    H, W, cin, cout, taps, eb = key
    analytic = int(lib.stv_conv_config(H, W, cin, cout, taps, 1 if eb == 2 else 0))
    top = max(cnt.values())
    best = [c for c, v in cnt.items() if v == top]
    cfg = analytic if analytic in best else min(best)
    entries.append(dict(H=H, W=W, cin=cin, cout=cout, taps=taps, elem_bytes=eb, cfg=cfg, analytic=analytic,
                        votes={str(c): v for c, v in sorted(cnt.items())}))
_props = torch.cuda.get_device_properties(0)
# torch reports the generic marketing string ("AMD Radeon Graphics") on the MI355X boxes of this pool: record what does
# identify the device and the box - arch with its feature string, CU count, memory, UUID, host name
doc = {"tool": "tools/tune_tiles.py", "rounds": args.rounds,
       "device": f"MI355X ({torch.cuda.get_device_name(0)}; {_props.gcnArchName}, {_props.multi_processor_count} CUs, "
                 f"{_props.total_memory / 2**30:.0f} GiB)",
       "box": {"host": __import__("socket").gethostname(), "uuid": str(getattr(_props, "uuid", "")),
               "rocm": str(torch.version.hip)},
       "arch": str(_props.gcnArchName).split(":")[0],
       "date": datetime.date.today().isoformat(), "entries": entries}
with open(args.out, "w") as fh:
    json.dump(doc, fh, indent=1)
print(f"wrote {len(entries)} entries to {args.out}; differ from the analytic choice: {sum(e['cfg'] != e['analytic'] for e in entries)}")

"""Per-step table of the L-BFGS long-run comparison (diagnostic; prints e_dev / e_32 per step)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import optim_ref
import tests.test_gpu_lbfgs_long as T

n, history = 20000, int(sys.argv[1]) if len(sys.argv) > 1 else 100
grad = T._objective(n)
devs = {"compact": T._Device(n, history, True), "twoloop": T._Device(n, history, False)}
x32 = torch.zeros(n); x64 = torch.zeros(n, dtype=torch.float64)
t32 = optim_ref.LbfgsRef(x32, history_size=history); t64 = optim_ref.LbfgsRef(x64, history_size=history)
zero = torch.tensor(0.0)
master = devs["compact"]
for step in range(1, history + 41):
    xb = {k: d.image() for k, d in devs.items()}
    g = grad(xb["compact"])
    b32, b64 = x32.clone(), x64.clone()
    t32.step(lambda: (zero, g.clone())); t64.step(lambda: (zero.double(), g.double()))
    for d in devs.values():
        d.step(g)
    u64 = x64 - b64; sc = float(u64.abs().max())
    e32 = float(((x32 - b32).double() - u64).abs().max()) / sc
    row = [f"{step:4d} m={len(t32.old_dirs):3d} e32 {e32:.2e}"]
    for k, d in devs.items():
        u = d.image().double() - xb[k].double()
        row.append(f"{k} {float((u - u64).abs().max()) / sc:.2e}")
    print("  ".join(row), flush=True)

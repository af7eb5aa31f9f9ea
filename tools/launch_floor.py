"""Fixed cost of one dependent kernel inside the captured graph: N tiny relu ops replayed (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import _lib, plan
dev = torch.device("cuda")
a = torch.zeros(1024, device=dev, dtype=torch.bfloat16); b = torch.zeros_like(a)
for n_ops in (1, 16, 64):
    ops_ = []
    for i in range(n_ops):
        o = _lib.StvOp(); o.op = _lib.OP_RELU_FWD; o.dtype = _lib.STV_BF16
        o.p0 = (a if i % 2 == 0 else b).data_ptr(); o.q0 = (b if i % 2 == 0 else a).data_ptr(); o.n = 1024
        ops_.append(o)
    prog = plan.Program(ops_, [a, b])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5): prog.run(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): prog.run(True)
        e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 200 * 1000
    print(f"{n_ops:3d} tiny kernels per graph: {t:8.1f} us per replay = {t / n_ops:6.2f} us per kernel")

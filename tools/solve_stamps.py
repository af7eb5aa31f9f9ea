"""Phase times of the L-BFGS solve kernel at full history (diagnostic; needs a -DSTV_SOLVE_STAMPS build).
  make -C style_transfer_visualizer_amd/csrc clean all CXXFLAGS+=-DSTV_SOLVE_STAMPS   (or build a side library)
  STV_LIB_PATH=tools/libstv_stamps.so python tools/solve_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import _lib, ops
dev = torch.device("cuda")
n, hist = 3 * 512 * 512, 100
a = 10.0 ** (torch.rand(n, device=dev) * 4 - 2)      # convex, badly conditioned: y.s > 0 every step
x = torch.randn(n, device=dev)
state, work = ops.lbfgs_alloc(n, hist, dev, compact=True)
lib = _lib.load()
names = ["header", "install+gdots", "table fill", "loop1 (alpha)", "loop2 (YY)", "loop3 (beta)", "tail"]
for k in range(140):
    g = a * x
    ops.lbfgs_step(x, g, state, work, hist, hist, 1.0, compact=True)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
lib.stv_debug_solve_stamps.argtypes = [ctypes.c_void_p]
assert lib.stv_debug_solve_stamps(out) == 0
t = list(out)
hdr = state.cpu().numpy().view("int32")[:8]
print("n_iter, hist_len:", hdr[0], hdr[1])
for i, nm in enumerate(names):
    print(f"{nm:16s} {(t[i + 1] - t[i]) / 100:7.2f} us")
print(f"total            {(t[7] - t[0]) / 100:7.2f} us")

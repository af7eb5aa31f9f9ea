"""Phase times of the bf16 Gram partial kernel (diagnostic; needs a -DSTV_GRAM_STAMPS build, STV_LIB_PATH)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from style_transfer_visualizer_amd import _lib, ops
dev = torch.device("cuda")
lib = _lib.load()
lib.stv_debug_gram_stamps.argtypes = [ctypes.c_void_p]
for C, N in ((512, 16384), (256, 65536), (128, 262144), (64, 1048576)):
    F = torch.randn(N, C, device=dev).bfloat16()
    parts = ops.gram_partial(F)
    for _ in range(5): ops.gram_partial(F, parts)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    assert lib.stv_debug_gram_stamps(out) == 0
    t = [v / 100 for v in out]
    n = int(out[6])
    print(f"C={C} N={N} stages={n}: prologue {t[4]:.2f} us | per stage: mfma {t[0]/n:.3f} loadwait {t[1]/n:.3f} "
          f"ldswrite {t[2]/n:.3f} barrier {t[3]/n:.3f} us | epilogue {t[5]:.2f} us")

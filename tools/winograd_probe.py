#!/usr/bin/env python3
"""A/B: Winograd F(2x2,3x3) against the direct implicit-GEMM kernel on one Cin >= 256 layer (bf16 mode).

    python tools/winograd_probe.py [--hw 256] [--cin 256] [--cout 256] [--json gpurun_out/wino.json]

Winograd needs 16 channel contractions of [tiles][Cin] x [Cin][Cout] (tiles = H*W/4) instead of 9
full-resolution ones: 2.25x fewer MACs.  This prototype runs the three stages as separate launches -
input transform (tools/winograd/wino.hip), 16 x the product library's 1x1 matrix-core kernel, output
transform - and reports, next to the direct kernel: time per stage, TFLOP/s-equivalent (direct-conv
FLOPs / time), and the error of each variant against the CPU oracle (F.conv2d in fp32 on the same
bf16-rounded operands; the direct kernel's own error is the yardstick):

  * "wino bf16 M": the pipeline as launched (products M stored in bf16 between GEMM and output transform);
  * "wino fp32 M": what a FUSED kernel would compute - V and U rounded to bf16 (MFMA operands), products
    accumulated and transformed back in fp32 (emulated with torch on the GPU; arithmetic only, no timing).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from style_transfer_visualizer_amd import ops  # noqa: E402

DEV = "cuda"
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # microseconds


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--hw", type=int, default=256)
    ap.add_argument("--cin", type=int, default=256)
    ap.add_argument("--cout", type=int, default=256)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "winograd", "libstv_wino.so"))
    H = W = a.hw
    C, N = a.cin, a.cout
    g = torch.Generator().manual_seed(0)
    x = F.relu(torch.randn(1, C, H, W, generator=g) * 0.8 + 0.1)               # post-ReLU activations
    w = (torch.rand(N, C, 3, 3, generator=g) * 2 - 1) * (6.0 / (9 * C)) ** 0.5   # He-scaled, as the synthetic VGG19
    b = torch.zeros(N)
    xq, wq = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(xq, wq, b, padding=1)                                         # CPU oracle, fp32
    scale = float(ref.abs().max())

    def err(y_nhwc: torch.Tensor) -> tuple[float, float]:
        d = ops.from_nhwc(y_nhwc).cpu() - ref
        return float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / float(ref.pow(2).mean().sqrt())
    flops = 2.0 * 9 * C * N * H * W
    res = {"layer": f"{H}x{W} {C}->{N} bf16", "direct_gflop": flops / 1e9, "winograd_gflop": flops / 2.25 / 1e9}

    # ---- direct kernel -----------------------------------------------------------------------
    xd = ops.to_nhwc(x, torch.bfloat16).to(DEV)
    wp = ops.block_weights(ops.pack_weights_fwd(w).bfloat16().to(DEV))
    bd = b.to(DEV)
    ops.conv_tune(H, W, C, N, 9, torch.bfloat16)
    y = torch.empty(H, W, N, device=DEV, dtype=torch.bfloat16)
    t_direct = timeit(lambda: ops.conv_igemm(xd, wp, bd, out=y))
    e_direct = err(y)
    # the oracle rounded to bf16: the floor any bf16-output kernel sits on
    e_floor = err(ops.to_nhwc(ref, torch.bfloat16))
    res["direct"] = {"us": t_direct, "tflops": flops / t_direct / 1e6, "max_err": e_direct[0], "rms_err": e_direct[1]}
    res["bf16_rounding_floor"] = {"max_err": e_floor[0], "rms_err": e_floor[1]}

    # ---- Winograd, three launches (+15) -----------------------------------------------------------
    U = torch.einsum("ij,ncjk,lk->ilnc", G, w, G)                                # [4][4][N][C] = G g G^T
    Uq = U.bfloat16()
    Ud = [ops.block_weights(Uq[i, j].reshape(1, N, C).contiguous().to(DEV)) for i in range(4) for j in range(4)]
    Th, Tw = H // 2, W // 2
    V = torch.empty(16, Th, Tw, C, device=DEV, dtype=torch.bfloat16)
    M = torch.empty(16, Th, Tw, N, device=DEV, dtype=torch.bfloat16)
    yw = torch.empty(H, W, N, device=DEV, dtype=torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream
    ops.conv_tune(Th, Tw, C, N, 1, torch.bfloat16)

    def t_in():
        assert lib.wino_input_transform(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(V.data_ptr()), H, W, C, 0, ctypes.c_void_p(st)) == 0

    def t_gemm():
        for k in range(16):
            ops.conv_igemm(V[k], Ud[k], None, out=M[k])

    def t_out():
        assert lib.wino_output_transform(ctypes.c_void_p(M.data_ptr()), 0, ctypes.c_void_p(bd.data_ptr()), ctypes.c_void_p(yw.data_ptr()),
                                         H, W, N, 0, ctypes.c_void_p(st)) == 0
    us_in, us_gemm, us_out = timeit(t_in), timeit(t_gemm), timeit(t_out)
    t_in(); t_gemm(); t_out()
    e_w16 = err(yw)
    total = us_in + us_gemm + us_out
    mb = lambda t: t.numel() * t.element_size() / 1e6                              # noqa: E731
    res["winograd_3_stage"] = {
        "us": {"input_transform": us_in, "gemm_16_launches": us_gemm, "output_transform": us_out, "total": total},
        "tflops_equivalent": flops / total / 1e6, "gemm_tflops": flops / 2.25 / us_gemm / 1e6,
        "hbm_MB": {"input_transform": mb(xd) + mb(V), "gemm": mb(V) + mb(M), "output_transform": mb(M) + mb(yw)},
        "max_err": e_w16[0], "rms_err": e_w16[1]}

    # ---- what a fused kernel would compute: bf16 V and U, fp32 M, fp32 output transform ------------
    Vf = V.float()                                                                   # V as the MFMA would read it
    Mf = torch.einsum("kyxc,knc->kyxn", Vf, torch.stack([Uq[i, j] for i in range(4) for j in range(4)]).float().to(DEV))
    yf = torch.empty(H, W, N, device=DEV, dtype=torch.bfloat16)
    assert lib.wino_output_transform(ctypes.c_void_p(Mf.contiguous().data_ptr()), 1, ctypes.c_void_p(bd.data_ptr()),
                                     ctypes.c_void_p(yf.data_ptr()), H, W, N, 0, ctypes.c_void_p(st)) == 0
    torch.cuda.synchronize()
    e_w32 = err(yf)
    res["winograd_fused_arithmetic"] = {"max_err": e_w32[0], "rms_err": e_w32[1]}
    print(json.dumps(res, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

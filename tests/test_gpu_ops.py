"""Per-kernel parity on a real MI355X: every C-ABI entry point vs the CPU oracle.

fp32 storage: results must agree with the torch-CPU fp32 oracle to rounding of
the accumulation order (tolerances below are relative to the tensor scale).
bf16 storage: the oracle is evaluated in fp32 on the *bf16-rounded* operands, so
only accumulation order and the final bf16 rounding (2^-8 relative) differ.
"""
from __future__ import annotations

import functools

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import core_model_ref as ocm
from oracle import optim_ref
from style_transfer_visualizer_amd import ops, synthetic

pytestmark = pytest.mark.gpu

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(shape, seed, lo=-1.0, hi=1.0):
    n = int(np.prod(shape))
    u = synthetic.hash_uniform(seed, 77, n).reshape(shape)
    return torch.from_numpy((u * (hi - lo) + lo).astype(np.float32))


def q(t, dtype):
    """Round to the storage dtype and come back to fp32 (oracle operand)."""
    return t.to(dtype).float()


def tol(dtype, k_terms=1):
    if dtype == torch.float32:
        return 2e-6 * max(1.0, k_terms ** 0.5)
    return 6e-3


def assert_close(got, ref, dtype, k_terms=1, what=""):
    got = got.float().cpu()
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max()) / scale
    assert err <= tol(dtype, k_terms), f"{what}: rel-to-scale err {err:.3e} > {tol(dtype, k_terms):.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("hw", [(40, 72), (5, 7), (64, 64), (33, 100)])
def test_conv_first_fwd_and_dgrad(dtype, hw, packed):
    """packed=True: weights prepared once (stv_conv_first_pack); in bf16 the forward then runs on the
    matrix cores with two-term bf16 splits of image and weights - same tolerance as the VALU kernel."""
    H, W = hw
    cout = 64
    x = rnd((1, 3, H, W), 1, -2, 2)
    w = rnd((cout, 3, 3, 3), 2, -0.5, 0.5)
    b = rnd((cout,), 3, -0.1, 0.1)
    ref = F.conv2d(x, w, b, padding=1)
    wf = ops.pack_weights_fwd(w).to(DEV)
    pk = ops.conv_first_pack(wf) if packed else None
    y = ops.conv_first_fwd(x.to(DEV), wf, b.to(DEV), dtype, packed=pk)
    assert y.shape == (H, W, cout)
    assert_close(ops.from_nhwc(y), ref, dtype, 27, "conv_first_fwd")
    if packed and dtype == torch.bfloat16:
        # the split product is fp32-faithful: what is left is only the bf16 rounding of the output
        # (two nearly equal values on either side of a rounding boundary differ by one ulp = 2^-7)
        d = (ops.from_nhwc(y).cpu() - ref.bfloat16().float()).abs() / ref.abs().max()
        assert float(d.max()) <= 2 ** -7, f"beyond one bf16 ulp of the output scale: {float(d.max()):.2e}"
        assert float((d > 0).float().mean()) < 0.02, "more than 2 % of the outputs round differently"
    # dgrad: dx = conv_transpose(dy, w)
    dy = rnd((1, cout, H, W), 4)
    dyq = q(dy, dtype)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, padding=1).backward(dyq)
    dx = ops.conv_first_dgrad(ops.to_nhwc(dy, dtype).to(DEV), wf, 3, packed=pk)
    assert_close(dx, xr.grad, torch.float32, 9 * cout, "conv_first_dgrad")


@pytest.mark.parametrize("hw", [(64, 64), (75, 101), (13, 7), (8, 32), (256, 320), (512, 512)])
def test_conv_first_fwd_leaves_gram_slabs(hw):
    """stv_conv_first_fwd_gram: the tapped first layer also writes the split-K slabs of F^T F that
    stv_gram_partial computes from the stored map - same map bit for bit, same Gram matrix up to the
    fp32 summation order (ragged tiles, single-tile images, idle workgroups that must write zero slabs)."""
    H, W = hw
    x = rnd((1, 3, H, W), 301, -2, 2)
    w = rnd((64, 3, 3, 3), 302, -0.5, 0.5)
    b = rnd((64,), 303, -0.1, 0.1)
    wf = ops.pack_weights_fwd(w).to(DEV)
    pk = ops.conv_first_pack(wf)
    assert ops.conv_first_gram_supported(H, W, 3, 64, torch.bfloat16)
    y_plain = ops.conv_first_fwd(x.to(DEV), wf, b.to(DEV), torch.bfloat16, packed=pk)
    slabs = torch.full((ops.gram_ksplit(H * W, 64), 64, 64), float("nan"), device=DEV)
    y = ops.conv_first_fwd(x.to(DEV), wf, b.to(DEV), torch.bfloat16, packed=pk, gram_partials=slabs)
    assert torch.equal(y, y_plain)
    assert bool(torch.isfinite(slabs).all())
    got = slabs.double().sum(0).cpu()
    f = y.reshape(H * W, 64).double().cpu()
    want = f.t() @ f                                    # Gram of the STORED values, in float64
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-6 * scale, f"{float((got - want).abs().max()) / scale:.2e}"
    assert torch.equal(got, got.t())                    # the lower block is written as the mirror image
    # and through the finish pass it is the Gram matrix the separate chain produces
    g_fused = torch.empty(64, 64, device=DEV)
    ops.gram_finish(slabs, H * W, 64, gram_out=g_fused, clamp_max=5e5, norm=float(64 * H * W))
    g_chain = torch.empty(64, 64, device=DEV)
    ops.gram_finish(ops.gram_partial(y), H * W, 64, gram_out=g_chain, clamp_max=5e5, norm=float(64 * H * W))
    assert float((g_fused - g_chain).abs().max()) <= 2e-6 * float(g_chain.abs().max())
    with pytest.raises(ValueError):
        ops.conv_first_fwd(x.to(DEV), wf, b.to(DEV), torch.bfloat16, gram_partials=slabs)      # needs packed weights


CONV_CASES = [
    # cin, cout, H, W
    (64, 64, 33, 70),
    (64, 128, 16, 40),
    (128, 128, 9, 33),
    (256, 512, 8, 8),
    (512, 512, 4, 4),
    (16, 8, 12, 20),
    (8, 16, 12, 20),
    (128, 256, 2, 3),
]


def _maybe_blocked(wp, blocked, H, W, cin, cout, dtype):
    """K-blocked weights (STV_W_BLOCKED) wherever the shape runs on the matrix cores."""
    if blocked and ops.conv_uses_mfma(H, W, cin, cout, dtype):
        return ops.block_weights(wp)
    if blocked:
        pytest.skip("direct-kernel shape: plain weights only")
    return wp


def test_blocked_weights_rejected_on_direct_shapes():
    x = torch.zeros(4, 4, 12, device=DEV)
    w = torch.zeros(9, 1, 8, 8, device=DEV)          # claims cin=8 (x has 12): host check
    with pytest.raises(RuntimeError):
        ops.conv_igemm(x, w)
    x = torch.zeros(4, 4, 8, device=DEV)
    w = torch.zeros(9, 1, 6, 8, device=DEV)          # cout=6 is not vectorisable -> direct kernel -> STV_ERR_ARG
    with pytest.raises(RuntimeError, match="STV_ERR_ARG"):
        ops.conv_igemm(x, w)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("blocked", [False, True])
@pytest.mark.parametrize("flags", [0, ops.RELU_IN, ops.RELU_OUT, ops.RELU_IN | ops.RELU_OUT])
def test_conv_igemm_forward(dtype, case, flags, blocked):
    cin, cout, H, W = case
    x = rnd((1, cin, H, W), 11)
    w = rnd((cout, cin, 3, 3), 12, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    b = rnd((cout,), 13, -0.2, 0.2)
    xq, wq = q(x, dtype), q(w, dtype)
    xin = F.relu(xq) if flags & ops.RELU_IN else xq
    ref = F.conv2d(xin, wq, b, padding=1)
    if flags & ops.RELU_OUT:
        ref = F.relu(ref)
    wp = _maybe_blocked(ops.pack_weights_fwd(w).to(dtype).to(DEV), blocked, H, W, cin, cout, dtype)
    y = ops.conv_igemm(ops.to_nhwc(x, dtype).to(DEV), wp, b.to(DEV), flags=flags)
    assert_close(ops.from_nhwc(y), ref, dtype, 9 * cin, f"conv fwd {case} flags={flags}")


@functools.lru_cache(maxsize=None)
def _dual_case(dtype, flags, case):
    """Operands (on the device) and the CPU reference of one dual-launch case: the same for every tile configuration
    the test forces (20 of them), so they are built once - the CPU convolution was most of the test's time."""
    cd, cs, H, W = case
    w = rnd((cd, cs, 3, 3), 71, -1, 1) * (2.0 / (9 * cd)) ** 0.5
    dy = rnd((1, cd, H, W), 72)
    z = rnd((1, cs, H, W), 73)                      # this layer's stored pre-ReLU output = F
    s_mat = rnd((cs, cs), 74, -0.02, 0.02)
    s_mat = (s_mat + s_mat.t()) * 0.5
    prev = rnd((1, cs, H, W), 75)
    wq, dyq, zq, sq, pq = q(w, dtype), q(dy, dtype), q(z, dtype), q(s_mat, dtype), q(prev, dtype)
    xr = torch.zeros(1, cs, H, W, requires_grad=True)
    F.conv2d(xr, wq, None, padding=1).backward(dyq)
    first = xr.grad * ((zq > 0).float() if flags & ops.MASK else 1.0)
    second = torch.einsum("bchw,nc->bnhw", zq, sq)
    ref = first + second + (pq if flags & ops.ACCUM else 0.0)
    dev = {"prev": ops.to_nhwc(prev, dtype).to(DEV), "wb": ops.block_weights(ops.pack_weights_bwd(w).to(dtype).to(DEV)),
           "z": ops.to_nhwc(z, dtype).to(DEV), "dy": ops.to_nhwc(dy, dtype).to(DEV), "s": sq.to(dtype).to(DEV).contiguous()}
    return ref, dev


# the four flag combinations on the tile the library picks; a FORCED tile runs the two extremes (no flag, both flags) - the
# epilogue is the same template code for every tile, the full cross product was 800 launches of it
_DUAL_CFG_FLAGS = ([(None, f) for f in (0, ops.MASK, ops.ACCUM, ops.MASK | ops.ACCUM)]
                   + [(c, f) for c in range(19) for f in (0, ops.MASK | ops.ACCUM)])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg,flags", _DUAL_CFG_FLAGS)
@pytest.mark.parametrize("case", [(128, 64, 33, 70), (256, 128, 16, 40), (64, 64, 9, 33), (512, 256, 8, 8),
                                  (96, 48, 20, 36)])   # 48 channels: not whole multi-slice stages -> general 1x1 loop
def test_conv_igemm_dual_dgrad_plus_gram_term(dtype, cfg, flags, case, monkeypatch):
    """out = [prev +] mask(z>0) * dgrad(dy, w) + F . S^T in one launch: the mask touches the first term only."""
    cd, cs, H, W = case            # channels of the layer above (dy) and of this layer (output, F, S)
    if cfg is not None:
        if cs <= 64 and cfg in (0, 2, 18):
            pytest.skip("128-channel tiles need more than 64 output channels")
        monkeypatch.setenv("STV_CONV_CFG", str(cfg))
    ref, dev = _dual_case(dtype, flags, case)
    out = dev["prev"].clone()
    ops.conv_igemm_dual(dev["dy"], dev["wb"], dev["z"], dev["s"], ref=dev["z"] if flags & ops.MASK else None, out=out, flags=flags)
    assert_close(ops.from_nhwc(out), ref, dtype, 9 * cd + cs, f"dual {case} flags={flags}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [None, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
@pytest.mark.parametrize("case", [(64, 64, 33, 70), (128, 128, 16, 40), (64, 128, 9, 33), (256, 256, 8, 8)])
def test_conv_igemm_with_fused_maxpool(dtype, cfg, case, monkeypatch):
    """One launch writes relu(conv) and MaxPool2d(2,2) of it (odd sizes drop the last row / column like torch)."""
    cin, cout, H, W = case
    if cfg is not None:
        if cout <= 64 and cfg in (0, 2, 18):
            pytest.skip("128-channel tiles need more than 64 output channels")
        monkeypatch.setenv("STV_CONV_CFG", str(cfg))
    x = rnd((1, cin, H, W), 51)
    w = rnd((cout, cin, 3, 3), 52, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    b = rnd((cout,), 53, -0.2, 0.2)
    ref = F.relu(F.conv2d(q(x, dtype), q(w, dtype), b, padding=1))
    wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
    idx = torch.full((H // 2, W // 2, cout), 255, device=DEV, dtype=torch.uint8)
    y, yp = ops.conv_igemm_pool(ops.to_nhwc(x, dtype).to(DEV), wp, b.to(DEV), flags=ops.RELU_OUT, pool_idx=idx)
    assert_close(ops.from_nhwc(y), ref, dtype, 9 * cin, f"conv {case}")
    assert yp.shape == (H // 2, W // 2, cout)
    # the pooled map is exactly the pool of the stored map (same rounding, max commutes with it)
    want = F.max_pool2d(ops.from_nhwc(y).cpu(), 2, 2)
    assert torch.equal(ops.from_nhwc(yp).cpu(), want)
    # the arg-max map is the decision a pooling pass over the stored map takes (first maximum in scan
    # order, winner > 0): routing the gradient through it equals the activation-based backward, bit
    # for bit, with and without the ReLU mask / accumulate
    assert int(idx.max()) <= 7
    dy = ops.to_nhwc(rnd((1, cout, H // 2, W // 2), 54), dtype).to(DEV)
    for flags in (0, ops.MASK, ops.MASK | ops.ACCUM):
        seed = ops.to_nhwc(rnd((1, cout, H, W), 55), dtype).to(DEV)
        a, b2 = seed.clone(), seed.clone()
        ops.maxpool_bwd(y, dy, out=a, flags=flags)
        ops.maxpool_bwd_idx(idx, dy, H, W, out=b2, flags=flags)
        assert torch.equal(a, b2), f"idx backward differs, flags={flags}"


@pytest.mark.parametrize("case", [(64, 64, 75, 101), (64, 128, 40, 72), (64, 64, 8, 32), (64, 64, 13, 7), (64, 64, 512, 512), (64, 128, 256, 320),
                                  (128, 128, 75, 101), (128, 128, 2, 32), (128, 128, 13, 7), (128, 128, 3, 40), (128, 128, 256, 320),
                                  (128, 128, 512, 512)])
def test_conv_ws_forward_pool_and_backward(case, monkeypatch):
    """The weight-stationary persistent kernel (csrc/conv_ws.hip; bf16; Cin = 64: 8 x 32-pixel tiles, 2 x 2 waves;
    Cin = 128, 128 -> 128: 2 x 32-pixel tiles, four waves along the output channels): forward with bias /
    ReLU-on-load / ReLU / fused max-pool + arg-max map (also without the full-resolution store: STV_POOL_ONLY), and
    the backward form mask(z>0)*dgrad + z.S^T, on ragged images, on single-tile images and on images with several
    tiles per workgroup."""
    cin, cout, H, W = case
    dtype = torch.bfloat16
    monkeypatch.setenv("STV_CONV_WS", "2")          # every supported shape, also those the default leaves to the general kernel
    monkeypatch.setenv("STV_CONV_WS128", "2")       # (default: only the backward form of large images, where it measured faster)
    assert ops.conv_uses_ws(H, W, cin, cout, dtype, flags=ops.RELU_IN | ops.RELU_OUT | ops.W_BLOCKED)
    x = rnd((1, cin, H, W), 141)
    w = rnd((cout, cin, 3, 3), 142, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    b = rnd((cout,), 143, -0.2, 0.2)
    xq, wq = q(x, dtype), q(w, dtype)
    for blocked in (True, False):
        wp = ops.pack_weights_fwd(w).to(dtype).to(DEV)
        if blocked:
            wp = ops.block_weights(wp)
        for flags in (0, ops.RELU_IN | ops.RELU_OUT):
            ref = F.conv2d(F.relu(xq) if flags & ops.RELU_IN else xq, wq, b, padding=1)
            ref = F.relu(ref) if flags & ops.RELU_OUT else ref
            y = ops.conv_igemm(ops.to_nhwc(x, dtype).to(DEV), wp, b.to(DEV), flags=flags)
            assert_close(ops.from_nhwc(y), ref, dtype, 9 * cin, f"ws fwd {case} flags={flags} blocked={blocked}")
    # fused pool + arg-max map
    if H >= 2 and W >= 2:
        assert ops.conv_uses_ws(H, W, cin, cout, dtype, flags=ops.RELU_OUT | ops.W_BLOCKED, has_pool=True)
        wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
        for relu_in in (0, ops.RELU_IN):
            idx = torch.full((H // 2, W // 2, cout), 255, device=DEV, dtype=torch.uint8)
            y, yp = ops.conv_igemm_pool(ops.to_nhwc(x, dtype).to(DEV), wp, b.to(DEV), flags=ops.RELU_OUT | relu_in, pool_idx=idx)
            ref = F.relu(F.conv2d(F.relu(xq) if relu_in else xq, wq, b, padding=1))
            assert_close(ops.from_nhwc(y), ref, dtype, 9 * cin, f"ws conv+pool {case} relu_in={relu_in}")
            assert torch.equal(ops.from_nhwc(yp).cpu(), F.max_pool2d(ops.from_nhwc(y).cpu(), 2, 2))
            assert int(idx.max()) <= 7
            # STV_POOL_ONLY: the full map is not written, the pooled map and the arg-max map are the same bits
            idx2 = torch.full_like(idx, 255)
            y2 = torch.full_like(y, 7.0)
            _, yp2 = ops.conv_igemm_pool(ops.to_nhwc(x, dtype).to(DEV), wp, b.to(DEV), flags=ops.RELU_OUT | relu_in | ops.POOL_ONLY,
                                         out=y2, pool_idx=idx2)
            assert torch.equal(yp2, yp) and torch.equal(idx2, idx) and bool((y2 == 7.0).all())
        dyp = ops.to_nhwc(rnd((1, cout, H // 2, W // 2), 144), dtype).to(DEV)
        for flags in (0, ops.MASK):
            a_, b_ = torch.zeros_like(y), torch.zeros_like(y)
            ops.maxpool_bwd(y, dyp, out=a_, flags=flags)
            ops.maxpool_bwd_idx(idx, dyp, H, W, out=b_, flags=flags)
            assert torch.equal(a_, b_), f"ws arg-max map differs from the activation-based routing, flags={flags}"
    # backward form (Cout = Cin only): mask and / or the fused Gram term
    if cout == cin:
        dy = rnd((1, cin, H, W), 145)
        z = rnd((1, cin, H, W), 146)
        s_mat = rnd((cin, cin), 147, -0.02, 0.02)
        s_mat = (s_mat + s_mat.t()) * 0.5
        dyq, zq, sq = q(dy, dtype), q(z, dtype), q(s_mat, dtype)
        xr = torch.zeros(1, cin, H, W, requires_grad=True)
        F.conv2d(xr, wq, None, padding=1).backward(dyq)
        wb = ops.block_weights(ops.pack_weights_bwd(w).to(dtype).to(DEV))
        zn = ops.to_nhwc(z, dtype).to(DEV)
        second = torch.einsum("bchw,nc->bnhw", zq, sq)
        for mask in (False, True):
            assert ops.conv_uses_ws(H, W, cin, cin, dtype, flags=(ops.MASK if mask else 0) | ops.W_BLOCKED, has_ref=True)
            first = xr.grad * ((zq > 0).float() if mask else 1.0)
            out = ops.conv_igemm_dual(ops.to_nhwc(dy, dtype).to(DEV), wb, zn, sq.to(dtype).to(DEV).contiguous(),
                                      ref=zn if mask else None, flags=ops.MASK if mask else 0)
            assert_close(ops.from_nhwc(out), first + second, dtype, 9 * cin + cin, f"ws dual {case} mask={mask}")
        out = ops.conv_igemm(ops.to_nhwc(dy, dtype).to(DEV), wb, None, ref=zn, flags=ops.MASK)
        assert_close(ops.from_nhwc(out), xr.grad * (zq > 0).float(), dtype, 9 * cin, f"ws masked dgrad {case}")
        # the general kernel on the same launch: the two agree to rounding of a few sums (another summation order)
        monkeypatch.setenv("STV_CONV_WS", "0")
        assert not ops.conv_uses_ws(H, W, cin, cin, dtype, flags=ops.MASK | ops.W_BLOCKED, has_ref=True)
        out_gen = ops.conv_igemm(ops.to_nhwc(dy, dtype).to(DEV), wb, None, ref=zn, flags=ops.MASK)
        d = (out.float() - out_gen.float()).abs()
        assert float((d / (2.0 ** -7 * torch.maximum(out.float().abs(), out_gen.float().abs()) + 1e-6)).max()) <= 1.0


@pytest.mark.parametrize("case,skew", [((64, 75, 101), "1"), ((64, 75, 101), "0"), ((128, 40, 72), "1"), ((64, 8, 32), "1"), ((64, 13, 7), "0"),
                                       ((64, 4, 33), "1"), ((64, 512, 512), "1"), ((128, 256, 320), "0")])
def test_conv_ws2_forward_and_pool(case, skew, monkeypatch):
    """The two-waves-per-SIMD form of the weight-stationary kernel (csrc/conv_ws2.hip, STV_CONV_WS2; Cin = 64, forward
    forms; K split between the two waves of a SIMD, partial sums exchanged through LDS): bias / ReLU-on-load / ReLU /
    fused max-pool + arg-max map / STV_POOL_ONLY against the CPU convolution, on ragged, single-tile and many-tile images,
    with the two wave classes finishing their tile at the same time and half a tile apart (STV_WS2_SKEW)."""
    cout, H, W = case
    cin, dtype = 64, torch.bfloat16
    monkeypatch.setenv("STV_CONV_WS2", "2")
    monkeypatch.setenv("STV_WS2_SKEW", skew)
    x = rnd((1, cin, H, W), 171)
    w = rnd((cout, cin, 3, 3), 172, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    b = rnd((cout,), 173, -0.2, 0.2)
    xq, wq = q(x, dtype), q(w, dtype)
    xn = ops.to_nhwc(x, dtype).to(DEV)
    for blocked in (True, False):
        wp = ops.pack_weights_fwd(w).to(dtype).to(DEV)
        if blocked:
            wp = ops.block_weights(wp)
        for flags in (0, ops.RELU_IN | ops.RELU_OUT):
            ref = F.conv2d(F.relu(xq) if flags & ops.RELU_IN else xq, wq, b, padding=1)
            ref = F.relu(ref) if flags & ops.RELU_OUT else ref
            y = ops.conv_igemm(xn, wp, b.to(DEV), flags=flags)
            assert_close(ops.from_nhwc(y), ref, dtype, 9 * cin, f"ws2 fwd {case} flags={flags} blocked={blocked}")
            # the one-wave-per-SIMD kernel on the same launch: another order of the K halves, same values to rounding
            monkeypatch.setenv("STV_CONV_WS2", "0")
            monkeypatch.setenv("STV_CONV_WS", "2")
            y1 = ops.conv_igemm(xn, wp, b.to(DEV), flags=flags)
            monkeypatch.setenv("STV_CONV_WS2", "2")
            d = (y.float() - y1.float()).abs()
            assert float((d / (2.0 ** -7 * torch.maximum(y.float().abs(), y1.float().abs()) + 1e-6)).max()) <= 1.0
            assert float((d > 0).float().mean()) < 5e-3
    wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
    for relu_in in (0, ops.RELU_IN):
        idx = torch.full((H // 2, W // 2, cout), 255, device=DEV, dtype=torch.uint8)
        y, yp = ops.conv_igemm_pool(xn, wp, b.to(DEV), flags=ops.RELU_OUT | relu_in, pool_idx=idx)
        ref = F.relu(F.conv2d(F.relu(xq) if relu_in else xq, wq, b, padding=1))
        assert_close(ops.from_nhwc(y), ref, dtype, 9 * cin, f"ws2 conv+pool {case} relu_in={relu_in}")
        assert torch.equal(ops.from_nhwc(yp).cpu(), F.max_pool2d(ops.from_nhwc(y).cpu(), 2, 2))
        assert int(idx.max()) <= 7
        idx2 = torch.full_like(idx, 255)
        y2 = torch.full_like(y, 7.0)
        _, yp2 = ops.conv_igemm_pool(xn, wp, b.to(DEV), flags=ops.RELU_OUT | relu_in | ops.POOL_ONLY, out=y2, pool_idx=idx2)
        assert torch.equal(yp2, yp) and torch.equal(idx2, idx) and bool((y2 == 7.0).all())
    dyp = ops.to_nhwc(rnd((1, cout, H // 2, W // 2), 174), dtype).to(DEV)
    for flags in (0, ops.MASK):
        a_, b_ = torch.zeros_like(y), torch.zeros_like(y)
        ops.maxpool_bwd(y, dyp, out=a_, flags=flags)
        ops.maxpool_bwd_idx(idx, dyp, H, W, out=b_, flags=flags)
        assert torch.equal(a_, b_), f"ws2 arg-max map differs from the activation-based routing, flags={flags}"


@pytest.mark.parametrize("cfg", [None, 0, 1, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
@pytest.mark.parametrize("case", [(128, 64, 32, 64), (256, 128, 16, 40), (512, 512, 8, 8), (512, 256, 12, 20)])
def test_conv_igemm_route_equals_dgrad_then_pool_backward(cfg, case, monkeypatch):
    """stv_conv_igemm_route: the dgrad of the conv behind a max-pool writes the pre-pool gradient directly.
    Must equal, bit for bit, the two launches it replaces (same dgrad arithmetic, then the arg-max routing
    of stv_maxpool_bwd with and without the ReLU mask)."""
    cd, cs, H, W = case                      # dy channels (the conv's Cout), routed channels (its Cin), pooled size
    dtype = torch.bfloat16
    if cfg is not None:
        if cs <= 64 and cfg in (0, 2, 18):
            pytest.skip("128-channel tiles need more than 64 output channels")
        monkeypatch.setenv("STV_CONV_CFG", str(cfg))
    # a forward conv + pool produces a genuine arg-max map (with ties from the ReLU zeros)
    xf = rnd((1, 64, 2 * H, 2 * W), 161)
    wf = rnd((cs, 64, 3, 3), 162, -1, 1) * (2.0 / (9 * 64)) ** 0.5
    idx = torch.empty(H, W, cs, device=DEV, dtype=torch.uint8)
    monkeypatch.setenv("STV_CONV_WS", "0")
    y, _ = ops.conv_igemm_pool(ops.to_nhwc(xf, dtype).to(DEV), ops.block_weights(ops.pack_weights_fwd(wf).to(dtype).to(DEV)),
                               None, flags=ops.RELU_OUT, pool_idx=idx)
    w = rnd((cd, cs, 3, 3), 163, -1, 1) * (2.0 / (9 * cd)) ** 0.5
    wb = ops.block_weights(ops.pack_weights_bwd(w).to(dtype).to(DEV))
    dy = ops.to_nhwc(rnd((1, cd, H, W), 164), dtype).to(DEV)
    pooled_grad = ops.conv_igemm(dy, wb, None)
    for flags in (0, ops.MASK):
        want = torch.full((2 * H, 2 * W, cs), 7.0, device=DEV, dtype=dtype)
        ops.maxpool_bwd_idx(idx, pooled_grad, 2 * H, 2 * W, out=want, flags=flags)
        got = torch.full((2 * H, 2 * W, cs), -3.0, device=DEV, dtype=dtype)
        ops.conv_igemm_route(dy, wb, idx, out=got, flags=flags)
        assert torch.equal(got, want), f"route {case} cfg={cfg} flags={flags}"


def test_conv_ws_matches_the_general_kernel(monkeypatch):
    """Same stage / tap accumulation order as conv_igemm.hip; only the bias enters first instead of last
    (it initialises the accumulators): the two kernels agree except for rare one-ulp roundings."""
    dtype = torch.bfloat16
    x = ops.to_nhwc(rnd((1, 64, 96, 160), 151), dtype).to(DEV)
    w = rnd((128, 64, 3, 3), 152, -1, 1) * (2.0 / (9 * 64)) ** 0.5
    b = rnd((128,), 153, -0.2, 0.2).to(DEV)
    wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
    monkeypatch.setenv("STV_CONV_WS", "2")
    assert ops.conv_uses_ws(96, 160, 64, 128, dtype, flags=ops.RELU_IN | ops.W_BLOCKED)
    y_ws = ops.conv_igemm(x, wp, b, flags=ops.RELU_IN)
    monkeypatch.setenv("STV_CONV_WS", "0")
    assert not ops.conv_uses_ws(96, 160, 64, 128, dtype, flags=ops.RELU_IN | ops.W_BLOCKED)
    y_gen = ops.conv_igemm(x, wp, b, flags=ops.RELU_IN)
    a_, g_ = y_ws.float(), y_gen.float()
    diff = (a_ - g_).abs()
    assert float((diff > 0).float().mean()) < 2e-3
    assert float((diff / (2.0 ** -7 * torch.maximum(a_.abs(), g_.abs()) + 1e-6)).max()) <= 1.0
    # without a bias the accumulation is the same sequence of MFMAs: bit for bit
    monkeypatch.setenv("STV_CONV_WS", "2")
    y_ws0 = ops.conv_igemm(x, wp, None, flags=ops.RELU_IN)
    monkeypatch.setenv("STV_CONV_WS", "0")
    assert torch.equal(y_ws0, ops.conv_igemm(x, wp, None, flags=ops.RELU_IN))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
@pytest.mark.parametrize("case", [(128, 128, 21, 70), (64, 64, 9, 33), (48, 136, 12, 40), (512, 128, 8, 8)])
def test_conv_igemm_every_tile_config(dtype, cfg, case, monkeypatch):
    """Each tile shape / wave layout / K split (forced with STV_CONV_CFG) on full, ragged and odd-K shapes."""
    cin, cout, H, W = case
    if cout <= 64 and cfg in (0, 2, 18):
        pytest.skip("128-channel tiles need more than 64 output channels")
    monkeypatch.setenv("STV_CONV_CFG", str(cfg))
    x = rnd((1, cin, H, W), 41)
    w = rnd((cout, cin, 3, 3), 42, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    b = rnd((cout,), 43, -0.2, 0.2)
    z = rnd((1, cout, H, W), 44)
    prev = rnd((1, cout, H, W), 45)
    ref = F.relu(F.conv2d(F.relu(q(x, dtype)), q(w, dtype), b, padding=1))
    wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
    y = ops.conv_igemm(ops.to_nhwc(x, dtype).to(DEV), wp, b.to(DEV), flags=ops.RELU_IN | ops.RELU_OUT)
    assert_close(ops.from_nhwc(y), ref, dtype, 9 * cin, f"cfg {cfg} fwd {case}")
    ref2 = F.conv2d(q(x, dtype), q(w, dtype), None, padding=1) * (q(z, dtype) > 0).float() + q(prev, dtype)
    out = ops.to_nhwc(prev, dtype).to(DEV)
    ops.conv_igemm(ops.to_nhwc(x, dtype).to(DEV), wp, None, ref=ops.to_nhwc(z, dtype).to(DEV), out=out,
                   flags=ops.MASK | ops.ACCUM)
    assert_close(ops.from_nhwc(out), ref2, dtype, 9 * cin, f"cfg {cfg} mask+accum {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(128, 512, 21, 70, None), (256, 512, 16, 64, 3), (64, 256, 40, 33, 5), (512, 512, 8, 8, 11), (128, 384, 12, 40, 1)])
def test_conv_xshare_changes_nothing(dtype, case, monkeypatch):
    """STV_CONV_XSHARE = 2 / 4 deals the channel blocks of a spatial tile to 2 / 4 XCDs (ConvArgs::xshare; the block
    decode of conv_igemm_kernel.h): another workgroup -> output-block mapping, the same per-block arithmetic - results
    bit-identical to the default mapping, also where the block count is not divisible by the request (384 / 128 = 3
    blocks: falls back to 1) and with a second pass (mask + accumulate) on top."""
    cin, cout, H, W, cfg = case
    if cfg is not None:
        monkeypatch.setenv("STV_CONV_CFG", str(cfg))
    x = ops.to_nhwc(rnd((1, cin, H, W), 141), dtype).to(DEV)
    w = rnd((cout, cin, 3, 3), 142, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
    b = rnd((cout,), 143, -0.2, 0.2).to(DEV)
    z = ops.to_nhwc(rnd((1, cout, H, W), 144), dtype).to(DEV)
    prev = ops.to_nhwc(rnd((1, cout, H, W), 145), dtype).to(DEV)

    def both():
        y = ops.conv_igemm(x, wp, b, flags=ops.RELU_IN | ops.RELU_OUT).clone()
        out = prev.clone()
        ops.conv_igemm(x, wp, None, ref=z, out=out, flags=ops.MASK | ops.ACCUM)
        return y, out
    monkeypatch.setenv("STV_CONV_XSHARE", "1")
    base = both()
    ref = F.relu(F.conv2d(F.relu(ops.from_nhwc(x).cpu()), q(w, dtype), b.cpu(), padding=1))
    assert_close(ops.from_nhwc(base[0]), ref, dtype, 9 * cin, f"xshare 1 {case}")
    for g in (2, 4):
        monkeypatch.setenv("STV_CONV_XSHARE", str(g))
        got = both()
        assert torch.equal(got[0], base[0]) and torch.equal(got[1], base[1]), f"xshare {g} {case}"


@pytest.fixture
def xk_workspace(monkeypatch):
    """K split across workgroups switched on for the test, with its scratch (stv_conv_workspace)."""
    monkeypatch.setenv("STV_CONV_XK", "1")
    ws = torch.zeros(ops.conv_workspace_bytes(), dtype=torch.uint8, device=DEV)
    ops.set_conv_workspace(ws)
    yield ws
    ops.set_conv_workspace(None)


@pytest.mark.parametrize("case", [(512, 512, 64, 64), (512, 448, 40, 96), (256, 512, 61, 64), (512, 384, 64, 64)])
def test_conv_split_k_across_workgroups(case, xk_workspace, monkeypatch):
    """ConvArgs::xk (STV_CONV_XK=1): two workgroups per output tile, each on half of K, the later one adds the other's fp32
    partial sums and runs the epilogue - every epilogue of the general kernel behind it (bias + ReLU, mask + accumulate, the
    fused 1x1 Gram term, the max-pool + arg-max map, the routed pooling backward), ragged tiles, a tile count that is not
    a multiple of eight (padding blocks) - against the CPU reference; the same bits on every run (a + b == b + a:
    whichever half finishes first); the scratch words are zero again after every launch; and the plain kernel for
    comparison (another summation order: same tolerance, not the same bits)."""
    cin, cout, H, W = case
    dtype = torch.bfloat16
    ws = xk_workspace
    x = rnd((1, cin, H, W), 151)
    w = rnd((cout, cin, 3, 3), 152, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    b = rnd((cout,), 153, -0.2, 0.2)
    z = rnd((1, cout, H, W), 154)
    prev = rnd((1, cout, H, W), 155)
    xd, bd = ops.to_nhwc(x, dtype).to(DEV), b.to(DEV)
    wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
    zd, pd = ops.to_nhwc(z, dtype).to(DEV), ops.to_nhwc(prev, dtype).to(DEV)

    def clean():
        torch.cuda.synchronize()
        assert int(ws[:65536].max()) == 0, "tickets / flags not reset"
    # forward: bias + ReLU in and out
    ref = F.relu(F.conv2d(F.relu(q(x, dtype)), q(w, dtype), b, padding=1))
    y1 = ops.conv_igemm(xd, wp, bd, flags=ops.RELU_IN | ops.RELU_OUT).clone()
    clean()
    assert_close(ops.from_nhwc(y1), ref, dtype, 9 * cin, f"xk fwd {case}")
    for _ in range(3):
        assert torch.equal(ops.conv_igemm(xd, wp, bd, flags=ops.RELU_IN | ops.RELU_OUT), y1), "not reproducible"
    monkeypatch.setenv("STV_CONV_XK", "0")
    y0 = ops.conv_igemm(xd, wp, bd, flags=ops.RELU_IN | ops.RELU_OUT)
    monkeypatch.setenv("STV_CONV_XK", "1")
    assert_close(ops.from_nhwc(y0), ref, dtype, 9 * cin, f"plain fwd {case}")
    assert not torch.equal(y0, y1) or cin < 256, "the K split did not engage (same bits as the plain kernel)"
    # mask + accumulate (a dgrad's epilogue)
    ref2 = F.conv2d(q(x, dtype), q(w, dtype), None, padding=1) * (q(z, dtype) > 0).float() + q(prev, dtype)
    out = pd.clone()
    ops.conv_igemm(xd, wp, None, ref=zd, out=out, flags=ops.MASK | ops.ACCUM)
    clean()
    assert_close(ops.from_nhwc(out), ref2, dtype, 9 * cin, f"xk mask+accum {case}")
    # fused max-pool + arg-max map (even sizes only)
    if H % 2 == 0 and W % 2 == 0:
        idx = torch.full((H // 2, W // 2, cout), 255, device=DEV, dtype=torch.uint8)
        y, yp = ops.conv_igemm_pool(xd, wp, bd, flags=ops.RELU_OUT, pool_idx=idx)
        clean()
        refp = F.relu(F.conv2d(q(x, dtype), q(w, dtype), b, padding=1))
        assert_close(ops.from_nhwc(y), refp, dtype, 9 * cin, f"xk conv+pool {case}")
        assert torch.equal(ops.from_nhwc(yp).cpu(), F.max_pool2d(ops.from_nhwc(y).cpu(), 2, 2))
        assert int(idx.max()) <= 7
    # the dgrad with the Gram-backward 1x1 term of its output: dy has `cin` channels (the layer above), the output,
    # F = z and S have `cout` (this layer) - as in test_conv_igemm_dual_dgrad_plus_gram_term
    wd = rnd((cin, cout, 3, 3), 157, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    s_mat = rnd((cout, cout), 156, -0.02, 0.02)
    sq = q((s_mat + s_mat.t()) * 0.5, dtype)
    xr = torch.zeros(1, cout, H, W, requires_grad=True)
    F.conv2d(xr, q(wd, dtype), None, padding=1).backward(q(x, dtype))
    ref3 = xr.grad * (q(z, dtype) > 0).float() + torch.einsum("bchw,nc->bnhw", q(z, dtype), sq)
    wb = ops.block_weights(ops.pack_weights_bwd(wd).to(dtype).to(DEV))
    out3 = torch.empty_like(zd)
    ops.conv_igemm_dual(xd, wb, zd, sq.to(dtype).to(DEV).contiguous(), ref=zd, out=out3, flags=ops.MASK)
    clean()
    assert_close(ops.from_nhwc(out3), ref3, dtype, 9 * cin + cout, f"xk dual {case}")
    # the routed pooling backward in the epilogue (a dgrad in front of a max-pool): against the plain kernel's routing of
    # the same sums rounded the same way is not available (another summation order), so against dgrad-then-pool-backward
    if H % 2 == 0 and W % 2 == 0 and 4 * H * W * cout * 2 < 2 ** 31:
        idx = torch.from_numpy((synthetic.hash_uniform(158, 3, H * W * cout) * 8).astype(np.uint8).reshape(H, W, cout)).to(DEV)
        routed = ops.conv_igemm_route(xd, wp, idx, flags=ops.MASK)
        clean()
        plain = ops.conv_igemm(xd, wp, None, flags=0)            # the same sums (K split as well), unrouted
        want = torch.zeros(2 * H, 2 * W, cout, device=DEV, dtype=dtype)
        ops.maxpool_bwd_idx(idx, plain, 2 * H, 2 * W, out=want, flags=ops.MASK)
        assert torch.equal(routed, want), f"xk route {case}"


@pytest.mark.parametrize("hint_bytes", [1, 100, 4096, 1 << 20, 5 << 20])
def test_next_weights_hint_changes_nothing_but_timing(hint_bytes):
    """stv_conv_next_weights: the launch that follows touches one 128-byte line per lane of the hinted range - fewer
    lanes than lines, more lanes than lines, a range that ends inside a line - and its own result is bit-identical
    to the launch without a hint; the hinted buffer is only read."""
    from style_transfer_visualizer_amd import _lib
    lib = _lib.load()
    cin, cout, H, W = 64, 128, 24, 40
    dtype = torch.bfloat16
    x = ops.to_nhwc(rnd((1, cin, H, W), 91), dtype).to(DEV)
    w = rnd((cout, cin, 3, 3), 92, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    wp = ops.block_weights(ops.pack_weights_fwd(w).to(dtype).to(DEV))
    b = rnd((cout,), 93, -0.2, 0.2).to(DEV)
    plain = ops.conv_igemm(x, wp, b, flags=ops.RELU_OUT).clone()
    hinted = torch.arange(hint_bytes, device=DEV, dtype=torch.int64).to(torch.uint8)
    before = hinted.clone()
    lib.stv_conv_next_weights(hinted.data_ptr(), hint_bytes)
    try:
        got = ops.conv_igemm(x, wp, b, flags=ops.RELU_OUT)
    finally:
        lib.stv_conv_next_weights(None, 0)
    torch.cuda.synchronize()
    assert torch.equal(got, plain)
    assert torch.equal(hinted, before)
    assert torch.equal(ops.conv_igemm(x, wp, b, flags=ops.RELU_OUT), plain)       # hint cleared: the plain launch again


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("blocked", [False, True])
def test_conv_igemm_dgrad_mask_accum(dtype, case, blocked):
    cin, cout, H, W = case
    w = rnd((cout, cin, 3, 3), 21, -1, 1) * (2.0 / (9 * cin)) ** 0.5
    dy = rnd((1, cout, H, W), 22)
    zref = rnd((1, cin, H, W), 23)        # stored activation whose ReLU mask applies
    prev = rnd((1, cin, H, W), 24)        # gradient already in the buffer (ACCUM)
    wq, dyq, zq, pq = q(w, dtype), q(dy, dtype), q(zref, dtype), q(prev, dtype)
    xr = torch.zeros(1, cin, H, W, requires_grad=True)
    F.conv2d(xr, wq, None, padding=1).backward(dyq)
    ref = xr.grad * (zq > 0).float() + pq
    out = ops.to_nhwc(prev, dtype).to(DEV)
    wp = _maybe_blocked(ops.pack_weights_bwd(w).to(dtype).to(DEV), blocked, H, W, cout, cin, dtype)
    ops.conv_igemm(ops.to_nhwc(dy, dtype).to(DEV), wp, None,
                   ref=ops.to_nhwc(zref, dtype).to(DEV), out=out, flags=ops.MASK | ops.ACCUM)
    assert_close(ops.from_nhwc(out), ref, dtype, 9 * cout, f"dgrad {case}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,H,W", [(64, 20, 36), (128, 8, 40), (256, 5, 9), (8, 6, 10)])
def test_conv_1x1_as_gram_backward(dtype, C, H, W):
    f = rnd((H, W, C), 31)
    s = rnd((C, C), 32, -0.01, 0.01)
    s = (s + s.t()) * 0.5
    fq, sq = q(f, dtype), q(s, dtype)
    ref = (fq.reshape(-1, C) @ sq.t()).reshape(H, W, C)
    out = ops.conv_igemm(f.to(dtype).to(DEV), s.reshape(1, C, C).to(dtype).to(DEV), None)
    got = out.float().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) / scale <= tol(dtype, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hwc", [(8, 12, 64), (7, 9, 16), (6, 6, 4), (32, 32, 128)])
def test_maxpool_forward_backward(dtype, hwc):
    H, W, C = hwc
    x = rnd((1, C, H, W), 41)
    x = torch.where(x < -0.3, torch.zeros_like(x), x)   # plenty of exact ties at 0
    x[0, :, :2, :2] = 0.5                                # positive ties: first max must win
    xq = q(x, dtype)
    xr = xq.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 2, 2)
    dy = rnd(tuple(y_ref.shape), 42)
    dyq = q(dy, dtype)
    y_ref.backward(dyq)
    xd = ops.to_nhwc(x, dtype).to(DEV)
    y = ops.maxpool_fwd(xd)
    assert torch.equal(ops.from_nhwc(y).cpu(), y_ref.detach())
    dx = ops.maxpool_bwd(xd, ops.to_nhwc(dy, dtype).to(DEV))
    assert torch.equal(ops.from_nhwc(dx).cpu(), xr.grad)
    # fused ReLU mask + accumulate
    prev = q(rnd((1, C, H, W), 43), dtype)
    out = ops.to_nhwc(prev, dtype).to(DEV)
    ops.maxpool_bwd(xd, ops.to_nhwc(dy, dtype).to(DEV), out=out, flags=ops.MASK | ops.ACCUM)
    ref = q(xr.grad * (xq > 0).float() + prev, dtype) if dtype == torch.bfloat16 else xr.grad * (xq > 0).float() + prev
    assert_close(ops.from_nhwc(out), ref, dtype, 1, "maxpool_bwd mask+accum")


@pytest.mark.parametrize("dtype", DTYPES)
def test_relu_forward_backward(dtype):
    x = rnd((9, 11, 20), 51)
    dy = rnd((9, 11, 20), 52)
    xd, dyd = x.to(dtype).to(DEV), dy.to(dtype).to(DEV)
    assert torch.equal(ops.relu_fwd(xd).float().cpu(), F.relu(q(x, dtype)))
    assert torch.equal(ops.relu_bwd(xd, dyd).float().cpu(), q(dy, dtype) * (q(x, dtype) > 0).float())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,H,W,scale", [(64, 37, 53, 1.0), (128, 16, 24, 1.0), (256, 9, 11, 1.0),
                                         (512, 8, 8, 1.0), (8, 10, 14, 1.0), (64, 64, 64, 40.0)])
def test_gram_forward_loss_and_seed(dtype, C, H, W, scale):
    n = H * W
    f = rnd((1, C, H, W), 61) * scale
    fq = q(f, dtype)
    clamp = 5e5
    tgt = ocm.gram_matrix(q(rnd((1, C, H, W), 62) * scale, dtype))
    fr = fq.clone().requires_grad_(True)
    g_ref = ocm.gram_matrix(fr, clamp)
    loss_ref = F.mse_loss(g_ref, tgt)
    coef = 1e5
    (coef * loss_ref).backward()
    raw = fq.reshape(C, n) @ fq.reshape(C, n).t()
    if scale > 1:
        assert int((raw > clamp).sum()) > 0, "case must engage the clamp"

    fd = ops.to_nhwc(f, dtype).to(DEV)
    partials = ops.gram_partial(fd)
    gram = torch.empty(C, C, device=DEV)
    parts = torch.zeros(ops.gram_loss_parts(C), device=DEV)
    sgrad = torch.empty(C, C, device=DEV, dtype=dtype)
    ops.gram_finish(partials, n, C, target=tgt.to(DEV), gram_out=gram, loss_part=parts, sgrad=sgrad,
                    clamp_max=clamp, coef=coef, dtype=dtype)
    assert_close(gram, g_ref.detach(), torch.float32, n, "gram")
    loss = float(parts.double().sum().cpu()) / (C * C)
    assert loss == pytest.approx(float(loss_ref), rel=1e-4)
    # backward product dF^T = F^T S as a 1x1 conv
    df = ops.conv_igemm(fd, sgrad.reshape(1, C, C), None)
    ref = fr.grad
    got = ops.from_nhwc(df).cpu()
    err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
    assert err <= (1e-4 if dtype == torch.float32 else 2e-2), f"gram bwd err {err:.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_content_loss_and_grad(dtype):
    f = rnd((17, 23, 32), 71)
    t = rnd((17, 23, 32), 72)
    fq, tq = q(f, dtype), q(t, dtype)
    fr = fq.clone().requires_grad_(True)
    loss_ref = F.mse_loss(fr, tq)
    (3.0 * loss_ref).backward()
    fd, td = f.to(dtype).to(DEV), t.to(dtype).to(DEV)
    parts = torch.zeros(256, device=DEV)
    ops.content_loss(fd, td, parts)
    assert float(parts.double().sum().cpu()) / f.numel() == pytest.approx(float(loss_ref), rel=1e-5)
    prev = q(rnd((17, 23, 32), 73), dtype)
    out = prev.to(dtype).to(DEV)
    ops.content_grad(fd, td, out, 3.0, flags=ops.ACCUM)
    assert_close(out, fr.grad + prev, dtype, 1, "content grad")
    # both in one pass (stv_content_loss_grad): the same partial sums, the same (written) gradient, bit for bit
    parts2 = torch.zeros(256, device=DEV)
    g_one = torch.full_like(fd, float("nan"))
    ops.content_loss_grad(fd, td, parts2, g_one, 3.0)
    g_two = torch.empty_like(fd)
    ops.content_grad(fd, td, g_two, 3.0)
    assert torch.equal(parts2, parts) and torch.equal(g_one, g_two)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gram_multi_matches_the_single_tap_calls(dtype):
    """One batched launch pair for five taps (tile sizes 64 and 128, 1..512 slabs) = the per-tap calls."""
    shapes = [(64, 48, 64), (32, 40, 128), (16, 24, 256), (8, 8, 512), (12, 20, 8)]
    feats = [rnd((H, W, C), 80 + i, -1.5, 1.5).to(dtype).to(DEV) for i, (H, W, C) in enumerate(shapes)]
    tgts = [ocm.gram_matrix(rnd((1, C, H, W), 90 + i)).to(DEV).contiguous() for i, (H, W, C) in enumerate(shapes)]
    grams, parts, seeds = ops.gram_multi(feats, tgts, coef=3.0)
    for f, t, g, lp, sg in zip(feats, tgts, grams, parts, seeds, strict=True):
        H, W, C = f.shape
        n = H * W
        partials = ops.gram_partial(f)
        g1 = torch.empty(C, C, device=DEV)
        lp1 = torch.empty(ops.gram_loss_parts(C), device=DEV)
        sg1 = torch.empty(C, C, device=DEV, dtype=dtype)
        ops.gram_finish(partials, n, C, target=t, gram_out=g1, loss_part=lp1, sgrad=sg1, coef=3.0, dtype=dtype)
        scale = float(g1.abs().max())
        assert float((g - g1).abs().max()) <= 2e-6 * scale          # summation order of the slabs may differ
        assert float(lp.sum()) == pytest.approx(float(lp1.sum()), rel=1e-5)
        assert float((sg.float() - sg1.float()).abs().max()) <= (2e-6 if dtype == torch.float32 else 8e-3) * float(sg1.float().abs().max())


def test_loss_combine():
    parts = torch.arange(1, 41, dtype=torch.float32, device=DEV)
    table = torch.tensor([[0, 10, 0], [10, 10, 0], [20, 20, 1]], dtype=torch.int32, device=DEV)
    scale = torch.tensor([0.5, 0.25, 2.0], device=DEV)
    losses = torch.zeros(3, device=DEV)
    scores = torch.zeros(4, device=DEV)
    ops.loss_combine(parts, table, scale, 1e5, 2.0, losses, scores)
    l = [55 * 0.5, 155 * 0.25, 610 * 2.0]
    assert losses.cpu().tolist() == l
    style = np.float32(l[0]) + np.float32(l[1])
    total = np.float32(1e5) * style + np.float32(2.0) * np.float32(l[2])
    assert scores.cpu().tolist() == [float(style), l[2], float(total), 1.0]


def _quartic(n, seed):
    a = rnd((n,), seed, 0.5, 2.0)
    b = rnd((n,), seed + 1, -1.0, 1.0)

    def f(x):
        return (0.5 * a * x * x - b * x + 0.05 * x ** 4).sum() + 0.1 * (x[1:] * x[:-1]).sum()
    return f


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("history", [4, 100])
def test_lbfgs_step_matches_oracle(history, compact):
    n, steps = 20000, 14
    f = _quartic(n, 81)
    x_ref = torch.zeros(n)
    ref = optim_ref.LbfgsRef(x_ref, lr=1.0, history_size=history)
    x = torch.zeros(n, device=DEV)
    state, work = ops.lbfgs_alloc(n, history, torch.device(DEV), compact=compact)
    for step in range(steps):
        def closure():
            with torch.enable_grad():
                xr = x_ref.detach().clone().requires_grad_(True)
                loss = f(xr)
                loss.backward()
            return loss.detach(), xr.grad
        ref.step(closure)
        with torch.enable_grad():
            xg = x.detach().cpu().clone().requires_grad_(True)
            f(xg).backward()
        ops.lbfgs_step(x, xg.grad.to(DEV), state, work, history, min(step, history), 1.0, compact=compact)
        err = float((x.cpu() - x_ref).abs().max()) / float(x_ref.abs().max())
        assert err < 2e-4, f"step {step + 1}: x diverged from the oracle by {err:.3e}"
    st = state.cpu().view(torch.int32)
    assert int(st[0]) == ref.n_iter
    assert int(st[1]) == len(ref.old_dirs)


@pytest.mark.parametrize("compact", [False, True])
def test_lbfgs_early_return_and_no_descent(compact):
    n = 4099                                            # not a multiple of the vector width
    x = torch.ones(n, device=DEV)
    state, work = ops.lbfgs_alloc(n, 100, torch.device(DEV), compact=compact)
    tiny = torch.full((n,), 5e-8, device=DEV)           # |g|max <= 1e-7 -> return before any update
    ops.lbfgs_step(x, tiny, state, work, 100, 0, 1.0, compact=compact)
    st = state.cpu().view(torch.int32)
    assert int(st[0]) == 0 and int(st[3]) == 1
    assert torch.equal(x.cpu(), torch.ones(n))
    g = torch.full((n,), 1.0, device=DEV)
    ops.lbfgs_step(x, g, state, work, 100, 0, 1.0, compact=compact)   # first real step: t = min(1, 1/|g|_1)
    st = state.cpu().view(torch.int32)
    assert int(st[0]) == 1 and int(st[3]) == 0
    np.testing.assert_allclose(x.cpu().numpy(), 1.0 - 1.0 / n, rtol=1e-6)


def test_adam_step_matches_oracle():
    n = 5000
    f = _quartic(n, 91)
    x_ref = rnd((n,), 93)
    x = x_ref.clone().to(DEV)
    ref = optim_ref.AdamRef(x_ref, lr=1e-2)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    for step in range(1, 9):
        def closure():
            with torch.enable_grad():
                xr = x_ref.detach().clone().requires_grad_(True)
                loss = f(xr)
                loss.backward()
            return loss.detach(), xr.grad
        ref.step(closure)
        with torch.enable_grad():
            xg = x.detach().cpu().clone().requires_grad_(True)
            f(xg).backward()
        ops.adam_step(x, xg.grad.to(DEV), m, v, step, lr=1e-2)
    np.testing.assert_allclose(x.cpu().numpy(), x_ref.numpy(), rtol=1e-5, atol=1e-6)

"""Row-strip partition of one image (BASELINE configs[4] pattern) vs the single-GPU path.

Two ranks (gloo, both on cuda:0 - this box has one GPU; on a node the same code runs over
RCCL) split a 512x96 image into strips.  Two partitions: ``HaloShard`` (a 1-row halo exchanged
before every 3x3 convolution, Adam and L-BFGS with all-reduced inner products) and the older
``SpatialShard`` (recomputed 160-row halos, Adam).  Losses and the image gradient must equal the
unsharded HIP result to fp32 rounding, and three optimizer steps must give the same image.
"""
from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

MINI = (8, 8, "M", 16, 16, "M", 32, 32, 32, 32, "M", 64, 64, 64, 64, "M", 64, 64, 64, 64, "M")
S_AT, C_AT = [0, 5, 10, 19, 28], [21]
H, W = 512, 96


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup():
    from style_transfer_visualizer_amd import core_model, synthetic
    dev = torch.device("cuda:0")
    weights = synthetic.synthetic_conv_weights(3, MINI)
    content = synthetic.synthetic_image(0, H, W).to(dev)
    style = synthetic.synthetic_image(1, 96, 128).to(dev)
    x0 = synthetic.synthetic_image(2, H, W).to(dev)
    saved = core_model.initialize_vgg                 # (also runs in spawned workers, where no monkeypatch fixture exists)
    core_model.initialize_vgg = lambda: core_model.build_vgg_features(weights, MINI).eval()
    try:
        model = core_model.StyleContentModel(S_AT, C_AT).to(dev)
    finally:
        core_model.initialize_vgg = saved
    model.set_targets(style, content)
    return model, content, x0, dev


def _worker(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from style_transfer_visualizer_amd import spatial
    model, content, x0, dev = _setup()
    shard = spatial.SpatialShard(model._layers(), S_AT, C_AT, content, model.style_targets,
                                 dtype=torch.float32, style_w=1e5, content_w=1.0)
    scores = shard.loss_and_grad(x0)
    x = x0.clone()
    for _ in range(3):
        x = shard.adam_step(x, lr=1e-2)
    torch.cuda.synchronize()
    # numpy (pickled by value): torch tensors in an mp.Queue need the sender alive on receipt
    q.put((rank, shard.c0, shard.c1, shard.e0, shard.e1, scores.cpu().numpy(), None, x.cpu().numpy()))
    # gradient of the first evaluation (adam_step overwrote g_core): recompute at x0
    shard.loss_and_grad(x0)
    torch.cuda.synchronize()
    q.put((rank, "grad", shard.g_core.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_strips_equal_the_unsharded_result():
    from style_transfer_visualizer_amd import optimizers
    model, content, x0, dev = _setup()
    x = x0.clone().requires_grad_(True)
    s_ref, c_ref, t_ref = model.loss_and_grad(x, 1e5, 1.0)
    g_ref = x.grad.clone().cpu()
    ref_scores = torch.stack((s_ref, c_ref, t_ref)).cpu()
    xa = x0.clone().requires_grad_(True)
    adam = optimizers.HipAdam([xa], lr=1e-2)
    for _ in range(3):
        adam.step(lambda: model.loss_and_grad(xa, 1e5, 1.0)[2])
    x_ref = xa.detach().cpu()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(4)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    finals = {g[0]: g for g in got if not isinstance(g[1], str)}
    grads = {g[0]: torch.from_numpy(g[2]) for g in got if isinstance(g[1], str)}
    assert (finals[0][1], finals[0][2]) == (0, 256) and (finals[1][1], finals[1][2]) == (256, 512)
    assert (finals[0][3], finals[0][4]) == (0, 416) and (finals[1][3], finals[1][4]) == (96, 512)
    for r in (0, 1):
        _, c0, c1, _, _, scores, _, x_fin = finals[r]
        scores, x_fin = torch.from_numpy(scores), torch.from_numpy(x_fin)
        assert torch.allclose(scores, ref_scores, rtol=2e-5, atol=0), f"rank {r}: {scores} vs {ref_scores}"
        gscale = float(g_ref.abs().max())
        err = float((grads[r] - g_ref[:, :, c0:c1]).abs().max()) / gscale
        assert err < 2e-5, f"rank {r}: core gradient differs from the unsharded one by {err:.2e}"
        assert float((x_fin - x_ref).abs().max()) < 1e-4, f"rank {r}: image after 3 Adam steps differs"  # lr 1e-2 x normalised update


# ------------------------------------------------------------------------------ per-layer halo exchange
def _halo_worker(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from style_transfer_visualizer_amd import spatial
    model, content, x0, dev = _setup()
    out = {}
    for opt_name, lr in (("adam", 1e-2), ("lbfgs", 1.0)):
        shard = spatial.HaloShard(model._layers(), S_AT, C_AT, content, model.style_targets,
                                  dtype=torch.float32, style_w=1e5, content_w=1.0)
        shard.set_image(x0)
        scores = shard.loss_and_grad()
        out[f"{opt_name}_scores"] = scores.cpu().numpy()
        out[f"{opt_name}_grad"] = shard.g_core.cpu().numpy()
        per_step = []
        for _ in range(3):
            per_step.append(shard.step(opt_name, lr=lr).cpu().numpy())
        out[f"{opt_name}_step_scores"] = per_step
        out[f"{opt_name}_image"] = shard.gather_image().cpu().numpy()
        if opt_name == "lbfgs":
            out["lbfgs_state"] = shard._opt.device_state()
        out["rows"] = (shard.c0, shard.c1)
        out["exchanges"] = shard.exchanges_per_closure
    torch.cuda.synchronize()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_halo_exchange_strips_equal_the_unsharded_result_adam_and_lbfgs(world):
    """world = 4: the two middle ranks exchange halo rows with BOTH neighbours (the configs[4] layout)."""
    import numpy as np

    from style_transfer_visualizer_amd import optimizers
    model, content, x0, dev = _setup()
    x = x0.clone().requires_grad_(True)
    s_ref, c_ref, t_ref = model.loss_and_grad(x, 1e5, 1.0)
    g_ref = x.grad.clone().cpu()
    ref_scores = torch.stack((s_ref, c_ref, t_ref)).cpu()
    refs = {}
    for name, make in (("adam", lambda p: optimizers.HipAdam([p], lr=1e-2)), ("lbfgs", lambda p: optimizers.HipLBFGS([p], lr=1.0))):
        xa = x0.clone().requires_grad_(True)
        opt = make(xa)
        totals = []
        for _ in range(3):
            totals.append(float(opt.step(lambda: model.loss_and_grad(xa, 1e5, 1.0)[2])))
        refs[name] = (xa.detach().cpu(), totals)
        if name == "lbfgs":
            ref_state = opt.device_state()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    rows = 512 // world
    assert [got[r]["rows"] for r in range(world)] == [(r * rows, (r + 1) * rows) for r in range(world)]
    # 13 convs forward + 13 backward: one exchange each (26 per closure)
    assert got[0]["exchanges"] == 26
    gscale = float(g_ref.abs().max())
    for r in range(world):
        c0, c1 = got[r]["rows"]
        for name in ("adam", "lbfgs"):
            scores = torch.from_numpy(got[r][f"{name}_scores"])
            assert torch.allclose(scores, ref_scores, rtol=2e-5, atol=0), f"rank {r} {name}: {scores} vs {ref_scores}"
            err = float((torch.from_numpy(got[r][f"{name}_grad"]) - g_ref[:, :, c0:c1]).abs().max()) / gscale
            assert err < 2e-5, f"rank {r}: own-rows gradient differs from the unsharded one by {err:.2e}"
            x_ref, totals = refs[name]
            step_totals = [float(sv[2]) for sv in got[r][f"{name}_step_scores"]]
            # Losses 1 and 2 are evaluated at x0 and x0 - t*g: identical arithmetic.  From the third
            # evaluation on L-BFGS has used a curvature pair: its update is built from DIFFERENCES of fp32
            # inner products, and the sharded sums (per-strip partials, then added) are ordered differently
            # from the unsharded ones - the 1e-4-of-range effect tests/test_gpu_fullsize.py documents
            # against float64.  Adam has no such cancellation.
            np.testing.assert_allclose(step_totals[:2], totals[:2], rtol=1e-5)
            np.testing.assert_allclose(step_totals[2:], totals[2:], rtol=1e-5 if name == "adam" else 2e-3)
            x_fin = torch.from_numpy(got[r][f"{name}_image"])
            dev_img = float((x_fin - x_ref).abs().max() / x_ref.abs().max())
            bound = 2e-4 if name == "adam" else 2e-3
            assert dev_img < bound, f"rank {r} {name}: image after 3 steps differs by {dev_img:.2e} of its range"
    # both ranks hold the same gathered image, and ran the SAME scalar recursion: after the all-reduce
    # the inner products are bit-identical on every rank, so the optimizer state must be too
    for r in range(1, world):
        assert np.array_equal(got[0]["lbfgs_image"], got[r]["lbfgs_image"])
        assert got[0]["lbfgs_state"] == got[r]["lbfgs_state"]
    st = got[0]["lbfgs_state"]
    assert (st["n_iter"], st["hist_len"], st["skip"]) == (ref_state["n_iter"], ref_state["hist_len"], ref_state["skip"]) == (3, 2, 0)
    assert st["H_diag"] == pytest.approx(ref_state["H_diag"], rel=1e-3)


# ------------------------------------------------------------------------------ closure as one captured graph
def test_strip_closure_replayed_as_one_graph_equals_the_eager_segments(monkeypatch):
    """HaloShard runs its closure - 27 program segments, the halo exchanges, the Gram all-reduce - as ONE
    captured graph after the first (eager) evaluation (RCCL operations are stream operations and capture like
    kernels; with a single strip the exchanges reduce to the zero-filled image border).  Replay vs
    STV_SPATIAL_GRAPH=0, same inputs: bit-identical scores, gradient and images over three L-BFGS steps."""
    from style_transfer_visualizer_amd import spatial
    model, content, x0, dev = _setup()
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("STV_SPATIAL_GRAPH", mode)
        shard = spatial.HaloShard(model._layers(), S_AT, C_AT, content, model.style_targets,
                                  dtype=torch.float32, style_w=1e5, content_w=1.0)
        shard.set_image(x0)
        first = shard.loss_and_grad().clone()              # eager in both modes (graph mode: warm-up + capture)
        second = shard.loss_and_grad().clone()             # graph mode: a replay
        grad = shard.g_core.clone()
        steps = [shard.step("lbfgs", lr=1.0).clone() for _ in range(3)]
        assert (shard._graph is not None) == (mode == "1")
        out[mode] = (first, second, grad, steps, shard.gather_image().clone())
    for a, b in zip(out["0"][:3], out["1"][:3], strict=True):
        assert torch.equal(a, b)
    assert all(torch.equal(a, b) for a, b in zip(out["0"][3], out["1"][3], strict=True))
    assert torch.equal(out["0"][4], out["1"][4])
    # and the single strip IS the unsharded problem
    x = x0.clone().requires_grad_(True)
    s_ref, c_ref, t_ref = model.loss_and_grad(x, 1e5, 1.0)
    assert torch.allclose(out["1"][0].cpu(), torch.stack((s_ref, c_ref, t_ref)).cpu(), rtol=2e-5, atol=0)


def _subgroup_worker(rank: int, world: int, port: int, q) -> None:
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    group = dist.new_group(ranks=[1, 2])                  # every rank takes part in creating it
    if rank == 0:
        dist.barrier()
        dist.destroy_process_group()
        return
    from style_transfer_visualizer_amd import spatial
    model, content, x0, dev = _setup()
    shard = spatial.HaloShard(model._layers(), S_AT, C_AT, content, model.style_targets, dtype=torch.float32,
                              style_w=1e5, content_w=1.0, group=group)
    shard.set_image(x0)
    scores = shard.loss_and_grad()
    torch.cuda.synchronize()
    q.put((rank, (shard.rank, shard.c0, shard.c1, scores.cpu().numpy(), shard.g_core.cpu().numpy())))
    dist.barrier()
    dist.destroy_process_group()


def test_halo_exchange_inside_a_subgroup_uses_global_peer_ranks():
    """A 2-rank strip group that is NOT ranks 0..1 of the world (world = 3, group = {1, 2}): isend / irecv name
    their peers by global rank, the strips by their rank in the group."""
    model, content, x0, dev = _setup()
    x = x0.clone().requires_grad_(True)
    s_ref, c_ref, t_ref = model.loss_and_grad(x, 1e5, 1.0)
    g_ref = x.grad.clone().cpu()
    ref_scores = torch.stack((s_ref, c_ref, t_ref)).cpu()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for world_rank, group_rank in ((1, 0), (2, 1)):
        grank, c0, c1, scores, grad = got[world_rank]
        assert (grank, c0, c1) == (group_rank, 256 * group_rank, 256 * (group_rank + 1))
        assert torch.allclose(torch.from_numpy(scores), ref_scores, rtol=2e-5, atol=0)
        err = float((torch.from_numpy(grad) - g_ref[:, :, c0:c1]).abs().max() / g_ref.abs().max())
        assert err < 2e-5, f"world rank {world_rank}: own-rows gradient differs by {err:.2e}"


# (configs[4]'s strips against the CPU oracle at full size - 544 / 528 x 3840 rows, fp32 - run in
#  tests/test_gpu_configs.py::test_configs4_200_adam_steps: all four strips, 200 Adam steps, oracle at the gathered image.)

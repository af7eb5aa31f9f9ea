"""Multi-GPU execution of the path: independent images, one process per GPU.

The per-step path of one image does not shard usefully at 512^2/1024^2 (26
halo exchanges per closure for ~1 ms of work), and a batch is
NOT a batch dimension (``gram_matrix`` folds batch into channels, reference
core_model.py:56-57).  So N GPUs run N independent content/style pairs -
replicas of the single-image path with their own L-BFGS state - and the only
collective is one all-gather of the finished images (RCCL over xGMI on GPUs,
gloo in CPU tests).  No collective sits on the per-step data path.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None, *, device: torch.device | None = None) -> tuple[int, int, int]:
    """Initialise from the torchrun environment; returns (rank, local_rank, world_size).

    ``device``: the GPU this rank drives (default ``cuda:LOCAL_RANK``); only used to bind the
    RCCL communicator ("nccl" backend)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = device if device is not None else torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shutdown() -> None:
    """Barrier + destroy the process group (no-op for a single process)."""
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def shard_items(n_items: int, rank: int, world: int) -> list[int]:
    """Indices of the independent image pairs this rank owns (round-robin)."""
    if world < 1 or not 0 <= rank < world:
        msg = f"invalid rank {rank} for world size {world}"
        raise ValueError(msg)
    return list(range(rank, n_items, world))


def gather_results(local: list[tuple[int, torch.Tensor]], n_items: int) -> list[torch.Tensor | None]:
    """All-gather (index, image) results so every rank ends with the full, ordered list.

    All images must share one shape/dtype (the benchmark's case); ranks with fewer items pad.
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        out: list[torch.Tensor | None] = [None] * n_items
        for idx, img in local:
            out[idx] = img
        return out
    per_rank = (n_items + world - 1) // world
    proto = local[0][1] if local else None
    shapes = [None] * world
    dist.all_gather_object(shapes, None if proto is None else (tuple(proto.shape), str(proto.dtype), str(proto.device)))
    shape, dtype_s, _ = next(s for s in shapes if s is not None)
    dtype = getattr(torch, dtype_s.split(".")[-1])
    device = proto.device if proto is not None else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu"))
    idx_t = torch.full((per_rank,), -1, dtype=torch.int64, device=device)
    buf = torch.zeros((per_rank, *shape), dtype=dtype, device=device)
    for k, (idx, img) in enumerate(local):
        idx_t[k] = idx
        buf[k].copy_(img)
    all_idx = [torch.empty_like(idx_t) for _ in range(world)]
    all_buf = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(all_idx, idx_t)
    dist.all_gather(all_buf, buf)
    out = [None] * n_items
    for ids, imgs in zip(all_idx, all_buf, strict=True):
        for k, idx in enumerate(ids.tolist()):
            if idx >= 0:
                out[idx] = imgs[k]
    return out


def images_in_flight(n_local: int, requested: int | None = None) -> int:
    """How many of this rank's images run at the same time (each on its own host thread and stream).

    One image leaves the GPU half idle in two different ways - its closure is bound by the matrix cores and their
    instruction issue, its L-BFGS update by HBM - so two or three independent images on one GPU overlap one's update
    with another's closure: `tools/two_images_probe.py`, one MI355X, aggregate steps/s by images in flight 1 / 2 / 3 / 4: 256^2 2,172 / 3,079 /
    3,429; 512^2 1,145 / 1,311 / 1,446 / 1,352; 1024^2 384 / 419 / 421 / 407.  The results do not depend on it (every
    kernel of a step is deterministic and an image shares nothing with its neighbours), but device memory does - every
    image in flight holds its activations and its 2 x 100 L-BFGS history vectors - so it is OPT-IN: the library default
    is 1; ask for more through the argument or ``STV_IMAGES_PER_GPU``."""
    if requested is None:
        requested = int(os.environ.get("STV_IMAGES_PER_GPU", "1"))
    return max(1, min(int(requested), n_local))


def run_sharded(items: list, fn, *, backend: str | None = None, concurrent: int | None = 1) -> list:
    """Run ``fn(index, item) -> Tensor`` on this rank's share of ``items`` (round-robin) and return the
    results of ALL items, in order, on every rank (one all-gather at the end; no collective in between).

    This is BASELINE configs[3]'s pattern - N independent content/style pairs, one per GPU - as a library
    call: ``main.style_transfer_batch`` and ``bench.py --gpus N`` are thin callers.  Works without a
    process group too (one process runs every item).  ``concurrent``: images of this rank in flight at once
    (``images_in_flight``; ``None`` = its default); worker threads inherit the caller's GPU.
    """
    rank, _local, world = init_distributed(backend)
    mine = shard_items(len(items), rank, world)
    k = images_in_flight(len(mine), concurrent) if mine else 1
    if k <= 1:
        local = [(i, fn(i, items[i])) for i in mine]
    else:
        from concurrent.futures import FIRST_EXCEPTION, ThreadPoolExecutor, wait  # noqa: PLC0415
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None

        def worker(i: int):
            if dev is not None:
                torch.cuda.set_device(dev)          # the current device is per thread
            return i, fn(i, items[i])
        with ThreadPoolExecutor(max_workers=k, thread_name_prefix="stv-image") as pool:
            futures = [pool.submit(worker, i) for i in mine]
            done, pending = wait(futures, return_when=FIRST_EXCEPTION)
            failed = next((f for f in done if f.exception() is not None), None)
            if failed is not None:                  # the first failure ends the batch: nothing queued behind it starts
                for f in pending:
                    f.cancel()
                raise failed.exception()
            local = [f.result() for f in futures]
    return gather_results(local, len(items))

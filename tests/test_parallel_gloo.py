"""N>1 path on CPU: two gloo ranks, independent items sharded round-robin, one all-gather at the end."""
from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from style_transfer_visualizer_amd import parallel


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, n_items: int, out_q) -> None:
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, lr, w) == (rank, rank, world)
    mine = parallel.shard_items(n_items, rank, world)
    # stand-in for the per-image optimisation: a deterministic function of the item index
    local = [(i, torch.full((1, 3, 4, 4), float(i) + 0.5)) for i in mine]
    full = parallel.gather_results(local, n_items)
    # the same through the library entry point main.style_transfer_batch / bench.py build on
    ran = []

    def fn(i, item):
        ran.append(i)
        return torch.full((1, 3, 4, 4), float(item) * 2.0)
    again = parallel.run_sharded([10.0 * k for k in range(n_items)], fn, backend="gloo")
    assert ran == mine
    out_q.put((rank, mine, [float(t.mean()) for t in full], [float(t.mean()) for t in again]))
    parallel.shutdown()


def test_two_ranks_shard_and_gather():
    world, n_items = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    by_rank = {r: (mine, vals, again) for r, mine, vals, again in results}
    assert by_rank[0][0] == [0, 2, 4] and by_rank[1][0] == [1, 3]
    expected = [i + 0.5 for i in range(n_items)]
    assert by_rank[0][1] == expected and by_rank[1][1] == expected      # every rank holds the ordered full set
    assert by_rank[0][2] == by_rank[1][2] == [20.0 * k for k in range(n_items)]


def test_single_process_paths():
    assert parallel.shard_items(5, 0, 1) == [0, 1, 2, 3, 4]
    out = parallel.gather_results([(1, torch.ones(2)), (0, torch.zeros(2))], 2)
    assert torch.equal(out[0], torch.zeros(2)) and torch.equal(out[1], torch.ones(2))
    res = parallel.run_sharded(["a", "bb", "ccc"], lambda i, item: torch.tensor([float(len(item) + i)]))
    assert [float(t) for t in res] == [1.0, 3.0, 5.0]


def test_several_items_in_flight_keep_their_order(monkeypatch):
    """``run_sharded(..., concurrent=k)``: k of a rank's items run at once on worker threads; results stay in item order
    whatever order they finish in, and the number in flight follows the argument / STV_IMAGES_PER_GPU / the item count."""
    import threading
    import time
    seen, lock = {"now": 0, "peak": 0, "threads": set()}, threading.Lock()

    def fn(i, item):
        with lock:
            seen["now"] += 1
            seen["peak"] = max(seen["peak"], seen["now"])
            seen["threads"].add(threading.current_thread().name)
        time.sleep(0.02 * (5 - i))                 # later items finish first
        with lock:
            seen["now"] -= 1
        return torch.tensor([float(item)])
    res = parallel.run_sharded([3.0, 1.0, 4.0, 1.5, 9.0], fn, concurrent=3)
    assert [float(t) for t in res] == [3.0, 1.0, 4.0, 1.5, 9.0]
    assert seen["peak"] == 3 and all(n.startswith("stv-image") for n in seen["threads"])
    assert parallel.images_in_flight(4, 1) == 1 and parallel.images_in_flight(2, 8) == 2 and parallel.images_in_flight(0, 3) == 1
    monkeypatch.setenv("STV_IMAGES_PER_GPU", "3")
    assert parallel.images_in_flight(10) == 3
    monkeypatch.delenv("STV_IMAGES_PER_GPU")
    assert parallel.images_in_flight(10) == 1                 # opt-in: the library default is one image at a time


def test_first_failure_cancels_what_has_not_started():
    """Several items in flight: the first exception ends the batch - items still queued behind it never start, and the
    caller sees that exception (not a later one, and not after every other item ran to completion)."""
    import threading
    import time
    started, lock = [], threading.Lock()

    def fn(i, item):
        with lock:
            started.append(i)
        if i == 0:
            time.sleep(0.05)
            raise RuntimeError("item 0 failed")
        time.sleep(0.3)
        return torch.tensor([float(item)])
    with pytest.raises(RuntimeError, match="item 0 failed"):
        parallel.run_sharded(list(range(8)), fn, concurrent=2)
    assert len(started) < 8, started

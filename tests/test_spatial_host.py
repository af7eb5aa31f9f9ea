"""Host-side logic of the row-strip exchange that cannot run on the one-GPU box: the RCCL branch of
HaloShard._p2p (batch_isend_irecv over uint8 views) with the collective replaced by a loop-back fake."""
from __future__ import annotations

import types

import torch
import torch.distributed as dist

from style_transfer_visualizer_amd import spatial


def test_p2p_rccl_branch_stages_non_contiguous_rows(monkeypatch):
    sent = {}

    class FakeOp:
        def __init__(self, fn, tensor, peer, group):
            self.fn, self.tensor, self.peer = fn, tensor, peer

    class Done:
        def wait(self):
            return None

    def fake_batch(ops):
        # loop-back world: what goes to peer p is what comes back from peer p
        for op in ops:
            assert op.tensor.is_contiguous() and op.tensor.dtype == torch.uint8      # what send/recv accept
            if op.fn is dist.isend:
                sent[op.peer] = op.tensor.clone()
        for op in ops:
            if op.fn is dist.irecv:
                op.tensor.copy_(sent[op.peer])
        return [Done() for _ in ops]
    monkeypatch.setattr(dist, "P2POp", FakeOp)
    monkeypatch.setattr(dist, "batch_isend_irecv", fake_batch)
    def make(rank, group=None):
        ns = types.SimpleNamespace(rank=rank, world=3, group=group, _host_p2p=False)
        ns._peer = types.MethodType(spatial.HaloShard._peer, ns)
        return ns
    shard = make(1)       # a middle rank: two neighbours
    # NCHW image rows (non-contiguous [1, 3, W] slices) and NHWC activation rows (contiguous [W, C], bf16)
    x = torch.arange(1 * 3 * 6 * 5, dtype=torch.float32).reshape(1, 3, 6, 5)
    want_top, want_bot = x[:, :, 1].clone(), x[:, :, 4].clone()
    spatial.HaloShard._p2p(shard, x[:, :, 1], x[:, :, 4], x[:, :, 0], x[:, :, 5])
    assert torch.equal(x[:, :, 0], want_top) and torch.equal(x[:, :, 5], want_bot)
    a = torch.arange(6 * 5 * 8, dtype=torch.float32).reshape(6, 5, 8).bfloat16()
    want_top, want_bot = a[1].clone(), a[4].clone()
    spatial.HaloShard._p2p(shard, a[1], a[4], a[0], a[5])
    assert torch.equal(a[0], want_top) and torch.equal(a[5], want_bot)
    # an edge rank only talks to its one neighbour
    edge = make(0)
    sent.clear()
    top_before = a[0].clone()
    spatial.HaloShard._p2p(edge, a[1], a[4], a[0], a[5])
    assert set(sent) == {1} and torch.equal(a[0], top_before)
    # inside a sub-group the peers are named by GLOBAL rank (group ranks 0, 1, 2 = world ranks 4, 5, 7)
    monkeypatch.setattr(dist, "get_global_rank", lambda group, r: group[r])
    sent.clear()
    sub = make(1, group=[4, 5, 7])
    spatial.HaloShard._p2p(sub, a[1], a[4], a[0], a[5])
    assert set(sent) == {4, 7}

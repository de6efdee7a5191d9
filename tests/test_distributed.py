"""world_size-2 gloo tests (CPU) of the batch-sharding layer: contiguous shards,
one fused SUM all-reduce of [loss, shared-kappa gradient], no collective on the data path.

The per-rank solve is injected: on a CPU box the HIP solve cannot run, so the ORACLE stands in
as the local solve (allowed for tests) -- what is exercised is the sharding / reduction logic,
whose result must equal the single-process loop of reference solves (fixture G9 semantics)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffhe.distributed import ShardedBatchSolve, allreduce_sum_fused, shard_range
from oracle import p1_oracle as orc
from _util import golden


def test_shard_range_partitions_contiguously():
    for B in (1, 7, 8, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_range(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(2048, 3, 8) == (768, 1024)          # BASELINE config 4: 256 per GPU


class _OracleSolve(torch.autograd.Function):
    """u(kappa, f) with the oracle's explicit adjoint; scalar kappa shared by the batch."""

    @staticmethod
    def forward(ctx, kappa, f, mesh):
        us = [orc.solve(*mesh, float(kappa), fb.numpy()) for fb in f]
        ctx.mesh, ctx.kappa, ctx.f = mesh, float(kappa), f
        return torch.from_numpy(np.stack(us))

    @staticmethod
    def backward(ctx, g):
        dk = 0.0
        for fb, gb in zip(ctx.f, g):
            _, dke, _ = orc.solve_with_adjoint(*ctx.mesh, ctx.kappa, fb.numpy(), lambda u, gb=gb: gb.numpy())
            dk += dke.sum()
        return torch.tensor(dk, dtype=torch.float64), None, None


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = golden("g9_batch_2d_8")
        mesh = (g["nodes"], g["elements"], g["bc_nodes"], g["bc_vals"])
        f = torch.from_numpy(g["f"])
        kappa = torch.tensor(float(g["kappa"][0]), dtype=torch.float64, requires_grad=True)
        drv = ShardedBatchSolve(lambda k, fl: _OracleSolve.apply(k, fl, mesh))
        assert drv.world == world and drv.rank == rank
        loss, u_local = drv.step(f, lambda u, lo, hi: (u ** 2).sum(), shared_kappa=kappa)
        lo, hi = shard_range(len(f), rank, world)
        assert u_local.shape[0] == hi - lo
        t = [torch.tensor([1.0 + rank]), torch.tensor([[2.0, 3.0 * (rank + 1)]])]
        allreduce_sum_fused(t)
        if rank == 0:
            torch.save(dict(loss=float(loss), grad=float(kappa.grad), t0=t[0], t1=t[1]), out)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process_reference_loop(tmp_path):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    g = golden("g9_batch_2d_8")
    mesh = (g["nodes"], g["elements"], g["bc_nodes"], g["bc_vals"])
    k0, B = float(g["kappa"][0]), len(g["f"])
    loss_ref, grad_ref = 0.0, 0.0
    for b in range(B):
        u, dk, _ = orc.solve_with_adjoint(*mesh, k0, g["f"][b], lambda u: 2 * u / B)
        loss_ref += (u ** 2).sum() / B
        grad_ref += dk.sum()
    assert abs(res["loss"] - loss_ref) <= 1e-12 * abs(loss_ref)
    assert abs(res["grad"] - grad_ref) <= 1e-12 * abs(grad_ref)
    assert float(res["t0"]) == 3.0 and res["t1"].tolist() == [[4.0, 9.0]]


def test_single_process_is_a_noop_for_collectives():
    t = torch.tensor([1.0, 2.0])
    allreduce_sum_fused([t])
    assert t.tolist() == [1.0, 2.0]
    drv = ShardedBatchSolve(lambda k, f: f * k)
    assert (drv.world, drv.rank) == (1, 0)
    k = torch.tensor(2.0, requires_grad=True)
    loss, u = drv.step(torch.ones(4, 3), lambda u, lo, hi: u.sum(), shared_kappa=k)
    assert float(loss) == 6.0 and float(k.grad) == 3.0

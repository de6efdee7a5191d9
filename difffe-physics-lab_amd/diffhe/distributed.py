"""Batch sharding of independent solves across the GPUs of one node.

Every sample (kappa_b, f_b) is an independent linear system on a shared mesh, so the
batch is split contiguously over ranks, the mesh plan is replicated, and NO collective
runs on the data path.  The only exchange is one SUM all-reduce per optimisation step of
[loss, shared-kappa gradient] fused in a single buffer (RCCL over xGMI with the "nccl"
backend; "gloo" on CPU for tests).  Per-sample kappa needs no gradient reduction.

The reference has no distributed code at all; the semantics are pinned by fixture G9
(gradient of a shared kappa = sum of the per-sample reference gradients).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of a batch of B for `rank`; sizes differ by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# messages from this size on are reduced as reduce-scatter + all-gather (every rank sums 1/world of the buffer and
# every pair of ranks exchanges 1/world of it: all 7 xGMI links of a GPU carry traffic at once), below it as one
# latency-bound all-reduce
_TWO_PHASE_BYTES = 1 << 20


def allreduce_sum_fused(tensors: Sequence[torch.Tensor], group=None) -> None:
    """In-place SUM all-reduce of several tensors as ONE message (loss scalar + gradient).
    8-16 B (scalar loss, shared scalar kappa): a single all-reduce.  MB-sized (a shared per-element kappa gradient,
    16.8 MB at 1024^2): reduce-scatter + all-gather on the RCCL backend; gloo (CPU tests) has no reduce-scatter and
    keeps the single all-reduce -- same result either way (SUM over ranks)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    nccl = dist.get_backend(group) == "nccl"
    # one staging device for the fused message: RCCL needs device memory; gloo reduces on the host (the loss may
    # live on the GPU while a shared kappa -- and hence its gradient -- is a CPU tensor, as in the reference's API)
    cuda = [t.device for t in tensors if t.is_cuda]
    if nccl and not cuda:
        raise RuntimeError("allreduce_sum_fused over RCCL needs at least one tensor on the GPU")
    target = cuda[0] if nccl else torch.device("cpu")
    flat = torch.cat([t.detach().reshape(-1).to(target) for t in tensors])
    if flat.numel() * flat.element_size() >= _TWO_PHASE_BYTES and nccl:
        n = flat.numel()
        per = (n + world - 1) // world
        padded = torch.zeros(per * world, dtype=flat.dtype, device=flat.device)
        padded[:n] = flat
        mine = torch.empty(per, dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM, group=group)
        dist.all_gather_into_tensor(padded, mine, group=group)
        flat = padded[:n]
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].reshape(t.shape))      # back to each tensor's own device
            off += t.numel()


class ShardedBatchSolve:
    """Data-parallel driver: this rank solves its shard of a global batch.

    `local_solve(kappa, f) -> u` is the per-rank differentiable solve (by default a
    `DifferentiableFESolver` call).  `step(...)` returns the GLOBAL mean loss and leaves
    the correctly reduced gradient in `shared_kappa.grad` when kappa is shared by the batch.
    """

    def __init__(self, local_solve: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], group=None):
        self.local_solve = local_solve
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0

    def shard(self, t: torch.Tensor) -> torch.Tensor:
        lo, hi = shard_range(t.shape[0], self.rank, self.world)
        return t[lo:hi]

    def step(self, f_global: torch.Tensor, loss_fn: Callable[[torch.Tensor, int, int], torch.Tensor],
             shared_kappa: Optional[torch.Tensor] = None, sample_kappa: Optional[torch.Tensor] = None):
        """One fwd + adjoint over the global batch.

        loss_fn(u_local, lo, hi) must return the SUM over this shard's samples; the
        global loss is sum / B.  Returns (loss_global, u_local)."""
        B = f_global.shape[0]
        lo, hi = shard_range(B, self.rank, self.world)
        kappa = shared_kappa if shared_kappa is not None else sample_kappa[lo:hi]
        return self.step_local(f_global[lo:hi], B, lambda u: loss_fn(u, lo, hi), kappa,
                               shared_kappa if shared_kappa is not None else None)

    def step_local(self, f_local: torch.Tensor, B_global: int, loss_sum_fn: Callable[[torch.Tensor], torch.Tensor],
                   kappa: torch.Tensor, shared_kappa: Optional[torch.Tensor] = None):
        """The same step when every rank already holds (only) its shard -- the layout of `bench.py`, where
        the global batch (2048 x 8.4 MB at BASELINE config 4) is never materialised on one GPU.
        loss_sum_fn(u_local) = SUM over this shard's samples.  Returns (loss_global, u_local)."""
        u = self.local_solve(kappa, f_local)
        loss_local = loss_sum_fn(u) / B_global
        loss_local.backward()
        loss = loss_local.detach().clone().reshape(1)
        bufs = [loss]
        if shared_kappa is not None and shared_kappa.grad is not None:
            bufs.append(shared_kappa.grad)
        allreduce_sum_fused(bufs, self.group)
        return loss[0], u

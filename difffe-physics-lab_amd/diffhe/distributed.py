"""Batch sharding of independent solves across the GPUs of one node.

Every sample (kappa_b, f_b) is an independent linear system on a shared mesh, so the
batch is split contiguously over ranks, the mesh plan is replicated, and NO collective
runs on the data path.  The only exchange is the SUM all-reduce per optimisation step of
[loss, shared-kappa gradient] (RCCL over xGMI with the "nccl" backend; "gloo" on CPU for
tests).  Per-sample kappa needs no gradient reduction -- only the scalar loss.

Message sizes and how they travel (`allreduce_sum_fused`):
  * small tensors (the scalar loss, a shared scalar kappa's gradient: 8-16 B) are packed into ONE
    latency-bound all-reduce;
  * a tensor of >= 1 MiB (the gradient of a per-element kappa field shared by the batch: 8 m B = 16.8 MB at
    1024^2, BASELINE config 4's "RCCL grad all-reduce") goes as reduce-scatter + all-gather: every rank sums
    1/world of the buffer and every pair of ranks exchanges 1/world of it, so all 7 xGMI links of a GPU carry
    traffic at once (a ring would be bound by ONE link).  When the length divides the world size both phases
    run straight on the gradient's own storage -- no staging copy, no padding.

The reference has no distributed code at all; the semantics are pinned by fixture G9
(gradient of a shared kappa = sum of the per-sample reference gradients).
"""
from __future__ import annotations

import time
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(B: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of a batch of B for `rank`; sizes differ by at most one."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# tensors from this size on are reduced as reduce-scatter + all-gather, smaller ones ride in one fused all-reduce
_TWO_PHASE_BYTES = 1 << 20


class TorchCollective:
    """The three collectives `allreduce_sum_fused` needs, on a torch.distributed process group.
    Any object with the same attributes (`world`, `two_phase`, `device_only`) and methods can be passed instead
    (the CPU tests inject an in-process fake to run the reduce-scatter branch without RCCL)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        backend = dist.get_backend(group)
        # Capability is decided HERE, once, identically on every rank (same backend, same torch build) -- never by
        # catching a failed collective: after a genuine RCCL failure a second collective on the same communicator
        # hangs, and a rank-local failure would leave the ranks in different collectives.  gloo has no
        # reduce_scatter_tensor: it keeps the single all-reduce.
        self.two_phase = (backend == "nccl" and hasattr(dist, "reduce_scatter_tensor")
                          and hasattr(dist, "all_gather_into_tensor"))
        self.device_only = backend == "nccl"    # RCCL moves device memory only

    def all_reduce(self, flat: torch.Tensor) -> None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        dist.all_gather_into_tensor(out, inp, group=self.group)


def _default_collective(group):
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return TorchCollective(group)


def allreduce_sum_fused(tensors: Sequence[torch.Tensor], group=None, collective=None, stats: Optional[dict] = None,
                        skip_single: bool = True) -> None:
    """In-place SUM all-reduce of several tensors (loss scalar + gradient) in as few messages as their sizes allow
    (module docstring).  `stats`, if given, receives what was sent: bytes, number of collectives, path taken.
    skip_single=False issues the collectives even in a group of one (the GPU test that runs every RCCL call of this
    function on the one GPU of the test box).  Errors of a collective propagate: the process must exit non-zero."""
    coll = collective if collective is not None else _default_collective(group)
    if stats is not None:
        stats.update(bytes=0, collectives=0, path="none", staged_bytes=0)
    if coll is None or (coll.world == 1 and skip_single):
        return
    world = coll.world
    cuda = [t.device for t in tensors if t.is_cuda]
    if coll.device_only and not cuda:
        raise RuntimeError("allreduce_sum_fused over RCCL needs at least one tensor on the GPU")
    # one staging device for what has to be packed: RCCL needs device memory; gloo reduces on the host (the loss
    # may live on the GPU while a shared kappa -- and hence its gradient -- is a CPU tensor, as in the reference's API)
    target = cuda[0] if coll.device_only else torch.device("cpu")
    big = [t for t in tensors if coll.two_phase and t.numel() * t.element_size() >= _TWO_PHASE_BYTES]
    small = [t for t in tensors if not any(t is b for b in big)]
    paths = []
    with torch.no_grad():
        if small:
            flat = torch.cat([t.detach().reshape(-1).to(target) for t in small])
            coll.all_reduce(flat)
            off = 0
            for t in small:
                t.copy_(flat[off:off + t.numel()].reshape(t.shape))      # back to each tensor's own device
                off += t.numel()
            paths.append("all_reduce")
            if stats is not None:
                stats["bytes"] += flat.numel() * flat.element_size()
                stats["collectives"] += 1
        for t in big:
            n = t.numel()
            direct = t.is_contiguous() and t.device == target and n % world == 0
            if direct:
                buf = t.detach().view(-1)                      # the gradient's own storage: nothing is copied
            else:
                per = (n + world - 1) // world
                buf = torch.zeros(per * world, dtype=t.dtype, device=target)
                buf[:n] = t.detach().reshape(-1).to(target)
                if stats is not None:
                    stats["staged_bytes"] += buf.numel() * buf.element_size()
            mine = torch.empty(buf.numel() // world, dtype=buf.dtype, device=buf.device)
            coll.reduce_scatter(mine, buf)                     # this rank's 1/world of the sum
            coll.all_gather(buf, mine)                         # ... handed to everyone
            paths.append("reduce_scatter+all_gather" + ("" if direct else "(staged)"))
            if not direct:
                t.copy_(buf[:n].reshape(t.shape))
            if stats is not None:
                stats["bytes"] += n * t.element_size()
                stats["collectives"] += 2
    if stats is not None:
        stats["path"] = " | ".join(paths)


class ShardedBatchSolve:
    """Data-parallel driver: this rank solves its shard of a global batch.

    `local_solve(kappa, f) -> u` is the per-rank differentiable solve (by default a
    `DifferentiableFESolver` call).  `step(...)` returns the GLOBAL mean loss and leaves
    the correctly reduced gradient in `shared_kappa.grad` when kappa is shared by the batch.
    `last_collective` describes the exchange of the last step (bytes, path, host-side milliseconds).
    """

    def __init__(self, local_solve: Callable[[torch.Tensor, torch.Tensor], torch.Tensor], group=None, collective=None):
        self.local_solve = local_solve
        self.group = group
        self.collective = collective if collective is not None else _default_collective(group)
        self.world = self.collective.world if self.collective is not None else 1
        self.rank = self.collective.rank if self.world > 1 else 0
        self.last_collective: dict = {}

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
                   kappa: torch.Tensor, shared_kappa: Optional[torch.Tensor] = None, time_collective: bool = False):
        """The same step when every rank already holds (only) its shard -- the layout of `bench.py`, where
        the global batch (2048 x 8.4 MB at BASELINE config 4) is never materialised on one GPU.
        loss_sum_fn(u_local) = SUM over this shard's samples.  Returns (loss_global, u_local).
        time_collective: synchronise the device before and after the exchange and record its wall time in
        `last_collective["ms"]` (a measurement aid: it serialises the step)."""
        u = self.local_solve(kappa, f_local)
        loss_local = loss_sum_fn(u) / B_global
        loss_local.backward()
        loss = loss_local.detach().clone().reshape(1)
        bufs = [loss]
        if shared_kappa is not None and shared_kappa.grad is not None:
            bufs.append(shared_kappa.grad)
        stats: dict = {}
        cuda_dev = next((t.device for t in bufs if t.is_cuda), None)
        if time_collective and cuda_dev is not None:
            torch.cuda.synchronize(cuda_dev)
        t0 = time.perf_counter()
        allreduce_sum_fused(bufs, self.group, self.collective, stats)
        if time_collective:
            if cuda_dev is not None:
                torch.cuda.synchronize(cuda_dev)
            stats["ms"] = 1e3 * (time.perf_counter() - t0)
        self.last_collective = stats
        return loss[0], u

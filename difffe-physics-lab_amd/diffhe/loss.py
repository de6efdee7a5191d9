"""Physics-informed losses on top of the HIP solve path.

Mirror of the reference `PhysicsLoss` (reference diffhe/loss.py:21-105): modes
"fem_match" (MSE against the FEM solution, solve under no_grad, loss.py:78-83) and
"variational" (1D finite-difference residual, loss.py:85-105), same constructor and
the same `ValueError` for an unknown mode (loss.py:52-53).

Differences that do not change results: the FEM target is cached while
(forcing values, kappa) are unchanged -- the reference re-solves an identical
system on every call (loss.py:81-82) -- and 2D meshes work (the reference's
`nodes.squeeze(1)` makes 2D fail): for dim 2 the forcing
callable receives the (n,2) coordinates and returns (n,) values.

RHS ensembles (the reference loops `PhysicsLoss` call by call, one identical-matrix solve
each, loss.py:78-83): `forcing_fns=[f_1, ..., f_B]`, or a `forcing_fn` that returns (B, n), makes the FEM target
ONE batched solve of B right-hand sides (cached like the single target).  `forward` then returns the mean of the
B per-member losses -- exactly `mse_loss` over the (B, n) target -- and `member_losses` the B values a loop of
reference calls would produce (fixture G12).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch
import torch.nn as nn

from .mesh import FEMesh
from .solver import DifferentiableFESolver


class PhysicsLoss(nn.Module):
    def __init__(self, mesh: FEMesh, forcing_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None,
                 mode: str = "fem_match", solver: Optional[DifferentiableFESolver] = None, *,
                 forcing_fns: Optional[Sequence[Callable[[torch.Tensor], torch.Tensor]]] = None):
        super().__init__()
        if mode not in ("fem_match", "variational"):
            raise ValueError(f"Unknown mode: {mode!r}")
        if (forcing_fn is None) == (forcing_fns is None):
            raise ValueError("give exactly one of forcing_fn and forcing_fns")
        if forcing_fns is not None:
            if mode != "fem_match":
                raise ValueError("forcing_fns (an ensemble of right-hand sides) needs mode='fem_match'")
            fns = tuple(forcing_fns)
            if not fns:
                raise ValueError("forcing_fns is empty")
            forcing_fn = lambda pts: torch.stack([fn(pts) for fn in fns])    # noqa: E731  (B, n)
        self.mesh = mesh
        self.forcing_fn = forcing_fn
        self.mode = mode
        self.solver = solver or DifferentiableFESolver(mesh)
        self._cache = None  # (f, kappa snapshot, u_fem)

    def forward(self, u_pred: torch.Tensor) -> torch.Tensor:
        if self.mode == "fem_match":
            return self._fem_match_loss(u_pred)
        return self._variational_loss(u_pred)

    def _points(self) -> torch.Tensor:
        return self.mesh.nodes.squeeze(1) if self.mesh.dim == 1 else self.mesh.nodes

    def fem_target(self) -> torch.Tensor:
        """FEM solution for the current forcing; solved once per distinct (f, kappa)."""
        f = self.forcing_fn(self._points())
        kappa = self.solver.kappa.detach()
        c = self._cache
        if (c is not None and c[0].shape == f.shape and torch.equal(c[0], f.detach())
                and torch.equal(c[1], kappa)):
            return c[2]
        with torch.no_grad():
            u_fem = self.solver(f)
        self._cache = (f.detach().clone(), kappa.clone(), u_fem)
        return u_fem

    def _fem_match_loss(self, u_pred: torch.Tensor) -> torch.Tensor:
        u_fem = self.fem_target().to(u_pred.device)
        if u_fem.dim() == 2 and u_pred.dim() == 1:      # one prediction against every member of the ensemble
            u_pred = u_pred.unsqueeze(0).expand_as(u_fem)
        return nn.functional.mse_loss(u_pred.double(), u_fem.double())

    def member_losses(self, u_pred: torch.Tensor) -> torch.Tensor:
        """(B,) fem_match losses, one per member of the ensemble -- what B separate reference
        `PhysicsLoss(mesh, f_b)(u_pred)` calls return (loss.py:78-83); their mean is `forward(u_pred)`."""
        u_fem = self.fem_target().to(u_pred.device).double()
        if u_fem.dim() == 1:
            u_fem = u_fem.unsqueeze(0)
        up = u_pred.double()
        if up.dim() == 1:
            up = up.unsqueeze(0)
        return ((up - u_fem) ** 2).mean(dim=1)

    def _variational_loss(self, u_pred: torch.Tensor) -> torch.Tensor:
        x = self.mesh.nodes.squeeze(1)
        f = self.forcing_fn(x)
        free = torch.as_tensor(self.mesh.free_nodes(), dtype=torch.long, device=u_pred.device)
        x_free = x[free.to(x.device)]
        u_free = u_pred[free].double()
        h = float(x_free[1] - x_free[0]) if len(x_free) > 1 else 1.0
        if len(x_free) >= 3:
            lap = (u_free[:-2] - 2 * u_free[1:-1] + u_free[2:]) / (h ** 2)
            residual = lap + f.to(u_pred.device)[free][1:-1].double()
        else:
            residual = torch.zeros(1, dtype=torch.float64, device=u_pred.device)
        return (residual ** 2).mean()

"""Heat equation on the differentiable P1 solve path (the reference's README roadmap item "heat equation",
README.md:139-143; SURVEY 8(f) rank 4 -- the reference itself has no time-dependent code).

    du/dt - div(kappa grad u) = f,   u = g on the Dirichlet nodes,   u(0) = u0

P1 in space with the LUMPED mass M_L (row sums of the load matrix of reference solver.py:95-96 / :143-145), the
theta scheme in time:

    (M_L / dt + theta K) u_{k+1} = (M_L / dt - (1 - theta) K) u_k + F

theta = 1 (default) is backward Euler: L-stable, first order, every step ONE solve of the reaction-diffusion
system `DifferentiableFESolver(..., reaction=1/dt)` with the extra load M_L u_k / dt -- the same HIP kernels, batch
layout and explicit adjoint as the stationary problem, so whole batches of kappa samples march together and
`backward()` runs the discrete adjoint heat equation (one adjoint solve per step, in reverse).
theta = 1/2 (Crank-Nicolson, second order) is written in its incremental form

    (2 M_L / dt + K) w = 2 M_L u_k / dt + F,   u_{k+1} = 2 w - u_k

(w = (u_k + u_{k+1}) / 2; Dirichlet rows of w hold g, so u_{k+1} = g there when u_k does) -- again one solve per
step, with reaction = 2 / dt, and no product K u_k to form.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .mesh import FEMesh
from .plan import get_plan
from .solver import DifferentiableFESolver, _resolve_device


class HeatEquation(nn.Module):
    """Time stepping of the heat equation on `mesh`.

    Parameters
    ----------
    mesh, kappa : as for `DifferentiableFESolver` (kappa: scalar, (B,), (n_elements,) or (B, n_elements)).
    dt : time step.
    theta : 1.0 (backward Euler) or 0.5 (Crank-Nicolson).
    solver_options : passed on to `DifferentiableFESolver` (device, tol, mg, warm_start, ...); `warm_start=True` is
        the natural choice for small time steps (the previous step IS a good guess) and is the default here.
    """

    def __init__(self, mesh: FEMesh, kappa=1.0, dt: float = 1e-2, theta: float = 1.0, **solver_options):
        super().__init__()
        if not dt > 0.0:
            raise ValueError(f"dt must be > 0, got {dt!r}")
        if theta not in (1.0, 0.5):
            raise ValueError(f"theta must be 1.0 (backward Euler) or 0.5 (Crank-Nicolson), got {theta!r}")
        self.mesh, self.dt, self.theta = mesh, float(dt), float(theta)
        solver_options.setdefault("warm_start", "forward")
        self.solver = DifferentiableFESolver(mesh, kappa, reaction=1.0 / (self.theta * self.dt), **solver_options)
        self._mass = None

    @property
    def kappa(self) -> torch.Tensor:
        return self.solver.kappa

    def lumped_mass(self) -> torch.Tensor:
        """(n,) lumped mass on the solver's device."""
        if self._mass is None:
            self._mass = get_plan(self.mesh, _resolve_device(self.solver._device)).lumped_mass()
        return self._mass

    def step(self, u: torch.Tensor, f: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One time step: u (n,) or (B,n) at t_k -> u at t_k + dt.  f: nodal forcing (n,) or (B,n) at the new time
        level (backward Euler) / at the midpoint (Crank-Nicolson); None = 0."""
        m = self.lumped_mass().to(u.device)
        if f is None:
            f = torch.zeros(self.mesh.n_nodes, dtype=torch.float64, device=u.device)
        load = m * u * self.solver.reaction         # M_L u_k / (theta dt); rows of Dirichlet nodes are ignored
        w = self.solver(f, load=load)
        return w if self.theta == 1.0 else 2.0 * w - u

    def forward(self, u0: torch.Tensor, n_steps: int, f=None, return_all: bool = False) -> torch.Tensor:
        """March `n_steps` steps from u0.  f: None, a tensor (constant in time) or a callable t -> tensor evaluated at
        t_{k+1} (backward Euler) / t_k + dt/2 (Crank-Nicolson).  Returns u(T), or the stacked (n_steps + 1, ...) history."""
        u = u0.to(torch.float64)
        bc = self.mesh.dirichlet_nodes
        if bc:     # the initial state takes the Dirichlet values, like every later one
            idx = torch.as_tensor(list(bc.keys()), dtype=torch.long, device=u.device)
            val = torch.as_tensor(list(bc.values()), dtype=torch.float64, device=u.device)
            u = u.index_copy(-1, idx, val.expand(u.shape[:-1] + val.shape) if u.dim() == 2 else val)
        hist = [u]
        for k in range(n_steps):
            t = (k + self.theta) * self.dt
            fk = f(t) if callable(f) else f
            u = self.step(u, fk)
            if return_all:
                hist.append(u)
        return torch.stack(hist) if return_all else u

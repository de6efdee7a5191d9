"""Small MLP PDE surrogate trained against `PhysicsLoss`.

Mirror of the reference `NeuralPDE` (reference diffhe/neural.py:19-149): fp64
tanh-MLP, Dirichlet handling by a boundary-zero mask, Adam loop `train_pde`.  The
network itself is stock PyTorch (it is not on the accelerated path);
what changes is that the FEM target behind `fem_match` comes from the HIP solver
and is solved once instead of once per epoch.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.nn as nn

from .mesh import FEMesh
from .loss import PhysicsLoss


class NeuralPDE(nn.Module):
    def __init__(self, mesh: FEMesh, hidden_dim: int = 32, n_layers: int = 3):
        super().__init__()
        self.mesh = mesh
        self.dim = mesh.dim
        widths = [self.dim] + [hidden_dim] * n_layers
        layers: List[nn.Module] = []
        for a, b in zip(widths[:-1], widths[1:]):
            layers += [nn.Linear(a, b), nn.Tanh()]
        layers.append(nn.Linear(widths[-1], 1))
        self.net = nn.Sequential(*layers).double()
        self._mask = self._compute_mask()

    def forward(self, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Network value at the mesh nodes (or at `x` (n, dim)), zero on Dirichlet nodes."""
        dev = self.net[0].weight.device          # the module may have been moved (model.cuda()): follow it
        x = (self.mesh.nodes if x is None else x).to(dev, torch.float64)
        raw = self.net(x).squeeze(1)
        if self._mask.device != dev:
            self._mask = self._mask.to(dev)
        return self._mask * raw

    def _compute_mask(self) -> torch.Tensor:
        """0 on Dirichlet nodes, ~1 inside (reference neural.py:80-101): in 1D the
        normalised parabola through the first and last Dirichlet node, in 2D an indicator."""
        nodes = self.mesh.nodes
        bc = list(self.mesh.dirichlet_nodes.keys())
        if self.dim == 1:
            if len(bc) < 2:
                return torch.ones(nodes.shape[0], dtype=torch.float64)
            x = nodes[:, 0]
            lo, hi = float(nodes[bc[0], 0]), float(nodes[bc[-1], 0])
            bump = (x - lo) * (hi - x)
            return bump / (bump.abs().max() + 1e-12)
        mask = torch.ones(nodes.shape[0], dtype=torch.float64)
        if bc:
            mask[torch.as_tensor(bc, dtype=torch.long)] = 0.0
        return mask

    def train_pde(self, forcing_fn: Callable[[torch.Tensor], torch.Tensor], n_epochs: int = 2000,
                  lr: float = 1e-3, mode: str = "fem_match", verbose: bool = True,
                  log_every: int = 200) -> List[float]:
        """Adam on PhysicsLoss(mesh, forcing_fn, mode); returns the loss history."""
        loss_fn = PhysicsLoss(self.mesh, forcing_fn, mode=mode)
        optimiser = torch.optim.Adam(self.parameters(), lr=lr)
        history: List[float] = []
        for epoch in range(1, n_epochs + 1):
            optimiser.zero_grad()
            loss = loss_fn(self.forward())
            loss.backward()
            optimiser.step()
            history.append(float(loss))
            if verbose and epoch % log_every == 0:
                print(f"  Epoch {epoch:5d}  loss = {float(loss):.3e}")
        return history

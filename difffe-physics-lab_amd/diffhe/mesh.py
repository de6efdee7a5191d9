"""P1 meshes in 1D (intervals) and 2D (triangles) with Dirichlet data.

Mirrors the reference container `FEMesh` (reference diffhe/mesh.py:14-143): same
fields, properties, factories, node / element / boundary ordering -- the ordering
is part of the results contract because `u` is indexed by node id.  The factories
are vectorised (the reference builds `rectangle` with Python loops, mesh.py:100-120).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch


@dataclass
class FEMesh:
    """Nodes `(n_nodes, dim)` float64, elements `(n_elements, dim+1)` int64 and a
    `{node: value}` Dirichlet map (reference mesh.py:38-40)."""

    nodes: torch.Tensor
    elements: torch.Tensor
    dirichlet_nodes: Dict[int, float] = field(default_factory=dict)

    @property
    def n_nodes(self) -> int:
        return self.nodes.shape[0]

    @property
    def n_elements(self) -> int:
        return self.elements.shape[0]

    @property
    def dim(self) -> int:
        return self.nodes.shape[1]

    # -- factories ---------------------------------------------------------------
    @classmethod
    def line(cls, n_elements: int = 10, x_left: float = 0.0, x_right: float = 1.0,
             bc_left: Optional[float] = 0.0, bc_right: Optional[float] = 0.0) -> "FEMesh":
        """Uniform chain on [x_left, x_right] (reference mesh.py:58-77)."""
        nodes = torch.linspace(x_left, x_right, n_elements + 1, dtype=torch.float64).unsqueeze(1)
        first = torch.arange(n_elements, dtype=torch.long)
        elements = torch.stack([first, first + 1], dim=1)
        bc: Dict[int, float] = {}
        if bc_left is not None:
            bc[0] = bc_left
        if bc_right is not None:
            bc[n_elements] = bc_right
        return cls(nodes=nodes, elements=elements, dirichlet_nodes=bc)

    @classmethod
    def rectangle(cls, nx: int = 4, ny: int = 4, x_range: Tuple[float, float] = (0.0, 1.0),
                  y_range: Tuple[float, float] = (0.0, 1.0), bc_value: float = 0.0) -> "FEMesh":
        """Uniform grid of right triangles, Dirichlet on the whole boundary
        (reference mesh.py:79-121): node id = row*(nx+1)+col, each quad (a,b,c,d)
        gives [a,b,d] then [b,c,d], boundary found with `np.isclose`."""
        xs = np.linspace(x_range[0], x_range[1], nx + 1)
        ys = np.linspace(y_range[0], y_range[1], ny + 1)
        gx, gy = np.meshgrid(xs, ys)
        coords = np.stack([gx.ravel(), gy.ravel()], axis=1)
        row, col = np.divmod(np.arange(nx * ny, dtype=np.int64), nx)
        a = row * (nx + 1) + col
        tris = np.empty((2 * nx * ny, 3), dtype=np.int64)
        tris[0::2, 0], tris[0::2, 1], tris[0::2, 2] = a, a + 1, a + nx + 1
        tris[1::2, 0], tris[1::2, 1], tris[1::2, 2] = a + 1, a + nx + 2, a + nx + 1
        on_edge = (np.isclose(coords[:, 0], x_range[0]) | np.isclose(coords[:, 0], x_range[1])
                   | np.isclose(coords[:, 1], y_range[0]) | np.isclose(coords[:, 1], y_range[1]))
        bc = dict.fromkeys(np.nonzero(on_edge)[0].tolist(), bc_value)
        return cls(nodes=torch.from_numpy(coords), elements=torch.from_numpy(tris), dirichlet_nodes=bc)

    @classmethod
    def rectangle_p2(cls, nx: int = 4, ny: int = 4, x_range: Tuple[float, float] = (0.0, 1.0),
                     y_range: Tuple[float, float] = (0.0, 1.0), bc_value: float = 0.0) -> "FEMesh":
        """QUADRATIC (P2, 6-node) triangles on the triangulation of `rectangle(nx, ny)` -- ours: "P2 elements" is an
        item of the reference's README roadmap (README.md:139-143), it has no such factory.  Nodes = the
        (2 nx + 1) x (2 ny + 1) lattice of vertices and edge midpoints, id = row * (2 nx + 1) + col (vertices sit at
        even row and column); each quad (a, b, c, d) gives [a, b, d, ab, bd, da] then [b, c, d, bc, cd, db]
        (three vertices, then the midpoints of edges 0-1, 1-2, 2-0); Dirichlet on the whole boundary."""
        Wn = 2 * nx + 1
        xs = np.linspace(x_range[0], x_range[1], Wn)
        ys = np.linspace(y_range[0], y_range[1], 2 * ny + 1)
        gx, gy = np.meshgrid(xs, ys)
        coords = np.stack([gx.ravel(), gy.ravel()], axis=1)
        row, col = np.divmod(np.arange(nx * ny, dtype=np.int64), nx)
        a = 2 * row * Wn + 2 * col                   # vertex a of quad (row, col) in the fine numbering
        b, d, c = a + 2, a + 2 * Wn, a + 2 * Wn + 2
        ab, da, bd, bc, cd = a + 1, a + Wn, a + Wn + 1, a + Wn + 2, a + 2 * Wn + 1
        tris = np.empty((2 * nx * ny, 6), dtype=np.int64)
        tris[0::2] = np.stack([a, b, d, ab, bd, da], axis=1)
        tris[1::2] = np.stack([b, c, d, bc, cd, bd], axis=1)
        on_edge = (np.isclose(coords[:, 0], x_range[0]) | np.isclose(coords[:, 0], x_range[1])
                   | np.isclose(coords[:, 1], y_range[0]) | np.isclose(coords[:, 1], y_range[1]))
        bc_ = dict.fromkeys(np.nonzero(on_edge)[0].tolist(), bc_value)
        return cls(nodes=torch.from_numpy(coords), elements=torch.from_numpy(tris), dirichlet_nodes=bc_)

    # -- convenience ---------------------------------------------------------------
    def free_nodes(self) -> List[int]:
        """Ascending ids of the unconstrained nodes (reference mesh.py:127-129)."""
        keep = np.ones(self.n_nodes, dtype=bool)
        if self.dirichlet_nodes:
            keep[np.fromiter(self.dirichlet_nodes.keys(), dtype=np.int64, count=len(self.dirichlet_nodes))] = False
        return np.nonzero(keep)[0].tolist()

    def h(self) -> float:
        """Smallest element length, 1D only (reference mesh.py:131-136)."""
        if self.dim == 1:
            x = self.nodes[:, 0]
            return float((x[self.elements[:, 1]] - x[self.elements[:, 0]]).abs().min())
        raise NotImplementedError("h() not implemented for dim>1 yet")

    def __repr__(self) -> str:
        return (f"FEMesh(dim={self.dim}, n_nodes={self.n_nodes}, "
                f"n_elements={self.n_elements}, "
                f"n_dirichlet={len(self.dirichlet_nodes)})")

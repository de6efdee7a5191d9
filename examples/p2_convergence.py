#!/usr/bin/env python3
"""P1 against P2 triangles on the HIP solve path (needs an MI355X): -lap u = 2 pi^2 sin(pi x) sin(pi y) on the unit
square, nodal error against the exact solution at equal numbers of unknowns.  `FEMesh.rectangle_p2` is this
implementation's answer to the reference's roadmap item "P2 elements" (README.md:139-143).

    python examples/p2_convergence.py
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import torch  # noqa: E402
from diffhe import FEMesh, DifferentiableFESolver  # noqa: E402


def nodal_error(mesh):
    x = mesh.nodes.cuda()
    exact = torch.sin(math.pi * x[:, 0]) * torch.sin(math.pi * x[:, 1])
    u = DifferentiableFESolver(mesh, 1.0)(2.0 * math.pi ** 2 * exact)
    return float((u - exact).abs().max()), mesh.n_nodes


if __name__ == "__main__":
    print("   N   unknowns   P1 (2N x 2N)   P2 (N x N)    ratio")
    for N in (8, 16, 32, 64, 128):
        e1, n1 = nodal_error(FEMesh.rectangle(2 * N, 2 * N))      # same (2N+1)^2 nodes as the P2 mesh
        e2, n2 = nodal_error(FEMesh.rectangle_p2(N, N))
        assert n1 == n2
        print(f"{N:4d} {n1:10d}   {e1:12.3e} {e2:12.3e} {e1 / e2:8.1f}")

#!/usr/bin/env python3
"""Topology optimisation on the HIP solve path (needs an MI355X): minimise the thermal compliance of a uniformly
heated plate that is cooled through a short sink on its left edge (the reference's README roadmap item
"Topology optimisation demo (minimise compliance)", README.md:141).

Design variable: one density rho in [0, 1] per QUAD of `FEMesh.rectangle` (both of its triangles share it),
conductivity kappa_e = k_min + (1 - k_min) rho^p (SIMP, p = 3), volume fraction mean(rho) <= V.
Objective: C = sum_i u_i ~ F^T u for uniform heating on a uniform mesh; its gradient with respect to the
per-element kappa is what `DifferentiableFESolver` returns through its explicit adjoint (one more solve with
the same operator).  Update: optimality criteria with a density filter (3 x 3 mean), bisection on the
volume multiplier.  Several designs (different volume fractions) are optimised AT ONCE as one batch:
kappa has shape (B, n_elements).

    python examples/topology_optimisation.py [N] [iterations] [warm]

`warm`: every solve starts from the previous design's solution (`warm_start=True`).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from diffhe import FEMesh, DifferentiableFESolver  # noqa: E402

T64 = torch.float64


def sink_mesh(N):
    """Unit square, N x N quads; Dirichlet u = 0 only on the middle fifth of the left edge (the heat sink)."""
    mesh = FEMesh.rectangle(N, N)
    keep = {}
    for node in mesh.dirichlet_nodes:
        x, y = float(mesh.nodes[node, 0]), float(mesh.nodes[node, 1])
        if x < 1e-12 and 0.4 <= y <= 0.6:
            keep[node] = 0.0
    mesh.dirichlet_nodes = keep
    return mesh


def optimise(N=96, iters=40, volumes=(0.3, 0.4, 0.5), p=3.0, k_min=1e-3, device="cuda", verbose=True, warm_start=False):
    mesh = sink_mesh(N)
    B = len(volumes)
    vol = torch.tensor(volumes, dtype=T64, device=device).view(B, 1, 1)
    rho = vol.expand(B, N, N).clone()                      # uniform start at the volume fraction
    f = torch.ones(B, mesh.n_nodes, dtype=T64, device=device)
    history, n_its = [], 0
    t0 = time.perf_counter()
    for it in range(iters):
        rho_f = F.avg_pool2d(F.pad(rho.unsqueeze(1), (1, 1, 1, 1), mode="replicate"), 3, stride=1).squeeze(1)
        rho_f.requires_grad_(True)
        kq = k_min + (1.0 - k_min) * rho_f ** p                         # (B, N, N) per quad
        kappa = kq.reshape(B, N * N, 1).expand(B, N * N, 2).reshape(B, 2 * N * N)   # both triangles of a quad
        solver = DifferentiableFESolver(mesh, kappa, device=device, warm_start=warm_start)
        u = solver(f)
        C = u.sum(dim=1)                                                # thermal compliance (up to h^2)
        C.sum().backward()
        dC = rho_f.grad                                                 # <= 0: more material never hurts
        # sensitivities through the filter (the filter is linear and self-adjoint up to the boundary)
        dC = F.avg_pool2d(F.pad(dC.unsqueeze(1), (1, 1, 1, 1), mode="replicate"), 3, stride=1).squeeze(1)
        lo = torch.full((B, 1, 1), 1e-12, dtype=T64, device=device)
        hi = torch.full((B, 1, 1), 1e12, dtype=T64, device=device)
        for _ in range(60):                                             # bisection on the volume multiplier
            mid = torch.sqrt(lo * hi)
            cand = (rho * torch.sqrt((-dC).clamp_min(0) / mid)).clamp(0.0, 1.0)
            cand = torch.minimum(torch.maximum(cand, rho - 0.2), rho + 0.2)
            too_much = cand.mean(dim=(1, 2), keepdim=True) > vol
            lo = torch.where(too_much, mid, lo)
            hi = torch.where(too_much, hi, mid)
        rho = cand.detach()
        history.append(C.detach().cpu())
        n_its += solver.last_info.iterations + solver.last_info.adj_iterations
        if verbose and (it % 10 == 0 or it == iters - 1):
            print(f"  iteration {it:3d}: compliance " + " ".join(f"{float(c):9.3f}" for c in C)
                  + "   volume " + " ".join(f"{float(v):.3f}" for v in rho.mean(dim=(1, 2))))
    dt = time.perf_counter() - t0
    if verbose:
        print(f"  mean CG iterations per step (forward + adjoint): {n_its / iters:.1f}")
    return rho, torch.stack(history), dt


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rho, hist, dt = optimise(N, iters, warm_start="warm" in sys.argv[3:])
    print(f"{N}x{N} quads, {hist.shape[1]} designs at once, {iters} iterations in {dt:.1f} s "
          f"({iters * hist.shape[1] / dt:.0f} differentiable solves/s); compliance "
          + ", ".join(f"{float(a):.2f} -> {float(b):.2f}" for a, b in zip(hist[0], hist[-1])))
    rows = ["".join(" .:-=+*#%@"[min(9, int(10 * float(v)))] for v in row[:: max(1, N // 64)]) for row in rho[1].flip(0)[:: max(1, N // 32)]]
    print("\n".join(rows))

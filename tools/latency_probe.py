#!/usr/bin/env python3
"""Latency of ONE unbatched differentiable solve (the reference's own call shape, solver.py:54: f (n,), scalar kappa):
forward + backward of L = sum u^2 through the Python boundary, wall time per call after warm-up."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import torch  # noqa: E402
from diffhe import FEMesh, DifferentiableFESolver, _hip  # noqa: E402
import ctypes  # noqa: E402

dev = "cuda:0"
for name, mesh in (("1D 20 (config 1)", FEMesh.line(20)), ("1D 10000", FEMesh.line(10000)),
                   ("2D 32^2", FEMesh.rectangle(32, 32)), ("2D 256^2", FEMesh.rectangle(256, 256)),
                   ("2D 1024^2", FEMesh.rectangle(1024, 1024))):
    k = torch.tensor(1.3, dtype=torch.float64, device=dev, requires_grad=True)
    f = torch.ones(mesh.n_nodes, dtype=torch.float64, device=dev)
    solver = DifferentiableFESolver(mesh, k, device=dev)

    def call():
        k.grad = None
        u = solver(f)
        (u ** 2).sum().backward()

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    nl = ctypes.c_longlong(0)
    _hip.lib().diffhe_traffic_account(1, None, None)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    _hip.lib().diffhe_traffic_account(1, None, ctypes.byref(nl))
    info = solver.last_info
    print(f"{name:18s} {1e3 * dt:8.3f} ms per fwd+bwd  ({nl.value / reps:7.0f} accounted launches, path {info.path}, "
          f"its {info.iterations}+{info.adj_iterations})")

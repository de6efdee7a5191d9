#!/usr/bin/env python3
"""Development probe: per-sample error of the bench workload against the EXACT solution of the assembled system.
On rectangle(N, N) with N a power of two the mesh size h = 1/N is a power of two, every assembled entry is exact and
the reference's matrix is exactly kappa_b * (5-point Laplacian): DST-I gives its exact solution (scipy, CPU)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import torch  # noqa: E402
from scipy.fft import dstn, idstn  # noqa: E402
from diffhe import FEMesh, DifferentiableFESolver  # noqa: E402
from oracle import p1_oracle as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
mesh = FEMesh.rectangle(N, N)
n = mesh.n_nodes
gen = torch.Generator().manual_seed(4096)
kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=torch.float64)).to(dev).requires_grad_(True)
f = torch.ones(B, n, dtype=torch.float64, device=dev)
bn = np.fromiter(mesh.dirichlet_nodes.keys(), dtype=np.int64)
F = orc.load_vector(mesh.nodes.numpy(), mesh.elements.numpy(), np.ones(n)).reshape(N + 1, N + 1)[1:-1, 1:-1]
k = np.arange(1, N)
lam = 4.0 - 2.0 * np.cos(np.pi * k / N)[:, None] - 2.0 * np.cos(np.pi * k / N)[None, :]
u1 = np.zeros((N + 1, N + 1))
u1[1:-1, 1:-1] = idstn(dstn(F, type=1) / lam, type=1)
u1 = torch.from_numpy(u1.ravel()).to(dev)
# adjoint of L = mean_b sum u^2: lambda_b = K_b^-1 (2 u_b / B); dL/dkappa_b = -2 L_b / kappa_b / B exactly
import warnings  # noqa: E402
warnings.simplefilter("ignore")
rough = len(sys.argv) > 3 and sys.argv[3] == "rough"
if rough:                                  # random forcing: exact solutions of a few samples via DST on the CPU
    gen2 = torch.Generator().manual_seed(77)
    f = (1 + 0.5 * torch.randn(B, n, generator=gen2, dtype=torch.float64)).to(dev)
    chk = [0, B // 2, B - 1]
    ue_rows = {}
    for b_ in chk:
        Fb = orc.load_vector(mesh.nodes.numpy(), mesh.elements.numpy(), f[b_].cpu().numpy()).reshape(N + 1, N + 1)[1:-1, 1:-1]
        ub = np.zeros((N + 1, N + 1))
        ub[1:-1, 1:-1] = idstn(dstn(Fb, type=1) / lam, type=1) / float(kappa[b_])
        ue_rows[b_] = torch.from_numpy(ub.ravel()).to(dev)
cases = [("default", {})] + [(f"max_iter {k}", dict(max_iter=k)) for k in (3, 4, 5, 6, 7)] + \
    [(f"tol_energy {t:g}", dict(mg=dict(tol_energy=t))) for t in (1e-10, 1e-11, 1e-12)]
for label, kw in cases:
    kappa.grad = None
    solver = DifferentiableFESolver(mesh, kappa, device=dev, **kw)
    u = solver(f)
    L = (u ** 2).sum(dim=1)
    L.mean().backward()
    info = solver.last_info
    if rough:
        err = np.array([float((u[b_].detach() - ue_rows[b_]).abs().max() / ue_rows[b_].abs().max()) for b_ in chk])
        gerr = np.zeros(1)
    else:
        ue = u1[None, :] / kappa.detach()[:, None]
        err = ((u.detach() - ue).abs().max(dim=1).values / ue.abs().max(dim=1).values).cpu().numpy()
        Le = (ue ** 2).sum(dim=1)
        gref = -2.0 * Le / kappa.detach() / B
        gerr = ((kappa.grad - gref).abs() / gref.abs()).cpu().numpy()
    print(f"{label:18s} its {info.iterations}+{info.adj_iterations} relres {info.max_relres:.1e}/{info.adj_max_relres:.1e} "
          f"est {info.err_est:.1e}/{info.adj_err_est:.1e} | u err vs exact: max {err.max():.2e} | dkappa err max {gerr.max():.2e}")

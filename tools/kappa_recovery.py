#!/usr/bin/env python3
"""BASELINE config 5 on one GPU's shard: recover per-sample kappa on a 512 x 512 mesh by 100 Adam steps
through the adjoint solve (64 samples per GPU; examples/poisson_1d_demo.py:102-110 generalised).
`python tools/kappa_recovery.py warm|forward` starts every solve / every forward solve from the previous step's
solution (warm_start=True / "forward")."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import torch  # noqa: E402
from diffhe import FEMesh, DifferentiableFESolver  # noqa: E402

N, B, STEPS = 512, 64, 100
WARM = True if "warm" in sys.argv[1:] else ("forward" if "forward" in sys.argv[1:] else False)
dev = torch.device("cuda", 0)
mesh = FEMesh.rectangle(N, N)
gen = torch.Generator().manual_seed(5)
k_true = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=torch.float64)).to(dev)
f = torch.ones(B, mesh.n_nodes, dtype=torch.float64, device=dev)
with torch.no_grad():
    u_data = DifferentiableFESolver(mesh, k_true)(f)
k = torch.ones(B, dtype=torch.float64, device=dev, requires_grad=True)
opt = torch.optim.Adam([k], lr=0.1)
scale = 1.0 / float((u_data ** 2).mean())
its = [0, 0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for step in range(STEPS):
    opt.zero_grad()
    solver = DifferentiableFESolver(mesh, k.abs(), warm_start=WARM)
    u = solver(f)
    loss = ((u - u_data) ** 2).mean(dim=1).sum() * scale
    loss.backward()
    opt.step()
    its[0] += solver.last_info.iterations
    its[1] += solver.last_info.adj_iterations
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"config 5 shard: {N}x{N}, {B} samples, {STEPS} Adam steps in {dt:.2f} s = {STEPS / dt:.1f} steps/s "
      f"({STEPS * B / dt:.0f} differentiable solves/s); max |kappa - kappa_true| = "
      f"{float((k.detach().abs() - k_true).abs().max()):.2e}, final loss {float(loss.detach()):.2e}; "
      f"mean iterations {its[0] / STEPS:.1f}+{its[1] / STEPS:.1f}" + (f" (warm_start={WARM!r})" if WARM else ""))

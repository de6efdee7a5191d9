"""Unstructured-mesh (aggregation-AMG PCG) timing: jittered N x N triangulation, batch of per-sample kappa.

    python tools/amg_bench.py [N] [batch] [gamma] [scale]
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "difffe-physics-lab_amd"))
import numpy as np, torch
import diffhe
from diffhe import FEMesh, DifferentiableFESolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
gammas = [int(sys.argv[3])] if len(sys.argv) > 3 else [1, 2]
scales = [float(sys.argv[4])] if len(sys.argv) > 4 else [None]   # None: the solver's default (1.3 smoothed, 1.8 piecewise constant)
m = FEMesh.rectangle(N, N)
rng = np.random.default_rng(0)
nodes = m.nodes.numpy().copy()
h = 1.0 / N
interior = (nodes[:, 0] > 1e-9) & (nodes[:, 0] < 1 - 1e-9) & (nodes[:, 1] > 1e-9) & (nodes[:, 1] < 1 - 1e-9)
nodes[interior] += rng.uniform(-0.25 * h, 0.25 * h, (int(interior.sum()), 2))
mesh = FEMesh(nodes=torch.from_numpy(nodes), elements=m.elements, dirichlet_nodes=dict(m.dirichlet_nodes))
kappa = torch.from_numpy(rng.uniform(0.5, 2.0, B)).cuda()
f = torch.ones(B, mesh.n_nodes, dtype=torch.float64, device="cuda")
for gamma in gammas:
    for scale in scales:
        s = DifferentiableFESolver(mesh, kappa, device="cuda", method="ell", operator=os.environ.get("AMG_BENCH_OPERATOR", "auto"))
        s.amg.update(gamma=gamma or None, scale=scale)   # 0: the solver's own choice
        print("pass_bytes", mesh.n_nodes * B * 8)
        t0 = time.time(); u = s(f); torch.cuda.synchronize(); t_first = time.time() - t0
        t0 = time.time(); u = s(f); torch.cuda.synchronize(); t = time.time() - t0
        print(f"N={N} B={B} gamma={gamma} scale={scale}: its={s.last_info.iterations} relres={s.last_info.max_relres:.2e} path={s.last_info.path} "
              f"first={t_first:.2f}s steady={t*1e3:.1f} ms  ({B/t:.1f} solves/s) factored={s.last_info.factored} "
              f"ms_per_iteration={t*1e3/max(s.last_info.iterations,1):.3f}", flush=True)

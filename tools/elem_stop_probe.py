#!/usr/bin/env python3
"""Development aid: error of u and dL/dkappa_e (per-element log-normal field per sample) against the refined oracle as a
function of the energy-norm tolerance, 512^2 (config 3 field variant) and a rough-data case (random forcing)."""
import concurrent.futures as cf
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def job(a):
    from oracle import p1_oracle as orc
    nodes, el, bn, bv, kap, f, scale = a
    u, dk, _ = orc.solve_with_adjoint(nodes, el, bn, bv, kap, f, lambda u_: scale * u_, sparse=True, refine=2)
    return u, dk


def main():
    from diffhe import FEMesh, DifferentiableFESolver
    shape = os.environ.get("PROBE_N", "512").split("x")
    nx, ny = int(shape[0]), int(shape[-1])
    N, B = f"{nx}x{ny}", 64
    dev = "cuda:0"
    mesh = FEMesh.rectangle(nx, ny)
    n, m = mesh.n_nodes, mesh.n_elements
    g = torch.Generator(device=dev).manual_seed(2025)
    sigma = float(os.environ.get("PROBE_SIGMA", 0.3))
    kappa = torch.exp(sigma * torch.randn(B, m, generator=g, dtype=torch.float64, device=dev))
    cases = {"f=1": torch.ones(B, n, dtype=torch.float64, device=dev),
             "f=1+0.5randn": 1 + 0.5 * torch.randn(B, n, generator=g, dtype=torch.float64, device=dev)}
    bn = np.fromiter(mesh.dirichlet_nodes.keys(), dtype=np.int64)
    bv = np.fromiter(mesh.dirichlet_nodes.values(), dtype=np.float64)
    idx = [0, 31, 63]
    for name, f in cases.items():
        jobs = [(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, kappa[b].cpu().numpy(), f[b].cpu().numpy(), 2.0 / B) for b in idx]
        with cf.ProcessPoolExecutor(len(jobs), mp_context=mp.get_context("spawn")) as ex:
            ref = list(ex.map(job, jobs))
        for te in (1e-10, 1e-11, 1e-12, 1e-13, 1e-14):
            k = kappa.clone().requires_grad_(True)
            s = DifferentiableFESolver(mesh, k, device=dev, mg=dict(tol_energy=te))
            u = s(f)
            ((u ** 2).sum(dim=1)).mean().backward()
            eu = max(float(np.max(np.abs(u[b].detach().cpu().numpy() - r[0])) / np.max(np.abs(r[0]))) for b, r in zip(idx, ref))
            eg = max(float(np.max(np.abs(k.grad[b].cpu().numpy() - r[1])) / np.max(np.abs(r[1]))) for b, r in zip(idx, ref))
            i = s.last_info
            print(f"{name:14s} N={N} tol_energy={te:7.0e}: its {i.iterations}+{i.adj_iterations}  u err {eu:.2e}  dkappa_e err {eg:.2e}  "
                  f"est {i.err_est:.1e}/{i.adj_err_est:.1e} stop {i.stop_rules}", flush=True)


if __name__ == "__main__":
    main()

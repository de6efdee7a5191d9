#!/usr/bin/env python3
"""Development aid: error of u and dL/dkappa_e (one log-normal kappa field per sample) against the refined oracle as a
function of the energy-norm tolerance, on several lattice shapes / contrasts / forcings.  Prints the ratio of the
measured max-norm errors to the solver's own estimate of the final iterate (0.3 x the CG iterate's estimate)."""
import concurrent.futures as cf
import multiprocessing as mp
import os
import sys
import warnings

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
    warnings.simplefilter("ignore")
    dev = "cuda:0"
    B = 64
    cases = [(512, 512, 0.3, "one"), (288, 296, 0.3, "rand"), (202, 70, 0.3, "rand"), (512, 512, 1.0, "rand"),
             (320, 300, 1.5, "rand"), (1024, 1024, 0.3, "one")]
    for nx, ny, sigma, fk in cases:
        mesh = FEMesh.rectangle(nx, ny)
        n, m = mesh.n_nodes, mesh.n_elements
        g = torch.Generator(device=dev).manual_seed(2025)
        kappa = torch.exp(sigma * torch.randn(B, m, generator=g, dtype=torch.float64, device=dev))
        f = torch.ones(B, n, dtype=torch.float64, device=dev) if fk == "one" else \
            1 + 0.5 * torch.randn(B, n, generator=g, dtype=torch.float64, device=dev)
        bn = np.fromiter(mesh.dirichlet_nodes.keys(), dtype=np.int64)
        bv = np.fromiter(mesh.dirichlet_nodes.values(), dtype=np.float64)
        idx = [0, 63]
        jobs = [(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, kappa[b].cpu().numpy(), f[b].cpu().numpy(), 2.0 / B) for b in idx]
        with cf.ProcessPoolExecutor(len(jobs), mp_context=mp.get_context("spawn")) as ex:
            ref = list(ex.map(job, jobs))
        for te in (1e-10, 1e-11, 1e-12, 1e-13):
            k = kappa.clone().requires_grad_(True)
            s = DifferentiableFESolver(mesh, k, device=dev, mg=dict(tol_energy=te), operator="assembled")
            u = s(f)
            ((u ** 2).sum(dim=1)).mean().backward()
            eu = max(float(np.max(np.abs(u[b].detach().cpu().numpy() - r[0])) / np.max(np.abs(r[0]))) for b, r in zip(idx, ref))
            eg = max(float(np.max(np.abs(k.grad[b].cpu().numpy() - r[1])) / np.max(np.abs(r[1]))) for b, r in zip(idx, ref))
            i = s.last_info
            est = 0.3 * max(i.err_est, i.adj_err_est)
            print(f"{nx}x{ny} sigma={sigma} f={fk:4s} tol_energy={te:7.0e}: its {i.iterations}+{i.adj_iterations}  u {eu:.1e}  dk_e {eg:.1e}  "
                  f"est_final {est:.1e}  dk_e/est {eg / est:6.1f}  {i.stop_rules}", flush=True)


if __name__ == "__main__":
    main()

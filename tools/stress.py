#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle (development aid; needs a GPU).
Varies mesh kind/size, Dirichlet sets, kappa layout, batch size and checks u, dL/dkappa, dL/df."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "difffe-physics-lab_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from diffhe import FEMesh, DifferentiableFESolver  # noqa: E402
from oracle import p1_oracle as orc  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
LARGE = len(sys.argv) > 3 and sys.argv[3] == "large"      # strip-kernel sizes, wave-sized batches, 3 samples checked
worst = 0.0
for case in range(n_cases):
    kind = rng.choice(["line", "rect", "rect_bc", "unstructured"])
    if kind == "line":
        N = int(rng.integers(2, 400)) if not LARGE else int(rng.integers(2000, 12000))
        bl, br = rng.choice([None, 0.0, 1.3]), rng.choice([0.0, -0.7])
        mesh = FEMesh.line(N, -1.0, 2.0, bl, br)
        if rng.random() < 0.5:     # graded node positions and a few interior Dirichlet nodes (chain segments)
            x = mesh.nodes.numpy()[:, 0].copy()
            h = x[1] - x[0]
            x[1:-1] += rng.uniform(-0.3, 0.3, N - 1) * h
            d = dict(mesh.dirichlet_nodes)
            for k in rng.choice(np.arange(1, N), size=min(N - 1, int(rng.integers(0, 4))), replace=False):
                d[int(k)] = float(rng.uniform(-1, 1))
            mesh = FEMesh(torch.from_numpy(x[:, None]), mesh.elements, d)
    else:
        nx, ny = (int(rng.integers(2, 70)), int(rng.integers(2, 70))) if not LARGE else \
            (int(rng.integers(190, 420)), int(rng.integers(64, 300)))
        mesh = FEMesh.rectangle(nx, ny, (0.0, float(rng.uniform(0.5, 6.0))), (0.0, 1.0), float(rng.uniform(-1, 1)))
        if rng.random() < 0.5:     # skewed lattice: jittered interior nodes (4 diagonals, some obtuse triangles)
            xy = mesh.nodes.numpy().copy().reshape(ny + 1, nx + 1, 2)
            hx, hy = xy[0, 1, 0] - xy[0, 0, 0], xy[1, 0, 1] - xy[0, 0, 1]
            xy[1:-1, 1:-1] += rng.uniform(-0.25, 0.25, (ny - 1, nx - 1, 2)) * np.array([hx, hy])
            mesh = FEMesh(torch.from_numpy(xy.reshape(-1, 2)), mesh.elements, dict(mesh.dirichlet_nodes))
        if kind == "rect" and rng.random() < 0.3:   # pinned interior nodes (0.05 % .. 5 % of the nodes)
            frac = 10 ** rng.uniform(-3.3, -1.3)
            d = dict(mesh.dirichlet_nodes)
            for k in rng.choice(mesh.n_nodes, max(1, int(frac * mesh.n_nodes)), replace=False):
                d[int(k)] = float(rng.uniform(-1, 1))
            mesh = FEMesh(mesh.nodes, mesh.elements, d)
        if kind == "rect_bc":      # Dirichlet only on part of the boundary (+ one interior node): Neumann elsewhere
            keys = list(mesh.dirichlet_nodes)
            keep = [k for k in keys if abs(float(mesh.nodes[k, 0])) < 1e-12] + [int(rng.integers(0, mesh.n_nodes))]
            mesh.dirichlet_nodes = {k: float(rng.uniform(-1, 1)) for k in keep}
        if kind == "unstructured":
            perm = rng.permutation(mesh.n_nodes)
            nodes = np.empty_like(mesh.nodes.numpy())
            nodes[perm] = mesh.nodes.numpy()
            el = perm[mesh.elements.numpy()][rng.permutation(mesh.n_elements)]
            mesh = FEMesh(torch.from_numpy(nodes), torch.from_numpy(el),
                          {int(perm[k]): v for k, v in mesh.dirichlet_nodes.items()})
    n, m = mesh.n_nodes, mesh.n_elements
    B = int(rng.choice([1, 2, 3, 17, 64, 70])) if not LARGE else int(rng.choice([64, 100, 128, 192]))
    kmode = rng.choice(["scalar", "sample", "elem", "sample_elem"])
    if kmode in ("sample", "sample_elem") and B == m:
        B += 1                     # (B,) and (m,) kappa must be distinguishable by shape
    kap = {"scalar": np.array(rng.uniform(0.5, 2.0)), "sample": rng.uniform(0.5, 2.0, B),
           "elem": np.exp(0.4 * rng.standard_normal(m)), "sample_elem": np.exp(0.4 * rng.standard_normal((B, m)))}[kmode]
    f = 1 + 0.5 * rng.standard_normal((B, n))
    if rng.random() < 0.3:         # forcing amplitudes spread over 12 decades across the batch (per-sample stops)
        f = f * (10.0 ** rng.uniform(-6, 6, (B, 1)))
    if kmode in ("elem", "sample_elem") and rng.random() < 0.3:   # high-contrast coefficient field (e^-4 .. e^4)
        kap = kap ** 3.3
    if os.environ.get("STRESS_ONLY") and case != int(os.environ["STRESS_ONLY"]):
        continue                   # every random draw of the case has been made: the stream stays in step
    kt = torch.from_numpy(np.atleast_1d(kap) if kmode != "scalar" else kap).requires_grad_(True)
    ft = torch.from_numpy(f).requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(ft)
    (u ** 2).sum().backward()
    bn = np.array(list(mesh.dirichlet_nodes.keys()), dtype=np.int64)
    bv = np.array(list(mesh.dirichlet_nodes.values()))
    errs = []
    dk_ref = np.zeros_like(np.atleast_1d(kap), dtype=np.float64) if kmode != "scalar" else 0.0
    dk_scale = np.zeros_like(np.atleast_1d(kap), dtype=np.float64) if kmode != "scalar" else 0.0   # sum |contributions|
    check = range(B) if not LARGE else sorted({0, B // 2, B - 1})
    if LARGE and kmode in ("scalar", "elem"):
        kmode_grad_partial = True       # gradient of a shared kappa sums over ALL samples: only compare u and df
    else:
        kmode_grad_partial = False
    for b in check:
        kb = kap if kmode in ("scalar", "elem") else kap[b]
        if kind == "line" and n > 1500:   # fp64 LU is itself ~cond*eps off there: use the extended-precision oracle
            uo, dko, dfo = orc.chain_solve_longdouble(mesh.nodes.numpy(), bn, bv, kb, f[b], lambda u: 2 * u,
                                                      reference_rounding=True)   # default chain mode: the reference's system
            cnd = np.abs(dko)
        else:
            uo, dko, dfo, cnd = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, kb, f[b],
                                                       lambda u: 2 * u, with_cond=True)
        sc = max(np.max(np.abs(uo)), 1e-300)
        errs.append(np.max(np.abs(u[b].detach().numpy() - uo)) / sc)
        errs.append(np.max(np.abs(ft.grad[b].numpy() - dfo)) / max(np.max(np.abs(dfo)), 1e-300))
        if kmode == "scalar":
            dk_ref += dko.sum()
            dk_scale += cnd.sum()
        elif kmode == "sample":
            dk_ref[b] = dko.sum()
            dk_scale[b] = cnd.sum()
        elif kmode == "elem":
            dk_ref += dko
            dk_scale += cnd
        else:
            dk_ref[b] = dko
            dk_scale[b] = cnd
    if not kmode_grad_partial:
        got = kt.grad.numpy()
        if LARGE:
            got, dk_ref, dk_scale = got[list(check)], dk_ref[list(check)], dk_scale[list(check)]
        # Gradients are judged against the magnitude of the terms they are made of (sum |lambda||k0||u|): with tiny
        # forcing and constant Dirichlet data u is nearly constant, the terms cancel, and dL/dkappa is only defined
        # to u * that magnitude in fp64 (the oracle's own value moves by 1e-8 relative there).  Where nothing cancels
        # the two measures coincide.
        if kmode in ("elem", "sample_elem"):
            errs.append(np.max(np.abs(got - dk_ref)) / max(np.max(np.abs(dk_ref)), np.max(dk_scale), 1e-300))
        else:
            errs.append(float(np.max(np.abs(got - dk_ref) / np.maximum(np.atleast_1d(dk_scale), 1e-300))))
    e = max(errs)
    if os.environ.get("STRESS_ONLY"):
        print("per-check errors", ["%.1e" % v for v in errs])
        print(solver.last_info, "n_bc", len(bn), "tol", solver.tol)
    worst = max(worst, e)
    flag = "" if e < 1e-10 else "   <-- ABOVE 1e-10"
    extra = ""
    if os.environ.get("STRESS_VERBOSE"):
        from diffhe.plan import get_plan
        pl = get_plan(mesh, torch.device("cuda:0"))
        xy = mesh.nodes.numpy()
        extra = (f"  closed={getattr(pl, 'closed_boundary', None)} regular={getattr(pl, 'regular_cells', None)} "
                 f"extent={np.ptp(xy, axis=0)} adj_iters={solver.last_info.adj_iterations} rules={solver.last_info.stop_rules}")
    print(f"case {case:3d} {kind:12s} n={n:5d} B={B:3d} kappa={kmode:11s} path={solver.last_info.path:14s} "
          f"iters={solver.last_info.iterations:4d} err={e:.1e}{flag}{extra}", flush=True)
print("worst", worst)

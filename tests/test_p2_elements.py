"""Quadratic (P2, 6-node) triangles through the general HIP path -- `FEMesh.rectangle_p2`, the reference's README
roadmap item "P2 elements" (README.md:139-143).  The reference has no quadratic element: the
checks are against `oracle/p2_oracle.py` (own quadrature, pinned to closed forms in tests/test_oracle_golden.py),
exact reproduction of quadratic solutions and the third-order L2 convergence that distinguishes P2 from P1."""
import numpy as np
import pytest
import torch

from diffhe import FEMesh, DifferentiableFESolver
from oracle import p2_oracle as p2
from _util import rel_err, RTOL_U, RTOL_GRAD

T64 = torch.float64


def _arrays(mesh):
    bn = np.fromiter(mesh.dirichlet_nodes.keys(), dtype=np.int64)
    bv = np.fromiter(mesh.dirichlet_nodes.values(), dtype=np.float64)
    return mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv


def test_p2_mesh_factory_layout():
    mesh = FEMesh.rectangle_p2(3, 2, (0.0, 1.5), (0.0, 1.0), 0.25)
    assert mesh.dim == 2 and mesh.n_nodes == 7 * 5 and mesh.n_elements == 12 and mesh.elements.shape == (12, 6)
    x = mesh.nodes.numpy()
    el = mesh.elements.numpy()
    for a, b, mid in ((0, 1, 3), (1, 2, 4), (2, 0, 5)):          # midpoints sit in the middle of their edges
        assert np.allclose(x[el[:, mid]], 0.5 * (x[el[:, a]] + x[el[:, b]]))
    assert len(mesh.dirichlet_nodes) == 2 * 7 + 2 * 5 - 4 and set(mesh.dirichlet_nodes.values()) == {0.25}
    assert sorted(set(el.ravel().tolist())) == list(range(35))    # every lattice point is a vertex or an edge midpoint


@pytest.mark.gpu
@pytest.mark.parametrize("kmode", ["scalar", "sample", "elem", "sample_elem"])
def test_p2_matches_oracle(kmode):
    """u, dL/dkappa and dL/df of L = sum u^2 on P2 triangles against the CPU oracle, every kappa layout, non-zero
    Dirichlet value, random forcing."""
    mesh = FEMesh.rectangle_p2(9, 7, (0.0, 1.5), (0.0, 1.0), 0.3)
    nodes, el, bn, bv = _arrays(mesh)
    n, m, B = mesh.n_nodes, mesh.n_elements, 3
    rng = np.random.default_rng(17)
    kap = {"scalar": np.array(1.3), "sample": rng.uniform(0.5, 2.0, B), "elem": np.exp(0.4 * rng.standard_normal(m)),
           "sample_elem": np.exp(0.4 * rng.standard_normal((B, m)))}[kmode]
    f = 1.0 + 0.5 * rng.standard_normal((B, n))
    kt = torch.from_numpy(np.asarray(kap)).cuda().requires_grad_(True)
    ft = torch.from_numpy(f).cuda().requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(ft)
    (u ** 2).sum().backward()
    assert solver.last_info.path.startswith("ell-") and solver.last_info.not_converged == 0
    dk_ref = np.zeros_like(np.atleast_1d(kap), dtype=np.float64)
    for b in range(B):
        prob = p2.P2Problem(nodes, el, bn, bv, kap if kmode in ("scalar", "elem") else kap[b])
        uo = prob.solve(f[b])
        dko, dfo = prob.adjoint(uo, 2.0 * uo)
        assert rel_err(u[b].detach().cpu().numpy(), uo) < RTOL_U
        assert rel_err(ft.grad[b].cpu().numpy(), dfo) < RTOL_GRAD
        if kmode == "scalar":
            dk_ref += dko.sum()
        elif kmode == "sample":
            dk_ref[b] = dko.sum()
        elif kmode == "elem":
            dk_ref += dko
        else:
            dk_ref[b] = dko
    assert rel_err(kt.grad.cpu().numpy().reshape(dk_ref.shape), dk_ref) < RTOL_GRAD


@pytest.mark.gpu
def test_p2_exact_on_quadratics_and_third_order_in_l2():
    """A quadratic solution is reproduced to solver accuracy (it lies in the P2 space), and for
    -lap u = 2 pi^2 sin(pi x) sin(pi y) the L2 error falls by ~8 per halving of h (the P1 path: ~4)."""
    mesh = FEMesh.rectangle_p2(12, 10, (0.0, 1.5), (0.0, 1.0))
    x = mesh.nodes.numpy()
    ue = x[:, 0] * (1.5 - x[:, 0]) / 2 + x[:, 1] * (1 - x[:, 1])
    mesh.dirichlet_nodes = {k: float(ue[k]) for k in mesh.dirichlet_nodes}
    u = DifferentiableFESolver(mesh, 1.7)(torch.full((mesh.n_nodes,), 3.0 * 1.7, dtype=T64, device="cuda"))
    assert rel_err(u.cpu().numpy(), ue) < 1e-11
    exact = lambda x_, y_: np.sin(np.pi * x_) * np.sin(np.pi * y_)      # noqa: E731
    errs = []
    for N in (8, 16, 32):
        mesh = FEMesh.rectangle_p2(N, N)
        nodes, el, bn, bv = _arrays(mesh)
        f = torch.from_numpy(2 * np.pi ** 2 * exact(nodes[:, 0], nodes[:, 1])).cuda()
        uh = DifferentiableFESolver(mesh, 1.0)(f).cpu().numpy()
        errs.append(p2.P2Problem(nodes, el, bn, bv, 1.0).l2_error(uh, exact))
    print("P2 L2 errors", ["%.2e" % e for e in errs])
    assert 6.5 < errs[0] / errs[1] < 10.0 and 6.5 < errs[1] / errs[2] < 10.0


@pytest.mark.gpu
def test_p2_batch_at_a_larger_size():
    """128 x 128 quads of P2 triangles (66 049 nodes, 19 entries per row), 64 kappa samples through the
    aggregation-multigrid PCG: first and last sample against the oracle, forward and dL/dkappa."""
    mesh = FEMesh.rectangle_p2(128, 128)
    nodes, el, bn, bv = _arrays(mesh)
    B = 64
    gen = torch.Generator().manual_seed(9)
    kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)).cuda().requires_grad_(True)
    f = torch.ones(B, mesh.n_nodes, dtype=T64, device="cuda")
    solver = DifferentiableFESolver(mesh, kappa)
    u = solver(f)
    (u ** 2).sum().backward()
    info = solver.last_info
    print(f"P2 128^2 x {B}: path {info.path}, iterations {info.iterations}+{info.adj_iterations}")
    assert info.not_converged == 0
    for b in (0, B - 1):
        prob = p2.P2Problem(nodes, el, bn, bv, float(kappa[b].detach()))
        uo = prob.solve(np.ones(mesh.n_nodes))
        dko, _ = prob.adjoint(uo, 2.0 * uo)
        assert rel_err(u[b].detach().cpu().numpy(), uo) < RTOL_U
        assert abs(float(kappa.grad[b]) - dko.sum()) < RTOL_GRAD * abs(dko.sum())


@pytest.mark.gpu
def test_p2_on_a_distorted_renumbered_mesh_with_partial_dirichlet_boundary():
    """Not only the factory's lattice: vertices jittered (edge midpoints moved with them, sides stay straight), nodes and
    elements in random order, Dirichlet data on the left edge only (Neumann elsewhere), one kappa field per sample."""
    base = FEMesh.rectangle_p2(11, 8, (0.0, 1.4), (0.0, 1.0))
    rng = np.random.default_rng(23)
    x = base.nodes.numpy().copy()
    el = base.elements.numpy()
    Wn, Hn = 2 * 11 + 1, 2 * 8 + 1
    grid = x.reshape(Hn, Wn, 2)
    hx, hy = 1.4 / 11, 1.0 / 8
    grid[2:-2:2, 2:-2:2] += rng.uniform(-0.2, 0.2, grid[2:-2:2, 2:-2:2].shape) * np.array([hx, hy])   # interior vertices
    for a, b, mid in ((0, 1, 3), (1, 2, 4), (2, 0, 5)):
        x[el[:, mid]] = 0.5 * (x[el[:, a]] + x[el[:, b]])
    perm = rng.permutation(len(x))
    nodes = np.empty_like(x)
    nodes[perm] = x
    el2 = perm[el][rng.permutation(len(el))]
    left = [int(perm[k]) for k in base.dirichlet_nodes if abs(float(base.nodes[k, 0])) < 1e-12]
    mesh = FEMesh(torch.from_numpy(nodes), torch.from_numpy(el2), {k: 0.4 for k in left})
    bn, bv = np.asarray(left, dtype=np.int64), np.full(len(left), 0.4)
    B, m, n = 5, mesh.n_elements, mesh.n_nodes
    kap = np.exp(0.4 * rng.standard_normal((B, m)))
    f = 1.0 + 0.5 * rng.standard_normal((B, n))
    kt = torch.from_numpy(kap).cuda().requires_grad_(True)
    ft = torch.from_numpy(f).cuda().requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(ft)
    (u ** 2).sum().backward()
    assert solver.last_info.not_converged == 0
    for b in (0, B - 1):
        prob = p2.P2Problem(nodes, el2, bn, bv, kap[b])
        uo = prob.solve(f[b])
        dko, dfo = prob.adjoint(uo, 2.0 * uo)
        assert rel_err(u[b].detach().cpu().numpy(), uo) < RTOL_U
        assert rel_err(kt.grad[b].cpu().numpy(), dko) < RTOL_GRAD
        assert rel_err(ft.grad[b].cpu().numpy(), dfo) < RTOL_GRAD

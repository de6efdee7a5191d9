"""Failure surfacing, determinism, thread safety and layout equivalence of the HIP solve path (SURVEY section 5:
the reference silently returns garbage on singular systems, solver.py:174; it has no threads and no global state).
All tests here run the real kernels through the C ABI (`-m gpu`)."""
import threading
import warnings

import numpy as np
import pytest
import torch

from diffhe import FEMesh, DifferentiableFESolver
from diffhe.plan import get_plan
from oracle import p1_oracle as orc
from _util import rel_err, RTOL_U, RTOL_GRAD

pytestmark = pytest.mark.gpu
T64 = torch.float64
DEV = "cuda:0"


def _unstructured(nx=40, ny=36, seed=3):
    """rectangle(nx, ny) with jittered interior nodes, nodes renumbered at random: a general (ELL) mesh."""
    base = FEMesh.rectangle(nx, ny)
    rng = np.random.default_rng(seed)
    nodes = base.nodes.numpy().copy()
    n = len(nodes)
    interior = np.array([i for i in range(n) if i not in base.dirichlet_nodes])
    nodes[interior] += rng.uniform(-0.2, 0.2, size=(len(interior), 2)) / max(nx, ny)
    perm = rng.permutation(n)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    return FEMesh(nodes=torch.from_numpy(nodes[perm]), elements=torch.from_numpy(inv[base.elements.numpy()]),
                  dirichlet_nodes={int(inv[k]): v for k, v in base.dirichlet_nodes.items()})


def _run(mesh, kappa0, f0, **kw):
    kappa = kappa0.clone().to(DEV).requires_grad_(True)
    f = f0.clone().to(DEV).requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kappa, device=DEV, **kw)
    u = solver(f)
    (u ** 2).sum().backward()
    return u.detach(), kappa.grad.detach(), f.grad.detach(), solver.last_info


# ---- failure surfacing ------------------------------------------------------------------------------------------
def test_iteration_cap_is_surfaced_not_silent():
    """max_iter = 3 on 256^2: the CG cannot finish -- RuntimeWarning, last_info.not_converged > 0, the 'cap' bucket of
    the stop-rule counts filled, and the output still finite (the best iterate so far)."""
    mesh = FEMesh.rectangle(256, 256)
    B = 64
    gen = torch.Generator().manual_seed(1)
    kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    solver = DifferentiableFESolver(mesh, kappa.to(DEV), device=DEV, max_iter=3)
    with pytest.warns(RuntimeWarning, match="did not reach tol"):
        u = solver(f.to(DEV))
    info = solver.last_info
    assert info.path == "lattice-mgpcg" and info.iterations == 3
    assert info.not_converged > 0 and info.stop_rules["cap"] == info.not_converged
    assert bool(torch.isfinite(u).all())
    # the same call without the cap converges, and says which rule ended each sample
    solver2 = DifferentiableFESolver(mesh, kappa.to(DEV), device=DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        solver2(f.to(DEV))
    r = solver2.last_info.stop_rules
    assert solver2.last_info.not_converged == 0 and r["cap"] == 0 and r["residual"] + r["energy"] == B


def test_general_path_iteration_cap_is_surfaced():
    mesh = _unstructured()
    solver = DifferentiableFESolver(mesh, 1.3, device=DEV, max_iter=2)
    with pytest.warns(RuntimeWarning, match="did not reach tol"):
        u = solver(torch.ones(4, mesh.n_nodes, dtype=T64, device=DEV))
    assert solver.last_info.not_converged == 4 and bool(torch.isfinite(u).all())


def test_pure_neumann_problem_warns_instead_of_returning_garbage_silently():
    """No Dirichlet node: K is singular.  The reference returns values of size 1e15 without a word (solver.py:174);
    here the 1D scan returns NaN and a RuntimeWarning names the cause; the 2D path warns too and reports the
    unconverged systems."""
    mesh = FEMesh.line(50, bc_left=None, bc_right=None)
    solver = DifferentiableFESolver(mesh, 1.0, device=DEV)
    with pytest.warns(RuntimeWarning, match="singular"):
        u = solver(torch.ones(mesh.n_nodes, dtype=T64, device=DEV))
    assert bool(torch.isnan(u).all())
    base = FEMesh.rectangle(24, 20)
    mesh2 = FEMesh(nodes=base.nodes, elements=base.elements, dirichlet_nodes={})
    solver2 = DifferentiableFESolver(mesh2, 1.0, device=DEV, max_iter=50)
    with pytest.warns(RuntimeWarning, match="singular"):
        solver2(torch.ones(2, mesh2.n_nodes, dtype=T64, device=DEV))


def test_explicit_tolerance_is_honoured_over_the_energy_stop():
    """tol= is a request on the RESIDUAL: the energy-norm stop (default on closed lattices, ends a solve at relative
    residuals up to ~3e-7) must step aside unless tol_energy is asked for as well."""
    mesh = FEMesh.rectangle(320, 300)
    B = 64
    gen = torch.Generator().manual_seed(5)
    kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)).to(DEV)
    f = (1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)).to(DEV)
    dflt = DifferentiableFESolver(mesh, kappa, device=DEV)
    dflt(f)
    assert dflt.last_info.stop_rules["energy"] > 0 and dflt.last_info.tol_energy > 0
    for tol in (1e-9, 1e-12):
        s = DifferentiableFESolver(mesh, kappa, device=DEV, tol=tol)
        s(f)
        info = s.last_info
        assert info.tol_energy == 0.0 and info.stop_rules["energy"] == 0 and info.not_converged == 0
        assert info.max_relres <= max(2.0 * tol, 1e-10), (tol, info)     # true residual; 1e-10: what fp64 can attain here
    both = DifferentiableFESolver(mesh, kappa, device=DEV, tol=1e-13, mg=dict(tol_energy=1e-11))
    both(f)
    assert both.last_info.tol_energy == 1e-11 and both.last_info.stop_rules["energy"] > 0


def test_energy_stop_is_off_on_a_lattice_whose_boundary_is_partly_neumann_even_with_pinned_interior_nodes():
    """`closed` used to be n_bc >= 2 (nx + ny), which pinned INTERIOR nodes could satisfy on a mesh whose
    outer boundary is mostly Neumann -- the regime where the energy estimate undershoots the nodal error 40x."""
    nx, ny = 200, 180
    base = FEMesh.rectangle(nx, ny)
    W = nx + 1
    bc = {i * W: 0.25 for i in range(ny + 1)}                                   # left edge only
    blk = [(i, j) for i in range(60, 60 + 26) for j in range(100, 100 + 26)]    # 676 pinned interior nodes (1.9 % of n)
    bc.update({i * W + j: 0.5 for i, j in blk})
    mesh = FEMesh(nodes=base.nodes, elements=base.elements, dirichlet_nodes=dict(sorted(bc.items())))
    plan = get_plan(mesh, torch.device(DEV))
    assert plan.n_bc >= 2 * (nx + ny) and not plan.closed_boundary
    kappa = torch.tensor([0.8, 1.9], dtype=T64)
    gen = torch.Generator().manual_seed(2)
    f = 1 + 0.5 * torch.randn(2, mesh.n_nodes, generator=gen, dtype=T64)
    u, gk, gf, info = _run(mesh, kappa, f)
    assert info.path == "lattice-mgpcg" and info.tol_energy == 0.0 and info.stop_rules["energy"] == 0
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in range(2):
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, float(kappa[b]), f[b].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=2)
        assert rel_err(u[b].cpu().numpy(), uo) < RTOL_U
        assert abs(float(gk[b]) - dk.sum()) < RTOL_GRAD * abs(dk.sum())
        assert rel_err(gf[b].cpu().numpy(), df) < RTOL_GRAD


# ---- determinism --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["lattice", "lattice-element", "general", "chain"])
def test_two_runs_are_bitwise_identical(kind):
    """Default (gather) path: no float atomics, fixed-order reductions -- u, dL/dkappa and dL/df of two runs are equal
    bit for bit."""
    gen = torch.Generator().manual_seed(11)
    if kind.startswith("lattice"):
        mesh, B = FEMesh.rectangle(224, 200), 64
    elif kind == "general":
        mesh, B = _unstructured(48, 44), 16
    else:
        mesh, B = FEMesh.line(3000, bc_left=1.0, bc_right=2.0), 32
    if kind == "lattice-element":
        kappa = torch.exp(0.3 * torch.randn(B, mesh.n_elements, generator=gen, dtype=T64))
    else:
        kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    a = _run(mesh, kappa, f)
    b = _run(mesh, kappa, f)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    assert a[3].iterations == b[3].iterations and a[3].not_converged == 0


# ---- two threads on one mesh ----------------------------------------------------------------------------------------
def test_forward_and_backward_of_two_solves_on_one_mesh_overlap_safely():
    """Two threads drive independent solves on the SAME mesh (same plan): each one's backward overlaps the other's
    forward.  Status buffers are per calling thread and the plan's caches are locked, so both get exactly what they
    get alone."""
    mesh = FEMesh.rectangle(160, 144)
    B = 64
    gen = torch.Generator().manual_seed(21)
    ks = [0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64) for _ in range(2)]
    fs = [1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64) for _ in range(2)]
    alone = [_run(mesh, ks[i], fs[i]) for i in range(2)]
    got, errs = [[], []], []
    gate = threading.Barrier(2)

    def body(i):
        try:
            torch.cuda.set_device(0)
            gate.wait()
            for _ in range(6):
                got[i].append(_run(mesh, ks[i], fs[i]))
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            gate.abort()

    ts = [threading.Thread(target=body, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    for i in range(2):
        for r in got[i]:
            assert torch.equal(r[0], alone[i][0]) and torch.equal(r[1], alone[i][1]) and torch.equal(r[2], alone[i][2])
            assert r[3].iterations == alone[i][3].iterations and r[3].not_converged == 0


# ---- second-order derivatives ---------------------------------------------------------------------------------------
def _dense_torch_solve(mesh, kappa_e, f):
    """The reference's formulation in differentiable dense torch on the CPU (assembly solver.py:86-96 / :125-145, row
    replacement for Dirichlet nodes :157-172, torch.linalg.solve :174): any-order autograd, the yardstick for Hessians."""
    X, el = mesh.nodes.to(T64).reshape(mesh.n_nodes, -1), mesh.elements.long()
    n, m = mesh.n_nodes, el.shape[0]
    if X.shape[1] == 1:
        h = (X[el[:, 1], 0] - X[el[:, 0], 0]).abs()
        k0 = torch.tensor([[1.0, -1.0], [-1.0, 1.0]], dtype=T64)[None] / h[:, None, None]
        m0 = torch.eye(2, dtype=T64)[None] * (0.5 * h)[:, None, None]
    else:
        x, y = X[el, 0], X[el, 1]
        b = torch.stack([y[:, 1] - y[:, 2], y[:, 2] - y[:, 0], y[:, 0] - y[:, 1]], 1)
        c = torch.stack([x[:, 2] - x[:, 1], x[:, 0] - x[:, 2], x[:, 1] - x[:, 0]], 1)
        area = 0.5 * ((x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])).abs()
        k0 = (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (4 * area)[:, None, None]
        m0 = (area / 9)[:, None, None].expand(-1, 3, 3)
    npe = el.shape[1]
    rows = el[:, :, None].expand(m, npe, npe).reshape(-1)
    cols = el[:, None, :].expand(m, npe, npe).reshape(-1)
    K = torch.zeros(n * n, dtype=T64).index_add(0, rows * n + cols, (kappa_e[:, None, None] * k0).reshape(-1)).reshape(n, n)
    F = torch.zeros(n, dtype=T64).index_add(0, el.reshape(-1), torch.einsum("epq,eq->ep", m0, f[el]).reshape(-1))
    bc = torch.tensor(sorted(mesh.dirichlet_nodes), dtype=torch.long)
    keep = torch.ones(n, dtype=T64); keep[bc] = 0
    eye_bc = torch.zeros(n, n, dtype=T64); eye_bc[bc, bc] = 1
    vals = torch.zeros(n, dtype=T64); vals[bc] = torch.tensor([mesh.dirichlet_nodes[int(i)] for i in bc], dtype=T64)
    return torch.linalg.solve(K * keep[:, None] + eye_bc, F * keep + vals)


@pytest.mark.parametrize("name,mesh", [("chain", FEMesh.line(40, bc_right=0.3)), ("lattice", FEMesh.rectangle(12, 10, bc_value=0.2)),
                                       ("general", _unstructured(10, 9, seed=11))])
def test_hessian_vector_products_match_dense_autograd(name, mesh):
    """backward(create_graph=True): the reference is differentiable to any order (autograd through torch.linalg.solve,
    solver.py:174); the second-order path here re-states the adjoint with differentiable solves.  d/dkappa and d/df of
    <dL/dkappa, v> + <dL/df, w> against double backward through the dense formulation, per-element kappa."""
    g = torch.Generator().manual_seed(5)
    m, n = mesh.n_elements, mesh.n_nodes
    k0 = torch.rand(m, generator=g, dtype=T64) + 0.5
    f0 = torch.rand(n, generator=g, dtype=T64) + 0.5
    v, w = torch.randn(m, generator=g, dtype=T64), torch.randn(n, generator=g, dtype=T64)

    def second(solve, dev):
        kap, f = k0.clone().to(dev).requires_grad_(True), f0.clone().to(dev).requires_grad_(True)
        u = solve(kap, f)
        L = (u ** 3).sum()                                   # not quadratic: d2L/du2 depends on u
        gk, gf = torch.autograd.grad(L, (kap, f), create_graph=True)
        assert gk.requires_grad and gf.requires_grad
        hk, hf = torch.autograd.grad((gk * v.to(dev)).sum() + (gf * w.to(dev)).sum(), (kap, f))
        return [t.detach().cpu() for t in (gk, gf, hk, hf)]

    ref = second(lambda kap, f: _dense_torch_solve(mesh, kap, f), "cpu")
    ours = second(lambda kap, f: DifferentiableFESolver(mesh, kap, device=DEV, tol=1e-14)(f), DEV)
    for tag, a, b in zip(("dL/dkappa", "dL/df", "H.kappa", "H.f"), ours, ref):
        assert rel_err(a.numpy(), b.numpy()) < 1e-9, (name, tag, rel_err(a.numpy(), b.numpy()))


def test_hessian_vector_product_of_a_batch_with_scalar_kappa_and_node_layout():
    """Per-sample scalar kappa (factored operator), (n, B) layout, a `load` term: second derivatives against central
    differences of the FIRST-order (explicit-adjoint) gradient."""
    mesh, B = FEMesh.rectangle(24, 20, bc_value=0.1), 64
    g = torch.Generator().manual_seed(9)
    k0 = (torch.rand(B, generator=g, dtype=T64) + 0.5).to(DEV)
    f = torch.ones(mesh.n_nodes, B, dtype=T64, device=DEV)
    load = (0.01 * torch.rand(mesh.n_nodes, B, generator=g, dtype=T64)).to(DEV)
    v = torch.randn(B, generator=g, dtype=T64).to(DEV)

    def grad(kv, create_graph=False):
        kap = kv.clone().requires_grad_(True)
        u = DifferentiableFESolver(mesh, kap, device=DEV, tol=1e-14)(f, load=load, layout="node")
        gk, = torch.autograd.grad((u ** 3).sum(), kap, create_graph=create_graph)
        return kap, gk

    kap, gk = grad(k0, create_graph=True)
    hv, = torch.autograd.grad((gk * v).sum(), kap)
    _, g_first = grad(k0)
    assert rel_err(gk.detach().cpu().numpy(), g_first.cpu().numpy()) < 1e-10
    eps = 1e-5
    fd = (grad(k0 + eps * v)[1] - grad(k0 - eps * v)[1]) / (2 * eps)      # H v for this per-sample (diagonal) Hessian
    assert rel_err(hv.cpu().numpy(), fd.cpu().numpy()) < 1e-7


def test_first_order_gradients_on_a_retained_graph_repeat_exactly():
    mesh = FEMesh.rectangle(72, 64)
    kappa = torch.tensor(1.4, dtype=T64, device=DEV, requires_grad=True)
    f = torch.ones(mesh.n_nodes, dtype=T64, device=DEV)
    u = DifferentiableFESolver(mesh, kappa, device=DEV)(f)
    L = (u ** 2).sum()
    g1, = torch.autograd.grad(L, kappa, retain_graph=True)
    g2, = torch.autograd.grad(L, kappa)
    assert torch.equal(g1, g2) and abs(float(g1) + 2 * float(L) / 1.4) < 1e-10 * abs(float(g1))
    with pytest.raises(RuntimeError):                        # graph (and the adjoint state) released by the second call
        torch.autograd.grad(L, kappa)


def test_adjoint_state_is_released_with_the_backward_pass_not_with_the_outputs():
    """The saved operators / iterates go when autograd drops its saved tensors, while u and the loss are still held."""
    from diffhe import solver as S
    mesh = FEMesh.rectangle(64, 64)
    kappa = (torch.rand(64, mesh.n_elements, dtype=T64, device=DEV) + 0.5).requires_grad_(True)
    u = DifferentiableFESolver(mesh, kappa, device=DEV)(torch.ones(64, mesh.n_nodes, dtype=T64, device=DEV))
    L = (u ** 2).sum()
    n0 = len(S._STATES)
    assert n0 >= 1
    L.backward()
    assert len(S._STATES) == n0 - 1 and u.grad_fn is not None


# ---- fp32 step length of the CG (cgstep2_kernel) -----------------------------------------------------------------------
def test_fp32_step_length_is_used_only_where_multigrid_is_fast_and_changes_nothing_there():
    """Flag bit 8 of diffhe_lattice_pcg_solve: the host sets it for lattices closed by Dirichlet data with near-square
    cells and a full hierarchy.  There the two-samples-per-lane CG step (p.Ap from a packed-fp32 stencil -- the step
    length only) gives the iteration counts and, to 1e-10, the results of the fp64 one-sample step."""
    dev = torch.device(DEV)
    good = get_plan(FEMesh.rectangle(256, 192), dev)
    assert good.closed_boundary and good.regular_cells and good.dense_level() is not None
    stretched = get_plan(FEMesh.rectangle(256, 192, x_range=(0.0, 6.0)), dev)
    assert stretched.closed_boundary and not stretched.regular_cells
    assert not get_plan(_skewed(64, 64, seed=1), dev).regular_cells
    shallow = get_plan(FEMesh.rectangle(382, 259), dev)                   # coarsens once: no small coarsest level
    assert shallow.closed_boundary and shallow.regular_cells and shallow.dense_level() is None
    open_edge = FEMesh.rectangle(64, 64)
    d = {k: v for k, v in open_edge.dirichlet_nodes.items() if k > 64}    # bottom edge left free
    assert not get_plan(FEMesh(nodes=open_edge.nodes, elements=open_edge.elements, dirichlet_nodes=d), dev).closed_boundary

    mesh, B = FEMesh.rectangle(256, 192, bc_value=0.1), 128
    g = torch.Generator().manual_seed(21)
    k0 = torch.rand(B, generator=g, dtype=T64) * 1.5 + 0.5
    f0 = torch.rand(B, mesh.n_nodes, generator=g, dtype=T64) + 0.5
    a = _run(mesh, k0, f0)
    b = _run(mesh, k0, f0, mg=dict(cg_fp32_steplength=0))
    assert a[3].iterations == b[3].iterations and a[3].adj_iterations == b[3].adj_iterations and a[3].not_converged == 0
    for x, y in zip(a[:3], b[:3]):
        assert rel_err(x.cpu().numpy(), y.cpu().numpy()) < 1e-10


# ---- coarsest-level dense solve on the matrix cores ---------------------------------------------------------------
@pytest.mark.parametrize("mesh_fn,B", [(lambda: FEMesh.rectangle(256, 192, bc_value=0.25), 128), (lambda: FEMesh.rectangle(320, 288), 64)])
def test_mfma_coarse_solve_agrees_with_the_scalar_kernel_and_the_oracle(mesh_fn, B):
    """mg_dense_mfma_kernel (fp32 accumulation on v_mfma_f32_32x32x2_f32, A operand read through the symmetry of the
    inverse) against the scalar-load kernel (fp64 accumulation): same iteration count, same u and dL/dkappa to 1e-10,
    both within the parity tolerance of the oracle on two samples."""
    mesh = mesh_fn()
    g = torch.Generator().manual_seed(12)
    k0 = torch.rand(B, generator=g, dtype=T64) * 1.5 + 0.5
    f0 = torch.rand(B, mesh.n_nodes, generator=g, dtype=T64) + 0.5
    out = {}
    for tag, mg in (("mfma", None), ("scalar", dict(dense_mfma=0))):
        out[tag] = _run(mesh, k0, f0, mg=mg)
    um, gm, fm, im = out["mfma"]
    us, gs, fs, is_ = out["scalar"]
    assert im.path == "lattice-mgpcg" and im.iterations == is_.iterations and im.not_converged == 0
    assert rel_err(um.cpu().numpy(), us.cpu().numpy()) < 1e-10 and rel_err(gm.cpu().numpy(), gs.cpu().numpy()) < 1e-10
    bn = np.array(list(mesh.dirichlet_nodes.keys()))
    bv = np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, B - 1):
        uo, dk, dfo = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, float(k0[b]), f0[b].numpy(),
                                             lambda u_: 2 * u_, sparse=True)
        assert rel_err(um[b].cpu().numpy(), uo) < RTOL_U
        assert abs(float(gm[b]) - dk.sum()) < RTOL_GRAD * abs(dk.sum())
        assert rel_err(fm[b].cpu().numpy(), dfo) < RTOL_GRAD


# ---- lattice form of the gather assembly ---------------------------------------------------------------------------
@pytest.mark.parametrize("mesh_fn,B", [(lambda: FEMesh.rectangle(37, 29, bc_value=0.3), 64), (lambda: _skewed(40, 33, seed=2), 128),
                                       (lambda: FEMesh.rectangle(8, 5), 3)])
def test_lattice_assembly_without_lists_is_bitwise_the_list_driven_gather(mesh_fn, B):
    """diffhe_lattice_assemble_rows (contribution lists written into the kernel) against diffhe_ell_assemble_rows with
    the lattice lists: stored diagonals and Dirichlet lift bit for bit, per-sample fields and a shared field, a mesh
    with interior Dirichlet nodes included."""
    from diffhe import _hip
    from diffhe.plan import padded_batch
    mesh = mesh_fn()
    d = dict(mesh.dirichlet_nodes)
    d[mesh.n_nodes // 2 + 3] = -0.7                       # an interior Dirichlet node: lower-triangle lift terms
    mesh = FEMesh(nodes=mesh.nodes, elements=mesh.elements, dirichlet_nodes=d)
    plan = get_plan(mesh, torch.device(DEV))
    assert plan.is_lattice
    lev = plan.levels[0]
    L = _hip.lib()
    Bp = padded_batch(B)
    g = torch.Generator().manual_seed(3)
    st = torch.cuda.current_stream(DEV).cuda_stream
    for Bv, kse, ksb, kap in ((Bp, Bp, 1, (torch.rand(lev.m, Bp, generator=g, dtype=T64) + 0.5).to(DEV)),
                              (1, 1, 0, (torch.rand(lev.m, 1, generator=g, dtype=T64) + 0.5).to(DEV))):
        out = []
        for which in (0, 1):
            v = torch.full((lev.nd, lev.n, Bv), float("nan"), dtype=T64, device=DEV)
            lf = torch.full((lev.n, Bv), float("nan"), dtype=T64, device=DEV)
            if which == 0:
                _hip.check(L.diffhe_ell_assemble_rows(_hip.ptr(lev.k0), _hip.ptr(kap), kse, ksb, _hip.ptr(lev.ent_ptr),
                                                      _hip.ptr(lev.contrib), _hip.ptr(lev.cols), _hip.ptr(lev.store_slot),
                                                      _hip.ptr(lev.is_bc), _hip.ptr(plan.g), _hip.ptr(v), _hip.ptr(lf), lev.n,
                                                      lev.m, 7, Bv, st), "lists")
            else:
                _hip.check(L.diffhe_lattice_assemble_rows(_hip.ptr(lev.k0), 0, _hip.ptr(kap), kse, ksb, _hip.ptr(lev.is_bc),
                                                          _hip.ptr(plan.g), _hip.ptr(v), _hip.ptr(lf), lev.nx, lev.ny, lev.nd,
                                                          Bv, st), "lattice")
            torch.cuda.synchronize()
            out.append((v, lf))
        assert not torch.isnan(out[1][0]).any() and not torch.isnan(out[1][1]).any()
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        assert float(out[1][1].abs().max()) > 0            # the lift is exercised


# ---- node-major entry -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,bc_value", [(64, 0.0), (5, 0.5), (128, 0.5)])
@pytest.mark.parametrize("per_element", [False, True])
def test_node_major_layout_gives_the_same_numbers(B, bc_value, per_element):
    """forward(f, layout='node') takes and returns (n, B): the same kernels on the same data as the (B, n) API, minus the
    transposing passes -- u, dL/dkappa and dL/df are equal bit for bit (padded batch, non-zero Dirichlet data and the
    unpadded zero-copy case)."""
    mesh = FEMesh.rectangle(208, 196, bc_value=bc_value)
    n = mesh.n_nodes
    gen = torch.Generator().manual_seed(B)
    kappa = (torch.exp(0.3 * torch.randn(B, mesh.n_elements, generator=gen, dtype=T64)) if per_element
             else 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64))
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    u_s, gk_s, gf_s, info_s = _run(mesh, kappa, f)
    kap = kappa.clone().to(DEV).requires_grad_(True)
    f_nm = f.t().contiguous().to(DEV).requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kap, device=DEV)
    u = solver(f_nm, layout="node")
    assert u.shape == (n, B)
    (u ** 2).sum().backward()
    assert torch.equal(u.detach().t(), u_s) and torch.equal(kap.grad, gk_s) and torch.equal(f_nm.grad.t(), gf_s)
    assert solver.last_info.iterations == info_s.iterations
    if bc_value == 0.0 and B == 64:
        # zero Dirichlet data, no padding: u is the solver's own iterate -- an in-place edit before backward is refused
        u2 = DifferentiableFESolver(mesh, kap, device=DEV)(f_nm, layout="node")
        u2.mul_(2.0)
        with pytest.raises(RuntimeError, match="modified by an inplace operation"):
            u2.sum().backward()
    with pytest.raises(ValueError, match="layout='node'"):
        solver(f.to(DEV), layout="node")                       # (B, n) given where (n, B) is expected


def test_node_major_layout_on_the_general_path_and_in_1d():
    mesh = _unstructured(44, 40)
    B = 8
    gen = torch.Generator().manual_seed(4)
    kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    u_s, gk_s, gf_s, _ = _run(mesh, kappa, f)
    kap = kappa.clone().to(DEV).requires_grad_(True)
    f_nm = f.t().contiguous().to(DEV).requires_grad_(True)
    u = DifferentiableFESolver(mesh, kap, device=DEV)(f_nm, layout="node")
    (u ** 2).sum().backward()
    assert torch.equal(u.detach().t(), u_s) and torch.equal(kap.grad, gk_s) and torch.equal(f_nm.grad.t(), gf_s)
    line = FEMesh.line(200, bc_left=1.0, bc_right=None)
    f1 = 1 + 0.5 * torch.randn(3, line.n_nodes, generator=gen, dtype=T64).to(DEV)
    s1 = DifferentiableFESolver(line, 1.7, device=DEV)
    assert torch.equal(s1(f1.t().contiguous(), layout="node").t(), s1(f1))


# ---- a kappa field shared by the batch: gradient summed in the kernel ---------------------------------------------------
@pytest.mark.parametrize("mesh_kind", ["lattice", "general", "chain"])
def test_shared_field_gradient_is_the_sum_of_the_per_sample_gradients(mesh_kind):
    """kappa (m,) shared by B samples: dL/dkappa_e = sum_b dL/dkappa_e,b.  The batch sum happens inside
    diffhe_p1_grad_kappa_shared (no (m, B) intermediate); compared with the per-sample-field run of the same data and
    with the oracle's loop of reference-order solves."""
    gen = torch.Generator().manual_seed(9)
    if mesh_kind == "lattice":
        mesh, B = FEMesh.rectangle(200, 196, bc_value=0.3), 70          # padded batch: 70 -> 128
    elif mesh_kind == "general":
        mesh, B = _unstructured(40, 36), 5
    else:
        mesh, B = FEMesh.line(500, bc_left=0.0, bc_right=1.0), 9
    m, n = mesh.n_elements, mesh.n_nodes
    field = torch.exp(0.3 * torch.randn(m, generator=gen, dtype=T64))
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    u, gk, gf, info = _run(mesh, field, f)
    assert gk.shape == (m,) and info.not_converged == 0
    u2, gk2, _, _ = _run(mesh, field.unsqueeze(0).expand(B, m).contiguous(), f)
    # the two runs assemble their matrices in different operation orders (reference order / kappa_e * fl(t / den)): cond * eps
    assert rel_err(gk.cpu().numpy(), gk2.sum(dim=0).cpu().numpy()) < 2e-11
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    ref = np.zeros(m)
    for b in range(min(B, 5)):
        uo, dk, _ = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, field.numpy(), f[b].numpy(),
                                           lambda u_: 2 * u_, sparse=n > 1500, refine=1)
        assert rel_err(u[b].cpu().numpy(), uo) < RTOL_U
        ref += dk
    if B <= 5:
        assert rel_err(gk.cpu().numpy(), ref) < RTOL_GRAD


# ---- two-samples-per-lane strip kernels (fp32 V-cycle, batch-shared matrix, batch a multiple of 128) ---------------------
def _skewed(nx, ny, seed=0):
    """rectangle connectivity with jittered interior nodes: obtuse triangles, 4 stored diagonals, lattice path."""
    base = FEMesh.rectangle(nx, ny, x_range=(0.0, 1.3), y_range=(0.0, 0.9), bc_value=0.2)
    rng = np.random.default_rng(seed)
    nodes = base.nodes.numpy().copy().reshape(ny + 1, nx + 1, 2)
    nodes[1:-1, 1:-1] += rng.uniform(-0.25, 0.25, size=(ny - 1, nx - 1, 2)) * np.array([1.3 / nx, 0.9 / ny])
    return FEMesh(nodes=torch.from_numpy(nodes.reshape(-1, 2)), elements=base.elements, dirichlet_nodes=base.dirichlet_nodes)


def _partly_neumann(nx, ny):
    base = FEMesh.rectangle(nx, ny)
    W = nx + 1
    bc = {i * W: 0.5 for i in range(ny + 1)}
    bc.update({j: 0.0 for j in range(W)})                        # left and bottom edges Dirichlet, the rest Neumann
    return FEMesh(nodes=base.nodes, elements=base.elements, dirichlet_nodes=dict(sorted(bc.items())))


@pytest.mark.parametrize("name,mesh,kind", [
    ("uniform 256^2, scalar kappa per sample", FEMesh.rectangle(256, 256), "sample"),
    ("odd sizes 333 x 207 (edge strips), non-zero Dirichlet data", FEMesh.rectangle(333, 207, bc_value=0.7), "sample"),
    ("skewed lattice (4 diagonals)", _skewed(272, 232), "sample"),
    ("partly Neumann boundary (assembled scalar-kappa operator, shared)", _partly_neumann(240, 224), "scalar"),
    ("partly Neumann boundary, one scalar PER SAMPLE (unfactored: per-sample matrices in the reference's operation order)",
     _partly_neumann(232, 216), "sample"),
    ("one per-element field shared by the batch", FEMesh.rectangle(288, 264), "field"),
    ("one field PER SAMPLE (per-sample matrices: fused passes on compact per-lane coefficients)", FEMesh.rectangle(272, 240, bc_value=0.3), "fields"),
    ("one field per sample, skewed lattice (4 diagonals)", _skewed(240, 216, seed=2), "fields"),
])
def test_two_samples_per_lane_kernels_agree_with_the_one_sample_kernels_and_the_oracle(name, mesh, kind):
    """B = 128 puts the fp32 V-cycle of a batch-shared matrix on dia_strip2_kernel (packed fp32 arithmetic, 8 B per
    lane); mg={'strip2': 0} keeps the fp64-in-registers kernels.  Both are preconditioners of the same fp64 CG: the
    solutions agree far inside the tolerance, the iteration counts do not move, and the result matches the oracle."""
    B, n, m = 128, mesh.n_nodes, mesh.n_elements
    gen = torch.Generator().manual_seed(77)
    if kind == "sample":
        kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    elif kind == "scalar":
        kappa = torch.tensor(1.37, dtype=T64)
    elif kind == "fields":
        kappa = torch.exp(0.3 * torch.randn(B, m, generator=gen, dtype=T64))
    else:
        kappa = torch.exp(0.3 * torch.randn(m, generator=gen, dtype=T64))
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    new = _run(mesh, kappa, f)
    old = _run(mesh, kappa, f, mg=dict(strip2=0))
    mid = _run(mesh, kappa, f, mg=dict(fused=0))     # two samples per lane, but the four single-stage passes per level
    assert new[3].path == "lattice-mgpcg" and new[3].not_converged == 0 and old[3].not_converged == 0
    for other in (old, mid):
        assert abs(new[3].iterations - other[3].iterations) <= 1 and abs(new[3].adj_iterations - other[3].adj_iterations) <= 1
        for a, b in zip(new[:3], other[:3]):
            assert float((a - b).abs().max() / b.abs().max()) < 2e-11, name
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, B - 1):
        kb = float(kappa[b]) if kind == "sample" else (kappa[b].numpy() if kind == "fields" else kappa.numpy())
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, kb, f[b].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=1)
        assert rel_err(new[0][b].cpu().numpy(), uo) < RTOL_U, name
        assert rel_err(new[2][b].cpu().numpy(), df) < RTOL_GRAD, name
        if kind == "sample":
            assert abs(float(new[1][b]) - dk.sum()) < RTOL_GRAD * abs(dk.sum()), name
        if kind == "fields":
            assert rel_err(new[1][b].cpu().numpy(), dk) < RTOL_GRAD, name


# ---- per-sample kappa fields: operator and coefficient-storage forms ----------------------------------------------------
@pytest.mark.parametrize("mesh_fn", [lambda: FEMesh.rectangle(256, 240, bc_value=0.4), lambda: _skewed(224, 208, seed=5)])
def test_per_sample_field_forms_agree(mesh_fn):
    """One kappa field per sample (one matrix per sample and level).  Default: entries sum_e kappa_e fl(t_e / den_e) (one
    rounding per contribution away from the reference's fl(fl(kappa_e t_e) / den_e)) and a V-cycle that reads an fp32
    diagonal + fp16 off-diagonals with the row sums kept; operator='assembled' = the reference's exact operation order;
    mg={'h16': 0} = plain fp32 coefficient copies.  All three meet the oracle; the default is within cond * eps of the
    bit-identical form and the compact coefficients do not cost iterations."""
    mesh = mesh_fn()
    B, n, m = 64, mesh.n_nodes, mesh.n_elements
    gen = torch.Generator().manual_seed(13)
    kappa = torch.exp(0.3 * torch.randn(B, m, generator=gen, dtype=T64))
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    dflt = _run(mesh, kappa, f)
    exact = _run(mesh, kappa, f, operator="assembled")
    plain = _run(mesh, kappa, f, mg=dict(h16=0))
    for r in (dflt, exact, plain):
        assert r[3].path == "lattice-mgpcg" and r[3].not_converged == 0
    assert abs(dflt[3].iterations - plain[3].iterations) <= 1 and abs(dflt[3].adj_iterations - plain[3].adj_iterations) <= 1
    worst = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(dflt[:3], exact[:3]))
    print(f"per-sample fields, default vs reference-order operator: {worst:.2e}")
    assert worst < 2e-11
    assert max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(dflt[:3], plain[:3])) < 2e-11
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, B - 1):
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, kappa[b].numpy(), f[b].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=2)
        for r in (dflt, exact):
            assert rel_err(r[0][b].cpu().numpy(), uo) < RTOL_U
            assert rel_err(r[1][b].cpu().numpy(), dk) < RTOL_GRAD
            assert rel_err(r[2][b].cpu().numpy(), df) < RTOL_GRAD
        assert rel_err(exact[1][b].cpu().numpy(), dk) < RTOL_GRAD / 5       # bit-identical matrix: margin against the refined oracle


def test_element_major_kappa_fields_in_node_layout():
    """layout='node' with kappa (m, B): per-sample fields element-major like f and u -- no transposing pass for kappa or
    its gradient; same numbers as the API layout, bit for bit."""
    mesh = FEMesh.rectangle(208, 200)
    B, n, m = 64, mesh.n_nodes, mesh.n_elements
    gen = torch.Generator().manual_seed(3)
    kappa = torch.exp(0.3 * torch.randn(B, m, generator=gen, dtype=T64))
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    u_s, gk_s, gf_s, _ = _run(mesh, kappa, f)
    kap = kappa.t().contiguous().to(DEV).requires_grad_(True)
    f_nm = f.t().contiguous().to(DEV).requires_grad_(True)
    u = DifferentiableFESolver(mesh, kap, device=DEV)(f_nm, layout="node")
    (u ** 2).sum().backward()
    assert kap.grad.shape == (m, B)
    assert torch.equal(u.detach().t(), u_s) and torch.equal(kap.grad.t(), gk_s) and torch.equal(f_nm.grad.t(), gf_s)


# ---- general path: smoothed aggregation -------------------------------------------------------------------------------------
def test_smoothed_aggregation_halves_the_iterations_of_the_general_path():
    """method='ell' (any mesh): the smoothed-aggregation hierarchy (default) against the piecewise-constant one
    (amg={'smoothed': 0}) on a jittered, renumbered mesh with one kappa per sample and a per-element field per sample:
    same answers (both meet the oracle), about half the iterations."""
    mesh = _unstructured(96, 88, seed=1)
    B, n, m = 16, mesh.n_nodes, mesh.n_elements
    gen = torch.Generator().manual_seed(5)
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for kappa in (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64), torch.exp(0.3 * torch.randn(B, m, generator=gen, dtype=T64))):
        sa = _run(mesh, kappa, f)
        pc = _run(mesh, kappa, f, amg=dict(smoothed=0))
        assert sa[3].path == "ell-amgpcg" and sa[3].not_converged == 0 and pc[3].not_converged == 0
        print(f"general path, {n} nodes: iterations {pc[3].iterations}+{pc[3].adj_iterations} piecewise constant -> "
              f"{sa[3].iterations}+{sa[3].adj_iterations} smoothed")
        assert sa[3].iterations <= 0.7 * pc[3].iterations
        for a, b in zip(sa[:3], pc[:3]):
            assert float((a - b).abs().max() / b.abs().max()) < 2e-11
        kb = float(kappa[0]) if kappa.dim() == 1 else kappa[0].numpy()
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, kb, f[0].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=1)
        assert rel_err(sa[0][0].cpu().numpy(), uo) < RTOL_U and rel_err(sa[2][0].cpu().numpy(), df) < RTOL_GRAD


@pytest.mark.timeout(600)
def test_general_path_at_512_takes_at_most_45_iterations():
    """rectangle(512, 512) forced through the general ELL path (method='ell'), 64 samples."""
    mesh = FEMesh.rectangle(512, 512)
    B = 64
    gen = torch.Generator().manual_seed(2024)
    kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)).to(DEV)
    f = torch.ones(B, mesh.n_nodes, dtype=T64, device=DEV)
    solver = DifferentiableFESolver(mesh, kappa, device=DEV, method="ell")
    with torch.no_grad():
        u = solver(f)
    info = solver.last_info
    print(f"512^2 through the general path: {info.iterations} iterations, max relres {info.max_relres:.1e}")
    assert info.path == "ell-amgpcg" and info.not_converged == 0 and info.iterations <= 45
    lat = DifferentiableFESolver(mesh, kappa, device=DEV)
    with torch.no_grad():
        u2 = lat(f)
    assert float((u - u2).abs().max() / u2.abs().max()) < 1e-10


# ---- fp16 coefficient copies: magnitudes across the batch, contrast inside a sample ---------------------------
def test_per_sample_fields_of_very_different_magnitudes_share_a_batch():
    """kappa_b(x) = c_b exp(0.3 randn) with c_b spread over 1e-6 .. 1e6 across the batch.  The fp16 couplings of the
    V-cycle are stored relative to a PER-SAMPLE power of two (the sample's largest free-row diagonal; the identity rows
    of Dirichlet nodes, 1.0 whatever kappa is, do not count), so every sample keeps 11 bits: compact coefficients stay in
    use and the iteration counts are those of an all-O(1) batch.  With homogeneous Dirichlet data u scales by 1 / c_b,
    dL/df by 1 / c_b^2 and dL/dkappa_e by 1 / c_b^3 EXACTLY (L = sum u^2), so every sample is checked against the c = 1
    run; three samples against the oracle as well."""
    mesh = FEMesh.rectangle(256, 240)
    B, n, m = 128, mesh.n_nodes, mesh.n_elements
    gen = torch.Generator().manual_seed(21)
    base = torch.exp(0.3 * torch.randn(B, m, generator=gen, dtype=T64))
    mag = 10.0 ** (12.0 * torch.rand(B, generator=gen, dtype=T64) - 6.0)
    mag[0], mag[-1] = 1e-6, 1e6
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    ref = _run(mesh, base, f)
    new = _run(mesh, base * mag[:, None], f)
    assert new[3].coeff_storage == "fp16-rowsum" and ref[3].coeff_storage == "fp16-rowsum"
    assert new[3].not_converged == 0 and bool(torch.isfinite(new[0]).all()) and bool(torch.isfinite(new[1]).all())
    assert new[3].iterations <= ref[3].iterations + 1 and new[3].adj_iterations <= ref[3].adj_iterations + 1
    mg_ = mag.to(DEV)[:, None]
    for a_, b_, p_ in ((new[0], ref[0], 1), (new[1], ref[1], 3), (new[2], ref[2], 2)):
        worst = float(((a_ * mg_ ** p_ - b_).abs().amax(dim=1) / b_.abs().amax(dim=1)).max())
        assert worst < RTOL_U, (p_, worst)
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, B // 2, B - 1):
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, (base[b] * mag[b]).numpy(),
                                            f[b].numpy(), lambda u_: 2 * u_, sparse=True, refine=2)
        assert rel_err(new[0][b].cpu().numpy(), uo) < RTOL_U
        assert rel_err(new[1][b].cpu().numpy(), dk) < RTOL_GRAD
        assert rel_err(new[2][b].cpu().numpy(), df) < RTOL_GRAD


def test_contrast_inside_a_sample_beyond_fp16_falls_back_to_fp32_copies():
    """A conductivity jump of 3e-7 inside every sample (left half O(1), right half 3e-7, log-normal noise on top): couplings
    of the weak half would land in fp16 subnormals / flush to 0 relative to the sample's scale, and with the row-sum rule an
    interior row of that half would get a vanishing diagonal (1 / 0 in the smoother).  The packing kernel reports it and the
    solve runs on plain fp32 copies instead: converged, finite, oracle-accurate."""
    mesh = FEMesh.rectangle(256, 224)
    B, n, m = 128, mesh.n_nodes, mesh.n_elements
    gen = torch.Generator().manual_seed(22)
    cx = mesh.nodes[mesh.elements].mean(dim=1)[:, 0]                 # element centroids
    jump = torch.where(cx < 0.5, torch.ones(m, dtype=T64), torch.full((m,), 3e-7, dtype=T64))
    kappa = jump[None, :] * torch.exp(0.3 * torch.randn(B, m, generator=gen, dtype=T64))
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    u, gk, gf, info = _run(mesh, kappa, f)
    assert info.coeff_storage == "fp32"
    assert info.not_converged == 0 and all(bool(torch.isfinite(t).all()) for t in (u, gk, gf))
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, B - 1):
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, kappa[b].numpy(), f[b].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=2)
        eu, ek, ef = rel_err(u[b].cpu().numpy(), uo), rel_err(gk[b].cpu().numpy(), dk), rel_err(gf[b].cpu().numpy(), df)
        print(f"jump 3e-7, sample {b}: u {eu:.1e} dkappa {ek:.1e} df {ef:.1e}, iterations {info.iterations}+{info.adj_iterations}")
        assert eu < 1e-9 and ek < 1e-9 and ef < 1e-9     # cond ~ 1e7 x the uniform mesh's: cond * eps, both codes


# ---- no load precedes any coefficient array (ABI v7; the fault of gpurun_out/r3i) -----------------------------------------
@pytest.mark.parametrize("B", [64, 128])
def test_no_kernel_reads_in_front_of_the_coefficient_arrays(B):
    """Every coefficient array handed to diffhe_lattice_pcg_solve -- fp64 values, fp32 diagonal, fp16 couplings, and for a
    shared matrix the fp32 copy, reciprocal diagonal and mask -- is placed directly behind a NaN-filled region of one
    allocation.  ABI v6 kernels read one element in front of a diagonal (times a zero window value: 0 x NaN = NaN, or a
    fault when the array starts an allocation); v7 kernels must not: results bitwise equal to the plain run.
    B = 64: one-sample strip kernels on per-lane fp16 couplings (the faulting instantiation); 128: fused passes."""
    from diffhe.solver import _Engine, K_SAMPLE_ELEM, K_SAMPLE
    from diffhe.plan import padded_batch
    mesh = FEMesh.rectangle(224, 200)
    plan = get_plan(mesh, torch.device(DEV))
    n, m = mesh.n_nodes, mesh.n_elements
    gen = torch.Generator().manual_seed(5)
    mg = dict(nu=2, n_coarse=8, omega=0.8, omegas=None, fp32=1, fmg=1, floor=1, tol_energy=1e-11)

    def behind_nan(t):
        pad = 4096
        raw = torch.empty(pad + t.numel() * t.element_size(), dtype=torch.uint8, device=DEV)
        raw[:pad] = 0xFF                                  # all-ones bytes: NaN in fp16, fp32 and fp64
        out = raw[pad:].view(t.dtype).view(t.shape)
        out.copy_(t)
        return out

    for mode, kappa in ((K_SAMPLE_ELEM, torch.exp(0.3 * torch.randn(B, m, generator=gen, dtype=T64))),
                        (K_SAMPLE, 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64))):
        eng = _Engine(plan, 1e-13, 200, 25, "gather")
        Bp = padded_batch(B)
        vals, Bv, scale, lift, _ = eng.lattice_assemble(kappa.to(DEV), mode, B, Bp, factor=True)
        rhs = torch.randn(n, Bp, generator=torch.Generator(device=DEV).manual_seed(1), dtype=T64, device=DEV)
        rhs[plan.bc_index()] = 0.0
        if Bv != 1:
            d32, o16, osc = eng.pack_cycle_coeffs(vals, Bv)
            assert d32 is not None
            plain = eng.lattice_pcg(vals, Bv, scale, rhs, Bp, mg, d32, None, off16=(o16, osc))
            moved = eng.lattice_pcg([behind_nan(v) for v in vals], Bv, scale, rhs, Bp, mg, [behind_nan(d) for d in d32], None,
                                    off16=([behind_nan(o) for o in o16], behind_nan(osc)))
        else:
            v32, rd32 = plan.shared_fp32(vals, cacheable=False)
            plain = eng.lattice_pcg(vals, Bv, scale, rhs, Bp, mg, v32, None, rdiag32=rd32)
            moved = eng.lattice_pcg([behind_nan(v) for v in vals], Bv, scale, rhs, Bp, mg, [behind_nan(v) for v in v32], None,
                                    rdiag32=[behind_nan(r) for r in rd32])
        assert plain[2] == 0 and moved[2] == 0 and bool(torch.isfinite(moved[0]).all())
        assert torch.equal(plain[0], moved[0]) and plain[1] == moved[1]


# ---- batches that are no multiple of 128: the fused passes with ONE sample per lane (BASELINE config 5: 64 per GPU) -----
@pytest.mark.parametrize("B", [64, 192])
@pytest.mark.parametrize("name,mesh_fn", [("uniform 256^2", lambda: FEMesh.rectangle(256, 256)),
                                          ("odd sizes, non-zero Dirichlet data", lambda: FEMesh.rectangle(333, 207, bc_value=0.7)),
                                          ("skewed lattice (4 diagonals)", lambda: _skewed(272, 232))])
def test_batches_of_64_and_192_take_the_fused_kernels_with_one_sample_per_lane(name, mesh_fn, B):
    """B = 64 (config 5's per-GPU shard) and 192 are whole waves but no multiple of 128: the V-cycle of a batch-shared matrix
    runs the same fused two-stage passes and the fp32 CG step with one sample per lane (round 3 sent them to the four
    single-stage fp64-in-register passes).  Same preconditioner up to fp32 rounding: iteration counts as with the old
    kernels (mg strip2 = 0), answers equal far inside the tolerance, and equal to the oracle."""
    mesh = mesh_fn()
    n = mesh.n_nodes
    gen = torch.Generator().manual_seed(78)
    kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    new = _run(mesh, kappa, f)
    old = _run(mesh, kappa, f, mg=dict(strip2=0))
    assert new[3].path == "lattice-mgpcg" and new[3].not_converged == 0 and old[3].not_converged == 0
    assert new[3].coeff_storage == "shared-fp32" and "one sample per lane" in new[3].precision
    assert abs(new[3].iterations - old[3].iterations) <= 1 and abs(new[3].adj_iterations - old[3].adj_iterations) <= 1
    for a, b in zip(new[:3], old[:3]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-11, name
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, B - 1):
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, float(kappa[b]), f[b].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=1)
        assert rel_err(new[0][b].cpu().numpy(), uo) < RTOL_U, name
        assert rel_err(new[2][b].cpu().numpy(), df) < RTOL_GRAD, name
        assert abs(float(new[1][b]) - dk.sum()) < RTOL_GRAD * abs(dk.sum()), name


@pytest.mark.parametrize("name,mesh_fn", [("uniform 256 x 224, non-zero Dirichlet data", lambda: FEMesh.rectangle(256, 224, bc_value=0.3)),
                                          ("odd sizes 301 x 263 (edge strips in both directions)", lambda: FEMesh.rectangle(301, 263))])
def test_four_samples_per_lane_pre_pass_agrees_with_the_two_sample_form(name, mesh_fn):
    """Batches of whole 256-sample waves run the fused way-down pass (two sweeps from zero + residual + restriction) with
    FOUR samples per lane (16-byte accesses: half the vector-memory instructions per byte); mg={'pre4': 0} keeps two.
    Same arithmetic per sample: identical iteration counts, answers equal to fp32-preconditioner noise, and the oracle."""
    mesh = mesh_fn()
    B, n = 256, mesh.n_nodes
    gen = torch.Generator().manual_seed(79)
    kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    new = _run(mesh, kappa, f)
    two = _run(mesh, kappa, f, mg=dict(pre4=0))
    assert new[3].path == "lattice-mgpcg" and new[3].not_converged == 0 and two[3].not_converged == 0
    assert "four in the way-down pass" in new[3].precision and "four in the way-down pass" not in two[3].precision
    assert new[3].iterations == two[3].iterations and new[3].adj_iterations == two[3].adj_iterations
    for a, b in zip(new[:3], two[:3]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-11, name
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, 100, B - 1):
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, float(kappa[b]), f[b].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=1)
        assert rel_err(new[0][b].cpu().numpy(), uo) < RTOL_U, name
        assert rel_err(new[2][b].cpu().numpy(), df) < RTOL_GRAD, name
        assert abs(float(new[1][b]) - dk.sum()) < RTOL_GRAD * abs(dk.sum()), name


def test_hessian_vector_products_on_a_mesh_with_a_degenerate_triangle():
    """A zero-area triangle is skipped silently by the reference (solver.py:120-121) and by the first-order kernels; the
    second-order path divided by its area: Hessian-vector products must be finite and equal to those of the
    same mesh WITHOUT that element (its kappa has no effect: zero gradient, zero Hessian row)."""
    base = _unstructured(10, 9, seed=12)
    el = base.elements
    extra = torch.tensor([[int(el[0, 0]), int(el[0, 1]), int(el[0, 0])]], dtype=el.dtype)       # repeated node: area 0
    mesh = FEMesh(nodes=base.nodes, elements=torch.cat([el, extra]), dirichlet_nodes=base.dirichlet_nodes)
    g = torch.Generator().manual_seed(6)
    m, n = base.n_elements, base.n_nodes
    k0 = torch.rand(m + 1, generator=g, dtype=T64) + 0.5
    f0 = torch.rand(n, generator=g, dtype=T64) + 0.5
    v, w = torch.randn(m + 1, generator=g, dtype=T64), torch.randn(n, generator=g, dtype=T64)

    def second(msh, kk, vv):
        kap, f = kk.clone().to(DEV).requires_grad_(True), f0.clone().to(DEV).requires_grad_(True)
        u = DifferentiableFESolver(msh, kap, device=DEV, tol=1e-14)(f)
        gk, gf = torch.autograd.grad((u ** 3).sum(), (kap, f), create_graph=True)
        hk, hf = torch.autograd.grad((gk * vv.to(DEV)).sum() + (gf * w.to(DEV)).sum(), (kap, f))
        return [t.detach().cpu() for t in (gk, gf, hk, hf)]

    with_deg = second(mesh, k0, v)
    without = second(base, k0[:m], v[:m])
    for tag, a, b in zip(("dL/dkappa", "dL/df", "H.kappa", "H.f"), with_deg, without):
        assert bool(torch.isfinite(a).all()), tag
        a_ = a[:m] if a.shape[0] == m + 1 else a
        assert rel_err(a_.numpy(), b.numpy()) < 1e-9, (tag, rel_err(a_.numpy(), b.numpy()))
    assert float(with_deg[0][m]) == 0.0 and float(with_deg[2][m]) == 0.0


# ---- general meshes: one scalar kappa per sample stays factored where the boundary is closed (round 4) --------------------
def test_general_path_keeps_a_per_sample_scalar_factored_on_closed_meshes():
    """Unstructured mesh, every boundary node Dirichlet, kappa (B,): K_b = kappa_b K_1 -- one unit matrix for the batch
    (the ELL kernels read it as broadcasts instead of one matrix per sample), the plan-constant aggregation hierarchy built
    once, K_1 x = F_b / kappa_b solved.  Same answers as operator='assembled' (one matrix per sample in the reference's
    operation order) far inside the tolerance, same iteration counts, the oracle met; with a Neumann part the mesh is NOT
    factored (cond * eps, as on lattices)."""
    mesh = _unstructured(48, 44, seed=21)
    B, n = 64, mesh.n_nodes
    gen = torch.Generator().manual_seed(31)
    kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    fac = _run(mesh, kappa, f)
    per = _run(mesh, kappa, f, operator="assembled")
    assert fac[3].path == "ell-amgpcg" and fac[3].factored and not per[3].factored
    assert fac[3].not_converged == 0 and per[3].not_converged == 0
    assert abs(fac[3].iterations - per[3].iterations) <= 1 and abs(fac[3].adj_iterations - per[3].adj_iterations) <= 1
    for a, b in zip(fac[:3], per[:3]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-11
    again = _run(mesh, kappa, f)                       # the cached hierarchy: bitwise the same answers
    assert all(torch.equal(a, b) for a, b in zip(fac[:3], again[:3]))
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    for b in (0, B - 1):
        uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, float(kappa[b]), f[b].numpy(),
                                            lambda u_: 2 * u_, sparse=True, refine=1)
        assert rel_err(fac[0][b].cpu().numpy(), uo) < RTOL_U
        assert rel_err(fac[2][b].cpu().numpy(), df) < RTOL_GRAD
        assert abs(float(fac[1][b]) - dk.sum()) < RTOL_GRAD * abs(dk.sum())
    # open boundary: Dirichlet data on part of it only
    keep = {k: v for i, (k, v) in enumerate(mesh.dirichlet_nodes.items()) if i % 3}
    open_mesh = FEMesh(nodes=mesh.nodes, elements=mesh.elements, dirichlet_nodes=keep)
    assert not get_plan(open_mesh, torch.device(DEV)).closed_boundary_general()
    assert get_plan(mesh, torch.device(DEV)).closed_boundary_general()
    assert not _run(open_mesh, kappa, f)[3].factored


def test_compact_element_matrices_of_congruent_lattices_give_bitwise_the_same_operator_and_gradient():
    """FEMesh.rectangle(64, 32) on the unit square (spacings 2^-6, 2^-5: exact): all triangles of one orientation have
    bitwise the same unit stiffness matrix, and the per-sample assembly / the per-element gradient read a (9, 2) table
    instead of the (9, m) arrays (plan.LatticeLevel.compact).  Same numbers, same order: bitwise equal matrices, lift and
    gradients; a lattice whose spacing is not exactly representable has no compact form."""
    from diffhe import _hip
    from diffhe.plan import _stream
    mesh = FEMesh.rectangle(64, 32, bc_value=0.4)
    plan = get_plan(mesh, torch.device(DEV))
    lev = plan.levels[0]
    assert lev.compact("k0") is not None and lev.compact("k0ref") is not None
    assert get_plan(FEMesh.rectangle(60, 33), torch.device(DEV)).levels[0].compact("k0ref") is None
    L, st = _hip.lib(), _stream(plan.device)
    Bv = 64
    kap = torch.exp(0.3 * torch.randn(lev.m, Bv, dtype=T64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4)))
    out = []
    for which in ("k0", "k0ref"):
        full = lev.k0 if which == "k0" else lev.k0ref()
        pair = []
        for tab, flag in ((full, 0), (lev.compact(which), 1)):
            v = torch.full((lev.nd, lev.n, Bv), float("nan"), dtype=T64, device=DEV)
            lf = torch.full((lev.n, Bv), float("nan"), dtype=T64, device=DEV)
            _hip.check(L.diffhe_lattice_assemble_rows(_hip.ptr(tab), flag, _hip.ptr(kap), Bv, 1, _hip.ptr(lev.is_bc),
                                                      _hip.ptr(plan.g), _hip.ptr(v), _hip.ptr(lf), lev.nx, lev.ny, lev.nd, Bv,
                                                      st), "assemble")
            pair.append((v, lf))
        torch.cuda.synchronize()
        assert torch.equal(pair[0][0], pair[1][0]) and torch.equal(pair[0][1], pair[1][1]), which
    lam = torch.randn(lev.n, Bv, dtype=T64, device=DEV)
    u = torch.randn(lev.n, Bv, dtype=T64, device=DEV)
    for tab, flag in ((lev.k0, 0), (lev.compact("k0"), 1)):
        dk = torch.full((lev.m, Bv), float("nan"), dtype=T64, device=DEV)
        _hip.check(L.diffhe_lattice_grad_kappa(lev.nx, lev.ny, _hip.ptr(tab), flag, _hip.ptr(lam), _hip.ptr(u), _hip.ptr(plan.g),
                                               _hip.ptr(dk), Bv, st), "grad")
        out.append(dk)
    torch.cuda.synchronize()
    assert torch.equal(out[0], out[1]) and bool(torch.isfinite(out[0]).all())


@pytest.mark.parametrize("nx,ny,Bv,bc", [(128, 64, 64, 0.4), (203, 77, 128, 0.25), (144, 81, 64, None)])
def test_strip_form_of_the_per_sample_assembly_is_bitwise_the_node_per_wave_kernel(nx, ny, Bv, bc, monkeypatch):
    """Per-sample kappa fields on a big level are assembled by a strip pass (a wave marches down RW node columns with the
    kappa of two quad rows in registers: every kappa_e loaded once per wave and quad row instead of once per incident
    node).  Same per-node arithmetic: bitwise the matrix and the lift of the node-per-wave kernel, for ragged widths and
    heights (partial last strip, partial last row chunk), with the (9, m) and the compact tables, with and without
    Dirichlet nodes."""
    from diffhe import _hip
    from diffhe.plan import _stream
    mesh = FEMesh.rectangle(nx, ny, bc_value=bc) if bc is not None else _partly_neumann(nx, ny)
    plan = get_plan(mesh, torch.device(DEV))
    lev = plan.levels[0]
    L, st = _hip.lib(), _stream(plan.device)
    kap = torch.exp(0.5 * torch.randn(lev.m, Bv, dtype=T64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(9)))
    tabs = [(lev.k0ref(), 0)] + ([(lev.compact("k0ref"), 1)] if lev.compact("k0ref") is not None else [])
    for tab, flag in tabs:
        res = []
        for strip in ("0", "1"):
            monkeypatch.setenv("DIFFHE_ASM_STRIP", strip)
            v = torch.full((lev.nd, lev.n, Bv), float("nan"), dtype=T64, device=DEV)
            lf = torch.full((lev.n, Bv), float("nan"), dtype=T64, device=DEV)
            _hip.check(L.diffhe_lattice_assemble_rows(_hip.ptr(tab), flag, _hip.ptr(kap), Bv, 1, _hip.ptr(lev.is_bc),
                                                      _hip.ptr(plan.g), _hip.ptr(v), _hip.ptr(lf), lev.nx, lev.ny, lev.nd, Bv,
                                                      st), "assemble")
            torch.cuda.synchronize()
            res.append((v, lf))
        assert bool(torch.isfinite(res[0][0]).all()) and bool(torch.isfinite(res[0][1]).all())
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (nx, ny, flag)


@pytest.mark.parametrize("case", ["scalar-per-sample (shared unit matrix, fp32 cycle)", "field-per-sample (per-sample matrices)",
                                  "field-per-sample, fp32 cycle", "P2 elements (rows of up to 19 entries)",
                                  "scalar-per-sample, batch of 128",
                                  "scalar-per-sample, batch of 128, fp64 cycle"])
def test_pipelined_wave_per_node_kernels_give_bitwise_the_plain_kernels_answers(case, monkeypatch):
    """The general path's sweep / residual / CG-product kernels for batches of >= 64 (`ellw_kernel`: a wave per node,
    column indices by one vector load + readlane, the next node's row data requested behind this node's gathers) take
    the same entries in the same order with the same operations as the plain kernels, and give each block the same
    nodes to sum: u, dL/dkappa and dL/df are BITWISE those of DIFFHE_ELL_PIPE=0, for batch-shared and per-sample
    matrices, fp64- and fp32-stored cycles, 7-entry rows and the wider rows of P2 elements and coarse levels, batches of one
    and two waves of samples."""
    gen = torch.Generator().manual_seed(77)
    B = 128 if "128" in case else 64
    kw = {"amg": {"fp32": 0}} if "fp64 cycle" in case else {}
    if case.startswith("P2"):
        mesh = FEMesh.rectangle_p2(14, 12, bc_value=0.2)
        kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    else:
        mesh = _unstructured(44, 40, seed=5)
        if case.startswith("scalar"):
            kappa = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
        else:
            kappa = torch.exp(0.4 * torch.randn(B, mesh.n_elements, generator=gen, dtype=T64))
            if "fp32" in case:
                kw["amg"] = {"fp32": 1}
    f = 1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    res = []
    for pipe in ("0", "1"):
        monkeypatch.setenv("DIFFHE_ELL_PIPE", pipe)
        res.append(_run(mesh, kappa, f, **kw))
    assert res[0][3].path.startswith("ell-") and res[0][3].not_converged == 0
    assert res[0][3].iterations == res[1][3].iterations and res[0][3].adj_iterations == res[1][3].adj_iterations
    for a, b in zip(res[0][:3], res[1][:3]):
        assert torch.equal(a, b), case


def test_general_path_keeps_one_scalar_kappa_factored_too():
    """kappa = one scalar for the whole batch (the reference's default call) on a closed unstructured mesh: the same factored
    form as one scalar per sample -- the unit matrix, its plan-cached hierarchy, K_1 x = F / kappa -- instead of a matrix
    that carries kappa and a Galerkin product per solve.  Same answers as operator='assembled', the oracle met, the
    gradient a scalar."""
    mesh = _unstructured(40, 36, seed=11)
    B, n = 64, mesh.n_nodes
    gen = torch.Generator().manual_seed(5)
    kappa = torch.tensor(1.7, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)
    fac = _run(mesh, kappa, f)
    per = _run(mesh, kappa, f, operator="assembled")
    assert fac[3].path == "ell-amgpcg" and fac[3].factored and not per[3].factored
    assert fac[1].shape == kappa.shape and fac[3].not_converged == 0
    for a, b in zip(fac[:3], per[:3]):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-11
    bn, bv = np.array(list(mesh.dirichlet_nodes.keys())), np.array(list(mesh.dirichlet_nodes.values()))
    uo, dk, df = orc.solve_with_adjoint(mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv, 1.7, f[3].numpy(),
                                        lambda u_: 2 * u_, sparse=True, refine=1)
    assert rel_err(fac[0][3].cpu().numpy(), uo) < RTOL_U
    assert rel_err(fac[2][3].cpu().numpy(), df) < RTOL_GRAD

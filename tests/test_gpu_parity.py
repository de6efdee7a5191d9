"""GPU parity tests: the HIP path (through the Python boundary and the C ABI) against
the golden vectors produced by the reference and against the CPU oracle on the same
seeded inputs.  Stated tolerance: 1e-10 relative (max-norm) on nodal u, dL/dkappa and
dL/df (BASELINE.json north_star)."""
import math

import numpy as np
import pytest
import torch

from diffhe import FEMesh, DifferentiableFESolver, PhysicsLoss
from diffhe import _hip
from diffhe.plan import get_plan
from oracle import p1_oracle as orc
from _util import golden, golden_names, rel_err, loss_grad, RTOL_U, RTOL_GRAD

pytestmark = pytest.mark.gpu
T64 = torch.float64


def mesh_from(g):
    bc = {int(k): float(v) for k, v in zip(g["bc_nodes"], g["bc_vals"])}
    return FEMesh(nodes=torch.from_numpy(g["nodes"]), elements=torch.from_numpy(g["elements"]), dirichlet_nodes=bc)


def arrays(mesh):
    bn = np.array(list(mesh.dirichlet_nodes.keys()), dtype=np.int64)
    bv = np.array(list(mesh.dirichlet_nodes.values()), dtype=np.float64)
    return mesh.nodes.numpy(), mesh.elements.numpy(), bn, bv


def torch_loss(kind, u, data=None):
    if kind == "sum":
        return u.sum()
    if kind == "sumsq":
        return (u ** 2).sum()
    return ((u - torch.from_numpy(data)) ** 2).mean()


def test_extension_is_loaded():
    assert _hip.lib().diffhe_abi_version() == _hip.ABI_VERSION
    assert torch.cuda.is_available()


@pytest.mark.parametrize("name", ["g1_config1"] + golden_names("g2_1d_fwd_") + ["g4_2d_fwd_32"])
def test_forward_golden(name):
    g = golden(name)
    solver = DifferentiableFESolver(mesh_from(g), float(g["kappa"]))
    u = solver(torch.from_numpy(g["f"]))
    assert u.dtype == T64 and u.shape == (len(g["f"]),) and u.device.type == "cpu"
    assert rel_err(u.numpy(), g["u"]) < RTOL_U, solver.last_info
    for node, val in zip(g["bc_nodes"], g["bc_vals"]):       # Dirichlet values exact
        assert abs(float(u[node]) - val) < 1e-12
    if name == "g1_config1":
        assert np.max(np.abs(u.numpy() - g["exact"])) < 1e-12


@pytest.mark.parametrize("assembly,method", [("gather", "auto"), ("gather", "ell"), ("atomic", "ell")])
@pytest.mark.parametrize("name", golden_names("g3_1d_grad_") + golden_names("g4_2d_0"))
def test_gradients_golden(name, assembly, method):
    g = golden(name)
    kind = str(g["loss_kind"])
    kappa = torch.tensor(float(g["kappa"]), dtype=T64, requires_grad=True)
    f = torch.from_numpy(g["f"]).clone().requires_grad_(True)
    solver = DifferentiableFESolver(mesh_from(g), kappa, assembly=assembly, method=method)
    u = solver(f)
    if name.startswith("g4"):
        # scalar kappa on a closed lattice this small: the direct dense path (no iteration at all)
        assert solver.last_info.path == ("lattice-direct" if method == "auto" else "ell-amgpcg") or \
            (method == "ell" and solver.last_info.path == "ell-pcg")      # tiny meshes: no coarse level
    loss = torch_loss(kind, u, g.get("data"))
    loss.backward()
    assert rel_err(u.detach().numpy(), g["u"]) < RTOL_U
    assert abs(float(loss) - float(g["loss"])) <= 1e-10 * abs(float(g["loss"]))
    assert abs(float(kappa.grad) - float(g["dkappa"])) <= RTOL_GRAD * abs(float(g["dkappa"])), solver.last_info
    assert rel_err(f.grad.numpy(), g["df"]) < RTOL_GRAD


@pytest.mark.parametrize("name", golden_names("g5_asm_"))
def test_assembled_system_golden(name):
    """K and F before BCs (reference solver.py:82-96 / :112-145), straight from the C ABI."""
    g = golden(name)
    mesh = mesh_from(g)
    if mesh.dim == 1:   # force the general path: a chain mesh never materialises K
        perm = np.arange(mesh.n_nodes)[::-1].copy()
    else:
        perm = np.arange(mesh.n_nodes)
    inv = np.argsort(perm)
    mesh2 = FEMesh(nodes=torch.from_numpy(g["nodes"][inv]), elements=torch.from_numpy(perm[g["elements"]]),
                   dirichlet_nodes={})
    dev = torch.device("cuda")
    plan = get_plan(mesh2, dev)
    plan.ensure_ell()            # lattice meshes build the general ELL pattern lazily
    assert not plan.is_chain
    L = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    n, m, W = plan.n, plan.m, plan.W
    kap = torch.full((1,), float(g["kappa"]), dtype=T64, device=dev)
    for mode in ("gather", "atomic"):
        Bv = 1 if mode == "gather" else 2
        vals = torch.zeros((W, n, Bv), dtype=T64, device=dev)
        if mode == "gather":
            _hip.check(L.diffhe_ell_assemble_rows(_hip.ptr(plan.k0), _hip.ptr(kap), 0, 0, _hip.ptr(plan.ent_ptr),
                                                  _hip.ptr(plan.contrib), _hip.ptr(plan.cols), None, None, None,
                                                  _hip.ptr(vals), None, n, m, W, Bv, st), "rows")
        else:
            _hip.check(L.diffhe_ell_assemble_atomic(_hip.ptr(plan.coords), _hip.ptr(plan.elems), plan.dim,
                                                    _hip.ptr(kap), 0, 0, _hip.ptr(plan.slot_of), _hip.ptr(vals), n,
                                                    m, W, Bv, st), "atomic")
        K = np.zeros((n, n))
        cols = plan.cols.cpu().numpy()
        v = vals.cpu().numpy()[:, :, 0]
        np.add.at(K, (np.tile(np.arange(n), W), cols.ravel()), v.ravel())
        K = K[np.ix_(perm, perm)]
        assert np.max(np.abs(K - g["K"])) < 1e-13 * np.max(np.abs(g["K"])), mode
    f_nm = torch.from_numpy(g["f"][inv]).to(dev).reshape(n, 1).contiguous()
    F = torch.empty((n, 1), dtype=T64, device=dev)
    _hip.check(L.diffhe_ell_spmv_shared(_hip.ptr(plan.Mvals), _hip.ptr(plan.cols), _hip.ptr(f_nm), None, 1, None,
                                        None, _hip.ptr(F), n, W, 1, st), "spmv")
    assert np.max(np.abs(F.cpu().numpy()[perm, 0] - g["F"])) < 1e-14


@pytest.mark.parametrize("method", ["auto", "ell"])
@pytest.mark.parametrize("name", golden_names("g9_batch_"))
def test_batched_golden(name, method):
    """Batch = loop of reference solves (G9): per-sample kappa (B,), f (B,n)."""
    g = golden(name)
    kappa = torch.from_numpy(g["kappa"]).clone().requires_grad_(True)
    f = torch.from_numpy(g["f"]).clone().requires_grad_(True)
    solver = DifferentiableFESolver(mesh_from(g), kappa, method=method)
    u = solver(f)
    assert u.shape == g["u"].shape
    (u ** 2).sum().backward()
    assert rel_err(u.detach().numpy(), g["u"]) < RTOL_U
    assert rel_err(kappa.grad.numpy(), g["dkappa"]) < RTOL_GRAD
    assert rel_err(f.grad.numpy(), g["df"]) < RTOL_GRAD
    # shared scalar kappa over a batch: gradient = sum over samples (all-reduce semantics)
    k0 = torch.tensor(float(g["kappa"][0]), dtype=T64, requires_grad=True)
    u0 = DifferentiableFESolver(mesh_from(g), k0, method=method)(torch.from_numpy(g["f"]))
    (u0 ** 2).sum().backward()
    ref = sum(orc.solve_with_adjoint(g["nodes"], g["elements"], g["bc_nodes"], g["bc_vals"], float(g["kappa"][0]),
                                     g["f"][b], lambda u: 2 * u)[1].sum() for b in range(len(g["f"])))
    assert abs(float(k0.grad) - ref) <= RTOL_GRAD * abs(ref)


def test_reference_known_answers():
    """reference tests/test_fem.py:85-179 run through the HIP path."""
    for N, atol in ((10, 1e-10), (100, 1e-9)):
        mesh = FEMesh.line(n_elements=N)
        x = mesh.nodes.squeeze(1)
        u = DifferentiableFESolver(mesh)(torch.ones_like(x))
        assert torch.allclose(u, x * (1 - x) / 2, atol=atol)
        assert abs(float(u[0])) < 1e-12 and abs(float(u[-1])) < 1e-12
    errs = []
    for N in (10, 20, 40, 80):
        mesh = FEMesh.line(n_elements=N)
        x = mesh.nodes.squeeze(1)
        u = DifferentiableFESolver(mesh)((math.pi ** 2) * torch.sin(math.pi * x))
        errs.append(float((u - torch.sin(math.pi * x)).abs().max()))
    for a, b in zip(errs, errs[1:]):
        assert a / (b + 1e-15) > 3.0
    mesh = FEMesh.line(n_elements=10, bc_left=1.0, bc_right=2.0)
    x = mesh.nodes.squeeze(1)
    assert torch.allclose(DifferentiableFESolver(mesh)(torch.zeros_like(x)), 1.0 + x, atol=1e-10)
    kappa = torch.tensor(1.0, dtype=T64, requires_grad=True)
    mesh = FEMesh.line(n_elements=5)
    DifferentiableFESolver(mesh, kappa=kappa)(torch.ones(6, dtype=T64)).sum().backward()
    assert kappa.grad is not None and kappa.grad.abs() > 1e-10
    mesh = FEMesh.rectangle(nx=4, ny=4)
    assert DifferentiableFESolver(mesh)(torch.zeros(25, dtype=T64)).abs().max() < 1e-10
    mesh = FEMesh.rectangle(nx=8, ny=8)
    u = DifferentiableFESolver(mesh)(torch.ones(81, dtype=T64))
    assert float(u[mesh.free_nodes()].min()) > 0.0
    # f given as float32 / (n,1) works and returns float64
    u32 = DifferentiableFESolver(mesh)(torch.ones(81, 1, dtype=torch.float32))
    assert u32.dtype == T64 and torch.allclose(u32, u)


@pytest.mark.parametrize("mk", [lambda: FEMesh.line(37, -1.0, 2.0, 0.3, -0.2), lambda: FEMesh.rectangle(7, 5, (0, 2), (0, 1), 0.1)])
def test_per_element_kappa_vs_oracle(mk):
    """Per-element kappa (documented by the reference, solver.py:28-29, but broken there)."""
    mesh = mk()
    nodes, el, bn, bv = arrays(mesh)
    rng = np.random.default_rng(11)
    B = 5
    kap = np.exp(0.3 * rng.standard_normal((B, mesh.n_elements)))
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    # (m,) shared by the batch, and (B,m)
    for kk, ff in ((kap[0], f), (kap, f), (kap[1], f[1])):
        kt = torch.from_numpy(kk).clone().requires_grad_(True)
        ft = torch.from_numpy(ff).clone().requires_grad_(True)
        u = DifferentiableFESolver(mesh, kt)(ft)
        (u ** 2).sum().backward()
        fb = ff if ff.ndim == 2 else ff[None]
        dk_ref = np.zeros_like(kk)
        for b in range(len(fb)):
            kb = kk[b] if kk.ndim == 2 else kk
            uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kb, fb[b], lambda u: 2 * u)
            ub = u[b] if u.dim() == 2 else u
            assert rel_err(ub.detach().numpy(), uo) < RTOL_U
            gf = ft.grad[b] if ft.grad.dim() == 2 else ft.grad
            assert rel_err(gf.numpy(), dfo) < RTOL_GRAD
            if kk.ndim == 2:
                dk_ref[b] = dko
            else:
                dk_ref += dko
        assert rel_err(kt.grad.numpy(), dk_ref) < RTOL_GRAD


def test_chain_with_interior_and_one_sided_dirichlet():
    rng = np.random.default_rng(5)
    for bc in ({0: 1.0}, {12: -0.5}, {3: 0.2, 9: 1.0}, {0: 0.0, 5: 0.3, 12: 1.0}):
        x = np.sort(rng.uniform(0, 2, 13))
        mesh = FEMesh(nodes=torch.from_numpy(x[:, None]), elements=FEMesh.line(12).elements, dirichlet_nodes=dict(bc))
        nodes, el, bn, bv = arrays(mesh)
        kap = rng.uniform(0.5, 2.0, 12)
        f = rng.standard_normal(13)
        kt = torch.from_numpy(kap).requires_grad_(True)
        ft = torch.from_numpy(f).requires_grad_(True)
        solver = DifferentiableFESolver(mesh, kt)
        u = solver(ft)
        assert solver.last_info.path == "chain1d-scan-ref"      # default: the reference's rounded system
        (u ** 2).sum().backward()
        uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kap, f, lambda u: 2 * u)
        assert rel_err(u.detach().numpy(), uo) < RTOL_U
        assert rel_err(kt.grad.numpy(), dko) < RTOL_GRAD
        assert rel_err(ft.grad.numpy(), dfo) < RTOL_GRAD


def test_unordered_1d_mesh_takes_general_path():
    g = golden("g2_1d_fwd_010")
    n = len(g["f"])
    perm = np.random.default_rng(2).permutation(n)
    inv = np.argsort(perm)
    bc = {int(perm[k]): float(v) for k, v in zip(g["bc_nodes"], g["bc_vals"])}
    mesh = FEMesh(nodes=torch.from_numpy(g["nodes"][inv]), elements=torch.from_numpy(perm[g["elements"]]),
                  dirichlet_nodes=bc)
    solver = DifferentiableFESolver(mesh, float(g["kappa"]))
    u = solver(torch.from_numpy(g["f"][inv]))
    assert solver.last_info.path in ("ell-pcg", "ell-amgpcg")
    assert rel_err(u.numpy()[perm], g["u"]) < RTOL_U


def test_kappa_recovery_trajectory():
    """examples/poisson_1d_demo.py:88-112 on the HIP path reproduces the reference run (G7)."""
    g = golden("g7_kappa_recovery")
    mesh = mesh_from(g)
    f = torch.from_numpy(g["f"])
    u_data = torch.from_numpy(g["u_data"])
    k = torch.tensor(1.0, dtype=T64, requires_grad=True)
    opt = torch.optim.Adam([k], lr=0.1)
    for step in range(200):
        opt.zero_grad()
        u = DifferentiableFESolver(mesh, kappa=k.abs())(f)
        loss = ((u - u_data) ** 2).mean()
        loss.backward()
        grad = float(k.grad)
        opt.step()
        if step in (0, 1, 2, 99, 199):
            ref = g["traj"][step]
            assert abs(float(loss) - ref[0]) <= 1e-9 * abs(ref[0]) + 1e-20
            assert abs(grad - ref[1]) <= 1e-9 * abs(ref[1]) + 1e-16
            assert abs(float(k.detach()) - ref[2]) < 1e-9
    assert abs(float(k.detach()) - 2.0) < 1e-4


def test_physics_loss_fem_match_value_and_cache():
    g = golden("g8_physics_loss")
    mesh = mesh_from(g)
    loss = PhysicsLoss(mesh, lambda x: torch.ones_like(x), mode="fem_match")
    u_pred = torch.from_numpy(g["u_pred"]).requires_grad_(True)
    v = loss(u_pred)
    assert abs(float(v) - float(g["fem_match"])) < 1e-15
    v.backward()
    assert u_pred.grad is not None
    target = loss.fem_target()
    assert loss.fem_target() is target                          # cached: not re-solved


def test_physics_loss_rhs_ensemble_is_one_batched_solve():
    """SURVEY 8(f) rank 1 / fixture G12: six forcing functions -> ONE batched HIP solve; the per-member values equal
    six separate reference `PhysicsLoss(mesh, f_k, "fem_match")(u_pred)` calls (loss.py:78-83), the targets the
    reference's own solves."""
    g = golden("g12_physics_loss_ensemble")
    mesh = mesh_from(g)
    fns = {"one": lambda x: torch.ones_like(x), "lin": lambda x: 1.0 + 2.0 * x,
           "sin1": lambda x: (math.pi ** 2) * torch.sin(math.pi * x), "sin3": lambda x: torch.sin(3 * math.pi * x),
           "quad": lambda x: 4.0 * x * (1.0 - x), "exp": lambda x: torch.exp(-x)}
    loss = PhysicsLoss(mesh, forcing_fns=[fns[str(nm)] for nm in g["names"]])
    u_pred = torch.from_numpy(g["u_pred"]).requires_grad_(True)
    per = loss.member_losses(u_pred)
    assert per.shape == (6,)
    assert rel_err(per.detach().numpy(), g["fem_match"]) < 1e-12
    assert rel_err(loss.fem_target().numpy(), g["u_fem"]) < RTOL_U
    assert loss.solver.last_info.path.startswith("chain1d") and loss.fem_target().shape == (6, mesh.n_nodes)
    v = loss(u_pred)
    assert abs(float(v) - float(g["fem_match"].mean())) < 1e-13 * float(g["fem_match"].mean())
    v.backward()
    assert u_pred.grad is not None
    # 2D ensemble on the lattice path: batched target == loop of single targets
    m2 = FEMesh.rectangle(16, 16)
    f2 = [lambda p: torch.ones(len(p), dtype=T64), lambda p: p[:, 0] + 2 * p[:, 1], lambda p: torch.sin(3 * p[:, 0])]
    ens = PhysicsLoss(m2, forcing_fns=f2).fem_target()
    for i, fn in enumerate(f2):
        single = PhysicsLoss(m2, fn).fem_target()
        assert rel_err(ens[i].numpy(), single.numpy()) < 1e-12


def test_config2_shape_1d_10000():
    """BASELINE config 2 at its stated size -- 1D, 10 000 elements, 1024 right-hand sides, fwd + adjoint -- against
    the REFERENCE ITSELF: fixture G10 holds rows 0, 1, 1023 of the reference's own solve (dense assembly +
    torch.linalg.solve, solver.py:73-98,153-183) of exactly this batch.  Strict 1e-10, no relaxed clause.
    cond(K) ~ 4e7 here; what makes the difference is the rounded diagonal fl(k_{i-1} + k_i) of the matrix the
    reference assembles (4e-10 in u): the default chain mode solves THAT system (chain1d.hip)."""
    g = golden("g10_config2_1d_10000")
    mesh = FEMesh.line(int(g["n_elements"]))
    nodes, el, bn, bv = arrays(mesh)
    B = int(g["batch"])
    assert B == 1024
    gen = torch.Generator().manual_seed(int(g["seed"]))
    f = 1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    for i, row in enumerate(g["rows"]):                      # same inputs as the reference saw
        assert np.array_equal(f[int(row)].numpy(), g["f"][i])
    k = torch.ones(B, dtype=T64, requires_grad=True)         # kappa = 1, one gradient per right-hand side
    fc = f.cuda().requires_grad_(True)
    solver = DifferentiableFESolver(mesh, k)
    u = solver(fc)
    assert u.is_cuda and solver.last_info.path == "chain1d-scan-ref"
    loss = 0.5 * (u ** 2).sum() / B
    loss.backward()
    uh = u.detach().cpu().numpy()
    worst = 0.0
    for i, row in enumerate(g["rows"]):
        worst = max(worst, rel_err(uh[int(row)], g["u"][i]))
    print(f"config 2: max rel err vs the reference's torch.linalg.solve over rows {list(g['rows'])}: {worst:.2e}")
    assert worst < RTOL_U
    # gradients (the reference's backward is O(N^3): infeasible at this size) against the fp64 oracle, which
    # reproduces the reference's u on these rows to 1e-14 (tests/test_oracle_golden.py) -- strict as well
    dfh = fc.grad.cpu().numpy()
    for b in (0, 1, 511, 1023):
        uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, 1.0, f[b].numpy(), lambda u_: u_ / B, sparse=True)
        assert rel_err(uh[b], uo) < RTOL_U
        assert rel_err(dfh[b], dfo) < RTOL_GRAD
        assert abs(float(k.grad[b]) - dko.sum()) < RTOL_GRAD * abs(dko.sum())
    # analytic identity dL/dkappa = -<gbar,u>/kappa (Appendix B): exact for the UNROUNDED system only -- for the
    # matrix the reference assembles it holds to cond * eps (measured 3.6e-10 here, on the oracle alike)
    ref = -float((u.detach() ** 2).sum() / B)
    assert abs(float(k.grad.sum()) - ref) <= 2e-9 * abs(ref)
    # f == 1: u = x(1-x)/2 at the nodes.  The reference's own solve is 4e-10 from that identity at this size (its
    # rounded diagonal); the plain scan (chain="exact") solves the unrounded system and meets it to 1e-14.
    x = mesh.nodes.squeeze(1)
    u1 = DifferentiableFESolver(mesh, chain="exact")(torch.ones_like(x))
    assert float((u1 - x * (1 - x) / 2).abs().max()) < 1e-13 * 0.125


def test_config2_exact_scan_mode_against_extended_precision():
    """chain="exact": the plain scan solves the UNROUNDED weighted Laplacian; judged against the extended-precision
    restatement (oracle.chain_solve_longdouble).  It is 4e-10 away from the reference at this size -- measured
    here and required to be outside the tolerance, so that the two modes cannot be confused."""
    g = golden("g10_config2_1d_10000")
    mesh = FEMesh.line(10_000)
    nodes, el, bn, bv = arrays(mesh)
    f = torch.from_numpy(g["f"])
    fc = f.cuda().requires_grad_(True)
    k = torch.tensor(1.0, dtype=T64, requires_grad=True)
    solver = DifferentiableFESolver(mesh, k, chain="exact")
    u = solver(fc)
    assert solver.last_info.path == "chain1d-scan"
    (0.5 * (u ** 2).sum()).backward()
    for b in range(f.shape[0]):
        ux, dkx, dfx = orc.chain_solve_longdouble(nodes, bn, bv, 1.0, f[b].numpy(), lambda u_: u_)
        assert rel_err(u[b].detach().cpu().numpy(), ux) < 1e-13
        assert rel_err(fc.grad[b].cpu().numpy(), dfx) < 1e-12
        d = rel_err(u[b].detach().cpu().numpy(), g["u"][b])
        assert 1e-10 < d < 1e-9, d


def test_1d_gradients_at_2000_elements_vs_reference_autograd():
    """G13: u, dL/dkappa, dL/df from the reference's own autograd at 2000 elements (cond 1.6e6)."""
    g = golden("g13_1d_grad_2000")
    mesh = FEMesh.line(int(g["n_elements"]))
    k = torch.tensor(float(g["kappa"]), dtype=T64, requires_grad=True)
    f = torch.from_numpy(g["f"]).requires_grad_(True)
    u = DifferentiableFESolver(mesh, k)(f)
    L = (u ** 2).sum()
    L.backward()
    assert rel_err(u.detach().numpy(), g["u"]) < RTOL_U
    assert abs(float(L) - float(g["loss"])) < RTOL_U * abs(float(g["loss"]))
    assert abs(float(k.grad) - float(g["dkappa"])) < RTOL_GRAD * abs(float(g["dkappa"]))
    assert rel_err(f.grad.numpy(), g["df"]) < RTOL_GRAD


def test_lattice_graded_mesh_and_per_element_kappa_all_levels():
    """Lattice connectivity with NON-uniform node positions (4 stored diagonals: the quad
    diagonal coupling is non-zero) and per-sample per-element kappa through 4 multigrid levels."""
    nx, ny = 24, 16
    base = FEMesh.rectangle(nx, ny, (0.0, 2.0), (0.0, 1.0), 0.3)
    rng = np.random.default_rng(9)
    xy = base.nodes.numpy().copy().reshape(ny + 1, nx + 1, 2)
    xy[1:-1, 1:-1] += rng.uniform(-0.2, 0.2, (ny - 1, nx - 1, 2)) * np.array([2.0 / nx, 1.0 / ny])
    mesh = FEMesh(nodes=torch.from_numpy(xy.reshape(-1, 2)), elements=base.elements,
                  dirichlet_nodes=dict(base.dirichlet_nodes))
    nodes, el, bn, bv = arrays(mesh)
    B = 3
    kap = np.exp(0.5 * rng.standard_normal((B, mesh.n_elements)))
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    kt = torch.from_numpy(kap).requires_grad_(True)
    ft = torch.from_numpy(f).requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(ft)
    assert solver.last_info.path == "lattice-mgpcg" and get_plan(mesh, torch.device("cuda", 0)).levels[0].nd == 4
    assert len(get_plan(mesh, torch.device("cuda", 0)).levels) == 4
    (u ** 2).sum().backward()
    for b in range(B):
        uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kap[b], f[b], lambda u: 2 * u)
        assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
        assert rel_err(kt.grad[b].numpy(), dko) < RTOL_GRAD
        assert rel_err(ft.grad[b].numpy(), dfo) < RTOL_GRAD
    assert solver.last_info.iterations < 40      # multigrid, not plain CG


@pytest.mark.parametrize("method", ["auto", "ell"])
def test_2d_64_batch_vs_oracle(method):
    mesh = FEMesh.rectangle(64, 64)
    nodes, el, bn, bv = arrays(mesh)
    B = 6
    gen = torch.Generator().manual_seed(2024)
    kap = 0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)
    f = 1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    kt = kap.clone().requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt, method=method)
    u = solver(f)
    (u ** 2).sum().backward()
    assert solver.last_info.not_converged == 0
    for b in range(B):
        uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, float(kap[b]), f[b].numpy(), lambda u: 2 * u)
        assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
        assert abs(float(kt.grad[b]) - dko.sum()) <= RTOL_GRAD * abs(dko.sum())


@pytest.mark.parametrize("shape", [(202, 70, 0.35), (203, 70, 1.5), (40, 48, 1.5)])
@pytest.mark.parametrize("graded", [False, True])
def test_lattice_strip_kernels_batch64(graded, shape):
    """Batch >= 64 on a grid of >= 192 columns and >= 64 rows: the register-window strip kernels
    (fused first sweeps, fused residual + restriction, fused prolongation, fused CG step, strip load
    vector) are the ones that run on the fine level -- per-sample scalar kappa (shared matrix + scale)
    and per-sample per-element kappa (matrix per sample), right-angled (3 diagonals) and skewed (4),
    non-zero Dirichlet data, a width that is not a multiple of the strip width.  Shape (202, 70) has square
    cells and halves once (full coarsening with the fused transfer kernels); (203, 70) cannot be coarsened,
    so its single level is solved by the Chebyshev semi-iteration alone -- on the skewed mesh (obtuse
    triangles, spectrum of D^-1 A beyond 2) that needs the Gershgorin bound of the coarsest operator."""
    nx, ny, yr = shape
    base = FEMesh.rectangle(nx, ny, (0.0, 1.0), (0.0, yr), 0.2)
    rng = np.random.default_rng(21)
    xy = base.nodes.numpy().copy().reshape(ny + 1, nx + 1, 2)
    if graded:
        xy[1:-1, 1:-1] += rng.uniform(-0.2, 0.2, (ny - 1, nx - 1, 2)) * np.array([1.0 / nx, yr / ny])
    mesh = FEMesh(nodes=torch.from_numpy(xy.reshape(-1, 2)), elements=base.elements,
                  dirichlet_nodes=dict(base.dirichlet_nodes))
    nodes, el, bn, bv = arrays(mesh)
    B = 64
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    for kap in (rng.uniform(0.5, 2.0, B), np.exp(0.4 * rng.standard_normal((B, mesh.n_elements)))):
        kt = torch.from_numpy(kap).requires_grad_(True)
        ft = torch.from_numpy(f).requires_grad_(True)
        solver = DifferentiableFESolver(mesh, kt)
        u = solver(ft)
        assert solver.last_info.path == "lattice-mgpcg"
        (u ** 2).sum().backward()
        for b in (0, 31, 63):
            uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kap[b], f[b], lambda u: 2 * u)
            assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
            got = kt.grad[b].numpy()
            assert rel_err(got, dko if kap.ndim == 2 else dko.sum()) < RTOL_GRAD
            assert rel_err(ft.grad[b].numpy(), dfo) < RTOL_GRAD


def test_lattice_operator_kernels_vs_dense():
    """diffhe_lattice_apply / diffhe_lattice_smooth (strip and simple variants) against the
    oracle's dense K with identity Dirichlet rows."""
    from diffhe.solver import _Engine, K_SAMPLE_ELEM
    from diffhe.plan import padded_batch
    nx, ny = 20, 36
    mesh = FEMesh.rectangle(nx, ny, (0.0, 1.0), (0.0, 2.0), 0.0)
    nodes, el, bn, bv = arrays(mesh)
    n, m = mesh.n_nodes, mesh.n_elements
    dev = torch.device("cuda", 0)
    plan = get_plan(mesh, dev)
    rng = np.random.default_rng(8)
    L = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    for B in (4, 64):
        kap = np.exp(0.3 * rng.standard_normal((B, m)))
        eng = _Engine(plan, 1e-12, 100, 1, "gather")
        Bp = padded_batch(B)
        vals, Bv, scale, lift, _ = eng.lattice_assemble(torch.from_numpy(kap), K_SAMPLE_ELEM, B, Bp)
        arr = eng.lattice_levels(vals)
        x = rng.standard_normal((n, Bp))
        x[bn] = 0.0
        rhs = rng.standard_normal((n, Bp))
        rhs[bn] = 0.0
        xd, rd = torch.from_numpy(x).to(dev), torch.from_numpy(rhs).to(dev)
        y = torch.empty_like(xd)
        xo = torch.empty_like(xd)
        part = torch.empty(L.diffhe_lattice_blocks(n, Bp) * Bp, dtype=T64, device=dev)
        _hip.check(L.diffhe_lattice_apply(arr, Bv, None, _hip.ptr(xd), _hip.ptr(y), _hip.ptr(part), Bp, st), "apply")
        _hip.check(L.diffhe_lattice_smooth(arr, Bv, None, _hip.ptr(rd), _hip.ptr(xd), _hip.ptr(xo), 0.8, Bp, st),
                   "smooth")
        for b in (0, B - 1):
            K, _ = orc.assemble_dense(nodes, el, kap[b], np.zeros(n))
            K[bn, :] = 0.0
            K[:, bn] = 0.0
            K[bn, bn] = 1.0
            yref = K @ x[:, b]
            assert rel_err(y[:, b].cpu().numpy(), yref) < 1e-13
            xref = x[:, b] + 0.8 * (rhs[:, b] - yref) / np.diag(K)
            assert rel_err(xo[:, b].cpu().numpy(), xref) < 1e-13


def _dst_solve_unit_square(N, kappa, F_interior):
    """Exact solve of kappa * (5-point Laplacian) u = F on the (N-1)^2 interior nodes of the
    uniform unit square via DST-I (K_free = kappa * (diag 4, off -1))."""
    from scipy.fft import dstn, idstn
    k = np.arange(1, N)
    lam = 4.0 - 2.0 * np.cos(np.pi * k / N)[:, None] - 2.0 * np.cos(np.pi * k / N)[None, :]
    return idstn(dstn(F_interior, type=1) / (kappa * lam), type=1)


@pytest.mark.parametrize("N,B", [(256, 64), (1024, 64)])
def test_full_size_properties_and_exact_dst(N, B):
    """BASELINE-size checks that need no O(n^3) oracle: (i) the exact DST-I solution of the
    assembled 5-point system for random f, (ii) dL/dkappa = -<gbar,u>/kappa = -2L/kappa,
    (iii) linearity in f, (iv) homogeneity u(c*kappa) = u(kappa)/c."""
    mesh = FEMesh.rectangle(N, N)
    n = mesh.n_nodes
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(4096)
    kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)).to(dev).requires_grad_(True)
    f = (1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)).to(dev)
    solver = DifferentiableFESolver(mesh, kappa)
    u = solver(f)
    assert solver.last_info.path == "lattice-mgpcg" and solver.last_info.not_converged == 0
    L = (u ** 2).sum(dim=1)
    L.sum().backward()
    ref = -2.0 * L.detach() / kappa.detach()
    assert float(((kappa.grad - ref).abs() / ref.abs()).max()) < RTOL_GRAD
    # (i) exact discrete solution: F = M f restricted to the interior (oracle load vector), DST solve
    nodes, el, bn, bv = arrays(mesh)
    for b in (0, B - 1):
        F = orc.load_vector(nodes, el, f[b].cpu().numpy()).reshape(N + 1, N + 1)[1:-1, 1:-1]
        ue = np.zeros((N + 1, N + 1))
        ue[1:-1, 1:-1] = _dst_solve_unit_square(N, float(kappa[b]), F)
        assert rel_err(u[b].detach().cpu().numpy(), ue.ravel()) < RTOL_U
    # (iii) linearity in f and (iv) homogeneity in kappa, on the first 64 samples
    with torch.no_grad():
        u2 = DifferentiableFESolver(mesh, 2.0 * kappa.detach())(3.0 * f)
    assert float((u2 - 1.5 * u.detach()).abs().max() / u.detach().abs().max()) < RTOL_U


def test_kappa_recovery_2d_inverse_problem():
    """BASELINE config 5 pattern at test size: recover per-sample kappa by Adam through the adjoint
    solve on a 2D mesh (examples/poisson_1d_demo.py:102-110 generalised to a batch)."""
    mesh = FEMesh.rectangle(32, 32)
    dev = torch.device("cuda", 0)
    B = 8
    gen = torch.Generator().manual_seed(5)
    k_true = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)).to(dev)
    f = torch.ones(B, mesh.n_nodes, dtype=T64, device=dev)
    with torch.no_grad():
        u_data = DifferentiableFESolver(mesh, k_true)(f)
    k = torch.ones(B, dtype=T64, device=dev, requires_grad=True)
    opt = torch.optim.Adam([k], lr=0.1)
    for _ in range(150):
        opt.zero_grad()
        u = DifferentiableFESolver(mesh, k.abs())(f)
        loss = ((u - u_data) ** 2).mean(dim=1).sum() * 1e4
        loss.backward()
        opt.step()
    assert float((k.detach().abs() - k_true).abs().max()) < 2e-2


def test_neural_pde_against_hip_fem_target():
    """reference tests/test_neural.py:39-90 with the FEM target coming from the HIP solver:
    the fem_match loss decreases, the trained net lands within 5 % of the FEM solution, and
    the Dirichlet mask keeps the boundary at zero."""
    from diffhe import NeuralPDE
    torch.manual_seed(42)
    mesh = FEMesh.line(n_elements=20)
    model = NeuralPDE(mesh, hidden_dim=32, n_layers=2)
    losses = model.train_pde(lambda x: torch.ones_like(x), n_epochs=1500, lr=3e-3, mode="fem_match", verbose=False)
    assert losses[-1] < 0.1 * losses[0]
    u_nn = model().detach()
    u_fem = DifferentiableFESolver(mesh)(torch.ones(21, dtype=T64))
    assert float((u_nn - u_fem).abs().max() / u_fem.abs().max()) < 0.05
    assert abs(float(u_nn[0])) < 1e-12 and abs(float(u_nn[-1])) < 1e-12
    # 2D fem_match works here (it raises in the reference)
    mesh2 = FEMesh.rectangle(8, 8)
    loss2 = PhysicsLoss(mesh2, lambda xy: torch.ones(xy.shape[0], dtype=T64), mode="fem_match")
    v = loss2(torch.zeros(mesh2.n_nodes, dtype=T64))
    u2 = DifferentiableFESolver(mesh2)(torch.ones(mesh2.n_nodes, dtype=T64))
    assert abs(float(v) - float((u2 ** 2).mean())) < 1e-15


@pytest.mark.parametrize("N,B", [(256, 100), (192, 640)])
def test_lattice_ragged_and_large_batches(N, B):
    """Batch sizes that are not a multiple of the wave (padding samples carry kappa = 1, f = 0 and
    must stay inert) and batches larger than one sample-chunk row of the grid, on the strip /
    fused / full-multigrid path; checked against the exact DST solution and the adjoint identity."""
    mesh = FEMesh.rectangle(N, N)
    n = mesh.n_nodes
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(7)
    kappa = (0.5 + 1.5 * torch.rand(B, generator=gen, dtype=T64)).to(dev).requires_grad_(True)
    f = (1 + 0.5 * torch.randn(B, n, generator=gen, dtype=T64)).to(dev)
    solver = DifferentiableFESolver(mesh, kappa)
    u = solver(f)
    assert u.shape == (B, n) and solver.last_info.not_converged == 0
    L = (u ** 2).sum(dim=1)
    L.sum().backward()
    ref = -2.0 * L.detach() / kappa.detach()
    assert float(((kappa.grad - ref).abs() / ref.abs()).max()) < RTOL_GRAD
    nodes, el, bn, bv = arrays(mesh)
    for b in (0, B // 2, B - 1):
        F = orc.load_vector(nodes, el, f[b].cpu().numpy()).reshape(N + 1, N + 1)[1:-1, 1:-1]
        ue = np.zeros((N + 1, N + 1))
        ue[1:-1, 1:-1] = _dst_solve_unit_square(N, float(kappa[b]), F)
        assert rel_err(u[b].detach().cpu().numpy(), ue.ravel()) < RTOL_U


def test_empty_and_degenerate_inputs():
    """Edge cases: zero forcing with non-zero Dirichlet data, a degenerate (zero-area) triangle is
    skipped (reference solver.py:120-121), wrong sizes raise."""
    mesh = FEMesh.rectangle(6, 6, bc_value=0.75)
    u = DifferentiableFESolver(mesh)(torch.zeros(mesh.n_nodes, dtype=T64))
    assert float((u - 0.75).abs().max()) < 1e-12                  # harmonic extension of a constant
    with pytest.raises(ValueError):
        DifferentiableFESolver(mesh)(torch.zeros(mesh.n_nodes + 1, dtype=T64))
    with pytest.raises(ValueError):
        DifferentiableFESolver(mesh, torch.ones(3, dtype=T64))(torch.zeros(5, mesh.n_nodes, dtype=T64))
    # unstructured mesh with one degenerate triangle appended: same solution as without it
    base = FEMesh.rectangle(4, 4)
    nodes, el, bn, bv = arrays(base)
    perm = np.random.default_rng(1).permutation(len(el))
    el2 = np.vstack([el[perm], [[0, 1, 1]]])                       # repeated vertex -> area 0
    mesh2 = FEMesh(nodes=base.nodes, elements=torch.from_numpy(el2), dirichlet_nodes=dict(base.dirichlet_nodes))
    f = torch.from_numpy(1 + 0.3 * np.random.default_rng(2).standard_normal(base.n_nodes))
    u2 = DifferentiableFESolver(mesh2, 1.3)(f)
    uo = orc.solve(nodes, el, bn, bv, 1.3, f.numpy())
    assert rel_err(u2.numpy(), uo) < RTOL_U


def test_custom_op_state_lifecycle():
    """The adjoint state lives exactly as long as the autograd graph of its solve: nothing is saved without grad,
    a plain backward frees it, retain_graph keeps it for further backward passes (the reference, being pure
    autograd, supports that), dropped graphs do not leak, and gradcheck (many backward passes) works."""
    import gc
    from diffhe import solver as S
    mesh = FEMesh.rectangle(8, 8)
    k = torch.tensor(1.3, dtype=T64, requires_grad=True)
    f = torch.ones(mesh.n_nodes, dtype=T64)
    gc.collect()
    S._STATES.clear()
    with torch.no_grad():
        DifferentiableFESolver(mesh, k)(f)
    assert len(S._STATES) == 0                                  # nothing saved without grad
    u = DifferentiableFESolver(mesh, k)(f)
    assert len(S._STATES) == 1
    L = (u ** 2).sum()
    L.backward(retain_graph=True)
    g1 = k.grad.clone()
    assert len(S._STATES) == 1                                  # graph retained: state kept
    L.backward()
    assert torch.allclose(k.grad, 2 * g1)                       # second pass accumulated the same gradient
    del u, L
    gc.collect()
    assert len(S._STATES) == 0                                  # freed with the graph
    for _ in range(10):                                         # graphs that are dropped do not leak
        DifferentiableFESolver(mesh, k)(f)
    gc.collect()
    assert len(S._STATES) == 0
    m = FEMesh.rectangle(4, 3, bc_value=0.2)
    kk = torch.tensor([0.8, 1.4], dtype=T64, requires_grad=True)
    ff = (1 + 0.1 * torch.arange(2 * m.n_nodes, dtype=T64).reshape(2, -1) / m.n_nodes).requires_grad_(True)
    s = DifferentiableFESolver(m, kk)

    def fn(kv, fv):
        s._kappa = kv
        return s(fv)
    assert torch.autograd.gradcheck(fn, (kk, ff), eps=1e-6, atol=1e-6, rtol=1e-4, nondet_tol=1e-10)


@pytest.mark.parametrize("nx,ny", [(100, 37), (250, 250), (96, 72)])
def test_lattice_sizes_that_do_not_halve(nx, ny):
    """Lattices whose sizes stop halving early keep a large coarsest level: its Chebyshev solve
    (degree from the level size) keeps the PCG iteration count mesh-independent."""
    mesh = FEMesh.rectangle(nx, ny, (0.0, 1.0), (0.0, 0.5), 0.1)
    nodes, el, bn, bv = arrays(mesh)
    rng = np.random.default_rng(31)
    B = 4
    kap = rng.uniform(0.5, 2.0, B)
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    kt = torch.from_numpy(kap).requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(torch.from_numpy(f))
    (u ** 2).sum().backward()
    assert solver.last_info.path == "lattice-mgpcg" and solver.last_info.iterations <= 16
    for b in (0, B - 1):
        uo, dko, _ = orc.solve_with_adjoint(nodes, el, bn, bv, kap[b], f[b], lambda u: 2 * u)
        assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
        assert abs(float(kt.grad[b]) - dko.sum()) <= RTOL_GRAD * abs(dko.sum())


def test_fused_cg_step_kernel_vs_reference_formulas():
    """diffhe_lattice_cg_step: p' = z + beta p, x += alpha p, Ap = A p', partial dots -- against torch on
    the same operator (dense K from the oracle), fp32 and fp64 z, first and later iterations."""
    from diffhe.solver import _Engine, K_SAMPLE
    nx, ny = 200, 70
    mesh = FEMesh.rectangle(nx, ny, (0.0, 2.0), (0.0, 0.7), 0.0)
    nodes, el, bn, bv = arrays(mesh)
    n = mesh.n_nodes
    dev = torch.device("cuda", 0)
    plan = get_plan(mesh, dev)
    L = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    B = 64
    rng = np.random.default_rng(3)
    kap = rng.uniform(0.5, 2.0, B)
    eng = _Engine(plan, 1e-12, 100, 1, "gather")
    vals, Bv, scale, _, _ = eng.lattice_assemble(torch.from_numpy(kap), K_SAMPLE, B, B)
    arr = eng.lattice_levels(vals)
    free = np.ones(n, bool)
    free[bn] = False
    mk = lambda: torch.from_numpy(rng.standard_normal((n, B)) * free[:, None]).to(dev)   # noqa: E731
    z64, p_in, x0 = mk(), mk(), mk()
    alpha = torch.from_numpy(rng.uniform(0.1, 1.0, B)).to(dev)
    beta = torch.from_numpy(rng.uniform(0.1, 1.0, B)).to(dev)
    import scipy.sparse as sp
    K1, _ = orc.assemble_sparse(nodes, el, 1.0, np.zeros(n))
    K1 = sp.diags(free.astype(float)) @ K1 @ sp.diags(free.astype(float)) + sp.diags((~free).astype(float))
    part = torch.empty(L.diffhe_lattice_blocks(n, B) * B, dtype=T64, device=dev)
    for z_fp32 in (0, 1):
        z = z64.float() if z_fp32 else z64
        zr = z.double().cpu().numpy()
        p_st = p_in.float() if z_fp32 else p_in          # the direction is stored in z's precision
        for first in (1, 0):
            x = x0.clone()
            p_out = torch.empty_like(p_st)
            Ap = torch.empty_like(p_in)
            rc = L.diffhe_lattice_cg_step(arr, Bv, _hip.ptr(scale), _hip.ptr(z), z_fp32, _hip.ptr(p_st), _hip.ptr(p_out),
                                          _hip.ptr(x), _hip.ptr(alpha), _hip.ptr(beta), first, _hip.ptr(Ap),
                                          _hip.ptr(part), B, st)
            assert rc == 0
            p_old = p_st.double().cpu().numpy()
            p_ref = zr if first else zr + beta.cpu().numpy() * p_old
            if z_fp32:
                p_ref = p_ref.astype(np.float32).astype(np.float64)     # stored rounded; Ap uses the stored p
            x_ref = x0.cpu().numpy() if first else x0.cpu().numpy() + alpha.cpu().numpy() * p_old
            Ap_ref = (K1 @ p_ref) * np.where(free[:, None], kap[None, :], 1.0)
            assert rel_err(p_out.double().cpu().numpy(), p_ref) < 1e-14
            assert rel_err(x.cpu().numpy(), x_ref) < 1e-14
            assert rel_err(Ap.cpu().numpy(), Ap_ref) < 1e-13


@pytest.mark.parametrize("nx,ny,xr", [(96, 96, (0.0, 8.0)), (256, 32, (0.0, 1.0)), (40, 160, (0.0, 1.0))])
def test_anisotropic_lattices_semi_coarsening(nx, ny, xr):
    """Elongated cells: levels are coarsened in the strongly coupled direction only until the cells are
    isotropic, so the iteration count stays at the isotropic level; values and gradients match the oracle."""
    mesh = FEMesh.rectangle(nx, ny, xr, (0.0, 1.0), 0.05)
    nodes, el, bn, bv = arrays(mesh)
    rng = np.random.default_rng(17)
    B = 3
    kap = np.exp(0.3 * rng.standard_normal((B, mesh.n_elements)))        # per-element: kappa restriction too
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    kt = torch.from_numpy(kap).requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(torch.from_numpy(f))
    (u ** 2).sum().backward()
    levels = [(lv.nx, lv.ny) for lv in get_plan(mesh, torch.device("cuda", 0)).levels]
    assert any(a[0] == b[0] or a[1] == b[1] for a, b in zip(levels, levels[1:])), levels   # a semi step exists
    assert solver.last_info.iterations <= 16, (solver.last_info, levels)
    for b in range(B):
        uo, dko, _ = orc.solve_with_adjoint(nodes, el, bn, bv, kap[b], f[b], lambda u: 2 * u)
        assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
        assert rel_err(kt.grad[b].numpy(), dko) < RTOL_GRAD


def _unstructured_mesh(N, seed=0):
    """A genuinely unstructured numbering and geometry: jittered lattice nodes, randomly permuted node ids
    and element order, random diagonal flips."""
    rng = np.random.default_rng(seed)
    base = FEMesh.rectangle(N, N)
    xy = base.nodes.numpy().copy().reshape(N + 1, N + 1, 2)
    xy[1:-1, 1:-1] += rng.uniform(-0.25, 0.25, (N - 1, N - 1, 2)) / N
    el = base.elements.numpy().copy().reshape(N * N, 2, 3)
    flip = rng.random(N * N) < 0.5                              # other diagonal: [a,b,c], [a,c,d]
    a, b, d = el[:, 0, 0], el[:, 0, 1], el[:, 0, 2]
    c = el[:, 1, 1]
    el[flip, 0] = np.stack([a, b, c], 1)[flip]
    el[flip, 1] = np.stack([a, c, d], 1)[flip]
    el = el.reshape(-1, 3)
    n = (N + 1) ** 2
    perm = rng.permutation(n)
    el = perm[el][rng.permutation(len(el))]
    nodes = np.empty((n, 2))
    nodes[perm] = xy.reshape(-1, 2)
    bc = {int(perm[k]): 0.1 for k in base.dirichlet_nodes}
    return FEMesh(nodes=torch.from_numpy(nodes), elements=torch.from_numpy(el), dirichlet_nodes=bc)


@pytest.mark.parametrize("N,B", [(24, 5), (96, 64)])
def test_unstructured_mesh_amg_vs_oracle(N, B):
    """General path on an unstructured mesh: aggregation-AMG PCG (values + both gradients vs the oracle),
    and far fewer iterations than the Jacobi-preconditioned CG."""
    mesh = _unstructured_mesh(N, seed=N)
    nodes, el, bn, bv = arrays(mesh)
    rng = np.random.default_rng(5)
    kap = np.exp(0.3 * rng.standard_normal((B, mesh.n_elements)))
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    kt = torch.from_numpy(kap).requires_grad_(True)
    ft = torch.from_numpy(f).requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(ft)
    assert solver.last_info.path == "ell-amgpcg" and solver.last_info.not_converged == 0
    its_amg = solver.last_info.iterations
    (u ** 2).sum().backward()
    for b in (0, B - 1):
        uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kap[b], f[b], lambda u: 2 * u)
        assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
        assert rel_err(kt.grad[b].numpy(), dko) < RTOL_GRAD
        assert rel_err(ft.grad[b].numpy(), dfo) < RTOL_GRAD
    with torch.no_grad():
        sj = DifferentiableFESolver(mesh, kt.detach(), method="ell-jacobi")
        uj = sj(ft.detach())
    assert sj.last_info.path == "ell-pcg" and rel_err(uj.numpy(), u.detach().numpy()) < RTOL_U
    if N >= 96:
        assert its_amg * 3 < sj.last_info.iterations, (its_amg, sj.last_info.iterations)


def _chain_oracle(mode, nodes, el, bn, bv, kappa, f, gbar_fn):
    """Extended-precision solve of the system each chain mode stands for: "reference" = the matrix and load the
    reference assembles in fp64 (rounded weights and diagonal), "exact" = the unrounded weighted Laplacian.
    (The fp64 LU of the oracle / the reference is itself up to 3e-10 off in per-element gradients at these sizes --
    measured against this routine -- so it cannot be the yardstick here; at 10^4 uniform elements the reference's
    own solve IS the yardstick: test_config2_shape_1d_10000.)"""
    return orc.chain_solve_longdouble(nodes, bn, bv, kappa, f, gbar_fn, reference_rounding=(mode == "reference"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["reference", "exact"])
def test_long_chain_global_staging_path(mode):
    """Chains longer than 10 240 elements leave the register kernel for the globally staged one."""
    mesh = FEMesh.line(12_345)
    nodes, el, bn, bv = arrays(mesh)
    B = 3
    gen = torch.Generator().manual_seed(7)
    f = 1 + 0.5 * torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64)
    k = torch.tensor([0.7, 1.0, 1.9], dtype=T64, requires_grad=True)
    fc = f.cuda().requires_grad_(True)
    u = DifferentiableFESolver(mesh, k, chain=mode)(fc)
    (0.5 * (u ** 2).sum()).backward()
    for b in range(B):
        ux, dkx, dfx = _chain_oracle(mode, nodes, el, bn, bv, float(k[b]), f[b].numpy(), lambda u: u)
        assert rel_err(u[b].detach().cpu().numpy(), ux) < RTOL_U
        assert rel_err(fc.grad[b].cpu().numpy(), dfx) < RTOL_GRAD
        assert abs(float(k.grad[b]) - dkx.sum()) < RTOL_GRAD * abs(dkx.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["reference", "exact"])
@pytest.mark.parametrize("N", [255, 256, 257, 1024, 1025, 4096, 4097, 10240, 10241])
def test_chain_kernel_size_boundaries(N, mode):
    """Each register-kernel instantiation at its largest size and the next kernel at its smallest,
    per-element kappa, non-uniform nodes, an interior Dirichlet node and a Neumann right end."""
    rng = np.random.default_rng(N)
    x = (np.arange(N + 1) + np.concatenate([[0.0], rng.uniform(-0.3, 0.3, N - 1), [0.0]])) / N
    mesh = FEMesh(nodes=torch.from_numpy(x[:, None]), elements=torch.stack([torch.arange(N), torch.arange(1, N + 1)], 1),
                  dirichlet_nodes={0: 0.3, N // 3: -0.2})
    nodes, el, bn, bv = arrays(mesh)
    kap = torch.from_numpy(rng.uniform(0.5, 2.0, N)).requires_grad_(True)
    f = torch.from_numpy(1 + 0.3 * rng.standard_normal(N + 1))
    fc = f.cuda().requires_grad_(True)
    u = DifferentiableFESolver(mesh, kap, chain=mode)(fc)
    w = rng.standard_normal(N + 1)
    (torch.from_numpy(w).cuda() * u).sum().backward()
    ux, dkx, dfx = _chain_oracle(mode, nodes, el, bn, bv, kap.detach().numpy(), f.numpy(), lambda u: w)
    assert rel_err(u.detach().cpu().numpy(), ux) < RTOL_U
    assert rel_err(fc.grad.cpu().numpy(), dfx) < RTOL_GRAD
    assert rel_err(kap.grad.numpy(), dkx) < RTOL_GRAD


@pytest.mark.gpu
def test_many_pending_differentiated_solves_one_backward():
    """100 unbatched solves summed into one loss, a single backward: every adjoint state stays alive with its graph
    (there is no cap on pending solves)."""
    mesh = FEMesh.line(24)
    k = torch.tensor(1.3, dtype=T64, requires_grad=True)
    gen = torch.Generator().manual_seed(3)
    fs = 1 + 0.5 * torch.randn(100, mesh.n_nodes, generator=gen, dtype=T64)
    solver = DifferentiableFESolver(mesh, k)
    total = sum((solver(fs[i]) ** 2).sum() for i in range(100))
    total.backward()
    kb = torch.tensor(1.3, dtype=T64, requires_grad=True)
    (DifferentiableFESolver(mesh, kb)(fs) ** 2).sum().backward()          # the same as one batched solve
    assert abs(float(k.grad) - float(kb.grad)) < 1e-12 * abs(float(kb.grad))


@pytest.mark.gpu
def test_attainable_accuracy_floor_of_the_stopping_test():
    """On a smooth right-hand side the fp64 residual stagnates near u |A| |x| / |b| (> tol for fine meshes): the
    default stop is floored at half that level.  It must save iterations, not accuracy: the error against the
    EXACT solution of the discrete system (DST-I) stays the one of the un-floored solve."""
    N, B = 512, 64
    mesh = FEMesh.rectangle(N, N)
    dev = torch.device("cuda", 0)
    kappa = torch.linspace(0.5, 2.0, B, dtype=T64, device=dev)
    f = torch.ones(B, mesh.n_nodes, dtype=T64, device=dev)
    nodes, el, bn, bv = arrays(mesh)
    F = orc.load_vector(nodes, el, np.ones(mesh.n_nodes)).reshape(N + 1, N + 1)[1:-1, 1:-1]
    out = {}
    for floor in (1, 0):
        solver = DifferentiableFESolver(mesh, kappa, mg=dict(floor=floor))
        u = solver(f)
        assert solver.last_info.not_converged == 0
        errs = []
        for b in (0, B - 1):
            ue = np.zeros((N + 1, N + 1))
            ue[1:-1, 1:-1] = _dst_solve_unit_square(N, float(kappa[b]), F)
            errs.append(rel_err(u[b].cpu().numpy(), ue.ravel()))
        out[floor] = (solver.last_info.iterations, max(errs))
    assert out[1][0] <= out[0][0]                     # never more iterations
    assert out[1][1] < RTOL_U and out[0][1] < RTOL_U
    assert out[1][1] < 3 * out[0][1] + 1e-12          # and the same accuracy


@pytest.mark.gpu
def test_neural_pde_trains_on_the_device():
    """with the module moved to the GPU the whole train_pde loop (network, mask, FEM target
    from the HIP solver, loss, Adam) runs there without host round trips in the loop."""
    from diffhe import NeuralPDE
    torch.manual_seed(0)
    mesh = FEMesh.rectangle(12, 12)
    model = NeuralPDE(mesh, hidden_dim=16, n_layers=2).cuda()
    losses = model.train_pde(lambda xy: torch.ones(xy.shape[0], dtype=T64, device=xy.device), n_epochs=300, lr=5e-3,
                             mode="fem_match", verbose=False)
    u = model()
    assert u.is_cuda and losses[-1] < 0.2 * losses[0]
    bc = torch.as_tensor(list(mesh.dirichlet_nodes.keys()), device=u.device)
    assert float(u[bc].abs().max()) == 0.0


@pytest.mark.gpu
def test_2d_second_order_convergence():
    """2D analogue of reference tests/test_fem.py:114-132 (on the reference's roadmap, README.md:139-143): for
    -lap u = 2 pi^2 sin(pi x) sin(pi y) on the unit square the nodal max error falls ~4x per mesh doubling."""
    errs = []
    for N in (16, 32, 64, 128):
        mesh = FEMesh.rectangle(N, N)
        x, y = mesh.nodes[:, 0], mesh.nodes[:, 1]
        exact = torch.sin(math.pi * x) * torch.sin(math.pi * y)
        u = DifferentiableFESolver(mesh)(2 * math.pi ** 2 * exact)
        errs.append(float((u.cpu() - exact).abs().max()))
    for a, b in zip(errs, errs[1:]):
        assert 3.3 < a / b < 4.7


@pytest.mark.gpu
def test_lattice_with_many_interior_dirichlet_nodes():
    """Pinned interior nodes: a few stay on the geometric-multigrid path; 5 % of the nodes (which the coarse
    lattices cannot represent) route the mesh to the aggregation-multigrid path, forward AND adjoint."""
    for frac, path in ((0.001, "lattice-mgpcg"), (0.05, "ell-amgpcg")):
        rng = np.random.default_rng(9)
        base = FEMesh.rectangle(96, 80, (0.0, 1.2), (0.0, 1.0), 0.0)
        d = dict(base.dirichlet_nodes)
        for k in rng.choice(base.n_nodes, int(frac * base.n_nodes), replace=False):
            d[int(k)] = float(rng.uniform(-1, 1))
        mesh = FEMesh(nodes=base.nodes, elements=base.elements, dirichlet_nodes=d)
        nodes, el, bn, bv = arrays(mesh)
        B = 5
        kap = rng.uniform(0.5, 2.0, B)
        f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
        kt = torch.from_numpy(kap).requires_grad_(True)
        ft = torch.from_numpy(f).requires_grad_(True)
        solver = DifferentiableFESolver(mesh, kt)
        u = solver(ft)
        (u ** 2).sum().backward()
        info = solver.last_info
        assert info.path == path and info.not_converged == 0
        assert info.adj_iterations <= 3 * info.iterations + 10      # the adjoint uses the same preconditioner
        for b in (0, B - 1):
            uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kap[b], f[b], lambda u: 2 * u)
            assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
            assert abs(float(kt.grad[b]) - dko.sum()) < RTOL_GRAD * abs(dko.sum())
            assert rel_err(ft.grad[b].numpy(), dfo) < RTOL_GRAD


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["auto", "ell"])
def test_solution_is_invariant_to_the_magnitude_of_the_data(method):
    """The fp32 copies that feed the multigrid preconditioners are taken of the residual scaled by a per-sample power
    of two (~ 1 / |b|), so the data may have any magnitude: the same forcing at amplitudes 1e-100 .. 1e100 gives the
    same iteration counts and (after rescaling) the same solution and gradients to the last digits."""
    mesh = FEMesh.rectangle(48, 40)
    rng = np.random.default_rng(3)
    f0 = 1 + 0.5 * rng.standard_normal((6, mesh.n_nodes))
    ref = None
    for amp in (1.0, 1e-100, 1e-35, 1e36, 1e100):
        k = torch.full((6,), 1.3, dtype=T64, requires_grad=True)
        solver = DifferentiableFESolver(mesh, k, method=method)
        u = solver(torch.from_numpy(f0 * amp))
        (u.sum() / amp).backward()
        un = u.detach().numpy() / amp
        info = solver.last_info
        assert info.not_converged == 0
        if ref is None:
            ref = (un, k.grad.clone(), info.iterations, info.adj_iterations)
            continue
        assert rel_err(un, ref[0]) < 1e-13 and float((k.grad - ref[1]).abs().max() / ref[1].abs().max()) < 1e-12
        assert abs(info.iterations - ref[2]) <= 1 and abs(info.adj_iterations - ref[3]) <= 1


@pytest.mark.gpu
def test_per_element_gradients_on_rough_data_at_strip_size():
    """Regression of a randomised-sweep finding: dL/dkappa_e differences the nodal fields, so it amplifies the rough
    part of the solver error by ~ the mesh resolution; with the floored 1e-13 stop a 288 x 296 lattice with a
    log-normal field and random forcing was off by 2.2e-10.  Per-element kappa now runs to 1e-14 without the floor."""
    nx, ny = 288, 296
    mesh = FEMesh.rectangle(nx, ny, (0.0, 1.0), (0.0, ny / nx), 0.2)
    nodes, el, bn, bv = arrays(mesh)
    rng = np.random.default_rng(53)
    B = 64
    kap = np.exp(0.4 * rng.standard_normal((B, mesh.n_elements)))
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    kt = torch.from_numpy(kap).cuda().requires_grad_(True)
    ft = torch.from_numpy(f).cuda().requires_grad_(True)
    solver = DifferentiableFESolver(mesh, kt)
    u = solver(ft)
    (u ** 2).sum().backward()
    assert solver.tol == 1e-14 and solver.last_info.not_converged == 0
    for b in (0, B - 1):
        uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kap[b], f[b], lambda u: 2 * u)
        assert rel_err(u[b].detach().cpu().numpy(), uo) < RTOL_U
        assert rel_err(ft.grad[b].cpu().numpy(), dfo) < RTOL_GRAD
        assert rel_err(kt.grad[b].cpu().numpy(), dko) < 0.5 * RTOL_GRAD


@pytest.mark.gpu
def test_assembled_operators_are_bit_identical_to_the_reference_order():
    """diffhe_ell_assemble_rows_ref forms every contribution as (kappa * t) / (4 area) with separately rounded
    operations and adds them in element order -- solver.py:139-140 verbatim -- so the values the lattice and the
    general path store equal, bit for bit, the oracle's K (which reproduces the reference's K bit for bit on the G5
    fixtures): uniform and skewed mesh, scalar and per-element kappa."""
    from diffhe.solver import _Engine, K_SCALAR, K_ELEM
    rng = np.random.default_rng(0)
    for jitter in (0.0, 0.25):
        nx, ny = 60, 44
        base = FEMesh.rectangle(nx, ny, (0, 1.7), (0, 1.0), 0.2)
        xy = base.nodes.numpy().copy().reshape(ny + 1, nx + 1, 2)
        if jitter: xy[1:-1, 1:-1] += rng.uniform(-jitter, jitter, (ny - 1, nx - 1, 2)) * np.array([1.7 / nx, 1.0 / ny])
        mesh = FEMesh(nodes=torch.from_numpy(xy.reshape(-1, 2)), elements=base.elements, dirichlet_nodes=dict(base.dirichlet_nodes))
        n, m = mesh.n_nodes, mesh.n_elements
        plan = get_plan(mesh, torch.device("cuda", 0))
        eng = _Engine(plan, 1e-12, 100, 1, "gather")
        for mode, kap in ((K_SCALAR, np.array(1.37)), (K_ELEM, np.exp(0.4 * rng.standard_normal(m)))):
            # factor=False: kappa folded into the stored values (what partly-Neumann lattices and per-element kappa
            # use).  On closed lattices scalar kappa is kept FACTORED, K = kappa * K_1 with the unit matrix stored:
            # bit-identical to the reference only where the assembly is exact (power-of-two meshes), else to a few ulp.
            vals, Bv, scale, lift, _ = eng.lattice_assemble(torch.from_numpy(kap), mode, 1, 1, factor=False)
            v = vals[0].cpu().numpy().reshape(-1, n)
            if mode == K_SCALAR:
                vf, Bvf, scf, _, _ = eng.lattice_assemble(torch.from_numpy(kap), mode, 1, 1, factor=True)
                assert Bvf == 1 and float(scf[0]) == float(kap)
                vu = vf[0].cpu().numpy().reshape(-1, n) * float(kap)
                bc_ = np.zeros(n, bool); bc_[list(mesh.dirichlet_nodes.keys())] = True
                for k_, off_ in enumerate([0, 1, nx + 1, nx][:v.shape[0]]):       # free rows x free columns only
                    i_ = np.arange(n - off_); ok_ = ~(bc_[i_] | bc_[i_ + off_]) & (v[k_, i_] != 0)
                    # a few ulp of the contributions an entry is summed from (on skewed meshes the quad-diagonal
                    # entry is a small difference of two of them): measured against the row's diagonal
                    assert np.max(np.abs(vu[k_, i_[ok_]] - v[k_, i_[ok_]]) / np.abs(v[0, i_[ok_]])) < 2e-15
            K, _ = orc.assemble_sparse(mesh.nodes.numpy(), mesh.elements.numpy(), kap, np.zeros(n)); K = K.tocsr()
            is_bc = np.zeros(n, bool); is_bc[list(mesh.dirichlet_nodes.keys())] = True
            W = nx + 1; offs = [0, 1, W, W - 1]; tot = diff = 0
            for k in range(v.shape[0]):
                i = np.arange(n - offs[k]); j = i + offs[k]; ok = ~(is_bc[i] | is_bc[j])
                ref = np.asarray(K[i[ok], j[ok]]).ravel(); got = v[k, i[ok]]; nz = ref != 0
                tot += nz.sum(); diff += (got[nz] != ref[nz]).sum()
            assert tot > 0 and diff == 0, (jitter, mode, diff)
        # general ELL path
        plan.ensure_ell()
        kap = np.exp(0.4 * rng.standard_normal(m))
        kdev, kse, ksb, Bv = eng.kappa_device(torch.from_numpy(kap), K_ELEM, 1, 1)
        vals, lift = eng.assemble(kdev, kse, ksb, Bv)
        cols = plan.cols.cpu().numpy(); v = vals.cpu().numpy()[:, :, 0]
        K, _ = orc.assemble_sparse(mesh.nodes.numpy(), mesh.elements.numpy(), kap, np.zeros(n)); K = K.tocsr()
        is_bc = np.zeros(n, bool); is_bc[list(mesh.dirichlet_nodes.keys())] = True
        tot = diff = 0
        for k in range(plan.W):
            i = np.arange(n); j = cols[k]; ok = ~(is_bc[i] | is_bc[j]) & ((k == 0) | (j != i))
            ref = np.asarray(K[i[ok], j[ok]]).ravel(); got = v[k, ok]; nz = ref != 0
            tot += nz.sum(); diff += (got[nz] != ref[nz]).sum()
        assert tot > 0 and diff == 0, (jitter, diff)


@pytest.mark.gpu
def test_per_call_options_do_not_stick_to_the_solver():
    """A per-element call runs without the residual floor and to 1e-14; the user's mg / amg settings and the
    scalar-kappa default tolerance are what the NEXT call sees (nothing is written back into the solver)."""
    mesh = FEMesh.rectangle(320, 320)
    f = torch.ones(2, mesh.n_nodes, dtype=T64)
    solver = DifferentiableFESolver(mesh, torch.tensor([1.0, 2.0], dtype=T64))
    mg0, amg0 = dict(solver.mg), dict(solver.amg)
    solver(f)
    tol_scalar = solver.tol
    solver._kappa = torch.rand(2, mesh.n_elements, dtype=T64) + 0.5
    solver(f)
    assert solver.tol == 1e-14 and solver.mg == mg0 and solver.amg == amg0
    solver._kappa = torch.tensor([1.0, 2.0], dtype=T64)
    solver(f)
    assert solver.tol == tol_scalar == 1e-12


@pytest.mark.gpu
def test_small_lattices_are_solved_directly_and_mid_sizes_use_the_dense_coarse_level():
    """The reference's own 2D sizes (up to 32 x 32, scalar kappa, unbatched): one dense product with the cached
    inverse of the unit matrix, no iteration; larger meshes cut their hierarchy at the 33^2 level the same way, for
    any batch size.  Per-element kappa (no plan-constant matrix) keeps the iterative path."""
    for N, B, path in ((32, 1, "lattice-direct"), (24, 3, "lattice-direct"), (32, 70, "lattice-direct"),
                       (64, 1, "lattice-mgpcg"), (128, 5, "lattice-mgpcg"), (256, 64, "lattice-mgpcg")):
        mesh = FEMesh.rectangle(N, N, bc_value=0.3)
        nodes, el, bn, bv = arrays(mesh)
        rng = np.random.default_rng(N + B)
        f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
        for kap in (np.float64(1.7), rng.uniform(0.5, 2.0, B)):
            kt = torch.tensor(kap, dtype=T64, requires_grad=True)
            ft = torch.from_numpy(f).requires_grad_(True)
            solver = DifferentiableFESolver(mesh, kt)
            u = solver(ft)
            assert solver.last_info.path == path, (N, B, solver.last_info)
            if path == "lattice-direct":
                assert solver.last_info.iterations == 0 and solver.last_info.max_relres < 1e-13
            (u ** 2).sum().backward()
            dk_tot = 0.0
            for b in (0, B - 1):
                kb = float(kap) if np.ndim(kap) == 0 else float(kap[b])
                uo, dko, dfo = orc.solve_with_adjoint(nodes, el, bn, bv, kb, f[b], lambda u_: 2 * u_)
                assert rel_err(u[b].detach().numpy(), uo) < RTOL_U
                assert rel_err(ft.grad[b].numpy(), dfo) < RTOL_GRAD
                if np.ndim(kap):
                    assert abs(float(kt.grad[b]) - dko.sum()) < RTOL_GRAD * abs(dko.sum())
            if np.ndim(kap) == 0 and B <= 3:
                for b in range(B):
                    kb = float(kap)
                    dk_tot += orc.solve_with_adjoint(nodes, el, bn, bv, kb, f[b], lambda u_: 2 * u_)[1].sum()
                assert abs(float(kt.grad) - dk_tot) < RTOL_GRAD * abs(dk_tot)
    mesh = FEMesh.rectangle(16, 16)
    s = DifferentiableFESolver(mesh, torch.rand(mesh.n_elements, dtype=T64) + 0.5)
    s(torch.ones(mesh.n_nodes, dtype=T64))
    assert s.last_info.path == "lattice-mgpcg"


@pytest.mark.gpu
def test_topology_optimisation_demo_reduces_compliance():
    """examples/topology_optimisation.py (the reference's roadmap item "minimise compliance"): per-element gradients
    through the adjoint drive an optimality-criteria update; two designs optimised as one batch."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "topology_optimisation.py")
    spec = importlib.util.spec_from_file_location("topology_optimisation", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rho, hist, _ = mod.optimise(N=48, iters=15, volumes=(0.3, 0.5), verbose=False)
    assert hist.shape == (15, 2)
    assert bool((hist[-1] < 0.75 * hist[0]).all())                  # compliance drops by > 25 % at fixed volume
    assert bool((rho.mean(dim=(1, 2)).cpu() <= torch.tensor([0.3, 0.5], dtype=T64) + 1e-6).all())
    assert float(rho.min()) >= 0.0 and float(rho.max()) <= 1.0
    # the gradient the update runs on is the adjoint's: check one design point against the oracle
    mesh = mod.sink_mesh(12)
    nodes, el, bn, bv = arrays(mesh)
    rng = np.random.default_rng(1)
    kap = 1e-3 + rng.uniform(0.1, 1.0, mesh.n_elements)
    kt = torch.from_numpy(kap).requires_grad_(True)
    u = DifferentiableFESolver(mesh, kt)(torch.ones(mesh.n_nodes, dtype=T64))
    u.sum().backward()
    uo, dko, _ = orc.solve_with_adjoint(nodes, el, bn, bv, kap, np.ones(mesh.n_nodes), lambda u_: np.ones_like(u_))
    assert rel_err(u.detach().numpy(), uo) < RTOL_U and rel_err(kt.grad.numpy(), dko) < RTOL_GRAD


@pytest.mark.gpu
def test_chain_many_short_segments_and_batches_beyond_the_grid_limit():
    """(i) A 20 000-element chain cut by 100 interior Dirichlet nodes: every segment fits the register kernel, so no
    staging buffer is allocated (it used to be sized by n - 1 per segment: ~3 TB at B = 1024).  (ii) 70 000
    right-hand sides on a tiny chain: beyond the 32 768 of one grid dimension, the batch continues in grid.z."""
    N = 20_000
    rng = np.random.default_rng(9)
    bc = {0: 0.1, N: -0.3}
    for k in rng.choice(np.arange(1, N), 100, replace=False):
        bc[int(k)] = float(rng.uniform(-1, 1))
    mesh = FEMesh(nodes=FEMesh.line(N).nodes, elements=FEMesh.line(N).elements, dirichlet_nodes=bc)
    plan = get_plan(mesh, torch.device("cuda", 0))
    assert plan.max_seg_len <= 10240 and plan.n_seg == 101
    assert _hip.lib().diffhe_chain1d_stage_doubles(mesh.n_nodes, 1024, plan.max_seg_len, 1) == 0
    nodes, el, bn, bv = arrays(mesh)
    B = 6
    f = 1 + 0.5 * rng.standard_normal((B, mesh.n_nodes))
    kt = torch.tensor(1.4, dtype=T64, requires_grad=True)
    ft = torch.from_numpy(f).cuda().requires_grad_(True)
    u = DifferentiableFESolver(mesh, kt)(ft)
    (u ** 2).sum().backward()
    dk = 0.0
    for b in range(B):
        uo, dko, dfo = orc.chain_solve_longdouble(nodes, bn, bv, 1.4, f[b], lambda u_: 2 * u_, reference_rounding=True)
        assert rel_err(u[b].detach().cpu().numpy(), uo) < RTOL_U
        assert rel_err(ft.grad[b].cpu().numpy(), dfo) < RTOL_GRAD
        dk += dko.sum()
    assert abs(float(kt.grad) - dk) < RTOL_GRAD * abs(dk)
    # (ii)
    mesh = FEMesh.line(6)
    B = 70_000
    f = torch.rand(B, 7, dtype=T64, device="cuda") + 0.5
    kap = torch.rand(B, dtype=T64, device="cuda") + 0.5
    u = DifferentiableFESolver(mesh, kap)(f)
    nodes, el, bn, bv = arrays(mesh)
    for b in (0, 32_767, 32_768, 65_535, 65_536, B - 1):
        uo = orc.solve(nodes, el, bn, bv, float(kap[b]), f[b].cpu().numpy())
        assert rel_err(u[b].cpu().numpy(), uo) < 1e-13


@pytest.mark.gpu
def test_warm_start_meets_the_same_tolerance_in_fewer_iterations():
    """`warm_start=True` (lattice path): the second solve of a slightly moved per-element kappa field starts from the
    first one's solution plus a full-multigrid pass on its residual.  Same answer as the cold solve to the parity
    tolerance, forward and gradient, and no more iterations than it."""
    N, B = 192, 64
    DEV = torch.device("cuda", 0)
    mesh = FEMesh.rectangle(N, N)
    gen = torch.Generator().manual_seed(77)
    k0 = torch.exp(0.5 * torch.randn(B, mesh.n_elements, generator=gen, dtype=T64)).to(DEV)
    k1 = k0 * (1.0 + 0.01 * torch.randn(B, mesh.n_elements, generator=gen, dtype=T64).to(DEV))
    f = torch.randn(B, mesh.n_nodes, generator=gen, dtype=T64).to(DEV)

    def run(kappa, warm):
        kk = kappa.clone().requires_grad_(True)
        s = DifferentiableFESolver(mesh, kk, device=DEV, warm_start=warm)
        u = s(f)
        (u ** 2).sum().backward()
        return u.detach(), kk.grad, s.last_info

    get_plan(mesh, DEV).warm.clear()
    u_cold, g_cold, i_cold = run(k1, False)
    run(k0, True)                                    # fills the plan's warm-start vectors
    u_warm, g_warm, i_warm = run(k1, True)
    assert i_warm.not_converged == 0
    assert float((u_warm - u_cold).abs().max() / u_cold.abs().max()) < 1e-10
    assert float((g_warm - g_cold).abs().max() / g_cold.abs().max()) < 1e-10
    assert i_warm.iterations <= i_cold.iterations and i_warm.adj_iterations <= i_cold.adj_iterations
    print(f"warm start: iterations {i_cold.iterations}+{i_cold.adj_iterations} cold -> "
          f"{i_warm.iterations}+{i_warm.adj_iterations} warm")
    # a guess from a different batch size is ignored, not misused
    u2 = DifferentiableFESolver(mesh, k1[:3], device=DEV, warm_start=True)(f[:3])
    assert float((u2 - u_cold[:3]).abs().max() / u_cold.abs().max()) < 1e-10
    get_plan(mesh, DEV).warm.clear()

"""CPU tests of the host side: mesh factories against the reference's golden data,
plan/pattern construction against the oracle, API quirks of the boundary
(SURVEY 8(b)).  No HIP kernel runs here."""
import hashlib
import os

import numpy as np
import pytest
import torch

from diffhe import FEMesh, DifferentiableFESolver, PhysicsLoss, NeuralPDE
from diffhe.plan import (build_ell_pattern, build_dia_pattern, detect_lattice, lattice_elements, chain_segments,
                         padded_batch, _bc_arrays)
from diffhe.solver import _kappa_mode, K_SCALAR, K_SAMPLE, K_ELEM, K_SAMPLE_ELEM
from oracle import p1_oracle as orc
from _util import golden, golden_json


def _arrays(mesh):
    bc_nodes = np.array(list(mesh.dirichlet_nodes.keys()), dtype=np.int64)
    bc_vals = np.array(list(mesh.dirichlet_nodes.values()), dtype=np.float64)
    return mesh.nodes.numpy(), mesh.elements.numpy(), bc_nodes, bc_vals


@pytest.mark.parametrize("name,mesh", [
    ("line_10", lambda: FEMesh.line(10)),
    ("line_7_shift", lambda: FEMesh.line(7, -1.0, 2.5, 0.25, None)),
    ("rect_4_4", lambda: FEMesh.rectangle(4, 4)),
    ("rect_3_2", lambda: FEMesh.rectangle(3, 2, (0.0, 3.0), (0.0, 1.0), 0.5)),
])
def test_mesh_factories_verbatim(name, mesh):
    g = golden("g6_mesh_" + name)
    m = mesh()
    nodes, elements, bc_nodes, bc_vals = _arrays(m)
    assert m.nodes.dtype == torch.float64 and m.elements.dtype == torch.int64
    assert np.array_equal(nodes, g["nodes"])
    assert np.array_equal(elements, g["elements"])
    assert np.array_equal(bc_nodes, g["bc_nodes"])          # dict order too
    assert np.array_equal(bc_vals, g["bc_vals"])
    assert m.free_nodes() == g["free"].tolist()
    assert repr(m) == str(g["repr"])


def test_mesh_factories_sha256_of_large_meshes():
    pins = golden_json("g6_mesh_sha256.json")
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    for key, pin in pins.items():
        kind, *dims = key.split("_")
        m = FEMesh.line(int(dims[0])) if kind == "line" else FEMesh.rectangle(int(dims[0]), int(dims[1]))
        nodes, elements, bc_nodes, _ = _arrays(m)
        assert sha(nodes) == pin["nodes"], key
        assert sha(elements) == pin["elements"], key
        assert sha(bc_nodes) == pin["bc_nodes"], key


def test_mesh_api_reference_tests():
    """reference tests/test_fem.py:43-72."""
    m = FEMesh.line(n_elements=10)
    assert (m.n_nodes, m.n_elements, m.dim) == (11, 10, 1)
    assert m.dirichlet_nodes[0] == 0.0 and m.dirichlet_nodes[10] == 0.0
    free = m.free_nodes()
    assert len(free) == 9 and 0 not in free and 10 not in free
    assert abs(m.h() - 0.1) < 1e-15
    m2 = FEMesh.rectangle(nx=4, ny=4)
    assert (m2.n_nodes, m2.n_elements, m2.dim) == (25, 32, 2)
    assert len(m2.dirichlet_nodes) == 16
    with pytest.raises(NotImplementedError):
        m2.h()


def _emulate_gather(pat, local, kappa_e, n):
    """numpy model of diffhe_ell_assemble_rows without BCs -> dense K."""
    W, cols, ptr, contrib = pat["W"], pat["cols"], pat["ent_ptr"], pat["contrib"]
    m = local.shape[1]
    K = np.zeros((n, n))
    e, pq = contrib >> 6, contrib & 63
    term = kappa_e[e] * local[pq, e]
    ent_of = np.repeat(np.arange(W * n), np.diff(ptr))
    vals = np.bincount(ent_of, weights=term, minlength=W * n)
    rows = np.tile(np.arange(n), W)
    np.add.at(K, (rows, cols.ravel()), vals)
    return K


@pytest.mark.parametrize("mesh", [
    lambda: FEMesh.rectangle(6, 6), lambda: FEMesh.rectangle(3, 2, (0.0, 3.0), (0.0, 1.0), 0.5),
    lambda: FEMesh.line(9), lambda: FEMesh.rectangle(5, 3),
])
def test_ell_pattern_reproduces_dense_assembly(mesh):
    m = mesh()
    nodes, elements, _, _ = _arrays(m)
    rng = np.random.default_rng(3)
    perm = rng.permutation(m.n_nodes)                      # also an unstructured numbering
    for el in (elements, perm[elements]):
        nd = nodes if el is elements else nodes[np.argsort(perm)]
        pat = build_ell_pattern(el, m.n_nodes)
        k0, _ = orc.element_matrices(nd, el)
        npe = el.shape[1]
        local = k0.reshape(len(el), npe * npe).T.copy()
        kap = rng.uniform(0.5, 2.0, len(el))
        K = _emulate_gather(pat, local, kap, m.n_nodes)
        Kref, _ = orc.assemble_dense(nd, el, kap, np.zeros(m.n_nodes))
        assert np.max(np.abs(K - Kref)) < 1e-12 * np.max(np.abs(Kref))
        assert np.array_equal(pat["cols"][0], np.arange(m.n_nodes))       # slot 0 = diagonal
        # slot_of points at the entry holding (row, col) of every local (p, q)
        for pq in range(npe * npe):
            r, c = el[:, pq // npe], el[:, pq % npe]
            assert np.array_equal(pat["cols"][pat["slot_of"][pq], r], c)


@pytest.mark.parametrize("nx,ny", [(2, 2), (5, 3), (8, 6)])
def test_dia_pattern_reproduces_dense_assembly(nx, ny):
    """numpy model of the symmetric-diagonal gather (store_slot semantics) vs the oracle."""
    m = FEMesh.rectangle(nx, ny, (0.0, 2.0), (0.0, 1.0))
    nodes, elements, _, _ = _arrays(m)
    rng = np.random.default_rng(4)
    nodes = nodes + rng.uniform(-0.05, 0.05, nodes.shape)          # skewed: all 4 diagonals live
    assert detect_lattice(elements, m.n_nodes) == (nx, ny)
    assert np.array_equal(lattice_elements(nx, ny), elements)
    assert detect_lattice(elements[::-1].copy(), m.n_nodes) is None
    pat = build_dia_pattern(nx, ny)
    n, W = m.n_nodes, nx + 1
    k0, _ = orc.element_matrices(nodes, elements)
    local = k0.reshape(len(elements), 9).T.copy()
    kap = rng.uniform(0.5, 2.0, len(elements))
    e, pq = pat["contrib"] >> 6, pat["contrib"] & 63
    ent_of = np.repeat(np.arange(7 * n), np.diff(pat["ent_ptr"]))
    vals = np.bincount(ent_of, weights=kap[e] * local[pq, e], minlength=7 * n).reshape(7, n)
    Kref, _ = orc.assemble_dense(nodes, elements, kap, np.zeros(n))
    K = np.zeros((n, n))
    for k, off in enumerate((0, 1, W, nx)):                         # stored upper diagonals
        idx = np.arange(n - off)
        K[idx, idx + off] += vals[k, : n - off]
        if off:
            K[idx + off, idx] += vals[k, : n - off]
    assert np.max(np.abs(K - Kref)) < 1e-12 * np.max(np.abs(Kref))
    for k, off in enumerate((0, 1, W, nx, -1, -W, -nx)):            # every entry = K[i, i+off]
        i = np.arange(max(0, -off), min(n, n - off))
        assert np.max(np.abs(vals[k, i] - Kref[i, i + off])) < 1e-12 * np.max(np.abs(Kref))
        used = np.diff(pat["ent_ptr"]).reshape(7, n)[k] > 0
        assert np.array_equal(pat["cols"][k][used], np.nonzero(used)[0] + off)
        assert np.array_equal(pat["cols"][k][~used], np.nonzero(~used)[0])


def test_chain_segments():
    bc = np.zeros(8, dtype=np.uint8)
    assert chain_segments(8, bc).tolist() == [[0, 7, 0]]
    bc[[0, 7]] = 1
    assert chain_segments(8, bc).tolist() == [[0, 7, 3]]
    bc[:] = 0
    bc[[2, 5]] = 1
    assert chain_segments(8, bc).tolist() == [[0, 2, 2], [2, 5, 3], [5, 7, 1]]
    bc[:] = 0
    bc[0] = 1
    assert chain_segments(8, bc).tolist() == [[0, 7, 1]]


def test_padded_batch():
    assert [padded_batch(b) for b in (1, 2, 3, 5, 33, 64, 65, 128, 200, 256, 1000)] == \
        [1, 2, 4, 8, 64, 64, 128, 128, 256, 256, 1024]


def test_kappa_modes():
    m = 32
    assert _kappa_mode(torch.tensor(1.0), m, None)[0] == K_SCALAR
    assert _kappa_mode(torch.ones(1), m, 7)[0] == K_SCALAR
    assert _kappa_mode(torch.ones(m), m, None) == (K_ELEM, None)
    assert _kappa_mode(torch.ones(m), m, 5) == (K_ELEM, None)
    assert _kappa_mode(torch.ones(5), m, 5) == (K_SAMPLE, 5)
    assert _kappa_mode(torch.ones(5, 1), m, None) == (K_SAMPLE, 5)
    assert _kappa_mode(torch.ones(5, m), m, 5) == (K_SAMPLE_ELEM, 5)
    with pytest.raises(ValueError):
        _kappa_mode(torch.ones(5, 3), m, 5)


def test_solver_boundary_quirks():
    """kappa wrapping, Parameter registration, errors."""
    mesh = FEMesh.line(5)
    s = DifferentiableFESolver(mesh)
    assert s.kappa.dtype == torch.float64 and s.kappa.dim() == 0 and float(s.kappa) == 1.0
    assert list(s.parameters()) == []
    p64 = torch.nn.Parameter(torch.tensor(2.0, dtype=torch.float64))
    s2 = DifferentiableFESolver(mesh, p64)
    assert [n for n, _ in s2.named_parameters()] == ["_kappa"] and s2.kappa is p64
    p32 = torch.nn.Parameter(torch.tensor(2.0))
    s3 = DifferentiableFESolver(mesh, p32)
    assert list(s3.parameters()) == [] and s3.kappa.dtype == torch.float64 and s3.kappa.requires_grad
    bad = FEMesh(nodes=torch.zeros(4, 3, dtype=torch.float64), elements=torch.zeros(1, 4, dtype=torch.long))
    with pytest.raises(NotImplementedError, match="Only 1D and 2D supported"):
        DifferentiableFESolver(bad)(torch.zeros(4))
    for opt in (dict(warm_start="adjoint"), dict(chain="fast"), dict(method="lu"), dict(assembly="scatter")):
        with pytest.raises(ValueError, match="Unknown"):
            DifferentiableFESolver(mesh, **opt)
    assert DifferentiableFESolver(mesh, warm_start="forward").warm_start == "forward"


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback():
    s = DifferentiableFESolver(FEMesh.line(5))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s(torch.ones(6, dtype=torch.float64))


def test_physics_loss_modes():
    mesh = FEMesh.line(10)
    with pytest.raises(ValueError, match="Unknown mode: 'bogus'"):
        PhysicsLoss(mesh, lambda x: x, mode="bogus")
    g = golden("g8_physics_loss")
    loss = PhysicsLoss(mesh, lambda x: torch.ones_like(x), mode="variational")
    assert abs(float(loss(torch.from_numpy(g["u_pred"]))) - float(g["variational"])) < 1e-13


def test_physics_loss_ensemble_host_logic():
    """RHS ensembles: stacking, one (stubbed) batched solve, caching, per-member values, argument errors.
    The solve is stubbed (no GPU here); the HIP-backed value test is tests/test_gpu_parity.py (fixture G12)."""
    mesh = FEMesh.line(10)

    class Stub(torch.nn.Module):
        calls = 0
        kappa = torch.tensor(1.0, dtype=torch.float64)

        def forward(self, f):
            Stub.calls += 1
            assert f.shape == (3, 11)                      # ONE call with the whole ensemble
            return 2.0 * f

    fns = [lambda x: torch.ones_like(x), lambda x: x, lambda x: x ** 2]
    loss = PhysicsLoss(mesh, forcing_fns=fns, solver=Stub())
    u_pred = torch.linspace(0, 1, 11, dtype=torch.float64)
    per = loss.member_losses(u_pred)
    x = mesh.nodes.squeeze(1)
    expect = torch.stack([((u_pred - 2 * fn(x)) ** 2).mean() for fn in fns])
    assert torch.allclose(per, expect, rtol=0, atol=1e-15)
    assert abs(float(loss(u_pred)) - float(expect.mean())) < 1e-15        # mean of the member losses
    assert Stub.calls == 1                                                 # cached across both calls
    same = PhysicsLoss(mesh, lambda x: torch.stack([fn(x) for fn in fns]), solver=Stub())   # forcing_fn -> (B, n)
    assert abs(float(same(u_pred)) - float(expect.mean())) < 1e-15
    with pytest.raises(ValueError, match="exactly one"):
        PhysicsLoss(mesh)
    with pytest.raises(ValueError, match="exactly one"):
        PhysicsLoss(mesh, lambda x: x, forcing_fns=fns)
    with pytest.raises(ValueError, match="fem_match"):
        PhysicsLoss(mesh, forcing_fns=fns, mode="variational")


def test_neural_pde_mask_and_shapes():
    """reference tests/test_neural.py:21-37 (BC mask) without any solve."""
    mesh = FEMesh.line(10)
    model = NeuralPDE(mesh, hidden_dim=8, n_layers=2)
    u = model()
    assert u.shape == (11,) and u.dtype == torch.float64
    assert abs(float(u[0])) < 1e-12 and abs(float(u[-1])) < 1e-12
    m2 = NeuralPDE(FEMesh.rectangle(3, 3), hidden_dim=4, n_layers=1)
    assert float(m2._mask.sum()) == 4.0
    losses = model.train_pde(lambda x: torch.ones_like(x), n_epochs=5, mode="variational", verbose=False)
    assert len(losses) == 5


def test_bc_arrays():
    mesh = FEMesh.line(4, bc_left=1.5, bc_right=None)
    is_bc, g = _bc_arrays(mesh)
    assert is_bc.tolist() == [1, 0, 0, 0, 0] and g.tolist() == [1.5, 0, 0, 0, 0]


def test_custom_ops_are_registered_with_shape_inference():
    """The solve is exposed as torch.library ops (diffhe::fe_solve / diffhe::fe_solve_backward) with
    fake (meta) implementations, so tracing needs no GPU."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from diffhe import solver as S
    assert hasattr(torch.ops.diffhe, "fe_solve") and hasattr(torch.ops.diffhe, "fe_solve_backward")
    mesh = FEMesh.rectangle(4, 4)
    s = DifferentiableFESolver(mesh)
    S._SOLVERS[id(s)] = s
    with FakeTensorMode():
        k1, f1 = torch.empty((), dtype=torch.float64), torch.empty(25, dtype=torch.float64)
        u, tok = torch.ops.diffhe.fe_solve(k1, f1, f1.new_empty(0), id(s), False)
        assert u.shape == (25,) and u.dtype == torch.float64 and tok.shape == ()
        kb, fb = torch.empty(6, dtype=torch.float64), torch.empty(6, 25, dtype=torch.float64)
        u, _ = torch.ops.diffhe.fe_solve(kb, fb, fb.new_empty(0), id(s), False)
        assert u.shape == (6, 25)
        u, _ = torch.ops.diffhe.fe_solve(kb, f1, f1.new_empty(0), id(s), False)          # kappa batch, shared f
        assert u.shape == (6, 25)
        gk, gf, gl = torch.ops.diffhe.fe_solve_backward(u, tok, True, False, False, kb, f1, f1.new_empty(0))
        assert gk.shape == (6,) and gf.numel() == 0 and gl.numel() == 0
        gk, gf, gl = torch.ops.diffhe.fe_solve_backward(u, tok, False, True, True, kb, f1, fb)   # load (B, n)
        assert gk.numel() == 0 and gf.shape == (25,) and gl.shape == (6, 25)


def test_coarsening_step_rules():
    from diffhe.plan import coarsening_step
    grid = lambda nx, ny, lx, ly: FEMesh.rectangle(nx, ny, (0.0, lx), (0.0, ly)).nodes.numpy().reshape(ny + 1, nx + 1, 2)  # noqa: E731
    assert coarsening_step(grid(64, 64, 1.0, 1.0)) == (2, 2)
    assert coarsening_step(grid(64, 64, 8.0, 1.0)) == (2, 1)      # short vertical edges: halve the rows only
    assert coarsening_step(grid(64, 64, 1.0, 8.0)) == (1, 2)
    assert coarsening_step(grid(512, 64, 1.0, 1.0)) == (1, 2)     # short horizontal edges: halve the columns
    assert coarsening_step(grid(6, 6, 1.0, 1.0)) == (2, 2)
    assert coarsening_step(grid(3, 3, 1.0, 1.0)) is None          # odd: stop
    assert coarsening_step(grid(64, 5, 8.0, 1.0)) is None         # would need the rows, which do not halve


def test_aggregation_hierarchy_galerkin_lists():
    """diffhe/amg.py: aggregates cover every free node once, roots are >= 3 apart, and the gather lists
    reproduce P^T A P for piecewise-constant P (numpy model of diffhe_ell_galerkin)."""
    from diffhe.amg import build_hierarchy
    m = FEMesh.rectangle(14, 11, (0.0, 2.0), (0.0, 1.0))
    nodes, elements, bc_nodes, _ = _arrays(m)
    n = m.n_nodes
    rng = np.random.default_rng(0)
    perm = rng.permutation(n)
    el = perm[elements]
    nd = nodes[np.argsort(perm)]
    is_bc = np.zeros(n, dtype=np.uint8)
    is_bc[perm[bc_nodes]] = 1
    pat = build_ell_pattern(el, n)
    K, _ = orc.assemble_dense(nd, el, rng.uniform(0.5, 2.0, len(el)), np.zeros(n))
    free = is_bc == 0
    K[~free, :] = 0.0
    K[:, ~free] = 0.0
    K[~free, ~free] = 1.0
    W, cols = pat["W"], pat["cols"]
    vals = np.stack([K[np.arange(n), cols[k]] * ((k == 0) | (cols[k] != np.arange(n))) for k in range(W)])
    levels = build_hierarchy(cols, is_bc, min_coarse=6)
    assert len(levels) >= 2
    A, v = K, vals
    for lv in levels:
        agg, nc = lv["agg"], lv["n"]
        act = agg >= 0
        assert np.array_equal(np.unique(agg[act]), np.arange(nc))
        P = np.zeros((A.shape[0], nc))
        P[np.nonzero(act)[0], agg[act]] = 1.0
        Ac = P.T @ A @ P
        ptr, contrib = lv["ent_ptr"], lv["contrib"]
        ent_of = np.repeat(np.arange(lv["W"] * nc), np.diff(ptr))
        vc = np.bincount(ent_of, weights=v.ravel()[contrib], minlength=lv["W"] * nc).reshape(lv["W"], nc)
        Ac2 = np.zeros((nc, nc))
        used = np.diff(ptr).reshape(lv["W"], nc) > 0
        for k in range(lv["W"]):
            Ac2[np.arange(nc)[used[k]], lv["cols"][k][used[k]]] += vc[k][used[k]]
        assert np.max(np.abs(Ac - Ac2)) < 1e-12 * np.max(np.abs(Ac))
        assert np.array_equal(lv["cols"][0], np.arange(nc))
        members = lv["agg_members"]
        assert np.array_equal(np.sort(members), np.nonzero(act)[0])
        A, v = Ac, vc


def test_lattice_gather_lists_are_the_sorted_definition():
    """The analytic gather lists of a lattice (plan.build_dia_pattern) equal the stable sort over all local entries."""
    from diffhe.plan import _build_dia_pattern_sorted
    for nx, ny in ((2, 2), (3, 2), (2, 5), (7, 5), (16, 9)):
        a, b = build_dia_pattern(nx, ny), _build_dia_pattern_sorted(nx, ny)
        for key in ("cols", "ent_ptr", "contrib"):
            assert np.array_equal(a[key], b[key]) and a[key].dtype == b[key].dtype, (nx, ny, key)


def test_plan_host_arrays_are_built_once_per_node(tmp_path, monkeypatch):
    """DIFFHE_PLAN_CACHE: the first rank builds and publishes the host arrays of a lattice level, the others map them."""
    from diffhe import plan as P
    calls = []

    def build():
        calls.append(1)
        return P.build_dia_pattern(6, 5)

    assert P.host_arrays("x", build)["contrib"].dtype == np.int32 and len(calls) == 1      # no cache dir: plain build
    monkeypatch.setenv("DIFFHE_PLAN_CACHE", str(tmp_path))
    first = P.host_arrays("dia_6x5", build)
    again = P.host_arrays("dia_6x5", build)
    assert len(calls) == 2                                                                 # the second call mapped the files
    for key in ("cols", "ent_ptr", "contrib"):
        assert np.array_equal(first[key], again[key]) and again[key].dtype == first[key].dtype
    assert int(again["We"]) == 7
    assert not [d for d in os.listdir(tmp_path) if ".tmp" in d]


def test_gather_codes_refuse_meshes_beyond_int32():
    """e * 64 + pq must fit an int32: 2^25 elements at most -- lattices beyond 4096^2 raise instead of wrapping."""
    from diffhe import plan as P
    with pytest.raises(ValueError, match="int32 gather lists"):
        P.build_dia_pattern(4096, 4097)
    with pytest.raises(ValueError, match="int32 gather lists"):
        P._build_dia_pattern_sorted(8192, 2049)
    P._check_gather_code_range(2 * 4096 * 4095)


def test_smoothed_aggregation_hierarchy_reproduces_the_galerkin_products():
    """diffhe.amg.build_hierarchy_sa: the weighted gather lists give P^T A_b P of a PER-SAMPLE matrix (same pattern, other
    values) to rounding, P^T as CSR is the transpose of P as ELL rows, Dirichlet rows interpolate nothing."""
    import scipy.sparse as sp
    from diffhe import amg
    nodes, el, bn, bv = orc.mesh_rectangle(40, 36)
    rng = np.random.default_rng(0)
    n = len(nodes)
    isbc = np.zeros(n, dtype=bool)
    isbc[bn] = True
    nodes = nodes + np.where(isbc[:, None], 0.0, rng.uniform(-0.2, 0.2, nodes.shape) / 40)
    pat = build_ell_pattern(el, n)
    cols = pat["cols"]

    def eliminated(kappa):
        K, _ = orc.assemble_sparse(nodes, el, kappa, np.ones(n))
        D = sp.diags((~isbc).astype(float))
        return sp.csr_matrix(D @ K @ D + sp.diags(isbc.astype(float)))

    def ell_values(A, c):
        W, nn = c.shape
        v = np.zeros((W, nn))
        for k in range(W):
            real = (k == 0) | (c[k] != np.arange(nn))
            v[k, real] = np.asarray(A[np.arange(nn)[real], c[k][real]]).ravel()
        return v

    levels = amg.build_hierarchy_sa(cols, ell_values(eliminated(1.0), cols), isbc)
    assert len(levels) >= 2 and levels[0]["n"] < 0.2 * n
    Ab = eliminated(np.exp(0.5 * rng.standard_normal(len(el))))
    vals_f, A = ell_values(Ab, cols).ravel(), Ab
    for lev in levels:
        nc, Wc = lev["n"], lev["W"]
        pw, nf = lev["p_cols"].shape
        ok = lev["p_cols"].ravel() >= 0
        P = sp.csr_matrix((lev["p_vals"].ravel()[ok], (np.tile(np.arange(nf), pw)[ok], lev["p_cols"].ravel()[ok])), shape=(nf, nc))
        if lev is levels[0]:
            assert abs(P[np.nonzero(isbc)[0]]).sum() == 0.0
        vc = np.zeros(Wc * nc)
        np.add.at(vc, np.repeat(np.arange(Wc * nc), np.diff(lev["ent_ptr"])), lev["weights"] * vals_f[lev["contrib"]])
        Ac = sp.csr_matrix(P.T @ A @ P)
        ref = ell_values(Ac, lev["cols"]).ravel()
        assert np.max(np.abs(vc - ref)) <= 1e-13 * np.max(np.abs(ref))
        PT = sp.csr_matrix((lev["agg_weights"], lev["agg_members"], lev["agg_ptr"]), shape=(nc, nf))
        assert abs(PT - P.T).max() == 0.0
        assert np.array_equal(lev["cols"][0], np.arange(nc))           # slot 0 = diagonal
        vals_f, A = vc, Ac

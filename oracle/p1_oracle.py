"""CPU oracle for the differentiable P1-FEM solve path.  TEST INFRASTRUCTURE ONLY.

This module is a numpy/scipy *restatement* of the algorithm of the reference
(`/root/reference/diffhe/solver.py`, `mesh.py`) -- it is NOT part of the shipped
product.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / reported baseline.  The
product path (`difffe-physics-lab_amd/diffhe`) never imports anything from here
and fails loudly when its HIP extension is missing.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here
against the golden vectors in `tests/golden/*.npz`, which were produced by
importing the reference itself (`tests/golden/make_golden.py`), and against the
known-answer tests of the reference's own suite (`tests/test_fem.py:85-179`).

Third-party arithmetic on the reference path: `torch.linalg.solve`
(`solver.py:174`, LAPACK LU with partial pivoting, pin `torch>=2.0`); restated
here with `numpy.linalg.solve` (same LAPACK `gesv`) on the dense path and with
`scipy.sparse.linalg.splu` on the sparse path used beyond the dense ceiling.

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import numpy as np

try:  # scipy is only needed for the large-size (sparse) oracle
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
except Exception:  # pragma: no cover
    sp = None
    spla = None


# ---------------------------------------------------------------------------
# Mesh factories (reference: diffhe/mesh.py)
# ---------------------------------------------------------------------------

def mesh_line(n_elements=10, x_left=0.0, x_right=1.0, bc_left=0.0, bc_right=0.0):
    """`FEMesh.line` (mesh.py:58-77): linspace nodes, elements [e, e+1],
    BC dict {0: bc_left, N: bc_right} with `None` entries omitted."""
    import torch  # torch.linspace is what the reference uses (mesh.py:68); its
    # rounding differs from np.linspace in the last bit for some points.
    x = torch.linspace(x_left, x_right, n_elements + 1, dtype=torch.float64).numpy().copy()
    nodes = x[:, None]
    idx = np.arange(n_elements, dtype=np.int64)
    elements = np.stack([idx, idx + 1], axis=1)
    bc_nodes, bc_vals = [], []
    if bc_left is not None:
        bc_nodes.append(0)
        bc_vals.append(float(bc_left))
    if bc_right is not None:
        bc_nodes.append(n_elements)
        bc_vals.append(float(bc_right))
    return nodes, elements, np.asarray(bc_nodes, dtype=np.int64), np.asarray(bc_vals, dtype=np.float64)


def mesh_rectangle(nx=4, ny=4, x_range=(0.0, 1.0), y_range=(0.0, 1.0), bc_value=0.0):
    """`FEMesh.rectangle` (mesh.py:79-121): node id = i*(nx+1)+j (i row/y, j col/x,
    mesh.py:92-98); quad (a,b,c,d) -> triangles [a,b,d], [b,c,d] (mesh.py:100-105);
    every node with x or y `np.isclose` to a range end is Dirichlet (mesh.py:110-120)."""
    xs = np.linspace(x_range[0], x_range[1], nx + 1)
    ys = np.linspace(y_range[0], y_range[1], ny + 1)
    xx, yy = np.meshgrid(xs, ys)
    nodes = np.stack([xx.ravel(), yy.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    a = (i * (nx + 1) + j).ravel()
    b = a + 1
    c = a + (nx + 1) + 1
    d = a + (nx + 1)
    elements = np.empty((2 * nx * ny, 3), dtype=np.int64)
    elements[0::2] = np.stack([a, b, d], axis=1)
    elements[1::2] = np.stack([b, c, d], axis=1)
    x, y = nodes[:, 0], nodes[:, 1]
    on_bnd = (np.isclose(x, x_range[0]) | np.isclose(x, x_range[1])
              | np.isclose(y, y_range[0]) | np.isclose(y, y_range[1]))
    bc_nodes = np.nonzero(on_bnd)[0].astype(np.int64)
    bc_vals = np.full(bc_nodes.shape, float(bc_value), dtype=np.float64)
    return nodes, elements, bc_nodes, bc_vals


def free_nodes(n_nodes, bc_nodes):
    """`FEMesh.free_nodes` (mesh.py:127-129): ascending ids not in the BC dict."""
    mask = np.ones(n_nodes, dtype=bool)
    mask[np.asarray(bc_nodes, dtype=np.int64)] = False
    return np.nonzero(mask)[0]


# ---------------------------------------------------------------------------
# Element integrals (reference: solver.py:82-96 and solver.py:112-145)
# ---------------------------------------------------------------------------

def element_matrices(nodes, elements):
    """Per-element unit-kappa stiffness `k0[e]` (npe x npe) and lumped load weight.

    1D (solver.py:84-96): h = x_j - x_i; k0 = (1/h) [[1,-1],[-1,1]]; load: F_p += h/2 f_p.
    2D (solver.py:119-145): area = 0.5 |det|, skipped when < 1e-15;
        b = [yj-yk, yk-yi, yi-yj], c = [xk-xj, xi-xk, xj-xi];
        k0[p,q] = (b_p b_q + c_p c_q) / (4 area);  load: F_p += area/3 * (f_i+f_j+f_k)/3.
    Returns (k0 (m,npe,npe), w (m,)) with w = h (1D) or area (2D; 0 for skipped elements).
    """
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    dim = nodes.shape[1]
    if dim == 1:
        xi = nodes[elements[:, 0], 0]
        xj = nodes[elements[:, 1], 0]
        h = xj - xi
        k0 = np.empty((len(elements), 2, 2))
        k0[:, 0, 0] = 1.0 / h
        k0[:, 0, 1] = -1.0 / h
        k0[:, 1, 0] = -1.0 / h
        k0[:, 1, 1] = 1.0 / h
        return k0, h
    if dim == 2:
        xi, yi = nodes[elements[:, 0], 0], nodes[elements[:, 0], 1]
        xj, yj = nodes[elements[:, 1], 0], nodes[elements[:, 1], 1]
        xk, yk = nodes[elements[:, 2], 0], nodes[elements[:, 2], 1]
        area = 0.5 * np.abs((xj - xi) * (yk - yi) - (xk - xi) * (yj - yi))
        keep = area >= 1e-15
        b = np.stack([yj - yk, yk - yi, yi - yj], axis=1)
        c = np.stack([xk - xj, xi - xk, xj - xi], axis=1)
        safe = np.where(keep, area, 1.0)
        k0 = (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (4.0 * safe)[:, None, None]
        k0[~keep] = 0.0
        return k0, np.where(keep, area, 0.0)
    raise NotImplementedError("Only 1D and 2D supported")  # solver.py:67


def _kappa_per_element(kappa, m):
    kappa = np.asarray(kappa, dtype=np.float64)
    if kappa.size == 1:
        return np.full(m, float(kappa.reshape(())), dtype=np.float64)
    if kappa.shape != (m,):
        raise ValueError(f"kappa must be scalar or per-element ({m},), got {kappa.shape}")
    return kappa


def load_vector(nodes, elements, f, w=None):
    """F before BCs.  1D: F_p += h/2 f_p (solver.py:95-96).
    2D: F_p += area/3 * (f_i+f_j+f_k)/3 for each vertex p (solver.py:143-145)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    f = np.asarray(f, dtype=np.float64).reshape(-1)
    n = nodes.shape[0]
    if w is None:
        _, w = element_matrices(nodes, elements)
    F = np.zeros(n)
    # np.add.at accumulates in index order: element-major with the vertices innermost is the order of the reference's
    # loops, so F is bit-identical to the reference's (checked against the G5 fixtures)
    if nodes.shape[1] == 1:
        vals = (w / 2.0)[:, None] * f[elements]                       # h_e / 2.0 * f[i], solver.py:95-96
        np.add.at(F, elements.ravel(), vals.ravel())
    else:
        fc = (f[elements[:, 0]] + f[elements[:, 1]] + f[elements[:, 2]]) / 3.0   # solver.py:143
        vals = np.repeat((w / 3.0 * fc)[:, None], 3, axis=1)          # area / 3.0 * f_centroid, solver.py:145
        np.add.at(F, elements.ravel(), vals.ravel())
    return F


def local_stiffness(nodes, elements, kappa):
    """Per-element stiffness contributions with the reference's operation order (each op rounded on its own):
    1D  k_e = kappa / h_e, [[k,-k],[-k,k]] (solver.py:88-92); 2D  k_pq = kappa * (b_p b_q + c_p c_q) / (4.0 area)
    = (kappa * t) / (4.0 area) (solver.py:139).  Returns (m, npe, npe); zero for skipped (degenerate) elements."""
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    m = elements.shape[0]
    kap = _kappa_per_element(kappa, m)
    if nodes.shape[1] == 1:
        h = nodes[elements[:, 1], 0] - nodes[elements[:, 0], 0]
        k = kap / h
        ke = np.empty((m, 2, 2))
        ke[:, 0, 0] = k; ke[:, 0, 1] = -k; ke[:, 1, 0] = -k; ke[:, 1, 1] = k
        return ke
    xi, yi = nodes[elements[:, 0], 0], nodes[elements[:, 0], 1]
    xj, yj = nodes[elements[:, 1], 0], nodes[elements[:, 1], 1]
    xk, yk = nodes[elements[:, 2], 0], nodes[elements[:, 2], 1]
    area = 0.5 * np.abs((xj - xi) * (yk - yi) - (xk - xi) * (yj - yi))
    keep = area >= 1e-15
    b = np.stack([yj - yk, yk - yi, yi - yj], axis=1)
    c = np.stack([xk - xj, xi - xk, xj - xi], axis=1)
    t = b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]
    ke = (kap[:, None, None] * t) / (4.0 * np.where(keep, area, 1.0))[:, None, None]
    ke[~keep] = 0.0
    return ke


def assemble_dense(nodes, elements, kappa, f):
    """Dense (n,n) K and (n,) F before BCs: the scatter-add loops of
    solver.py:82-96 (1D) / solver.py:112-145 (2D), same operation and accumulation order
    (element-major, p outer, q inner): bit-identical to the reference's K and F (G5 fixtures)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    n = nodes.shape[0]
    ke = local_stiffness(nodes, elements, kappa)
    npe = elements.shape[1]
    rows = np.repeat(elements[:, :, None], npe, axis=2).ravel()
    cols = np.repeat(elements[:, None, :], npe, axis=1).ravel()
    K = np.zeros((n, n))
    np.add.at(K, (rows, cols), ke.ravel())
    return K, load_vector(nodes, elements, f)


def assemble_sparse(nodes, elements, kappa, f):
    """Same system as `assemble_dense`, stored CSR; every entry accumulates its contributions in the same
    (element) order, so the values are bit-identical to the dense ones."""
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    n = nodes.shape[0]
    ke = local_stiffness(nodes, elements, kappa)
    npe = elements.shape[1]
    rows = np.repeat(elements[:, :, None], npe, axis=2).ravel()
    cols = np.repeat(elements[:, None, :], npe, axis=1).ravel()
    uniq, inv = np.unique(rows * n + cols, return_inverse=True)
    vals = np.zeros(len(uniq))
    np.add.at(vals, inv, ke.ravel())
    K = sp.csr_matrix((vals, (uniq // n, uniq % n)), shape=(n, n))
    return K, load_vector(nodes, elements, f)


# ---------------------------------------------------------------------------
# Dirichlet elimination + solve (reference: solver.py:153-183)
# ---------------------------------------------------------------------------

def refine_solution(Kff, solve_fn, x, rhs, steps):
    """Iterative refinement of an fp64 LU solve with residuals evaluated in extended precision (numpy.longdouble,
    x87 80-bit here): x <- x + LU^{-1}(rhs - Kff x).  Converges to the EXACT solution of the fp64-stored system --
    the matrix the reference assembled (solver.py:89-92, :137-140) -- to ~1e-15 relative, where the plain LU result
    (what solver.py:174 returns) carries a forward error of ~cond * eps: 7e-12 at 1D 10^4 elements, 3e-11 (u) /
    9e-11 (dL/dkappa) at 2D 1024^2.  A yardstick for the large-size parity checks, NOT what the reference computes:
    `tests/test_oracle_golden.py` pins its distance to the reference's own LU results (G10, G11)."""
    if steps <= 0:
        return x
    A = Kff.astype(np.longdouble)
    b = np.asarray(rhs, dtype=np.longdouble)
    x = np.asarray(x, dtype=np.longdouble)
    for _ in range(steps):
        r = b - A @ x
        x = x + solve_fn(np.asarray(r, dtype=np.float64))
    return np.asarray(x, dtype=np.float64)


def apply_bc_and_solve(K, F, bc_nodes, bc_vals, refine=0):
    """F_free = F[free] - sum_bc K[free,bc] g (solver.py:165-169);
    K_free = K[free][:,free] (solver.py:171); u_free = solve(K_free, F_free)
    (solver.py:174); u[bc] = g, u[free] = u_free (solver.py:177-181).
    refine > 0: that many steps of `refine_solution` on top (ours, yardstick only)."""
    n = F.shape[0]
    bc_nodes = np.asarray(bc_nodes, dtype=np.int64)
    bc_vals = np.asarray(bc_vals, dtype=np.float64)
    free = free_nodes(n, bc_nodes)
    u = np.zeros(n)
    u[bc_nodes] = bc_vals
    # F_free[fi] = F_free[fi] - K[f_node, bc] * g, one Dirichlet node after the other in dict order (solver.py:166-169):
    # the subtractions are sequential, not F - (sum of products)
    F_free = F[free].copy()
    if sp is not None and sp.issparse(K):
        K = K.tocsr()
        Kfb = K[free][:, bc_nodes].tocsc()
        for j in range(len(bc_nodes)):
            lo, hi = Kfb.indptr[j], Kfb.indptr[j + 1]
            if hi > lo:
                F_free[Kfb.indices[lo:hi]] -= Kfb.data[lo:hi] * bc_vals[j]
        Kff = K[free][:, free]
        lu = spla.splu(Kff.tocsc())
        u[free] = refine_solution(Kff, lu.solve, lu.solve(F_free), F_free, refine)
        return u, lu
    for j, b in enumerate(bc_nodes):
        F_free = F_free - K[free, b] * bc_vals[j]
    Kff = K[np.ix_(free, free)]
    u[free] = refine_solution(Kff, lambda r: np.linalg.solve(Kff, r), np.linalg.solve(Kff, F_free), F_free, refine)
    return u, None


def solve(nodes, elements, bc_nodes, bc_vals, kappa, f, sparse=None, refine=0):
    """`DifferentiableFESolver.forward` (solver.py:49-67) for one sample."""
    n = np.asarray(nodes).shape[0]
    if sparse is None:
        sparse = n > 1500
    asm = assemble_sparse if sparse else assemble_dense
    K, F = asm(nodes, elements, kappa, f)
    u, _ = apply_bc_and_solve(K, F, bc_nodes, bc_vals, refine)
    return u


# ---------------------------------------------------------------------------
# Adjoint (what autograd computes through solver.py:89-96,139-145,169-181)
# ---------------------------------------------------------------------------

def solve_with_adjoint(nodes, elements, bc_nodes, bc_vals, kappa, f, gbar_fn, sparse=None, with_cond=False, refine=0):
    """Forward solve, then the adjoint of it for the cotangent `gbar = gbar_fn(u)`.

    lambda_free = K_free^{-T} gbar_free, lambda_bc = 0   (LinalgSolveExBackward of solver.py:174)
    dL/dkappa_e = - sum_{p,q in e} lambda_p k0_e[p,q] u_q  (reverse of solver.py:89-92 / 137-140,
                  with the full u so the lifting term of solver.py:169 is included)
    dL/df       = M^T lambda, M the load map of solver.py:95-96 (1D) / 143-145 (2D).
    Returns (u, dL/dkappa per element (m,), dL/df (n,)); sum the per-element
    vector for a scalar kappa.  with_cond=True appends sum_{p,q} |lambda_p| |k0_e[p,q]| |u_q| per element: the
    magnitude of the terms each dL/dkappa_e is made of (when u is nearly constant over an element they cancel, and
    the gradient is only defined to u * that magnitude in fp64 -- used by tools/stress.py to judge gradients).
    refine=k: k steps of `refine_solution` after the forward AND the adjoint LU solve (extended-precision residuals on
    the reference-order assembled matrix): the yardstick with margin for the 512^2 / 1024^2 parity checks, where the
    plain LU's own forward error (cond * eps) is most of the 1e-10 tolerance.
    """
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    n, m = nodes.shape[0], elements.shape[0]
    if sparse is None:
        sparse = n > 1500
    asm = assemble_sparse if sparse else assemble_dense
    K, F = asm(nodes, elements, kappa, f)
    u, lu = apply_bc_and_solve(K, F, bc_nodes, bc_vals, refine)
    gbar = np.asarray(gbar_fn(u), dtype=np.float64)
    free = free_nodes(n, bc_nodes)
    lam = np.zeros(n)
    if lu is not None:
        solve_t = lambda r: lu.solve(r, trans="T")                      # noqa: E731
        KffT = K.tocsr()[free][:, free].T.tocsr() if refine > 0 else None
    else:
        KffT = K[np.ix_(free, free)].T
        solve_t = lambda r: np.linalg.solve(KffT, r)                    # noqa: E731
    lam[free] = refine_solution(KffT, solve_t, solve_t(gbar[free]), gbar[free], refine)
    k0, w = element_matrices(nodes, elements)
    lam_e = lam[elements]            # (m, npe)
    u_e = u[elements]
    dkappa = -np.einsum("ep,epq,eq->e", lam_e, k0, u_e)
    df = np.zeros(n)
    if nodes.shape[1] == 1:
        np.add.at(df, elements[:, 0], w / 2.0 * lam_e[:, 0])
        np.add.at(df, elements[:, 1], w / 2.0 * lam_e[:, 1])
    else:
        s = (lam_e[:, 0] + lam_e[:, 1] + lam_e[:, 2]) * (w / 9.0)
        for p in range(3):
            np.add.at(df, elements[:, p], s)
    if with_cond:
        return u, dkappa, df, np.einsum("ep,epq,eq->e", np.abs(lam_e), np.abs(k0), np.abs(u_e))
    return u, dkappa, df


def solve_batch(nodes, elements, bc_nodes, bc_vals, kappa_b, f_b, sparse=None):
    """Loop of independent reference solves over a batch (the reference has no
    batch dimension, solver.py:54).  `kappa_b` is (B,) or
    (B,m) or a scalar shared by all; `f_b` is (B,n) or (n,) shared by all."""
    f_b = np.asarray(f_b, dtype=np.float64)
    kappa_b = np.asarray(kappa_b, dtype=np.float64)
    B = f_b.shape[0] if f_b.ndim == 2 else (kappa_b.shape[0] if kappa_b.ndim >= 1 and kappa_b.size > 1 else 1)
    out = []
    for s in range(B):
        f = f_b[s] if f_b.ndim == 2 else f_b
        k = kappa_b[s] if (kappa_b.ndim >= 1 and kappa_b.shape[0] == B and kappa_b.size > 1) else kappa_b
        out.append(solve(nodes, elements, bc_nodes, bc_vals, k, f, sparse=sparse))
    return np.stack(out)


# ---------------------------------------------------------------------------
# PhysicsLoss (reference: diffhe/loss.py) -- 1D only, as in the reference
# ---------------------------------------------------------------------------

def physics_loss_fem_match(nodes, elements, bc_nodes, bc_vals, kappa, f, u_pred):
    """MSE(u_pred, solver(f)) (loss.py:78-83)."""
    u = solve(nodes, elements, bc_nodes, bc_vals, kappa, f)
    return float(np.mean((np.asarray(u_pred, dtype=np.float64) - u) ** 2))


def physics_loss_variational(nodes, bc_nodes, f, u_pred):
    """FD-Laplacian residual on the free nodes (loss.py:85-105)."""
    x = np.asarray(nodes)[:, 0]
    free = free_nodes(len(x), bc_nodes)
    x_free = x[free]
    u_free = np.asarray(u_pred, dtype=np.float64)[free]
    h = float(x_free[1] - x_free[0]) if len(x_free) > 1 else 1.0
    if len(x_free) >= 3:
        lap = (u_free[:-2] - 2 * u_free[1:-1] + u_free[2:]) / h ** 2
        res = lap + np.asarray(f, dtype=np.float64)[free][1:-1]
    else:
        res = np.zeros(1)
    return float(np.mean(res ** 2))


# ---------------------------------------------------------------------------
# Extended-precision reference for ill-conditioned 1D chains
# ---------------------------------------------------------------------------

def chain_solve_longdouble(nodes, bc_nodes, bc_vals, kappa, f, gbar_fn=None, reference_rounding=False):
    """The SAME discrete system as `assemble_dense` + `apply_bc_and_solve` for a chain mesh
    (elements[e] = (e, e+1); reference solver.py:73-98, :153-183), but assembled and
    LU-factorised (tridiagonal, no pivoting: the matrix is SPD) in numpy.longdouble.

    Why: cond(K_free) ~ 0.4 N^2, so at N = 10^4 the fp64 LU the reference runs
    (torch.linalg.solve, solver.py:174) is itself ~4e-10 away from the exact discrete
    solution (measured: dense LAPACK and SuperLU agree with each other to 5e-15 and both
    differ from this routine by 4.04e-10).  Parity at that size is judged against this
    routine, and the fp64 oracle's own distance to it is reported next to ours.
    reference_rounding=True: the matrix and load are the ones the reference ASSEMBLES IN FP64 -- weights
    k_e = fl(kappa/h_e), diagonal fl(k_{i-1} + k_i) (solver.py:88-92), F_i = fl(h/2 f_i) summed in element order
    (solver.py:95-96) -- and only the SOLVE runs in extended precision: the exact solution of the reference's own
    rounded system, the yardstick of the HIP path's reference-order chain mode where the fp64 LU's forward error
    (up to 3e-10 in per-element gradients at 10^4 elements, measured) is in the way.
    Returns u (and, with gbar_fn, dkappa per element and df) as float64.
    """
    LD = np.longdouble
    x64 = np.asarray(nodes, dtype=np.float64)[:, 0]
    x = x64.astype(LD)
    n = len(x)
    h = x[1:] - x[:-1]                                   # solver.py:84-86  (exact: a difference of two doubles
    #                                                      is representable in 64-bit-mantissa long double)
    k = _kappa_per_element(kappa, n - 1).astype(LD) / h  # solver.py:88
    w = np.zeros(n, dtype=LD)
    w[:-1] += h / 2
    w[1:] += h / 2                                       # solver.py:95-96
    if reference_rounding:
        h64 = x64[1:] - x64[:-1]
        k = (_kappa_per_element(kappa, n - 1) / h64).astype(LD)
    is_bc = np.zeros(n, dtype=bool)
    g = np.zeros(n, dtype=LD)
    is_bc[np.asarray(bc_nodes, dtype=np.int64)] = True
    g[np.asarray(bc_nodes, dtype=np.int64)] = np.asarray(bc_vals, dtype=np.float64)
    diag = np.zeros(n, dtype=LD)
    diag[:-1] += k
    diag[1:] += k                                        # solver.py:89,92
    if reference_rounding:
        d64 = np.zeros(n)
        d64[1:] += k.astype(np.float64)                  # K[j,j] += k_e comes first for node j = e+1 ...
        d64[:-1] = d64[:-1] + k.astype(np.float64)       # ... then K[i,i] += k_{e+1}: fl(k_{i-1} + k_i)
        diag = d64.astype(LD)
    off = -k                                             # solver.py:90-91

    def tri_solve(rhs):
        """Thomas on the eliminated system: identity rows at Dirichlet nodes."""
        d = diag.copy()
        lo = off.copy()   # couples i+1 -> i  (sub-diagonal entry of row i+1)
        up = off.copy()   # couples i -> i+1
        r = rhs.copy()
        for i in range(n):
            if is_bc[i]:
                d[i] = 1.0
                if i > 0:
                    lo[i - 1] = 0.0
                if i < n - 1:
                    up[i] = 0.0
        for i in range(n - 1):      # also cut couplings INTO Dirichlet columns
            if is_bc[i + 1]:
                up[i] = 0.0
            if is_bc[i]:
                lo[i] = 0.0
        for i in range(1, n):
            mlt = lo[i - 1] / d[i - 1]
            d[i] = d[i] - mlt * up[i - 1]
            r[i] = r[i] - mlt * r[i - 1]
        sol = np.zeros(n, dtype=LD)
        sol[-1] = r[-1] / d[-1]
        for i in range(n - 2, -1, -1):
            sol[i] = (r[i] - up[i] * sol[i + 1]) / d[i]
        return sol

    F = np.asarray(f, dtype=np.float64).astype(LD) * w
    if reference_rounding:
        f64 = np.asarray(f, dtype=np.float64)
        F64 = np.zeros(n)
        F64[1:] = (h64 / 2.0) * f64[1:]
        F64[:-1] = F64[:-1] + (h64 / 2.0) * f64[:-1]
        F = F64.astype(LD)
    rhs = F.copy()
    # lifting: F_free -= K[free,bc] g  (solver.py:166-169)
    for i in range(n):
        if is_bc[i]:
            continue
        if i > 0 and is_bc[i - 1]:
            rhs[i] -= off[i - 1] * g[i - 1]
        if i < n - 1 and is_bc[i + 1]:
            rhs[i] -= off[i] * g[i + 1]
    rhs[is_bc] = 0.0
    u = tri_solve(rhs) + g
    if gbar_fn is None:
        return u.astype(np.float64)
    gb = np.asarray(gbar_fn(u.astype(np.float64)), dtype=np.float64).astype(LD)
    gb[is_bc] = 0.0
    lam = tri_solve(gb)
    dk = -(lam[1:] - lam[:-1]) * (u[1:] - u[:-1]) / h
    df = lam * w
    return u.astype(np.float64), dk.astype(np.float64), df.astype(np.float64)

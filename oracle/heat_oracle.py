"""CPU oracle for the reaction-diffusion solve and the heat-equation time stepping.  TEST INFRASTRUCTURE ONLY.

Parity status: PARITY UNPINNED against the reference -- the reference has no reaction term and no time-dependent
code ("heat equation" is an item of its README roadmap, README.md:139-143); there is nothing of its own to pin to.
This module restates OUR discretisation on the CPU with the pinned pieces of `oracle/p1_oracle.py` (the
reference's stiffness matrix, load vector and Dirichlet elimination, solver.py:73-183) plus the lumped mass
M_L = row sums of the reference's load map (solver.py:95-96, :143-145), and is itself checked against closed-form
solutions in tests/test_oracle_golden.py (eigenmode decay of the discrete operator, manufactured reaction-diffusion
solution).  Only `tests/` import it; the product never does.
"""
from __future__ import annotations

import numpy as np

try:
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
except Exception:  # pragma: no cover
    sp = None

from . import p1_oracle as orc


def lumped_mass(nodes, elements):
    """m_i = sum over the elements at node i of |e| / (dim + 1)."""
    nodes = np.asarray(nodes, dtype=np.float64)
    elements = np.asarray(elements, dtype=np.int64)
    _, w = orc.element_matrices(nodes, elements)          # w = element length (1D) / area (2D)
    m = np.zeros(nodes.shape[0])
    for p in range(elements.shape[1]):
        np.add.at(m, elements[:, p], w / elements.shape[1])
    return m


class ReactionDiffusion:
    """(K(kappa) + c M_L) u = M f + load with Dirichlet elimination; factorised once, reused for every right-hand
    side and for the adjoint (the operator is symmetric)."""

    def __init__(self, nodes, elements, bc_nodes, bc_vals, kappa, reaction):
        self.nodes = np.asarray(nodes, dtype=np.float64)
        self.elements = np.asarray(elements, dtype=np.int64)
        self.n = self.nodes.shape[0]
        self.bc_nodes = np.asarray(bc_nodes, dtype=np.int64)
        self.bc_vals = np.asarray(bc_vals, dtype=np.float64)
        self.free = orc.free_nodes(self.n, self.bc_nodes)
        self.mass = lumped_mass(self.nodes, self.elements)
        self.c = float(reaction)
        K, _ = orc.assemble_sparse(self.nodes, self.elements, kappa, np.zeros(self.n))
        self.A = (K + self.c * sp.diags(self.mass)).tocsr()
        self.lu = spla.splu(self.A[self.free][:, self.free].tocsc())
        self.k0, _ = orc.element_matrices(self.nodes, self.elements)

    def _refined(self, x_free, rhs_free, steps):
        """Iterative refinement of an LU solve with residuals in extended precision: the exact solution of the
        fp64-stored system to ~1e-15, where the plain LU result carries cond * eps (near-singular operators: no
        Dirichlet node and a small reaction coefficient, cond 1e8-1e9, measured 2e-10 in u)."""
        if steps <= 0:
            return x_free
        if getattr(self, "_Aff_ld", None) is None:
            self._Aff_ld = self.A[self.free][:, self.free].astype(np.longdouble)
        x = x_free.astype(np.longdouble)
        b = np.asarray(rhs_free, dtype=np.longdouble)
        for _ in range(steps):
            r = b - self._Aff_ld @ x
            x = x + self.lu.solve(np.asarray(r, dtype=np.float64))
        return np.asarray(x, dtype=np.float64)

    def solve(self, f, load=None, refine=0):
        F = orc.load_vector(self.nodes, self.elements, np.asarray(f, dtype=np.float64))
        if load is not None:
            F = F + np.asarray(load, dtype=np.float64)
        u = np.zeros(self.n)
        u[self.bc_nodes] = self.bc_vals
        rhs = F[self.free] - self.A[self.free][:, self.bc_nodes] @ self.bc_vals
        u[self.free] = self._refined(self.lu.solve(rhs), rhs, refine)
        return u

    def adjoint(self, u, gbar, refine=0):
        """lambda, dL/dkappa per element, dL/df, dL/dload for the cotangent gbar of `u = solve(...)`."""
        lam = np.zeros(self.n)
        g = np.asarray(gbar, dtype=np.float64)[self.free]
        lam[self.free] = self._refined(self.lu.solve(g, trans="T"), g, refine)      # the operator is symmetric
        lam_e, u_e = lam[self.elements], u[self.elements]
        dkappa = -np.einsum("ep,epq,eq->e", lam_e, self.k0, u_e)
        df = _load_transpose(self.nodes, self.elements, lam)
        return lam, dkappa, df, lam.copy()

    def gradient_scale(self, u, lam):
        """sum_{p,q} |lambda_p| |k0_e[p,q]| |u_q| per element: the magnitude of the terms dL/dkappa_e is made of.  Where
        u is nearly constant over an element they cancel, and fp64 defines the gradient only to u * that magnitude."""
        return np.einsum("ep,epq,eq->e", np.abs(lam[self.elements]), np.abs(self.k0), np.abs(u[self.elements]))


def _load_transpose(nodes, elements, lam):
    """M^T lambda for the load map of solver.py:95-96 (1D) / :143-145 (2D) (as p1_oracle.solve_with_adjoint)."""
    _, w = orc.element_matrices(nodes, elements)
    lam_e = lam[elements]
    df = np.zeros(nodes.shape[0])
    if nodes.shape[1] == 1:
        np.add.at(df, elements[:, 0], w / 2.0 * lam_e[:, 0])
        np.add.at(df, elements[:, 1], w / 2.0 * lam_e[:, 1])
    else:
        s = (lam_e[:, 0] + lam_e[:, 1] + lam_e[:, 2]) * (w / 9.0)
        for p in range(3):
            np.add.at(df, elements[:, p], s)
    return df


def heat_march(nodes, elements, bc_nodes, bc_vals, kappa, u0, dt, n_steps, f=None, theta=1.0, gbar_fn=None,
               with_scale=False):
    """Theta-scheme time stepping exactly as diffhe/heat.py states it (lumped mass; theta = 1 backward Euler,
    theta = 1/2 Crank-Nicolson in incremental form).  f: None, array, or callable t -> array.
    Returns the (n_steps + 1, n) history; with gbar_fn (cotangent of the FINAL state) also
    (dL/dkappa per element, dL/du0) by the discrete adjoint marched backwards; with_scale=True appends the summed
    `gradient_scale` of the steps."""
    rd = ReactionDiffusion(nodes, elements, bc_nodes, bc_vals, kappa, 1.0 / (theta * dt))
    n = rd.n
    u = np.array(u0, dtype=np.float64)
    u[rd.bc_nodes] = rd.bc_vals
    hist, ws = [u.copy()], []
    for k in range(n_steps):
        fk = f((k + theta) * dt) if callable(f) else (np.zeros(n) if f is None else f)
        w = rd.solve(fk, load=rd.mass * u * rd.c)
        ws.append(w)
        u = w if theta == 1.0 else 2.0 * w - u
        hist.append(u.copy())
    hist = np.stack(hist)
    if gbar_fn is None:
        return hist
    ubar = np.asarray(gbar_fn(hist[-1]), dtype=np.float64).copy()
    dkappa = np.zeros(rd.elements.shape[0])
    scale = np.zeros(rd.elements.shape[0])
    for k in reversed(range(n_steps)):
        wbar = ubar if theta == 1.0 else 2.0 * ubar
        lam, dk, _, dload = rd.adjoint(ws[k], wbar)
        dkappa += dk
        scale += rd.gradient_scale(ws[k], lam)
        prev = rd.mass * rd.c * dload                  # through load = M_L u_k c (Dirichlet rows: lambda = 0)
        ubar = prev if theta == 1.0 else prev - ubar   # u_{k+1} = 2 w - u_k
    ubar[rd.bc_nodes] = 0.0                            # the initial state is overwritten by the Dirichlet values there
    return (hist, dkappa, ubar, scale) if with_scale else (hist, dkappa, ubar)

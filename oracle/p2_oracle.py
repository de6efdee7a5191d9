"""CPU oracle for QUADRATIC (P2, 6-node) triangles.  TEST INFRASTRUCTURE ONLY.

Parity status: PARITY UNPINNED against the reference -- "P2 elements" is an item of the reference's README roadmap
(README.md:139-143); the reference has no quadratic element, no mesh for one and nothing to generate fixtures from.
This module restates OUR P2 discretisation on the CPU, independently of the product's host code (its own integrals:
a 7-point degree-5 quadrature for stiffness and mass instead of the product's closed forms), with the reference's
Dirichlet elimination (solver.py:160-183), and is pinned to closed forms in tests/test_oracle_golden.py: exact
reproduction of quadratic solutions, third-order convergence in L2.  Only `tests/` import it; the product never does.

Element node order: [v0, v1, v2, m01, m12, m20] (three vertices, then the midpoints of edges 0-1, 1-2, 2-0);
phi_i = L_i (2 L_i - 1), phi_ij = 4 L_i L_j in barycentric coordinates L.
"""
from __future__ import annotations

import numpy as np

try:
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
except Exception:  # pragma: no cover
    sp = None

# Radon's 7-point rule on the reference triangle (degree 5): barycentric points and weights (sum = 1)
_A1, _B1 = 0.0597158717897698, 0.4701420641051151
_A2, _B2 = 0.7974269853530873, 0.1012865073234563
_QP = np.array([[1 / 3, 1 / 3, 1 / 3], [_A1, _B1, _B1], [_B1, _A1, _B1], [_B1, _B1, _A1],
                [_A2, _B2, _B2], [_B2, _A2, _B2], [_B2, _B2, _A2]])
_QW = np.array([0.225, 0.1323941527885062, 0.1323941527885062, 0.1323941527885062,
                0.1259391805448271, 0.1259391805448271, 0.1259391805448271])


def mesh_rectangle_p2(nx=4, ny=4, x_range=(0.0, 1.0), y_range=(0.0, 1.0), bc_value=0.0):
    """The mesh of `FEMesh.rectangle_p2`, restated with loops (small sizes only)."""
    Wn = 2 * nx + 1
    xs = np.linspace(x_range[0], x_range[1], Wn)
    ys = np.linspace(y_range[0], y_range[1], 2 * ny + 1)
    nodes = np.array([[x, y] for y in ys for x in xs])
    el = []
    for i in range(ny):
        for j in range(nx):
            a = 2 * i * Wn + 2 * j
            b, d, c = a + 2, a + 2 * Wn, a + 2 * Wn + 2
            el.append([a, b, d, a + 1, a + Wn + 1, a + Wn])
            el.append([b, c, d, a + Wn + 2, a + 2 * Wn + 1, a + Wn + 1])
    on = [k for k, (x, y) in enumerate(nodes) if np.isclose(x, x_range[0]) or np.isclose(x, x_range[1])
          or np.isclose(y, y_range[0]) or np.isclose(y, y_range[1])]
    return nodes, np.asarray(el, dtype=np.int64), np.asarray(on, dtype=np.int64), np.full(len(on), float(bc_value))


def shape(L):
    """phi (6,) at barycentric point L."""
    return np.array([L[0] * (2 * L[0] - 1), L[1] * (2 * L[1] - 1), L[2] * (2 * L[2] - 1),
                     4 * L[0] * L[1], 4 * L[1] * L[2], 4 * L[2] * L[0]])


def dshape(L):
    """d phi / d L (6, 3) at barycentric point L."""
    return np.array([[4 * L[0] - 1, 0, 0], [0, 4 * L[1] - 1, 0], [0, 0, 4 * L[2] - 1],
                     [4 * L[1], 4 * L[0], 0], [0, 4 * L[2], 4 * L[1]], [4 * L[2], 0, 4 * L[0]]])


def element_matrices(nodes, elements):
    """Unit-kappa stiffness k0 (m, 6, 6), mass m0 (m, 6, 6) and areas (m,) by quadrature."""
    nodes = np.asarray(nodes, dtype=np.float64)
    m = len(elements)
    k0, m0, area = np.zeros((m, 6, 6)), np.zeros((m, 6, 6)), np.zeros(m)
    for e, el in enumerate(elements):
        (xi, yi), (xj, yj), (xk, yk) = nodes[el[0]], nodes[el[1]], nodes[el[2]]
        det = (xj - xi) * (yk - yi) - (xk - xi) * (yj - yi)
        A = 0.5 * abs(det)
        area[e] = A
        if A < 1e-15:
            continue
        gL = np.array([[yj - yk, xk - xj], [yk - yi, xi - xk], [yi - yj, xj - xi]]) / det     # grad L_i
        for L, w in zip(_QP, _QW):
            G = dshape(L) @ gL                       # (6, 2)
            N = shape(L)
            k0[e] += w * A * (G @ G.T)
            m0[e] += w * A * np.outer(N, N)
    return k0, m0, area


class P2Problem:
    """K(kappa) u = M f with Dirichlet elimination (solver.py:160-183) on 6-node triangles."""

    def __init__(self, nodes, elements, bc_nodes, bc_vals, kappa):
        self.nodes = np.asarray(nodes, dtype=np.float64)
        self.elements = np.asarray(elements, dtype=np.int64)
        self.n, self.m = len(self.nodes), len(self.elements)
        self.bc_nodes = np.asarray(bc_nodes, dtype=np.int64)
        self.bc_vals = np.asarray(bc_vals, dtype=np.float64)
        mask = np.ones(self.n, dtype=bool)
        mask[self.bc_nodes] = False
        self.free = np.nonzero(mask)[0]
        self.k0, self.m0, self.area = element_matrices(self.nodes, self.elements)
        kap = np.broadcast_to(np.asarray(kappa, dtype=np.float64), (self.m,))
        rows = np.repeat(self.elements, 6, axis=1).ravel()
        cols = np.tile(self.elements, (1, 6)).ravel()
        self.K = sp.csr_matrix(((kap[:, None, None] * self.k0).ravel(), (rows, cols)), shape=(self.n, self.n))
        self.M = sp.csr_matrix((self.m0.ravel(), (rows, cols)), shape=(self.n, self.n))
        self.lu = spla.splu(self.K[self.free][:, self.free].tocsc())

    def solve(self, f):
        F = self.M @ np.asarray(f, dtype=np.float64)
        u = np.zeros(self.n)
        u[self.bc_nodes] = self.bc_vals
        u[self.free] = self.lu.solve(F[self.free] - self.K[self.free][:, self.bc_nodes] @ self.bc_vals)
        return u

    def adjoint(self, u, gbar):
        """(dL/dkappa per element, dL/df) for the cotangent gbar of u = solve(f)."""
        lam = np.zeros(self.n)
        lam[self.free] = self.lu.solve(np.asarray(gbar, dtype=np.float64)[self.free], trans="T")
        dk = -np.einsum("ep,epq,eq->e", lam[self.elements], self.k0, u[self.elements])
        return dk, self.M.T @ lam

    def l2_error(self, u, exact):
        """sqrt(int (u_h - exact)^2) with the 7-point rule per element."""
        err = 0.0
        for e, el in enumerate(self.elements):
            v = self.nodes[el[:3]]
            for L, w in zip(_QP, _QW):
                xy = L @ v
                err += w * self.area[e] * (shape(L) @ u[el] - exact(xy[0], xy[1])) ** 2
        return float(np.sqrt(err))

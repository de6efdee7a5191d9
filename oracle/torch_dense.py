"""Reference-faithful DENSE CPU baseline on PyTorch.  TEST INFRASTRUCTURE / REPORTED BASELINE ONLY.

The path BASELINE.json's north star names -- dense K, `torch.linalg.solve`, backward by autograd
(reference diffhe/solver.py:73-183) -- restated with VECTORISED assembly, i.e. minus the reference's
per-element Python loops and its O(n_bc n_free) elimination loop, so that it runs at
the sizes a dense matrix allows (1D 1000, 2D 64^2) in seconds instead of minutes.  SURVEY 8(d)
"B-dense".  Only `tests/` and the `cpu_baseline` leg of `bench.py` import this module; the product
never does.  Pinned against the reference's golden vectors in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import numpy as np
import torch

T64 = torch.float64


def assemble_dense(nodes, elements, kappa: torch.Tensor, f: torch.Tensor):
    """Dense K (n,n) and F (n) as the reference assembles them (solver.py:82-96 1D, :112-145 2D),
    one `index_put_(accumulate=True)` instead of the element loop; differentiable in kappa and f."""
    x = torch.as_tensor(nodes, dtype=T64)
    el = torch.as_tensor(elements, dtype=torch.long)
    n, npe = x.shape[0], el.shape[1]
    K = torch.zeros(n * n, dtype=T64)
    F = torch.zeros(n, dtype=T64)
    if npe == 2:
        i, j = el[:, 0], el[:, 1]
        h = x[j, 0] - x[i, 0]                                   # solver.py:84-86
        k = kappa / h                                           # solver.py:88
        K = K.index_put((torch.cat([i * n + i, i * n + j, j * n + i, j * n + j]),),
                        torch.cat([k, -k, -k, k]), accumulate=True)       # solver.py:89-92
        F = F.index_put((torch.cat([i, j]),), torch.cat([h / 2.0 * f[i], h / 2.0 * f[j]]), accumulate=True)
        return K.reshape(n, n), F
    i, j, k_ = el[:, 0], el[:, 1], el[:, 2]
    xi, yi, xj, yj, xk, yk = x[i, 0], x[i, 1], x[j, 0], x[j, 1], x[k_, 0], x[k_, 1]
    area = 0.5 * torch.abs((xj - xi) * (yk - yi) - (xk - xi) * (yj - yi))     # solver.py:119
    keep = area >= 1e-15                                                      # solver.py:120-121
    b = torch.stack([yj - yk, yk - yi, yi - yj])                              # solver.py:125-134 (detached there too)
    c = torch.stack([xk - xj, xi - xk, xj - xi])
    idx = torch.stack([i, j, k_])
    rows, vals = [], []
    den = torch.where(keep, 4.0 * area, torch.ones_like(area))
    for p in range(3):
        for q in range(3):
            rows.append(idx[p] * n + idx[q])
            vals.append(torch.where(keep, kappa * (b[p] * b[q] + c[p] * c[q]) / den, torch.zeros_like(area)))
    K = K.index_put((torch.cat(rows),), torch.cat(vals), accumulate=True)     # solver.py:137-140
    fc = (f[i] + f[j] + f[k_]) / 3.0                                          # solver.py:143
    contrib = torch.where(keep, area / 3.0 * fc, torch.zeros_like(area))
    F = F.index_put((torch.cat([i, j, k_]),), torch.cat([contrib, contrib, contrib]), accumulate=True)
    return K.reshape(n, n), F


def solve(nodes, elements, bc_nodes, bc_vals, kappa: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
    """u (n) through Dirichlet elimination + torch.linalg.solve (solver.py:153-183)."""
    n = len(nodes)
    K, F = assemble_dense(nodes, elements, kappa, f)
    bc = torch.as_tensor(np.asarray(bc_nodes), dtype=torch.long)
    g = torch.as_tensor(np.asarray(bc_vals), dtype=T64)
    mask = torch.ones(n, dtype=torch.bool)
    mask[bc] = False
    free = torch.nonzero(mask).squeeze(1)
    F_free = F[free] - K[free][:, bc] @ g                       # solver.py:165-169
    u_free = torch.linalg.solve(K[free][:, free], F_free)       # solver.py:171-174
    u = torch.zeros(n, dtype=T64)
    u = u.index_put((bc,), g)
    return u.index_put((free,), u_free)                         # solver.py:177-181


def differentiable_solve(nodes, elements, bc_nodes, bc_vals, kappa: float, f):
    """One fwd + autograd backward of L = sum u^2: returns (u, dL/dkappa, dL/df) as numpy."""
    k = torch.tensor(float(kappa), dtype=T64, requires_grad=True)
    ff = torch.as_tensor(np.asarray(f), dtype=T64).clone().requires_grad_(True)
    u = solve(nodes, elements, bc_nodes, bc_vals, k, ff)
    (u ** 2).sum().backward()
    return u.detach().numpy(), float(k.grad), ff.grad.numpy()

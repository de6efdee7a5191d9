"""Per-mesh solve plan: the batch-invariant integer metadata the HIP kernels need.

Built once per (mesh, device) with vectorised numpy and cached on the mesh object:
  * chain detection + Dirichlet-delimited segments for the 1D scan solver;
  * the ELL pattern shared by the whole batch, the per-entry contribution lists
    of the deterministic gather assembly, the per-element slot map of the atomic
    assembly;
  * element integrals k0 / m0 and the load matrix M (computed ON THE DEVICE by
    `diffhe_p1_element_integrals` / `diffhe_ell_assemble_rows`).

Nothing here solves anything: it replaces the bookkeeping of reference
mesh.py:127-129 (`free_nodes`) and the index arithmetic implicit in the dense
`K[i, j]` writes of solver.py:89-92 / :137-140.
"""
from __future__ import annotations

import collections
import hashlib
import os
import shutil
import threading
from typing import Optional

import numpy as np
import torch

from . import _hip


_TLS = threading.local()


def status_buffer() -> torch.Tensor:
    """Pinned 4-int host buffer the solve entry points report through (iterations, unconverged count, ...).
    One per calling THREAD: a forward on the main thread and the backward of another solve on autograd's thread
    never share it (the library reads/writes it while the GIL is released inside ctypes)."""
    buf = getattr(_TLS, "status", None)
    if buf is None:
        buf = _TLS.status = torch.zeros(4, dtype=torch.int32).pin_memory()
    return buf


def host_arrays(tag: str, build):
    """`build()` -> dict of numpy arrays, through an optional on-disk cache shared by the ranks of a node.

    With DIFFHE_PLAN_CACHE=<dir> set, the first process to need `tag` builds it and publishes one .npy per array
    (written under a private name, then renamed: readers never see a partial entry); every other process maps the
    files instead of repeating the numpy work (the fine-level gather lists and reference-order integrals of a
    1024^2 lattice are ~5 s of host time per rank -- eight ranks starting together would each pay it).  Unset: no
    cache, `build()` runs."""
    root = os.environ.get("DIFFHE_PLAN_CACHE")
    if not root:
        return build()
    path = os.path.join(root, tag)
    if not os.path.isdir(path):
        arrays = build()
        tmp = f"{path}.tmp{os.getpid()}.{threading.get_ident()}"
        os.makedirs(tmp, exist_ok=True)
        for name, a in arrays.items():
            np.save(os.path.join(tmp, name + ".npy"), np.asarray(a))
        try:
            os.rename(tmp, path)
        except OSError:                      # another rank published the same entry first
            shutil.rmtree(tmp, ignore_errors=True)
        return arrays
    return {fn[:-4]: np.load(os.path.join(path, fn), mmap_mode="r") for fn in sorted(os.listdir(path))
            if fn.endswith(".npy")}


def padded_batch(B: int) -> int:
    """Bp: next power of two up to 64, else the next multiple of 64."""
    if B <= 64:
        p = 1
        while p < B:
            p *= 2
        return p
    return ((B + 63) // 64) * 64


def _bc_arrays(mesh):
    n = mesh.n_nodes
    is_bc = np.zeros(n, dtype=np.uint8)
    g = np.zeros(n, dtype=np.float64)
    if mesh.dirichlet_nodes:
        keys = np.fromiter(mesh.dirichlet_nodes.keys(), dtype=np.int64, count=len(mesh.dirichlet_nodes))
        vals = np.fromiter((float(v) for v in mesh.dirichlet_nodes.values()), dtype=np.float64,
                           count=len(mesh.dirichlet_nodes))
        is_bc[keys] = 1
        g[keys] = vals
    return is_bc, g


def chain_segments(n: int, is_bc: np.ndarray) -> np.ndarray:
    """(n_seg, 3) int32 rows (first node, last node, flags); flags bit0/bit1 = the
    first/last node is Dirichlet.  Segments are the maximal runs between Dirichlet nodes."""
    d = np.nonzero(is_bc)[0]
    if len(d) == 0:
        return np.array([[0, n - 1, 0]], dtype=np.int32)
    segs = []
    if d[0] > 0:
        segs.append((0, int(d[0]), 2))
    for a, b in zip(d[:-1], d[1:]):
        segs.append((int(a), int(b), 3))
    if d[-1] < n - 1:
        segs.append((int(d[-1]), n - 1, 1))
    if not segs:  # n == 1
        segs.append((0, n - 1, 3))
    return np.asarray(segs, dtype=np.int32)


def build_ell_pattern(elements: np.ndarray, n: int):
    """ELL pattern + gather lists from the connectivity.

    Returns dict(W, cols (W,n) i32, ent_ptr (W*n+1) i32, contrib i32, slot_of (npe*npe, m) i32).
    Entry (row i, slot k) is stored at k*n + i; slot 0 is the diagonal; unused slots
    point at the row itself and carry no contribution.
    """
    m, npe = elements.shape
    _check_gather_code_range(m)
    nloc = npe * npe
    e_idx = np.repeat(np.arange(m, dtype=np.int64), nloc)
    pq = np.tile(np.arange(nloc, dtype=np.int64), m)
    rows = elements[e_idx, pq // npe].astype(np.int64)
    cols = elements[e_idx, pq % npe].astype(np.int64)
    # every row owns a diagonal entry even if no element touches the node
    rows_all = np.concatenate([np.arange(n, dtype=np.int64), rows])
    cols_all = np.concatenate([np.arange(n, dtype=np.int64), cols])
    # sort key: row, then (diagonal first), then column
    offdiag = (rows_all != cols_all).astype(np.int64)
    key = (rows_all * 2 + offdiag) * n + cols_all
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    new_entry = np.ones(len(key_s), dtype=bool)
    new_entry[1:] = key_s[1:] != key_s[:-1]
    entry_id = np.cumsum(new_entry) - 1                     # entry index in (row, slot) order
    ent_rows = rows_all[order][new_entry]
    ent_cols = cols_all[order][new_entry]
    row_start = np.searchsorted(ent_rows, np.arange(n))
    slot = np.arange(len(ent_rows)) - row_start[ent_rows]   # slot within the row
    W = int(slot.max()) + 1
    ell_cols = np.tile(np.arange(n, dtype=np.int32), W)
    ell_index = slot * n + ent_rows                          # k*n + i
    ell_cols[ell_index] = ent_cols.astype(np.int32)
    # contributions (drop the synthetic diagonals, which sit in the first n inputs)
    inv = np.empty(len(order), dtype=np.int64)
    inv[order] = np.arange(len(order))
    contrib_entry = ell_index[entry_id[inv[n:]]]             # ELL index of each (e, pq)
    code = (e_idx * 64 + pq).astype(np.int64)                # decoded by assemble_rows_kernel: e = code >> 6
    corder = np.argsort(contrib_entry, kind="stable")        # fixed order: deterministic sums
    contrib = code[corder].astype(np.int32)
    counts = np.bincount(contrib_entry, minlength=W * n)
    ent_ptr = np.zeros(W * n + 1, dtype=np.int64)
    np.cumsum(counts, out=ent_ptr[1:])
    slot_of = (contrib_entry // n).reshape(m, nloc).T.copy().astype(np.int32)
    if ent_ptr[-1] >= 2 ** 31:
        raise ValueError("mesh too large for int32 gather lists")
    return dict(W=W, cols=ell_cols.reshape(W, n), ent_ptr=ent_ptr.astype(np.int32), contrib=contrib,
                slot_of=slot_of)


def _check_gather_code_range(m: int) -> None:
    """Gather codes are e * 64 + pq in an int32 (assemble_rows_kernel decodes e = code >> 6): 2^25 elements at most
    (a 4096 x 4096 lattice).  Beyond that the cast would wrap silently and the kernel would read out of bounds."""
    if m >= 2 ** 25:
        raise ValueError(f"mesh too large for int32 gather lists: {m} elements >= 2**25")


def detect_lattice(elements: np.ndarray, n: int):
    """(nx, ny) if `elements` is exactly the FEMesh.rectangle connectivity
    (reference mesh.py:92-105: quad (a,b,c,d) -> [a,b,d], [b,c,d], row-major), else None."""
    m = len(elements)
    if elements.shape[1] != 3 or m < 8 or m % 2:
        return None
    nx = int(elements[0, 2]) - 1
    if nx < 2 or (m // 2) % nx:
        return None
    ny = (m // 2) // nx
    if ny < 2 or (nx + 1) * (ny + 1) != n:
        return None
    return (nx, ny) if np.array_equal(elements, lattice_elements(nx, ny)) else None


def lattice_elements(nx: int, ny: int) -> np.ndarray:
    row, col = np.divmod(np.arange(nx * ny, dtype=np.int64), nx)
    a = row * (nx + 1) + col
    tris = np.empty((2 * nx * ny, 3), dtype=np.int64)
    tris[0::2, 0], tris[0::2, 1], tris[0::2, 2] = a, a + 1, a + nx + 1
    tris[1::2, 0], tris[1::2, 1], tris[1::2, 2] = a + 1, a + nx + 2, a + nx + 1
    return tris


def build_dia_pattern(nx: int, ny: int):
    """Gather lists of the symmetric-diagonal lattice format, written down analytically from the FEMesh.rectangle
    connectivity (reference mesh.py:100-105: quad (r, c) -> T0 = [a, b, d] = element 2q, T1 = [b, c, d] = element
    2q + 1, q = r nx + c) -- no sort over the 9 m local entries (that cost 8 s of host time at 1024^2).

    Seven entries per row in the fixed order offsets (0, +1, +W, +nx, -1, -W, -nx), W = nx+1;
    the first four can be stored (store_slot), the lower three only feed the Dirichlet lift.  The contributions
    of an entry are listed in ELEMENT ORDER (then local-entry order): the order the reference's loop adds them.
    Returns dict(We=7, cols (7,n) i32, ent_ptr (7n+1) i32, contrib i32 packed e * 64 + p * 3 + q)."""
    n, W = (nx + 1) * (ny + 1), nx + 1
    _check_gather_code_range(2 * nx * ny)
    r, c = np.divmod(np.arange(n, dtype=np.int64), W)
    up, dn, lf, rt = r < ny, r >= 1, c >= 1, c < nx          # a quad exists above / below / left / right of the node
    q = lambda rr, cc: rr * nx + cc                         # noqa: E731  quad index
    T0 = lambda rr, cc: 2 * q(rr, cc)                       # noqa: E731
    T1 = lambda rr, cc: 2 * q(rr, cc) + 1                   # noqa: E731
    # per entry kind: (column offset, [(valid mask, element id, local entry p*3+q), ...] in increasing element id)
    kinds = [
        (0, [(dn & lf, T1(r - 1, c - 1), 4), (dn & rt, T0(r - 1, c), 8), (dn & rt, T1(r - 1, c), 8),
             (up & lf, T0(r, c - 1), 4), (up & lf, T1(r, c - 1), 0), (up & rt, T0(r, c), 0)]),
        (1, [(rt & dn, T1(r - 1, c), 7), (rt & up, T0(r, c), 1)]),                    # east: edges d-c, a-b
        (W, [(up & lf, T1(r, c - 1), 1), (up & rt, T0(r, c), 2)]),                    # north: edges b-c, a-d
        (nx, [(up & lf, T0(r, c - 1), 5), (up & lf, T1(r, c - 1), 2)]),               # quad diagonal b-d, seen from b
        (-1, [(lf & dn, T1(r - 1, c - 1), 5), (lf & up, T0(r, c - 1), 3)]),           # west
        (-W, [(dn & lf, T1(r - 1, c - 1), 3), (dn & rt, T0(r - 1, c), 6)]),           # south
        (-nx, [(dn & rt, T0(r - 1, c), 7), (dn & rt, T1(r - 1, c), 6)]),              # quad diagonal, seen from d
    ]
    counts = np.zeros(7 * n, dtype=np.int32)
    cols = np.tile(np.arange(n, dtype=np.int32), 7)
    parts = []
    ar = np.arange(n, dtype=np.int32)
    for k, (off, cands) in enumerate(kinds):
        valid = np.stack([v for v, _, _ in cands], axis=1)                           # (n, C)
        codes = np.stack([(e * 64 + pq).astype(np.int32) for _, e, pq in cands], axis=1)
        cnt = valid.sum(axis=1, dtype=np.int32)
        counts[k * n:(k + 1) * n] = cnt
        np.copyto(cols[k * n:(k + 1) * n], ar + np.int32(off), where=cnt > 0)
        parts.append(codes[valid])               # row-major: per entry, candidates in increasing element id
    ent_ptr = np.zeros(7 * n + 1, dtype=np.int64)
    np.cumsum(counts, out=ent_ptr[1:])
    contrib = np.concatenate(parts)
    return dict(We=7, cols=cols.reshape(7, n), ent_ptr=ent_ptr.astype(np.int32), contrib=contrib)


def _build_dia_pattern_sorted(nx: int, ny: int):
    """The same lists from a stable sort over all 9 m local entries (the definition; kept as the test's yardstick)."""
    n, W = (nx + 1) * (ny + 1), nx + 1
    _check_gather_code_range(2 * nx * ny)
    el = lattice_elements(nx, ny)
    m = len(el)
    e_idx = np.repeat(np.arange(m, dtype=np.int64), 9)
    pq = np.tile(np.arange(9, dtype=np.int64), m)
    rows = el[e_idx, pq // 3]
    cols = el[e_idx, pq % 3]
    d = cols - rows
    k = np.full(d.shape, -1, dtype=np.int64)
    for kk, off in enumerate((0, 1, W, nx, -1, -W, -nx)):
        k[d == off] = kk
    assert (k >= 0).all()
    ent = k * n + rows
    ell_cols = np.tile(np.arange(n, dtype=np.int32), 7)
    ell_cols[ent] = cols.astype(np.int32)
    order = np.argsort(ent, kind="stable")
    contrib = (e_idx * 64 + pq)[order].astype(np.int32)
    ent_ptr = np.zeros(7 * n + 1, dtype=np.int64)
    np.cumsum(np.bincount(ent, minlength=7 * n), out=ent_ptr[1:])
    return dict(We=7, cols=ell_cols.reshape(7, n), ent_ptr=ent_ptr.astype(np.int32), contrib=contrib)


def coarsening_step(nodes2d: np.ndarray):
    """(row step, column step) for the next multigrid level of a lattice with node array (ny+1, nx+1, 2),
    or None to stop.  While the cells are anisotropic (mean edge ratio > 1.6) only the direction of the
    SHORT edges -- the strongly coupled one, which point smoothing cannot handle -- is coarsened
    (semi-coarsening); otherwise both are halved.  A direction is only halved while it stays even and >= 4."""
    ny, nx = nodes2d.shape[0] - 1, nodes2d.shape[1] - 1
    hx = float(np.mean(np.linalg.norm(nodes2d[:, 1:] - nodes2d[:, :-1], axis=2)))
    hy = float(np.mean(np.linalg.norm(nodes2d[1:, :] - nodes2d[:-1, :], axis=2)))
    can_x = nx % 2 == 0 and nx >= 4
    can_y = ny % 2 == 0 and ny >= 4
    if hy * 1.6 < hx:                      # short vertical edges: coarsen rows (y) only
        return (2, 1) if can_y else None
    if hx * 1.6 < hy:
        return (1, 2) if can_x else None
    if can_x and can_y:
        return (2, 2)
    return None


def reference_order_integrals(coords: np.ndarray, elems: np.ndarray):
    """Numerators and denominators of the element stiffness entries in the reference's operation order, computed
    with numpy (every operation rounded on its own, like torch's): 2D t[pq] = b_p b_q + c_p c_q, den = 4.0 area
    (solver.py:119-139; skipped elements contribute 0); 1D t = +-1, den = h (solver.py:86-92).  coords (dim, n),
    elems (npe, m) as uploaded to the device.  Returns (tnum (npe*npe, m), den (m))."""
    dim = coords.shape[0]
    e = elems.astype(np.int64)
    if dim == 1:
        h = coords[0, e[1]] - coords[0, e[0]]
        m = e.shape[1]
        return np.stack([np.ones(m), -np.ones(m), -np.ones(m), np.ones(m)]), h
    xi, yi = coords[0, e[0]], coords[1, e[0]]
    xj, yj = coords[0, e[1]], coords[1, e[1]]
    xk, yk = coords[0, e[2]], coords[1, e[2]]
    area = 0.5 * np.abs((xj - xi) * (yk - yi) - (xk - xi) * (yj - yi))
    keep = area >= 1e-15
    b = np.stack([yj - yk, yk - yi, yi - yj])
    c = np.stack([xk - xj, xi - xk, xj - xi])
    t = (b[:, None, :] * b[None, :, :] + c[:, None, :] * c[None, :, :]).reshape(9, -1)
    t[:, ~keep] = 0.0
    return t, np.where(keep, 4.0 * area, 1.0)


def p2_element_integrals(nodes: np.ndarray, elements: np.ndarray):
    """(k0 (36, m), m0 (36, m)) of straight-sided 6-node triangles [v0, v1, v2, m01, m12, m20]: unit-kappa stiffness
    int grad phi_p . grad phi_q (3-point edge-midpoint rule: exact, the integrand is quadratic) and the consistent mass
    int phi_p phi_q (closed form).  phi_i = L_i (2 L_i - 1), phi_ij = 4 L_i L_j in barycentric coordinates L."""
    v = nodes[elements[:, :3]]                                   # (m, 3, 2)
    xi, yi, xj, yj, xk, yk = v[:, 0, 0], v[:, 0, 1], v[:, 1, 0], v[:, 1, 1], v[:, 2, 0], v[:, 2, 1]
    det = (xj - xi) * (yk - yi) - (xk - xi) * (yj - yi)
    area = 0.5 * np.abs(det)
    keep = area >= 1e-15                                         # degenerate triangles contribute nothing (solver.py:120)
    safe = np.where(keep, det, 1.0)
    gL = np.stack([np.stack([yj - yk, xk - xj], axis=1), np.stack([yk - yi, xi - xk], axis=1),
                   np.stack([yi - yj, xj - xi], axis=1)], axis=1) / safe[:, None, None]      # grad L_i, (m, 3, 2)
    k0 = np.zeros((len(elements), 6, 6))
    for L in ((0.5, 0.5, 0.0), (0.0, 0.5, 0.5), (0.5, 0.0, 0.5)):
        G = np.empty((len(elements), 6, 2))
        for i in range(3):
            G[:, i] = (4.0 * L[i] - 1.0) * gL[:, i]
        for e_, (i, j) in enumerate(((0, 1), (1, 2), (2, 0))):
            G[:, 3 + e_] = 4.0 * (L[i] * gL[:, j] + L[j] * gL[:, i])
        k0 += np.einsum("epd,eqd->epq", G, G) * (area / 3.0)[:, None, None]
    k0[~keep] = 0.0
    Mref = np.array([[6, -1, -1, 0, -4, 0], [-1, 6, -1, 0, 0, -4], [-1, -1, 6, -4, 0, 0],
                     [0, 0, -4, 32, 16, 16], [-4, 0, 0, 16, 32, 16], [0, -4, 0, 16, 16, 32]], dtype=np.float64) / 180.0
    m0 = Mref[None] * np.where(keep, area, 0.0)[:, None, None]
    return (np.ascontiguousarray(k0.reshape(-1, 36).T), np.ascontiguousarray(m0.reshape(-1, 36).T))


def _lumped_mass(nodes: np.ndarray, elements: np.ndarray) -> np.ndarray:
    n = nodes.shape[0]
    if nodes.shape[1] == 1:
        size = np.abs(nodes[elements[:, 1], 0] - nodes[elements[:, 0], 0])
    else:
        a, b, c = (nodes[elements[:, k]] for k in range(3))
        size = 0.5 * np.abs((b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (c[:, 0] - a[:, 0]) * (b[:, 1] - a[:, 1]))
    m = np.zeros(n)
    for k in range(elements.shape[1]):
        np.add.at(m, elements[:, k], size / elements.shape[1])
    return m


class LatticeLevel:
    """One level of the multigrid hierarchy of a lattice mesh: geometry, element integrals
    and gather lists on the device.  Level l uses every 2^l-th node of the fine mesh."""

    def __init__(self, nodes2d: np.ndarray, is_bc2d: np.ndarray, device, with_load_matrix: bool = False):
        L = _hip.lib()
        ny, nx = nodes2d.shape[0] - 1, nodes2d.shape[1] - 1
        self.nx, self.ny = nx, ny
        self.n, self.m = (nx + 1) * (ny + 1), 2 * nx * ny
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        self.coords = dev(nodes2d.reshape(self.n, 2).T)
        self.elems = dev(lattice_elements(nx, ny).T.astype(np.int32))
        self.is_bc = dev(is_bc2d.reshape(self.n).astype(np.uint8))
        pat = host_arrays(f"dia_{nx}x{ny}", lambda: build_dia_pattern(nx, ny))
        self.cols, self.ent_ptr, self.contrib = dev(pat["cols"]), dev(pat["ent_ptr"]), dev(pat["contrib"])
        self.k0 = torch.empty((9, self.m), dtype=torch.float64, device=device)
        m0 = torch.empty((9, self.m), dtype=torch.float64, device=device)
        _hip.check(L.diffhe_p1_element_integrals(_hip.ptr(self.coords), _hip.ptr(self.elems), 2, self.n, self.m,
                                                 _hip.ptr(self.k0), _hip.ptr(m0), _stream(device)),
                   "diffhe_p1_element_integrals")
        self._m0 = m0 if with_load_matrix else None
        flat = np.ascontiguousarray(nodes2d.reshape(self.n, 2).T)
        ref = host_arrays(f"refint_{nx}x{ny}_{hashlib.blake2b(flat.tobytes(), digest_size=8).hexdigest()}",
                          lambda: dict(zip(("tn", "dn"), reference_order_integrals(flat, lattice_elements(nx, ny).T))))
        self.tnum, self.den = dev(ref["tn"]), dev(ref["dn"])   # reference-order assembly (fine level; kappa folded in)
        # quad-diagonal coupling b-d: local (1,2) of [a,b,d], local (0,2) of [b,c,d]; exactly 0 for
        # right triangles -> 3 stored diagonals instead of 4
        hyp = max(float(self.k0[5, 0::2].abs().max()), float(self.k0[2, 1::2].abs().max()))
        self.nd = 3 if hyp == 0.0 else 4
        self.store_slot = dev(np.array([0, 1, 2, 3 if self.nd == 4 else -1, -1, -1, -1], dtype=np.int32))
        self._zero_g = None
        self._device = device
        self._nodes2d = nodes2d
        self._lumped = None
        # load matrix M in symmetric diagonals (all four: the quad diagonal couples through the centroid rule)
        self.Mvals = None
        if with_load_matrix:
            all4 = dev(np.array([0, 1, 2, 3, -1, -1, -1], dtype=np.int32))
            self.Mvals = torch.empty((4, self.n), dtype=torch.float64, device=device)
            _hip.check(L.diffhe_ell_assemble_rows(_hip.ptr(self._m0), None, 0, 0, _hip.ptr(self.ent_ptr),
                                                  _hip.ptr(self.contrib), _hip.ptr(self.cols), _hip.ptr(all4), None,
                                                  None, _hip.ptr(self.Mvals), None, self.n, self.m, 7, 1,
                                                  _stream(device)), "diffhe_ell_assemble_rows(M, lattice)")
            self._m0 = None

    def lumped_mass(self) -> torch.Tensor:
        """(n,) lumped mass of this level's own triangulation (coarse levels are re-discretised, like their stiffness)."""
        if self._lumped is None:
            m = _lumped_mass(np.ascontiguousarray(self._nodes2d.reshape(self.n, 2)), lattice_elements(self.nx, self.ny))
            self._lumped = torch.from_numpy(m).to(self._device)
        return self._lumped

    def k0ref(self) -> torch.Tensor:
        """(9, m) unit-kappa element stiffness entries fl(t / den) with t and den in the reference's operation order
        (solver.py:125-139): the batch-shared factors of the fast per-sample assembly."""
        if getattr(self, "_k0ref", None) is None:
            self._k0ref = (self.tnum / self.den.unsqueeze(0)).contiguous()
        return self._k0ref

    def mask32(self) -> torch.Tensor:
        """(n) fp32: 0 on Dirichlet rows, 1 elsewhere (what the two-samples-per-lane strip kernels scalar-load)."""
        if getattr(self, "_mask32", None) is None:
            self._mask32 = (1 - self.is_bc.to(torch.float32)).contiguous()
        return self._mask32

    def compact(self, which: str):
        """(9, 2) tensor = the element matrices `which` ("k0" | "k0ref") of the two triangle orientations, if those of all
        even and of all odd elements are BITWISE equal (a lattice with exactly representable spacing, e.g. N a power of two
        on the unit square: the bench mesh) -- what diffhe_lattice_assemble_rows / diffhe_lattice_grad_kappa take in
        compact form; None otherwise (jittered or skewed lattices, spacings with rounding).  Same numbers either way."""
        cache = self.__dict__.setdefault("_compact", {})
        if which not in cache:
            t = self.k0 if which == "k0" else self.k0ref()
            same = bool((t[:, 0::2] == t[:, 0:1]).all()) and bool((t[:, 1::2] == t[:, 1:2]).all())
            cache[which] = t[:, :2].contiguous() if same else None
        return cache[which]

    def zero_g(self):
        """Dirichlet values of a coarse level: corrections vanish there."""
        if self._zero_g is None:
            self._zero_g = torch.zeros(self.n, dtype=torch.float64, device=self._device)
        return self._zero_g


class SolvePlan:
    """Device-resident metadata of one mesh (see module docstring)."""

    def __init__(self, mesh, device: torch.device):
        L = _hip.lib()
        self.device = device
        self.dim = mesh.dim
        self.n = mesh.n_nodes
        self.m = mesh.n_elements
        self.npe = int(mesh.elements.shape[1])                        # 2 (1D), 3 (P1 triangles) or 6 (P2 triangles)
        if self.dim not in (1, 2):
            raise NotImplementedError("Only 1D and 2D supported")  # reference solver.py:67
        nodes = mesh.nodes.detach().to("cpu", torch.float64).numpy()
        elements = mesh.elements.detach().to("cpu", torch.int64).numpy()
        if self.npe != self.dim + 1 and not (self.dim == 2 and self.npe == 6):
            raise ValueError(f"expected {self.dim + 1} nodes per element (or 6: P2 triangles), got {self.npe}")
        self.is_p2 = self.npe == 6
        is_bc, g = _bc_arrays(mesh)
        self.n_bc = int(is_bc.sum())
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        self.is_bc = dev(is_bc)
        self.g = dev(g)
        self.has_dirichlet_data = bool(np.any(g != 0.0))     # some g_i != 0: u = x + g needs its own buffer
        self._bc_index = None
        self.coords = dev(nodes.T)                                   # (dim, n) SoA
        self.elems = dev(elements.T.astype(np.int32))                # (npe, m) SoA
        # warm-start vectors of DifferentiableFESolver(warm_start=...): (kind, Bp, kappa layout, reaction) -> previous
        # (n, Bp) solution.  Kept on the PLAN so that a loop that builds a new solver per step (the reference's pattern)
        # still benefits; keyed by what makes two solves comparable, so solvers with another kappa layout or reaction
        # term never feed each other's guesses; at most `_WARM_MAX` vectors live (least recently used goes first).
        self.warm = collections.OrderedDict()
        self._lock = threading.Lock()
        self._build_lock = threading.RLock()     # lazily built parts (ELL pattern, AMG hierarchy): one builder at a time

        self._nodes_host = nodes
        self._lumped_mass = None
        # the general ELL path is built lazily (`ensure_ell`) for chains and lattices, eagerly for everything else
        self._elements = elements
        self._ell_ready = False
        self.levels = []
        self.n_bc_interior = 0
        self.closed_boundary = False
        self.regular_cells = False
        # --- 1D chain fast path ---------------------------------------------------------
        self.is_lattice = False
        self.is_chain = bool(self.dim == 1 and self.n == self.m + 1
                             and np.array_equal(elements[:, 0], np.arange(self.m))
                             and np.array_equal(elements[:, 1], np.arange(1, self.m + 1)))
        if self.is_chain:
            seg = chain_segments(self.n, is_bc)
            self.n_seg = len(seg)
            self.max_seg_len = int((seg[:, 1] - seg[:, 0]).max())   # elements of the longest segment: kernel choice
            self.seg = dev(seg)
            self.x = self.coords[0].contiguous()
            return

        # --- lattice fast path: symmetric diagonals + multigrid hierarchy --------------------
        lat = detect_lattice(elements, self.n) if self.dim == 2 else None
        self.is_lattice = lat is not None
        if self.is_lattice:
            nx, ny = lat
            nodes2d = nodes.reshape(ny + 1, nx + 1, 2)
            bc2d = is_bc.reshape(ny + 1, nx + 1)
            # Dirichlet nodes strictly inside the lattice: the geometric hierarchy only keeps those that fall on
            # coarse nodes, so many of them (pinned regions, holes) weaken the V-cycle -- the solver then routes
            # the mesh to the aggregation-multigrid path, whose Galerkin operators see every one of them
            self.n_bc_interior = int(bc2d[1:-1, 1:-1].sum())
            # every node of the four edges is a Dirichlet node (what FEMesh.rectangle produces): the regime in which
            # the energy norm controls nodal values and a scalar kappa per sample may stay factored
            self.closed_boundary = bool(bc2d[0, :].all() and bc2d[-1, :].all() and bc2d[:, 0].all() and bc2d[:, -1].all())
            # cells close to square and axis-aligned (what FEMesh.rectangle on a near-square domain gives): where the
            # multigrid cycle converges at its textbook rate and the CG may take its step LENGTH from an fp32 stencil
            # (cgstep2_kernel; the solver also asks for a full hierarchy: where multigrid converges slowly an inexact
            # step length costs iterations)
            dx = np.abs(np.diff(nodes2d[:, :, 0], axis=1))
            dy = np.abs(np.diff(nodes2d[:, :, 1], axis=0))
            skew = max(float(np.abs(np.diff(nodes2d[:, :, 1], axis=1)).max(initial=0.0)),
                       float(np.abs(np.diff(nodes2d[:, :, 0], axis=0)).max(initial=0.0)))
            hmin, hmax = min(float(dx.min()), float(dy.min())), max(float(dx.max()), float(dy.max()))
            self.regular_cells = bool(hmin > 0 and hmax <= 2.0 * hmin and skew <= 1e-12 * hmax)
            while len(self.levels) < 16:
                self.levels.append(LatticeLevel(nodes2d, bc2d, device, with_load_matrix=not self.levels))
                step = coarsening_step(nodes2d)
                if step is None:
                    break
                nodes2d, bc2d = nodes2d[::step[0], ::step[1]], bc2d[::step[0], ::step[1]]

        # --- general ELL path: built eagerly for general meshes, lazily for lattice meshes (only
        # method="ell" needs it there; the pattern build costs ~20 s of numpy at 1024^2) ------------
        if not self.is_lattice:
            self._ensure_ell()

    def bc_index(self) -> torch.Tensor:
        """Device int64 indices of the Dirichlet nodes."""
        if self._bc_index is None:
            self._bc_index = torch.nonzero(self.is_bc).reshape(-1)
        return self._bc_index

    _WARM_MAX = 4

    def warm_get(self, key):
        with self._lock:
            x = self.warm.get(key)
            if x is not None:
                self.warm.move_to_end(key)
            return x

    def warm_put(self, key, x):
        with self._lock:
            self.warm[key] = x
            self.warm.move_to_end(key)
            while len(self.warm) > self._WARM_MAX:
                self.warm.popitem(last=False)

    def shared_fp32(self, vals, cacheable: bool):
        """(fp32 copies, fp32 reciprocal main diagonals) of the batch-shared per-level matrices `vals` (each (nd, n, 1)):
        what the two-samples-per-lane strip kernels read.  cacheable: `vals` is the unit-kappa operator of the mesh
        (plan-constant), so the copies are made once."""
        key = len(vals)
        cache = self.__dict__.setdefault("_fp32_cache", {})
        if cacheable and key in cache:
            return cache[key]
        v32 = [v.float() for v in vals]
        rd32 = [(1.0 / v[0]).float().contiguous() for v in vals]
        if cacheable:
            with self._lock:
                cache[key] = (v32, rd32)
        return v32, rd32

    def dense_level(self, max_nodes: int = 1200):
        """Index of the first level with at most `max_nodes` nodes (33 x 33 for power-of-two meshes), or None: the
        level whose solve a dense inverse replaces for factored operators.  0 = the mesh itself is that small."""
        return next((i for i, lev in enumerate(self.levels) if lev.n <= max_nodes), None)

    def dense_coarse(self, idx: int, vals, fp32: bool):
        """(level index, dense inverse) for the factored lattice operator.  `vals` are the UNIT-kappa symmetric
        diagonals of the levels (plan-constant, so the inverse is built once and cached): level `idx` is inverted on
        the host -- identity rows stay identity -- and uploaded in the V-cycle's storage type (fp64 for idx 0)."""
        key = (idx, bool(fp32))
        cache = self.__dict__.setdefault("_dense_cache", {})
        with self._lock:
            self._dense_coarse_build(cache, key, idx, vals, fp32)
        return idx, cache[key]

    def _dense_coarse_build(self, cache, key, idx, vals, fp32):
        if key not in cache:
            lev = self.levels[idx]
            d = vals[idx].detach().to("cpu", torch.float64).numpy().reshape(lev.nd, lev.n)
            n, W = lev.n, lev.nx + 1
            K = np.zeros((n, n))
            i = np.arange(n)
            K[i, i] = d[0]
            for k, off in ((1, 1), (2, W)) + (((3, lev.nx),) if lev.nd == 4 else ()):
                j = i[: n - off]
                K[j, j + off] = d[k][: n - off]
                K[j + off, j] = d[k][: n - off]
            inv = np.linalg.inv(K)
            inv = 0.5 * (inv + inv.T)                        # symmetric to the last bit: the cycle stays symmetric
            cache[key] = torch.from_numpy(inv.astype(np.float32) if fp32 else inv).to(self.device).contiguous()

    def lumped_mass(self) -> torch.Tensor:
        """(n,) lumped P1 mass m_i = sum over the elements at node i of |e| / (dim + 1): the row sums of the load
        matrix of solver.py:95-96 / :143-145.  The reaction term c * u and the heat equation's time derivative use it."""
        if self._lumped_mass is None:
            self._lumped_mass = torch.from_numpy(_lumped_mass(self._nodes_host, self._elements)).to(self.device)
        return self._lumped_mass

    def ensure_ell(self):
        """ELL pattern, gather lists, element integrals and the ELL load matrix of the general path."""
        with self._build_lock:
            self._ensure_ell()

    def _ensure_ell(self):
        if self._ell_ready:
            return
        L = _hip.lib()
        device = self.device
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        pat = build_ell_pattern(self._elements, self.n)
        self.W = pat["W"]
        self.cols = dev(pat["cols"])
        self.ent_ptr = dev(pat["ent_ptr"])
        self.contrib = dev(pat["contrib"])
        self.slot_of = dev(pat["slot_of"])
        stream = _stream(device)
        nloc = self.npe * self.npe
        if self.is_p2:      # quadratic triangles (ours): batch-shared integrals from the host, plain kappa * k0 assembly
            k0h, m0h = p2_element_integrals(self._nodes_host, self._elements)
            self.k0, self.m0 = dev(k0h), dev(m0h)
            self.tnum = self.den = None
        else:
            self.k0 = torch.empty((nloc, self.m), dtype=torch.float64, device=device)
            self.m0 = torch.empty((nloc, self.m), dtype=torch.float64, device=device)
            _hip.check(L.diffhe_p1_element_integrals(_hip.ptr(self.coords), _hip.ptr(self.elems), self.dim, self.n,
                                                     self.m, _hip.ptr(self.k0), _hip.ptr(self.m0), stream),
                       "diffhe_p1_element_integrals")
            tn, dn = reference_order_integrals(self.coords.cpu().numpy(), self.elems.cpu().numpy())
            self.tnum, self.den = dev(tn), dev(dn)       # reference-order assembly
        # load matrix M (batch-shared ELL values): F = M f, df = M^T lambda
        self.Mvals = torch.empty((self.W, self.n), dtype=torch.float64, device=device)
        _hip.check(L.diffhe_ell_assemble_rows(_hip.ptr(self.m0), None, 0, 0, _hip.ptr(self.ent_ptr),
                                              _hip.ptr(self.contrib), _hip.ptr(self.cols), None, None, None,
                                              _hip.ptr(self.Mvals), None, self.n, self.m, self.W, 1, stream),
                   "diffhe_ell_assemble_rows(M)")
        self._ell_ready = True

    def closed_boundary_general(self) -> bool:
        """Every node of the mesh boundary is a Dirichlet node (general meshes, P1: a boundary edge belongs to exactly one
        triangle; 1D: a boundary node to exactly one element).  The regime in which one scalar kappa per sample may stay
        FACTORED, K_b = kappa_b K_1 (closed lattices have `closed_boundary`): with Neumann parts the system is
        ill-conditioned enough for the last-bit difference to the reference's rounded matrix to show.  Cached."""
        cached = self.__dict__.get("_closed_general")
        if cached is not None:
            return cached
        closed = False
        if not self.is_p2:
            el = self._elements
            is_bc = self.is_bc.cpu().numpy().astype(bool)
            if self.dim == 1:
                deg = np.bincount(el.reshape(-1), minlength=self.n)
                bnodes = np.nonzero(deg == 1)[0]
            else:
                e = np.concatenate([el[:, [0, 1]], el[:, [1, 2]], el[:, [2, 0]]])
                e.sort(axis=1)
                key = e[:, 0].astype(np.int64) * self.n + e[:, 1]
                uniq, cnt = np.unique(key, return_counts=True)
                bkeys = uniq[cnt == 1]
                bnodes = np.unique(np.concatenate([bkeys // self.n, bkeys % self.n]))
            closed = bool(len(bnodes) > 0 and is_bc[bnodes].all())
        self.__dict__["_closed_general"] = closed
        return closed

    def unit_amg(self, key, build):
        """The aggregation-multigrid hierarchy (Galerkin coarse operators included) of the UNIT-kappa operator: plan-constant,
        built once per (smoothed, fp32) by `build()` and shared by every factored solve on this mesh."""
        cache = self.__dict__.setdefault("_unit_amg", {})
        with self._lock:
            if key not in cache:
                cache[key] = build()
            return cache[key]

    def ensure_amg(self, smoothed: bool = False):
        """Aggregation hierarchy of the general path (diffhe/amg.py), uploaded once per mesh.  smoothed: the
        smoothed-aggregation hierarchy (batch-shared P from the unit-kappa operator, weighted Galerkin lists) instead
        of the piecewise-constant one.  Returns the list of device-side level dicts."""
        with self._build_lock:
            return self._ensure_amg(smoothed)

    def _ensure_amg(self, smoothed: bool = False):
        attr = "amg_levels_sa" if smoothed else "amg_levels"
        if getattr(self, attr, None) is not None:
            return getattr(self, attr)
        from .amg import build_hierarchy, build_hierarchy_sa
        self._ensure_ell()
        device = self.device
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
        if smoothed:
            host = build_hierarchy_sa(self.cols.cpu().numpy(), self._unit_ell_values(), self.is_bc.cpu().numpy())
        else:
            host = build_hierarchy(self.cols.cpu().numpy(), self.is_bc.cpu().numpy())
        levels = []
        for lv in host:
            d = dict(n=lv["n"], W=lv["W"], cols=dev(lv["cols"]), ent_ptr=dev(lv["ent_ptr"]), contrib=dev(lv["contrib"]),
                     agg=dev(lv["agg"]), agg_ptr=dev(lv["agg_ptr"]), agg_members=dev(lv["agg_members"]))
            if smoothed:
                d.update(weights=dev(lv["weights"]), agg_weights=dev(lv["agg_weights"]), p_cols=dev(lv["p_cols"]),
                         p_vals=dev(lv["p_vals"]))
            levels.append(d)
        setattr(self, attr, levels)
        return levels

    def _unit_ell_values(self) -> np.ndarray:
        """(W, n) host copy of the Dirichlet-eliminated UNIT-kappa matrix in the ELL pattern (identity rows on Dirichlet
        nodes): what the smoothed prolongation is built from.  Assembled by the device kernels the solve itself uses."""
        L = _hip.lib()
        vals = torch.empty((self.W, self.n, 1), dtype=torch.float64, device=self.device)
        lift = torch.empty((self.n, 1), dtype=torch.float64, device=self.device)
        one = torch.ones(1, dtype=torch.float64, device=self.device)
        st = _stream(self.device)
        if self.is_p2:
            _hip.check(L.diffhe_ell_assemble_rows(_hip.ptr(self.k0), _hip.ptr(one), 0, 0, _hip.ptr(self.ent_ptr),
                                                  _hip.ptr(self.contrib), _hip.ptr(self.cols), None, _hip.ptr(self.is_bc),
                                                  _hip.ptr(self.g), _hip.ptr(vals), _hip.ptr(lift), self.n, self.m, self.W, 1,
                                                  st), "diffhe_ell_assemble_rows(unit)")
        else:
            _hip.check(L.diffhe_ell_assemble_rows_ref(_hip.ptr(self.tnum), _hip.ptr(self.den), _hip.ptr(one), 0, 0,
                                                      _hip.ptr(self.ent_ptr), _hip.ptr(self.contrib), _hip.ptr(self.cols), None,
                                                      _hip.ptr(self.is_bc), _hip.ptr(self.g), _hip.ptr(vals), _hip.ptr(lift),
                                                      self.n, self.m, self.W, 1, st), "diffhe_ell_assemble_rows_ref(unit)")
        return vals.reshape(self.W, self.n).cpu().numpy()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _fingerprint(mesh):
    bc = mesh.dirichlet_nodes
    return (id(mesh.nodes), mesh.nodes._version, id(mesh.elements), mesh.elements._version, len(bc),
            hash(tuple(bc.items())) if len(bc) <= 1 << 16 else (hash(tuple(bc.keys())), hash(tuple(bc.values()))))


_PLAN_LOCK = threading.Lock()


def get_plan(mesh, device: torch.device) -> SolvePlan:
    """Cached plan for (mesh, device); rebuilt when nodes/elements/BCs change.  One builder at a time: two threads
    that meet on a new mesh get the same plan."""
    key = (str(device), _fingerprint(mesh))
    with _PLAN_LOCK:
        cache = mesh.__dict__.setdefault("_diffhe_plans", {})
        plan: Optional[SolvePlan] = cache.get(key)
        if plan is None:
            cache.clear()
            plan = SolvePlan(mesh, device)
            cache[key] = plan
    return plan

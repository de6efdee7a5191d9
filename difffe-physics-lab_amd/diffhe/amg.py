"""Aggregation hierarchy for general (unstructured) P1 meshes -- host side.

The general ELL path has no geometric hierarchy to lean on, so its multigrid preconditioner is
algebraic: nodes are grouped into aggregates (roots = a distance-2 maximal independent set of the
mesh graph, every other node joins an adjacent aggregate), the prolongation is piecewise constant,
and the coarse operators are the Galerkin products P^T A P -- which for piecewise-constant P are
plain sums of fine-matrix entries, so they are rebuilt PER SAMPLE on the device from gather lists
(`diffhe_ell_galerkin`) while the aggregates and patterns, computed here once per mesh with
vectorised numpy, are shared by the whole batch.  Dirichlet nodes belong to no aggregate (their
rows are identity rows and their residual is zero).

Nothing here has a counterpart in the reference (it solves with a dense LU, solver.py:174).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


def aggregate(cols: np.ndarray, active: np.ndarray, seed: int = 0) -> np.ndarray:
    """Node -> aggregate id (or -1 for inactive nodes) for the graph given by ELL columns
    `cols` (W, n) (slot 0 = the node itself, unused slots point at the node itself).

    Roots are picked as strict priority maxima within graph distance 2 (Luby rounds), so two
    roots are never adjacent or share a neighbour; each remaining node joins the aggregate of an
    adjacent root, else of an adjacent already-aggregated node; isolated leftovers become singletons.
    Deterministic for a given seed.
    """
    W, n = cols.shape
    rng = np.random.default_rng(seed)
    prio = rng.permutation(n).astype(np.float64) + 1.0
    prio[~active] = -np.inf
    state = np.zeros(n, dtype=np.int8)            # 0 undecided, 1 root, 2 within distance 2 of a root
    state[~active] = 2
    idx_all = np.arange(n)
    while True:
        cand = state == 0
        if not cand.any():
            break
        pc = np.where(cand, prio, -np.inf)
        m1 = pc[cols].max(axis=0)                  # max over the closed neighbourhood
        m2 = m1[cols].max(axis=0)                  # ... over distance 2
        new_root = cand & (pc >= m2)
        state[new_root] = 1
        near1 = new_root[cols].any(axis=0)
        near2 = near1[cols].any(axis=0)
        state[near2 & (state == 0)] = 2
    roots = np.nonzero(state == 1)[0]
    agg = np.full(n, -1, dtype=np.int64)
    agg[roots] = np.arange(len(roots))
    # distance-1 neighbours of a root join it (a node is adjacent to at most one root)
    nb = agg[cols]                                 # (W, n) aggregate of each neighbour, -1 if none
    join = nb.max(axis=0)
    take = active & (agg < 0) & (join >= 0)
    agg[take] = join[take]
    # the rest (distance 2 from every root) joins any adjacent aggregate; repeat until stable
    for _ in range(8):
        left = active & (agg < 0)
        if not left.any():
            break
        join = agg[cols].max(axis=0)
        take = left & (join >= 0)
        if not take.any():
            break
        agg[take] = join[take]
    left = np.nonzero(active & (agg < 0))[0]       # disconnected leftovers: singletons
    if len(left):
        base = int(agg.max()) + 1
        agg[left] = base + np.arange(len(left))
    del idx_all
    return agg


def coarse_pattern(cols: np.ndarray, agg: np.ndarray):
    """Galerkin gather lists for piecewise-constant aggregation.

    Returns dict(n, W, cols (W,n) i32, ent_ptr (W*n+1) i32, contrib i32) for the coarse level:
    coarse entry (I, slot) at slot*n + I sums the fine entries listed in contrib[ent_ptr[.]:ent_ptr[.+1]]
    (fine entry index = k*n_fine + i).  Slot 0 is the diagonal."""
    Wf, nf = cols.shape
    nc = int(agg.max()) + 1
    k_idx, i_idx = np.meshgrid(np.arange(Wf, dtype=np.int64), np.arange(nf, dtype=np.int64), indexing="ij")
    j_idx = cols.astype(np.int64)
    real = (k_idx == 0) | (j_idx != i_idx)         # drop the padding slots (they point at the row itself)
    I = agg[i_idx]
    J = agg[j_idx]
    keep = real & (I >= 0) & (J >= 0)
    I, J = I[keep], J[keep]
    fine_entry = (k_idx * nf + i_idx)[keep]
    # coarse entries: unique (I, J), diagonal first within each row
    rows_all = np.concatenate([np.arange(nc, dtype=np.int64), I])
    cols_all = np.concatenate([np.arange(nc, dtype=np.int64), J])
    offd = (rows_all != cols_all).astype(np.int64)
    key = (rows_all * 2 + offd) * nc + cols_all
    order = np.argsort(key, kind="stable")
    ks = key[order]
    new = np.ones(len(ks), dtype=bool)
    new[1:] = ks[1:] != ks[:-1]
    entry_id = np.cumsum(new) - 1
    ent_rows = rows_all[order][new]
    ent_cols = cols_all[order][new]
    row_start = np.searchsorted(ent_rows, np.arange(nc))
    slot = np.arange(len(ent_rows)) - row_start[ent_rows]
    W = int(slot.max()) + 1
    ccols = np.tile(np.arange(nc, dtype=np.int32), W)
    cidx = slot * nc + ent_rows
    ccols[cidx] = ent_cols.astype(np.int32)
    inv = np.empty(len(order), dtype=np.int64)
    inv[order] = np.arange(len(order))
    contrib_entry = cidx[entry_id[inv[nc:]]]
    corder = np.argsort(contrib_entry, kind="stable")
    contrib = fine_entry[corder]
    ent_ptr = np.zeros(W * nc + 1, dtype=np.int64)
    np.cumsum(np.bincount(contrib_entry, minlength=W * nc), out=ent_ptr[1:])
    if ent_ptr[-1] >= 2 ** 31 or Wf * nf >= 2 ** 31:
        raise ValueError("mesh too large for int32 gather lists")
    return dict(n=nc, W=W, cols=ccols.reshape(W, nc), ent_ptr=ent_ptr.astype(np.int32), contrib=contrib.astype(np.int32))


def members_csr(agg: np.ndarray, nc: int):
    """CSR of aggregate members: (ptr (nc+1) i32, members i32 sorted by aggregate then node id)."""
    act = np.nonzero(agg >= 0)[0]
    order = np.argsort(agg[act], kind="stable")
    members = act[order].astype(np.int32)
    ptr = np.zeros(nc + 1, dtype=np.int64)
    np.cumsum(np.bincount(agg[act], minlength=nc), out=ptr[1:])
    return ptr.astype(np.int32), members


def build_hierarchy(cols: np.ndarray, is_bc: np.ndarray, min_coarse: int = 64, max_levels: int = 12) -> List[Dict]:
    """Levels below the fine one: each dict holds the aggregation of the PREVIOUS level (agg, agg_ptr,
    agg_members) and the pattern / Galerkin lists of this level (n, W, cols, ent_ptr, contrib)."""
    levels: List[Dict] = []
    active = ~is_bc.astype(bool)
    cur_cols = cols
    while len(levels) < max_levels - 1:
        n_active = int(active.sum())
        if n_active <= min_coarse:
            break
        agg = aggregate(cur_cols, active, seed=len(levels))
        nc = int(agg.max()) + 1
        if nc >= 0.7 * n_active:                   # not coarsening any more (e.g. disconnected graph)
            break
        pat = coarse_pattern(cur_cols, agg)
        ptr, members = members_csr(agg, nc)
        pat.update(agg=agg.astype(np.int32), agg_ptr=ptr, agg_members=members)
        levels.append(pat)
        cur_cols = pat["cols"]
        active = np.ones(nc, dtype=bool)
    return levels


# ---------------------------------------------------------------------------------------------------------------------
# Smoothed aggregation (round 3): same aggregates, prolongation smoothed by one damped-Jacobi step of the UNIT-kappa
# operator, P = (I - omega D_1^-1 A_1) P_0.  P is batch-shared (built once per mesh from A_1); the coarse operators stay
# per sample, A_c^b = P^T A_b P, as WEIGHTED sums of fine entries (gather lists with weights P_iI P_jJ).  Halves the
# iteration count of the piecewise-constant hierarchy (256^2, CPU prototype: 68 -> 34) for ~1.6x denser coarse levels.
# ---------------------------------------------------------------------------------------------------------------------
def _ell_to_csr(cols: np.ndarray, vals: np.ndarray):
    """(csr matrix, ELL index k*n+i of every stored nonzero in csr order) from an ELL pattern (slot 0 = diagonal,
    padding slots point at the row itself)."""
    import scipy.sparse as sp
    W, n = cols.shape
    k_idx, i_idx = np.meshgrid(np.arange(W, dtype=np.int64), np.arange(n, dtype=np.int64), indexing="ij")
    real = (k_idx == 0) | (cols != i_idx)
    r, c, e = i_idx[real], cols[real].astype(np.int64), (k_idx * n + i_idx)[real]
    order = np.lexsort((c, r))
    A = sp.csr_matrix((vals[real][order], (r[order], c[order])), shape=(n, n))
    A.sum_duplicates()
    assert A.nnz == len(order), "duplicate ELL entries"
    return A, e[order]


def _csr_to_ell(A):
    """ELL pattern (W, n) with slot 0 = diagonal for a square CSR matrix with a full diagonal; also the slot index k*n+i of
    every CSR nonzero (in CSR order)."""
    n = A.shape[0]
    A = A.tocsr()
    A.sort_indices()
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(A.indptr))
    colsA = A.indices.astype(np.int64)
    isd = colsA == rows
    # rank within the row with the diagonal first
    pos = np.arange(A.nnz) - A.indptr[rows]
    dpos = np.zeros(n, dtype=np.int64)
    dpos[rows[isd]] = pos[isd]
    slot = np.where(isd, 0, np.where(pos < dpos[rows], pos + 1, pos))
    W = int(slot.max()) + 1
    ell = np.tile(np.arange(n, dtype=np.int32), W).reshape(W, n)
    ell[slot, rows] = colsA.astype(np.int32)
    return ell, slot * n + rows


def smoothed_level(A1, ell_index_f, cols_f, active, seed, omega=2.0 / 3.0, drop=0.0):
    """One coarsening step.  A1: unit-kappa CSR matrix of the fine level; ell_index_f: ELL entry index (k*n_f + i) of each
    of its nonzeros (CSR order); cols_f: its ELL pattern (for the aggregation).  Returns the level dict (device arrays as
    numpy) and the coarse unit matrix with ITS ell index."""
    import scipy.sparse as sp
    nf = A1.shape[0]
    agg = aggregate(cols_f, active, seed=seed)
    nc = int(agg.max()) + 1
    rows = np.nonzero(agg >= 0)[0]
    P0 = sp.csr_matrix((np.ones(len(rows)), (rows, agg[rows])), shape=(nf, nc))
    d = A1.diagonal()
    S = sp.identity(nf, format="csr") - sp.diags(omega / d) @ A1
    P = (sp.diags((agg >= 0).astype(np.float64)) @ S @ P0).tocsr()       # inactive (Dirichlet) rows interpolate nothing
    if drop > 0.0:
        P.data[np.abs(P.data) < drop] = 0.0
    P.eliminate_zeros()
    P.sort_indices()
    # --- weighted Galerkin gather lists: coarse entry (I, J) = sum over fine nonzeros (i, j) of P_iI a_ij P_jJ.  The
    # coarse PATTERN is the set of (I, J) these lists reach (structural: a fine entry whose unit value happens to be
    # exactly 0 -- the quad diagonal of a right-angled lattice -- still carries a per-sample value) ---
    fi = np.repeat(np.arange(nf, dtype=np.int64), np.diff(A1.indptr))
    fj = A1.indices.astype(np.int64)
    pn = np.diff(P.indptr).astype(np.int64)
    keep = (pn[fi] > 0) & (pn[fj] > 0)
    fi, fj, fe, fa = fi[keep], fj[keep], ell_index_f[keep], A1.data[keep]
    ca = pn[fi]                                                    # expand over the nonzeros of P's row i ...
    t1 = np.repeat(np.arange(len(fi)), ca)
    a_loc = np.arange(len(t1)) - np.repeat(np.cumsum(ca) - ca, ca)
    pa = P.indptr[fi[t1]] + a_loc
    cb = pn[fj[t1]]                                                # ... then over those of row j
    t2 = np.repeat(np.arange(len(t1)), cb)
    b_loc = np.arange(len(t2)) - np.repeat(np.cumsum(cb) - cb, cb)
    pb = P.indptr[fj[t1[t2]]] + b_loc
    I, J = P.indices[pa[t2]].astype(np.int64), P.indices[pb].astype(np.int64)
    w = P.data[pa[t2]] * P.data[pb]
    fine_entry = fe[t1[t2]]
    # coarse unit matrix P^T A_1 P on that pattern (+ a structural diagonal for every coarse node)
    keys = np.concatenate([I * nc + J, np.arange(nc, dtype=np.int64) * (nc + 1)])
    data = np.concatenate([w * fa[t1[t2]], np.zeros(nc)])
    ckey, inv = np.unique(keys, return_inverse=True)
    cval = np.zeros(len(ckey))
    np.add.at(cval, inv, data)
    crow_u = ckey // nc
    indptr = np.zeros(nc + 1, dtype=np.int64)
    np.cumsum(np.bincount(crow_u, minlength=nc), out=indptr[1:])
    Ac = sp.csr_matrix((cval, (ckey % nc).astype(np.int32), indptr), shape=(nc, nc))
    cols_c, ell_index_c = _csr_to_ell(Ac)
    Wc = cols_c.shape[0]
    loc = inv[:len(I)]
    centry = ell_index_c[loc]
    order = np.argsort(centry, kind="stable")
    ent_ptr = np.zeros(Wc * nc + 1, dtype=np.int64)
    np.cumsum(np.bincount(centry, minlength=Wc * nc), out=ent_ptr[1:])
    if ent_ptr[-1] >= 2 ** 31:
        raise ValueError("mesh too large for int32 gather lists")
    # --- transfer operators for the device: P as ELL rows (prolongation), P^T as CSR (restriction) ---
    pw = int(pn.max()) if len(pn) else 1
    p_cols = np.full((pw, nf), -1, dtype=np.int32)
    p_vals = np.zeros((pw, nf), dtype=np.float64)
    prow = np.repeat(np.arange(nf, dtype=np.int64), pn)
    ploc = np.arange(P.nnz) - P.indptr[prow]
    p_cols[ploc, prow] = P.indices
    p_vals[ploc, prow] = P.data
    PT = P.T.tocsr()
    PT.sort_indices()
    level = dict(n=nc, W=Wc, cols=cols_c, ent_ptr=ent_ptr.astype(np.int32), contrib=fine_entry[order].astype(np.int32),
                 weights=w[order], agg=agg.astype(np.int32), agg_ptr=PT.indptr.astype(np.int32),
                 agg_members=PT.indices.astype(np.int32), agg_weights=PT.data.astype(np.float64),
                 p_cols=p_cols, p_vals=p_vals)
    return level, Ac, ell_index_c


def build_hierarchy_sa(cols: np.ndarray, unit_vals: np.ndarray, is_bc: np.ndarray, min_coarse: int = 64,
                       max_levels: int = 12) -> List[Dict]:
    """Smoothed-aggregation levels below the fine one.  cols (W, n), unit_vals (W, n): ELL pattern and UNIT-kappa values
    of the Dirichlet-eliminated fine matrix (identity rows on Dirichlet nodes).  Each dict: the coarse pattern and
    weighted Galerkin lists (n, W, cols, ent_ptr, contrib, weights) and the transfers from the level above (p_cols,
    p_vals: P as ELL rows; agg_ptr, agg_members, agg_weights: P^T as CSR; agg: the underlying aggregates)."""
    levels: List[Dict] = []
    A1, ell_idx = _ell_to_csr(cols, unit_vals)
    active = ~is_bc.astype(bool)
    cur_cols = cols
    while len(levels) < max_levels - 1:
        n_active = int(active.sum())
        if n_active <= min_coarse:
            break
        lev, Ac, ell_c = smoothed_level(A1, ell_idx, cur_cols, active, seed=len(levels))
        if lev["n"] >= 0.7 * n_active:
            break
        levels.append(lev)
        A1, ell_idx, cur_cols = Ac, ell_c, lev["cols"]
        active = np.ones(lev["n"], dtype=bool)
    return levels

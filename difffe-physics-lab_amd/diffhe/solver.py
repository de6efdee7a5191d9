"""Differentiable P1-FEM solve of -div(kappa grad u) = f with Dirichlet data, on MI355X.

Mirror of the reference operator `DifferentiableFESolver` (reference
diffhe/solver.py:21-183): same constructor, `.kappa`, `forward(f) -> u`, same
unbatched semantics (float64 output, Dirichlet values exact, gradients to kappa
and f).  The work is done by hand-written HIP kernels behind the C ABI of
include/diffhe_hip.h; the adjoint is explicit, not autograd
replay.  There is no CPU fallback: without the HIP library or a GPU, `forward`
raises.

Extensions over the reference (which has no batch dimension, solver.py:54):
  f      (n,) | (n,1) | (B,n)
  kappa  python scalar | 0-dim / 1-element tensor | (m,) per element |
         (B,1) or (B,) per sample | (B,m) per sample and element
"""
from __future__ import annotations

import ctypes
import itertools
import os
import types
import warnings
import weakref
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _hip
from .mesh import FEMesh
from .plan import SolvePlan, get_plan, padded_batch, status_buffer, _stream

# kappa layouts
K_SCALAR, K_SAMPLE, K_ELEM, K_SAMPLE_ELEM = 0, 1, 2, 3


@dataclass
class SolveInfo:
    """Diagnostics of the last solve (the reference silently returns garbage on
    failure; we surface it instead)."""
    path: str = ""
    iterations: int = 0
    not_converged: int = 0
    max_relres: float = 0.0
    adj_iterations: int = 0
    adj_max_relres: float = 0.0
    err_est: float = 0.0        # lattice path: max over samples of the estimated relative energy-norm error
    adj_err_est: float = 0.0
    # lattice path: how many samples each rule stopped, forward / adjoint: {"residual": ., "energy": ., "cap": .}
    # ("cap" = neither rule fired before the iteration cap; the direct dense path iterates nothing and reports {})
    stop_rules: Optional[dict] = None
    adj_stop_rules: Optional[dict] = None
    tol_energy: float = 0.0     # the energy-norm tolerance in force for this call (0: residual rule alone)
    # lattice path, fp32-stored V-cycle: where its coefficients come from -- "shared-fp32" (batch-shared matrix, scalar
    # loads), "fp16-rowsum" (per-sample matrices: fp32 diagonal + scaled fp16 couplings), "fp32" (per-sample plain fp32
    # copies: asked for, or the fallback when a sample's couplings span more than fp16 holds), "fp64" (no copies)
    coeff_storage: str = ""
    factored: bool = False      # one scalar kappa per sample (or for all) kept as K_b = kappa_b K_1: ONE unit matrix for the batch
    flags: int = 0              # lattice path: the `precond_fp32` word handed to diffhe_lattice_pcg_solve (include/diffhe_hip.h)
    precision: str = ""         # what is stored / computed in which precision in THIS solve, derived from those flags


def _resolve_device(device) -> torch.device:
    if device is not None:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("diffhe runs on a ROCm GPU only (no CPU fallback)")
        return device
    if not torch.cuda.is_available():
        raise RuntimeError("diffhe: no ROCm GPU visible -- the HIP solve path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _kappa_mode(kappa: torch.Tensor, m: int, B: Optional[int]):
    """Classify kappa -> (mode, B implied or None)."""
    if kappa.numel() == 1:
        return K_SCALAR, None
    if kappa.dim() == 1:
        if kappa.shape[0] == m and (B is None or B == 1 or B != m):
            return K_ELEM, None
        return K_SAMPLE, kappa.shape[0]
    if kappa.dim() == 2:
        if kappa.shape[1] == 1:
            return K_SAMPLE, kappa.shape[0]
        if kappa.shape[1] == m:
            return K_SAMPLE_ELEM, kappa.shape[0]
    raise ValueError(f"kappa shape {tuple(kappa.shape)} not understood for a mesh with {m} elements")


class _Engine:
    """Thin, stateless driver of the C ABI for one plan and one batch geometry."""

    def __init__(self, plan: SolvePlan, tol: float, max_iter: int, check_every: int, assembly: str):
        self.p = plan
        self.tol, self.max_iter, self.check_every, self.assembly = tol, max_iter, check_every, assembly
        self.ref_order = False      # per-sample lattice matrices in the reference's exact operation order (operator="assembled")
        self.L = _hip.lib()

    # -- layout helpers -----------------------------------------------------------------
    def to_node_major(self, src, B, Bp, n, zero_mask=None):
        """(B, n) rows -- or one (n,) vector broadcast to all B samples -- to (n, Bp)."""
        p = self.p
        dst = torch.empty((n, Bp), dtype=torch.float64, device=p.device)
        ld = 0 if src.dim() == 1 else src.stride(0)
        _hip.check(self.L.diffhe_to_node_major(_hip.ptr(src), ld, _hip.ptr(zero_mask), _hip.ptr(dst), n, B, Bp,
                                               _stream(p.device)), "diffhe_to_node_major")
        return dst

    def to_sample_major(self, src, B, Bp, n, add=None):
        p = self.p
        dst = torch.empty((B, n), dtype=torch.float64, device=p.device)
        _hip.check(self.L.diffhe_to_sample_major(_hip.ptr(src), _hip.ptr(add), _hip.ptr(dst), n, n, B, Bp,
                                                 _stream(p.device)), "diffhe_to_sample_major")
        return dst

    def field_node_major(self, k, B, Bp, em):
        """(m, Bp) per-sample kappa fields (padding samples = 1) from the API's (B, m) -- one transposing pass -- or from
        an element-major (m, B) tensor (layout='node'): used as it is when B needs no padding."""
        p = self.p
        if em:
            if Bp == B and k.is_contiguous():
                return k
            kp = torch.ones((p.m, Bp), dtype=torch.float64, device=p.device)
            kp[:, :B] = k
            return kp
        kp = self.to_node_major(k.reshape(B, p.m).contiguous(), B, Bp, p.m)
        if Bp > B:
            kp[:, B:] = 1.0
        return kp

    # -- general path ---------------------------------------------------------------------
    def kappa_device(self, kappa, mode, B, Bp, em=False):
        """-> (tensor, stride_e, stride_b, Bv) in the layout the kernels index."""
        p = self.p
        k = kappa.detach().to(p.device, torch.float64)
        if mode == K_SCALAR:
            return k.reshape(1).contiguous(), 0, 0, 1
        if mode == K_ELEM:
            return k.reshape(p.m).contiguous(), 1, 0, 1
        if mode == K_SAMPLE:
            kp = torch.ones(Bp, dtype=torch.float64, device=p.device)
            kp[:B] = k.reshape(B)
            return kp, 0, 1, Bp
        return self.field_node_major(k, B, Bp, em), Bp, 1, Bp

    def assemble(self, kdev, kse, ksb, Bv):
        p, L = self.p, self.L
        st = _stream(p.device)
        vals = torch.empty((p.W, p.n, Bv), dtype=torch.float64, device=p.device)
        lift = torch.empty((p.n, Bv), dtype=torch.float64, device=p.device)
        if p.is_p2:
            # quadratic triangles (ours; no reference operator to be bit-identical with): kappa * k0, gathered
            _hip.check(L.diffhe_ell_assemble_rows(_hip.ptr(p.k0), _hip.ptr(kdev), kse, ksb, _hip.ptr(p.ent_ptr),
                                                  _hip.ptr(p.contrib), _hip.ptr(p.cols), None, _hip.ptr(p.is_bc),
                                                  _hip.ptr(p.g), _hip.ptr(vals), _hip.ptr(lift), p.n, p.m, p.W, Bv, st),
                       "diffhe_ell_assemble_rows(P2)")
        elif self.assembly == "atomic" and Bv > 1:
            vals.zero_()
            _hip.check(L.diffhe_ell_assemble_atomic(_hip.ptr(p.coords), _hip.ptr(p.elems), p.dim, _hip.ptr(kdev), kse,
                                                    ksb, _hip.ptr(p.slot_of), _hip.ptr(vals), p.n, p.m, p.W, Bv, st),
                       "diffhe_ell_assemble_atomic")
            lift.zero_()  # apply_dirichlet returns F - lift: feed F = 0, negate
            _hip.check(L.diffhe_ell_apply_dirichlet(_hip.ptr(p.cols), _hip.ptr(p.is_bc), _hip.ptr(p.g),
                                                    _hip.ptr(vals), _hip.ptr(lift), p.n, p.W, Bv, st),
                       "diffhe_ell_apply_dirichlet")
            lift.neg_()
        else:
            # reference operation order: values bit-identical to the reference's K (include/diffhe_hip.h)
            _hip.check(L.diffhe_ell_assemble_rows_ref(_hip.ptr(p.tnum), _hip.ptr(p.den), _hip.ptr(kdev), kse, ksb,
                                                      _hip.ptr(p.ent_ptr), _hip.ptr(p.contrib), _hip.ptr(p.cols), None,
                                                      _hip.ptr(p.is_bc), _hip.ptr(p.g), _hip.ptr(vals), _hip.ptr(lift),
                                                      p.n, p.m, p.W, Bv, st), "diffhe_ell_assemble_rows_ref")
        return vals, lift

    def reaction_shifts(self, c, n_levels):
        """Per-level (n,) diagonal shifts c * M_L (0 on Dirichlet rows) of a FACTORED lattice operator, cached on the plan."""
        p = self.p
        with p._lock:       # solvers with different c may share the plan (and run on different threads)
            cache = p.__dict__.setdefault("_shift_cache", {})
            if c not in cache:
                while len(cache) >= 4:
                    cache.pop(next(iter(cache)))
                cache[c] = [c * torch.where(lev.is_bc.bool(), torch.zeros_like(lev.lumped_mass()), lev.lumped_mass())
                            for lev in p.levels]
            return cache[c][:n_levels]

    def add_reaction(self, vals, c, lattice):
        """A += c M_L on the free rows (M_L = lumped mass, diagonal): the stored main diagonal is slot 0 of both
        formats.  Lattice: one entry of `vals` per multigrid level, each with its own (re-discretised) lumped mass."""
        p = self.p
        for li, v in enumerate(vals):
            lev = p.levels[li] if lattice else None
            mass = lev.lumped_mass() if lattice else p.lumped_mass()
            is_bc = lev.is_bc if lattice else p.is_bc
            v[0] += (c * torch.where(is_bc.bool(), torch.zeros_like(mass), mass)).unsqueeze(1)

    def load_vector(self, f_nm, lift, Bv, Bp, lift_scale=None, lattice=False):
        """F = M f - lift_scale * lift on the free rows, 0 on Dirichlet rows."""
        p = self.p
        F = torch.empty((p.n, Bp), dtype=torch.float64, device=p.device)
        if lattice:   # load matrix stored as symmetric diagonals of level 0: no ELL pattern needed
            lev = p.levels[0]
            _hip.check(self.L.diffhe_lattice_apply_shared(lev.nx, lev.ny, 4, _hip.ptr(lev.Mvals), _hip.ptr(f_nm),
                                                          _hip.ptr(lift), Bv, _hip.ptr(lift_scale), _hip.ptr(p.is_bc),
                                                          _hip.ptr(F), Bp, _stream(p.device)),
                       "diffhe_lattice_apply_shared")
            return F
        _hip.check(self.L.diffhe_ell_spmv_shared(_hip.ptr(p.Mvals), _hip.ptr(p.cols), _hip.ptr(f_nm), _hip.ptr(lift),
                                                 Bv, _hip.ptr(lift_scale), _hip.ptr(p.is_bc), _hip.ptr(F), p.n, p.W,
                                                 Bp, _stream(p.device)), "diffhe_ell_spmv_shared")
        return F

    def apply_M(self, x_nm, Bp, lattice=False):
        p = self.p
        y = torch.empty((p.n, Bp), dtype=torch.float64, device=p.device)
        if lattice:
            lev = p.levels[0]
            _hip.check(self.L.diffhe_lattice_apply_shared(lev.nx, lev.ny, 4, _hip.ptr(lev.Mvals), _hip.ptr(x_nm), None,
                                                          1, None, None, _hip.ptr(y), Bp, _stream(p.device)),
                       "diffhe_lattice_apply_shared")
            return y
        _hip.check(self.L.diffhe_ell_spmv_shared(_hip.ptr(p.Mvals), _hip.ptr(p.cols), _hip.ptr(x_nm), None, 1, None,
                                                 None, _hip.ptr(y), p.n, p.W, Bp, _stream(p.device)),
                   "diffhe_ell_spmv_shared")
        return y

    def cg(self, vals, rhs, Bp, Bv):
        p, L = self.p, self.L
        x = torch.empty((p.n, Bp), dtype=torch.float64, device=p.device)
        work = torch.empty(L.diffhe_cg_workspace_doubles(p.n, Bp), dtype=torch.float64, device=p.device)
        relres = torch.empty(Bp, dtype=torch.float64, device=p.device)
        iters = torch.empty(Bp, dtype=torch.int32, device=p.device)
        st = status_buffer()
        _hip.check(L.diffhe_ell_cg_solve(_hip.ptr(vals), _hip.ptr(p.cols), _hip.ptr(rhs), _hip.ptr(x), p.n, p.W, Bp,
                                         Bv, self.tol, self.max_iter, self.check_every, _hip.ptr(work),
                                         _hip.ptr(relres), _hip.ptr(iters), _hip.ptr(st),
                                         _stream(p.device)), "diffhe_ell_cg_solve")
        return x, int(st[0]), int(st[1]), relres

    # -- lattice path -----------------------------------------------------------------------
    def lattice_assemble(self, kappa, mode, B, Bp, factor=True, n_levels=None, em=False):
        """Per-level symmetric-diagonal operators.  -> (vals per level, Bv, scale, lift, lift_scale).

        One scalar kappa per sample is kept factored, K_b = kappa_b * K_1 (solver.py:88,139 are
        linear in kappa): ONE unit matrix per level is assembled for the whole batch and the
        kernels scale the free rows by kappa_b.  factor=False assembles one matrix per sample
        instead (entries sum_e kappa_b k0_e, rounded like the reference's): kappa_b (K_1 x) differs from
        that in the last bit of every entry, which ill-conditioned systems amplify by their condition number."""
        p, L = self.p, self.L
        st = _stream(p.device)
        k = kappa.detach().to(p.device, torch.float64)
        scale = None
        if mode == K_SCALAR and factor:       # K = kappa K_1: the same factored form, one scale for every sample
            kl, kse, ksb, Bv = None, 0, 0, 1
            scale = k.reshape(1).expand(Bp).contiguous()
        elif mode == K_SCALAR:
            kl, kse, ksb, Bv = k.reshape(1).contiguous(), 0, 0, 1
        elif mode == K_SAMPLE and factor:
            kl, kse, ksb, Bv = None, 0, 0, 1
            scale = torch.ones(Bp, dtype=torch.float64, device=p.device)
            scale[:B] = k.reshape(B)
        elif mode == K_SAMPLE:
            kl = torch.ones(Bp, dtype=torch.float64, device=p.device)
            kl[:B] = k.reshape(B)
            kse, ksb, Bv = 0, 1, Bp
        elif mode == K_ELEM:
            kl, kse, ksb, Bv = k.reshape(p.m, 1).contiguous(), 1, 0, 1
        else:
            kl = self.field_node_major(k, B, Bp, em)
            kse, ksb, Bv = Bp, 1, Bp
        vals, lift = [], None
        for li, lev in enumerate(p.levels[:n_levels] if n_levels else p.levels):
            if li > 0 and mode in (K_ELEM, K_SAMPLE_ELEM):   # coarse kappa = mean of the 4 children
                kc = torch.empty((lev.m, Bv), dtype=torch.float64, device=p.device)
                prev = p.levels[li - 1]
                _hip.check(L.diffhe_lattice_restrict_kappa(_hip.ptr(kl), _hip.ptr(kc), lev.nx, lev.ny,
                                                           prev.nx // lev.nx, prev.ny // lev.ny, Bv, st),
                           "diffhe_lattice_restrict_kappa")
                kl = kc
            v = torch.empty((lev.nd, lev.n, Bv), dtype=torch.float64, device=p.device)
            lf = torch.empty((lev.n, Bv), dtype=torch.float64, device=p.device) if li == 0 else None
            if li == 0 and mode == K_SAMPLE_ELEM and p.closed_boundary and not self.ref_order:
                # one kappa FIELD per sample on a lattice closed by Dirichlet data, default: entries
                # sum_e kappa_e * fl(t_e / den_e) with the batch-shared quotients precomputed in the reference's
                # rounding -- ONE rounding away from the reference's fl(fl(kappa_e t_e) / den_e) per contribution (a
                # cond * eps effect in u, like the factored form) instead of twelve IEEE fp64 divisions per node and
                # sample (10.3 -> 3 ms at 1024^2 x 256).  NOT on lattices with Neumann parts (cond ~ 1e7 there: a last-bit
                # difference of the entries shows as 4e-10 in u, which is why per-sample scalars are left unfactored on
                # them, `closed_` in _solve_forward) -- they, per-sample SCALARS that are not factored, and
                # operator="assembled" keep the bit-identical order below
                self._lattice_rows(lev, "k0ref", kl, kse, ksb, p.g, v, lf, Bv, st)
            elif li == 0 and kl is not None:   # the operator the solution is defined by: reference operation order
                _hip.check(L.diffhe_ell_assemble_rows_ref(_hip.ptr(lev.tnum), _hip.ptr(lev.den), _hip.ptr(kl), kse, ksb,
                                                          _hip.ptr(lev.ent_ptr), _hip.ptr(lev.contrib),
                                                          _hip.ptr(lev.cols), _hip.ptr(lev.store_slot),
                                                          _hip.ptr(lev.is_bc), _hip.ptr(p.g), _hip.ptr(v), _hip.ptr(lf),
                                                          lev.n, lev.m, 7, Bv, st), "diffhe_ell_assemble_rows_ref(lattice)")
            else:
                self._lattice_rows(lev, "k0", kl, kse, ksb, p.g if li == 0 else lev.zero_g(), v, lf, Bv, st)
            vals.append(v)
            if li == 0:
                lift = lf
        return vals, Bv, scale, lift, scale

    def _lattice_rows(self, lev, which, kl, kse, ksb, g, v, lf, Bv, st):
        """kappa * k0 gathered into the symmetric diagonals of a lattice level (+ the Dirichlet lift).  The lattice form of
        the gather (contribution lists written into the kernel, each kappa_e read once per node) gives bitwise the values
        of the list-driven kernel (tests/test_robustness.py); DIFFHE_LATTICE_ASSEMBLE=0 keeps the latter."""
        L = self.L
        local = lev.k0 if which == "k0" else lev.k0ref()
        if os.environ.get("DIFFHE_LATTICE_ASSEMBLE", "1") != "0":
            # congruent triangles (bit for bit): one (9, 2) table instead of the (9, m) array -- same values
            small = lev.compact(which) if os.environ.get("DIFFHE_COMPACT_K0", "1") != "0" else None
            _hip.check(L.diffhe_lattice_assemble_rows(_hip.ptr(small if small is not None else local),
                                                      1 if small is not None else 0, _hip.ptr(kl), kse, ksb,
                                                      _hip.ptr(lev.is_bc), _hip.ptr(g), _hip.ptr(v), _hip.ptr(lf), lev.nx,
                                                      lev.ny, lev.nd, Bv, st), "diffhe_lattice_assemble_rows")
        else:
            _hip.check(L.diffhe_ell_assemble_rows(_hip.ptr(local), _hip.ptr(kl), kse, ksb, _hip.ptr(lev.ent_ptr),
                                                  _hip.ptr(lev.contrib), _hip.ptr(lev.cols), _hip.ptr(lev.store_slot),
                                                  _hip.ptr(lev.is_bc), _hip.ptr(g), _hip.ptr(v), _hip.ptr(lf), lev.n, lev.m,
                                                  7, Bv, st), "diffhe_ell_assemble_rows(lattice)")

    def pack_cycle_coeffs(self, vals, Bv):
        """Per-sample matrices, fp32-stored V-cycle: (fp32 diagonals, fp16 off-diagonals, per-sample scales) per level --
        8 instead of 12 B of coefficients per node and sample (3 diagonals); every row sum of the fp64 matrix is kept.
        The off-diagonals of sample b are stored divided by a power of two >= that sample's largest free-row diagonal of
        the fine level (one reduction pass), so kappa of any magnitude -- and samples of very different magnitudes in
        one batch -- stay inside the fp16 range.  Contrast INSIDE a sample is what fp16 cannot hold: couplings below
        2^-19 of the scale (fewer than 5 bits left; 0 from 2^-25 on, which would leave rows with a vanishing diagonal)
        are reported by the packing kernel and the caller falls back to plain fp32 copies: returns (None, None, None)."""
        p, L = self.p, self.L
        st = _stream(p.device)
        lev0 = (_hip.MgLevel * 1)()
        lev0[0].nx, lev0[0].ny, lev0[0].nd = p.levels[0].nx, p.levels[0].ny, p.levels[0].nd
        lev0[0].vals, lev0[0].is_bc = vals[0].data_ptr(), p.levels[0].is_bc.data_ptr()
        dmax = torch.empty(Bv, dtype=torch.float64, device=p.device)
        _hip.check(L.diffhe_lattice_max_diag(lev0, Bv, _hip.ptr(dmax), st), "diffhe_lattice_max_diag")
        # power of two >= dmax (exact: frexp); samples without a positive finite diagonal (padding is kappa = 1) get 1
        mant, expo = torch.frexp(dmax)
        scales = torch.ldexp(torch.ones_like(dmax), expo - (mant == 0.5).to(expo.dtype))
        ok = torch.isfinite(dmax) & (dmax > 0)
        scales = torch.where(ok, scales, torch.ones_like(scales)).contiguous()
        flags = torch.zeros(1, dtype=torch.int32, device=p.device)
        d32, o16 = [], []
        for lev, v in zip(p.levels, vals):
            one = (_hip.MgLevel * 1)()
            one[0].nx, one[0].ny, one[0].nd, one[0].vals = lev.nx, lev.ny, lev.nd, v.data_ptr()
            d = torch.empty((lev.n, Bv), dtype=torch.float32, device=p.device)
            o = torch.empty((lev.nd - 1, lev.n, Bv), dtype=torch.float16, device=p.device)
            _hip.check(L.diffhe_lattice_pack_h16(one, Bv, _hip.ptr(scales), _hip.ptr(d), _hip.ptr(o), _hip.ptr(flags), st),
                       "diffhe_lattice_pack_h16")
            d32.append(d)
            o16.append(o)
        if not bool(ok.all()) or int(flags[0]) != 0:
            return None, None, None
        return d32, o16, scales

    def lattice_levels(self, vals, vals32=None, dense=None, shift=None, rdiag32=None, off16=None):
        off16, oscales = off16 if off16 is not None else (None, None)
        """Level descriptors for the C ABI.  dense = (level index, inverse tensor): the hierarchy is cut at that
        level, whose solve becomes one dense product (diffhe_mg_level.dense_inv).  shift = per-level (n,) diagonal
        shifts of a factored operator (diffhe_mg_level.shift)."""
        nl = len(vals) if dense is None else dense[0] + 1
        arr = (_hip.MgLevel * nl)()
        for i, (lev, v) in enumerate(zip(self.p.levels[:nl], vals[:nl])):
            arr[i].nx, arr[i].ny, arr[i].nd, arr[i].reserved = lev.nx, lev.ny, lev.nd, 0
            arr[i].vals, arr[i].is_bc = v.data_ptr(), lev.is_bc.data_ptr()
            arr[i].vals32 = vals32[i].data_ptr() if vals32 is not None and vals32[i] is not None else None
            arr[i].dense_inv = dense[1].data_ptr() if dense is not None and i == nl - 1 else None
            arr[i].shift = shift[i].data_ptr() if shift is not None else None
            arr[i].rdiag32 = rdiag32[i].data_ptr() if rdiag32 is not None and rdiag32[i] is not None else None
            arr[i].offdiag16 = off16[i].data_ptr() if off16 is not None and off16[i] is not None else None
            arr[i].mask32 = lev.mask32().data_ptr() if (arr[i].rdiag32 or arr[i].offdiag16) else None
            arr[i].offdiag_scales = oscales.data_ptr() if arr[i].offdiag16 else None
        return arr

    def lattice_pcg(self, vals, Bv, scale, rhs, Bp, mg, vals32=None, dense=None, x0=None, shift=None, rdiag32=None,
                    off16=None):
        """x0: (n, Bp) initial guess (warm start; left untouched) or None for the cold full-multigrid start."""
        p, L = self.p, self.L
        arr = self.lattice_levels(vals, vals32, dense, shift, rdiag32, off16)
        nl = len(arr)
        warm = x0 is not None and x0.shape == (p.n, Bp)
        x = x0.clone() if warm else torch.empty((p.n, Bp), dtype=torch.float64, device=p.device)
        work = torch.empty(L.diffhe_lattice_pcg_workspace_doubles(arr, nl, Bp), dtype=torch.float64, device=p.device)
        relres = torch.empty(Bp, dtype=torch.float64, device=p.device)
        iters = torch.empty(Bp, dtype=torch.int32, device=p.device)
        # per-sweep Jacobi damping: Chebyshev weights for the interval [0.5, 2] of D^-1 A when nu == 2
        omegas = mg.get("omegas") or ([0.56, 1.39] if mg["nu"] == 2 else [mg["omega"]] * mg["nu"])
        om = (ctypes.c_double * len(omegas))(*omegas)
        # a multigrid-preconditioned CG that has not converged in a few hundred iterations never will:
        # bound the loop so a defect surfaces as `not_converged` instead of minutes of GPU time
        est = torch.empty(Bp, dtype=torch.float64, device=p.device)
        rule = torch.empty(Bp, dtype=torch.int32, device=p.device)
        st = status_buffer()
        flags = (int(mg.get("fp32", 0)) | (int(mg.get("fmg", 0)) << 1) | ((int(mg.get("fmg_cycles", 1)) - 1) << 2)
                 | ((0 if int(mg.get("floor", 1)) else 1) << 4) | (32 if warm else 0)
                 | (0 if int(mg.get("fused", 1)) else 64) | (0 if int(mg.get("dense_mfma", 1)) else 128)
                 | (0 if int(mg.get("pre4", 1)) else 512)
                 | (256 if (p.closed_boundary and p.regular_cells and p.dense_level() is not None
                            and int(mg.get("cg_fp32_steplength", 1))) else 0))
        self.last_flags = flags
        _hip.check(L.diffhe_lattice_pcg_solve(arr, nl, Bv, _hip.ptr(scale), _hip.ptr(rhs), _hip.ptr(x), Bp, self.tol,
                                              float(mg.get("tol_energy", 0.0) or 0.0),
                                              min(self.max_iter, 500), len(omegas), mg["n_coarse"], om, flags,
                                              _hip.ptr(work),
                                              _hip.ptr(relres), _hip.ptr(est), _hip.ptr(iters), _hip.ptr(rule),
                                              _hip.ptr(st), _stream(p.device)), "diffhe_lattice_pcg_solve")
        self.last_est = est
        self.last_rule = rule
        return x, int(st[0]), int(st[1]), relres

    # -- general path with the aggregation-multigrid preconditioner ---------------------------------
    def amg_setup(self, vals, Bv, fp32=True, levels=None, dense_coarse=False):
        """Per-solve coarse operators (Galerkin sums of the fine values) + the level descriptor array.
        fp32: per-sample matrices also get an fp32 copy of every level's values for the fp32 cycle.
        dense_coarse (batch-shared, plan-cached hierarchies): the last level (<= 128 nodes) gets the inverse of its
        matrix and is solved by one dense product instead of n_coarse Jacobi sweeps."""
        p, L = self.p, self.L
        st = _stream(p.device)
        chain = [dict(n=p.n, W=p.W, vals=vals, cols=p.cols)]
        for lv in (levels if levels is not None else p.amg_levels):
            vc = torch.empty((lv["W"], lv["n"], Bv), dtype=torch.float64, device=p.device)
            _hip.check(L.diffhe_ell_galerkin(_hip.ptr(chain[-1]["vals"]), _hip.ptr(lv["ent_ptr"]), _hip.ptr(lv["contrib"]),
                                             _hip.ptr(lv.get("weights")), _hip.ptr(vc), lv["n"], lv["W"], Bv, st),
                       "diffhe_ell_galerkin")
            chain[-1].update(agg=lv["agg"], agg_ptr=lv["agg_ptr"], agg_members=lv["agg_members"],
                             agg_weights=lv.get("agg_weights"), p_cols=lv.get("p_cols"), p_vals=lv.get("p_vals"))
            chain.append(dict(n=lv["n"], W=lv["W"], vals=vc, cols=lv["cols"]))
        arr = (_hip.AmgLevel * len(chain))()
        for i, lv in enumerate(chain):
            arr[i].n, arr[i].W = lv["n"], lv["W"]
            arr[i].vals, arr[i].cols = lv["vals"].data_ptr(), lv["cols"].data_ptr()
            if fp32 and Bv != 1:
                lv["vals32"] = lv["vals"].to(torch.float32)
                arr[i].vals32 = lv["vals32"].data_ptr()
            if "agg" in lv:
                arr[i].agg, arr[i].agg_ptr = lv["agg"].data_ptr(), lv["agg_ptr"].data_ptr()
                arr[i].agg_members = lv["agg_members"].data_ptr()
                if lv.get("p_cols") is not None:      # smoothed aggregation: P as ELL rows, P^T weights
                    arr[i].agg_weights, arr[i].p_cols = lv["agg_weights"].data_ptr(), lv["p_cols"].data_ptr()
                    arr[i].p_vals, arr[i].p_width = lv["p_vals"].data_ptr(), int(lv["p_cols"].shape[0])
        last = chain[-1]
        if dense_coarse and Bv == 1 and len(chain) > 1 and last["n"] <= 128 and os.environ.get("DIFFHE_AMG_DENSE", "1") != "0":
            nc = last["n"]
            dense = torch.zeros((nc, nc), dtype=torch.float64, device=p.device)
            rows = torch.arange(nc, device=p.device).repeat(last["W"])
            dense.index_put_((rows, last["cols"].reshape(-1).long()), last["vals"].reshape(-1), accumulate=True)
            inv = torch.linalg.inv(dense)
            inv = 0.5 * (inv + inv.t())            # symmetric to the last bit: the cycle stays a symmetric preconditioner
            if bool(torch.isfinite(inv).all()):
                last["dense_inv"] = inv.contiguous()
                arr[len(chain) - 1].dense_inv = last["dense_inv"].data_ptr()
        return arr, chain      # keep `chain` alive: it owns the coarse value tensors

    def amg_pcg(self, amg, rhs, Bp, Bv, opts):
        p, L = self.p, self.L
        arr, chain = amg
        nl = len(chain)
        x = torch.empty((p.n, Bp), dtype=torch.float64, device=p.device)
        work = torch.empty(L.diffhe_ell_amg_workspace_doubles(arr, nl, Bp), dtype=torch.float64, device=p.device)
        relres = torch.empty(Bp, dtype=torch.float64, device=p.device)
        iters = torch.empty(Bp, dtype=torch.int32, device=p.device)
        st = status_buffer()
        _hip.check(L.diffhe_ell_amg_pcg_solve(arr, nl, Bv, _hip.ptr(rhs), _hip.ptr(x), Bp, self.tol,
                                              min(self.max_iter, int(opts.get("max_iter", 20000))), int(opts["n_coarse"]),
                                              int(opts["gamma"]),
                                              float(opts["scale"]),
                                              int(opts.get("fp32", 0)) | ((0 if int(opts.get("floor", 1)) else 1) << 4),
                                              _hip.ptr(work),
                                              _hip.ptr(relres), _hip.ptr(iters),
                                              _hip.ptr(st), _stream(p.device)), "diffhe_ell_amg_pcg_solve")
        return x, int(st[0]), int(st[1]), relres

    def grad_kappa_factored(self, vals, lift, lam, x, Bp):
        """dL/dkappa_b = -lam^T K_1 u for a batch-shared (factored) lattice operator in one strip pass:
        lam^T (A_1 x + K_1[free,bc] g).  Returns None below the strip-kernel threshold."""
        p, L = self.p, self.L
        arr = self.lattice_levels(vals[:1])
        part = torch.empty(L.diffhe_lattice_blocks(p.n, Bp) * Bp, dtype=torch.float64, device=p.device)
        out = torch.empty(Bp, dtype=torch.float64, device=p.device)
        rc = L.diffhe_lattice_bilinear(arr, 1, None, _hip.ptr(x), _hip.ptr(lam), _hip.ptr(lift), _hip.ptr(part),
                                       _hip.ptr(out), Bp, _stream(p.device))
        if rc == -3:
            return None
        _hip.check(rc, "diffhe_lattice_bilinear")
        return -out

    def grad_kappa(self, lam, x, Bp, want_elem):
        p, L = self.p, self.L
        if want_elem and p.is_lattice and Bp >= 64:
            # lattice mesh, per-element gradient of every sample: strip pass (each nodal value read once per wave)
            lev = p.levels[0]
            dk_e = torch.empty((p.m, Bp), dtype=torch.float64, device=p.device)
            small = lev.compact("k0") if os.environ.get("DIFFHE_COMPACT_K0", "1") != "0" else None
            _hip.check(L.diffhe_lattice_grad_kappa(lev.nx, lev.ny, _hip.ptr(small if small is not None else lev.k0),
                                                   1 if small is not None else 0, _hip.ptr(lam), _hip.ptr(x), _hip.ptr(p.g),
                                                   _hip.ptr(dk_e), Bp, _stream(p.device)), "diffhe_lattice_grad_kappa")
            return dk_e, None
        nblk = L.diffhe_grad_kappa_blocks(p.m, Bp)
        dk_e = torch.empty((p.m, Bp), dtype=torch.float64, device=p.device) if want_elem else None
        part = torch.empty((nblk, Bp), dtype=torch.float64, device=p.device)
        dk_sum = torch.empty(Bp, dtype=torch.float64, device=p.device)
        k0 = p.k0 if p._ell_ready else p.levels[0].k0      # lattice meshes keep k0 on level 0
        _hip.check(L.diffhe_p1_grad_kappa(_hip.ptr(p.elems), _hip.ptr(k0), _hip.ptr(lam), _hip.ptr(x), _hip.ptr(p.g),
                                          p.npe, p.m, Bp, _hip.ptr(dk_e), _hip.ptr(part), _hip.ptr(dk_sum),
                                          _stream(p.device)), "diffhe_p1_grad_kappa")
        return dk_e, dk_sum

    def grad_kappa_shared(self, lam, x, B, Bp):
        """dL/dkappa_e summed over the batch, (m,): the gradient of ONE per-element field shared by all samples,
        without the (m, Bp) per-sample gradient (what the multi-GPU gradient all-reduce carries)."""
        p, L = self.p, self.L
        dk = torch.empty(p.m, dtype=torch.float64, device=p.device)
        k0 = p.k0 if p._ell_ready else p.levels[0].k0
        _hip.check(L.diffhe_p1_grad_kappa_shared(_hip.ptr(p.elems), _hip.ptr(k0), _hip.ptr(lam), _hip.ptr(x),
                                                 _hip.ptr(p.g), p.npe, p.m, B, Bp, _hip.ptr(dk), _stream(p.device)),
                   "diffhe_p1_grad_kappa_shared")
        return dk


def _precision_text(flags: int, coeff_storage: str, Bv: int, Bp: int, fused_lib: int, recompute_ap: bool) -> str:
    """Plain-words account of the precisions of one lattice solve, from the flag word actually passed to
    diffhe_lattice_pcg_solve and the library's own switches (nothing here is a literal about 'the' configuration)."""
    if not flags & 1:
        return "fp64 throughout: every vector stored fp64, all arithmetic fp64"
    spl2 = Bp % 128 == 0 and fused_lib == 2
    packed = not flags & 64 and fused_lib != 0 and ((coeff_storage == "shared-fp32" and Bp % 64 == 0)
                                                    or (coeff_storage == "fp16-rowsum" and spl2))
    parts = ["iterate x, residual r, right-hand side, the updates x += alpha p and r -= alpha A p and EVERY reduction "
             "(r.r, r.z, p.Ap accumulation, energy estimate): fp64",
             "CG search directions p and all V-cycle (preconditioner) vectors: STORED fp32"]
    if packed:
        parts.append("V-cycle arithmetic: fp32 (" + (("packed, two samples per lane" + (
            " (four in the way-down pass)" if Bp % 256 == 0 and not flags & 512 and coeff_storage == "shared-fp32" else ""))
            if spl2 else "one sample per lane")
                     + "; coefficients: "
                     + {"shared-fp32": "batch-shared fp32 copies, scalar loads",
                        "fp16-rowsum": "per-sample fp32 diagonal + scaled fp16 couplings, row sums kept"}[coeff_storage]
                     + "); it is a preconditioner only")
    else:
        parts.append("V-cycle arithmetic: fp64 in registers on the fp32-stored vectors"
                     + ("" if coeff_storage in ("", "fp64") else f" (coefficients: {coeff_storage})"))
    if Bv == 1 and recompute_ap:
        parts.append("A p is never stored: the residual update recomputes it in fp64 from the stored fp32 p")
        if flags & 256 and Bp % 64 == 0 and coeff_storage == "shared-fp32":
            parts.append("p.Ap: stencil in fp32 on the stored p, accumulated fp64 -- enters the STEP LENGTH alpha only "
                         "(closed regular lattice); r = b - A x holds in fp64 whatever alpha is")
        else:
            parts.append("p.Ap: fp64 stencil")
    else:
        parts.append("A p: fp64, stored")
    return "; ".join(parts)


def _rule_counts(rule: torch.Tensor, B: int) -> dict:
    c = torch.bincount(rule[:B].to(torch.int64), minlength=3).tolist()
    return {"cap": c[0], "residual": c[1], "energy": c[2]}


def _solve_forward(solver, kappa, f, load=None, node_major=False):
    """u = (K(kappa) + c M_L)^{-1} (F(f) + load) with Dirichlet elimination (c = solver.reaction, 0 for the reference's
    problem).  Returns (u, state); `state` carries what the explicit adjoint needs (assembled operators, the
    eliminated solution, layout facts).
    node_major (2D paths): f, load and u are (n, B) -- the solver's own layout -- instead of the API's (B, n): no
    layout change on the way in or out (with B a valid padded batch and zero Dirichlet data, u IS the solver's
    iterate, no copy at all)."""
    ctx = types.SimpleNamespace()
    reaction = float(solver.reaction)
    if load is not None and load.numel() == 0:
        load = None
    plan: SolvePlan = solver._plan()
    eng = _Engine(plan, solver.tol, solver.max_iter, solver.check_every, solver.assembly)
    eng.ref_order = solver.operator == "assembled"
    out_device = f.device
    batched = f.dim() == 2
    m, n = plan.m, plan.n
    if node_major and (plan.is_chain or not batched):
        raise ValueError("layout='node' takes (n, B) tensors on 2D meshes")
    if node_major:
        f = f.t()                       # a (B, n) VIEW for the shape logic below; the data stays (n, B)
        load = load.t() if load is not None else None
    ctx.node_major = node_major
    B_f = f.shape[0] if batched else None
    # layout='node': per-sample kappa fields may come element-major, (m, B), like f and u -- no transposing pass for
    # kappa or its gradient either (a square (m, m) tensor is read that way)
    kappa_em = bool(node_major and kappa.dim() == 2 and tuple(kappa.shape) == (m, B_f))
    ctx.kappa_em = kappa_em
    mode, B_k = (K_SAMPLE_ELEM, B_f) if kappa_em else _kappa_mode(kappa, m, B_f)
    B = B_f if B_f is not None else (B_k if B_k is not None else 1)
    if B_k is not None and B_k != B:
        raise ValueError(f"kappa batch {B_k} does not match f batch {B}")
    # lattice fast path unless more than 2 % of the nodes are interior Dirichlet nodes (measured: 5 % on 256^2 needs
    # 99 geometric-multigrid iterations and misses the parity tolerance; the aggregation path takes 33 and meets it)
    lattice = plan.is_lattice and solver.method == "auto" and plan.n_bc_interior <= 0.02 * plan.n
    # Options of THIS call: the user's settings stay as given (a solver object serves scalar and per-element kappa
    # alike; nothing chosen for one call sticks to the next).
    mg, amg, tol = dict(solver.mg), dict(solver.amg), solver._tol_user
    if mode in (K_ELEM, K_SAMPLE_ELEM):
        # Per-element gradients difference the nodal fields, so they amplify the ROUGH part of the solver error by
        # ~ the mesh resolution; that part keeps converging with the recurrence residual after the true residual
        # norm has stalled, so per-element kappa runs to 1e-14 without the attainable-accuracy floor (measured:
        # dL/dkappa_e against the oracle 2.2e-10 -> 1.4e-11 on a 288 x 296 mesh for two more iterations).
        if "floor" not in solver._mg_user:
            mg["floor"] = 0
        amg.setdefault("floor", 0)     # honoured as given when the caller put a "floor" key into solver.amg
    if tol is None:
        # Default stop (relative residual).  Fully Dirichlet-bounded lattices with one kappa per sample are well
        # conditioned for their size and multigrid keeps error ~ residual: 1e-12 (validated against the oracle
        # in every bench run).  Per-element fields (their per-element gradients amplify solver error), partly
        # Neumann boundaries and the general path get one or two more decades.
        closed = lattice and plan.closed_boundary
        simple = mode in (K_SCALAR, K_SAMPLE)
        # small systems (< 10^5 nodes) get one more decade whatever their kind: there an iteration costs next to
        # nothing, and odd shapes (7 x 61 cells of aspect 50, say) converge slowly enough for the error to sit well
        # above the residual
        tol = 1e-12 if (plan.is_chain or (closed and simple and plan.n >= 100_000)) else \
            (1e-13 if (closed or not lattice) and simple else 1e-14)
    # The energy norm controls nodal values only where the boundary is (almost) all Dirichlet (Friedrichs: no
    # near-null mode).  With large Neumann parts the nearly constant mode carries next to no energy per unit of
    # amplitude, and the estimate fell 40x below the nodal error (330 x 125 lattice, Dirichlet data on one edge and
    # one interior node: 4e-10 in u at an estimate of 1e-11) -- such meshes stop on the residual alone, as before.
    closed_lattice = lattice and plan.closed_boundary
    if not closed_lattice and "tol_energy" not in solver._mg_user:
        mg["tol_energy"] = 0.0
    # an explicit `tol` is a request on the RESIDUAL: the energy-norm stop (which ends a solve at relative residuals
    # up to ~3e-7) steps aside unless the caller asked for it too
    if solver._tol_user is not None and "tol_energy" not in solver._mg_user:
        mg["tol_energy"] = 0.0
    if "tol_energy" not in solver._mg_user and mg.get("tol_energy"):
        # The energy-norm stop is calibrated on NODAL error (the estimate sits 3-10x above it).  Per-element
        # gradients are pointwise products of the gradients of u and lambda: their max-norm error ran 20-60x
        # above the estimate on rough data (288 x 296 and 202 x 70 lattices, log-normal fields, random forcing), so
        # per-element kappa asks for two more decades.  Small systems get one more whatever their kind, like `tol`.
        if mode in (K_ELEM, K_SAMPLE_ELEM):
            mg["tol_energy"] *= 1e-2
        if plan.n < 100_000:
            mg["tol_energy"] *= 0.1
    solver.tol = eng.tol = tol          # `solver.tol` reports the tolerance of the last call
    ctx.mg, ctx.amg = mg, amg
    f_dev = f.detach().to(plan.device, torch.float64)
    f_dev = f_dev if node_major else f_dev.contiguous()      # node-major: a transposed view of contiguous (n, B) data
    info = SolveInfo()
    ctx.solver, ctx.plan, ctx.eng = solver, plan, eng
    ctx.mode, ctx.B, ctx.batched_f, ctx.out_device = mode, B, batched, out_device
    ctx.kappa_shape, ctx.kappa_device = kappa.shape, kappa.device
    ctx.kappa_value = kappa.detach().to(plan.device, torch.float64).reshape(-1)[0] if mode == K_SCALAR else None

    load_dev = None
    if load is not None:      # extra nodal load, added to the assembled F on the free rows
        load_dev = load.detach().to(plan.device, torch.float64)
        load_dev = (load_dev.reshape(1, n).expand(B, n) if load_dev.dim() == 1 else load_dev).contiguous()   # (B, n)
        if load_dev.shape != (B, n):
            raise ValueError(f"load must be (n,) or (B,n) with B={B}, n={n}, got {tuple(load.shape)}")
    ctx.load_batched = load is not None and load.dim() == 2
    ctx.reaction = reaction
    if plan.n_bc == 0 and reaction == 0.0:
        # no Dirichlet node and no reaction term: K is singular (constants are in its null space).  The reference
        # solves it anyway and returns garbage of size 1e15 (solver.py:174, no check); here the 1D scan returns NaN
        # and the iterative paths stop at the iteration cap -- either way it is said out loud
        warnings.warn("diffhe: the system is singular (pure Neumann problem: no Dirichlet node, no reaction term); "
                      "the returned values are not a solution", RuntimeWarning)
    # the scan solver inverts a pure path-graph Laplacian: with a reaction term the chain takes the general path
    if reaction and plan.is_p2:
        raise NotImplementedError("reaction term with P2 elements: the lumped P2 mass vanishes at the vertices")
    use_chain = plan.is_chain and reaction == 0.0
    if use_chain and load_dev is not None:
        # the 1D load map of solver.py:95-96 is diagonal (h/2 from each side): an extra load is a change of forcing
        f_dev = (f_dev if batched else f_dev.reshape(1, n).expand(B, n)) + load_dev / plan.lumped_mass()
        batched_dev = True
    else:
        batched_dev = batched
    if use_chain:
        L = eng.L
        kdev = kappa.detach().to(plan.device, torch.float64).contiguous()
        ksb, kse = {K_SCALAR: (0, 0), K_SAMPLE: (1, 0), K_ELEM: (0, 1), K_SAMPLE_ELEM: (m, 1)}[mode]
        u = torch.empty((B, n), dtype=torch.float64, device=plan.device)
        # reference-order mode (default): the system the reference assembled in fp64 (rounded diagonal), see chain1d.hip
        cflags = _hip.CHAIN_REFERENCE_ORDER if solver.chain == "reference" else 0
        info.path = "chain1d-scan-ref" if cflags else "chain1d-scan"
        ns = L.diffhe_chain1d_stage_doubles(n, B, plan.max_seg_len, cflags)   # 0: every segment fits the registers
        stage = torch.empty(ns, dtype=torch.float64, device=plan.device) if ns > 0 else None
        _hip.check(L.diffhe_chain1d_solve(_hip.ptr(plan.x), _hip.ptr(kdev), ksb, kse, _hip.ptr(f_dev),
                                          n if batched_dev else 0, _hip.ptr(plan.seg), plan.n_seg, _hip.ptr(plan.g),
                                          _hip.ptr(u), n, n, B, plan.max_seg_len, cflags, _hip.ptr(stage),
                                          _stream(plan.device)), "diffhe_chain1d_solve")
        ctx.chain_flags = cflags
        ctx.saved = (kdev, ksb, kse, u)
    elif lattice:
        info.path = "lattice-mgpcg"
        Bp = padded_batch(B)
        # per-sample scalar kappa stays factored on closed lattices only: with large Neumann parts the system is
        # ill-conditioned enough (cond ~ 1e7 in the randomised sweep) for the last-bit difference between
        # kappa_b (K_1 x) and (sum_e kappa_b k0_e) x to show as 4e-10 in u
        closed_ = plan.closed_boundary and solver.operator != "assembled"
        # factored operator (one plan-constant unit matrix per level, scalar kappa per sample or for all): the levels
        # from ~33^2 nodes down are replaced by ONE dense product with the cached inverse of that level's matrix (they
        # cost ~45 launch-bound launches per cycle); a mesh that small as a whole -- the reference's own 2D sizes -- is
        # solved DIRECTLY by that product (level index 0: no iteration at all)
        # a reaction term c M_L does not scale with kappa: K_b + c M_L is assembled per sample (or once, scalar kappa)
        factored = closed_ and mode in (K_SCALAR, K_SAMPLE)
        kappa_free_unit = factored           # the stored matrix is the unit-kappa K_1 of the mesh: plan-constant
        # a reaction term c M_L does not scale with kappa: a factored operator carries it as a batch-shared diagonal
        # SHIFT, A_b = kappa_b K_1 + diag(c m) (coefficients stay scalar loads; no cached dense inverse then: it would
        # depend on c / kappa_b); per-sample matrices get it added to their diagonals
        didx = plan.dense_level() if (factored and mg.get("dense_coarse", 1) and reaction == 0.0) else None
        n_levels = None if didx is None else didx + 1
        # (cutting the hierarchy of per-sample matrices at 9^2 / 17^2 / 33^2 -- 11 to 30 fewer launches per cycle, the last
        # level on its Chebyshev iteration -- was measured: 9 + 9 -> 10 + 10 iterations, 208.6 -> 224.0 ... 226.9 ms; r4ab)
        vals, Bv, scale, lift, lift_scale = eng.lattice_assemble(kappa, mode, B, Bp, factor=closed_,
                                                                 n_levels=n_levels, em=kappa_em)
        shift = None
        if reaction and factored:
            shift = eng.reaction_shifts(reaction, len(vals))
        elif reaction:
            eng.add_reaction(vals, reaction, lattice=True)
        f_nm = _as_node_major(eng, f_dev, B, Bp, n, node_major)
        rhs = eng.load_vector(f_nm, lift, Bv, Bp, lift_scale, lattice=True)
        if load_dev is not None:
            rhs += eng.to_node_major(load_dev, B, Bp, n, zero_mask=plan.is_bc)
        # fp32-stored V-cycle: per-sample matrices are read from an fp32 copy of the coefficients; a batch-SHARED
        # matrix gets an fp32 copy and the reciprocal of its main diagonal (a few MB), which switch the strip levels
        # to the two-samples-per-lane kernels (packed fp32 arithmetic; batches that are multiples of 128, no shift)
        vals32 = rdiag32 = off16 = None
        if mg.get("fp32") and Bv != 1 and mg.get("h16", 1) and Bp > 1:
            d32_, o16_, osc_ = eng.pack_cycle_coeffs(vals, Bv)   # fp32 diagonal + fp16 off-diagonals (row sums kept)
            if d32_ is not None:
                vals32, off16 = d32_, (o16_, osc_)
        if mg.get("fp32") and Bv != 1 and vals32 is None:
            vals32 = [v.float() for v in vals]
        elif mg.get("fp32") and Bv == 1 and mg.get("strip2", 1) and Bp % 64 == 0 and shift is None:
            vals32, rdiag32 = plan.shared_fp32(vals, cacheable=factored and kappa_free_unit)
        info.coeff_storage = ("fp64" if not mg.get("fp32") else
                              ("fp16-rowsum" if off16 is not None else "fp32") if Bv != 1 else "shared-fp32")
        dense = None
        if didx is not None:
            if didx == 0:
                info.path = "lattice-direct"
                mg = ctx.mg = dict(mg, fp32=0)        # the direct product runs in fp64
            dense = plan.dense_coarse(didx, vals, bool(mg.get("fp32")))
        wkey = (Bp, mode, reaction)
        x, its, bad, relres = eng.lattice_pcg(vals, Bv, scale, rhs, Bp, mg, vals32, dense,
                                              x0=plan.warm_get(("u",) + wkey) if solver.warm_start else None, shift=shift,
                                              rdiag32=rdiag32, off16=off16)
        ctx.shift, ctx.wkey, ctx.rdiag32, ctx.off16 = shift, wkey, rdiag32, off16
        if solver.warm_start and not bad:
            # never written by the solver again (the next solve starts from a copy) -- but with layout='node' and no padding
            # the caller's u IS this tensor, and an in-place edit under no_grad would silently move the next warm start:
            # keep a private copy then
            shares = node_major and Bp == B and not plan.has_dirichlet_data
            plan.warm_put(("u",) + wkey, x.clone() if shares else x)
        info.stop_rules = _rule_counts(eng.last_rule, B) if info.path != "lattice-direct" else {}
        info.flags = eng.last_flags
        info.precision = _precision_text(eng.last_flags, info.coeff_storage, Bv, Bp, int(eng.L.diffhe_lattice_fused_passes()),
                                         bool(eng.L.diffhe_lattice_recompute_ap()))
        info.tol_energy = float(mg.get("tol_energy", 0.0) or 0.0)
        ctx.dense = dense
        ctx.factored = factored
        info.factored = bool(factored)
        info.iterations, info.not_converged = its, bad
        info.max_relres = float(relres[:B].max())
        info.err_est = float(eng.last_est[:B].max())
        u = _from_node_major(eng, x, B, Bp, n, node_major)
        ctx.saved = (vals, x, Bp, Bv, scale)
        ctx.vals32 = vals32
        ctx.lift = lift if Bv == 1 else None
    else:
        plan.ensure_ell()
        Bp = padded_batch(B)
        # One scalar kappa per sample on a general mesh whose boundary is closed by Dirichlet data (round 4): kept FACTORED
        # like on closed lattices, K_b = kappa_b K_1 -- ONE unit matrix for the batch (Bv = 1: the ELL kernels read it as
        # wave-uniform broadcasts instead of 8 W bytes per node and sample) and the system K_1 x = F_b / kappa_b, whose
        # solution, residual ratio and adjoint are those of the original one; the aggregation hierarchy and its Galerkin
        # operators are then plan-constant and built once.  Not with a reaction term (kappa_b K_1 + c M is no multiple of
        # one matrix), not for operator="assembled", not with Neumann parts (cond * eps, as on lattices).
        ell_factored = (mode in (K_SCALAR, K_SAMPLE) and reaction == 0.0 and solver.operator != "assembled" and not plan.is_p2
                        and solver.method != "ell-jacobi" and plan.closed_boundary_general())
        ctx.ell_inv_kappa = None
        if ell_factored and "fp32" not in solver._amg_user:
            # the fp32-stored cycle is off by default because high-contrast kappa FIELDS break it; the factored operator
            # is the unit-kappa matrix of the mesh -- no coefficient contrast at all: 2.33 -> 2.08 ms per iteration, same 39
            # iterations (jittered 512^2 x 64, gpurun_out/r4v)
            amg["fp32"] = 1
        if ell_factored:
            one = torch.ones(1, dtype=torch.float64, device=plan.device)
            vals, lift = eng.assemble(one, 0, 0, 1)
            Bv = 1
            kpad = torch.ones(Bp, dtype=torch.float64, device=plan.device)
            kpad[:B] = kappa.detach().to(plan.device, torch.float64).reshape(-1)      # (B,), or one scalar for all
            ctx.ell_inv_kappa = 1.0 / kpad
            f_nm = _as_node_major(eng, f_dev, B, Bp, n, node_major)
            rhs = eng.load_vector(f_nm, lift, 1, Bp, kpad)          # F_b = M f_b - kappa_b lift_1
            if load_dev is not None:
                rhs += eng.to_node_major(load_dev, B, Bp, n, zero_mask=plan.is_bc)
            rhs *= ctx.ell_inv_kappa                                 # ... / kappa_b: K_1 x = F_b / kappa_b
        else:
            kdev, kse, ksb, Bv = eng.kappa_device(kappa, mode, B, Bp, em=kappa_em)
            vals, lift = eng.assemble(kdev, kse, ksb, Bv)
            if reaction:
                eng.add_reaction([vals], reaction, lattice=False)
            f_nm = _as_node_major(eng, f_dev, B, Bp, n, node_major)
            rhs = eng.load_vector(f_nm, lift, Bv, Bp)
            if load_dev is not None:
                rhs += eng.to_node_major(load_dev, B, Bp, n, zero_mask=plan.is_bc)
        ctx.amg_hier = None
        if solver.method != "ell-jacobi":
            smoothed = bool(amg.get("smoothed", 1))
            amg_levels = plan.ensure_amg(smoothed=smoothed)
            if amg.get("scale") is None:
                amg["scale"] = 1.3 if amg.get("smoothed", 1) else 1.8
            if amg.get("gamma") is None:      # None = automatic (the default); an explicit 1 or 2 is honoured
                amg["gamma"] = 1
                big = smoothed and plan.n * Bp >= 12_000_000
                # W-cycle where the FINE level dominates the cycle (>= 12 M node-samples): the second visit of the
                # coarser levels costs latency-bound launches on a few thousand nodes, the fine-level sweeps are the
                # bill -- 39 -> 24 iterations, 38.3 -> 35.0 ms at jittered 512^2 x 64, 143.6 -> 118.9 ms at 512^2 x 256,
                # per-sample matrices 102.7 -> 94.8; below that size the V-cycle wins (256^2 x 64: 11.1 against 14.4 ms;
                # 128^2 x 64: 5.9 against 9.3; gpurun_out/r4be, r4bf)
                if big:
                    amg["gamma"] = 2
            if amg_levels and ell_factored:          # plan-constant hierarchy of the unit operator: built once
                ctx.amg_hier = plan.unit_amg((smoothed, bool(amg.get("fp32", 0))),
                                             lambda: eng.amg_setup(vals, 1, bool(amg.get("fp32", 0)), amg_levels, dense_coarse=True))
            elif amg_levels:                         # at least one coarse level: aggregation-AMG PCG
                ctx.amg_hier = eng.amg_setup(vals, Bv, bool(amg.get("fp32", 0)), amg_levels)
        if ctx.amg_hier is not None:
            info.path = "ell-amgpcg"
            x, its, bad, relres = eng.amg_pcg(ctx.amg_hier, rhs, Bp, Bv, amg)
        else:
            info.path = "ell-pcg"
            x, its, bad, relres = eng.cg(vals, rhs, Bp, Bv)
        info.iterations, info.not_converged = its, bad
        info.max_relres = float(relres[:B].max())
        info.factored = bool(ell_factored)
        u = _from_node_major(eng, x, B, Bp, n, node_major)
        ctx.saved = (vals, x, Bp, Bv, None)
    solver.last_info = info
    ctx.path = info.path
    if info.not_converged:
        warnings.warn(f"diffhe: {info.not_converged} of {B} systems did not reach tol={solver.tol:g} "
                      f"(max relative residual {info.max_relres:.2e}, path {info.path})", RuntimeWarning)
    out = u if batched or B > 1 or node_major else u[0]
    return out.to(out_device), ctx


def _as_node_major(eng, f_dev, B, Bp, n, node_major):
    """The (n, Bp) forcing the kernels read.  API layout: one transposing pass.  Node-major input ((B, n) view of
    contiguous (n, B) data): used as it is when B needs no padding, else copied into the padded buffer."""
    if not node_major:
        return eng.to_node_major(f_dev, B, Bp, n)
    f_nb = f_dev.t()
    if Bp == B and f_nb.is_contiguous():
        return f_nb
    out = torch.zeros((n, Bp), dtype=torch.float64, device=f_dev.device)
    out[:, :B] = f_nb
    return out


def _from_node_major(eng, x, B, Bp, n, node_major):
    """u in the caller's layout from the eliminated-system solution x (n, Bp), Dirichlet values added."""
    p = eng.p
    if not node_major:
        return eng.to_sample_major(x, B, Bp, n, add=p.g)
    xo = x if Bp == B else x[:, :B]
    return xo + p.g.unsqueeze(1) if p.has_dirichlet_data else xo   # zero Dirichlet data: u IS x, nothing is copied


def _solve_backward(ctx, gbar, need_k, need_f, need_load=False):
    """Explicit adjoint: lambda = K_free^{-1} gbar_free with the saved operators,
    dL/dkappa = -lambda^T k0 u, dL/df = M^T lambda, dL/dload = lambda.
    Returns (grad_kappa | None, grad_f | None, grad_load | None)."""
    plan, eng, mode, B = ctx.plan, ctx.eng, ctx.mode, ctx.B
    m, n = plan.m, plan.n
    node_major = getattr(ctx, "node_major", False)
    if node_major:
        g_dev = gbar.detach().to(plan.device, torch.float64).reshape(n, B)
    else:
        g_dev = gbar.detach().to(plan.device, torch.float64).reshape(B, n).contiguous()
    info = ctx.solver.last_info
    grad_load = None
    dk_shared = None
    if ctx.path.startswith("chain1d"):
        need_f_user, need_f = need_f, need_f or need_load    # the chain's extra load went in as forcing
        kdev, ksb, kse, u = ctx.saved
        L = eng.L
        df = torch.empty((B, n), dtype=torch.float64, device=plan.device)
        want_e = mode in (K_ELEM, K_SAMPLE_ELEM)
        dk_e = torch.empty((B, m), dtype=torch.float64, device=plan.device) if want_e else None
        part = torch.empty((B, plan.n_seg), dtype=torch.float64, device=plan.device)
        ns = L.diffhe_chain1d_stage_doubles(n, B, plan.max_seg_len, ctx.chain_flags)
        stage = torch.empty(ns, dtype=torch.float64, device=plan.device) if ns > 0 else None
        _hip.check(L.diffhe_chain1d_adjoint(_hip.ptr(plan.x), _hip.ptr(kdev), ksb, kse, _hip.ptr(g_dev), n,
                                            _hip.ptr(u), n, _hip.ptr(plan.seg), plan.n_seg, _hip.ptr(df), n,
                                            _hip.ptr(dk_e), m, _hip.ptr(part), n, B, plan.max_seg_len,
                                            ctx.chain_flags, _hip.ptr(stage), _stream(plan.device)),
                   "diffhe_chain1d_adjoint")
        dk_sample = part.sum(dim=1)                      # (B,) tiny host-side glue
        dk_elem = dk_e
        if need_load:
            grad_load = df / plan.lumped_mass()
        need_f = need_f_user
    else:
        vals, x, Bp, Bv, scale = ctx.saved
        if node_major:
            # the adjoint right-hand side must vanish on Dirichlet rows (lambda_bc = 0): a cotangent that already does
            # (L = sum u^2 with zero Dirichlet data) is used as it is
            dirty = plan.n_bc > 0 and bool((g_dev[plan.bc_index()] != 0).any())
            if Bp == B and g_dev.is_contiguous() and not dirty:
                rhs = g_dev
            else:
                rhs = torch.zeros((n, Bp), dtype=torch.float64, device=plan.device)
                rhs[:, :B] = g_dev
                if dirty:
                    rhs[plan.bc_index()] = 0.0
        else:
            rhs = eng.to_node_major(g_dev, B, Bp, n, zero_mask=plan.is_bc)
        if ctx.path in ("lattice-mgpcg", "lattice-direct"):
            ws = ctx.solver.warm_start is True
            lam, its, bad, relres = eng.lattice_pcg(vals, Bv, scale, rhs, Bp, ctx.mg, ctx.vals32, ctx.dense,
                                                    x0=plan.warm_get(("lambda",) + ctx.wkey) if ws else None,
                                                    shift=ctx.shift, rdiag32=ctx.rdiag32, off16=ctx.off16)
            if ws and not bad:
                plan.warm_put(("lambda",) + ctx.wkey, lam)
            info.adj_stop_rules = _rule_counts(eng.last_rule, B) if ctx.path != "lattice-direct" else {}
            info.adj_err_est = float(eng.last_est[:B].max())
        elif ctx.path == "ell-amgpcg":    # same preconditioner (and the saved per-sample coarse operators) as forward
            if getattr(ctx, "ell_inv_kappa", None) is not None:   # factored: lambda_b = K_1^-1 (gbar_b / kappa_b)
                rhs = rhs * ctx.ell_inv_kappa
            lam, its, bad, relres = eng.amg_pcg(ctx.amg_hier, rhs, Bp, Bv, ctx.amg)
        else:
            if getattr(ctx, "ell_inv_kappa", None) is not None:
                rhs = rhs * ctx.ell_inv_kappa
            lam, its, bad, relres = eng.cg(vals, rhs, Bp, Bv)
        info.adj_iterations = its
        info.adj_max_relres = float(relres[:B].max())
        info.not_converged += bad
        want_e = mode in (K_ELEM, K_SAMPLE_ELEM)
        dk_nm = dk_sum = None
        if need_k and ctx.path in ("lattice-mgpcg", "lattice-direct") and mode in (K_SCALAR, K_SAMPLE) and Bv == 1 \
                and (ctx.factored or not ctx.reaction):   # the bilinear form must see K alone: factored operators keep
            # the reaction term apart (ctx.shift), assembled ones carry it in `vals`
            dk_sum = eng.grad_kappa_factored(vals, ctx.lift, lam, x, Bp)   # shared matrix: one strip pass
            if dk_sum is not None and mode == K_SCALAR and not ctx.factored:
                dk_sum = dk_sum / ctx.kappa_value                           # vals carry kappa: K = kappa K_1
        if need_k and mode == K_ELEM:
            dk_shared = eng.grad_kappa_shared(lam, x, B, Bp)       # (m,): summed over the batch in the kernel
        elif need_k and dk_sum is None:
            dk_nm, dk_sum = eng.grad_kappa(lam, x, Bp, want_e)
        dk_sample = dk_sum[:B] if need_k and dk_sum is not None else None
        dk_elem = None
        if need_k and mode == K_SAMPLE_ELEM:
            # (B, m) like the API's kappa, or left element-major (m, B) when kappa came that way
            dk_elem = (dk_nm if Bp == B else dk_nm[:, :B]) if ctx.kappa_em else eng.to_sample_major(dk_nm, B, Bp, m)
        df = None
        if need_f:
            df = eng.apply_M(lam, Bp, lattice=ctx.path.startswith("lattice-"))
            df = df[:, :B] if node_major else eng.to_sample_major(df, B, Bp, n)
        if need_load:
            grad_load = lam[:, :B].clone() if node_major else eng.to_sample_major(lam, B, Bp, n)

    grad_k = None
    if need_k:
        if mode == K_SCALAR:
            grad_k = dk_sample.sum().reshape(ctx.kappa_shape)
        elif mode == K_SAMPLE:
            grad_k = dk_sample.reshape(ctx.kappa_shape)
        elif mode == K_ELEM:
            grad_k = (dk_shared if dk_shared is not None else dk_elem.sum(dim=0)).reshape(ctx.kappa_shape)
        else:
            grad_k = dk_elem.reshape(ctx.kappa_shape)
        grad_k = grad_k.to(ctx.kappa_device)
    grad_f = None
    if need_f:
        grad_f = df if ctx.batched_f else df.sum(dim=0)
        grad_f = grad_f.to(ctx.out_device)
    if grad_load is not None:
        grad_load = (grad_load if ctx.load_batched else grad_load.sum(dim=1 if node_major else 0)).to(ctx.out_device)
    return grad_k, grad_f, grad_load




# ---------------------------------------------------------------------------------------------
# torch.library custom ops: diffhe::fe_solve (forward) and diffhe::fe_solve_backward (adjoint).
# Non-tensor context (the solver, the per-call adjoint state) travels as integer handles.
# ---------------------------------------------------------------------------------------------
_SOLVERS: "weakref.WeakValueDictionary[int, DifferentiableFESolver]" = weakref.WeakValueDictionary()
_STATES: Dict[int, object] = {}
_TOKENS = itertools.count(1)


class _StateGuard:
    """Owned by the autograd context of one differentiated solve: when the graph is freed (backward without
    retain_graph, or the output tensor dropped) the saved adjoint state goes with it -- the lifetime autograd gives
    its own saved tensors, so `backward(retain_graph=True)`, repeated backward passes and gradcheck work."""

    def __init__(self, token: int):
        self.token = token

    def __del__(self):
        _STATES.pop(self.token, None)


@torch.library.custom_op("diffhe::fe_solve", mutates_args=())
def fe_solve(kappa: torch.Tensor, f: torch.Tensor, load: torch.Tensor, handle: int,
             save: bool, node_major: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """(u, token) = solve with the solver registered under `handle`; `token` names the saved
    adjoint state (0 when `save` is false).  `load`: extra nodal load vector, empty for none.
    node_major: f, load and u are (n, B) instead of (B, n)."""
    solver = _SOLVERS[handle]
    u, state = _solve_forward(solver, kappa, f, load, node_major)
    token = 0
    if save:
        token = next(_TOKENS)
        _STATES[token] = state     # freed with the autograd graph of this solve (_StateGuard): no cap on pending solves
    return u, torch.tensor(token, dtype=torch.int64)


@fe_solve.register_fake
def _fe_solve_fake(kappa, f, load, handle, save, node_major=False):
    solver = _SOLVERS[handle]
    n, m = solver.mesh.n_nodes, solver.mesh.n_elements
    if node_major:
        return f.new_empty(tuple(f.shape), dtype=torch.float64), torch.empty((), dtype=torch.int64)
    B_f = f.shape[0] if f.dim() == 2 else None
    _, B_k = _kappa_mode(kappa, m, B_f)
    B = B_f if B_f is not None else (B_k if B_k is not None else 1)
    shape = (B, n) if (f.dim() == 2 or B > 1) else (n,)
    return f.new_empty(shape, dtype=torch.float64), torch.empty((), dtype=torch.int64)


@torch.library.custom_op("diffhe::fe_solve_backward", mutates_args=())
def fe_solve_backward(gbar: torch.Tensor, token: torch.Tensor, need_k: bool, need_f: bool, need_load: bool,
                      kappa_like: torch.Tensor, f_like: torch.Tensor,
                      load_like: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(dL/dkappa, dL/df, dL/dload) for the forward call named by `token`; unused gradients come back empty."""
    state = _STATES.get(int(token))
    if state is None:
        raise RuntimeError("diffhe: adjoint state of this solve is gone (its autograd graph was freed)")
    gk, gf, gl = _solve_backward(state, gbar, need_k, need_f, need_load)
    return (gk if gk is not None else kappa_like.new_empty(0), gf if gf is not None else f_like.new_empty(0),
            gl.to(load_like.dtype) if gl is not None else load_like.new_empty(0))


@fe_solve_backward.register_fake
def _fe_solve_backward_fake(gbar, token, need_k, need_f, need_load, kappa_like, f_like, load_like):
    return (torch.empty_like(kappa_like) if need_k else kappa_like.new_empty(0),
            torch.empty_like(f_like) if need_f else f_like.new_empty(0),
            torch.empty_like(load_like) if need_load else load_like.new_empty(0))


def _fe_setup_context(ctx, inputs, output):
    kappa, f, load, _, _, node_major = inputs
    # node-major: u may BE the solver's saved iterate -- saving it lets autograd refuse a backward after an in-place edit
    real = not isinstance(output[1], torch._subclasses.FakeTensor)
    # A sentinel among the saved tensors: autograd drops its saved tensors at the end of a backward pass that does not
    # retain the graph, the sentinel dies with them and takes the adjoint state (per-sample operators, iterates: tens of
    # GB at the bench size) along -- BEFORE the caller lets go of u / the loss.  A state that lived until then overlaps
    # the next step's forward solve and the caching allocator has to find a second set of blocks for it: measured as
    # new device allocations (hipMalloc, 70-700 ms each) in otherwise steady 93 ms steps.
    sentinel = (torch.empty(0),) if real else ()
    ctx.save_for_backward(output[1], kappa, f, load, *((output[0],) if node_major else ()), *sentinel)
    ctx.handle, ctx.node_major = inputs[3], bool(node_major)
    if real:
        weakref.finalize(sentinel[0], _STATES.pop, int(output[1]), None)
        ctx.state_guard = _StateGuard(int(output[1]))       # and in any case together with the graph


def _element_forms(plan: SolvePlan):
    """Unit-kappa element stiffness and load matrices (m, npe, npe) and the (m, npe) connectivity, as torch tensors
    built from the plan's coordinates -- the differentiable restatement of reference solver.py:86-96 (1D) and :125-145
    (2D P1) used by the second-order path only."""
    cached = plan.__dict__.get("_element_forms")
    if cached is not None:
        return cached
    if plan.npe != plan.dim + 1:
        raise NotImplementedError("diffhe: second-order derivatives are implemented for P1 elements only")
    el = plan.elems.long().t().contiguous()                          # (m, npe)
    X = plan.coords.to(torch.float64)                                # (dim, n)
    if plan.dim == 1:
        h = (X[0][el[:, 1]] - X[0][el[:, 0]]).abs()
        k0 = torch.tensor([[1.0, -1.0], [-1.0, 1.0]], dtype=torch.float64, device=h.device)[None] / h[:, None, None]
        m0 = torch.eye(2, dtype=torch.float64, device=h.device)[None] * (0.5 * h)[:, None, None]
    else:
        x, y = X[0][el], X[1][el]                                    # (m, 3)
        b = torch.stack([y[:, 1] - y[:, 2], y[:, 2] - y[:, 0], y[:, 0] - y[:, 1]], 1)
        c = torch.stack([x[:, 2] - x[:, 1], x[:, 0] - x[:, 2], x[:, 1] - x[:, 0]], 1)
        area = 0.5 * ((x[:, 1] - x[:, 0]) * (y[:, 2] - y[:, 0]) - (x[:, 2] - x[:, 0]) * (y[:, 1] - y[:, 0])).abs()
        # degenerate triangles are skipped silently, as in the reference (solver.py:120-121), the first-order kernels
        # (ell.hip) and plan.py: no contribution to K or F instead of a division by ~0
        keep = (area >= 1e-15)[:, None, None]
        safe = torch.where(area >= 1e-15, area, torch.ones_like(area))
        k0 = torch.where(keep, (b[:, :, None] * b[:, None, :] + c[:, :, None] * c[:, None, :]) / (4.0 * safe)[:, None, None],
                         torch.zeros((), dtype=torch.float64, device=area.device))
        m0 = torch.where(keep, (safe / 9.0)[:, None, None].expand(-1, 3, 3), torch.zeros((), dtype=torch.float64,
                                                                                      device=area.device)).contiguous()
    plan.__dict__["_element_forms"] = (el, k0, m0)
    return el, k0, m0


def _second_order_backward(ctx, grad_u):
    """The adjoint written with differentiable pieces, for backward(create_graph=True) / Hessian-vector products:

        lambda = A(kappa)^-1 gbar          a solve of the SAME solver class on the mesh with homogeneous Dirichlet data
                                           (gbar enters as `load`: rows of Dirichlet nodes dropped), itself differentiable;
        u      = the forward solve again   (recorded this time: the first one's graph ends at the custom op);
        dL/dkappa_e = - lambda_e^T k0_e u_e,   dL/df = M^T lambda,   dL/dload = lambda      plain torch gathers / sums.

    Costs two HIP solves per first-order gradient instead of one, and element-local torch temporaries of (B, m, npe):
    meant for the sizes Hessian-vector products are taken at, not for the first-order hot path (which never comes here).
    Each further derivative of the result runs the explicit adjoint of those two solves (or, with create_graph again,
    this function recursively)."""
    solver = _SOLVERS.get(ctx.handle)
    if solver is None:
        raise RuntimeError("diffhe: the solver of this solve is gone; second-order backward needs it alive")
    _token, kappa, f, load = ctx.saved_tensors[:4]
    node_major = ctx.node_major
    need_k, need_f, need_load = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
    plan = solver._plan()
    el, k0, m0 = _element_forms(plan)
    m, n = plan.m, plan.n
    twin = solver._adjoint_twin()
    _SOLVERS[id(twin)] = twin
    g = grad_u.to(torch.float64)
    lam, _ = torch.ops.diffhe.fe_solve(kappa, torch.zeros_like(g), g, id(twin), True, node_major)
    u, _ = torch.ops.diffhe.fe_solve(kappa, f, load, ctx.handle, True, node_major)
    # (B, n) views for the element-local maps
    lam_b = (lam.t() if node_major else lam).reshape(-1, n)
    u_b = (u.t() if node_major else u).reshape(-1, n)
    B = lam_b.shape[0]
    gk = gf = gl = None
    if need_k:
        B_f = B if (f.dim() == 2) else None
        kappa_em = bool(node_major and kappa.dim() == 2 and tuple(kappa.shape) == (m, B))
        mode, _ = (K_SAMPLE_ELEM, B) if kappa_em else _kappa_mode(kappa, m, B_f)
        s_be = -torch.einsum("bep,epq,beq->be", lam_b[:, el], k0, u_b[:, el])      # (B, m)
        if mode == K_SCALAR:
            gk = s_be.sum().reshape(kappa.shape)
        elif mode == K_SAMPLE:
            gk = s_be.sum(1).reshape(kappa.shape)
        elif mode == K_ELEM:
            gk = s_be.sum(0).reshape(kappa.shape)
        else:
            gk = (s_be.t() if kappa_em else s_be).reshape(kappa.shape)
    if need_f:
        y = torch.einsum("epq,beq->bep", m0, lam_b[:, el])                          # M^T lambda, M symmetric per element
        gf_b = torch.zeros_like(lam_b).index_add_(1, el.reshape(-1), y.reshape(B, -1))
        gf = (gf_b.t() if node_major else gf_b).reshape(f.shape) if f.dim() == lam.dim() else gf_b.sum(0).reshape(f.shape)
    if need_load:
        gl = lam.reshape(load.shape) if load.dim() == lam.dim() else lam_b.sum(0).reshape(load.shape)
    return gk, gf, gl, None, None, None


def _fe_backward(ctx, grad_u, _grad_token):
    if torch.is_grad_enabled():
        # backward(create_graph=True) / autograd.grad(..., create_graph=True): the caller wants a gradient it can
        # differentiate again.  The explicit adjoint below is not recorded by autograd (what it returns would carry no
        # graph, a Hessian-vector product through it would silently be zero): take the differentiable restatement.
        return _second_order_backward(ctx, grad_u)
    token, kappa, f, load = ctx.saved_tensors[:4]
    need_k, need_f, need_load = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
    gk, gf, gl = torch.ops.diffhe.fe_solve_backward(grad_u, token, need_k, need_f, need_load, kappa, f, load)
    return (gk if need_k else None), (gf if need_f else None), (gl if need_load else None), None, None, None


torch.library.register_autograd("diffhe::fe_solve", _fe_backward, setup_context=_fe_setup_context)


class DifferentiableFESolver(nn.Module):
    """Assemble and solve the P1 system of a mesh for a forcing (reference solver.py:21-43).

    Parameters
    ----------
    mesh : FEMesh
    kappa : float or torch.Tensor -- diffusion coefficient (see module docstring).
    device, tol, max_iter, check_every, assembly, method, mg, chain : HIP-path knobs (ours; the
        reference has none).  `assembly` is "gather" (deterministic) or "atomic" (general
        path); `method="ell"` forces the general ELL path on lattice meshes; `mg` overrides
        the multigrid parameters (nu, n_coarse, omega) of the lattice path.
    """

    def __init__(self, mesh: FEMesh, kappa: float = 1.0, *, device=None, tol: Optional[float] = None,
                 max_iter: int = 20000, check_every: int = 25, assembly: str = "gather", method: str = "auto",
                 mg: Optional[dict] = None, chain: str = "reference", warm_start=False, reaction: float = 0.0,
                 operator: str = "auto", amg: Optional[dict] = None):
        super().__init__()
        self.mesh = mesh
        if isinstance(kappa, (int, float)):
            self._kappa = torch.tensor(kappa, dtype=torch.float64)        # reference solver.py:36-37
        else:
            self._kappa = kappa.to(dtype=torch.float64)                  # reference solver.py:38-39
        if assembly not in ("gather", "atomic"):
            raise ValueError(f"Unknown assembly: {assembly!r}")
        if method not in ("auto", "ell", "ell-jacobi"):
            raise ValueError(f"Unknown method: {method!r}")
        if chain not in ("reference", "exact"):
            raise ValueError(f"Unknown chain mode: {chain!r}")
        if operator not in ("auto", "assembled"):
            raise ValueError(f"Unknown operator: {operator!r}")
        # 2D lattices with one scalar kappa per sample (or for all): "auto" keeps the operator FACTORED on closed
        # lattices, K_b = kappa_b K_1 -- no matrix traffic, but not the matrix the reference assembles: its entries
        # fl(sum_e fl(fl(kappa_b t_e) / den_e)) carry roundings that depend on kappa_b, and the two solutions differ by
        # ~cond * eps (1024^2: 3e-11 in u, 8e-11 in dL/dkappa -- inside the 1e-10 tolerance, measured against the
        # refined oracle).  "assembled" stores one matrix per sample in the reference's operation
        # order (bit-identical to its K): ~1e-12 from the refined oracle, at per-element-kappa speed.
        self.operator = operator
        # 1D chains: "reference" reproduces the solution of the matrix the reference assembles in fp64 (its rounded
        # diagonal costs 4e-10 in u at 10^4 elements); "exact" is the plain scan, 1e-15 from the exact solution of the
        # unrounded system and ~1.5x faster
        self.chain = chain
        # reaction >= 0 (ours; the reference solves pure diffusion): the problem becomes -div(kappa grad u) + c u = f,
        # discretised as (K + c M_L) u = F with the LUMPED mass M_L = row sums of the reference's load matrix
        # (solver.py:95-96, :143-145).  One backward-Euler step of the heat equation is this with c = 1 / dt
        # (diffhe/heat.py).  A fixed number, not differentiated.
        if not (float(reaction) >= 0.0):
            raise ValueError(f"reaction must be >= 0, got {reaction!r}")
        self.reaction = float(reaction)
        # warm_start (lattice path): the forward and the adjoint solve start from the previous solution on this mesh
        # and batch size (kept on the mesh's plan, so a loop that builds a new solver per step -- the reference's
        # pattern -- still benefits), corrected by one full-multigrid pass on its residual.  For optimisation loops
        # whose kappa moves a little per step; results meet the same stopping rule, only the iteration count changes.
        # Costs two (n, B) fp64 vectors of device memory per mesh; `plan.warm.clear()` drops them.
        # "forward": only the forward solve (an adjoint right-hand side that changes direction from step to step, as
        # the data misfit of an inverse problem does near its minimum, makes a poor guess: config 5 took 3.6 + 6.3
        # iterations warm against 5 + 5 cold, 3.6 + 5 with "forward").
        if warm_start not in (False, True, "forward"):
            raise ValueError(f"Unknown warm_start: {warm_start!r}")
        self.warm_start = warm_start
        # "ell": general path (aggregation-AMG PCG) even on lattice meshes; "ell-jacobi": general path
        # with the plain Jacobi preconditioner
        self.method = method
        # aggregation AMG of the general path: V-cycle (gamma = 1) with the coarse correction scaled by 1.8
        # (over-correction compensates the piecewise-constant interpolation; < 2 keeps the cycle a contraction)
        # fp32 = 1 stores the cycle's vectors (and reads copies of per-sample matrix values) in fp32, like mg["fp32"]:
        # +13 % on benign fields, but OFF by default -- this cycle is a much weaker preconditioner than the geometric one
        # (50-250 iterations), and on high-contrast fields (kappa spanning 1e5) the fp32 roundings inside it stall or
        # break the CG long before 1e-14 (randomised sweep: 2000 iterations, diverging samples; fp64 cycle: 123)
        # max_iter: this cycle's iteration counts grow with coefficient contrast and element anisotropy (78 on a benign
        # 84k-node mesh, 4467 on an 86k-node one with an iid e^-4..e^4 field per sample and 6:1 elements, randomised sweep
        # seed 802 case 6) but the CG keeps converging; a cap of 2000 left that case at 2e-9.
        # smoothed = 1 (round 3): smoothed aggregation -- the prolongation of the same aggregates smoothed by one damped
        # Jacobi step of the unit-kappa operator (batch-shared), coarse operators as weighted Galerkin sums per sample:
        # about half the iterations of the piecewise-constant hierarchy (512^2 through this path: 82 -> see DESIGN);
        # scale None = 1.3 smoothed / 1.8 piecewise constant
        self.amg = dict(n_coarse=16, gamma=None, scale=None, fp32=0, max_iter=20000, smoothed=1)   # gamma None: 1, or 2 on big problems
        for item in filter(None, os.environ.get("DIFFHE_AMG", "").split(",")):  # e.g. "scale=1.0,gamma=2,max_iter=20000"
            key, val = item.split("=")
            self.amg[key] = float(val) if key == "scale" else int(val)
        self.amg.update(amg or {})
        self._amg_user = set((amg or {}).keys()) | {i.split("=")[0] for i in os.environ.get("DIFFHE_AMG", "").split(",") if i}
        # fp32 = 1: the V-cycle (a preconditioner) STORES its vectors in fp32; all arithmetic, the
        # outer CG, its residual, the solution and every dot product stay fp64 (same 1e-10 parity)
        # fmg = 1: the CG starts from a full-multigrid iterate instead of 0 (3 iterations fewer at 1024^2)
        # floor = 1: the stop is `tol` or half the residual level fp64 can attain (u |A| |x|), whichever is larger
        # tol_energy: per sample the CG also stops once the ESTIMATED relative energy-norm error of the iterate,
        # sqrt(r.z / u^T A u) (r.z is the dot the CG computes anyway; with a multigrid preconditioner it is e^T A e),
        # is below it.  Nodal values and per-element gradients -- what the 1e-10 parity tolerance is stated in -- are
        # bounded by the energy norm far more tightly than by the residual: on the 1024^2 bench workload the
        # estimate is 3-10x ABOVE the measured nodal error at every iteration (2.6e-11 vs 7.8e-12 after 5), while the
        # relative residual is still 6e-9 there.  1e-11 leaves >= 10x to the tolerance; 0 turns the criterion off.
        self.mg = dict(nu=2, n_coarse=8, omega=0.8, omegas=None, fp32=1, fmg=1, floor=1, tol_energy=1e-11)
        for item in filter(None, os.environ.get("DIFFHE_MG", "").split(",")):   # e.g. "nu=1,omega=0.85"
            key, val = item.split("=")
            if key == "omegas":
                self.mg[key] = [float(v) for v in val.split(":")]
            else:
                self.mg[key] = float(val) if key in ("omega", "tol_energy") else int(val)
        self.mg.update(mg or {})
        self._mg_user = set((mg or {}).keys()) | {i.split("=")[0] for i in os.environ.get("DIFFHE_MG", "").split(",") if i}
        self._device = device
        if os.environ.get("DIFFHE_TOL"):
            tol = float(os.environ["DIFFHE_TOL"])
        # relative-residual stop; None = chosen per mesh at the first solve (1e-12 or 1e-13, see _solve_forward)
        self._tol_user = tol
        self.tol, self.max_iter, self.check_every, self.assembly = tol, max_iter, check_every, assembly
        self.last_info = SolveInfo()

    @property
    def kappa(self) -> torch.Tensor:
        return self._kappa

    def _plan(self) -> SolvePlan:
        return get_plan(self.mesh, _resolve_device(self._device))

    def _adjoint_twin(self) -> "DifferentiableFESolver":
        """This solver on the same mesh with HOMOGENEOUS Dirichlet data: u = A^-1 load, the adjoint solve as a
        differentiable call (second-order path).  Built once per solver; options are re-read at every use."""
        twin = self.__dict__.get("_twin")
        if twin is None:
            mesh0 = self.mesh.__dict__.get("_diffhe_zero_twin")
            if mesh0 is None:
                mesh0 = FEMesh(nodes=self.mesh.nodes, elements=self.mesh.elements,
                               dirichlet_nodes={k: 0.0 for k in self.mesh.dirichlet_nodes})
                self.mesh.__dict__["_diffhe_zero_twin"] = mesh0
            twin = DifferentiableFESolver(mesh0, self._kappa, device=self._device)
            self.__dict__["_twin"] = twin
        for name in ("tol", "_tol_user", "max_iter", "check_every", "assembly", "method", "chain", "warm_start",
                     "reaction", "operator", "_mg_user", "_amg_user"):
            setattr(twin, name, getattr(self, name))
        twin.mg, twin.amg, twin.warm_start = dict(self.mg), dict(self.amg), False
        return twin

    def forward(self, f: torch.Tensor, load: Optional[torch.Tensor] = None, layout: str = "sample") -> torch.Tensor:
        """Solve for nodal u.  f: (n,), (n,1) or (B,n); returns float64 (n,) or (B,n)
        on f's device (reference solver.py:49-67 returns CPU float64).
        load (ours): (n,) or (B,n) nodal load added to the assembled load vector F on the free rows (differentiable) --
        the M_L u_prev / dt term of a time step, point sources, a Neumann flux integrated by the caller.
        layout (ours): "sample" = the shapes above; "node" = f, load and u are (n, B), the batch INNERMOST -- the layout
        the kernels work in, so a caller that keeps its batch that way (an optimisation loop over kappa, say) pays no
        transposing pass in or out (3 x 16 B per node and sample of a fwd + adjoint step); 2D meshes, same results."""
        if self.mesh.dim not in (1, 2):
            raise NotImplementedError("Only 1D and 2D supported")       # reference solver.py:67
        if layout not in ("sample", "node"):
            raise ValueError(f"Unknown layout: {layout!r}")
        n = self.mesh.n_nodes
        if layout == "node":
            if f.dim() != 2 or f.shape[0] != n or (load is not None and tuple(load.shape) != tuple(f.shape)):
                raise ValueError(f"layout='node': f (and load) must be (n, B) with n={n}, got {tuple(f.shape)}")
            if self.mesh.dim == 1:     # the 1D scan works sample-major: transposing views in and out
                return self.forward(f.t(), None if load is None else load.t()).t()
            f64 = f.to(torch.float64)
            _SOLVERS[id(self)] = self
            load64 = f64.new_empty(0) if load is None else load.to(torch.float64)
            save = torch.is_grad_enabled() and (self._kappa.requires_grad or f64.requires_grad or load64.requires_grad)
            u, _token = torch.ops.diffhe.fe_solve(self._kappa, f64, load64, id(self), save, True)
            return u
        f64 = f.to(torch.float64)
        if f64.dim() == 2 and f64.shape == (n, 1):
            f64 = f64.reshape(n)                                          # (n,1) works in the reference too
        elif f64.dim() == 2 and f64.shape[1] != n:
            raise ValueError(f"f must be (n,) or (B,n) with n={n}, got {tuple(f.shape)}")
        elif f64.dim() == 1 and f64.shape[0] != n:
            raise ValueError(f"f must have {n} nodal values, got {f64.shape[0]}")
        _SOLVERS[id(self)] = self
        if load is None:
            load64 = f64.new_empty(0)
        else:
            load64 = load.to(torch.float64)
            if load64.shape[-1] != n or load64.dim() not in (1, 2):
                raise ValueError(f"load must be (n,) or (B,n) with n={n}, got {tuple(load.shape)}")
            if load64.dim() == 2 and f64.dim() == 1:
                f64 = f64.reshape(1, n).expand(load64.shape[0], n)
        save = torch.is_grad_enabled() and (self._kappa.requires_grad or f64.requires_grad or load64.requires_grad)
        u, _token = torch.ops.diffhe.fe_solve(self._kappa, f64, load64, id(self), save, False)
        return u

    # reference-private names kept as aliases; both run the HIP path
    def _solve_1d(self, f: torch.Tensor) -> torch.Tensor:
        return self.forward(f)

    def _solve_2d(self, f: torch.Tensor) -> torch.Tensor:
        return self.forward(f)

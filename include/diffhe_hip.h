/*
 * diffhe_hip.h -- C ABI of libdiffhe_hip.so: the MI355X (gfx950) differentiable
 * P1-FEM solve path behind diffhe.DifferentiableFESolver.
 *
 * The reference (danieleschmidt/DiffFE-Physics-Lab) has no FFI of its own: its hot
 * path is three Python methods (diffhe/solver.py:73-98 `_solve_1d`, :104-147
 * `_solve_2d`, :153-183 `_apply_bc_and_solve`) whose adjoint is whatever autograd
 * replays.  Each entry point below replaces a slice of those methods; the slice is
 * cited per function.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in `_host`;
 *   - values are fp64, indices int32, sizes `int`/`long long`;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered;
 *   - functions return 0 (DIFFHE_OK) or a negative DIFFHE_E_* code, never throw,
 *     never allocate: workspaces are caller-owned;
 *   - 1D chain path: sample-major (B, n) arrays with an explicit row stride;
 *   - 2D/general path: node-major, batch-innermost arrays `(n, Bp)`:
 *     element (i, b) lives at `i * Bp + b`, Bp = padded batch (power of two <= 64
 *     or a multiple of 64).  The mesh pattern is shared by the whole batch.
 */
#ifndef DIFFHE_HIP_H
#define DIFFHE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DIFFHE_ABI_VERSION 7

#define DIFFHE_OK 0
#define DIFFHE_E_BADARG (-1)
#define DIFFHE_E_LAUNCH (-2)   /* a HIP launch/runtime call failed */
#define DIFFHE_E_TOOBIG (-3)   /* problem exceeds what this entry point supports */
#define DIFFHE_E_BATCHPAD (-4) /* Bp is not a power of two <= 64 or a multiple of 64 */

int diffhe_abi_version(void);
const char* diffhe_status_string(int status);
/* hipGetLastError()-style text of the last DIFFHE_E_LAUNCH on this thread. */
const char* diffhe_last_hip_error(void);

/* Algorithmic-byte accounting of the solve path.  Every kernel launch of the lattice path and of the shared
 * set-up / layout kernels adds the unique bytes it must read + write once (batch-shared
 * data counts 0; per-sample matrices count their stored diagonals) to a process-wide counter.
 * Returns the totals since the last reset; reset != 0 clears them.  bench.py divides the bytes of the timed
 * steps by their duration: the step-level fraction of the HBM roofline. */
int diffhe_traffic_account(int reset, double* bytes_host, long long* launches_host);

/* ------------------------------------------------------------------------------
 * 1D chain path (elements[e] = (e, e+1)).  Fused assemble + solve; K is never
 * materialised.  The P1 stiffness of a chain is a weighted path-graph Laplacian:
 * its inverse is two prefix sums (flux scan, then potential scan), evaluated per
 * Dirichlet-delimited segment by one workgroup per (segment, sample).
 *
 * Replaces solver.py:73-98 (element integrals h_e, k_e = kappa/h_e, lumped load
 * h_e/2 f_i) + solver.py:153-183 (elimination, solve, scatter-back).
 *
 *   x        (n)            node coordinates
 *   kappa    kappa[b*kappa_sb + e*kappa_se]   (strides 0 broadcast)
 *   rhs      rhs[b*rhs_sb + i]  nodal forcing f (rhs_sb = 0 shares one f)
 *   seg      (n_seg, 3) int32: first node, last node, flags
 *            (bit0: first node is Dirichlet, bit1: last node is Dirichlet)
 *   g        (n) Dirichlet values (0 at free nodes)
 *   u        (B, n) out, row stride ldu
 *   max_seg_len  elements of the longest segment (1 .. n-1): selects the kernel.  Segments up to
 *            10 240 elements are solved in registers; longer ones stage through `stage`.
 *   flags    DIFFHE_CHAIN_REFERENCE_ORDER: solve the system the reference ASSEMBLED in fp64 --
 *            weights k_e = fl(kappa_e/h_e), diagonal fl(k_{i-1} + k_i) (solver.py:88-92), load
 *            fl(fl(h/2) f_i) summed in element order (solver.py:95-96) -- by correcting the exact
 *            scan solution to first order in the diagonal's rounding error (a second scan in the
 *            same kernel; see chain1d.hip).  Measured 7e-12 from torch.linalg.solve at N = 1e4,
 *            where the plain scan -- which solves the unrounded weighted Laplacian to 1e-15 -- is
 *            4e-10 away.  0: plain scan.
 *   stage    global staging buffer of diffhe_chain1d_stage_doubles(...) doubles (0 when every
 *            segment fits the register kernel: may be NULL then; DIFFHE_E_TOOBIG if needed and NULL)
 * ---------------------------------------------------------------------------- */
#define DIFFHE_CHAIN_REFERENCE_ORDER 1

long long diffhe_chain1d_stage_doubles(int n, int B, int max_seg_len, int flags);

int diffhe_chain1d_solve(const double* x, const double* kappa, long long kappa_sb, long long kappa_se,
                         const double* rhs, long long rhs_sb, const int* seg, int n_seg, const double* g,
                         double* u, long long ldu, int n, int B, int max_seg_len, int flags, double* stage,
                         void* stream);

/* Adjoint of the above (reverse of solver.py:89-96,169-181):
 *   lambda = K_free^{-1} gbar_free (lambda = 0 on Dirichlet nodes);
 *   df[b,i]      = lambda_i * sum_{e ni i} h_e/2
 *   dkappa_e[b,e] = -(lambda_{e+1}-lambda_e)(u_{e+1}-u_e)/h_e        (optional, may be NULL)
 *   dkappa_part[b*n_seg + s] = sum over the elements of segment s of dkappa_e
 * With DIFFHE_CHAIN_REFERENCE_ORDER lambda is the solution of the reference's rounded matrix (what
 * autograd's LinalgSolveExBackward computes), same correction as the forward solve.
 */
int diffhe_chain1d_adjoint(const double* x, const double* kappa, long long kappa_sb, long long kappa_se,
                           const double* gbar, long long gbar_sb, const double* u, long long ldu,
                           const int* seg, int n_seg, double* df, long long lddf, double* dkappa_e,
                           long long lddk, double* dkappa_part, int n, int B, int max_seg_len, int flags,
                           double* stage, void* stream);

/* ------------------------------------------------------------------------------
 * General P1 path (any 1D/2D mesh): per-element integrals, deterministic gather
 * assembly into a batch-shared ELL pattern, Dirichlet elimination, batched PCG.
 * ---------------------------------------------------------------------------- */

/* Element integrals, batch-invariant (solver.py:84-88 1D; solver.py:119-139 2D).
 *   coords (dim, n) SoA; elems (npe, m) SoA int32, npe = dim + 1
 *   k0 (npe*npe, m): unit-kappa local stiffness, entry p*npe+q
 *   m0 (npe*npe, m): local load map (1D: diag h/2; 2D: area/9 everywhere)
 *   degenerate triangles (area < 1e-15, solver.py:120-121) get zeros. */
int diffhe_p1_element_integrals(const double* coords, const int* elems, int dim, int n, int m,
                                double* k0, double* m0, void* stream);

/* Deterministic row-gather assembly with fused Dirichlet elimination.
 * Replaces the scatter-add of solver.py:89-92 / :137-140 and the elimination of
 * solver.py:165-171.
 *   local     (npe*npe, m) local matrices (k0 or m0)
 *   kappa     kappa[e*kappa_se + b*kappa_sb] or NULL for kappa == 1
 *   ent_ptr   (n*W + 1) CSR over ELL entries (row i, slot k) -> contributions
 *   contrib   packed (e * 64 + p*npe+q), npe <= 6
 *   cols      (W, n) column id of entry (row i, k) at k*n + i; entry 0 = diagonal;
 *             unused entries point at the row itself
 *   store_slot (W) or NULL: entry k is stored at vals[(store_slot[k], i, b)]; -1 = not
 *             stored (lower triangle of the symmetric-diagonal lattice format: such
 *             entries only feed the Dirichlet lift).  NULL = identity (ELL).
 *   is_bc     (n) bytes or NULL (no elimination: raw K);  g (n) Dirichlet values
 *   vals      out (W, n, Bv); Bv = padded batch, or 1 when kappa is batch-shared
 *   lift      out (n, Bv) or NULL: lift_i = sum_{j in bc} K_ij g_j (0 on Dirichlet rows);
 *             Dirichlet rows become identity rows, couplings to Dirichlet columns 0.
 */
int diffhe_ell_assemble_rows(const double* local, const double* kappa, long long kappa_se, long long kappa_sb,
                             const int* ent_ptr, const int* contrib, const int* cols, const int* store_slot,
                             const unsigned char* is_bc, const double* g, double* vals, double* lift, int n, int m,
                             int W, int Bv, void* stream);

/* diffhe_ell_assemble_rows for FEMesh.rectangle connectivity (reference mesh.py:100-105) without the gather lists: the
 * contributions of the seven entry kinds (0, +1, +W, +nx, -1, -W, -nx; the first `nd` stored as symmetric diagonals
 * (nd, n, Bv), the rest feeding the Dirichlet lift) are written into the kernel in the element order of the
 * reference's loops (solver.py:137-140) -- bitwise the values of diffhe_ell_assemble_rows with the lattice lists, each
 * kappa_e read once per node.  local: (9, m) unit element matrices, or -- local_compact != 0, ABI v7 -- (9, 2): one matrix
 * per triangle orientation (element parity) of a lattice whose triangles are congruent bit for bit (the caller has
 * compared the columns); kappa (optional) strided by kappa_se per element and kappa_sb per sample. */
int diffhe_lattice_assemble_rows(const double* local, int local_compact, const double* kappa, long long kappa_se,
                                 long long kappa_sb, const unsigned char* is_bc, const double* g, double* vals,
                                 double* lift, int nx, int ny, int nd, int Bv, void* stream);
/* The same gather assembly in the REFERENCE'S OPERATION ORDER: tnum (npe*npe, m) holds t = b_p b_q + c_p c_q (2D;
 * +-1 in 1D), den (m) holds 4 area (2D; h in 1D), and every contribution is (kappa * t) / den with each operation
 * rounded on its own (no contracted multiply-adds, a true division), added in element order -- solver.py:88-92,
 * :139-140 verbatim.  The stored values and the lifting terms are then bit-identical to the reference's K; used
 * wherever the matrix is assembled with kappa folded in (everything except the factored per-sample-scalar mode). */
int diffhe_ell_assemble_rows_ref(const double* tnum, const double* den, const double* kappa, long long kappa_se,
                                 long long kappa_sb, const int* ent_ptr, const int* contrib, const int* cols,
                                 const int* store_slot, const unsigned char* is_bc, const double* g, double* vals,
                                 double* lift, int n, int m, int W, int Bv, void* stream);

/* Element-parallel assembly with fp64 global atomics (the literal scatter-add of
 * solver.py:89-92 / :137-140): element integrals are computed from coords and
 * staged in LDS, then scattered for all Bp samples.  `vals` must be zeroed by the
 * caller; `slot_of` (npe*npe, m) gives the ELL slot of entry (p,q) of element e.
 * No Dirichlet handling (use diffhe_ell_apply_dirichlet). */
int diffhe_ell_assemble_atomic(const double* coords, const int* elems, int dim, const double* kappa,
                               long long kappa_se, long long kappa_sb, const int* slot_of, double* vals, int n,
                               int m, int W, int Bp, void* stream);

/* Dirichlet elimination on an assembled ELL matrix (solver.py:165-171). */
int diffhe_ell_apply_dirichlet(const int* cols, const unsigned char* is_bc, const double* g, double* vals,
                               double* F, int n, int W, int Bp, void* stream);

/* y = is_bc ? 0 : (M x - sub_scale[b] * sub) for a batch-shared ELL matrix (values (W, n)): the
 * load vector F = M f (solver.py:95-96 / :143-145) minus the Dirichlet lift
 * (solver.py:166-169), and the adjoint df = M^T lambda (M symmetric; sub = is_bc = NULL).
 *   sub (n, sub_B) with sub_B == Bp or 1 (batch-shared), or NULL; sub_scale (Bp) or NULL (= 1) */
int diffhe_ell_spmv_shared(const double* vals, const int* cols, const double* x, const double* sub, int sub_B,
                           const double* sub_scale, const unsigned char* is_bc, double* y, int n, int W, int Bp,
                           void* stream);

/* Batched Jacobi-preconditioned CG on (W, n, Bv) ELL values, all samples at once.
 * Replaces torch.linalg.solve (solver.py:174) and, for the adjoint, the solve in
 * its autograd backward.  x is the output (initial guess 0).
 *   vals     (W, n, Bv), Bv = Bp or 1 (matrix shared by the batch)
 *   work     diffhe_cg_workspace_doubles(n, Bp) doubles
 *   status_host  pinned host int[4]: [0] iterations run, [1] samples not converged, [2] scratch
 *   relres   (Bp) out: true relative residual |b - A x| / |b| per sample
 *   iters    (Bp) out: iterations each sample was active
 */
long long diffhe_cg_workspace_doubles(int n, int Bp);
int diffhe_ell_cg_solve(const double* vals, const int* cols, const double* b, double* x, int n, int W, int Bp,
                        int Bv, double tol, int max_iter, int check_every, double* work, double* relres,
                        int* iters, int* status_host, void* stream);

/* ------------------------------------------------------------------------------
 * Aggregation multigrid for general meshes.  The hierarchy (aggregates, coarse ELL patterns,
 * Galerkin gather lists) is batch-shared and built on the host once per mesh; coarse VALUES are
 * P^T A P with piecewise-constant P = sums of fine entries, rebuilt per solve / per sample.
 * ---------------------------------------------------------------------------- */
typedef struct diffhe_amg_level {
  int n, W;                /* nodes and ELL width of this level */
  const double* vals;      /* (W, n, Bv) */
  const int* cols;         /* (W, n) */
  const int* agg;          /* (n) node -> node of the NEXT level, -1 = none (Dirichlet); NULL on the last level */
  const int* agg_ptr;      /* (n_next + 1) CSR of the members of each next-level node; NULL on the last level */
  const int* agg_members;  /* node ids, grouped by aggregate */
  const float* vals32;     /* optional fp32 copy of vals, read by the fp32 cycle when Bv == Bp (may be NULL) */
  /* SMOOTHED aggregation (ABI v6; all NULL / 0 = piecewise-constant aggregation as above): the batch-shared
   * prolongation P = (I - omega D_1^-1 A_1) P_0 to the NEXT level, as ELL rows, and its transpose as weights of the
   * CSR above (agg_ptr / agg_members then list the support of each next-level node's column of P) */
  const double* agg_weights; /* (nnz P) weight of each entry of agg_members */
  const int* p_cols;         /* (p_width, n) next-level node of each entry of P's row, -1 = none */
  const double* p_vals;      /* (p_width, n) */
  int p_width;
  int reserved;
  /* LAST level of a batch-shared hierarchy (Bv == 1), optional: the (n, n) row-major INVERSE of this level's matrix,
   * n <= 128 -- the level is then solved by one dense product instead of n_coarse Jacobi sweeps (NULL: sweeps) */
  const double* dense_inv;
} diffhe_amg_level;

/* vals_coarse[(k*n_coarse + I)*Bv + b] = sum of weights[c] * vals_fine[contrib[c]*Bv + b], c in ent_ptr[k*n_coarse+I] ..
 * (weights NULL = 1: piecewise-constant aggregation; smoothed aggregation: weights[c] = P_iI P_jJ, ABI v6) */
int diffhe_ell_galerkin(const double* vals_fine, const int* ent_ptr, const int* contrib, const double* weights,
                        double* vals_coarse, int n_coarse, int W_coarse, int Bv, void* stream);
/* Batched CG preconditioned by one aggregation-multigrid cycle: V(2,2) Chebyshev-weighted Jacobi,
 * `gamma` coarse corrections per level (2 = W-cycle), coarse correction scaled by `scale`, n_coarse
 * sweeps on the last level.  Replaces torch.linalg.solve (solver.py:174) on general meshes.
 * precond_fp32 bit 0: the cycle stores its vectors in fp32 (arithmetic fp64 in registers) and reads levels[].vals32
 * where given; the CG vectors, residuals and dot products stay fp64.  Bit 4: stop on `tol` alone (no floor).
 * Arguments as diffhe_ell_cg_solve; levels is a HOST array.  The stop is floored like
 * diffhe_lattice_pcg_solve's: sample b stops at |r| <= max(tol |b|, 0.5 u |A_b| |x_b|), with the running
 * iterate x (this path starts from 0). */
long long diffhe_ell_amg_workspace_doubles(const diffhe_amg_level* levels, int n_levels, int Bp);
int diffhe_ell_amg_pcg_solve(const diffhe_amg_level* levels, int n_levels, int Bv, const double* b, double* x, int Bp,
                             double tol, int max_iter, int n_coarse, int gamma, double scale, int precond_fp32,
                             double* work,
                             double* relres, int* iters, int* status_host, void* stream);

/* One application of the batched operator, y = A x, with the per-sample dots x.y left as
 * block partials in `part` (diffhe_grad_kappa_blocks(n, Bp) * Bp doubles): the very kernel
 * the CG loop launches once per iteration, exposed so it can be timed and tested alone. */
int diffhe_ell_apply(const double* vals, const int* cols, const double* x, double* y, double* part, int n, int W,
                     int Bp, int Bv, void* stream);

/* ------------------------------------------------------------------------------
 * Lattice fast path: meshes with the connectivity of FEMesh.rectangle (mesh.py:100-105).
 * Operator = symmetric diagonals D0 (i,i), D1 (i,i+1), D2 (i,i+nx+1), D3 (i,i+nx; only when
 * nd == 4), values (nd, n, Bv) produced by diffhe_ell_assemble_rows with
 * store_slot = {0,1,2,3|-1,-1,-1,-1}.  Level l+1 halves level l in x, in y (semi-coarsening of
 * anisotropic meshes) or in both.
 * ---------------------------------------------------------------------------- */
typedef struct diffhe_mg_level {
  int nx, ny;                  /* elements per direction; n = (nx+1)*(ny+1) */
  int nd;                      /* stored diagonals: 3 or 4 */
  int reserved;
  const double* vals;          /* (nd, n, Bv) */
  const unsigned char* is_bc;  /* (n) */
  const float* vals32;         /* optional fp32 copy of vals (NULL = none): read by the fp32-stored
                                  V-cycle (Bv == Bp: per-sample copies; Bv == 1: see rdiag32); the outer CG always
                                  applies the fp64 values */
  const void* dense_inv;       /* LAST level only, Bv == 1, optional: dense inverse (n, n) row-major of this level's
                                  matrix (identity rows -> zero rows), fp32 when the V-cycle is stored fp32 else
                                  fp64.  The coarsest-level solve is then ONE dense product per cycle (scaled by
                                  1 / scale[b]) instead of the Chebyshev iteration; NULL = Chebyshev */
  const double* shift;         /* optional (n), batch-shared: the level operator is scale[b] * K + diag(shift) on the
                                  free rows (reaction term c M_L of a FACTORED operator; must be 0 on Dirichlet rows and
                                  must not be combined with dense_inv); NULL = none */
  const float* rdiag32;        /* optional (n), Bv == 1 only (ABI v6): fp32 reciprocal of the main diagonal vals[0].  Given
                                  together with vals32 (here (nd, n): fp32 copy of the SHARED values) it lets the
                                  fp32-stored V-cycle run its two-samples-per-lane strip kernels (packed fp32
                                  arithmetic, no division); needs mask32, Bp % 128 == 0 and no shift; NULL = the
                                  fp64-in-registers kernels */
  const float* mask32;         /* with rdiag32: (n) 0.0f on Dirichlet rows, 1.0f elsewhere (is_bc as scalar-loadable floats) */
  const void* offdiag16;       /* optional, Bv == Bp only (ABI v6): fp16 off-diagonals (nd - 1, n, Bv), those of sample b
                                  divided by offdiag_scales[b]; vals32 is then the (n, Bv) fp32 main diagonal adjusted to
                                  keep every row sum of `vals` (diffhe_lattice_pack_h16 produces both): 4 + 2 (nd - 1)
                                  instead of 4 nd bytes of coefficients per node and sample in the fp32-stored V-cycle.
                                  No kernel reads in front of (or behind) any array of this struct (ABI v7: the guard
                                  elements v6 asked for in front of vals32 / offdiag16 are gone) */
  const double* offdiag_scales; /* (Bv) device array, ABI v7: per-sample powers of two >= the sample's largest FREE-row
                                  diagonal entry (diffhe_lattice_max_diag), so samples of any magnitude share a batch;
                                  NULL = offdiag16 unused */
} diffhe_mg_level;

/* Batched CG preconditioned by one multigrid V(nu,nu) cycle (weighted Jacobi with the
 * per-sweep damping factors omegas_host[0..nu-1] -- Chebyshev weights; post-smoothing runs
 * them in reverse so the cycle stays symmetric -- P1 transfers, n_coarse sweeps on the last
 * level; n_levels == 1 degenerates to a Jacobi polynomial).  Replaces torch.linalg.solve
 * (solver.py:174) forward and adjoint.
 *   levels   HOST array of n_levels descriptors (device pointers inside)
 *   Bv       Bp (matrix per sample) or 1 (shared); scale (Bp) or NULL: K_b = scale[b]*K on
 *            the free rows (one scalar kappa per sample, solver.py:88,139)
 *   precond_fp32  bit 0: the V-cycle AND the CG search direction p are stored in fp32 (arithmetic
 *            stays fp64 in registers; x, r, Ap and all dot products are fp64, and every update
 *            uses the stored p, so the recursion r = b - A x stays exact);
 *            bit 1: start the CG from a full-multigrid iterate x0 instead of 0;
 *            bits 2-3: extra V-cycles per coarse level of that start (0..3);
 *            bit 5: WARM START -- `x` holds an initial guess on entry (the previous solution of an optimisation
 *            loop): the solve starts from x + FMG(b - A x) (bit 1 set) or from x itself;
 *            bit 6: keep the four single-stage strip passes per level in the fp32 V-cycle of a batch-shared matrix
 *            (default: the fused two-stage passes, 22 instead of 42 B per node and sample and cycle);
 *            bit 7: coarsest-level dense solve of the fp32 cycle with the scalar-load kernel (fp64 accumulation)
 *            instead of the MFMA kernel (fp32 accumulation; the default for batches of >= 64);
 *            bit 8: the lattice is closed by Dirichlet data on all four edges, its cells are near-square and its
 *            hierarchy reaches a small coarsest level (multigrid at its textbook rate): the CG step of a batch-shared matrix may
 *            form p.Ap -- the step length only -- with a packed-fp32 stencil, two samples per lane (cgstep2_kernel);
 *            x and r are updated with the exact fp64 A p either way;
 *            bit 9 (ABI v7): keep the fused PRE pass at two samples per lane (default: four where the batch has whole
 *            waves of 256 samples and the level has 3 diagonals -- half the vector-memory instructions per byte);
 *            bit 4: stop on `tol` alone.  By default (bit 4 clear, bit 1 set) sample b stops at
 *            |r| <= max(tol |b|, 0.5 u |A_b| |x0_b|), u = 2^-53, |A_b| = 2 scale[b] max_i K_ii: fp64 cannot
 *            bring |b - A x| below ~ u |A| |x|, the recurrence residual keeps falling past that level but the
 *            iterate no longer improves (the same backward error a direct fp64 solve, solver.py:174, reaches)
 *   tol      relative residual |r|_2 / |b|_2 per sample
 *   tol_energy  > 0: sample b ALSO stops once the estimated relative energy-norm error of its iterate,
 *            sqrt(r.z / u^T A u), is <= tol_energy.  r.z = r^T M^-1 r is the dot the CG needs for beta anyway;
 *            with the multigrid preconditioner M ~ A it equals e^T A e to the spectral equivalence of M and A;
 *            u^T A u ~ b.x0 of the full-multigrid start (r0.z0 from a zero start).  Nodal and per-element-gradient
 *            errors are what the parity tolerance is stated in, and the energy norm bounds both far more tightly
 *            than the residual does (1024^2, f = 1: relative residual 6e-9 <-> nodal error 6e-12).  0: off.
 *   err_est  (Bp) out or NULL: the last estimate per sample
 *   stop_rule (Bp) out or NULL (ABI v6): the rule that ended each sample -- 1 residual, 2 energy-norm estimate,
 *            0 neither (iteration cap reached, or the direct dense path where nothing iterates)
 *   b, x     (n, Bp) right-hand side / solution (x is read only with flag bit 5; otherwise the start is 0 / FMG(b))
 *   work     diffhe_lattice_pcg_workspace_doubles(...) doubles
 *   relres, iters, status_host: as diffhe_ell_cg_solve */
long long diffhe_lattice_pcg_workspace_doubles(const diffhe_mg_level* levels, int n_levels, int Bp);
int diffhe_lattice_pcg_solve(const diffhe_mg_level* levels, int n_levels, int Bv, const double* scale,
                             const double* b, double* x, int Bp, double tol, double tol_energy, int max_iter, int nu,
                             int n_coarse, const double* omegas_host, int precond_fp32, double* work, double* relres,
                             double* err_est, int* iters, int* stop_rule, int* status_host, void* stream);
/* Opt-in timing of the fused CG-step kernel (the dominant one) inside diffhe_lattice_pcg_solve's own loop, for
 * bench.py's roofline entry: enable = 1 creates two HIP events (per calling thread, the only hidden state in the
 * library, and only in this mode) and resets the counters, 0 stops, < 0 only reads.  The events bracket each launch
 * on the solve's stream and are read after the loop's per-iteration synchronisation.  Returns the totals so far. */
int diffhe_lattice_pcg_profile(int enable, double* total_ms, long long* launches);
/* The same events exist for the other main kernels of an iteration (fine-level launches only); read-only access:
 *   id 0 fused CG step (full-work launches), 1 residual update r -= alpha Ap, 2 first two sweeps from 0,
 *      3 residual + restriction, 4 prolongation + sweep, 5 sweep.   enable / reset through diffhe_lattice_pcg_profile. */
int diffhe_lattice_kernel_profile(int id, double* total_ms, long long* launches);
/* Single kernels of that loop, exposed for timing/tests: y = A x (+ x.y block partials in
 * `part`, diffhe_lattice_blocks(n, Bp) * Bp doubles) and one damped-Jacobi sweep
 * xout = xin + omega (rhs - A xin)/D  (xin NULL = 0). */
int diffhe_lattice_blocks(int n, int Bp);
/* 0: the fp32 V-cycle of a batch-shared matrix runs four single-stage strip passes per level; 1 / 2: the fused
 * two-stage passes (pre-smoothing + residual + restriction; prolongation + post-smoothing) with that many samples per
 * lane (environment DIFFHE_FUSED / DIFFHE_FUSED_SPL; flag bit 6 of diffhe_lattice_pcg_solve switches them off per call).
 * With them, kernel-profile id 2 times the fused PRE pass (9 B per node and sample), id 4 the fused POST pass (13 B),
 * ids 3 and 5 see no launches. */
int diffhe_lattice_fused_passes(void);
/* 1: with a batch-shared matrix and fp32-stored search directions the PCG never stores A p -- the fused CG step keeps it
 * in registers for p.Ap (12 B per node and sample: z, p_old read, p written) and the residual update recomputes it from
 * the stored p (24 B: p, r read; r and its fp32 copy written; kernel-profile id 1).  0 (environment DIFFHE_RUPD=0): A p
 * is written by the CG step (20 B) and read back by pcg_update_kernel (28 B). */
int diffhe_lattice_recompute_ap(void);
int diffhe_lattice_apply(const diffhe_mg_level* level, int Bv, const double* scale, const double* x, double* y,
                         double* part, int Bp, void* stream);
int diffhe_lattice_smooth(const diffhe_mg_level* level, int Bv, const double* scale, const double* rhs,
                          const double* xin, double* xout, double omega, int Bp, void* stream);
/* The fused CG step of diffhe_lattice_pcg_solve as a single launch (timing / tests):
 *   p_out = z + beta[b] p_in (first != 0: p_out = z);  x += alpha[b] p_in (skipped when first);
 *   Ap = A p_out;  part = block partials of p_out . Ap.   z, p_in and p_out are (n, Bp) fp32
 *   (z_fp32 != 0) or fp64: the search direction is stored in the precision of the preconditioner
 *   output; x, Ap and the dots are always fp64 and use the STORED p, so r = b - A x stays exact.
 * x == NULL: the iterate is left alone -- the form diffhe_lattice_pcg_solve launches since ABI v4: it keeps its
 *   directions in a ring of slots and forms x += sum_j alpha_j p_j when the ring is full / at the end, which takes
 *   the x read-modify-write (16 of 36 B per node) out of every iteration.
 * p_in and p_out must be different buffers (halo reads of p_in).  DIFFHE_E_TOOBIG below the
 * strip-kernel threshold. */
int diffhe_lattice_cg_step(const diffhe_mg_level* level, int Bv, const double* scale, const void* z, int z_fp32,
                           const void* p_in, void* p_out, double* x, const double* alpha, const double* beta,
                           int first, double* Ap, double* part, int Bp, void* stream);
/* out[b] = sum_i lam[i,b] * ((A x)[i,b] + add[i]): the dL/dkappa contraction of a batch-FACTORED
 * operator (K_b = kappa_b K_1: -lam^T K_1 u with u = x + g, add = K_1[free,bc] g), one pass over
 * x and lam instead of the element loop.  Returns DIFFHE_E_TOOBIG when the mesh/batch is below the
 * strip-kernel threshold (callers then use diffhe_p1_grad_kappa).  add (n) may be NULL;
 * part: diffhe_lattice_blocks(n, Bp) * Bp doubles; out (Bp). */
int diffhe_lattice_bilinear(const diffhe_mg_level* level, int Bv, const double* scale, const double* x,
                            const double* lam, const double* add, double* part, double* out, int Bp, void* stream);
/* y = mask ? 0 : (M x - sub_scale[b] * sub) for a batch-shared symmetric-diagonal matrix M (nd, n) of a
 * lattice mesh: the load vector F = M f minus the Dirichlet lift (solver.py:143-145, :166-169) and the
 * adjoint df = M^T lambda, without the general ELL pattern.  Arguments as diffhe_ell_spmv_shared. */
int diffhe_lattice_apply_shared(int nx, int ny, int nd, const double* vals, const double* x, const double* sub,
                                int sub_B, const double* sub_scale, const unsigned char* mask, double* y, int Bp,
                                void* stream);
/* dk[e, b] = - sum_{p,q} lambda[node_p, b] k0[p*3+q, e] (u[node_q, b] + g[node_q]) for every element of a LATTICE mesh
 * (FEMesh.rectangle connectivity) and every sample: the per-element gradient of diffhe_p1_grad_kappa as a strip pass
 * (each nodal value loaded once per wave instead of once per incident element).  k0 (9, m) unit-kappa element
 * matrices (k0_compact != 0, ABI v7: (9, 2), one per triangle orientation, as in diffhe_lattice_assemble_rows), lam / u
 * (n, Bp), g (n) or NULL, dk (m, Bp).  DIFFHE_E_TOOBIG for Bp < 64 (use diffhe_p1_grad_kappa). */
int diffhe_lattice_grad_kappa(int nx, int ny, const double* k0, int k0_compact, const double* lam, const double* u,
                              const double* g, double* dk, int Bp, void* stream);
/* Compact coefficient copies of a per-sample matrix for the fp32-stored V-cycle (diffhe_mg_level.vals32 / offdiag16):
 * diag32 (n, Bv) fp32, offdiag16 (nd - 1, n, Bv) fp16 of vals / offdiag_scales[b] (per-sample powers of two >= the
 * sample's max free-row diagonal, device array of Bv doubles); the diagonal absorbs the rounding differences of its row's
 * couplings, so the row sums -- what the smooth error modes see -- are those of `vals`.
 * flags (device int, may be NULL; the caller zeroes it): bit 0 is set when a non-zero coupling is smaller than 2^-19 of
 * its sample's scale -- fp16 keeps fewer than 5 bits there and flushes to 0 from 2^-25 on, which would leave rows with a
 * vanishing diagonal: the packed copies of a matrix with that much contrast INSIDE a sample must not be used (plain fp32
 * copies in vals32 have no such limit).  ABI v7. */
int diffhe_lattice_pack_h16(const diffhe_mg_level* level, int Bv, const double* offdiag_scales, float* diag32,
                            void* offdiag16, int* flags, void* stream);
/* out[b] = max over the FREE rows (is_bc == 0; identity rows carry 1.0 whatever kappa is) of the main diagonal of sample
 * b's level matrix, Bv doubles (zeroed inside).  ABI v7. */
int diffhe_lattice_max_diag(const diffhe_mg_level* level, int Bv, double* out, void* stream);
/* kappa of the coarse triangulation, arrays (m, Bv); sx, sy in {1, 2} = coarsening factor per direction.
 * Full coarsening: mean of the 4 children of each coarse triangle; semi: mean over the 2 covered fine cells. */
int diffhe_lattice_restrict_kappa(const double* kappa_fine, double* kappa_coarse, int nx_coarse, int ny_coarse,
                                  int sx, int sy, int Bv, void* stream);

/* dL/dkappa contraction (reverse of solver.py:89-92 / :137-140, Appendix A step 2):
 *   dk[e,b] = - sum_{p,q} lambda[elem_p,b] * k0[p*npe+q, e] * (u[elem_q,b] + g[elem_q])
 *   (u is the eliminated-system solution, 0 on Dirichlet nodes; g (n) adds the Dirichlet
 *   values back, may be NULL)
 *   dk_e (m, Bp) optional; dk_part (nblk, Bp) block partial sums over elements,
 *   dk_sum (Bp) their deterministic total. nblk = diffhe_grad_kappa_blocks(m, Bp). */
int diffhe_grad_kappa_blocks(int m, int Bp);
int diffhe_p1_grad_kappa(const int* elems, const double* k0, const double* lam, const double* u, const double* g,
                         int npe, int m, int Bp, double* dk_e, double* dk_part, double* dk_sum, void* stream);
/* The same contraction SUMMED OVER THE BATCH, dk[e] = sum_{b < B} dk[e,b] (m doubles): the gradient of a per-element
 * kappa field shared by all samples (reverse of solver.py:137-140 for kappa (m,)) -- what the RCCL gradient
 * all-reduce of BASELINE config 4 carries.  Fixed summation order (bitwise reproducible); the (m, Bp) per-sample
 * gradient is never materialised.  ABI v6. */
int diffhe_p1_grad_kappa_shared(const int* elems, const double* k0, const double* lam, const double* u, const double* g,
                                int npe, int m, int B, int Bp, double* dk, void* stream);

/* Layout changes between the API's (B, n) and the solver's (n, Bp).
 * to_node_major: dst[i*Bp + b] = src[b*ld + i] (b < B), 0 for padding samples and
 *   where zero_mask[i] != 0 (zero_mask may be NULL; src row stride ld = 0 broadcasts).
 * to_sample_major: dst[b*ld + i] = src[i*Bp + b] + add[i] (add may be NULL). */
int diffhe_to_node_major(const double* src, long long ld, const unsigned char* zero_mask, double* dst, int n, int B,
                         int Bp, void* stream);
int diffhe_to_sample_major(const double* src, const double* add, double* dst, long long ld, int n, int B, int Bp,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFHE_HIP_H */

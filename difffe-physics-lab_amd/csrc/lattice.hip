// Lattice fast path: meshes with the connectivity of FEMesh.rectangle (reference mesh.py:79-121;
// node positions may be arbitrary).  The assembled operator is a 7-point stencil, stored as
// SYMMETRIC DIAGONALS (DIA-sym): D0[i] = K[i,i], D1[i] = K[i,i+1], D2[i] = K[i,i+W] (W = nx+1),
// D3[i] = K[i,i+nx] (the quad diagonal b-d; dropped when all triangles are right-angled, where
// it is exactly 0).  No column indices at all; batch-innermost (n, Bp) vectors as in ell.hip.
//
// Solver: batched CG preconditioned by one geometric-multigrid V-cycle (P1 interpolation on the
// nested triangulations, R = P^T, re-discretised coarse operators = Galerkin for nested P1,
// damped-Jacobi smoothing, nu_pre = nu_post so the preconditioner is symmetric).  Replaces
// torch.linalg.solve of reference solver.py:174 (forward) and of its autograd backward (adjoint).
//
// Matrix sharing: Bv = Bp (one matrix per sample) or Bv = 1 (one matrix for the batch) with an
// optional per-sample scale s_b on the free rows, K_b = s_b * K_1 -- the exact form of the
// assembled operator when kappa is one scalar per sample (solver.py:88,139: k_e = kappa * k0_e).
#include <stdlib.h>
#include <type_traits>
#include <string.h>

#include "common.h"

namespace {

using namespace diffhe;
typedef long long i64;

struct Level {
  int nx, ny, n, W, nd;
  const double* v;          // (nd, n, Bv)
  const float* v32;         // optional fp32 copy of v, used by the fp32 V-cycle's strip kernels
  const _Float16* o16;      // optional fp16 off-diagonals (nd - 1, n, Bv), times 1 / osc[b], of a per-sample matrix (Bv == Bp):
                            // with it v32 holds ONLY the main diagonal (n, Bv), adjusted so that every row sum equals
                            // the fp64 matrix's
  const double* osc;        // (Bv) per-sample powers of two >= the sample's largest free-row diagonal entry: stored
                            // off-diagonals lie in [-1, 1] whatever the magnitude of that sample's kappa
  const float* rd32;        // optional (n) fp32 reciprocal of the main diagonal of a batch-SHARED level matrix (Bv == 1):
                            // with v32 and mk32 it switches the fp32 V-cycle to the two-samples-per-lane strip kernels
  const float* mk32;        // (n) 0.0f on Dirichlet rows, 1.0f elsewhere (scalar-loadable form of bc)
  const unsigned char* bc;  // (n)
  const void* inv;          // optional dense inverse (n, n) of a batch-shared level matrix, in the V-cycle's storage type
  const double* shift;      // optional (n) batch-shared diagonal shift: A_b = scale_b * K + diag(shift) (reaction term
                            // c M_L on a FACTORED operator; 0 on Dirichlet rows); NULL = none
};

__device__ inline double shift_at(const Level& L, int i) { return L.shift ? L.shift[i] : 0.0; }

// Matrix-value storage of the strip kernels: fp64; fp32 copies (fp32 V-cycle, per-sample matrices); or `h16m`: fp32 main
// diagonal + fp16 off-diagonals (8 instead of 12 B per node and sample for 3 diagonals).  The preconditioner only has to
// be spectrally close to A: rounding an edge weight to fp16 (2^-11 relative) while the diagonal keeps every ROW SUM of
// the fp64 matrix perturbs A by a graph Laplacian with edge weights 2^-11 |a_ij| -- spectrally equivalent within 0.1 %,
// the null-space behaviour of the smooth modes untouched (a rounded diagonal would shift them by 2^-11 |a_ii| >>
// lambda_min).  Range: the off-diagonals of sample b are stored divided by a power of two >= that sample's largest
// free-row diagonal entry (|a_ij| <= max a_ii for an SPD matrix), so kappa of any magnitude -- and samples of very
// different magnitudes in one batch -- fit; what fp16 cannot hold is contrast INSIDE a sample: couplings below
// 2^-19 of the scale keep fewer than 5 bits (subnormals) and flush to 0 below 2^-25, so the packing kernel reports them and
// the host falls back to plain fp32 copies for that solve (dia_pack_h16_kernel).  bf16 (no scaling needed) was measured
// one PCG iteration worse on the bench workload (10 + 10 against 9 + 9).
struct h16m {};   // tag type
template <typename TM> struct MatTypes { typedef TM diag; typedef TM off; };
template <> struct MatTypes<h16m> { typedef float diag; typedef _Float16 off; };
__device__ __forceinline__ double ldc(const double* __restrict__ p, unsigned lv) { return p[lv]; }
__device__ __forceinline__ double ldc(const float* __restrict__ p, unsigned lv) { return (double)p[lv]; }
__device__ __forceinline__ double ldc(const _Float16* __restrict__ p, unsigned lv) { return (double)(float)p[lv]; }

// 1/d for the smoother: hardware v_rcp_f64 (~2^-23 relative) + one Newton step (~1e-14) -- 4 instructions
// instead of the ~30 of an IEEE fp64 division.  D^-1 only has to be the same positive diagonal everywhere in
// the preconditioner, so the remaining 1e-14 is immaterial.
__device__ inline double fast_rcp(double d) {
  const double r0 = __builtin_amdgcn_rcp(d);
  return fma(r0, fma(-d, r0, 1.0), r0);
}

__device__ inline int dia_off(const Level& L, int k) { return k == 1 ? 1 : (k == 2 ? L.W : L.nx); }

// sum_j K[i,j] x[j] for sample b (unscaled)
template <typename TV>
__device__ inline double dia_row(const Level& L, int Bv, int vb, const TV* __restrict__ x, int i, int b, int Bp) {
  const i64 n = L.n;
  double acc = L.v[(i64)i * Bv + vb] * (double)x[(i64)i * Bp + b];
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    if (k < L.nd) {
      const int off = dia_off(L, k);
      if (i + off < L.n) acc += L.v[((i64)k * n + i) * Bv + vb] * (double)x[(i64)(i + off) * Bp + b];
      if (i - off >= 0) acc += L.v[((i64)k * n + (i - off)) * Bv + vb] * (double)x[(i64)(i - off) * Bp + b];
    }
  }
  return acc;
}

__device__ inline double row_scale(const Level& L, const double* __restrict__ scale, int i, int b) {
  return (scale && !L.bc[i]) ? scale[b] : 1.0;
}

#define STORE_PARTIAL(part, val)                                                          \
  do {                                                                                    \
    const double t__ = block_sum_per_sample((val), Bp, lds);                              \
    if ((threadIdx.x >> 6) == 0 && (threadIdx.x & 63) < (Bp < kWave ? Bp : kWave))        \
      (part)[(i64)blockIdx.x * Bp + nm.b] = t__;                                          \
  } while (0)

// y = A x ; part = per-sample partial of x.y
__global__ __launch_bounds__(256) void dia_apply_dot_kernel(Level L, int Bv, const double* __restrict__ scale,
                                                             const double* __restrict__ x, double* __restrict__ y,
                                                             double* __restrict__ part, int Bp) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const int vb = Bv == 1 ? 0 : nm.b;
  double s = 0.0;
  for (int i = nm.node0; i < L.n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    const double acc = row_scale(L, scale, i, nm.b) * dia_row(L, Bv, vb, x, i, nm.b, Bp) + shift_at(L, i) * x[o];
    y[o] = acc;
    s += acc * x[o];
  }
  STORE_PARTIAL(part, s);
}

// r = b - A x ; optional part = per-sample partial of r.r
template <typename TV>
__global__ __launch_bounds__(256) void dia_residual_kernel(Level L, int Bv, const double* __restrict__ scale,
                                                            const TV* __restrict__ bvec, const TV* __restrict__ x,
                                                            TV* __restrict__ r, double* __restrict__ part, int Bp,
                                                            int dot_bx = 0, double* __restrict__ part2 = nullptr) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const int vb = Bv == 1 ? 0 : nm.b;
  double s = 0.0, s2 = 0.0;
  for (int i = nm.node0; i < L.n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    const double bi = (double)bvec[o];
    const double ri = bi - (row_scale(L, scale, i, nm.b) * dia_row(L, Bv, vb, x, i, nm.b, Bp) + shift_at(L, i) * (double)x[o]);
    if (r) r[o] = (TV)ri;
    s += dot_bx ? bi * (double)x[o] : ri * ri;   // dot_bx: b.x ...
    if (dot_bx) s2 += (double)x[o] * (bi - ri);  // ... and x.(A x): together a lower bound of the solution's energy
  }
  if (part) STORE_PARTIAL(part, s);
  if (part2) STORE_PARTIAL(part2, s2);
}

// damped Jacobi: xout = xin + omega (b - A xin) / D   (xin == NULL: xin = 0)
// optional part = per-sample partial of b.xout  (the r.z dot of the CG, fused into the last sweep)
template <typename TV>
__global__ __launch_bounds__(256) void dia_jacobi_kernel(Level L, int Bv, const double* __restrict__ scale,
                                                          const TV* __restrict__ bvec, const TV* __restrict__ xin,
                                                          TV* __restrict__ xout, double omega,
                                                          double* __restrict__ part, int Bp) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const int vb = Bv == 1 ? 0 : nm.b;
  double s = 0.0;
  for (int i = nm.node0; i < L.n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    const double sc = row_scale(L, scale, i, nm.b);
    const double sh = shift_at(L, i);
    const double dinv = fast_rcp(sc * L.v[(i64)i * Bv + vb] + sh);  // same reciprocal as the strip kernels
    const double bi = (double)bvec[o];
    double xo;
    if (xin)
      xo = (double)xin[o] + omega * (bi - (sc * dia_row(L, Bv, vb, xin, i, nm.b, Bp) + sh * (double)xin[o])) * dinv;
    else
      xo = omega * bi * dinv;
    xout[o] = (TV)xo;
    s += bi * xo;
  }
  if (part) STORE_PARTIAL(part, s);
}

// ---------------------------------------------------------------------------------------------
// Strip kernels: the same three stencil operations with register-level reuse.
//
// A wave owns RW consecutive grid COLUMNS x 64 samples (lanes) and marches down the rows of its
// tile keeping a 3-row window of x in registers: every x value is loaded once per wave (plus
// the 2 halo columns per strip, (RW+2)/RW loads per output) instead of once per stencil leg;
// the 4 waves of a block own 4 adjacent strips, so halo columns hit L1/L2.  Along a row the
// coefficients of a strip are CONTIGUOUS, so with a batch-shared matrix (Bv == 1) they arrive
// as scalar loads off one SGPR base per diagonal -- no per-lane traffic at all.
//
// Invariant relied upon (and kept by every kernel of the solver): all vectors vanish on
// Dirichlet rows, whose matrix rows are identity rows.  It lets the per-sample scale s_b be
// applied to every row without looking up the Dirichlet flag (0 * s_b == 0).
// ---------------------------------------------------------------------------------------------
enum { M_APPLY = 0, M_RESID = 1, M_JACOBI = 2 };
// Fusions folded into the window load:
//   F_PROLONG: the operand is x + P e (coarse-grid correction added on the fly; with M_JACOBI this
//              is "prolongate, correct and post-smooth" in one pass);
//   F_PUPD:    the operand is the NEW search direction p = z + beta p_old of the CG (with M_APPLY
//              this is "update p, apply A, dot p.Ap" in one pass); the kernel also stores p and
//              applies the pending iterate update x += alpha_prev p_old.
//   F_RESTRICT (with M_RESID): the residual is not stored; it is restricted on the fly (P1 full
//              weighting) into the coarse right-hand side.  The strip then covers the 2 CW + 1 fine
//              columns 2 J0 - 1 .. 2 J0 + 2 CW - 1 that feed the wave's CW coarse columns (one fine
//              column is shared with -- and recomputed by -- each neighbour strip) and the tile the
//              fine rows 2 I0 - 1 .. 2 I1 - 1 of the coarse rows I0 .. I1 - 1.
enum { F_NONE = 0, F_PROLONG = 1, F_PUPD = 2, F_RESTRICT = 3, F_PUPD_NX = 4, F_RUPD = 5 };
// F_RUPD (with M_APPLY): the CG's residual update with A p RECOMPUTED from the stored direction p (TA, ex.p_in) instead of
// read back: r -= alpha (A p), the fp32 copy of r and the partials of r.r in one pass -- for a batch-shared matrix (scalar
// loads, no coefficient traffic) reading p's window (4 B + halo) is cheaper than writing and re-reading A p (8 + 8 B).
// F_PUPD_NX: F_PUPD without the iterate update (the solver's form: x is assembled from the kept directions at the
// end); a compile-time variant so that the x stream costs neither registers nor instructions
constexpr bool is_pupd(int fuse) { return fuse == F_PUPD || fuse == F_PUPD_NX; }

struct Extra {
  const void* a0;           // F_PROLONG: coarse correction e (TA);  F_PUPD: z (TA)
  const void* p_in;         // F_PUPD: previous search direction, stored as TA (the type of z)
  void* p_out;              // F_PUPD: new search direction, stored as TA
  double* x;                // F_PUPD: iterate, updated in place (NULL: left alone -- the solver keeps its directions
                            //   and forms x once at the end, pcg_finish_kernel)
  const double* alpha;      // F_PUPD: per-sample alpha of the previous iteration
  const double* beta;       // F_PUPD
  int first;                // F_PUPD: first iteration (p = z, nothing pending)
  int cW;                   // F_PROLONG, F_RESTRICT: row width of the coarse level
  const unsigned char* bc;  // F_PROLONG: fine Dirichlet flags (no correction there); F_RESTRICT: coarse flags
  const double* dotv;       // M_APPLY, F_NONE: dot (A x + addv) against this vector instead of x
  const double* addv;       // M_APPLY, F_NONE: batch-shared (n) vector added to A x (may be NULL)
  float* r32;               // M_RESID, F_NONE, fp64 vectors: also store the residual rounded to fp32 (may be NULL)
  const double* rscale;     //   ... multiplied by this per-sample power of two first (may be NULL: 1)
  const double* sub;        // M_APPLY, F_NONE: y = A x - sub_scale[b] * sub[i], sub batch-shared (n) (may be NULL) ...
  const double* sub_scale;  //   per-sample factor of `sub` (NULL: 1)
  int sub_pb;               //   ... or, sub_pb != 0, one value per sample: sub is (n, Bp) (the Dirichlet lift of per-sample matrices)
  const unsigned char* mask;  // M_APPLY, F_NONE: rows with mask[i] != 0 are stored as 0 (may be NULL)
  int dot_bx;               // M_RESID, F_NONE: the partial sums hold b.x (energy of the iterate) instead of r.r ...
  double* part2;            //   ... and these (same layout as `part`) x.(A x)
};

template <typename TV, typename TA, typename TM, int MODE, int FUSE, int ND, bool SHARED, bool XFROMB, int RW,
          bool TAIL, bool SHIFT = false>
__device__ __forceinline__ double strip_body(const Level& L, double sb, const TV* __restrict__ src,
                                             const TV* __restrict__ bvec, TV* __restrict__ out, double omega,
                                             double omega_in, const Extra& ex, int Bp, int b, int c0w, int r0,
                                             int r1, double& s2) {
  const int W = L.W, nyp = L.ny + 1;
  const i64 n = L.n;
  const i64 Bv = SHARED ? 1 : Bp;
  // Addressing discipline: every pointer below is WAVE-UNIFORM (lives in SGPRs) and the lane's
  // sample index is added last as a 32-bit offset, so loads/stores use the "SGPR base + VGPR
  // offset" form and the kernel needs one address VGPR instead of one 64-bit pair per stream.
  const unsigned lb = (unsigned)b;            // lane offset into (.., Bp) vectors
  const unsigned lv = SHARED ? 0u : (unsigned)b;  // lane offset into the matrix values
  double s = 0.0;

  // Column offsets of the window (q <-> grid column c0w - 1 + q) and of the strip (k <-> c0w + k),
  // relative to column c0w.  The first strip's left halo and the columns past the right edge are clamped (both
  // only in the TAIL instantiation, which every edge strip takes); their window values are forced to 0.
  int dq[RW + 2];
  bool okq[RW + 2];
#pragma unroll
  for (int q = 0; q < RW + 2; ++q) {
    int c = c0w - 1 + q;
    okq[q] = c >= 0 && (!TAIL || c < W);
    if (c < 0) c = 0;
    if (TAIL && c > W - 1) c = W - 1;
    dq[q] = c - c0w;
  }
  // D_k[i] lives at V[(k*n + i)*Bv + vb].  Row r0 - 1 of D_2 / D_3 is read for the south couplings also when r0 == 0:
  // that is the tail of the previous diagonal in the same array (finite, multiplied by a window value of 0).  Nothing is
  // read in front of an array: the east coupling of column c0w - 1 is an in-grid entry for every non-TAIL strip.
  const i64 i0 = (i64)r0 * W + c0w;          // node (r0, c0w)
  typedef typename MatTypes<TM>::diag TD;
  typedef typename MatTypes<TM>::off TO;
  constexpr bool kSplit = sizeof(TO) == 2;   // h16m: diagonal in L.v32, off-diagonals in L.o16 (times 1 / L.osc[b])
  const double osc = kSplit ? L.osc[b] : 1.0;
  const TD* __restrict__ p0 = (sizeof(TD) == 4 ? (const TD*)L.v32 : (const TD*)L.v) + i0 * Bv;
  const TO* __restrict__ p1 = kSplit ? (const TO*)L.o16 + i0 * Bv : (const TO*)(const void*)(p0 + n * Bv);
  const TO* __restrict__ p2 = p1 + n * Bv;
  const TO* __restrict__ p3 = p2 + n * Bv;
  const double* __restrict__ psh = SHIFT ? L.shift + i0 : nullptr;   // diagonal shift at (row, c0w): wave-uniform loads
  const TV* __restrict__ px = src + i0 * Bp;
  const TV* __restrict__ pb = bvec ? bvec + i0 * Bp : nullptr;
  TV* __restrict__ po = (out && FUSE != F_RESTRICT) ? out + i0 * Bp : nullptr;
  const i64 rowV = (i64)W * Bv, rowX = (i64)W * Bp;

  const double inv_omega_in = XFROMB ? 1.0 / omega_in : 0.0;
  const double sub_fac = (MODE == M_APPLY && FUSE == F_NONE && ex.sub && ex.sub_scale) ? ex.sub_scale[b] : 1.0;
  const double rsc = (MODE == M_RESID && FUSE == F_NONE && ex.r32 && ex.rscale) ? ex.rscale[b] : 1.0;
  const double beta = (is_pupd(FUSE) && !ex.first) ? ex.beta[b] : 0.0;
  const double alpha_prev = (FUSE == F_PUPD && !ex.first && ex.x) ? ex.alpha[b] : 0.0;  // alpha is NULL when x is
  const TA* __restrict__ aux = (const TA*)ex.a0;
  // F_PUPD row pointers at (row, c0w), advanced with the others
  const TA* __restrict__ pz = (is_pupd(FUSE)) ? aux + i0 * Bp : nullptr;
  const TA* __restrict__ ppi = (is_pupd(FUSE) || FUSE == F_RUPD) ? (const TA*)ex.p_in + i0 * Bp : nullptr;
  double* __restrict__ pr = (FUSE == F_RUPD) ? ex.x + i0 * Bp : nullptr;          // F_RUPD: ex.x is the residual r
  float* __restrict__ pr32 = (FUSE == F_RUPD && ex.r32) ? ex.r32 + i0 * Bp : nullptr;
  const double alpha_cur = (FUSE == F_RUPD) ? ex.alpha[b] : 0.0;
  const double rsc_u = (FUSE == F_RUPD && ex.r32 && ex.rscale) ? ex.rscale[b] : 1.0;
  TA* __restrict__ ppo = (is_pupd(FUSE)) ? (TA*)ex.p_out + i0 * Bp : nullptr;
  double* __restrict__ pxx = (FUSE == F_PUPD && ex.x) ? ex.x + i0 * Bp : nullptr;  // NULL: the iterate is not touched

  // `row` is the grid row being loaded; xrow / d0row point at (row, c0w); roff = offset of that
  // row from the current one in vector elements
  auto load_window = [&](int row, i64 roff, const TV* __restrict__ xrow, const TD* __restrict__ d0row,
                         double* dst, const double* __restrict__ shrow = nullptr) {
    double ce[RW / 2 + 2], ce2[RW / 2 + 2];
    if (FUSE == F_PROLONG) {  // coarse values around this strip: coarse columns c0w/2 - 1 + j
      const int cr = row >> 1;
#pragma unroll
      for (int j = 0; j < RW / 2 + 2; ++j) {
        int cj = (c0w >> 1) - 1 + j;
        cj = cj < 0 ? 0 : (cj > ex.cW - 1 ? ex.cW - 1 : cj);
        ce[j] = (double)(aux + ((i64)cr * ex.cW + cj) * Bp)[lb];
        ce2[j] = (row & 1) ? (double)(aux + ((i64)(cr + 1) * ex.cW + cj) * Bp)[lb] : 0.0;
      }
    }
#pragma unroll
    for (int q = 0; q < RW + 2; ++q) {
      double v;
      if (is_pupd(FUSE)) {
        const i64 o = roff + (i64)dq[q] * Bp;
        v = (double)(pz + o)[lb];
        if (!ex.first) v += beta * (double)(ppi + o)[lb];
        // the direction is STORED as TA: use the stored (rounded) value everywhere, so that Ap = A p,
        // x += alpha p and r -= alpha Ap stay exactly consistent (r == b - A x is independent of p)
        v = (double)(TA)v;
      } else if (FUSE == F_RUPD) {
        v = (double)(ppi + roff + (i64)dq[q] * Bp)[lb];
      } else {
        v = (double)(xrow + (i64)dq[q] * Bp)[lb];
      }
      if (XFROMB) v = omega_in * v * fast_rcp(sb * ldc(d0row + (i64)dq[q] * Bv, lv) + (SHIFT ? shrow[dq[q]] : 0.0));
      if (FUSE == F_PROLONG) {
        double corr;  // c0w is even: window column q has the parity of q + 1
        if (q & 1)
          corr = (row & 1) ? 0.5 * (ce[(q - 1) / 2 + 1] + ce2[(q - 1) / 2 + 1]) : ce[(q - 1) / 2 + 1];
        else
          corr = (row & 1) ? 0.5 * (ce[q / 2 + 1] + ce2[q / 2]) : 0.5 * (ce[q / 2] + ce[q / 2 + 1]);
        if (ex.bc[(i64)row * W + c0w + dq[q]]) corr = 0.0;
        v += corr;
      }
      dst[q] = okq[q] ? v : 0.0;
    }
  };

  double xm[RW + 2], xc[RW + 2], xp[RW + 2];
  double n2p[RW], d3p[RW + 1];
#pragma unroll
  for (int q = 0; q < RW + 2; ++q) xm[q] = 0.0;
  if (r0 > 0) load_window(r0 - 1, -rowX, px - rowX, p0 - rowV, xm, SHIFT ? psh - W : nullptr);
  load_window(r0, 0, px, p0, xc, psh);
#pragma unroll
  for (int k = 0; k < RW; ++k) n2p[k] = kSplit ? osc * ldc(p2 - rowV + (i64)dq[k + 1] * Bv, lv) : ldc(p2 - rowV + (i64)dq[k + 1] * Bv, lv);
#pragma unroll
  for (int k = 0; k < RW + 1; ++k)
    d3p[k] = (ND == 4) ? (kSplit ? osc * ldc(p3 - rowV + (i64)dq[k + 1] * Bv, lv) : ldc(p3 - rowV + (i64)dq[k + 1] * Bv, lv)) : 0.0;

  constexpr int CWR = (FUSE == F_RESTRICT) ? (RW - 1) / 2 : 1;  // coarse columns of an F_RESTRICT strip
  double racc[CWR], rnext[CWR];
#pragma unroll
  for (int j = 0; j < CWR; ++j) racc[j] = rnext[j] = 0.0;
  const int cI0 = (r0 + 1) >> 1, cJ0 = (c0w + 1) >> 1;          // F_RESTRICT: first coarse row / column

  for (int row = r0; row < r1; ++row) {
    if (row + 1 < nyp) {
      load_window(row + 1, rowX, px + rowX, p0 + rowV, xp, SHIFT ? psh + W : nullptr);
    } else {
#pragma unroll
      for (int q = 0; q < RW + 2; ++q) xp[q] = 0.0;
    }
    double d0[RW], e1[RW + 1], n2c[RW], d3c[RW + 1];
    double resrow[(FUSE == F_RESTRICT) ? RW : 1];
    if (FUSE == F_RESTRICT) {
#pragma unroll
      for (int k = 0; k < RW; ++k) resrow[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      d0[k] = ldc(p0 + (i64)dq[k + 1] * Bv, lv);
      n2c[k] = kSplit ? osc * ldc(p2 + (i64)dq[k + 1] * Bv, lv) : ldc(p2 + (i64)dq[k + 1] * Bv, lv);
    }
#pragma unroll
    for (int k = 0; k < RW + 1; ++k) {
      // east coupling of column c0w - 1 + k (interior strips: c0w >= RW, the column exists; edge strips: clamped)
      const int dc = TAIL ? dq[k] : k - 1;
      e1[k] = kSplit ? osc * ldc(p1 + (i64)dc * Bv, lv) : ldc(p1 + (i64)dc * Bv, lv);
      d3c[k] = (ND == 4) ? (kSplit ? osc * ldc(p3 + (i64)dq[k + 1] * Bv, lv) : ldc(p3 + (i64)dq[k + 1] * Bv, lv)) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const int q = k + 1;
      if (TAIL && (c0w + k >= W || c0w + k < 0)) continue;
      double acc = d0[k] * xc[q];
      acc += e1[k + 1] * xc[q + 1] + e1[k] * xc[q - 1];
      acc += n2c[k] * xp[q] + n2p[k] * xm[q];
      if (ND == 4) acc += d3c[k] * xp[q - 1] + d3p[k + 1] * xm[q + 1];
      const i64 o = (i64)k * Bp;
      const double sh = SHIFT ? psh[dq[k + 1]] : 0.0;   // A = sb K + diag(shift)
      const double diag = SHIFT ? sb * d0[k] + sh : sb * d0[k];
      const double Ax = SHIFT ? sb * acc + sh * xc[q] : sb * acc;
      if (MODE == M_APPLY && FUSE == F_RUPD) {
        double* ra = &(pr + o)[lb];
        const double ri = __builtin_nontemporal_load(ra) - alpha_cur * Ax;
        __builtin_nontemporal_store(ri, ra);
        if (pr32) (pr32 + o)[lb] = (float)(ri * rsc_u);   // read again right away by the V-cycle: left cacheable
        s += ri * ri;
      } else if (MODE == M_APPLY) {
        double y = Ax;
        if (FUSE == F_NONE && (ex.sub || ex.mask)) {  // load vector of a lattice mesh: F = M f - lift, 0 on Dirichlet rows
          const i64 ig = (i64)row * W + c0w + k;
          if (ex.sub) y -= sub_fac * (ex.sub_pb ? (ex.sub + ig * Bp)[lb] : ex.sub[ig]);
          if (ex.mask && ex.mask[ig]) y = 0.0;
        }
        if (po) {
          // CG-step streams (Ap, p, x) are touched once per iteration, 1-2 GB each: nontemporal accesses keep them
          // from evicting the halo columns and the V-cycle's vectors from L2 / Infinity Cache (fused step -4 %)
          if (is_pupd(FUSE)) __builtin_nontemporal_store((TV)y, &(po + o)[lb]);
          else (po + o)[lb] = (TV)y;
        }
        if (FUSE == F_NONE && ex.dotv) {  // bilinear form lam^T (A x + add): dL/dkappa of a factored operator
          const i64 ig = (i64)row * W + c0w + k;
          s += (y + (ex.addv ? ex.addv[ig] : 0.0)) * (ex.dotv + ig * Bp)[lb];
        } else {
          s += y * xc[q];
        }
        if (is_pupd(FUSE)) {  // store the new direction; apply the pending x += alpha_prev * p_old
          __builtin_nontemporal_store((TA)xc[q], &(ppo + o)[lb]);
          if (FUSE == F_PUPD && !ex.first && pxx) {
            double* xa_ = &(pxx + o)[lb];
            __builtin_nontemporal_store(__builtin_nontemporal_load(xa_) + alpha_prev * (double)(ppi + o)[lb], xa_);
          }
        }
      } else {
        const double dinv = (MODE == M_JACOBI) ? fast_rcp(diag) : 0.0;
        // XFROMB: the window holds x1 = omega_in * rhs * dinv, so rhs = x1 / (omega_in * dinv)
        const double bi = XFROMB ? xc[q] * diag * inv_omega_in : (double)(pb + o)[lb];
        const double res = bi - Ax;
        if (MODE == M_RESID && FUSE == F_RESTRICT) {
          resrow[k] = res;
        } else if (MODE == M_RESID) {
          if (po) (po + o)[lb] = (TV)res;
          if (FUSE == F_NONE && sizeof(TV) == 8 && ex.r32) (ex.r32 + ((i64)row * W + c0w + k) * Bp)[lb] = (float)(res * rsc);
          if (FUSE == F_NONE && ex.dot_bx) {
            s += bi * xc[q];
            s2 += xc[q] * (bi - res);   // x.(A x)
          } else {
            s += res * res;
          }
        } else {
          const double xo = xc[q] + omega * res * dinv;
          (po + o)[lb] = (TV)xo;
          s += bi * xo;
        }
      }
    }
    if (FUSE == F_RESTRICT) {
      // strip column k <-> fine column 2 cJ0 - 1 + k, so coarse column cJ0 + j sits at k = 2 j + 1.
      // Full weighting of the P1 lattice: centre 1; W, E, N, S, NE-of-the-row-above, SW-of-the-row-below 1/2.
      const bool store = (row & 1) || row + 1 >= nyp;  // coarse row complete after its odd row (or at the last row)
      if (!(row & 1)) {
#pragma unroll
        for (int j = 0; j < CWR; ++j) racc[j] += resrow[2 * j + 1] + 0.5 * (resrow[2 * j] + resrow[2 * j + 2]);
      } else {
#pragma unroll
        for (int j = 0; j < CWR; ++j) {
          racc[j] += 0.5 * (resrow[2 * j + 1] + resrow[2 * j]);
          rnext[j] = 0.5 * (resrow[2 * j + 1] + resrow[2 * j + 2]);
        }
      }
      if (store) {
        const int I = row >> 1;
        if (I >= cI0) {
#pragma unroll
          for (int j = 0; j < CWR; ++j) {
            const int J = cJ0 + j;
            if (J < ex.cW) {
              const i64 Ic = (i64)I * ex.cW + J;
              (out + Ic * Bp)[lb] = (TV)(ex.bc[Ic] ? 0.0 : racc[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CWR; ++j) {
          racc[j] = rnext[j];
          rnext[j] = 0.0;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RW + 2; ++q) {
      xm[q] = xc[q];
      xc[q] = xp[q];
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) n2p[k] = n2c[k];
#pragma unroll
    for (int k = 0; k < RW + 1; ++k) d3p[k] = d3c[k];
    p0 += rowV; p1 += rowV; p2 += rowV; p3 += rowV;
    if (SHIFT) psh += W;
    px += rowX;
    if (pb) pb += rowX;
    if (po) po += rowX;
    if (is_pupd(FUSE)) { pz += rowX; ppi += rowX; ppo += rowX; if (FUSE == F_PUPD && pxx) pxx += rowX; }
    if (FUSE == F_RUPD) { ppi += rowX; pr += rowX; if (pr32) pr32 += rowX; }
  }
  return s;
}

// Workgroups are handed to the 8 XCDs round-robin by linear id, so blocks x = k (mod 8) share one L2.
// Give each such class a contiguous range of tiles: spatially adjacent strips (which read each other's
// halo columns / rows) then run on the same XCD at about the same time and the halo hits its L2.
__device__ inline int xcd_tile(int x, int gx) {
  const int q = gx >> 3, rem = gx & 7;
  const int k = x & 7, j = x >> 3;
  return k * q + (k < rem ? k : rem) + j;
}

// Body of the strip kernels: tile -> (column strip, row chunk) of this wave, the strip march, the per-sample partials.
// One call level below the __global__ functions on purpose: written directly into the kernel the same code gets a
// different register allocation for the batch-shared (SHARED) fp64 variants -- 72 VGPRs + 60 B of scratch instead of
// 62 for the fused CG step, 141-148 instead of 75-95 for the fp64 residual / apply strips -- and the step measures
// 2 % slower that way (A/B on one MI355X, 1024^2 x 256: 116.1 vs 113.9 ms; fused CG step 1.20 vs 1.17 ms; only the
// fp64-stored Jacobi sweep of mg fp32=0 prefers the direct form, 1.27 vs 1.31 ms).
template <typename TV, typename TA, typename TM, int MODE, int FUSE, int ND, bool SHARED, bool XFROMB, int RW, bool SHIFT>
__device__ __forceinline__ void strip_kernel_body(Level L, const double* __restrict__ scale,
                                                  const TV* __restrict__ xin, const TV* __restrict__ bvec,
                                                  TV* __restrict__ out, double omega, double omega_in, Extra ex,
                                                  double* __restrict__ part, int Bp, int ncb, int TR) {
  __shared__ double lds[4 * kWave];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y * kWave + lane;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int rc = tile / ncb, cb = tile - rc * ncb;
  const int nyp = L.ny + 1;
  int c0w, r0, r1;
  bool active;
  if (FUSE == F_RESTRICT) {  // TR counts COARSE rows, the wave owns (RW - 1) / 2 coarse columns
    const int J0 = (cb * 4 + wave) * ((RW - 1) / 2), I0 = rc * TR;
    const int cnyp = (nyp + 1) >> 1;
    const int I1 = (I0 + TR < cnyp) ? I0 + TR : cnyp;
    c0w = 2 * J0 - 1;
    r0 = I0 > 0 ? 2 * I0 - 1 : 0;
    r1 = (2 * I1 < nyp) ? 2 * I1 : nyp;
    active = J0 < ex.cW && I0 < I1;
  } else {
    c0w = (cb * 4 + wave) * RW;
    r0 = rc * TR;
    r1 = (r0 + TR < nyp) ? r0 + TR : nyp;
    active = c0w < L.W && r0 < r1;
  }
  const double sb = scale ? scale[b] : 1.0;
  const TV* __restrict__ src = XFROMB ? bvec : xin;
  double s = 0.0, s2 = 0.0;
  if (active) {
    // strips that touch the left or right edge take the clamped body -- the first strip (c0w == 0) too: its window column
    // -1 would otherwise read the east coupling one element BEFORE the row, which for row 0 lies in front of the array
    if (c0w + RW + 1 > L.W || c0w <= 0)
      s = strip_body<TV, TA, TM, MODE, FUSE, ND, SHARED, XFROMB, RW, true, SHIFT>(L, sb, src, bvec, out, omega, omega_in, ex,
                                                                              Bp, b, c0w, r0, r1, s2);
    else
      s = strip_body<TV, TA, TM, MODE, FUSE, ND, SHARED, XFROMB, RW, false, SHIFT>(L, sb, src, bvec, out, omega, omega_in, ex,
                                                                               Bp, b, c0w, r0, r1, s2);
  }
  if (part) {
    const double t = block_sum_per_sample(s, Bp, lds);
    if (wave == 0) part[(i64)blockIdx.x * Bp + b] = t;
  }
  if (MODE == M_RESID && FUSE == F_NONE && ex.part2) {
    const double t = block_sum_per_sample(s2, Bp, lds);
    if (wave == 0) ex.part2[(i64)blockIdx.x * Bp + b] = t;
  }
}

template <typename TV, typename TA, typename TM, int MODE, int FUSE, int ND, bool SHARED, bool XFROMB, int RW,
          int MINW = 1>
__global__ __launch_bounds__(256, MINW) void dia_strip_kernel(Level L, const double* __restrict__ scale,
                                                         const TV* __restrict__ xin, const TV* __restrict__ bvec,
                                                         TV* __restrict__ out, double omega, double omega_in,
                                                         Extra ex, double* __restrict__ part, int Bp, int ncb,
                                                         int TR) {
  strip_kernel_body<TV, TA, TM, MODE, FUSE, ND, SHARED, XFROMB, RW, false>(L, scale, xin, bvec, out, omega, omega_in, ex, part,
                                                                          Bp, ncb, TR);
}

// The same strips for a FACTORED operator with a batch-shared diagonal shift, A_b = scale_b K_1 + diag(L.shift)
// (reaction term / heat-equation steps with one scalar kappa per sample): coefficients stay scalar loads.
template <typename TV, typename TA, int MODE, int FUSE, int ND, bool XFROMB, int RW, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void dia_strip_shift_kernel(Level L, const double* __restrict__ scale,
                                                               const TV* __restrict__ xin, const TV* __restrict__ bvec,
                                                               TV* __restrict__ out, double omega, double omega_in,
                                                               Extra ex, double* __restrict__ part, int Bp, int ncb,
                                                               int TR) {
  strip_kernel_body<TV, TA, double, MODE, FUSE, ND, true, XFROMB, RW, true>(L, scale, xin, bvec, out, omega, omega_in, ex, part,
                                                                           Bp, ncb, TR);
}


// ---------------------------------------------------------------------------------------------
// Two samples per lane: the strip kernels of the fp32-stored V-cycle for a batch-SHARED matrix
// (factored operator K_b = s_b K_1, or one per-element field for the whole batch).
//
// A lane owns TWO adjacent samples, a wave 128: every vector access is 8 B per lane / 512 B per wave
// instead of 4 / 256 (this GPU streams 4 B-per-lane accesses at ~4.8 TB/s, 8 B at 5.3-5.5:
// profiles/r02_stream_bench.txt), and the arithmetic is PACKED fp32 (v_pk_fma_f32: both samples per
// instruction) on fp32 coefficient copies that arrive as scalar loads -- about a fifth of the
// instructions per sample of the fp64-in-registers form.  The vectors of this cycle are stored
// fp32 anyway: a stored x carries a 2^-24 relative rounding that enters A x with weight |A||x|, and
// fp32 accumulation of the seven stencil terms adds the same order (measured: same iteration
// counts, same parity).  Written in "unit" form: with ib = 1 / s_b,
//     Jacobi   x' = x + omega rd0 (b ib - K_1 x)          (rd0 = 1 / diag K_1, batch-shared)
//     residual r  = b - s_b (K_1 x)
// so a sweep needs no division at all.  Same strips, tiles, window and fusions as strip_body.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f ld2(const float* __restrict__ p, unsigned lb) { return *(const v2f*)(p + lb); }
__device__ __forceinline__ void st2(float* __restrict__ p, unsigned lb, v2f v) { *(v2f*)(p + lb) = v; }
// Buffer addressing: one resource descriptor per stream (base = the tile's first window row, one column left of the
// strip), a loop-invariant 32-bit per-lane byte offset per column (VGPR) and a wave-uniform 32-bit byte offset per
// row (SGPR, one s_add per iteration): "buffer_load_dwordx2 v, v_off, s[rsrc], s_row offen" -- no 64-bit address
// arithmetic per access (the flat-pointer form cost a v_lshl_add_u64 per load and ~50 scalar adds per row).
// Offsets are relative to the TILE, so they stay far below 2^32 whatever the size of the vector (checked on the host).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ v2f bld(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, unsigned voff, unsigned soff, v2f v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, 0);
}

template <int MODE, int FUSE, int ND, bool XFROMB, int RW, bool TAIL, bool DOT, bool BST>
__device__ __forceinline__ void strip2_body(const Level& L, v2f ib, v2f sb, const float* __restrict__ src,
                                            const float* __restrict__ bvec, float* __restrict__ out, float omega,
                                            float omega_in, const Extra& ex, int Bp, unsigned lb, int c0w, int r0,
                                            int r1, double& s0, double& s1) {
  const int W = L.W, nyp = L.ny + 1;
  const i64 n = L.n;
  const v2f zero2 = {0.0f, 0.0f};
  // Bases sit at window row r0 - 1, one column LEFT of the strip: window column q (grid column c0w - 1 + q) has the
  // non-negative lane offset (dq[q] + 1) * Bp, row `row` the uniform offset (row - r0 + 1) * W * Bp.
  int dq[RW + 2];
  bool okq[RW + 2];
  unsigned offq[RW + 2];
#pragma unroll
  for (int q = 0; q < RW + 2; ++q) {
    int c = c0w - 1 + q;
    okq[q] = !TAIL || (c >= 0 && c < W);   // interior strips (TAIL = false) have every window column inside the grid
    if (TAIL && c < 0) c = 0;
    if (TAIL && c > W - 1) c = W - 1;
    dq[q] = c - c0w;
    offq[q] = 4u * ((unsigned)((dq[q] + 1) * Bp) + lb);   // bytes
  }
  const i64 i0 = (i64)r0 * W + c0w;          // node (r0, c0w)
  const float* __restrict__ p0 = L.v32 + i0;
  const float* __restrict__ p1 = p0 + n;
  const float* __restrict__ p2 = p1 + n;
  const float* __restrict__ p3 = p2 + n;
  const float* __restrict__ prd = L.rd32 + i0;
  const float* __restrict__ pmk = (FUSE == F_PROLONG) ? L.mk32 + i0 : nullptr;
  const i64 tile0 = (i0 - W - 1) * Bp;                                     // element (r0 - 1, c0w - 1)
  const rsrc_t rx = make_rsrc(src + tile0);
  const rsrc_t rb = make_rsrc(bvec ? bvec + tile0 : nullptr);
  const rsrc_t ro = make_rsrc((out && FUSE != F_RESTRICT) ? out + tile0 : nullptr);
  float* __restrict__ po = (out && FUSE != F_RESTRICT) ? out + tile0 + (i64)W * Bp : nullptr;   // row r0, column c0w - 1
  const unsigned rowB = 4u * (unsigned)W * (unsigned)Bp;                   // bytes per grid row
  const float inv_omega_in = XFROMB ? 1.0f / omega_in : 0.0f;
  const float* __restrict__ aux = (const float*)ex.a0;
  unsigned offc[RW / 2 + 2];   // F_PROLONG: coarse columns c0w/2 - 1 + j (clamped), as offsets into a coarse row
#pragma unroll
  for (int j = 0; j < RW / 2 + 2; ++j) {
    int cj = (c0w >> 1) - 1 + j;
    cj = cj < 0 ? 0 : (cj > ex.cW - 1 ? ex.cW - 1 : cj);
    offc[j] = (FUSE == F_PROLONG) ? 4u * ((unsigned)(cj * Bp) + lb) : 0u;
  }
  const int cr0 = (r0 > 0 ? r0 - 1 : 0) >> 1;                              // first coarse row this tile reads
  const rsrc_t rc = make_rsrc(FUSE == F_PROLONG ? aux + (i64)cr0 * ex.cW * Bp : nullptr);
  const unsigned rowCB = 4u * (unsigned)ex.cW * (unsigned)Bp;

  // sx = byte offset of window row `row` in the tile
  auto load_window = [&](int row, unsigned sx, const float* __restrict__ rdrow, const float* __restrict__ mkrow,
                         v2f* dst) {
    v2f ce[RW / 2 + 2], ce2[RW / 2 + 2];
    if (FUSE == F_PROLONG) {  // coarse values around this strip
      const unsigned sc = (unsigned)((row >> 1) - cr0) * rowCB;
#pragma unroll
      for (int j = 0; j < RW / 2 + 2; ++j) {
        ce[j] = bld(rc, offc[j], sc);
        ce2[j] = (row & 1) ? bld(rc, offc[j], sc + rowCB) : zero2;
      }
    }
#pragma unroll
    for (int q = 0; q < RW + 2; ++q) {
      v2f v = bld(rx, offq[q], sx);
      if (XFROMB) v = (v * ib) * (omega_in * rdrow[dq[q]]);   // x1 = omega_in D^-1 rhs, formed on the fly
      if (FUSE == F_PROLONG) {
        v2f corr;  // c0w is even: window column q has the parity of q + 1
        if (q & 1)
          corr = (row & 1) ? 0.5f * (ce[(q - 1) / 2 + 1] + ce2[(q - 1) / 2 + 1]) : ce[(q - 1) / 2 + 1];
        else
          corr = (row & 1) ? 0.5f * (ce[q / 2 + 1] + ce2[q / 2]) : 0.5f * (ce[q / 2] + ce[q / 2 + 1]);
        v += mkrow[dq[q]] * corr;   // mask: 0 on Dirichlet rows (no correction there), 1 elsewhere
      }
      dst[q] = okq[q] ? v : zero2;
    }
  };

  v2f xm[RW + 2], xc[RW + 2], xp[RW + 2];
  float n2p[RW], d3p[RW + 1];
#pragma unroll
  for (int q = 0; q < RW + 2; ++q) xm[q] = zero2;
  if (r0 > 0) load_window(r0 - 1, 0u, prd - W, pmk ? pmk - W : nullptr, xm);
  load_window(r0, rowB, prd, pmk, xc);
  unsigned sx = rowB;                                                      // byte offset of the current row
#pragma unroll
  for (int k = 0; k < RW; ++k) n2p[k] = (p2 - W)[dq[k + 1]];
#pragma unroll
  for (int k = 0; k < RW + 1; ++k) d3p[k] = (ND == 4) ? (p3 - W)[dq[k + 1]] : 0.0f;

  constexpr int CWR = (FUSE == F_RESTRICT) ? (RW - 1) / 2 : 1;  // coarse columns of an F_RESTRICT strip
  v2f racc[CWR], rnext[CWR];
#pragma unroll
  for (int j = 0; j < CWR; ++j) racc[j] = rnext[j] = zero2;
  const int cI0 = (r0 + 1) >> 1, cJ0 = (c0w + 1) >> 1;          // F_RESTRICT: first coarse row / column

  for (int row = r0; row < r1; ++row) {
    if (row + 1 < nyp) {
      load_window(row + 1, sx + rowB, prd + W, pmk ? pmk + W : nullptr, xp);
    } else {
#pragma unroll
      for (int q = 0; q < RW + 2; ++q) xp[q] = zero2;
    }
    float d0[RW], e1[RW + 1], n2c[RW], d3c[RW + 1];
    v2f resrow[(FUSE == F_RESTRICT) ? RW : 1];
    if (FUSE == F_RESTRICT) {
#pragma unroll
      for (int k = 0; k < RW; ++k) resrow[k] = zero2;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      d0[k] = p0[dq[k + 1]];
      n2c[k] = p2[dq[k + 1]];
    }
#pragma unroll
    for (int k = 0; k < RW + 1; ++k) {
      const int dc = TAIL ? dq[k] : k - 1;  // east coupling of column c0w - 1 + k (TAIL covers c0w < 1: clamped)
      e1[k] = p1[dc];
      d3c[k] = (ND == 4) ? p3[dq[k + 1]] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const int q = k + 1;
      if (TAIL && (c0w + k >= W || c0w + k < 0)) continue;
      if (MODE == M_JACOBI) {
        // unit form: bu = b / s_b; XFROMB: the window holds x1 = omega_in rd0 bu, so bu = x1 d0 / omega_in
        v2f braw = zero2, res;
        if (XFROMB) {
          res = xc[q] * (d0[k] * inv_omega_in);
          if (DOT) braw = res * sb;
        } else {
          braw = bld(rb, offq[q], sx);
          res = braw * ib;
        }
        res -= d0[k] * xc[q];
        res -= e1[k + 1] * xc[q + 1];
        res -= e1[k] * xc[q - 1];
        res -= n2c[k] * xp[q];
        res -= n2p[k] * xm[q];
        if (ND == 4) {
          res -= d3c[k] * xp[q - 1];
          res -= d3p[k + 1] * xm[q + 1];
        }
        const v2f xo = xc[q] + (omega * prd[dq[q]]) * res;
        if (BST) bst(ro, offq[q], sx, xo);
        else *(v2f*)((char*)po + offq[q]) = xo;
        if (DOT) {
          const v2f pr = braw * xo;
          s0 += (double)pr.x;
          s1 += (double)pr.y;
        }
      } else {  // M_RESID (+ F_RESTRICT): r = b - s_b (K_1 x)
        v2f acc = d0[k] * xc[q];
        acc += e1[k + 1] * xc[q + 1];
        acc += e1[k] * xc[q - 1];
        acc += n2c[k] * xp[q];
        acc += n2p[k] * xm[q];
        if (ND == 4) {
          acc += d3c[k] * xp[q - 1];
          acc += d3p[k + 1] * xm[q + 1];
        }
        const v2f res = bld(rb, offq[q], sx) - sb * acc;
        if (FUSE == F_RESTRICT) resrow[k] = res;
        else if (BST) bst(ro, offq[q], sx, res);
        else *(v2f*)((char*)po + offq[q]) = res;
      }
    }
    if (FUSE == F_RESTRICT) {
      // strip column k <-> fine column 2 cJ0 - 1 + k, so coarse column cJ0 + j sits at k = 2 j + 1.
      // Full weighting of the P1 lattice: centre 1; W, E, N, S, NE-of-the-row-above, SW-of-the-row-below 1/2.
      const bool store = (row & 1) || row + 1 >= nyp;  // coarse row complete after its odd row (or at the last row)
      if (!(row & 1)) {
#pragma unroll
        for (int j = 0; j < CWR; ++j) racc[j] += resrow[2 * j + 1] + 0.5f * (resrow[2 * j] + resrow[2 * j + 2]);
      } else {
#pragma unroll
        for (int j = 0; j < CWR; ++j) {
          racc[j] += 0.5f * (resrow[2 * j + 1] + resrow[2 * j]);
          rnext[j] = 0.5f * (resrow[2 * j + 1] + resrow[2 * j + 2]);
        }
      }
      if (store) {
        const int I = row >> 1;
        if (I >= cI0) {
#pragma unroll
          for (int j = 0; j < CWR; ++j) {
            const int J = cJ0 + j;
            if (J < ex.cW) {
              const i64 Ic = (i64)I * ex.cW + J;
              st2(out + Ic * Bp, lb, ex.bc[Ic] ? zero2 : racc[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CWR; ++j) {
          racc[j] = rnext[j];
          rnext[j] = zero2;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RW + 2; ++q) {
      xm[q] = xc[q];
      xc[q] = xp[q];
    }
#pragma unroll
    for (int k = 0; k < RW; ++k) n2p[k] = n2c[k];
#pragma unroll
    for (int k = 0; k < RW + 1; ++k) d3p[k] = d3c[k];
    p0 += W; p1 += W; p2 += W; p3 += W; prd += W;
    if (FUSE == F_PROLONG) pmk += W;
    sx += rowB;
    if (!BST && po) po += (i64)W * Bp;
  }
}

template <int MODE, int FUSE, int ND, bool XFROMB, int RW, bool DOT, bool BST>
__global__ __launch_bounds__(256) void dia_strip2_kernel(Level L, const double* __restrict__ scale,
                                                          const float* __restrict__ xin, const float* __restrict__ bvec,
                                                          float* __restrict__ out, float omega, float omega_in, Extra ex,
                                                          double* __restrict__ part, int Bp, int ncb, int TR) {
  __shared__ double lds[4 * kWave];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = blockIdx.y * (2 * kWave) + 2 * lane;   // first of this lane's two samples
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int rc = tile / ncb, cb = tile - rc * ncb;
  const int nyp = L.ny + 1;
  int c0w, r0, r1;
  bool active;
  if (FUSE == F_RESTRICT) {  // TR counts COARSE rows, the wave owns (RW - 1) / 2 coarse columns
    const int J0 = (cb * 4 + wave) * ((RW - 1) / 2), I0 = rc * TR;
    const int cnyp = (nyp + 1) >> 1;
    const int I1 = (I0 + TR < cnyp) ? I0 + TR : cnyp;
    c0w = 2 * J0 - 1;
    r0 = I0 > 0 ? 2 * I0 - 1 : 0;
    r1 = (2 * I1 < nyp) ? 2 * I1 : nyp;
    active = J0 < ex.cW && I0 < I1;
  } else {
    c0w = (cb * 4 + wave) * RW;
    r0 = rc * TR;
    r1 = (r0 + TR < nyp) ? r0 + TR : nyp;
    active = c0w < L.W && r0 < r1;
  }
  v2f sb = {1.0f, 1.0f};
  if (scale) {
    sb.x = (float)scale[lb];
    sb.y = (float)scale[lb + 1];
  }
  const v2f ib = 1.0f / sb;
  const float* __restrict__ src = XFROMB ? bvec : xin;
  double s0 = 0.0, s1 = 0.0;
  if (active) {
    if (c0w + RW + 1 > L.W || c0w < 1)    // strips that touch the left or right edge: clamped window columns
      strip2_body<MODE, FUSE, ND, XFROMB, RW, true, DOT, BST>(L, ib, sb, src, bvec, out, omega, omega_in, ex, Bp, lb, c0w, r0, r1,
                                                         s0, s1);
    else
      strip2_body<MODE, FUSE, ND, XFROMB, RW, false, DOT, BST>(L, ib, sb, src, bvec, out, omega, omega_in, ex, Bp, lb, c0w, r0,
                                                          r1, s0, s1);
  }
  if (DOT) {
    const double t0 = block_sum_per_sample(s0, Bp, lds);
    const double t1 = block_sum_per_sample(s1, Bp, lds);
    if (wave == 0) {
      part[(i64)blockIdx.x * Bp + lb] = t0;
      part[(i64)blockIdx.x * Bp + lb + 1] = t1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// FUSED two-stage passes of the fp32 V-cycle (batch-shared matrix), round 3.
//
// The four strip passes of a level -- first two sweeps, residual + restriction, prolongation + sweep, sweep -- read the
// right-hand side four times and write / re-read two intermediate iterates: 42 B per node and sample, all of it HBM
// traffic, at the HBM rate (section 6 of DESIGN.md: these kernels run at 4.4-5.2 TB/s of REAL traffic; their inner
// loops are not the limit).  The packed-fp32 form leaves most of the issue slots idle, so they are spent on
// RECOMPUTATION instead: two chained stencil stages per pass, the intermediate iterate kept in registers on a
// one-column / one-row wider window and never stored.
//   PRE : x2 = two sweeps from 0, coarse rhs = R (r - A x2)       reads r; writes x2 and the coarse rhs:     9 B  (was 17)
//   POST: z  = two sweeps on (x2 + P e)                           reads x2, r, e; writes z:                 13 B  (was 25)
// 22 instead of 42 B per node and sample and cycle.  Same arithmetic per node as the unfused kernels (unit form, packed
// fp32), evaluated once more on the halo ring; results agree with them to fp32 rounding (different association only).
// A wave owns its columns for both stages; VT = v2f (two samples per lane) or float (one).
// ---------------------------------------------------------------------------------------------
// per-sample sum over the NW waves of a block (lanes hold distinct samples); valid in wave 0.  NW == 4: the same order
// of additions as block_sum_per_sample (results of the default geometry stay bitwise what they were)
template <int NW>
__device__ __forceinline__ double block_sum_waves(double v, double* lds /* >= NW * 64 doubles */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  lds[wave * kWave + lane] = v;
  __syncthreads();
  double s = 0.0;
  if (wave == 0) {
    if (NW == 4) {
      s = (lds[lane] + lds[kWave + lane]) + (lds[2 * kWave + lane] + lds[3 * kWave + lane]);
    } else {
#pragma unroll
      for (int w = 0; w < NW; ++w) s += lds[w * kWave + lane];
    }
  }
  __syncthreads();
  return s;
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Acc { double v[4] = {0.0, 0.0, 0.0, 0.0}; };   // per-sample dot-product accumulators of one lane (kSpl used)

template <typename VT> struct VLane;
template <> struct VLane<float> {
  static constexpr int kSpl = 1;
  static __device__ __forceinline__ float zero() { return 0.0f; }
  static __device__ __forceinline__ float ld(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  }
  static __device__ __forceinline__ float from_scale(const double* __restrict__ s, unsigned lb) { return s ? (float)s[lb] : 1.0f; }
  static __device__ __forceinline__ void dot(Acc& s, float a, float b) { s.v[0] += (double)(a * b); }
};
template <> struct VLane<v2f> {
  static constexpr int kSpl = 2;
  static __device__ __forceinline__ v2f zero() { return v2f{0.0f, 0.0f}; }
  static __device__ __forceinline__ v2f ld(rsrc_t r, unsigned voff, unsigned soff) { return bld(r, voff, soff); }
  static __device__ __forceinline__ v2f from_scale(const double* __restrict__ s, unsigned lb) {
    return s ? v2f{(float)s[lb], (float)s[lb + 1]} : v2f{1.0f, 1.0f};
  }
  static __device__ __forceinline__ void dot(Acc& s, v2f a, v2f b) {
    const v2f p = a * b;
    s.v[0] += (double)p.x;
    s.v[1] += (double)p.y;
  }
};
// FOUR samples per lane, 256 per wave: one 16-byte access per lane and node -- half the vector-memory instructions per
// byte of the two-sample form (the fused passes are bound by the NUMBER of those instructions),
// twice the registers per lane (2 waves per SIMD instead of 4: the same bytes in flight per SIMD).
template <> struct VLane<v4f> {
  static constexpr int kSpl = 4;
  static __device__ __forceinline__ v4f zero() { return v4f{0.0f, 0.0f, 0.0f, 0.0f}; }
  static __device__ __forceinline__ v4f ld(rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  }
  static __device__ __forceinline__ v4f from_scale(const double* __restrict__ s, unsigned lb) {
    return s ? v4f{(float)s[lb], (float)s[lb + 1], (float)s[lb + 2], (float)s[lb + 3]} : v4f{1.0f, 1.0f, 1.0f, 1.0f};
  }
  static __device__ __forceinline__ void dot(Acc& s, v4f a, v4f b) {
    const v4f p = a * b;
    s.v[0] += (double)p.x;
    s.v[1] += (double)p.y;
    s.v[2] += (double)p.z;
    s.v[3] += (double)p.w;
  }
};

// A vector stream of the fused kernels: buffer-resource addressing (raw_buffer_load, 32-bit per-lane offset + uniform
// SGPR row offset) or, with -DDIFFHE_FLAT_LD=1, global loads off a wave-uniform 64-bit base (SGPR pair, the row offset
// added by the scalar unit) + the 32-bit per-lane offset (the form strip_body uses).  A/B switch of round 4: the PMC
// counters show the texture-addresser FIFOs full 28-35 % of the time in the buffer-load kernels and never in strip_body's.
#ifndef DIFFHE_FLAT_LD
#define DIFFHE_FLAT_LD 0
#endif
struct Src {
  rsrc_t r;
  const char* p;
};
__device__ __forceinline__ Src make_src(const void* p) { return Src{make_rsrc(p), (const char*)p}; }
template <typename VT>
__device__ __forceinline__ VT ldsrc(const Src& s, unsigned voff, unsigned soff) {
  if constexpr (DIFFHE_FLAT_LD != 0) return *(const VT*)((s.p + (size_t)soff) + (size_t)voff);
  else return VLane<VT>::ld(s.r, voff, soff);
}

// Where the matrix coefficients of the fused passes come from.
//   SHARED: batch-shared fp32 copies + reciprocal diagonal, wave-uniform scalar loads (values are plain floats);
//   per sample: fp32 diagonal + scaled fp16 off-diagonals (Level.v32 / o16 / osc), one value per sample and lane,
//   buffer loads with tile-relative offsets; the reciprocal diagonal is v_rcp_f32 of the loaded diagonal.
template <typename VT, bool SHARED> struct Coef;
template <typename VT> struct Coef<VT, true> {
  typedef float T;
  const float *v0, *v1, *v2, *v3, *rdp;
  __device__ __forceinline__ Coef(const Level& L, i64, unsigned, int) : v0(L.v32), v1(L.v32 + L.n), v2(L.v32 + 2 * (i64)L.n),
                                                                        v3(L.v32 + 3 * (i64)L.n), rdp(L.rd32) {}
  __device__ __forceinline__ T d(i64 i) const { return v0[i]; }
  __device__ __forceinline__ T e(i64 i) const { return v1[i]; }
  __device__ __forceinline__ T n2(i64 i) const { return v2[i]; }
  __device__ __forceinline__ T q3(i64 i) const { return v3[i]; }
  __device__ __forceinline__ T rd(i64 i, T) const { return rdp[i]; }
};
__device__ __forceinline__ float ldh(rsrc_t r, unsigned voff, float) {
  return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, 0));
}
__device__ __forceinline__ v2f ldh(rsrc_t r, unsigned voff, v2f) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_bit_cast(h2, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
  return v2f{(float)h.x, (float)h.y};
}
__device__ __forceinline__ unsigned ldraw(rsrc_t r, unsigned voff, float) {
  return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, voff, 0, 0);
}
__device__ __forceinline__ unsigned ldraw(rsrc_t r, unsigned voff, v2f) {
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
}
__device__ __forceinline__ float unraw(unsigned raw, float) { return (float)__builtin_bit_cast(_Float16, (unsigned short)raw); }
__device__ __forceinline__ v2f unraw(unsigned raw, v2f) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_bit_cast(h2, raw);
  return v2f{(float)h.x, (float)h.y};
}
template <typename VT> struct Coef<VT, false> {
  typedef VT T;
  rsrc_t r0, r1, r2, r3;
  i64 base;       // node index the resources are based at (<= every index the tile touches)
  unsigned lb, Bp;
  VT osc;         // this lane's sample scale(s) of the fp16 couplings
  __device__ __forceinline__ Coef(const Level& L, i64 base_, unsigned lb_, int Bp_)
      : base(base_), lb(lb_), Bp((unsigned)Bp_), osc(VLane<VT>::from_scale(L.osc, lb_)) {
    const i64 n = L.n;
    r0 = make_rsrc(L.v32 + base * Bp_);
    r1 = make_rsrc(L.o16 + base * Bp_);
    r2 = make_rsrc(L.o16 + (n + base) * Bp_);
    r3 = make_rsrc(L.o16 + (2 * n + base) * Bp_);
  }
  __device__ __forceinline__ unsigned off(i64 i) const { return (unsigned)(i - base) * Bp + lb; }
  __device__ __forceinline__ T d(i64 i) const { return VLane<VT>::ld(r0, 4u * off(i), 0u); }
  __device__ __forceinline__ T e(i64 i) const { return osc * ldh(r1, 2u * off(i), VT{}); }
  __device__ __forceinline__ T n2(i64 i) const { return osc * ldh(r2, 2u * off(i), VT{}); }
  __device__ __forceinline__ T q3(i64 i) const { return osc * ldh(r3, 2u * off(i), VT{}); }
  __device__ __forceinline__ T rd(i64, T dv) const { return 1.0f / dv; }
  // raw fp16 storage words (one per sample of the lane), for the register-cached coefficient rows of the fused POST pass
  __device__ __forceinline__ unsigned e_raw(i64 i) const { return ldraw(r1, 2u * off(i), VT{}); }
  __device__ __forceinline__ unsigned n2_raw(i64 i) const { return ldraw(r2, 2u * off(i), VT{}); }
  __device__ __forceinline__ T cvt(unsigned raw) const { return osc * unraw(raw, VT{}); }
};

// K_1 x at the NC columns col0 .. col0 + NC - 1 of grid row R, handed column by column to `use(k, K1x, d0, rd)`.
// xm / xc / xp hold x on rows R - 1 / R / R + 1 at the NC + 2 columns col0 - 1 .. col0 + NC (index j <-> column
// col0 - 1 + j); out-of-grid positions must hold 0.  EDGE: the strip / tile touches a grid edge, so the coefficient
// indices of non-existent couplings are clamped into the arrays (their values meet a zero x).
template <typename VT, int NC, int ND, bool EDGE, typename CF, typename F>
__device__ __forceinline__ void k1_row(const CF& cf, i64 n, int W, int R, int col0, const VT* xm, const VT* xc,
                                       const VT* xp, F&& use) {
  const i64 base = (i64)R * W + col0;
  auto at = [&](i64 i) -> i64 { return EDGE ? (i < 0 ? 0 : (i > n - 1 ? n - 1 : i)) : i; };
  typename CF::T ew = cf.e(at(base - 1));       // west coupling of the first column; then carried along the row
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const i64 i = base + k;
    const typename CF::T d0 = cf.d(at(i));
    const typename CF::T ee = cf.e(at(i));
    VT acc = d0 * xc[k + 1];
    acc += ee * xc[k + 2];                       // east  (R, c) - (R, c + 1)
    acc += ew * xc[k];                           // west
    acc += cf.n2(at(i)) * xp[k + 1];             // north (R, c) - (R + 1, c)
    acc += cf.n2(at(i - W)) * xm[k + 1];         // south
    if (ND == 4) {
      acc += cf.q3(at(i)) * xp[k];               // (R, c) - (R + 1, c - 1)
      acc += cf.q3(at(i - W + 1)) * xm[k + 2];   // (R - 1, c + 1) - (R, c)
    }
    use(k, acc, d0, cf.rd(at(i), d0));
    ew = ee;
  }
}

// ---- PRE: first two sweeps from a zero guess + residual + full-weighting restriction -------------------------------
// Geometry of the F_RESTRICT strips: the wave owns CW coarse columns J0 .. J0 + CW - 1, i.e. the RW = 2 CW + 1 fine
// residual columns c0w = 2 J0 - 1 .. 2 J0 + 2 CW - 1 (the last one shared with -- and recomputed by -- the next strip),
// and stores x2 on the first 2 CW of them; tile rows: coarse I0 .. I1 - 1 = fine residual rows r0 .. r1 - 1
// (r0 = 2 I0 - 1, r1 = 2 I1), x2 stored on rows r0 .. r1 - 2 (all the way up on the last tile).
template <typename VT, int ND, int CW, bool EDGE, bool SHARED>
__device__ __forceinline__ void fused_pre_body(const Level& L, VT ib, VT sb, const float* __restrict__ rhs,
                                               float* __restrict__ x2out, float* __restrict__ crhs, float w0, float w1,
                                               int cW, const unsigned char* __restrict__ cbc, int Bp, unsigned lb, int c0w,
                                               int r0, int r1) {
  constexpr int RW = 2 * CW + 1;
  constexpr int N1 = RW + 4, N2 = RW + 2;    // columns of the x1 / x2 windows: c0w - 2 + j / c0w - 1 + j
  const int W = L.W, nyp = L.ny + 1;
  const i64 n = L.n;
  const VT Z = VLane<VT>::zero();
  typedef Coef<VT, SHARED> CF;
  i64 cbase = (i64)(r0 - 3) * W;             // coefficient resources: based below everything the tile touches
  if (cbase < 0) cbase = 0;
  const CF cf(L, cbase, lb, Bp);
  bool ok1[N1];
  unsigned off1[N1];
#pragma unroll
  for (int j = 0; j < N1; ++j) {
    int c = c0w - 2 + j;
    ok1[j] = !EDGE || (c >= 0 && c < W);
    if (EDGE) c = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
    off1[j] = 4u * ((unsigned)(c - (c0w - 2) + 2) * (unsigned)Bp + lb);   // base sits two columns further left
  }
  // base: element (r0 - 2, c0w - 4): every offset below is non-negative
  const i64 tile0 = ((i64)(r0 - 2) * W + (c0w - 4)) * Bp;
  const Src rr = make_src(rhs + tile0);
  const unsigned rowB = 4u * (unsigned)W * (unsigned)Bp;
  const float inv_w0 = 1.0f / w0;

  // x1 on grid row R (window N1): w0 rd (r ib); 0 outside the grid
  auto x1_row = [&](int R, VT* dst) {
    if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
      for (int j = 0; j < N1; ++j) dst[j] = Z;
      return;
    }
    const unsigned sx = (unsigned)(R - (r0 - 2)) * rowB;
    const i64 rb = (i64)R * W + (c0w - 2);
#pragma unroll
    for (int j = 0; j < N1; ++j) {
      const VT v = ldsrc<VT>(rr, off1[j], sx);
      i64 i = rb + j;
      if (EDGE) i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
      const typename CF::T dv = SHARED ? typename CF::T{} : cf.d(i);
      dst[j] = ok1[j] ? (v * ib) * (w0 * cf.rd(i, dv)) : Z;
    }
  };
  // x2 on grid row R (window N2) from x1 rows R - 1, R, R + 1
  auto x2_row = [&](int R, const VT* am, const VT* ac, const VT* ap, VT* dst) {
    if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
      for (int j = 0; j < N2; ++j) dst[j] = Z;
      return;
    }
    k1_row<VT, N2, ND, EDGE>(cf, n, W, R, c0w - 1, am, ac, ap, [&](int j, VT kx, typename CF::T d0, typename CF::T rd) {
      const VT bu = ac[j + 1] * (d0 * inv_w0);              // x1 = w0 rd bu  ->  bu = x1 d0 / w0
      const VT v = ac[j + 1] + (w1 * rd) * (bu - kx);
      dst[j] = ok1[j + 1] ? v : Z;
    });
  };

  VT a0[N1], a1[N1], a2[N1];   // x1 rows R - 1, R, R + 1 of the x2 row being formed
  VT b0[N2], b1[N2], b2[N2];   // x2 rows row - 1, row, row + 1
  if constexpr (!SHARED && ND == 3) {
    // Per-sample coefficients: every coefficient row loaded ONCE into a register window (as in fused_post_body): the
    // diagonal when the row's x1 is formed (9 columns), its couplings one iteration later for the x2 stage (raw fp16
    // words), both kept one more iteration for the residual stage; the row below contributes its north couplings.
    struct CR { VT d[N2]; unsigned e[N2 + 1]; unsigned n[N2]; };   // columns c0w - 1 + j; e[t] = east coupling of column c0w - 2 + t
    auto at = [&](i64 i) -> i64 { return EDGE ? (i < 0 ? 0 : (i > n - 1 ? n - 1 : i)) : i; };
    auto load_d = [&](int R, VT* D) {          // diagonal of row R on the N1 columns c0w - 2 + j
      const i64 base = (i64)R * W + (c0w - 2);
#pragma unroll
      for (int j = 0; j < N1; ++j) D[j] = cf.d(at(base + j));
    };
    auto load_en = [&](int R, const VT* D, CR& c) {
      const i64 base = (i64)R * W + (c0w - 2);
#pragma unroll
      for (int t = 0; t < N2 + 1; ++t) c.e[t] = cf.e_raw(at(base + t));
#pragma unroll
      for (int j = 0; j < N2; ++j) {
        c.n[j] = cf.n2_raw(at(base + 1 + j));
        c.d[j] = D[j + 1];
      }
    };
    auto x1c = [&](int R, const VT* D, VT* dst) {
      if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
        for (int j = 0; j < N1; ++j) dst[j] = Z;
        return;
      }
      const unsigned sx = (unsigned)(R - (r0 - 2)) * rowB;
#pragma unroll
      for (int j = 0; j < N1; ++j) {
        const VT v = ldsrc<VT>(rr, off1[j], sx);
        dst[j] = ok1[j] ? (v * ib) * (w0 * (1.0f / D[j])) : Z;
      }
    };
    auto k1c = [&](auto nc_tag, auto off_tag, const CR& c, const unsigned* sn, const VT* xm, const VT* xc, const VT* xq,
                   auto&& use) {
      constexpr int NC = decltype(nc_tag)::value, OFF = decltype(off_tag)::value;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int j = k + OFF;
        const VT d0 = c.d[j];
        VT acc = d0 * xc[k + 1];
        acc += cf.cvt(c.e[j + 1]) * xc[k + 2];
        acc += cf.cvt(c.e[j]) * xc[k];
        acc += cf.cvt(c.n[j]) * xq[k + 1];
        acc += cf.cvt(sn[j]) * xm[k + 1];
        use(k, acc, d0, 1.0f / d0);
      }
    };
    typedef std::integral_constant<int, N2> tN2;
    typedef std::integral_constant<int, RW> tRW;
    typedef std::integral_constant<int, 0> t0;
    typedef std::integral_constant<int, 1> t1;
    auto x2c = [&](int R, const CR& c, const unsigned* sn, const VT* am, const VT* ac, const VT* ap, VT* dst) {
      if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
        for (int j = 0; j < N2; ++j) dst[j] = Z;
        return;
      }
      k1c(tN2{}, t0{}, c, sn, am, ac, ap, [&](int j, VT kx, VT d0, VT rd) {
        const VT bu = ac[j + 1] * (d0 * inv_w0);
        const VT v = ac[j + 1] + (w1 * rd) * (bu - kx);
        dst[j] = ok1[j + 1] ? v : Z;
      });
    };
    VT Dq[N1], Dn[N1];           // diagonals of the newest two x1 rows
    CR cA, cB;                   // coefficient rows of the x2 row being formed / of the residual row
    unsigned sS[N2];             // north couplings of the row below the residual row
    load_d(r0 - 2, Dq);
    x1c(r0 - 2, Dq, a0);
    load_en(r0 - 2, Dq, cB);     // only its n is used: south of row r0 - 1
    load_d(r0 - 1, Dq);
    x1c(r0 - 1, Dq, a1);
    load_en(r0 - 1, Dq, cA);
    load_d(r0, Dn);
    x1c(r0, Dn, a2);
    x2c(r0 - 1, cA, cB.n, a0, a1, a2, b0);
#pragma unroll
    for (int j = 0; j < N2; ++j) sS[j] = cA.n[j];      // n of row r0 - 1
#pragma unroll
    for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
    load_d(r0 + 1, Dq);
    x1c(r0 + 1, Dq, a2);
    load_en(r0, Dn, cB);         // coefficient row r0
    x2c(r0, cB, sS, a0, a1, a2, b1);
    // loop invariant at the top of iteration `row`: cB = coefficient row `row`, sS = n of row - 1, Dq = diagonal of row + 1

    VT racc[CW], rnext[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) racc[j] = rnext[j] = Z;
    const int cI0 = (r0 + 1) >> 1, cJ0 = (c0w + 1) >> 1;
    const int last_store = (r1 >= nyp) ? nyp - 1 : r1 - 2;
    float* __restrict__ px2 = x2out + ((i64)r0 * W + c0w) * Bp;
    for (int row = r0; row < r1; ++row) {
#pragma unroll
      for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
      load_d(row + 2, Dn);
      x1c(row + 2, Dn, a2);
      load_en(row + 1, Dq, cA);
      x2c(row + 1, cA, cB.n, a0, a1, a2, b2);
      VT res[RW];
      k1c(tRW{}, t1{}, cB, sS, b0, b1, b2, [&](int k, VT kx, VT d0, VT) {
        const VT bu = a0[k + 2] * (d0 * inv_w0);
        res[k] = (!EDGE || (c0w + k >= 0 && c0w + k < W)) ? bu - kx : Z;
      });
      if (row <= last_store) {
#pragma unroll
        for (int k = 0; k < RW - 1; ++k) {
          if (!EDGE || (c0w + k >= 0 && c0w + k < W)) *(VT*)(px2 + (i64)k * Bp + lb) = b1[k + 1];
        }
      }
      px2 += (i64)W * Bp;
      const bool store = (row & 1) || row + 1 >= nyp;
      if (!(row & 1)) {
#pragma unroll
        for (int j = 0; j < CW; ++j) racc[j] += res[2 * j + 1] + 0.5f * (res[2 * j] + res[2 * j + 2]);
      } else {
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          racc[j] += 0.5f * (res[2 * j + 1] + res[2 * j]);
          rnext[j] = 0.5f * (res[2 * j + 1] + res[2 * j + 2]);
        }
      }
      if (store) {
        const int I = row >> 1;
        if (I >= cI0) {
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            const int J = cJ0 + j;
            if (J < cW) {
              const i64 Ic = (i64)I * cW + J;
              *(VT*)(crhs + Ic * Bp + lb) = cbc[Ic] ? Z : sb * racc[j];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < CW; ++j) { racc[j] = rnext[j]; rnext[j] = Z; }
      }
#pragma unroll
      for (int j = 0; j < N2; ++j) { b0[j] = b1[j]; b1[j] = b2[j]; sS[j] = cB.n[j]; }
      cB = cA;
#pragma unroll
      for (int j = 0; j < N1; ++j) Dq[j] = Dn[j];
    }
    return;
  }
  x1_row(r0 - 2, a0);
  x1_row(r0 - 1, a1);
  x1_row(r0, a2);
  x2_row(r0 - 1, a0, a1, a2, b0);
#pragma unroll
  for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
  x1_row(r0 + 1, a2);
  x2_row(r0, a0, a1, a2, b1);

  VT racc[CW], rnext[CW];
#pragma unroll
  for (int j = 0; j < CW; ++j) racc[j] = rnext[j] = Z;
  const int cI0 = (r0 + 1) >> 1, cJ0 = (c0w + 1) >> 1;
  const int last_store = (r1 >= nyp) ? nyp - 1 : r1 - 2;
  float* __restrict__ px2 = x2out + ((i64)r0 * W + c0w) * Bp;

  for (int row = r0; row < r1; ++row) {
    // x1 row + 2 -> x2 row + 1
#pragma unroll
    for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
    x1_row(row + 2, a2);
    x2_row(row + 1, a0, a1, a2, b2);
    // residual of row `row` on the RW columns c0w .. c0w + RW - 1 (unit form, times s_b at the store)
    VT res[RW];
    k1_row<VT, RW, ND, EDGE>(cf, n, W, row, c0w, b0, b1, b2, [&](int k, VT kx, typename CF::T d0, typename CF::T) {
      // bu at (row, c0w + k) from the x1 window kept for this row (a0 after the shift above = x1 row `row`)
      const VT bu = a0[k + 2] * (d0 * inv_w0);
      res[k] = (!EDGE || (c0w + k >= 0 && c0w + k < W)) ? bu - kx : Z;
    });
    // store x2 of this row on the owned columns
    if (row <= last_store) {
#pragma unroll
      for (int k = 0; k < RW - 1; ++k) {
        if (!EDGE || (c0w + k >= 0 && c0w + k < W)) *(VT*)(px2 + (i64)k * Bp + lb) = b1[k + 1];
      }
    }
    px2 += (i64)W * Bp;
    // full weighting, as in strip2_body
    const bool store = (row & 1) || row + 1 >= nyp;
    if (!(row & 1)) {
#pragma unroll
      for (int j = 0; j < CW; ++j) racc[j] += res[2 * j + 1] + 0.5f * (res[2 * j] + res[2 * j + 2]);
    } else {
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        racc[j] += 0.5f * (res[2 * j + 1] + res[2 * j]);
        rnext[j] = 0.5f * (res[2 * j + 1] + res[2 * j + 2]);
      }
    }
    if (store) {
      const int I = row >> 1;
      if (I >= cI0) {
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const int J = cJ0 + j;
          if (J < cW) {
            const i64 Ic = (i64)I * cW + J;
            *(VT*)(crhs + Ic * Bp + lb) = cbc[Ic] ? Z : sb * racc[j];
          }
        }
      }
#pragma unroll
      for (int j = 0; j < CW; ++j) { racc[j] = rnext[j]; rnext[j] = Z; }
    }
#pragma unroll
    for (int j = 0; j < N2; ++j) { b0[j] = b1[j]; b1[j] = b2[j]; }
  }
}

// NW = waves per block (a block owns NW * CW adjacent coarse columns).  Round 4 measured 8 and 16 against 4 on the
// 1024^2 x 256 bench (gpurun_out/r4c): WIDER blocks read MORE from the fabric, not less (POST 1.56 -> 1.65 / 1.66 read
// passes: the waves of a larger block drift apart and miss each other's halo lines) and run slower (PRE 0.708 -> 0.740 /
// 0.818 ms, POST 1.000 -> 0.978 / 1.101 ms, step 81.7 -> 82.6 / 86.6 ms).  4 stays; the parameter documents the experiment.
template <typename VT, int ND, int CW, bool SHARED, int NW = 4, int MW = (SHARED ? 4 : 1)>
__global__ __launch_bounds__(64 * NW, MW) void fused_pre_kernel(Level L, const double* __restrict__ scale,
                                                         const float* __restrict__ rhs, float* __restrict__ x2out,
                                                         float* __restrict__ crhs, float w0, float w1, int cW,
                                                         const unsigned char* __restrict__ cbc, int Bp, int ncb, int TR) {
  constexpr int SPL = VLane<VT>::kSpl;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = blockIdx.y * (SPL * kWave) + SPL * lane;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int rc = tile / ncb, cb = tile - rc * ncb;
  const int nyp = L.ny + 1;
  const int J0 = (cb * NW + wave) * CW, I0 = rc * TR;
  const int cnyp = (nyp + 1) >> 1;
  const int I1 = (I0 + TR < cnyp) ? I0 + TR : cnyp;
  if (!(J0 < cW && I0 < I1)) return;
  const int c0w = 2 * J0 - 1;
  const int r0 = I0 > 0 ? 2 * I0 - 1 : 0;
  const int r1 = (2 * I1 < nyp) ? 2 * I1 : nyp;
  const VT sb = VLane<VT>::from_scale(scale, lb);
  const VT ib = 1.0f / sb;
  constexpr int RW = 2 * CW + 1;
  // interior tiles: every window column (c0w - 2 .. c0w + RW + 1) and row (r0 - 2 .. r1 + 1) lies inside the grid
  const bool edge = c0w - 2 < 0 || c0w + RW + 1 > L.W - 1 || r0 - 2 < 0 || r1 + 1 > nyp - 1;
  if (edge) fused_pre_body<VT, ND, CW, true, SHARED>(L, ib, sb, rhs, x2out, crhs, w0, w1, cW, cbc, Bp, lb, c0w, r0, r1);
  else fused_pre_body<VT, ND, CW, false, SHARED>(L, ib, sb, rhs, x2out, crhs, w0, w1, cW, cbc, Bp, lb, c0w, r0, r1);
}

// ---- POST: prolongation + correction + both post-smoothing sweeps (+ the partials of rhs . z) ------------------------
// The wave owns the RW columns c0w .. c0w + RW - 1 (c0w a multiple of RW, even); rows r0 .. r1 - 1.
// Round 4 built and measured a variant in which the 4 waves of a block EXCHANGE their halo columns through LDS instead of
// each loading (and prolongating) them again: 12 instead of 23 vector-memory loads per wave and fine row, x' and r / s_b
// written to a two-slot exchange area, ONE workgroup barrier per row.  Correct (28 GPU tests, same iteration counts) and
// TWICE as slow: 0.999 -> 2.035 ms per fine-level launch at 1024^2 x 256 (gpurun_out/r4o).  The barrier puts the block's
// waves in lockstep, and these passes live on their waves being at DIFFERENT points of the row loop (one wave's loads in
// flight under another's arithmetic); any block-cooperative staging of rows pays the same price.  Removed.
template <typename VT, int ND, int RW, bool EDGE, bool DOT, bool SHARED, bool XZ>
__device__ __forceinline__ void fused_post_body(const Level& L, VT ib, const float* __restrict__ xin,
                                                const float* __restrict__ rhs, const float* __restrict__ ec,
                                                float* __restrict__ zout, float wA, float wB, int cW, int Bp, unsigned lb,
                                                int c0w, int r0, int r1, Acc& acc) {
  constexpr int N1 = RW + 4, N2 = RW + 2;    // x' window: columns c0w - 2 + j; x3 window: c0w - 1 + j
  constexpr int NCE = RW / 2 + 3;            // coarse columns (c0w - 2) / 2 .. (c0w + RW + 1 + 1) / 2
  const int W = L.W, nyp = L.ny + 1;
  const i64 n = L.n;
  const VT Z = VLane<VT>::zero();
  typedef Coef<VT, SHARED> CF;
  i64 cbase = (i64)(r0 - 3) * W;
  if (cbase < 0) cbase = 0;
  const CF cf(L, cbase, lb, Bp);
  bool ok1[N1];
  unsigned off1[N1];
#pragma unroll
  for (int j = 0; j < N1; ++j) {
    int c = c0w - 2 + j;
    ok1[j] = !EDGE || (c >= 0 && c < W);
    if (EDGE) c = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
    off1[j] = 4u * ((unsigned)(c - (c0w - 2) + 2) * (unsigned)Bp + lb);
  }
  const int cj0 = (c0w >> 1) - 1;            // first coarse column of the window (c0w is even)
  unsigned offc[NCE];
#pragma unroll
  for (int j = 0; j < NCE; ++j) {
    int cj = cj0 + j;
    cj = cj < 0 ? 0 : (cj > cW - 1 ? cW - 1 : cj);
    offc[j] = 4u * ((unsigned)cj * (unsigned)Bp + lb);
  }
  const i64 tile0 = ((i64)(r0 - 2) * W + (c0w - 4)) * Bp;
  const Src rx = make_src(XZ ? rhs + tile0 : xin + tile0);   // XZ: x = 0, never loaded
  const Src rr = make_src(rhs + tile0);
  const int cr0 = (r0 - 2 > 0 ? r0 - 2 : 0) >> 1;
  const Src rc = make_src(ec + (i64)cr0 * cW * Bp);
  const unsigned rowB = 4u * (unsigned)W * (unsigned)Bp, rowCB = 4u * (unsigned)cW * (unsigned)Bp;

  // x' = x + mask (P e) on grid row R
  auto xp_row = [&](int R, VT* dst) {
    if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
      for (int j = 0; j < N1; ++j) dst[j] = Z;
      return;
    }
    const unsigned sx = (unsigned)(R - (r0 - 2)) * rowB;
    const unsigned sc = (unsigned)((R >> 1) - cr0) * rowCB;
    // (Keeping the coarse row in registers between fine rows -- it is loaded three times, 7.5 of a row's 23 loads -- was
    // built and measured in round 4: + 10 live VGPRs spill (100 B of scratch at the 128-VGPR cap of the shared form, 16-100 B
    // at the 168 cap of the per-sample one): POST 1.000 -> 1.034 ms, per-element-field step 213.5 -> 232.0 ms; gpurun_out/r4i.)
    // ... and kept in a lane-private LDS ring instead (no barrier, no cross-lane traffic: LDS as a second register file
    // that bypasses the texture addresser; 2.5 instead of 7.5 memory loads per fine row): correct and 1.002 -> 1.417 ms --
    // LDS and scalar loads share one counter (lgkmcnt), and the waits for the ring serialise the coefficient loads of
    // both stages (gpurun_out/r4r).  Removed as well.
    VT ce[NCE], ce2[NCE];
#pragma unroll
    for (int j = 0; j < NCE; ++j) {
      ce[j] = ldsrc<VT>(rc, offc[j], sc);
      ce2[j] = (R & 1) ? ldsrc<VT>(rc, offc[j], sc + rowCB) : Z;
    }
    const i64 rb = (i64)R * W + (c0w - 2);
#pragma unroll
    for (int j = 0; j < N1; ++j) {
      VT corr;
      if (!(j & 1))                      // even window column <-> coarse column cj0 + j / 2
        corr = (R & 1) ? 0.5f * (ce[j / 2] + ce2[j / 2]) : ce[j / 2];
      else                               // between coarse columns cj0 + (j - 1) / 2 and + 1
        corr = (R & 1) ? 0.5f * (ce[(j + 1) / 2] + ce2[(j - 1) / 2]) : 0.5f * (ce[(j - 1) / 2] + ce[(j + 1) / 2]);
      i64 i = rb + j;
      if (EDGE) i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
      const VT v = XZ ? L.mk32[i] * corr : ldsrc<VT>(rx, off1[j], sx) + L.mk32[i] * corr;
      dst[j] = ok1[j] ? v : Z;
    }
  };
  // bu = r / s_b on grid row R at the N2 columns c0w - 1 + j
  auto bu_row = [&](int R, VT* dst) {
    if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
      for (int j = 0; j < N2; ++j) dst[j] = Z;
      return;
    }
    const unsigned sx = (unsigned)(R - (r0 - 2)) * rowB;
#pragma unroll
    for (int j = 0; j < N2; ++j) dst[j] = ok1[j + 1] ? ldsrc<VT>(rr, off1[j + 1], sx) * ib : Z;
  };
  // x3 on grid row R (window N2) from x' rows R - 1, R, R + 1 and bu row R
  auto x3_row = [&](int R, const VT* am, const VT* ac, const VT* ap, const VT* bu, VT* dst) {
    if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
      for (int j = 0; j < N2; ++j) dst[j] = Z;
      return;
    }
    k1_row<VT, N2, ND, EDGE>(cf, n, W, R, c0w - 1, am, ac, ap, [&](int j, VT kx, typename CF::T, typename CF::T rd) {
      const VT v = ac[j + 1] + (wA * rd) * (bu[j] - kx);
      dst[j] = ok1[j + 1] ? v : Z;
    });
  };

  VT a0[N1], a1[N1], a2[N1];   // x' rows
  VT b0[N2], b1[N2], b2[N2];   // x3 rows row - 1, row, row + 1
  VT u1[N2], u2[N2];           // bu rows row, row + 1
  if constexpr (!SHARED && ND == 3) {
    // Per-sample coefficients (fp32 diagonal + fp16 couplings, one value per lane and sample): every coefficient row is
    // needed four times -- by the x3 stage and the z stage, as the row's own couplings and as the south couplings of the
    // row above -- and re-loading it each time missed the caches about half the time (PMC: 4.54 passes of traffic for
    // 2.6 algorithmic, 5.4 TB/s: bandwidth-bound on wasted re-reads).  Rows are loaded ONCE into a register window, the
    // couplings kept as raw fp16 words (one VGPR per pair of samples).
    struct CRow { VT d[N2]; unsigned e[N2 + 1]; unsigned n[N2]; };   // columns c0w - 1 + j; e[j + 1] = east coupling of column j
    auto load_crow = [&](int R, CRow& c) {
      const i64 base = (i64)R * W + (c0w - 1);
      auto at = [&](i64 i) -> i64 { return EDGE ? (i < 0 ? 0 : (i > n - 1 ? n - 1 : i)) : i; };
      c.e[0] = cf.e_raw(at(base - 1));
#pragma unroll
      for (int j = 0; j < N2; ++j) {
        const i64 i = at(base + j);
        c.d[j] = cf.d(i);
        c.e[j + 1] = cf.e_raw(i);
        c.n[j] = cf.n2_raw(i);
      }
    };
    // K_1 x on NC columns starting at cached column OFF, row couplings c, south couplings sn (the n of the row below)
    auto k1c = [&](auto nc_tag, auto off_tag, const CRow& c, const unsigned* sn, const VT* xm, const VT* xc, const VT* xq,
                   auto&& use) {
      constexpr int NC = decltype(nc_tag)::value, OFF = decltype(off_tag)::value;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const int j = k + OFF;
        const VT d0 = c.d[j];
        VT acc = d0 * xc[k + 1];
        acc += cf.cvt(c.e[j + 1]) * xc[k + 2];
        acc += cf.cvt(c.e[j]) * xc[k];
        acc += cf.cvt(c.n[j]) * xq[k + 1];
        acc += cf.cvt(sn[j]) * xm[k + 1];
        use(k, acc, d0, 1.0f / d0);
      }
    };
    typedef std::integral_constant<int, N2> tN2;
    typedef std::integral_constant<int, RW> tRW;
    typedef std::integral_constant<int, 0> t0;
    typedef std::integral_constant<int, 1> t1;
    auto x3c = [&](int R, const CRow& c, const unsigned* sn, const VT* am, const VT* ac, const VT* ap, const VT* bu, VT* dst) {
      if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
        for (int j = 0; j < N2; ++j) dst[j] = Z;
        return;
      }
      k1c(tN2{}, t0{}, c, sn, am, ac, ap, [&](int j, VT kx, VT, VT rd) {
        const VT v = ac[j + 1] + (wA * rd) * (bu[j] - kx);
        dst[j] = ok1[j + 1] ? v : Z;
      });
    };
    CRow cS, cC, cN;             // coefficient rows R - 1, R, R + 1 of the x3 row being formed
    load_crow(r0 - 2, cS);
    load_crow(r0 - 1, cC);
    xp_row(r0 - 2, a0);
    xp_row(r0 - 1, a1);
    xp_row(r0, a2);
    bu_row(r0 - 1, u1);
    x3c(r0 - 1, cC, cS.n, a0, a1, a2, u1, b0);
#pragma unroll
    for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
    xp_row(r0 + 1, a2);
    bu_row(r0, u1);
    load_crow(r0, cN);
    x3c(r0, cN, cC.n, a0, a1, a2, u1, b1);
    // from here on: cS = row - 1 (only its n is used), cC = row, cN = row + 1
#pragma unroll
    for (int j = 0; j < N2; ++j) cS.n[j] = cC.n[j];
    cC = cN;
    float* __restrict__ pz = zout + ((i64)r0 * W + c0w) * Bp;
    for (int row = r0; row < r1; ++row) {
#pragma unroll
      for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
      xp_row(row + 2, a2);
      bu_row(row + 1, u2);
      load_crow(row + 1, cN);
      x3c(row + 1, cN, cC.n, a0, a1, a2, u2, b2);
      k1c(tRW{}, t1{}, cC, cS.n, b0, b1, b2, [&](int k, VT kx, VT, VT rd) {
        if (!EDGE || c0w + k < W) {
          const VT z = b1[k + 1] + (wB * rd) * (u1[k + 1] - kx);
          *(VT*)(pz + (i64)k * Bp + lb) = z;
          if (DOT) VLane<VT>::dot(acc, u1[k + 1], z);
        }
      });
      pz += (i64)W * Bp;
#pragma unroll
      for (int j = 0; j < N2; ++j) { b0[j] = b1[j]; b1[j] = b2[j]; u1[j] = u2[j]; cS.n[j] = cC.n[j]; }
      cC = cN;
    }
    return;
  }
  xp_row(r0 - 2, a0);
  xp_row(r0 - 1, a1);
  xp_row(r0, a2);
  bu_row(r0 - 1, u1);
  x3_row(r0 - 1, a0, a1, a2, u1, b0);
#pragma unroll
  for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
  xp_row(r0 + 1, a2);
  bu_row(r0, u1);
  x3_row(r0, a0, a1, a2, u1, b1);
  float* __restrict__ pz = zout + ((i64)r0 * W + c0w) * Bp;

  // (A software-pipelined form of this loop -- the raw loads of the next row requested before this row's two stencil stages,
  // 48 more live VGPRs, 3 instead of 4 waves per SIMD -- was built and measured at the end of round 4: correct, 168 VGPRs with
  // 80 B of scratch, 0.999 -> 1.321 ms per fine-level launch, step 80.8 -> 87.5 ms, gpurun_out/r4bj.  Independent waves hide the
  // row's load latency better than one wave overlapping its own rows.)
  for (int row = r0; row < r1; ++row) {
#pragma unroll
    for (int j = 0; j < N1; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
    xp_row(row + 2, a2);
    bu_row(row + 1, u2);
    x3_row(row + 1, a0, a1, a2, u2, b2);
    k1_row<VT, RW, ND, EDGE>(cf, n, W, row, c0w, b0, b1, b2, [&](int k, VT kx, typename CF::T, typename CF::T rd) {
      if (!EDGE || c0w + k < W) {
        const VT z = b1[k + 1] + (wB * rd) * (u1[k + 1] - kx);
        *(VT*)(pz + (i64)k * Bp + lb) = z;
        if (DOT) VLane<VT>::dot(acc, u1[k + 1], z);     // (r / s_b) . z; times s_b after the loop
      }
    });
    pz += (i64)W * Bp;
#pragma unroll
    for (int j = 0; j < N2; ++j) { b0[j] = b1[j]; b1[j] = b2[j]; u1[j] = u2[j]; }
  }
}

// XZ: the operand is P e alone (x = 0 is not read): two sweeps from a prolonged initial guess, the first stage of a
// full-multigrid level (vcycle with `guess`)
template <typename VT, int ND, int RW, bool DOT, bool SHARED, bool XZ = false, int NW = 4, int MW = (SHARED ? 4 : 1)>
__global__ __launch_bounds__(64 * NW, MW) void fused_post_kernel(Level L, const double* __restrict__ scale,
                                                          const float* __restrict__ xin, const float* __restrict__ rhs,
                                                          const float* __restrict__ ec, float* __restrict__ zout, float wA,
                                                          float wB, int cW, double* __restrict__ part, int Bp, int ncb,
                                                          int TR) {
  __shared__ double lds[NW * kWave];
  constexpr int SPL = VLane<VT>::kSpl;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = blockIdx.y * (SPL * kWave) + SPL * lane;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int rc = tile / ncb, cb = tile - rc * ncb;
  const int nyp = L.ny + 1;
  const int c0w = (cb * NW + wave) * RW;
  const int r0 = rc * TR;
  const int r1 = (r0 + TR < nyp) ? r0 + TR : nyp;
  const bool active = c0w < L.W && r0 < r1;
  const VT sb = VLane<VT>::from_scale(scale, lb);
  const VT ib = 1.0f / sb;
  Acc acc;
  if (active) {
    const bool edge = c0w - 2 < 0 || c0w + RW + 1 > L.W - 1 || r0 - 2 < 0 || r1 + 1 > nyp - 1;
    if (edge)
      fused_post_body<VT, ND, RW, true, DOT, SHARED, XZ>(L, ib, xin, rhs, ec, zout, wA, wB, cW, Bp, lb, c0w, r0, r1, acc);
    else
      fused_post_body<VT, ND, RW, false, DOT, SHARED, XZ>(L, ib, xin, rhs, ec, zout, wA, wB, cW, Bp, lb, c0w, r0, r1, acc);
  }
  if (DOT) {
#pragma unroll
    for (int q = 0; q < SPL; ++q) {
      const double f = scale ? scale[lb + q] : 1.0;
      const double t = block_sum_waves<NW>(acc.v[q] * f, lds);
      if (wave == 0) part[(i64)blockIdx.x * Bp + lb + q] = t;
    }
  }
}

// ---- CG step of a batch-shared matrix with fp32-stored directions, two samples per lane -------------------------------
// p = z + beta p_old (fp32 fused multiply-add: the stored value), p . (K_1 p) with the stencil in packed fp32 on the fp32
// coefficient copies, accumulated per sample in fp64; A p itself is never stored (the residual update recomputes it in
// fp64 from the stored p, F_RUPD).  The fp32 stencil only enters the STEP LENGTH alpha = r.z / p.Ap: x += alpha p and
// r -= alpha A p use the same alpha and the exact (fp64) A p, so r = b - A x holds to fp64 whatever alpha is, and an
// error delta in alpha costs delta^2 of the energy reduction of the step (the minimum of a parabola).
template <typename VT, int ND, int RW, bool EDGE>
__device__ __forceinline__ void cgstep2_body(const Level& L, VT beta, bool first, const float* __restrict__ z,
                                             const float* __restrict__ pin, float* __restrict__ pout, int Bp, unsigned lb,
                                             int c0w, int r0, int r1, Acc& acc) {
  constexpr int N = RW + 2;                  // window columns c0w - 1 + j
  const int W = L.W, nyp = L.ny + 1;
  const i64 n = L.n;
  const VT Z = VLane<VT>::zero();
  const Coef<VT, true> cf(L, 0, lb, Bp);
  bool ok[N];
  unsigned off[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    int c = c0w - 1 + j;
    ok[j] = !EDGE || (c >= 0 && c < W);
    if (EDGE) c = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
    off[j] = 4u * ((unsigned)(c - (c0w - 1) + 1) * (unsigned)Bp + lb);      // base sits one column further left
  }
  const i64 tile0 = ((i64)(r0 - 1) * W + (c0w - 2)) * Bp;                  // element (r0 - 1, c0w - 2)
  const Src rz = make_src(z + tile0);
  const Src rp = make_src(first ? z + tile0 : pin + tile0);
  const unsigned rowB = 4u * (unsigned)W * (unsigned)Bp;
  auto p_row = [&](int R, VT* dst) {
    if (EDGE && (R < 0 || R >= nyp)) {
#pragma unroll
      for (int j = 0; j < N; ++j) dst[j] = Z;
      return;
    }
    const unsigned sx = (unsigned)(R - (r0 - 1)) * rowB;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      VT v = ldsrc<VT>(rz, off[j], sx);
      if (!first) v += beta * ldsrc<VT>(rp, off[j], sx);
      dst[j] = ok[j] ? v : Z;
    }
  };
  VT a0[N], a1[N], a2[N];
  p_row(r0 - 1, a0);
  p_row(r0, a1);
  float* __restrict__ pp = pout + ((i64)r0 * W + c0w) * Bp;
  for (int row = r0; row < r1; ++row) {
    p_row(row + 1, a2);
    k1_row<VT, RW, ND, EDGE>(cf, n, W, row, c0w, a0, a1, a2, [&](int k, VT kx, float, float) {
      if (!EDGE || c0w + k < W) {
        __builtin_nontemporal_store(a1[k + 1], (VT*)(pp + (i64)k * Bp + lb));
        VLane<VT>::dot(acc, a1[k + 1], kx);
      }
    });
    pp += (i64)W * Bp;
#pragma unroll
    for (int j = 0; j < N; ++j) { a0[j] = a1[j]; a1[j] = a2[j]; }
  }
}

template <typename VT, int ND, int RW, int MW>
__global__ __launch_bounds__(256, MW) void cgstep2_kernel(Level L, const double* __restrict__ scale,
                                                           const double* __restrict__ beta, int first,
                                                           const float* __restrict__ z, const float* __restrict__ pin,
                                                           float* __restrict__ pout, double* __restrict__ part, int Bp,
                                                           int ncb, int TR) {
  __shared__ double lds[4 * kWave];
  constexpr int SPL = VLane<VT>::kSpl;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = blockIdx.y * (SPL * kWave) + SPL * lane;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int rc = tile / ncb, cb = tile - rc * ncb;
  const int nyp = L.ny + 1;
  const int c0w = (cb * 4 + wave) * RW;
  const int r0 = rc * TR;
  const int r1 = (r0 + TR < nyp) ? r0 + TR : nyp;
  Acc acc;
  if (c0w < L.W && r0 < r1) {
    const VT bt = first ? VLane<VT>::zero() : VLane<VT>::from_scale(beta, lb);
    const bool edge = c0w - 1 < 0 || c0w + RW > L.W - 1 || r0 - 1 < 0 || r1 > nyp - 1;
    if (edge) cgstep2_body<VT, ND, RW, true>(L, bt, first != 0, z, pin, pout, Bp, lb, c0w, r0, r1, acc);
    else cgstep2_body<VT, ND, RW, false>(L, bt, first != 0, z, pin, pout, Bp, lb, c0w, r0, r1, acc);
  }
#pragma unroll
  for (int q = 0; q < SPL; ++q) {
    const double f = scale ? scale[lb + q] : 1.0;
    const double t = block_sum_per_sample(acc.v[q] * f, Bp, lds);
    if (wave == 0) part[(i64)blockIdx.x * Bp + lb + q] = t;
  }
}

constexpr int kStripCols = 8;
// fp32-stored V-cycle vectors run best on 4-column strips (kernel trace, same box: prolongation + sweep -7 %,
// first two sweeps -5 % against 8 columns); fp64 vectors keep 8 (half the register footprint per column there)
template <typename TV>
constexpr int strip_cols() { return sizeof(TV) == 4 ? 4 : kStripCols; }
constexpr int kRestrictCols = 2;  // coarse columns per wave of the fused residual + restriction (5 fine columns; 3, 4, 6: slower)
constexpr int kPupdCols = 4;  // narrower strips for the 3-stream fused CG kernel: fewer VGPRs, more waves
constexpr int kPartBlocks = 2048;  // capacity (in blocks) of every partial-sum buffer

struct StripGeom {
  bool use;
  int ncb, nrc, TR;
};

inline StripGeom strip_geom(const Level& L, int Bp, int rw = kStripCols, int spl = 1, int nw = 4) {
  StripGeom g{false, 0, 0, 0};
  // small levels: one wave marching down a strip is latency-bound; the simple kernels win below ~200^2
  // 128: the 129^2 level of a 1024^2 hierarchy takes the strip / fused kernels too (-0.7 ms per step; 64: slower again)
  static const int minw = getenv("DIFFHE_STRIP_MINW") ? atoi(getenv("DIFFHE_STRIP_MINW")) : 128;
  if (Bp < kWave || L.W < minw || L.ny + 1 < 64) return g;
  g.use = true;
  g.ncb = (L.W + nw * rw - 1) / (nw * rw);   // nw = waves (strips) per block
  const int gy = Bp / (kWave * spl);   // spl = samples per lane (2: dia_strip2_kernel)
  if (gy < 1) { g.use = false; return g; }   // fewer samples than one wave of that form holds
  static const int target4 = getenv("DIFFHE_STRIP_BLOCKS") ? atoi(getenv("DIFFHE_STRIP_BLOCKS")) : 6144;
  const int target = target4 * 4 / nw;       // the same number of WAVES (and tile height) whatever the block width
  int nrc = (target + g.ncb * gy - 1) / (g.ncb * gy);
  const int nyp = L.ny + 1;
  if (nrc > nyp / 8) nrc = nyp / 8;
  if (nrc < 1) nrc = 1;
  while (g.ncb * nrc > kPartBlocks) --nrc;
  g.TR = (nyp + nrc - 1) / nrc;
  g.nrc = (nyp + g.TR - 1) / g.TR;
  return g;
}

template <typename TV, int MODE, bool XFROMB, int FUSE = F_NONE, typename TA = TV, int RW = kStripCols, int MINW = 1>
void launch_strip(const Level& L, int Bv, const double* scale, const TV* xin, const TV* bvec, TV* out,
                  double omega, double omega_in, double* part, int Bp, const StripGeom& g, hipStream_t st,
                  const Extra& ex = Extra{}) {
  dim3 grid(g.ncb * g.nrc, Bp / kWave);
  // per-sample matrices inside the fp32 V-cycle read the fp32 copy of the coefficients
  const bool m32 = (sizeof(TV) == 4) && Bv != 1 && L.v32 != nullptr;
  const bool m16 = m32 && L.o16 != nullptr;   // fp32 diagonal + fp16 off-diagonals
  {  // algorithmic bytes per (node, sample) of this launch (diffhe_traffic_account)
    const double tv = sizeof(TV), ta = sizeof(TA);
    double bpn;
    if (MODE == M_JACOBI) bpn = (XFROMB ? 2.0 : 3.0) * tv + (FUSE == F_PROLONG ? 0.25 * tv : 0.0);
    else if (MODE == M_RESID) bpn = 2.0 * tv + (FUSE == F_RESTRICT ? 0.25 * tv : (out ? tv : 0.0) + (ex.r32 ? 4.0 : 0.0));
    else if (is_pupd(FUSE)) bpn = (ex.first ? 2.0 * ta : 3.0 * ta) + (out ? 8.0 : 0.0) + ((FUSE == F_PUPD && ex.x) ? 16.0 : 0.0);
    else if (FUSE == F_RUPD) bpn = ta + 16.0 + (ex.r32 ? 4.0 : 0.0);
    else bpn = tv + (out ? tv : 0.0) + (ex.dotv ? 8.0 : 0.0);
    if (Bv != 1) bpn += m16 ? 4.0 + 2.0 * (L.nd - 1) : L.nd * (m32 ? 4.0 : 8.0);
    diffhe::account(bpn * (double)L.n * Bp);
  }
  // development knob: DIFFHE_S1_LDS=<bytes> of dynamic LDS per block caps the blocks resident per CU (160 KB / bytes)
  const unsigned dyn_lds = getenv("DIFFHE_S1_LDS") ? (unsigned)atoi(getenv("DIFFHE_S1_LDS")) : 0u;
#define STRIP(ND_, SH_, TM_)                                                                                       \
  hipLaunchKernelGGL((dia_strip_kernel<TV, TA, TM_, MODE, FUSE, ND_, SH_, XFROMB, RW, MINW>), grid, dim3(256), dyn_lds, st, L,   \
                     scale, xin, bvec, out, omega, omega_in, ex, part, Bp, g.ncb, g.TR)
#define STRIP_SHIFT(ND_)                                                                                           \
  hipLaunchKernelGGL((dia_strip_shift_kernel<TV, TA, MODE, FUSE, ND_, XFROMB, RW, MINW>), grid, dim3(256), 0, st, L, scale, \
                     xin, bvec, out, omega, omega_in, ex, part, Bp, g.ncb, g.TR)
  if (Bv == 1 && L.shift) {
    if (L.nd == 3) STRIP_SHIFT(3); else STRIP_SHIFT(4);
  } else if (L.nd == 3) {
    if (Bv == 1) STRIP(3, true, double); else if (m16) STRIP(3, false, h16m); else if (m32) STRIP(3, false, float); else STRIP(3, false, double);
  } else {
    if (Bv == 1) STRIP(4, true, double); else if (m16) STRIP(4, false, h16m); else if (m32) STRIP(4, false, float); else STRIP(4, false, double);
  }
#undef STRIP
#undef STRIP_SHIFT
}

// fp32 V-cycle, batch-shared matrix with fp32 coefficient copies and reciprocal diagonal, batch a multiple of 128,
// no diagonal shift: the two-samples-per-lane kernels apply (DIFFHE_STRIP2=0 switches them off: A/B runs)
inline bool shared32_ok(const Level& L, int Bv, int Bp) {   // the fp32 copies of a batch-shared matrix are there, whole waves
  static const int on = getenv("DIFFHE_STRIP2") ? atoi(getenv("DIFFHE_STRIP2")) : 1;
  return on && Bv == 1 && L.v32 && L.rd32 && L.mk32 && !L.shift && Bp % kWave == 0;
}
inline bool strip2_ok(const Level& L, int Bv, int Bp) { return shared32_ok(L, Bv, Bp) && Bp % (2 * kWave) == 0; }
// the kernels address their tile (`rows` fine rows + the window's two halo rows) with 32-bit byte offsets
inline bool strip2_tile_fits(const Level& L, int Bp, int rows) { return 4LL * (rows + 3) * L.W * Bp < (1LL << 31); }
// geometry for the two-samples-per-lane kernels if they apply to this level (and its tiles fit), else the usual one
template <typename TV>
inline bool strip2_pick(const Level& L, int Bv, int Bp, int rw, StripGeom* g) {
  if (sizeof(TV) == 4 && strip2_ok(L, Bv, Bp)) {
    *g = strip_geom(L, Bp, rw, 2);
    if (!g->use || strip2_tile_fits(L, Bp, g->TR)) return g->use;
  }
  *g = strip_geom(L, Bp, rw, 1);
  return false;
}

template <int MODE, bool XFROMB, int FUSE, int RW>
void launch_strip2(const Level& L, const double* scale, const float* xin, const float* bvec, float* out, double omega,
                   double omega_in, double* part, int Bp, const StripGeom& g, hipStream_t st, const Extra& ex = Extra{}) {
  dim3 grid(g.ncb * g.nrc, Bp / (2 * kWave));
  double bpn;  // algorithmic bytes per (node, sample), as launch_strip
  if (MODE == M_JACOBI) bpn = (XFROMB ? 2.0 : 3.0) * 4.0 + (FUSE == F_PROLONG ? 1.0 : 0.0);
  else bpn = 8.0 + (FUSE == F_RESTRICT ? 1.0 : 4.0);
  diffhe::account(bpn * (double)L.n * Bp);
  // 40 000 B of dynamic LDS per block cap the residency at 4 blocks (16 waves) per CU: measured best for these kernels
  // (sweep over 3 .. 7 blocks per CU on one box, gpurun_out/r3d: first two sweeps 0.506 / prolongation 0.803 / restriction
  // 0.582 ms at 4 against 0.514-0.523 / 0.815-0.818 / 0.589-0.590 unrestricted); DIFFHE_S2_LDS=<bytes> overrides
  static const unsigned dyn_lds = getenv("DIFFHE_S2_LDS") ? (unsigned)atoi(getenv("DIFFHE_S2_LDS")) : 40000u;
#define STRIP2(ND_, DOT_, BST_)                                                                                            \
  hipLaunchKernelGGL((dia_strip2_kernel<MODE, FUSE, ND_, XFROMB, RW, DOT_, BST_>), grid, dim3(256), dyn_lds, st, L, scale, xin, \
                     bvec, out, (float)omega, (float)omega_in, ex, part, Bp, g.ncb, g.TR)
#define STRIP2B(ND_, DOT_) STRIP2(ND_, DOT_, false)
  if (MODE == M_JACOBI && part) {   // the sweep that leaves the partials of rhs . x (the CG's r.z)
    if (L.nd == 3) STRIP2B(3, (MODE == M_JACOBI)); else STRIP2B(4, (MODE == M_JACOBI));
  } else {
    if (L.nd == 3) STRIP2B(3, false); else STRIP2B(4, false);
  }
#undef STRIP2B
#undef STRIP2
}

// Fused two-stage passes (fused_pre_kernel / fused_post_kernel): DIFFHE_FUSED=0 keeps the four single-stage passes,
// DIFFHE_FUSED_SPL=1 runs them with one sample per lane
inline int fused_mode() {
  static const int on = getenv("DIFFHE_FUSED") ? atoi(getenv("DIFFHE_FUSED")) : 1;
  static const int spl = getenv("DIFFHE_FUSED_SPL") ? atoi(getenv("DIFFHE_FUSED_SPL")) : 2;
  return on ? (spl == 1 ? 1 : 2) : 0;
}
inline unsigned fused_lds() {
  // dynamic LDS per block = a cap on the blocks resident per CU; the fused passes (118-125 VGPRs: 4 waves per SIMD
  // anyway) run best without one (gpurun_out/r3w: 93.9 ms per step at 0, 94.4 at 40 000, 103.8 at 54 000)
  static const unsigned v = getenv("DIFFHE_FUSED_LDS") ? (unsigned)atoi(getenv("DIFFHE_FUSED_LDS")) : 0u;
  return v;
}

// fused passes apply to: a batch-shared matrix with its fp32 copy, reciprocal diagonal and mask (strip2_ok), or a
// per-sample matrix with the compact copies (fp32 diagonal + scaled fp16 off-diagonals) and the mask, no per-sample scale
// returns a bit mask: 1 = the PRE pass may be fused, 2 = the POST pass
// development knob: "a:b:c" = one integer per multigrid level (0 / missing = keep the default)
// DIFFHE_RUPD: 0 = store A p and read it back in pcg_update_kernel; 1 = recompute it in the residual update (F_RUPD);
// 2 / 4 = the same with other launch shapes (development)
inline int rupd_env() {
  static const int v = getenv("DIFFHE_RUPD") ? atoi(getenv("DIFFHE_RUPD")) : 1;
  return v;
}

inline int env_level_int(const char* name, int level) {
  const char* e = getenv(name);
  if (!e) return 0;
  for (int l = 0; l < level; ++l) {
    e = strchr(e, ':');
    if (!e) return 0;
    ++e;
  }
  return atoi(e);
}

inline int fused_ok(const Level& L, int Bv, int Bp, const double* scale) {
  // batch-shared matrix: two samples per lane for multiples of 128, else ONE per lane (batches of 64 or 192 per GPU --
  // BASELINE config 5's shard: same fused passes, fp32 arithmetic, 4-byte accesses)
  if (shared32_ok(L, Bv, Bp)) return 3;
  // per-sample matrices: both passes fused, with the coefficient rows cached in registers (fp16 couplings as raw words:
  // 174 / 206 VGPRs, 2 waves per SIMD, no spills).  Without the cache the PRE pass needed 256 VGPRs and measured slower
  // than its two single passes (forward solve 153 ms against 140), and the POST pass re-read every coefficient row four
  // times (PMC 4.5 passes of traffic for 2.6 algorithmic); 1024^2 x 256 step: 245 (POST only, uncached) -> 236 (POST
  // cached) -> 219 ms (both, cached; gpurun_out/r5b, r5d).  DIFFHE_FUSED_PS: bit 0 = PRE, bit 1 = POST.
  static const int per_sample = getenv("DIFFHE_FUSED_PS") ? atoi(getenv("DIFFHE_FUSED_PS")) : 3;
  return (Bv == Bp && Bp % (2 * kWave) == 0 && L.v32 && L.o16 && L.mk32 && !L.shift && !scale) ? per_sample : 0;
}

void launch_fused_pre(const Level& L, const Level& C, int Bv, const double* scale, const float* rhs, float* x2, float* crhs,
                      double w0, double w1, int Bp, const StripGeom& g, int spl, hipStream_t st, int nw = 4) {
  constexpr int CW = kRestrictCols;
  // r read, x2 and the coarse rhs written; per-sample matrices: + the compact coefficients (read by both stages)
  diffhe::account((9.0 + (Bv == 1 ? 0.0 : 4.0 + 2.0 * (L.nd - 1))) * (double)L.n * Bp);
  const dim3 grid(g.ncb * g.nrc, Bp / (spl * kWave));
#define FPRE(VT_, ND_, SH_)                                                                                               \
  hipLaunchKernelGGL((fused_pre_kernel<VT_, ND_, CW, SH_>), grid, dim3(256), fused_lds(), st, L, scale, rhs, x2, crhs,     \
                     (float)w0, (float)w1, C.W, C.bc, Bp, g.ncb, g.TR)
  // per-sample coefficients: 206 VGPRs = 2 waves per SIMD; capped at 168 (3 waves) it spills and loses (launch_fused_post)
  static const int ps_mw = getenv("DIFFHE_FUSED_PS_MW") ? atoi(getenv("DIFFHE_FUSED_PS_MW")) : 2;
  if (Bv != 1 && L.nd == 3 && (ps_mw & 1))
    hipLaunchKernelGGL((fused_pre_kernel<v2f, 3, CW, false, 4, 3>), grid, dim3(256), fused_lds(), st, L, scale, rhs, x2, crhs,
                       (float)w0, (float)w1, C.W, C.bc, Bp, g.ncb, g.TR);
  else if (Bv != 1) { if (L.nd == 3) FPRE(v2f, 3, false); else FPRE(v2f, 4, false); }
  else if (spl == 4 && L.nd == 3)
    hipLaunchKernelGGL((fused_pre_kernel<v4f, 3, CW, true, 4, 2>), grid, dim3(256), fused_lds(), st, L, scale, rhs, x2, crhs,
                       (float)w0, (float)w1, C.W, C.bc, Bp, g.ncb, g.TR);
  else if (spl >= 2) { if (L.nd == 3) FPRE(v2f, 3, true); else FPRE(v2f, 4, true); }
  else { if (L.nd == 3) FPRE(float, 3, true); else FPRE(float, 4, true); }
#undef FPRE
}

void launch_fused_post(const Level& L, const Level& C, int Bv, const double* scale, const float* xin, const float* rhs,
                       const float* ec, float* z, double wA, double wB, double* part, int Bp, const StripGeom& g, int spl,
                       hipStream_t st, int nw = 4) {
  // x2, r, a quarter of e read; z written (+ compact coefficients of a per-sample matrix); xin == NULL: x2 = 0, not read
  diffhe::account(((xin ? 13.0 : 9.0) + (Bv == 1 ? 0.0 : 4.0 + 2.0 * (L.nd - 1))) * (double)L.n * Bp);
  const dim3 grid(g.ncb * g.nrc, Bp / (spl * kWave));
#define FPOST(VT_, ND_, DOT_, SH_, XZ_)                                                                                      \
  hipLaunchKernelGGL((fused_post_kernel<VT_, ND_, 4, DOT_, SH_, XZ_>), grid, dim3(256), fused_lds(), st, L, scale, xin, rhs, ec, \
                     z, (float)wA, (float)wB, C.W, part, Bp, g.ncb, g.TR)
#define FPOSTD(VT_, ND_, SH_)                                                                                              \
  do {                                                                                                                     \
    if (!xin) FPOST(VT_, ND_, false, SH_, true);                                                                          \
    else if (part) FPOST(VT_, ND_, true, SH_, false);                                                                     \
    else FPOST(VT_, ND_, false, SH_, false);                                                                              \
  } while (0)
  // per-sample coefficients: the POST pass needs 173 VGPRs uncapped (176 allocated: 2 waves per SIMD); capped at 168 it runs
  // 3 waves per SIMD without spills: 218.1 -> 213.5 ms per 1024^2 x 256 step of the per-element-field workload; the PRE pass
  // (206 VGPRs) spills under the same cap: 248.9 ms (gpurun_out/r4e).  DIFFHE_FUSED_PS_MW: bit 0 = PRE, bit 1 = POST capped
  static const int ps_mw = getenv("DIFFHE_FUSED_PS_MW") ? atoi(getenv("DIFFHE_FUSED_PS_MW")) : 2;
  if (Bv != 1 && L.nd == 3 && (ps_mw & 2)) {
#define FPOSTM(DOT_, XZ_)                                                                                                  \
  hipLaunchKernelGGL((fused_post_kernel<v2f, 3, 4, DOT_, false, XZ_, 4, 3>), grid, dim3(256), fused_lds(), st, L, scale, xin, \
                     rhs, ec, z, (float)wA, (float)wB, C.W, part, Bp, g.ncb, g.TR)
    if (!xin) FPOSTM(false, true); else if (part) FPOSTM(true, false); else FPOSTM(false, false);
#undef FPOSTM
  }
  else if (Bv != 1) { if (L.nd == 3) FPOSTD(v2f, 3, false); else FPOSTD(v2f, 4, false); }
  else if (spl >= 2) { if (L.nd == 3) FPOSTD(v2f, 3, true); else FPOSTD(v2f, 4, true); }
  else { if (L.nd == 3) FPOSTD(float, 3, true); else FPOSTD(float, 4, true); }
#undef FPOSTD
#undef FPOST
}

// One step of the Chebyshev semi-iteration (three-term form) on the coarsest level:
//   d_out = c1 d_in + c2 D^-1 (b - A x_in) ;  x_out = x_in + d_out        (d_in == NULL: c1 = 0)
// part (optional): per-sample partial of b.x_out, as in dia_jacobi_kernel.
template <typename TV>
__global__ __launch_bounds__(256) void dia_cheby_kernel(Level L, int Bv, const double* __restrict__ scale,
                                                         const TV* __restrict__ bvec, const TV* __restrict__ xin,
                                                         const TV* __restrict__ din, TV* __restrict__ xout,
                                                         TV* __restrict__ dout, double c1, double c2,
                                                         double* __restrict__ part, int Bp) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const int vb = Bv == 1 ? 0 : nm.b;
  double s = 0.0;
  for (int i = nm.node0; i < L.n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    const double sc = row_scale(L, scale, i, nm.b);
    const double sh = shift_at(L, i);
    const double dinv = fast_rcp(sc * L.v[(i64)i * Bv + vb] + sh);
    const double bi = (double)bvec[o];
    const double xi = xin ? (double)xin[o] : 0.0;
    const double res = xin ? bi - (sc * dia_row(L, Bv, vb, xin, i, nm.b, Bp) + sh * xi) : bi;
    const double dn = (din ? c1 * (double)din[o] : 0.0) + c2 * res * dinv;
    dout[o] = (TV)dn;
    const double xo = xi + dn;
    xout[o] = (TV)xo;
    s += bi * xo;
  }
  if (part) STORE_PARTIAL(part, s);
}

// Coarsening of a level pair: both directions (2:1 nested triangulations, P = P1 interpolation with the
// quad-diagonal midpoints) or ONE direction only (semi-coarsening, used while the mesh is anisotropic:
// P = 1D linear interpolation along the coarsened direction).
__device__ inline int coarsen_x(const Level& F, const Level& C) { return F.nx == 2 * C.nx ? 2 : 1; }
__device__ inline int coarsen_y(const Level& F, const Level& C) { return F.ny == 2 * C.ny ? 2 : 1; }

// coarse rhs = P^T r, 0 on coarse Dirichlet rows
template <typename TV>
__global__ __launch_bounds__(256) void mg_restrict_kernel(Level F, Level C, const TV* __restrict__ r,
                                                           TV* __restrict__ rc, int Bp) {
  const NodeMap nm = node_map(Bp);
  const int sx = coarsen_x(F, C), sy = coarsen_y(F, C);
  for (int I = nm.node0; I < C.n; I += nm.stride) {
    double out = 0.0;
    if (!C.bc[I]) {
      const int ci = I / C.W, cj = I - ci * C.W;
      const int fi = sy * ci, fj = sx * cj;
      const i64 c = (i64)fi * F.W + fj;
      double h = 0.0;
      if (sx == 2) {
        if (fj > 0) h += (double)r[(c - 1) * Bp + nm.b];
        if (fj < F.nx) h += (double)r[(c + 1) * Bp + nm.b];
      }
      if (sy == 2) {
        if (fi > 0) h += (double)r[(c - F.W) * Bp + nm.b];
        if (fi < F.ny) h += (double)r[(c + F.W) * Bp + nm.b];
      }
      if (sx == 2 && sy == 2) {  // midpoints of the quad diagonals b-d
        if (fi > 0 && fj < F.nx) h += (double)r[(c - F.W + 1) * Bp + nm.b];
        if (fi < F.ny && fj > 0) h += (double)r[(c + F.W - 1) * Bp + nm.b];
      }
      out = (double)r[c * Bp + nm.b] + 0.5 * h;
    }
    rc[(i64)I * Bp + nm.b] = (TV)out;
  }
}

// x += P e  (0 on fine Dirichlet rows)
template <typename TV>
__global__ __launch_bounds__(256) void mg_prolong_add_kernel(Level F, Level C, const TV* __restrict__ e,
                                                              TV* __restrict__ x, int Bp, int set = 0) {
  const NodeMap nm = node_map(Bp);
  for (int i = nm.node0; i < F.n; i += nm.stride) {
    if (F.bc[i]) {
      if (set) x[(i64)i * Bp + nm.b] = (TV)0.0;
      continue;
    }
    const int fi = i / F.W, fj = i - fi * F.W;
    const int sx = coarsen_x(F, C), sy = coarsen_y(F, C);
    const bool oi = sy == 2 && (fi & 1), oj = sx == 2 && (fj & 1);  // between two coarse rows / columns
    const int ci = sy == 2 ? fi >> 1 : fi, cj = sx == 2 ? fj >> 1 : fj;
    const i64 c = (i64)ci * C.W + cj;
    double v;
    if (!oi && !oj)
      v = (double)e[c * Bp + nm.b];
    else if (!oi)
      v = 0.5 * ((double)e[c * Bp + nm.b] + (double)e[(c + 1) * Bp + nm.b]);
    else if (!oj)
      v = 0.5 * ((double)e[c * Bp + nm.b] + (double)e[(c + C.W) * Bp + nm.b]);
    else  // midpoint of the quad diagonal b-d (full coarsening only)
      v = 0.5 * ((double)e[(c + 1) * Bp + nm.b] + (double)e[(c + C.W) * Bp + nm.b]);
    x[(i64)i * Bp + nm.b] = set ? (TV)v : (TV)((double)x[(i64)i * Bp + nm.b] + v);
  }
}

// The two transfers for fp32 vectors, full coarsening and batches that are multiples of 128: a wave owns ONE node and 128
// samples (8-byte accesses), the node index and everything derived from it (row / column, parities, Dirichlet flag,
// coarse index) is wave-uniform scalar arithmetic instead of one integer division per lane.  Same fp64 arithmetic per
// sample as mg_prolong_add_kernel / mg_restrict_kernel: bitwise the same values.
__global__ __launch_bounds__(256) void mg_prolong2_kernel(Level F, Level C, const float* __restrict__ e,
                                                           float* __restrict__ x, int Bp, int set) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = blockIdx.y * (2 * kWave) + 2 * lane;
  for (int i = blockIdx.x * 4 + wave; i < F.n; i += gridDim.x * 4) {
    float* __restrict__ px = x + (i64)i * Bp + lb;
    if (F.bc[i]) {
      if (set) *(v2f*)px = v2f{0.0f, 0.0f};
      continue;
    }
    const int fi = i / F.W, fj = i - fi * F.W;
    const bool oi = fi & 1, oj = fj & 1;
    const float* __restrict__ pe = e + ((i64)(fi >> 1) * C.W + (fj >> 1)) * Bp + lb;
    double v0, v1;
    if (!oi && !oj) {
      const v2f a = *(const v2f*)pe;
      v0 = (double)a.x; v1 = (double)a.y;
    } else {
      const v2f a = *(const v2f*)(pe + ((oi && oj) ? (i64)Bp : 0));                       // c (or c + 1 on a quad diagonal)
      const v2f b = *(const v2f*)(pe + (!oi ? (i64)Bp : (i64)C.W * Bp));                  // c + 1 (odd column only) or c + C.W
      v0 = 0.5 * ((double)a.x + (double)b.x);
      v1 = 0.5 * ((double)a.y + (double)b.y);
    }
    if (set) {
      *(v2f*)px = v2f{(float)v0, (float)v1};
    } else {
      const v2f o = *(const v2f*)px;
      *(v2f*)px = v2f{(float)((double)o.x + v0), (float)((double)o.y + v1)};
    }
  }
}

__global__ __launch_bounds__(256) void mg_restrict2_kernel(Level F, Level C, const float* __restrict__ r,
                                                            float* __restrict__ rc, int Bp) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = blockIdx.y * (2 * kWave) + 2 * lane;
  for (int I = blockIdx.x * 4 + wave; I < C.n; I += gridDim.x * 4) {
    float* __restrict__ po = rc + (i64)I * Bp + lb;
    if (C.bc[I]) {
      *(v2f*)po = v2f{0.0f, 0.0f};
      continue;
    }
    const int ci = I / C.W, cj = I - ci * C.W;
    const int fi = 2 * ci, fj = 2 * cj;
    const float* __restrict__ pc = r + ((i64)fi * F.W + fj) * Bp + lb;
    const i64 row = (i64)F.W * Bp;
    double h0 = 0.0, h1 = 0.0;
    auto acc = [&](const float* q) { const v2f t = *(const v2f*)q; h0 += (double)t.x; h1 += (double)t.y; };
    if (fj > 0) acc(pc - Bp);
    if (fj < F.nx) acc(pc + Bp);
    if (fi > 0) acc(pc - row);
    if (fi < F.ny) acc(pc + row);
    if (fi > 0 && fj < F.nx) acc(pc - row + Bp);
    if (fi < F.ny && fj > 0) acc(pc + row - Bp);
    const v2f cc = *(const v2f*)pc;
    *(v2f*)po = v2f{(float)((double)cc.x + 0.5 * h0), (float)((double)cc.y + 0.5 * h1)};
  }
}

inline bool transfers2_ok(const Level& F, const Level& C, int Bp, size_t esz) {
  static const int on = getenv("DIFFHE_TRANSFER2") ? atoi(getenv("DIFFHE_TRANSFER2")) : 1;
  return on && esz == 4 && F.nx == 2 * C.nx && F.ny == 2 * C.ny && Bp % (2 * kWave) == 0;
}
inline dim3 transfer2_grid(int n, int Bp) { return dim3((unsigned)(((i64)n + 3) / 4 < 4096 ? ((i64)n + 3) / 4 : 4096), Bp / (2 * kWave)); }

// per-element kappa of the coarse triangulation.  Full coarsening (sx = sy = 2): mean of the 4 children
// (Galerkin for nested P1).  Semi-coarsening: both coarse triangles of a cell take the mean of the 4 fine
// triangles of the 2 fine cells it covers.
__global__ __launch_bounds__(256) void mg_restrict_kappa_kernel(const double* __restrict__ kf, double* __restrict__ kc,
                                                                 int nxc, int nyc, int sx, int sy, int Bv) {
  const NodeMap nm = node_map(Bv);
  const int mc = 2 * nxc * nyc, nxf = sx * nxc;
  for (int E = nm.node0; E < mc; E += nm.stride) {
    const int q = E >> 1, up = E & 1;
    const int I = q / nxc, J = q - I * nxc;
    // fine element id = 2*(row*nxf + col) + upper
    auto fe = [&](int r, int c, int u) { return (i64)(2 * ((i64)r * nxf + c) + u) * Bv + nm.b; };
    double s;
    if (sx == 2 && sy == 2) {
      if (!up)
        s = kf[fe(2 * I, 2 * J, 0)] + kf[fe(2 * I, 2 * J, 1)] + kf[fe(2 * I, 2 * J + 1, 0)] + kf[fe(2 * I + 1, 2 * J, 0)];
      else
        s = kf[fe(2 * I + 1, 2 * J + 1, 1)] + kf[fe(2 * I + 1, 2 * J + 1, 0)] + kf[fe(2 * I, 2 * J + 1, 1)] +
            kf[fe(2 * I + 1, 2 * J, 1)];
    } else if (sx == 2) {
      s = kf[fe(I, 2 * J, 0)] + kf[fe(I, 2 * J, 1)] + kf[fe(I, 2 * J + 1, 0)] + kf[fe(I, 2 * J + 1, 1)];
    } else {
      s = kf[fe(2 * I, J, 0)] + kf[fe(2 * I, J, 1)] + kf[fe(2 * I + 1, J, 0)] + kf[fe(2 * I + 1, J, 1)];
    }
    kc[(i64)E * Bv + nm.b] = 0.25 * s;
  }
}


// Gershgorin bound of D^-1 A: max over rows (and samples) of sum_j |a_ij| / a_ii, as the bit pattern of a
// non-negative double (ordered like an unsigned integer, so atomicMax gives a deterministic result).
// Meshes with obtuse triangles have positive off-diagonal entries and a spectrum that reaches beyond 2.
__global__ __launch_bounds__(256) void dia_gershgorin_kernel(Level L, int Bv, unsigned long long* __restrict__ out) {
  const NodeMap nm = node_map(Bv);
  const i64 n = L.n;
  double m = 0.0;
  if (nm.b < Bv) {
    for (int i = nm.node0; i < L.n; i += nm.stride) {
      double sum = 0.0;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        if (k < L.nd) {
          const int off = dia_off(L, k);
          if (i + off < L.n) sum += fabs(L.v[((i64)k * n + i) * Bv + nm.b]);
          if (i - off >= 0) sum += fabs(L.v[((i64)k * n + (i - off)) * Bv + nm.b]);
        }
      }
      const double r = 1.0 + sum / L.v[(i64)i * Bv + nm.b];
      m = r > m ? r : m;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const double o = __shfl_xor(m, d);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(m));
}

// ---- CG vector kernels ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void pcg_init_kernel(const double* __restrict__ bvec, double* __restrict__ x,
                                                        double* __restrict__ r, double* __restrict__ part, int n,
                                                        int Bp) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  double s = 0.0;
  for (int i = nm.node0; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    const double bi = bvec[o];
    if (x) {  // x == NULL (full-multigrid start): x and r are set after the start, only b.b is due here
      x[o] = 0.0;
      r[o] = bi;
    }
    s += bi * bi;
  }
  STORE_PARTIAL(part, s);
}

__global__ __launch_bounds__(256) void pcg_update_kernel(const double* __restrict__ p, const double* __restrict__ Ap,
                                                          const double* __restrict__ alpha, double* __restrict__ x,
                                                          double* __restrict__ r, float* __restrict__ r32,
                                                          const double* __restrict__ rs, double* __restrict__ part,
                                                          int n, int Bp, int unroll = 0) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const double a = alpha[nm.b];
  const double sc = (r32 && rs) ? rs[nm.b] : 1.0;
  double s = 0.0;
  int i = nm.node0;
  // four nodes per trip (the loads of all four in flight together; one node per trip left a wave with two loads
  // outstanding: 4.8 TB/s); same nodes, same order of the partial sum
  if (unroll && !x)
    for (; (i64)i + 3LL * nm.stride < n; i += 4 * nm.stride) {
      double rv[4], av[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const i64 o = (i64)(i + u * nm.stride) * Bp + nm.b;
        rv[u] = __builtin_nontemporal_load(r + o);
        av[u] = __builtin_nontemporal_load(Ap + o);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const i64 o = (i64)(i + u * nm.stride) * Bp + nm.b;
        const double ri = rv[u] - a * av[u];
        __builtin_nontemporal_store(ri, r + o);
        if (r32) r32[o] = (float)(ri * sc);
        s += ri * ri;
      }
    }
  for (; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    if (x) x[o] += a * p[o];  // x == NULL: the iterate update is fused into the next operator apply
    const double ri = __builtin_nontemporal_load(r + o) - a * __builtin_nontemporal_load(Ap + o);
    __builtin_nontemporal_store(ri, r + o);
    if (r32) r32[o] = (float)(ri * sc);  // read again right away by the V-cycle: left cacheable
    s += ri * ri;
  }
  STORE_PARTIAL(part, s);
}

// (Two samples per lane -- 16-byte loads and stores, a wave moving 1 KB per instruction -- were measured for this kernel
// and for pcg_finish_kernel: 212.9 / 213.8 -> 214.6 / 214.2 ms per step of the per-element-field variant, headline step
// unchanged, gpurun_out/r4an.  At 4.9 TB/s these passes run at the rate of the box's own device-to-device copy.)
// y += x (TV) ; and the start of the CG from a full-multigrid iterate: x64 = (double) x0
template <typename TV>
__global__ __launch_bounds__(256) void mg_add_kernel(const TV* __restrict__ x, TV* __restrict__ y, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  for (int i = nm.node0; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    y[o] = (TV)((double)y[o] + (double)x[o]);
  }
}

template <typename TV>
__global__ __launch_bounds__(256) void pcg_setx_kernel(const TV* __restrict__ x0, const double* __restrict__ rs,
                                                        double* __restrict__ x, double* __restrict__ part, int n,
                                                        int Bp, int add = 0, const TV* __restrict__ e0 = nullptr,
                                                        int unroll = 0) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const double inv = rs ? 1.0 / rs[nm.b] : 1.0;  // the start was computed from the scaled right-hand side
  double s = 0.0;
  int i = nm.node0;
  if (unroll && !add)   // four nodes per trip (see pcg_update_kernel)
    for (; (i64)i + 3LL * nm.stride < n; i += 4 * nm.stride) {
      TV xv[4], ev[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const i64 o = (i64)(i + u * nm.stride) * Bp + nm.b;
        xv[u] = x0[o];
        ev[u] = e0 ? e0[o] : (TV)0;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const i64 o = (i64)(i + u * nm.stride) * Bp + nm.b;
        const double v = ((double)xv[u] + (e0 ? (double)ev[u] : 0.0)) * inv + 0.0;
        x[o] = v;
        s += v * v;
      }
    }
  for (; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    // add: x0 is a correction of the caller's iterate;  e0: the last cycle's correction of x0, not yet added (fmg_start)
    const double v = ((double)x0[o] + (e0 ? (double)e0[o] : 0.0)) * inv + (add ? x[o] : 0.0);
    x[o] = v;
    s += v * v;
  }
  if (part) STORE_PARTIAL(part, s);  // |x0|^2: scale of the attainable residual (S_FLOOR)
}

// Per-sample max of the matrix diagonal (bit pattern of a non-negative double, atomicMax: deterministic).
// out has Bv entries, zeroed by the caller.
__global__ __launch_bounds__(256) void dia_maxdiag_kernel(Level L, int Bv, unsigned long long* __restrict__ out) {
  const NodeMap nm = node_map(Bv);
  double m = 0.0;
  if (nm.b < Bv)
    for (int i = nm.node0; i < L.n; i += nm.stride) {
      const double d = L.v[(i64)i * Bv + nm.b];
      m = d > m ? d : m;
    }
  const int LB = Bv < kWave ? Bv : kWave;
  for (int off = LB; off < kWave; off <<= 1) {  // lanes that hold the same sample
    const double o = __shfl_xor(m, off);
    m = o > m ? o : m;
  }
  if ((int)(threadIdx.x & 63) < LB && nm.b < Bv) atomicMax(out + nm.b, (unsigned long long)__double_as_longlong(m));
}

// r32 = fp32(rs * r): the fp32 copies that feed the preconditioner are taken of the residual scaled by a per-sample
// power of two rs ~ 1 / |b| (S_INIT), so they stay inside the fp32 range whatever the magnitude of the data
// (forcing of amplitude 1e-35 used to underflow them); powers of two make the scaling exact, so nothing else changes.
__global__ __launch_bounds__(256) void pcg_cvt_kernel(const double* __restrict__ r, const double* __restrict__ rs,
                                                       float* __restrict__ r32, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  const double sc = rs ? rs[nm.b] : 1.0;
  int i = nm.node0;
  for (; (i64)i + 3LL * nm.stride < n; i += 4 * nm.stride) {   // four nodes per trip: four loads in flight per wave
    double rv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) rv[u] = r[(i64)(i + u * nm.stride) * Bp + nm.b];
#pragma unroll
    for (int u = 0; u < 4; ++u) r32[(i64)(i + u * nm.stride) * Bp + nm.b] = (float)(rv[u] * sc);
  }
  for (; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    r32[o] = (float)(r[o] * sc);
  }
}

// x += alpha p  (flush of the pending iterate update of the fused CG loop)
template <typename TP>
__global__ __launch_bounds__(256) void pcg_axpy_kernel(const double* __restrict__ alpha, const TP* __restrict__ p,
                                                        double* __restrict__ x, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  const double a = alpha[nm.b];
  for (int i = nm.node0; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    x[o] += a * (double)p[o];
  }
}

// Directions kept before the iterate is touched: 10 fp32 slots (5 fp64) -- solves of up to 10 iterations (the 9 + 9 of a
// per-element field per sample) form x ONCE, in pcg_finish_kernel; round 3's 6 slots flushed such a solve twice
constexpr int kRingSlots = 10;

// End of the solve: x += alpha p (the pending iterate update of the fused loop; p == NULL: none) + z / rs, where
// z = V(r) is the preconditioned residual of the FINAL iterate -- every iteration ends with that V-cycle (its r.z is
// the error estimate the stop is decided on), and samples that stopped earlier kept r, hence z, unchanged since.
// Adding it is one step of the stationary multigrid iteration: e <- (I - M^-1 A) e, a further reduction by the
// V-cycle's own convergence factor (< 0.3) for no extra pass.
template <typename TP>
__global__ __launch_bounds__(256) void pcg_finish_kernel(const double* __restrict__ alpha, const TP* __restrict__ p,
                                                          long long slot_stride, int j0, int count, int n_slots,
                                                          const TP* __restrict__ z, const double* __restrict__ rs,
                                                          double* __restrict__ x, int n, int Bp, int unroll = 0) {
  // x += sum_{j = j0 .. j0 + count - 1} alpha_j p_j (+ z / rs): direction j lives in slot j % n_slots of `p`, its
  // step lengths in row j % n_slots of `alpha` (0 for samples that had stopped)
  const NodeMap nm = node_map(Bp);
  const double zi = z ? (rs ? 1.0 / rs[nm.b] : 1.0) : 0.0;   // rs is a power of two: exact
  double a[kRingSlots];
#pragma unroll
  for (int k = 0; k < kRingSlots; ++k) a[k] = k < count ? alpha[(long long)((j0 + k) % n_slots) * Bp + nm.b] : 0.0;
  int i = nm.node0;
  if (unroll)   // two nodes per trip: twice the loads in flight per wave (same operations per node)
    for (; (i64)i + nm.stride < n; i += 2 * nm.stride) {
      const i64 o0 = (i64)i * Bp + nm.b, o1 = (i64)(i + nm.stride) * Bp + nm.b;
      double v0 = x[o0], v1 = x[o1];
      TP z0 = z ? z[o0] : (TP)0, z1 = z ? z[o1] : (TP)0;
      TP p0[kRingSlots], p1[kRingSlots];
#pragma unroll
      for (int k = 0; k < kRingSlots; ++k)
        if (k < count) {
          const long long so = (long long)((j0 + k) % n_slots) * slot_stride;
          p0[k] = p[so + o0];
          p1[k] = p[so + o1];
        }
      if (z) { v0 += zi * (double)z0; v1 += zi * (double)z1; }
#pragma unroll
      for (int k = 0; k < kRingSlots; ++k)
        if (k < count) { v0 += a[k] * (double)p0[k]; v1 += a[k] * (double)p1[k]; }
      x[o0] = v0;
      x[o1] = v1;
    }
  for (; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    double v = x[o];
    if (z) v += zi * (double)z[o];
#pragma unroll
    for (int k = 0; k < kRingSlots; ++k)
      if (k < count) v += a[k] * (double)p[(long long)((j0 + k) % n_slots) * slot_stride + o];
    x[o] = v;
  }
}

// p = z + beta p   (first: p = z)
template <typename TV>
__global__ __launch_bounds__(256) void pcg_update_p_kernel(const TV* __restrict__ z, const double* __restrict__ beta,
                                                            double* __restrict__ p, int first, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  const double be = first ? 0.0 : beta[nm.b];
  for (int i = nm.node0; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    p[o] = first ? (double)z[o] : (double)z[o] + be * p[o];
  }
}

// ---- per-sample scalars -----------------------------------------------------------------------
struct PcgScalars {
  double *rz, *alpha, *beta, *bb, *tol2;
  double* rs;             // per-sample power of two ~ 1 / |b| applied to the fp32 copies of the residual (NULL: none)
  const double* maxdiag;  // S_FLOOR: per-sample (Bv entries) max diagonal of the unscaled level-0 matrix
  const double* scale;    // S_FLOOR: per-sample operator scale (may be NULL)
  int Bv;
  int *active, *iters, *n_active;
  // Energy-norm stop.  With a multigrid preconditioner M ~ A the dot r.z = r^T M^-1 r the CG computes anyway is the
  // squared ENERGY norm of the error e^T A e (to the spectral equivalence of M and A, ~20 %), and b.x that of the
  // solution: sample b stops once r.z <= tol_e2 * energy[b].
  double* energy;         // u^T A u >= (b.x0)^2 / (x0^T A x0) (Cauchy-Schwarz in the A inner product: a LOWER bound for
                          // any x0, tight for the full-multigrid start), or r0.z0 = b^T M^-1 b from a zero start
  double* rr;             // last r.r per sample (guard of the energy stop)
  double* est;            // out: last estimate sqrt(r.z / energy) per sample
  double tol_e2;          // 0: residual criterion only
  int e_max_it;           // the energy rule is trusted within this many iterations (10 at tol_energy 1e-11, one more per decade)
  int have_energy;        // energy[] was set from the full-multigrid start (S_ENERGY)
  int* rule;              // out: which rule ended each sample: 0 none (iteration cap), 1 residual, 2 energy-norm estimate
};
enum { S_INIT = 0, S_RZ0 = 1, S_ALPHA = 2, S_CONV = 3, S_BETA = 4, S_RELRES = 5, S_SUM = 6, S_FLOOR = 7, S_ENERGY = 8,
       S_ENERGY2 = 9 };

// First stage of a long partial list: block (x, y) sums the rows k = y, y + S, y + 2 S, ... of `part` for the samples of
// chunk x into row y of `slice` (S = gridDim.y rows).  One block of pcg_scalar_kernel summing 1500-2000 rows reads ~1 MB
// through ONE CU (23 us per phase at 1024^2 x 256, 43 phases per step); 16 blocks + the final phase over 16 rows take ~8.
// Fixed assignment and fixed order of additions: bitwise reproducible.
constexpr int kScalarSlices = 16;
__global__ __launch_bounds__(256) void pcg_slice_kernel(const double* __restrict__ part, int nblk, int Bp,
                                                         double* __restrict__ slice) {
  __shared__ double lds[4 * kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * kWave + lane;
  const int S = gridDim.y, y = blockIdx.y;
  double s0 = 0.0, s1 = 0.0;
  if (b < Bp) {
    int k = y + S * wave;
    for (; k + 4 * S < nblk; k += 8 * S) {     // two independent chains per wave, four waves: eight loads in flight
      s0 += part[(i64)k * Bp + b];
      s1 += part[(i64)(k + 4 * S) * Bp + b];
    }
    if (k < nblk) s0 += part[(i64)k * Bp + b];
  }
  lds[wave * kWave + lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && b < Bp)
    slice[(i64)y * Bp + b] = (lds[lane] + lds[kWave + lane]) + (lds[2 * kWave + lane] + lds[3 * kWave + lane]);
}

// 1024 threads: lanes over samples, 16 waves over slices of the partial list (fixed order)
__global__ __launch_bounds__(1024) void pcg_scalar_kernel(int phase, const double* __restrict__ part, int nblk, int Bp,
                                                           double tol, PcgScalars S, double* __restrict__ relres) {
  __shared__ double lds[16 * kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * kWave + lane;
  double s = 0.0;
  if (b < Bp) {  // 4 independent chains keep several loads in flight (fixed order: still deterministic)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = wave;
    for (; k + 48 < nblk; k += 64) {
      s0 += part[(i64)k * Bp + b];
      s1 += part[(i64)(k + 16) * Bp + b];
      s2 += part[(i64)(k + 32) * Bp + b];
      s3 += part[(i64)(k + 48) * Bp + b];
    }
    for (; k < nblk; k += 16) s0 += part[(i64)k * Bp + b];
    s = (s0 + s1) + (s2 + s3);
  }
  lds[wave * kWave + lane] = s;
  __syncthreads();
  if (wave != 0 || b >= Bp) return;
  double a = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) a += lds[w * kWave + lane];
  switch (phase) {
    case S_INIT:  // a = b.b
      S.bb[b] = a;
      if (S.rs) S.rs[b] = a > 0.0 ? ldexp(1.0, -ilogb(sqrt(a))) : 1.0;  // rs |b| in [1, 2)
      S.tol2[b] = tol * tol * a;
      S.active[b] = a > 0.0 ? 1 : 0;
      S.rule[b] = a > 0.0 ? 0 : 1;   // a zero right-hand side is solved by x = 0
      S.iters[b] = 0;
      S.alpha[b] = 0.0;
      S.beta[b] = 0.0;
      S.rz[b] = 0.0;
      S.rr[b] = a;
      break;
    case S_RZ0: {  // a = r.z
      S.rz[b] = a;
      const double rs2 = S.rs ? S.rs[b] * S.rs[b] : 1.0;   // z carries rs, so does the copy of r it is dotted with
      if (!S.have_energy) S.energy[b] = a / rs2;           // zero start: r0.z0 = b^T M^-1 b ~ u^T A u
      S.est[b] = S.energy[b] > 0.0 ? sqrt(a / rs2 / S.energy[b]) : 0.0;
      break;
    }
    case S_ENERGY:  // a = b.x0
      S.energy[b] = a;
      break;
    case S_ENERGY2:  // a = x0^T A x0: energy of the solution >= (b.x0)^2 / (x0^T A x0), whatever x0 is
      S.energy[b] = (a > 0.0 && S.energy[b] > 0.0) ? S.energy[b] * (S.energy[b] / a) : 0.0;  // no squares: any data magnitude
      break;
    case S_ALPHA:  // a = p.Ap
      // with scaled fp32 copies z, p and Ap carry the factor rs and both dots rs^2: alpha is unchanged, and the
      // updates x += alpha p, r -= alpha Ap take alpha / rs
      S.alpha[b] = (S.active[b] && a > 0.0) ? (S.rz[b] / a) / (S.rs ? S.rs[b] : 1.0) : 0.0;
      if (b == 0) *S.n_active = 0;
      break;
    case S_CONV:  // a = r.r after the update
      if (S.active[b]) {
        S.iters[b] += 1;
        S.rr[b] = a;
        if (a <= S.tol2[b]) {
          S.active[b] = 0;
          S.rule[b] = 1;
        }
      }
      break;
    case S_BETA:  // a = r.z (new)
      if (S.active[b]) {
        S.beta[b] = a / S.rz[b];
        S.rz[b] = a;
        const double rs2 = S.rs ? S.rs[b] * S.rs[b] : 1.0;
        const double e2 = a / rs2;                                   // ~ e^T A e of the current iterate
        S.est[b] = S.energy[b] > 0.0 ? sqrt(fmax(e2, 0.0) / S.energy[b]) : 0.0;
        // The estimate stands on M ~ A.  It is trusted only where the iteration is visibly healthy: a positive r.z
        // (a V-cycle that lost definiteness -- obtuse meshes, fp32 overflow -- can return anything), a positive
        // energy bound, and a residual already within 1e4 x the target (|r|/|b| is 6e-9 .. 2e-10 at the iterations
        // where the bench workload stops); otherwise the residual criterion decides alone.
        // ... and only within the first 10 iterations (at tol_energy = 1e-11; one more per decade asked beyond that --
        // a healthy cycle gains a decade per iteration): r.z equals e^T A e up to lambda_min(M^-1 A), and a CG that needs
        // more than that to get here is telling that this constant is small (skewed lattices with pinned interior
        // nodes: 12 and 35 iterations, error 8e-11 at an estimate of 1e-11).
        if (S.tol_e2 > 0.0 && a > 0.0 && S.energy[b] > 0.0 && e2 <= S.tol_e2 * S.energy[b] &&
            S.rr[b] <= 1e8 * S.tol_e2 * S.bb[b] && S.iters[b] <= S.e_max_it) {
          S.active[b] = 0;
          S.rule[b] = 2;
        }
      } else {
        S.beta[b] = 0.0;
      }
      if (S.active[b]) atomicAdd(S.n_active, 1);                      // read by the host after this phase
      break;
    case S_SUM:  // plain per-sample total
      relres[b] = a;
      break;
    case S_FLOOR: {  // a = |x0|^2.  fp64 cannot bring |b - A x| below ~ u |A| |x| (u = 2^-53): the recurrence
      // residual keeps falling past that level but the iterate no longer improves, so the stop is floored at
      // HALF of it -- the backward-stability level a direct fp64 solve (the reference's LU) reaches too.
      const double anorm = 2.0 * S.maxdiag[S.Bv == 1 ? 0 : b] * (S.scale ? S.scale[b] : 1.0);  // >= |A|_inf
      const double fl = 0.5 * 1.1102230246251565e-16 * anorm;
      const double floor2 = fl * fl * a;
      if (floor2 > S.tol2[b]) S.tol2[b] = floor2;
      break;
    }
    default:  // S_RELRES: a = |b - A x|^2
      relres[b] = S.bb[b] > 0.0 ? sqrt(a / S.bb[b]) : 0.0;
  }
}

// ---- opt-in timing of the step's main kernels INSIDE the solver loop (bench.py's roofline entries) ----------
// HIP events on the solve's stream around the fine-level launch of each kernel family, read after the per-iteration
// stream synchronisation the loop performs anyway.  Per calling thread (the adjoint solves run on autograd's thread
// and are not sampled); the only hidden state of the library, and only while enabled.
enum { KP_CGSTEP = 0, KP_UPDATE = 1, KP_FIRST2 = 2, KP_RESTRICT = 3, KP_PROLONG = 4, KP_SWEEP = 5, KP_COUNT = 6 };
struct KernelProfile {
  bool on = false;
  hipEvent_t e0[KP_COUNT] = {}, e1[KP_COUNT] = {};
  bool have[KP_COUNT] = {};
  double ms[KP_COUNT] = {};
  long long n[KP_COUNT] = {};
};
thread_local KernelProfile g_kp;
inline void kp_begin(int id, hipStream_t st) {
  if (g_kp.on) (void)hipEventRecord(g_kp.e0[id], st);
}
inline void kp_end(int id, hipStream_t st) {
  if (g_kp.on) {
    (void)hipEventRecord(g_kp.e1[id], st);
    g_kp.have[id] = true;
  }
}
inline void kp_collect() {  // call with the stream idle: every recorded event has completed
  if (!g_kp.on) return;
  for (int id = 0; id < KP_COUNT; ++id) {
    if (!g_kp.have[id]) continue;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, g_kp.e0[id], g_kp.e1[id]) == hipSuccess) {
      g_kp.ms[id] += ms;
      g_kp.n[id] += 1;
    }
    g_kp.have[id] = false;
  }
}

// ---- host-side hierarchy --------------------------------------------------------------------
constexpr int kMaxLevels = 16;

struct Hier {
  Level lev[kMaxLevels];
  int nl, Bv, Bp;
  const double* scale;
  double omega[8];  // per-sweep damping (Chebyshev-weighted Jacobi); post-smoothing runs them in reverse
  int nu, n_coarse, fmg_coarse_cycles;
  int fuse;  // 0: four single-stage strip passes per level; 1 / 2: fused two-stage passes, samples per lane
  int pre4;  // the fused PRE pass may take four samples per lane (fused_spl)
  int dense_mfma;  // coarsest-level dense solve of an fp32 cycle on the matrix cores (0: scalar-load kernel)
  double coarse_lmax;  // upper bound of the spectrum of D^-1 A on the coarsest level (2 for an M-matrix)
  // per-level work vectors
  void *xa[kMaxLevels], *xb[kMaxLevels], *res[kMaxLevels], *rhs[kMaxLevels];  // TV vectors of the V-cycle
  void *bF[kMaxLevels], *xF[kMaxLevels];  // full-multigrid start: restricted right-hand sides, iterates
};

inline dim3 lgrid(int n, int Bp) { return node_grid(n, Bp, 2048); }  // <= kPartBlocks partial rows

// bpn = algorithmic bytes per (node, sample) of the launch, for diffhe_traffic_account
#define LAUNCH(bpn, kernel, n, ...)                                                  \
  do {                                                                               \
    diffhe::account((double)(bpn) * (double)(n) * H.Bp);                             \
    hipLaunchKernelGGL(kernel, lgrid((n), H.Bp), dim3(256), 0, st, __VA_ARGS__);     \
  } while (0)
// per-sample matrices: bytes of the nd stored diagonals (fp64) per node; a batch-shared matrix is amortised to 0
#define MATB(L) (H.Bv == 1 ? 0.0 : 8.0 * (L).nd)

// ---- operator dispatch: strip kernels on big levels, simple kernels on small ones ----------------
// Each returns the number of partial blocks it wrote (when `part` != NULL).  TV is the storage
// type of the vectors (double, or float inside a single-precision preconditioner); arithmetic is
// always fp64 in registers.
template <typename TV>
int op_jacobi(const Hier& H, int l, const TV* rhs, const TV* xin, TV* xout, double omega, double* part,
              hipStream_t st) {
  const Level& L = H.lev[l];
  // the plain sweep runs at the HBM rate of its real traffic either way (0.69 ms one sample per lane, 0.70-0.72 two):
  // it keeps the one-sample kernel; DIFFHE_S2_SWEEP=1 switches it over too
  static const int sweep2 = getenv("DIFFHE_S2_SWEEP") ? atoi(getenv("DIFFHE_S2_SWEEP")) : 0;
  StripGeom g;
  const bool two = sweep2 ? strip2_pick<TV>(L, H.Bv, H.Bp, strip_cols<TV>(), &g)
                          : (g = strip_geom(L, H.Bp, strip_cols<TV>()), false);
  if (g.use && xin) {
    if (l == 0) kp_begin(KP_SWEEP, st);
    if (two)
      launch_strip2<M_JACOBI, false, F_NONE, 4>(L, H.scale, (const float*)xin, (const float*)rhs, (float*)xout, omega, 0.0,
                                                part, H.Bp, g, st);
    else
      launch_strip<TV, M_JACOBI, false, F_NONE, TV, strip_cols<TV>()>(L, H.Bv, H.scale, xin, rhs, xout, omega, 0.0, part,
                                                                     H.Bp, g, st);
    if (l == 0) kp_end(KP_SWEEP, st);
    return g.ncb * g.nrc;
  }
  LAUNCH((xin ? 3 : 2) * sizeof(TV) + MATB(L), dia_jacobi_kernel<TV>, L.n, L, H.Bv, H.scale, rhs, xin, xout, omega, part, H.Bp);
  return lgrid(L.n, H.Bp).x;
}

// two sweeps from a zero guess in one pass over rhs: x1 = w0 D^-1 rhs is formed on the fly
template <typename TV>
int op_jacobi_first2(const Hier& H, int l, const TV* rhs, TV* xa, TV* xb, double w0, double w1, double* part,
                     TV** result, hipStream_t st) {
  const Level& L = H.lev[l];
  StripGeom g;
  const bool two = strip2_pick<TV>(L, H.Bv, H.Bp, strip_cols<TV>(), &g);
  if (g.use) {
    if (l == 0) kp_begin(KP_FIRST2, st);
    if (two)
      launch_strip2<M_JACOBI, true, F_NONE, 4>(L, H.scale, (const float*)nullptr, (const float*)rhs, (float*)xa, w1, w0,
                                               part, H.Bp, g, st);
    else
      launch_strip<TV, M_JACOBI, true, F_NONE, TV, strip_cols<TV>()>(L, H.Bv, H.scale, (const TV*)nullptr, rhs, xa, w1, w0,
                                                                    part, H.Bp, g, st);
    if (l == 0) kp_end(KP_FIRST2, st);
    *result = xa;
    return g.ncb * g.nrc;
  }
  LAUNCH(2 * sizeof(TV) + MATB(L) / L.nd, dia_jacobi_kernel<TV>, L.n, L, H.Bv, H.scale, rhs, (const TV*)nullptr, xa, w0, (double*)nullptr, H.Bp);
  LAUNCH(3 * sizeof(TV) + MATB(L), dia_jacobi_kernel<TV>, L.n, L, H.Bv, H.scale, rhs, (const TV*)xa, xb, w1, part, H.Bp);
  *result = xb;
  return lgrid(L.n, H.Bp).x;
}

template <typename TV>
int op_residual(const Hier& H, int l, const TV* rhs, const TV* x, TV* res, double* part, hipStream_t st,
                int dot_bx = 0, double* part2 = nullptr) {
  const Level& L = H.lev[l];
  const StripGeom g = strip_geom(L, H.Bp);
  if (g.use) {
    Extra ex{};
    ex.dot_bx = dot_bx;
    ex.part2 = part2;
    launch_strip<TV, M_RESID, false>(L, H.Bv, H.scale, x, rhs, res, 0.0, 0.0, part, H.Bp, g, st, ex);
    return g.ncb * g.nrc;
  }
  LAUNCH((res ? 3 : 2) * sizeof(TV) + MATB(L), dia_residual_kernel<TV>, L.n, L, H.Bv, H.scale, rhs, x, res, part, H.Bp,
         dot_bx, part2);
  return lgrid(L.n, H.Bp).x;
}

int op_apply_dot(const Hier& H, const double* x, double* y, double* part, hipStream_t st) {
  const Level& L = H.lev[0];
  const StripGeom g = strip_geom(L, H.Bp);
  if (g.use) {
    launch_strip<double, M_APPLY, false>(L, H.Bv, H.scale, x, (const double*)nullptr, y, 0.0, 0.0, part, H.Bp, g, st);
    return g.ncb * g.nrc;
  }
  LAUNCH(16.0 + MATB(L), dia_apply_dot_kernel, L.n, L, H.Bv, H.scale, x, y, part, H.Bp);
  return lgrid(L.n, H.Bp).x;
}

// Coarsest-level solve with a precomputed dense inverse of the batch-shared level matrix (K_1 of a factored
// operator, plan-constant): x[i, b] = (1 / s_b) sum_j inv[i, j] rhs[j, b].  A wave owns RPW rows x 64 samples: rhs is
// read once per wave (lanes over samples, 256-512 B per load, L2-resident at these sizes), the inverse arrives as
// wave-uniform scalar loads.  33^2 nodes x 256 samples: 3e8 multiply-adds in ONE launch instead of the ~45 launches
// (5 levels of sweeps, transfers and the Chebyshev solve of the 3 x 3 grid) it replaces -- those were
// launch-latency-bound at ~5 us each.  Exact (to fp32/fp64 rounding) and symmetric, so the cycle stays an SPD
// preconditioner.
template <typename TV, int RPB>
__global__ __launch_bounds__(256) void mg_dense_solve_kernel(int n, const TV* __restrict__ inv,
                                                              const double* __restrict__ scale,
                                                              const TV* __restrict__ rhs, TV* __restrict__ x,
                                                              double* __restrict__ part, int Bp) {
  // block = RPB rows x 64 samples; its 4 waves split the sum over j (a quarter each, 4 loads in flight per wave:
  // one wave per SIMD with one dependent L2 load per step ran 260 us), partial rows meet in LDS
  __shared__ double red[4 * RPB * kWave];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y * kWave + lane;
  const int i0 = blockIdx.x * RPB;
  const int nq = (n + 3) / 4;
  const int j0 = wave * nq, j1 = (j0 + nq < n) ? j0 + nq : n;
  double acc[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) acc[r] = 0.0;
  const TV* __restrict__ row[RPB];
#pragma unroll
  for (int r = 0; r < RPB; ++r) row[r] = inv + (i64)(i0 + r < n ? i0 + r : n - 1) * n;
  const TV* __restrict__ rb = rhs + b;
  int j = j0;
  for (; j + 4 <= j1; j += 4) {
    const double v0 = (double)rb[(i64)j * Bp], v1 = (double)rb[(i64)(j + 1) * Bp];
    const double v2 = (double)rb[(i64)(j + 2) * Bp], v3 = (double)rb[(i64)(j + 3) * Bp];
#pragma unroll
    for (int r = 0; r < RPB; ++r)
      acc[r] += ((double)row[r][j] * v0 + (double)row[r][j + 1] * v1) + ((double)row[r][j + 2] * v2 + (double)row[r][j + 3] * v3);
  }
  for (; j < j1; ++j) {
    const double v = (double)rb[(i64)j * Bp];
#pragma unroll
    for (int r = 0; r < RPB; ++r) acc[r] += (double)row[r][j] * v;
  }
#pragma unroll
  for (int r = 0; r < RPB; ++r) red[(wave * RPB + r) * kWave + lane] = acc[r];
  __syncthreads();
  double s = 0.0;
  if (wave == 0) {
    const double si = scale ? 1.0 / scale[b] : 1.0;
#pragma unroll
    for (int r = 0; r < RPB; ++r) {
      if (i0 + r < n) {
        const double t = (red[r * kWave + lane] + red[(RPB + r) * kWave + lane]) +
                         (red[(2 * RPB + r) * kWave + lane] + red[(3 * RPB + r) * kWave + lane]);
        const double xo = si * t;
        x[(i64)(i0 + r) * Bp + b] = (TV)xo;
        s += (double)rb[(i64)(i0 + r) * Bp] * xo;
      }
    }
    if (part) part[(i64)blockIdx.x * Bp + b] = s;  // rhs . x partials (only when this level is the whole cycle)
  }
}

// The same product on the matrix cores (fp32 storage only): X (n x Bp) = inv (n x n) . R (n x Bp) is a plain GEMM, the one
// GEMM-shaped piece of the path.  v_mfma_f32_32x32x2_f32: a block owns 32 rows x 32 samples, its 4 waves split the sum
// over j and meet in LDS.  A-operand: lane l supplies inv[i0 + l % 32][j + l / 32] -- read as inv[j + l / 32][i0 + l % 32]
// (the inverse of a symmetric matrix is symmetric), so the 32 lanes of a half-wave read 128 contiguous bytes;
// B-operand: rhs[j + l / 32][b0 + l % 32], contiguous as well.  Accumulates in fp32 where the scalar kernel above
// accumulates in fp64: inside an fp32-stored preconditioner the 1e-6 this costs on the coarsest-level solve is immaterial
// (same iteration counts, tests/test_robustness.py).  D layout: lane l holds column l % 32, rows 8 (v / 4) + 4 (l / 32) + v % 4.
typedef float f16v __attribute__((ext_vector_type(16)));
template <int NW>
__global__ __launch_bounds__(64 * NW) void mg_dense_mfma_kernel(int n, const float* __restrict__ inv,
                                                             const double* __restrict__ scale,
                                                             const float* __restrict__ rhs, float* __restrict__ x, int Bp) {
  __shared__ float red[NW - 1][16][kWave];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int il = lane & 31, kh = lane >> 5;
  const int i0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
  const int ia = (i0 + il < n) ? i0 + il : n - 1;
  const float* __restrict__ pa = inv + ia;
  const float* __restrict__ pb = rhs + b0 + il;
  const int nkp = (n + 1) >> 1, q = (nkp + NW - 1) / NW;
  const int kp0 = wave * q, kp1 = (kp0 + q < nkp) ? kp0 + q : nkp;
  f16v acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
  int kp = kp0;
  for (; kp + 4 <= kp1; kp += 4) {
    float a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = 2 * (kp + u) + kh;
      const bool ok = j < n;
      const int jj = ok ? j : 0;
      a[u] = ok ? pa[(i64)jj * n] : 0.0f;
      b[u] = ok ? pb[(i64)jj * Bp] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
  }
  for (; kp < kp1; ++kp) {
    const int j = 2 * kp + kh;
    const bool ok = j < n;
    const int jj = ok ? j : 0;
    const float a = ok ? pa[(i64)jj * n] : 0.0f;
    const float b = ok ? pb[(i64)jj * Bp] : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < 16; ++v) red[wave - 1][v][lane] = acc[v];
  }
  __syncthreads();
  if (wave == 0) {
    const float si = scale ? (float)(1.0 / scale[b0 + il]) : 1.0f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int i = i0 + 8 * (v >> 2) + 4 * kh + (v & 3);
      float t = acc[v];
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) t += red[w][v][lane];
      if (i < n) x[(i64)i * Bp + b0 + il] = si * t;
    }
  }
}

// The same product for batches below a wave (Bp = 1 .. 32, the unbatched call shape of the reference): one wave per
// row, lanes over the columns j, a wave reduction per sample.
template <typename TV>
__global__ __launch_bounds__(64) void mg_dense_small_kernel(int n, const TV* __restrict__ inv,
                                                             const double* __restrict__ scale,
                                                             const TV* __restrict__ rhs, TV* __restrict__ x,
                                                             double* __restrict__ part, int Bp) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const TV* __restrict__ row = inv + (i64)i * n;
  for (int b = 0; b < Bp; ++b) {
    double s = 0.0;
    for (int j = lane; j < n; j += kWave) s += (double)row[j] * (double)rhs[(i64)j * Bp + b];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) {
      const double xo = (scale ? 1.0 / scale[b] : 1.0) * s;
      x[(i64)i * Bp + b] = (TV)xo;
      if (part) part[(i64)i * Bp + b] = (double)rhs[(i64)i * Bp + b] * xo;   // one partial row per matrix row
    }
  }
}

// Coarsest-level solve: Chebyshev semi-iteration for D^-1 A with the spectrum bounds of the P1 Laplacian
// on an nx x ny lattice, lambda in [ (1 - cos(pi/nx))/2 + (1 - cos(pi/ny))/2 , 2 ]; the lower bound is halved
// for safety (below it the polynomial stays < 1, it only damps less).  The degree follows from the size, so a
// 3 x 3 coarsest grid costs ~5 steps and a 125 x 125 one (sizes that cannot be halved further) ~170 --
// a fixed polynomial in A, hence still a symmetric preconditioner.  Returns the solution buffer.
template <typename TV>
TV* coarse_solve(const Hier& H, int l, const TV* rhs, double* part, int* nblocks, hipStream_t st) {
  const Level& L = H.lev[l];
  if (L.inv && H.Bv == 1 && L.n <= kPartBlocks) {  // dense inverse of the shared level matrix: one launch
    diffhe::account(2.0 * sizeof(TV) * (double)L.n * H.Bp);
    static const int use_mfma = getenv("DIFFHE_DENSE_MFMA") ? atoi(getenv("DIFFHE_DENSE_MFMA")) : 1;
    if (use_mfma && H.dense_mfma && sizeof(TV) == 4 && H.Bp >= kWave && !part) {
      // 8 waves per 32 x 32 tile split the sum over j: 1089 nodes x 256 samples = 280 blocks, a chain of 17 dependent
      // 4-step groups per wave (4 waves: 28 us, the scalar fp64-accumulating kernel: 51 us)
      if (use_mfma == 4)
        hipLaunchKernelGGL(mg_dense_mfma_kernel<4>, dim3((L.n + 31) / 32, H.Bp / 32), dim3(256), 0, st, L.n, (const float*)L.inv,
                           H.scale, (const float*)rhs, (float*)H.xa[l], H.Bp);
      else
        hipLaunchKernelGGL(mg_dense_mfma_kernel<8>, dim3((L.n + 31) / 32, H.Bp / 32), dim3(512), 0, st, L.n, (const float*)L.inv,
                           H.scale, (const float*)rhs, (float*)H.xa[l], H.Bp);
      if (nblocks) *nblocks = 0;
    } else if (H.Bp >= kWave) {
      constexpr int RPB = 4;
      const dim3 grid((L.n + RPB - 1) / RPB, H.Bp / kWave);
      hipLaunchKernelGGL((mg_dense_solve_kernel<TV, RPB>), grid, dim3(256), 0, st, L.n, (const TV*)L.inv, H.scale, rhs,
                         (TV*)H.xa[l], part, H.Bp);
      if (nblocks) *nblocks = grid.x;
    } else {
      hipLaunchKernelGGL(mg_dense_small_kernel<TV>, dim3(L.n), dim3(64), 0, st, L.n, (const TV*)L.inv, H.scale, rhs,
                         (TV*)H.xa[l], part, H.Bp);
      if (nblocks) *nblocks = L.n;
    }
    return (TV*)H.xa[l];
  }
  const double pi = 3.14159265358979323846;
  const double lmin = 0.5 * (0.5 * (1.0 - cos(pi / L.nx)) + 0.5 * (1.0 - cos(pi / L.ny)));
  const double lmax = H.coarse_lmax;
  int deg = (int)ceil(1.5 * sqrt(lmax / lmin));
  if (deg < H.n_coarse) deg = H.n_coarse;
  if (deg > 400) deg = 400;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  TV* xa = (TV*)H.xa[l];
  TV* xb = (TV*)H.xb[l];
  TV* d = (TV*)H.res[l];  // the coarsest level never restricts: its residual buffer holds d
  double rho = 1.0 / sigma;
  LAUNCH(3 * sizeof(TV) + MATB(L) / L.nd, dia_cheby_kernel<TV>, L.n, L, H.Bv, H.scale, rhs, (const TV*)nullptr, (const TV*)nullptr, xa, d, 0.0,
         1.0 / theta, (deg == 1) ? part : (double*)nullptr, H.Bp);
  for (int k = 1; k < deg; ++k) {
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    LAUNCH(5 * sizeof(TV) + MATB(L), dia_cheby_kernel<TV>, L.n, L, H.Bv, H.scale, rhs, (const TV*)xa, (const TV*)d, xb, d, rho_new * rho,
           2.0 * rho_new / delta, (k == deg - 1) ? part : (double*)nullptr, H.Bp);
    rho = rho_new;
    TV* t = xa; xa = xb; xb = t;
  }
  if (nblocks) *nblocks = lgrid(L.n, H.Bp).x;
  return xa;
}

template <typename TV>
void launch_restrict(const Hier& H, const Level& F, const Level& C, const TV* r, TV* rc, hipStream_t st) {
  if (transfers2_ok(F, C, H.Bp, sizeof(TV))) {
    diffhe::account(((double)F.n / C.n + 1.0) * sizeof(TV) * (double)C.n * H.Bp);
    hipLaunchKernelGGL(mg_restrict2_kernel, transfer2_grid(C.n, H.Bp), dim3(256), 0, st, F, C, (const float*)r, (float*)rc, H.Bp);
  } else {
    LAUNCH(((double)F.n / C.n + 1.0) * sizeof(TV), mg_restrict_kernel<TV>, C.n, F, C, r, rc, H.Bp);
  }
}

template <typename TV>
void launch_prolong(const Hier& H, const Level& F, const Level& C, const TV* e, TV* x, int set, hipStream_t st) {
  if (transfers2_ok(F, C, H.Bp, sizeof(TV))) {
    diffhe::account(((set ? 1.0 : 2.0) + (double)C.n / F.n) * sizeof(TV) * (double)F.n * H.Bp);
    hipLaunchKernelGGL(mg_prolong2_kernel, transfer2_grid(F.n, H.Bp), dim3(256), 0, st, F, C, (const float*)e, (float*)x, H.Bp, set);
  } else {
    LAUNCH(((set ? 1.0 : 2.0) + (double)C.n / F.n) * sizeof(TV), mg_prolong_add_kernel<TV>, F.n, F, C, e, x, H.Bp, set);
  }
}

// residual + full-weighting restriction of level l in one pass (the residual is never stored): H.rhs[l + 1] = R (rhs - A x)
template <typename TV>
void resid_restrict(const Hier& H, int l, const TV* x, const TV* rhs_l, hipStream_t st) {
  const Level& L = H.lev[l];
  const Level& C = H.lev[l + 1];
  constexpr int CW = kRestrictCols;
  bool two = sizeof(TV) == 4 && strip2_ok(L, H.Bv, H.Bp);
  StripGeom g{true, 0, 0, 0};
  g.ncb = (C.W + 4 * CW - 1) / (4 * CW);
  for (int pass = 0; pass < 2; ++pass) {
    const int gy = H.Bp / (two ? 2 * kWave : kWave);
    int nrc = (6144 + g.ncb * gy - 1) / (g.ncb * gy);
    if (nrc > (C.ny + 1) / 4) nrc = (C.ny + 1) / 4;
    if (nrc < 1) nrc = 1;
    g.TR = (C.ny + 1 + nrc - 1) / nrc;  // coarse rows per tile
    g.nrc = (C.ny + 1 + g.TR - 1) / g.TR;
    if (!two || strip2_tile_fits(L, H.Bp, 2 * g.TR + 1)) break;
    two = false;                         // tiles beyond 32-bit offsets: the one-sample-per-lane kernel
  }
  Extra ex{};
  ex.cW = C.W;
  ex.bc = C.bc;
  if (l == 0) kp_begin(KP_RESTRICT, st);
  if (two)
    launch_strip2<M_RESID, false, F_RESTRICT, 2 * CW + 1>(L, H.scale, (const float*)x, (const float*)rhs_l,
                                                          (float*)H.rhs[l + 1], 0.0, 0.0, nullptr, H.Bp, g, st, ex);
  else
    launch_strip<TV, M_RESID, false, F_RESTRICT, TV, 2 * CW + 1>(L, H.Bv, H.scale, x, rhs_l, (TV*)H.rhs[l + 1], 0.0, 0.0,
                                                                   nullptr, H.Bp, g, st, ex);
  if (l == 0) kp_end(KP_RESTRICT, st);
}

// samples per lane of the fused passes on level L: per-sample matrices always two; a batch-shared matrix what the batch
// allows (H.fuse), the four-sample form for 3-diagonal levels only
// The PRE pass of a 3-diagonal batch-shared level takes FOUR samples per lane where the batch has whole waves of 256:
// half the vector-memory instructions per byte at half the waves (219 VGPRs, 2 waves per SIMD).  Measured on the
// 1024^2 x 256 bench, same box (gpurun_out/r4k): PRE 0.707 -> 0.659 ms; the POST pass (240 VGPRs) 0.998 -> 1.042 ms:
// it keeps two.  DIFFHE_FUSED_PRE4=0 switches the four-sample form off.
inline int fused_spl(const Hier& H, const Level& L, bool pre) {
  const int spl = H.Bv == 1 ? H.fuse : 2;
  static const int pre4 = getenv("DIFFHE_FUSED_PRE4") ? atoi(getenv("DIFFHE_FUSED_PRE4")) : 1;
  if (pre && pre4 && H.pre4 && H.Bv == 1 && spl == 2 && H.Bp % (4 * kWave) == 0 && L.nd == 3) return 4;
  return spl;
}

// Can level l of the fp32 cycle run the fused POST pass (and with it the initial-guess form of the cycle)?  Fills the
// tile geometries of the fused PRE (gpre) and POST (gpost) passes; returns the fused_ok mask (0: no fused pass here).
template <typename TV>
int fused_level(const Hier& H, int l, StripGeom* gpre, StripGeom* gpost) {
  if (l >= H.nl - 1) return 0;
  const Level& L = H.lev[l];
  const Level& C = H.lev[l + 1];
  const int fmask = (sizeof(TV) == 4 && H.fuse && H.nu == 2) ? fused_ok(L, H.Bv, H.Bp, H.scale) : 0;
  if (!fmask || !(L.nx == 2 * C.nx && L.ny == 2 * C.ny && strip_geom(L, H.Bp).use)) return 0;
  const int spl = fused_spl(H, L, false), spl_pre = fused_spl(H, L, true);
  const int nw = 4;   // waves per block (fused_pre_kernel: wider blocks measured slower)
  constexpr int CW = kRestrictCols;
  StripGeom g{true, 0, 0, 0};
  g.ncb = (C.W + nw * CW - 1) / (nw * CW);
  const int gy = H.Bp / (spl_pre * kWave);
  // ~6144 blocks whatever the samples per lane: the four-sample form gets tiles of half the height (6 instead of 11 coarse
  // rows at 1024^2 x 256).  Measured (gpurun_out/r4l, same box): 6 rows 0.660 ms, 11 rows 0.681, 16 rows 0.778 -- the
  // number of independent marches matters more than the halo rows
  int nrc = (6144 * 4 / nw + g.ncb * gy - 1) / (g.ncb * gy);
  // Levels of <= 300 columns cannot fill the GPU with 4-coarse-row tiles: shorter tiles (2 coarse rows going down, ~5 fine
  // rows going up) double the independent marches; -1.4 ms per 1024^2 step, neutral on the 513^2 level (gpurun_out/r5j, r5k)
  const bool small = L.W <= 300;
  const int cap = small ? (C.ny + 1) / 2 : (C.ny + 1) / 4;
  if (nrc > cap) nrc = cap;
  if (nrc < 1) nrc = 1;
  g.TR = (C.ny + 1 + nrc - 1) / nrc;  // coarse rows per tile
  if (const int tr = env_level_int("DIFFHE_FUSED_TR_PRE", l)) g.TR = tr;
  g.nrc = (C.ny + 1 + g.TR - 1) / g.TR;
  *gpre = g;
  *gpost = strip_geom(L, H.Bp, 4, spl, nw);
  if (small) {
    const int nyp = L.ny + 1;
    int tr = 4;
    while (gpost->ncb * ((nyp + tr - 1) / tr) > kPartBlocks) ++tr;
    gpost->TR = tr;
    gpost->nrc = (nyp + tr - 1) / tr;
  }
  if (const int tr = env_level_int("DIFFHE_FUSED_TR_POST", l)) {
    if (gpost->ncb * ((L.ny + 1 + tr - 1) / tr) <= kPartBlocks) {
      gpost->TR = tr;
      gpost->nrc = (L.ny + 1 + tr - 1) / tr;
    }
  }
  const bool fits = strip2_tile_fits(L, H.Bp, 2 * g.TR + 6) && strip2_tile_fits(L, H.Bp, gpost->TR + 5);
  return fits ? fmask : 0;
}

// z = V(rhs0): returns the buffer holding the result at level 0.  If rz_part != NULL the last
// fine sweep also leaves the partials of rhs0.z there (*rz_blocks of them).
// guess != NULL (only where fused_level(H, l0) & 2): the cycle starts from the initial guess P guess instead of 0 and
// returns the new ITERATE for the right-hand side rhs0 -- in exact arithmetic P guess + V(rhs0 - A P guess), without the
// prolongation, residual and addition passes of that form (full-multigrid start).
template <typename TV>
TV* vcycle(const Hier& H, const TV* rhs0, double* rz_part, int* rz_blocks, hipStream_t st, int l0 = 0,
           const TV* guess = nullptr) {
  const TV* rhs[kMaxLevels];
  TV* cur[kMaxLevels];
  bool fused[kMaxLevels];
  StripGeom gpost[kMaxLevels];
  rhs[l0] = rhs0;  // the cycle runs on levels l0 .. last (l0 > 0: inside full multigrid)
  const int last = H.nl - 1;
  for (int l = l0; l <= last; ++l) {  // downward leg
    const Level& L = H.lev[l];
    if (l == last) {  // coarsest level: Chebyshev solve (also the whole cycle when there is one level)
      int nb = 0;
      cur[l] = coarse_solve<TV>(H, l, rhs[l], (l0 == last) ? rz_part : nullptr, &nb, st);
      if (l0 == last && rz_part && rz_blocks) *rz_blocks = nb;
      break;
    }
    const int sweeps = H.nu;
    const bool only = false;
    TV* a = (TV*)H.xa[l];
    TV* b2 = (TV*)H.xb[l];
    fused[l] = false;
    StripGeom gpre;
    const int fmask = fused_level<TV>(H, l, &gpre, &gpost[l]);
    if (fmask) {
      const Level& C = H.lev[l + 1];
      const int spl = fused_spl(H, L, false);
      fused[l] = (fmask & 2) != 0;             // the way up: fused POST pass
      if (l == l0 && guess && fused[l]) {
        // two sweeps from the prolonged guess (the POST kernel with x = 0), then residual + restriction
        launch_fused_post(L, C, H.Bv, H.scale, (const float*)nullptr, (const float*)rhs[l], (const float*)guess, (float*)a,
                          H.omega[0], H.omega[1], nullptr, H.Bp, gpost[l], spl, st);
        resid_restrict<TV>(H, l, a, rhs[l], st);
        cur[l] = a;
        rhs[l + 1] = (const TV*)H.rhs[l + 1];
        continue;
      }
      if (fmask & 1) {
        // both sweeps + residual + restriction in ONE pass (fused_pre_kernel)
        if (l == 0) kp_begin(KP_FIRST2, st);
        launch_fused_pre(L, C, H.Bv, H.scale, (const float*)rhs[l], (float*)a, (float*)H.rhs[l + 1], H.omega[0],
                         H.omega[1], H.Bp, gpre, fused_spl(H, L, true), st);
        if (l == 0) kp_end(KP_FIRST2, st);
        cur[l] = a;
        rhs[l + 1] = (const TV*)H.rhs[l + 1];
        continue;
      }
    }
    int done;
    if (sweeps >= 2) {
      TV* resu;
      const int nb = op_jacobi_first2<TV>(H, l, rhs[l], a, b2, H.omega[0], H.omega[1 % H.nu],
                                          (only && sweeps == 2) ? rz_part : nullptr, &resu, st);
      if (only && sweeps == 2 && rz_blocks) *rz_blocks = nb;
      if (resu != a) { TV* t = a; a = b2; b2 = t; }
      done = 2;
    } else {
      const int nb = op_jacobi<TV>(H, l, rhs[l], nullptr, a, H.omega[0], (only && sweeps == 1) ? rz_part : nullptr, st);
      if (only && sweeps == 1 && rz_blocks) *rz_blocks = nb;
      done = 1;
    }
    for (int s = done; s < sweeps; ++s) {
      const int nb = op_jacobi<TV>(H, l, rhs[l], a, b2, H.omega[s % H.nu],
                                   (only && s == sweeps - 1) ? rz_part : nullptr, st);
      if (only && s == sweeps - 1 && rz_blocks) *rz_blocks = nb;
      TV* t = a; a = b2; b2 = t;
    }
    cur[l] = a;
    if (l < last) {
      const Level& C = H.lev[l + 1];
      if (strip_geom(L, H.Bp).use && L.nx == 2 * C.nx && L.ny == 2 * C.ny) {
        resid_restrict<TV>(H, l, a, rhs[l], st);
      } else {
        op_residual<TV>(H, l, rhs[l], a, (TV*)H.res[l], nullptr, st);
        launch_restrict<TV>(H, L, C, (const TV*)H.res[l], (TV*)H.rhs[l + 1], st);
      }
      rhs[l + 1] = (const TV*)H.rhs[l + 1];
    }
  }
  for (int l = last - 1; l >= l0; --l) {  // upward leg
    const Level& L = H.lev[l];
    const Level& C = H.lev[l + 1];
    TV* a = cur[l];
    TV* b2 = (a == (TV*)H.xa[l]) ? (TV*)H.xb[l] : (TV*)H.xa[l];
    if (fused[l]) {   // prolongation + correction + both post-sweeps (+ the partials of rhs . z) in ONE pass
      const bool dot = (l == l0) && rz_part;
      if (l == 0) kp_begin(KP_PROLONG, st);
      launch_fused_post(L, C, H.Bv, H.scale, (const float*)a, (const float*)rhs[l], (const float*)cur[l + 1], (float*)b2,
                        H.omega[1], H.omega[0], dot ? rz_part : nullptr, H.Bp, gpost[l], fused_spl(H, L, false), st);
      if (l == 0) kp_end(KP_PROLONG, st);
      if (dot && rz_blocks) *rz_blocks = gpost[l].ncb * gpost[l].nrc;
      cur[l] = b2;
      continue;
    }
    int s0 = 0;
    StripGeom g;
    const bool two = strip2_pick<TV>(L, H.Bv, H.Bp, strip_cols<TV>(), &g);
    if (g.use && L.nx == 2 * C.nx && L.ny == 2 * C.ny) {  // prolongate + correct + first post-sweep in one pass
      const bool lastsweep = (l == l0 && H.nu == 1);
      Extra ex{};
      ex.a0 = cur[l + 1]; ex.cW = C.W; ex.bc = L.bc;
      if (l == 0) kp_begin(KP_PROLONG, st);
      if (two)
        launch_strip2<M_JACOBI, false, F_PROLONG, 4>(L, H.scale, (const float*)a, (const float*)rhs[l], (float*)b2,
                                                     H.omega[H.nu - 1], 0.0, lastsweep ? rz_part : nullptr, H.Bp, g, st, ex);
      else
        launch_strip<TV, M_JACOBI, false, F_PROLONG, TV, strip_cols<TV>()>(L, H.Bv, H.scale, (const TV*)a, rhs[l], b2,
                                                                            H.omega[H.nu - 1], 0.0,
                                                                            lastsweep ? rz_part : nullptr, H.Bp, g, st, ex);
      if (l == 0) kp_end(KP_PROLONG, st);
      if (lastsweep && rz_blocks) *rz_blocks = g.ncb * g.nrc;
      TV* t = a; a = b2; b2 = t;
      s0 = 1;
    } else {
      launch_prolong<TV>(H, L, C, (const TV*)cur[l + 1], a, 0, st);
    }
    for (int s = s0; s < H.nu; ++s) {
      const bool lastsweep = (l == l0 && s == H.nu - 1);
      const int nb = op_jacobi<TV>(H, l, rhs[l], a, b2, H.omega[H.nu - 1 - s], lastsweep ? rz_part : nullptr, st);
      if (lastsweep && rz_blocks) *rz_blocks = nb;
      TV* t = a; a = b2; b2 = t;
    }
    cur[l] = a;
  }
  return cur[l0];
}

// Full multigrid start: solve on the coarsest level, then per level interpolate, take the residual
// and apply one V-cycle.  Gives the CG an iterate whose error is already smooth (about 3-4 CG
// iterations ahead of a zero guess) for ~0.8 of an iteration.  b0 = right-hand side in TV storage.
// *pending (optional): the fine level's last correction is NOT added to the returned iterate but handed back -- the
// caller's conversion pass (pcg_setx_kernel) adds the two in fp64, one pass over x less.
template <typename TV>
TV* fmg_start(const Hier& H, const TV* b0, hipStream_t st, const TV** pending = nullptr) {
  const int last = H.nl - 1;
  const TV* bl[kMaxLevels];
  bl[0] = b0;
  for (int l = 0; l < last; ++l) {
    launch_restrict<TV>(H, H.lev[l], H.lev[l + 1], bl[l], (TV*)H.bF[l + 1], st);
    bl[l + 1] = (const TV*)H.bF[l + 1];
  }
  {  // coarsest level: the V-cycle from `last` is n_coarse Jacobi sweeps
    TV* e = vcycle<TV>(H, bl[last], nullptr, nullptr, st, last);
    if (diffhe::check(hipMemcpyAsync(H.xF[last], e, (size_t)H.lev[last].n * H.Bp * sizeof(TV), hipMemcpyDeviceToDevice, st)))
      return nullptr;  // error text recorded for diffhe_last_hip_error()
  }
  const TV* coarse = (const TV*)H.xF[last];      // the iterate of level l + 1
  if (pending) *pending = nullptr;
  // 2: the initial-guess form of the cycle below the fine level only.  On the fine level it stores the full ITERATE in
  // fp32 between its passes where the correction form stores a correction ~1e-3 of it: rounding noise of 6e-8 |u|, rough,
  // ~3e-5 of the solution's energy -- measured one PCG iteration more (6 + 6 against 5 + 5 at 1024^2; gpurun_out/r4m)
  static const int guess_form = getenv("DIFFHE_FMG_GUESS") ? atoi(getenv("DIFFHE_FMG_GUESS")) : 2;
  for (int l = last - 1; l >= 0; --l) {
    const Level& L = H.lev[l];
    const int cycles = (l == 0) ? 1 : H.fmg_coarse_cycles;  // extra cycles on the cheap coarse levels
    StripGeom g1, g2;
    TV* x = (TV*)H.xF[l];
    int c0 = 0;
    if (guess_form && (l > 0 || guess_form == 1) && (fused_level<TV>(H, l, &g1, &g2) & 2)) {
      // levels with the fused passes: ONE cycle from the prolonged guess -- no prolongation, residual or addition pass
      TV* it = vcycle<TV>(H, bl[l], nullptr, nullptr, st, l, coarse);
      if (cycles == 1) {
        coarse = it;          // consumed by the first launch of the next level, before that level's cycle reuses the buffer
        continue;
      }
      if (diffhe::check(hipMemcpyAsync(x, it, (size_t)L.n * H.Bp * sizeof(TV), hipMemcpyDeviceToDevice, st))) return nullptr;
      c0 = 1;
    } else {
      launch_prolong<TV>(H, L, H.lev[l + 1], coarse, x, 1, st);
    }
    for (int c = c0; c < cycles; ++c) {
      op_residual<TV>(H, l, bl[l], (const TV*)x, (TV*)H.rhs[l], nullptr, st);
      TV* e = vcycle<TV>(H, (const TV*)H.rhs[l], nullptr, nullptr, st, l);
      if (l == 0 && c == cycles - 1 && pending) {
        *pending = e;
        break;
      }
      LAUNCH(3 * sizeof(TV), mg_add_kernel<TV>, L.n, (const TV*)e, x, L.n, H.Bp);
    }
    coarse = x;
  }
  return (TV*)coarse;
}


}  // namespace

extern "C" int diffhe_lattice_pcg_profile(int enable, double* total_ms, long long* launches) {
  if (total_ms) *total_ms = g_kp.ms[KP_CGSTEP];
  if (launches) *launches = g_kp.n[KP_CGSTEP];
  if (enable >= 0) {
    if (enable && !g_kp.e0[0]) {
      for (int id = 0; id < KP_COUNT; ++id)
        if (hipEventCreate(&g_kp.e0[id]) != hipSuccess || hipEventCreate(&g_kp.e1[id]) != hipSuccess) return DIFFHE_E_LAUNCH;
    }
    g_kp.on = enable != 0;
    for (int id = 0; id < KP_COUNT; ++id) {
      g_kp.ms[id] = 0.0;
      g_kp.n[id] = 0;
      g_kp.have[id] = false;
    }
  }
  return DIFFHE_OK;
}

extern "C" int diffhe_lattice_kernel_profile(int id, double* total_ms, long long* launches) {
  if (id < 0 || id >= KP_COUNT) return DIFFHE_E_BADARG;
  if (total_ms) *total_ms = g_kp.ms[id];
  if (launches) *launches = g_kp.n[id];
  return DIFFHE_OK;
}

// =========================================================================================
// C ABI
// =========================================================================================
static int fill_hier(Hier& H, const diffhe_mg_level* levels, int n_levels, int Bv, int Bp, const double* scale,
                     const double* omegas, int nu, int n_coarse) {
  if (!levels || n_levels < 1 || n_levels > kMaxLevels) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  if (Bv != 1 && Bv != Bp) return DIFFHE_E_BADARG;
  if (nu < 1 || nu > 8 || n_coarse < 1 || !omegas) return DIFFHE_E_BADARG;
  for (int l = 0; l < n_levels; ++l) {
    const diffhe_mg_level& s = levels[l];
    if (s.nx < 2 || s.ny < 2 || (s.nd != 3 && s.nd != 4) || !s.vals || !s.is_bc) return DIFFHE_E_BADARG;
    if (l > 0) {  // each level halves the previous one in x, in y, or in both
      const bool hx = levels[l - 1].nx == 2 * s.nx, hy = levels[l - 1].ny == 2 * s.ny;
      const bool kx = levels[l - 1].nx == s.nx, ky = levels[l - 1].ny == s.ny;
      if (!((hx && hy) || (hx && ky) || (kx && hy))) return DIFFHE_E_BADARG;
    }
    if ((long long)(s.nx + 1) * (s.ny + 1) > 0x7fffffffLL) return DIFFHE_E_TOOBIG;
    Level& L = H.lev[l];
    L.nx = s.nx; L.ny = s.ny; L.W = s.nx + 1; L.n = (s.nx + 1) * (s.ny + 1); L.nd = s.nd;
    L.v = s.vals; L.v32 = s.vals32; L.bc = s.is_bc; L.inv = s.dense_inv; L.shift = s.shift; L.rd32 = s.rdiag32; L.mk32 = s.mask32; L.o16 = (Bv == Bp && Bp > 1 && s.offdiag_scales) ? (const _Float16*)s.offdiag16 : nullptr;
    L.osc = s.offdiag_scales;
  }
  H.nl = n_levels; H.Bv = Bv; H.Bp = Bp; H.scale = scale; H.nu = nu; H.n_coarse = n_coarse;
  H.coarse_lmax = 2.0;
  H.fmg_coarse_cycles = 1;
  H.fuse = fused_mode();
  H.pre4 = 1;
  if (H.fuse == 2 && Bp % (2 * kWave) != 0) H.fuse = 1;   // one sample per lane where the batch is no multiple of 128
  H.dense_mfma = 1;
  for (int k = 0; k < 8; ++k) H.omega[k] = omegas[k < nu ? k : nu - 1];
  return DIFFHE_OK;
}

// V-cycle vectors, carved in units of doubles (fp32 vectors take half, rounded up to 64 B)
static long long carve(Hier& H, double* work, bool fp32) {
  long long off = 0;
  auto take = [&](long long cnt) {
    if (fp32) cnt = (cnt + 1) / 2;
    cnt = (cnt + 7) & ~7LL;
    double* p = work ? work + off : nullptr;
    off += cnt;
    return (void*)p;
  };
  for (int l = 0; l < H.nl; ++l) {
    const long long nb = (long long)H.lev[l].n * H.Bp;
    H.xa[l] = take(nb);
    H.xb[l] = take(nb);
    H.res[l] = take(nb);
    H.rhs[l] = take(nb);  // level 0: fp32 copy of the CG residual / FMG residual
    H.bF[l] = l > 0 ? take(nb) : nullptr;
    H.xF[l] = take(nb);
  }
  return off;
}

extern "C" long long diffhe_lattice_pcg_workspace_doubles(const diffhe_mg_level* levels, int n_levels, int Bp) {
  Hier H;
  const double w1 = 0.8;
  if (fill_hier(H, levels, n_levels, 1, Bp, nullptr, &w1, 1, 1)) return -1;
  const long long nb = (long long)H.lev[0].n * Bp;
  const long long nblk = lgrid(H.lev[0].n, Bp).x;
  (void)nblk;
  // r, the direction ring (kRingSlots fp32 = kRingSlots / 2 fp64 vectors), A p; the fp64 layout of the cycle is the larger
  return carve(H, nullptr, false) + (2 + kRingSlots / 2) * nb + 2LL * kPartBlocks * Bp + (32LL + kScalarSlices) * Bp + 64;
}

extern "C" int diffhe_lattice_pcg_solve(const diffhe_mg_level* levels, int n_levels, int Bv, const double* scale,
                                        const double* b, double* x, int Bp, double tol, double tol_energy, int max_iter,
                                        int nu, int n_coarse, const double* omegas_host, int precond_fp32, double* work,
                                        double* relres, double* err_est, int* iters, int* stop_rule, int* status_host,
                                        void* stream) {
  if (!b || !x || !work || !relres || !iters || !status_host || max_iter < 0) return DIFFHE_E_BADARG;
  Hier H;
  int rc = fill_hier(H, levels, n_levels, Bv, Bp, scale, omegas_host, nu, n_coarse);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const Level& L0 = H.lev[0];
  const int n = L0.n;
  const long long nb = (long long)n * Bp;
  const int nblk = lgrid(n, Bp).x;
  const bool f32 = (precond_fp32 & 1) != 0;
  const bool use_fmg = (precond_fp32 & 2) != 0 && H.nl > 1;
  const bool warm = (precond_fp32 & 32) != 0;   // x holds an initial guess (e.g. the previous step of an optimisation)
  H.fmg_coarse_cycles = 1 + ((precond_fp32 >> 2) & 3);
  if (precond_fp32 & 64) H.fuse = 0;             // bit 6: keep the four single-stage passes (A/B runs, tests)
  if (precond_fp32 & 128) H.dense_mfma = 0;      // bit 7: scalar-load dense coarse solve
  if (precond_fp32 & 512) H.pre4 = 0;            // bit 9: fused PRE pass with two samples per lane as well (A/B runs, tests)
  double* w = work + carve(H, work, f32);
  float* r32 = f32 ? (float*)H.rhs[0] : nullptr;
  double* r = w;
  // Search directions.  Fused loop: the iterate is NOT touched inside the loop (that cost 16 of the fused step's 36
  // bytes per node); the directions p_j stay in a ring of slots (10 fp32 / 5 fp64 vectors in these 5 nb doubles) with
  // their step lengths alpha_j, and x += sum_j alpha_j p_j is formed when the ring is full or the solve ends.
  // Unfused loop (small meshes / batches): one fp64 p in the same region, x updated every iteration.
  double* p = r + nb;
  const int n_slots = f32 ? kRingSlots : kRingSlots / 2;
  const long long slot_stride = nb;              // in elements of the stored type: fp32 slots are nb floats apart
  double* Ap = p + (kRingSlots / 2) * nb;
  double* partA = Ap + nb;
  double* partB = partA + (long long)kPartBlocks * Bp;
  double* sc = partB + (long long)kPartBlocks * Bp;
  PcgScalars S;
  S.rz = sc; S.alpha = sc + Bp; S.beta = sc + 2 * Bp; S.bb = sc + 3 * Bp; S.tol2 = sc + 4 * Bp;
  S.active = (int*)(sc + 5 * Bp);
  S.iters = iters;
  S.n_active = (int*)(sc + 6 * Bp);
  const bool use_floor = (precond_fp32 & 16) == 0;  // bit 4 set: stop on `tol` alone
  S.maxdiag = sc + 9 * Bp;  // Bv entries (Bv <= Bp)
  S.rs = f32 ? sc + 11 * Bp : nullptr;
  S.scale = scale;
  S.Bv = Bv;
  S.energy = sc + 12 * Bp;
  S.est = err_est ? err_est : sc + 13 * Bp;
  S.rr = sc + 14 * Bp;
  S.rule = stop_rule ? stop_rule : (int*)(sc + 7 * Bp);
  double* alpha_ring = sc + 16 * Bp;              // n_slots (<= 10) rows of Bp step lengths (the scalar block has 32 rows)
  double* const alpha_single = S.alpha;
  // tol_energy is asked of the FINAL iterate, which receives one more multigrid correction after the decision
  // (pcg_finish_kernel): the CG iterate's own estimate may be 1 / 0.3 of it (0.3: a cautious bound of the V(2,2)
  // cycle's convergence factor; measured reductions of the nodal error by that step: 5-8x)
  S.tol_e2 = tol_energy > 0.0 ? (tol_energy / 0.3) * (tol_energy / 0.3) : 0.0;
  S.e_max_it = 10 + ((tol_energy > 0.0 && tol_energy < 1e-11) ? (int)ceil(log10(1e-11 / tol_energy) - 1e-9) : 0);
  S.have_energy = 0;
  if (use_floor && use_fmg) {
    rc = diffhe::check(hipMemsetAsync((void*)S.maxdiag, 0, sizeof(double) * Bv, st));
    if (rc) return rc;
    hipLaunchKernelGGL(dia_maxdiag_kernel, node_grid(L0.n, Bv, 512), dim3(256), 0, st, L0, Bv,
                       (unsigned long long*)S.maxdiag);
  }
  {  // spectrum bound for the coarsest-level Chebyshev solve: 2 unless the mesh has obtuse triangles
    const Level& Lc = H.lev[H.nl - 1];
    if (Lc.nd == 4) {
      unsigned long long* gb = (unsigned long long*)(sc + 8 * Bp);
      rc = diffhe::check(hipMemsetAsync(gb, 0, sizeof(unsigned long long), st));
      if (rc) return rc;
      hipLaunchKernelGGL(dia_gershgorin_kernel, node_grid(Lc.n, Bv, 256), dim3(256), 0, st, Lc, Bv, gb);
      double bound = 0.0;
      rc = diffhe::check(hipMemcpyAsync(&bound, gb, sizeof(double), hipMemcpyDeviceToHost, st));
      if (!rc) rc = diffhe::check(hipStreamSynchronize(st));
      if (rc) return rc;
      if (bound > 2.0 && bound < 1e3) H.coarse_lmax = bound * (1.0 + 1e-9);
    }
  }
  const dim3 sgrid((Bp + 63) / 64);
  double* const slices = sc + 32LL * Bp;          // kScalarSlices rows: first stage of long partial lists
  static const int two_stage = getenv("DIFFHE_SCALAR2") ? atoi(getenv("DIFFHE_SCALAR2")) : 1;
  // several nodes per trip in the grid-stride vector kernels (A/B: DIFFHE_VEC_UNROLL=0)
  static const int vec_unroll = getenv("DIFFHE_VEC_UNROLL") ? atoi(getenv("DIFFHE_VEC_UNROLL")) : 1;
#define SCALAR(phase, part, nb_)                                                                                           \
  do {                                                                                                                     \
    if (two_stage && (int)(nb_) >= 256) {                                                                                  \
      hipLaunchKernelGGL(pcg_slice_kernel, dim3(sgrid.x, kScalarSlices), dim3(256), 0, st, (const double*)(part), (int)(nb_), \
                         Bp, slices);                                                                                      \
      hipLaunchKernelGGL(pcg_scalar_kernel, sgrid, dim3(1024), 0, st, (int)(phase), (const double*)slices, kScalarSlices, Bp, \
                         tol, S, relres);                                                                                  \
    } else {                                                                                                               \
      hipLaunchKernelGGL(pcg_scalar_kernel, sgrid, dim3(1024), 0, st, (int)(phase), (const double*)(part), (int)(nb_), Bp,   \
                         tol, S, relres);                                                                                  \
    }                                                                                                                      \
  } while (0)

  for (int l = 0; l < H.nl; ++l)
    if (H.lev[l].shift && (H.lev[l].inv || Bv != 1)) return DIFFHE_E_BADARG;  // a shift belongs to a factored operator
  if (H.nl == 1 && L0.inv && Bv == 1 && !f32 && L0.n <= kPartBlocks) {
    // DIRECT solve: the whole system is small enough for the dense inverse of its (batch-shared) matrix -- the
    // reference's own regime (2D meshes up to 32 x 32).  x = (1 / s_b) K_1^{-1} b in one launch, then the true residual.
    diffhe::account(16.0 * (double)n * Bp);
    if (Bp >= kWave)
      hipLaunchKernelGGL((mg_dense_solve_kernel<double, 4>), dim3((n + 3) / 4, Bp / kWave), dim3(256), 0, st, n,
                         (const double*)L0.inv, scale, b, x, (double*)nullptr, Bp);
    else
      hipLaunchKernelGGL(mg_dense_small_kernel<double>, dim3(n), dim3(64), 0, st, n, (const double*)L0.inv, scale, b, x,
                         (double*)nullptr, Bp);
    LAUNCH(8.0, pcg_init_kernel, n, b, (double*)nullptr, (double*)nullptr, partA, n, Bp);
    SCALAR(S_INIT, partA, nblk);                                 // b.b (and the bookkeeping S_RELRES reads)
    const int nbr = op_residual<double>(H, 0, b, (const double*)x, (double*)nullptr, partA, st);
    SCALAR(S_RELRES, partA, nbr);
    rc = diffhe::check(hipMemsetAsync(S.est, 0, sizeof(double) * Bp, st));
    if (rc) return rc;
    if (stop_rule) {  // direct solve: nothing iterated, nothing stopped
      rc = diffhe::check(hipMemsetAsync(stop_rule, 0, sizeof(int) * Bp, st));
      if (rc) return rc;
    }
    rc = diffhe::check_launch();
    if (rc) return rc;
    status_host[0] = 0;
    status_host[1] = 0;
    return DIFFHE_OK;
  }

  int nbz = 0, nba = 0;
  const bool light_init = (use_fmg && f32) || warm;  // the start overwrites x and r (cold) / x is the caller's guess (warm)
  LAUNCH(light_init ? 8.0 : 24.0, pcg_init_kernel, n, b, light_init ? (double*)nullptr : x, r, partA, n, Bp);
  SCALAR(S_INIT, partA, nblk);
  if (f32 && !warm) LAUNCH(12.0, pcg_cvt_kernel, n, b, (const double*)S.rs, r32, n, Bp);  // fp32 copy of rs * b (rs from S_INIT)
  // Fused loop (fine level runs the strip kernels): per iteration
  //   [p = z + beta p ; x += alpha_prev p_old ; Ap = A p ; p.Ap]  ->  alpha  ->  [r -= alpha Ap ; r.r]
  //   -> convergence flags  ->  z = V(r) (last sweep leaves r.z)  ->  beta
  // Unfused fallback (small meshes / batches): separate p-update, apply and x/r update kernels.
  // development knob: strip width / occupancy of the fused CG step (gpurun_out/r2l/variants.txt)
  static const int pupd_variant = getenv("DIFFHE_PUPD_VARIANT") ? atoi(getenv("DIFFHE_PUPD_VARIANT")) : 0;
  const StripGeom g0 = strip_geom(L0, Bp, kPupdCols);
  const bool fused = g0.use;
  // batch-shared matrix, fp32-stored directions: A p is never stored -- the residual update recomputes it from p (F_RUPD)
  const int rupd_mode = rupd_env();
  const bool rupd = fused && f32 && Bv == 1 && rupd_mode != 0;
  const StripGeom g8 = strip_geom(L0, Bp, 8);
  // cgstep2_kernel: two samples per lane for batches that are multiples of 128, else one
  // (four samples per lane -- what pays in the fused PRE pass -- measured here too: 0.699 -> 0.711 ms at 4 waves per SIMD
  // instead of 8, gpurun_out/r4n; not kept)
  const int cspl = (Bp % (2 * kWave) == 0) ? 2 : 1;
  const StripGeom g2 = (Bp % kWave == 0) ? strip_geom(L0, Bp, 4, cspl) : StripGeom{false, 0, 0, 0};
  const void* z = nullptr;
  int it = 0, flushed = 0;       // iterations done / directions already folded into x (fused loop)
  // x += sum_{j = flushed .. it-1} alpha_j p_j  (+ z / rs at the end of the solve: pcg_finish_kernel)
  auto flush_directions = [&](bool with_z) {
    const int count = it - flushed;
    if (count == 0 && !with_z) return;
    const double bytes = 16.0 + (f32 ? 4.0 : 8.0) * (count + (with_z ? 1 : 0));
    if (f32)
      LAUNCH(bytes, pcg_finish_kernel<float>, n, (const double*)alpha_ring, (const float*)(const void*)p, slot_stride,
             flushed, count, n_slots, with_z ? (const float*)z : (const float*)nullptr, (const double*)S.rs, x, n, Bp, vec_unroll);
    else
      LAUNCH(bytes, pcg_finish_kernel<double>, n, (const double*)alpha_ring, (const double*)p, slot_stride, flushed, count,
             n_slots, with_z ? (const double*)z : (const double*)nullptr, (const double*)nullptr, x, n, Bp, vec_unroll);
    flushed = it;
  };
  auto precondition = [&](int first) {
    if (f32) z = vcycle<float>(H, (const float*)r32, partB, &nbz, st);
    else z = vcycle<double>(H, (const double*)r, partB, &nbz, st);
    SCALAR(first ? S_RZ0 : S_BETA, partB, nbz);
  };
  auto apply_step = [&](int first) {
    if (fused) {
      if (!first) kp_begin(KP_CGSTEP, st);  // the first step of a solve (p = z) moves fewer bytes: not timed
      if (it - flushed == n_slots) flush_directions(false);   // ring full: fold everything so far into x
      const size_t esz = f32 ? sizeof(float) : sizeof(double);
      char* ring = (char*)p;
      Extra ex{};
      ex.a0 = z;
      ex.p_in = ring + (size_t)((it + n_slots - 1) % n_slots) * slot_stride * esz;
      ex.p_out = ring + (size_t)(it % n_slots) * slot_stride * esz;
      ex.x = nullptr;            // deferred (flush_directions)
      ex.alpha = nullptr; ex.beta = S.beta; ex.first = first;
#define NXV(RW_, MINW_)                                                                                               \
  launch_strip<double, M_APPLY, false, F_PUPD_NX, float, RW_, MINW_>(L0, Bv, scale, (const double*)nullptr,                \
                                                                      (const double*)nullptr, rupd ? (double*)nullptr : Ap, \
                                                                      0.0, 0.0, partA, Bp, g0, st, ex)
      static const int cg2 = getenv("DIFFHE_CG2") ? atoi(getenv("DIFFHE_CG2")) : 1;
      // flag bit 8: the caller vouches for a lattice closed by Dirichlet data (lambda_min of the scaled operator bounded
      // away from 0).  With large Neumann parts the search directions are dominated by near-null modes, for which the
      // fp32 stencil cancels to noise: measured 13 / 11 instead of 12 / 9 iterations to 1e-14 there (gpurun_out/r6g)
      // (the host also asks for near-square cells and a hierarchy that reaches the dense coarsest level: on a 382 x 259
      // lattice, which coarsens once, 36 iterations to 1e-14 became 38 -- tools/stress.py seed 6301 case 39)
      if (f32 && rupd && cg2 && (precond_fp32 & 256) && g2.use && shared32_ok(L0, Bv, Bp) &&
          strip2_tile_fits(L0, Bp, g2.TR + 3)) {
        // fp32 stencil for p.Ap (cgstep2_kernel; packed, two samples per lane, where the batch allows): the step length only
        const dim3 grid(g2.ncb * g2.nrc, Bp / (cspl * kWave));
        diffhe::account((first ? 8.0 : 12.0) * (double)n * Bp);
        const float* zz = (const float*)z;
        const float* pi_ = (const float*)ex.p_in;
        float* po_ = (float*)ex.p_out;
#define CG2(VT_, ND_, MW_) hipLaunchKernelGGL((cgstep2_kernel<VT_, ND_, 4, MW_>), grid, dim3(256), 0, st, L0, scale, \
                                              (const double*)S.beta, first, zz, pi_, po_, partA, Bp, g2.ncb, g2.TR)
        // 8 waves per SIMD: 0.70 ms at 1024^2 x 256 (6: 0.75, 4: 0.75; the one-sample fp64 strip: 0.87; gpurun_out/r6e)
        if (cspl == 2) { if (L0.nd == 3) CG2(v2f, 3, 8); else CG2(v2f, 4, 8); }
        else { if (L0.nd == 3) CG2(float, 3, 8); else CG2(float, 4, 8); }
#undef CG2
        S.alpha = alpha_ring + (long long)(it % n_slots) * Bp;
        nba = g2.ncb * g2.nrc;
        if (!first) kp_end(KP_CGSTEP, st);
        return;
      }
      if (f32) {
        // 72 VGPRs (18 spilled), 7 waves per SIMD: 1.16 ms against 1.28 at the compiler's own 85 / 5; 8-column strips
        // (142 VGPRs) 1.96, 2-column strips at 8 waves 1.27, 6 or 8 waves 1.25 / 1.18 (same box, gpurun_out/r2l/variants*.txt).
        // The same cap on the V-cycle's strip kernels (already 6-7 waves) made them slower: -2...-6 % end to end.
        // per-sample matrices (coefficients in VGPRs): 4 waves per SIMD, 245.2 ms per step of the per-element-field variant
        // against 249.5 at 7 (gpurun_out/r4w)
        static const int pupd_ps = getenv("DIFFHE_PUPD_PS") ? atoi(getenv("DIFFHE_PUPD_PS")) : 4;
        const int pv = Bv != 1 ? pupd_ps : pupd_variant;
        if (pv == 5 || pv == 1) NXV(4, 1); else if (pv == 4) NXV(4, 4); else if (pv == 6) NXV(4, 6); else NXV(4, 7);
      }
#undef NXV
      else
        launch_strip<double, M_APPLY, false, F_PUPD_NX, double, kPupdCols>(L0, Bv, scale, (const double*)nullptr,
                                                                (const double*)nullptr, Ap, 0.0, 0.0, partA, Bp, g0, st, ex);
      S.alpha = alpha_ring + (long long)(it % n_slots) * Bp;   // alpha_it goes next to p_it
      nba = g0.ncb * g0.nrc;
      if (!first) kp_end(KP_CGSTEP, st);
    } else {
      if (f32) LAUNCH(first ? 12.0 : 20.0, pcg_update_p_kernel<float>, n, (const float*)z, (const double*)S.beta, p, first, n, Bp);
      else LAUNCH(first ? 16.0 : 24.0, pcg_update_p_kernel<double>, n, (const double*)z, (const double*)S.beta, p, first, n, Bp);
      nba = op_apply_dot(H, p, Ap, partA, st);
    }
  };
  // r = b - A x (+ its fp32 copy, + the partials b.x and x.(A x) of the energy bound when asked for)
  auto residual_pass = [&](bool energy) {
    const StripGeom gr = strip_geom(L0, Bp);
    if (gr.use && f32) {  // r and its fp32 copy in one pass
      Extra ex{};
      ex.r32 = r32;
      ex.rscale = S.rs;
      ex.dot_bx = energy ? 1 : 0;   // partial sums of this pass: b.x0 and x0.(A x0) (S_ENERGY / S_ENERGY2)
      ex.part2 = energy ? partB : nullptr;
      launch_strip<double, M_RESID, false>(L0, Bv, scale, (const double*)x, b, r, 0.0, 0.0, energy ? partA : (double*)nullptr,
                                           Bp, gr, st, ex);
      nba = gr.ncb * gr.nrc;
    } else {
      nba = op_residual<double>(H, 0, b, (const double*)x, r, energy ? partA : (double*)nullptr, st, energy ? 1 : 0,
                                energy ? partB : (double*)nullptr);
      if (f32) LAUNCH(12.0, pcg_cvt_kernel, n, (const double*)r, (const double*)S.rs, r32, n, Bp);
    }
    if (energy) {
      SCALAR(S_ENERGY, partA, nba);
      SCALAR(S_ENERGY2, partB, nba);
      S.have_energy = 1;
    }
  };
  if (use_fmg) {
    // cold: x0 = FMG(b).  warm: x0 = x + FMG(b - A x) -- the full-multigrid start applied to the residual equation of
    // the caller's guess (an optimisation loop's previous solution): the start is then as accurate as the guess is
    // close, times the ~1e-3 of the full-multigrid step itself.
    if (warm) residual_pass(false);
    if (f32) {
      const float* e0 = nullptr;
      const float* x0 = fmg_start<float>(H, (const float*)r32, st, &e0);
      if (!x0) return DIFFHE_E_LAUNCH;
      LAUNCH((warm ? 20.0 : 12.0) + (e0 ? 4.0 : 0.0), pcg_setx_kernel<float>, n, x0, (const double*)S.rs, x,
             use_floor ? partA : (double*)nullptr, n, Bp, warm ? 1 : 0, e0, vec_unroll);
    } else {
      const double* e0 = nullptr;
      const double* x0 = fmg_start<double>(H, warm ? (const double*)r : b, st, &e0);
      if (!x0) return DIFFHE_E_LAUNCH;
      LAUNCH((warm ? 24.0 : 16.0) + (e0 ? 8.0 : 0.0), pcg_setx_kernel<double>, n, x0, (const double*)nullptr, x,
             use_floor ? partA : (double*)nullptr, n, Bp, warm ? 1 : 0, e0, vec_unroll);
    }
    if (use_floor) SCALAR(S_FLOOR, partA, nblk);
    residual_pass(true);
  } else if (warm) {
    residual_pass(true);
  }
  precondition(1);
  rc = diffhe::check_launch();
  if (rc) return rc;

  int n_active = -1;
  while (it < max_iter) {
    apply_step(it == 0);
    SCALAR(S_ALPHA, partA, nba);
    kp_begin(KP_UPDATE, st);
    if (rupd) {
      Extra ex{};
      ex.p_in = (char*)p + (size_t)(it % n_slots) * slot_stride * sizeof(float);   // the direction apply_step just stored
      ex.x = r;
      ex.r32 = r32;
      ex.rscale = S.rs;
      ex.alpha = S.alpha;
#define RUV(RW_, MINW_, G_)                                                                                            \
  launch_strip<double, M_APPLY, false, F_RUPD, float, RW_, MINW_>(L0, Bv, scale, (const double*)nullptr,                   \
                                                                   (const double*)nullptr, (double*)nullptr, 0.0, 0.0,   \
                                                                   partA, Bp, G_, st, ex)
      // 5 waves per SIMD: 1.30 ms at 1024^2 x 256 (compiler's own choice 1.30, 7 waves 2.31 with spills; gpurun_out/r4q)
      if (rupd_mode == 2) RUV(kPupdCols, 1, g0); else if (rupd_mode == 4) RUV(8, 1, g8); else RUV(kPupdCols, 5, g0);
#undef RUV
    } else {
      LAUNCH(24.0 + (r32 ? 4.0 : 0.0) + (fused ? 0.0 : 24.0), pcg_update_kernel, n, (const double*)p, (const double*)Ap, (const double*)S.alpha, fused ? (double*)nullptr : x,
             r, r32, (const double*)S.rs, partA, n, Bp, vec_unroll);
    }
    kp_end(KP_UPDATE, st);
    SCALAR(S_CONV, partA, rupd ? (rupd_mode == 4 ? g8.ncb * g8.nrc : g0.ncb * g0.nrc) : nblk);
    ++it;
    // z = V(r) and r.z: the new search direction's ingredients AND the energy-norm error estimate of the iterate;
    // the samples still active are counted in the scalar phase behind it (S_BETA)
    precondition(0);
    rc = diffhe::check(hipMemcpyAsync(&status_host[2], S.n_active, sizeof(int), hipMemcpyDeviceToHost, st));
    if (rc) return rc;
    rc = diffhe::check(hipStreamSynchronize(st));
    if (rc) return rc;
    n_active = status_host[2];
    kp_collect();  // the stream is idle here
    if (n_active == 0) break;
  }
  // fold the directions still in the ring into x and add the final V-cycle's correction z (pcg_finish_kernel);
  // the unfused loop kept x current: only z is due there
  if (!fused) flushed = it;
  flush_directions(true);
  S.alpha = alpha_single;
  nba = op_residual<double>(H, 0, b, (const double*)x, (double*)nullptr, partA, st);
  SCALAR(S_RELRES, partA, nba);
  rc = diffhe::check_launch();
  if (rc) return rc;
  status_host[0] = it;
  status_host[1] = n_active < 0 ? 0 : n_active;
  return DIFFHE_OK;
}

extern "C" int diffhe_lattice_recompute_ap(void) { return rupd_env() != 0; }

extern "C" int diffhe_lattice_blocks(int n, int Bp) { (void)n; (void)Bp; return kPartBlocks; }

extern "C" int diffhe_lattice_fused_passes(void) { return fused_mode(); }

extern "C" int diffhe_lattice_apply(const diffhe_mg_level* level, int Bv, const double* scale, const double* x,
                                    double* y, double* part, int Bp, void* stream) {
  if (!x || !y || !part) return DIFFHE_E_BADARG;
  Hier H;
  const double w1 = 0.8;
  int rc = fill_hier(H, level, 1, Bv, Bp, scale, &w1, 1, 1);
  if (rc) return rc;
  op_apply_dot(H, x, y, part, (hipStream_t)stream);
  return diffhe::check_launch();
}

extern "C" int diffhe_lattice_bilinear(const diffhe_mg_level* level, int Bv, const double* scale, const double* x,
                                       const double* lam, const double* add, double* part, double* out, int Bp,
                                       void* stream) {
  if (!x || !lam || !part || !out) return DIFFHE_E_BADARG;
  Hier H;
  const double w1 = 0.8;
  int rc = fill_hier(H, level, 1, Bv, Bp, scale, &w1, 1, 1);
  if (rc) return rc;
  const StripGeom g = strip_geom(H.lev[0], Bp);
  if (!g.use) return DIFFHE_E_TOOBIG;  // small problems: use diffhe_p1_grad_kappa
  hipStream_t st = (hipStream_t)stream;
  Extra ex{};
  ex.dotv = lam;
  ex.addv = add;
  launch_strip<double, M_APPLY, false>(H.lev[0], Bv, scale, x, (const double*)nullptr, (double*)nullptr, 0.0, 0.0, part,
                                       Bp, g, st, ex);
  PcgScalars S{};
  hipLaunchKernelGGL(pcg_scalar_kernel, dim3((Bp + 63) / 64), dim3(1024), 0, st, (int)S_SUM, (const double*)part,
                     g.ncb * g.nrc, Bp, 0.0, S, out);
  return diffhe::check_launch();
}

extern "C" int diffhe_lattice_cg_step(const diffhe_mg_level* level, int Bv, const double* scale, const void* z,
                                      int z_fp32, const void* p_in, void* p_out, double* x, const double* alpha,
                                      const double* beta, int first, double* Ap, double* part, int Bp, void* stream) {
  if (!z || !p_out || !Ap || !part || (!first && (!p_in || !beta || (x && !alpha)))) return DIFFHE_E_BADARG;
  Hier H;
  const double w1 = 0.8;
  int rc = fill_hier(H, level, 1, Bv, Bp, scale, &w1, 1, 1);
  if (rc) return rc;
  const StripGeom g = strip_geom(H.lev[0], Bp, kPupdCols);
  if (!g.use) return DIFFHE_E_TOOBIG;
  Extra ex{};
  ex.a0 = z; ex.p_in = p_in; ex.p_out = p_out; ex.x = x; ex.alpha = alpha; ex.beta = beta; ex.first = first;
  hipStream_t st = (hipStream_t)stream;
#define CGSTEP(FUSE_, TA_)                                                                                          \
  launch_strip<double, M_APPLY, false, FUSE_, TA_, kPupdCols>(H.lev[0], Bv, scale, (const double*)nullptr,              \
                                                              (const double*)nullptr, Ap, 0.0, 0.0, part, Bp, g, st, ex)
  if (x) {
    if (z_fp32) CGSTEP(F_PUPD, float); else CGSTEP(F_PUPD, double);
  } else {
    if (z_fp32)   // the solver's instantiation (7 waves per SIMD)
      launch_strip<double, M_APPLY, false, F_PUPD_NX, float, kPupdCols, 7>(H.lev[0], Bv, scale, (const double*)nullptr,
                                                                          (const double*)nullptr, Ap, 0.0, 0.0, part, Bp, g, st, ex);
    else
      CGSTEP(F_PUPD_NX, double);
  }
#undef CGSTEP
  return diffhe::check_launch();
}

// y = is_bc ? 0 : (M x - sub_scale[b] * sub) for a batch-shared symmetric-diagonal matrix M (the load
// matrix of a lattice mesh): F = M f - lift and df = M^T lambda without the general ELL pattern.
__global__ __launch_bounds__(256) void dia_shared_apply_kernel(Level L, const double* __restrict__ x,
                                                                const double* __restrict__ sub, int sub_B,
                                                                const double* __restrict__ sub_scale,
                                                                const unsigned char* __restrict__ mask,
                                                                double* __restrict__ y, int Bp) {
  const NodeMap nm = node_map(Bp);
  for (int i = nm.node0; i < L.n; i += nm.stride) {
    double acc = dia_row(L, 1, 0, x, i, nm.b, Bp);
    if (sub) acc -= (sub_scale ? sub_scale[nm.b] : 1.0) * sub[(i64)i * sub_B + (sub_B == 1 ? 0 : nm.b)];
    if (mask && mask[i]) acc = 0.0;
    y[(i64)i * Bp + nm.b] = acc;
  }
}

extern "C" int diffhe_lattice_apply_shared(int nx, int ny, int nd, const double* vals, const double* x,
                                           const double* sub, int sub_B, const double* sub_scale,
                                           const unsigned char* mask, double* y, int Bp, void* stream) {
  if (!vals || !x || !y || nx < 2 || ny < 2 || (nd != 3 && nd != 4)) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  if (sub && sub_B != 1 && sub_B != Bp) return DIFFHE_E_BADARG;
  Level L{};   // inv, shift: none
  L.nx = nx; L.ny = ny; L.W = nx + 1; L.n = (nx + 1) * (ny + 1); L.nd = nd; L.v = vals; L.v32 = nullptr; L.bc = nullptr;
  L.rd32 = nullptr; L.mk32 = nullptr; L.o16 = nullptr; L.osc = nullptr;
  const StripGeom g = strip_geom(L, Bp);
  if (g.use) {
    Extra ex{};
    ex.sub = sub; ex.sub_scale = sub_scale; ex.mask = mask;
    ex.sub_pb = (sub && sub_B != 1) ? 1 : 0;   // per-sample lift: read in the strip pass too (it used to fall to the
                                               // gather kernel below: 2.67 instead of ~1.4 ms at 1024^2 x 256)
    if (ex.sub_pb) diffhe::account(8.0 * (double)L.n * Bp);
    launch_strip<double, M_APPLY, false>(L, 1, nullptr, x, (const double*)nullptr, y, 0.0, 0.0, nullptr, Bp, g,
                                         (hipStream_t)stream, ex);
    return diffhe::check_launch();
  }
  diffhe::account((16.0 + (sub && sub_B != 1 ? 8.0 : 0.0)) * (double)L.n * Bp);
  hipLaunchKernelGGL(dia_shared_apply_kernel, lgrid(L.n, Bp), dim3(256), 0, (hipStream_t)stream, L, x, sub, sub_B,
                     sub_scale, mask, y, Bp);
  return diffhe::check_launch();
}

extern "C" int diffhe_lattice_smooth(const diffhe_mg_level* level, int Bv, const double* scale, const double* rhs,
                                     const double* xin, double* xout, double omega, int Bp, void* stream) {
  if (!rhs || !xout) return DIFFHE_E_BADARG;
  Hier H;
  int rc = fill_hier(H, level, 1, Bv, Bp, scale, &omega, 1, 1);
  if (rc) return rc;
  op_jacobi<double>(H, 0, rhs, xin, xout, omega, nullptr, (hipStream_t)stream);
  return diffhe::check_launch();
}

// dL/dkappa per element and sample on a lattice mesh (reverse of solver.py:137-140; Appendix A step 2):
//   dk[e, b] = - sum_{p,q} lambda[node_p, b] k0[p*3+q, e] (u[node_q, b] + g[node_q])
// Quad (r, c) = nodes a (r, c), b (r, c+1), c (r+1, c+1), d (r+1, c) carries T0 = [a, b, d] = element 2q and
// T1 = [b, c, d] = element 2q + 1 (mesh.py:100-105).  A wave owns GW quad columns x 64 samples and marches down the quad
// rows with a two-row window of lambda and u in registers: every nodal value is loaded once per wave (+ one halo
// column) instead of once per incident element (6x), k0 arrives as scalar loads, dk leaves as 512 B rows.
// 32 B per node and sample of algorithmic traffic (lambda, u, two dk): the element-loop kernel ran it at 1.2 TB/s.
constexpr int kGradCols = 4;
__global__ __launch_bounds__(256) void lattice_grad_kappa_kernel(int nx, int ny, const double* __restrict__ k0, i64 lm,
                                                                  i64 emask, const double* __restrict__ lam,
                                                                  const double* __restrict__ u,
                                                                  const double* __restrict__ g, double* __restrict__ dk,
                                                                  int Bp, int ncb, int TR) {
  constexpr int GW = kGradCols;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lb = blockIdx.y * kWave + lane;
  const int tile = xcd_tile(blockIdx.x, gridDim.x);
  const int rc = tile / ncb, cb = tile - rc * ncb;
  const int c0 = (cb * 4 + wave) * GW;                 // first quad column
  const int r0 = rc * TR;
  const int r1 = (r0 + TR < ny) ? r0 + TR : ny;        // quad rows r0 .. r1 - 1
  if (c0 >= nx || r0 >= r1) return;
  const int W = nx + 1;
  int dc[GW + 1];                                       // node columns c0 .. c0 + GW, clamped at the right edge
#pragma unroll
  for (int j = 0; j < GW + 1; ++j) dc[j] = (c0 + j < W) ? j : W - 1 - c0;
  double la[GW + 1], ua[GW + 1], lbn[GW + 1], ubn[GW + 1];   // node rows r (a, b) and r + 1 (d, c)
  const double* __restrict__ pl = lam + ((i64)r0 * W + c0) * Bp;
  const double* __restrict__ pu = u + ((i64)r0 * W + c0) * Bp;
  const double* __restrict__ pg = g ? g + (i64)r0 * W + c0 : nullptr;
  const i64 rowX = (i64)W * Bp;
#pragma unroll
  for (int j = 0; j < GW + 1; ++j) {
    la[j] = (pl + (i64)dc[j] * Bp)[lb];
    ua[j] = (pu + (i64)dc[j] * Bp)[lb] + (pg ? pg[dc[j]] : 0.0);
  }
  for (int r = r0; r < r1; ++r) {
#pragma unroll
    for (int j = 0; j < GW + 1; ++j) {
      lbn[j] = (pl + rowX + (i64)dc[j] * Bp)[lb];
      ubn[j] = (pu + rowX + (i64)dc[j] * Bp)[lb] + (pg ? pg[W + dc[j]] : 0.0);
    }
    const i64 e0 = 2 * ((i64)r * nx + c0);               // element 2 q of quad (r, c0)
#pragma unroll
    for (int j = 0; j < GW; ++j) {
      if (c0 + j >= nx) continue;
      const i64 e = e0 + 2 * j;
      // T0 = [a, b, d]: a = (r, c), b = (r, c + 1), d = (r + 1, c)
      {
        const double lp[3] = {la[j], la[j + 1], lbn[j]}, uq[3] = {ua[j], ua[j + 1], ubn[j]};
        double acc = 0.0;
#pragma unroll
        for (int p_ = 0; p_ < 3; ++p_)
#pragma unroll
          for (int q = 0; q < 3; ++q) acc += lp[p_] * k0[(i64)(p_ * 3 + q) * lm + (e & emask)] * uq[q];
        (dk + e * Bp)[lb] = -acc;
      }
      // T1 = [b, c, d]: b = (r, c + 1), c = (r + 1, c + 1), d = (r + 1, c)
      {
        const double lp[3] = {la[j + 1], lbn[j + 1], lbn[j]}, uq[3] = {ua[j + 1], ubn[j + 1], ubn[j]};
        double acc = 0.0;
#pragma unroll
        for (int p_ = 0; p_ < 3; ++p_)
#pragma unroll
          for (int q = 0; q < 3; ++q) acc += lp[p_] * k0[(i64)(p_ * 3 + q) * lm + ((e + 1) & emask)] * uq[q];
        (dk + (e + 1) * Bp)[lb] = -acc;
      }
    }
#pragma unroll
    for (int j = 0; j < GW + 1; ++j) {
      la[j] = lbn[j];
      ua[j] = ubn[j];
    }
    pl += rowX;
    pu += rowX;
    if (pg) pg += W;
  }
}

extern "C" int diffhe_lattice_grad_kappa(int nx, int ny, const double* k0, int k0_compact, const double* lam,
                                         const double* u, const double* g, double* dk, int Bp, void* stream) {
  if (!k0 || !lam || !u || !dk || nx < 2 || ny < 2) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  if (Bp < kWave) return DIFFHE_E_TOOBIG;              // small batches: diffhe_p1_grad_kappa
  const int ncb = (nx + 4 * kGradCols - 1) / (4 * kGradCols);
  const int gy = Bp / kWave;
  int nrc = (6144 + ncb * gy - 1) / (ncb * gy);
  if (nrc > ny / 8) nrc = ny / 8;
  if (nrc < 1) nrc = 1;
  const int TR = (ny + nrc - 1) / nrc;
  nrc = (ny + TR - 1) / TR;
  diffhe::account(32.0 * (double)(nx + 1) * (ny + 1) * Bp);   // lambda, u once per node; two dk per node
  const i64 mm = 2LL * nx * ny;
  hipLaunchKernelGGL(lattice_grad_kappa_kernel, dim3(ncb * nrc, gy), dim3(256), 0, (hipStream_t)stream, nx, ny, k0,
                     (i64)(k0_compact ? 2 : mm), (i64)(k0_compact ? 1 : -1), lam, u, g, dk, Bp, ncb, TR);
  return diffhe::check_launch();
}

// fp32 diagonal + fp16 off-diagonals of a per-sample symmetric-diagonal matrix (h16m above): the off-diagonals of sample b
// divided by `oscale[b]` and rounded to fp16, the diagonal moved by the sum of the rounding differences of the row's
// 2 (nd - 1) couplings so that the row sum is the fp64 matrix's (to the fp32 rounding of the diagonal itself, 6e-8
// relative).  flags[0] is set when a non-zero coupling falls below 2^-19 of its sample's scale: fp16 subnormals keep fewer
// than 5 bits there (and flush to 0 from 2^-25 on: the row-sum rule would then leave a row with a vanishing diagonal), so
// the caller must not use the packed copies of that matrix (high contrast INSIDE a sample; the fp32 copies have no such limit).
__global__ __launch_bounds__(256) void dia_pack_h16_kernel(Level L, int Bv, const double* __restrict__ oscale,
                                                            float* __restrict__ d32, _Float16* __restrict__ o16,
                                                            int* __restrict__ flags) {
  const NodeMap nm = node_map(Bv);
  if (nm.b >= Bv) return;
  const i64 n = L.n;
  const double osc = oscale[nm.b];
  const double inv = 1.0 / osc;   // power of two: exact
  const double tiny = 1.9073486328125e-06;   // 2^-19
  bool under = false;
  for (int i = nm.node0; i < L.n; i += nm.stride) {
    double d = L.v[(i64)i * Bv + nm.b];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      if (k < L.nd) {
        const int off = dia_off(L, k);
        const double up = L.v[((i64)k * n + i) * Bv + nm.b];            // coupling (i, i + off): stored here
        const _Float16 h = (_Float16)(float)(up * inv);
        o16[((i64)(k - 1) * n + i) * Bv + nm.b] = h;
        if (i + off < L.n) {
          d += up - osc * (double)(float)h;
          under = under || (up != 0.0 && fabs(up * inv) < tiny);
        }
        if (i - off >= 0) {                                             // coupling (i - off, i): stored at the other end
          const double lo = L.v[((i64)k * n + (i - off)) * Bv + nm.b];
          d += lo - osc * (double)(float)(_Float16)(float)(lo * inv);
        }
      }
    }
    d32[(i64)i * Bv + nm.b] = (float)d;
  }
  if (flags && __any(under) && (threadIdx.x & 63) == 0) atomicOr(flags, 1);
}

extern "C" int diffhe_lattice_pack_h16(const diffhe_mg_level* level, int Bv, const double* offdiag_scales, float* diag32,
                                       void* offdiag16, int* flags, void* stream) {
  if (!level || !diag32 || !offdiag16 || !level->vals || (level->nd != 3 && level->nd != 4) || !offdiag_scales)
    return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bv)) return DIFFHE_E_BATCHPAD;
  Level L{};
  L.nx = level->nx; L.ny = level->ny; L.W = level->nx + 1; L.n = (level->nx + 1) * (level->ny + 1); L.nd = level->nd;
  L.v = level->vals;
  diffhe::account((8.0 * L.nd + 4.0 + 2.0 * (L.nd - 1)) * (double)L.n * Bv);
  hipLaunchKernelGGL(dia_pack_h16_kernel, node_grid(L.n, Bv), dim3(256), 0, (hipStream_t)stream, L, Bv, offdiag_scales,
                     diag32, (_Float16*)offdiag16, flags);
  return diffhe::check_launch();
}

// Per-sample maximum of the main diagonal over the FREE rows (identity rows of Dirichlet nodes carry 1.0 whatever the
// magnitude of kappa and are skipped): the quantity the per-sample fp16 scale is derived from.  out: Bv doubles.
__global__ __launch_bounds__(256) void dia_maxdiag_free_kernel(Level L, int Bv, unsigned long long* __restrict__ out) {
  const NodeMap nm = node_map(Bv);
  double m = 0.0;
  if (nm.b < Bv)
    for (int i = nm.node0; i < L.n; i += nm.stride) {
      const double d = L.bc[i] ? 0.0 : L.v[(i64)i * Bv + nm.b];
      m = d > m ? d : m;
    }
  const int LB = Bv < kWave ? Bv : kWave;
  for (int off = LB; off < kWave; off <<= 1) {  // lanes that hold the same sample
    const double o = __shfl_xor(m, off);
    m = o > m ? o : m;
  }
  if ((int)(threadIdx.x & 63) < LB && nm.b < Bv) atomicMax(out + nm.b, (unsigned long long)__double_as_longlong(m));
}

extern "C" int diffhe_lattice_max_diag(const diffhe_mg_level* level, int Bv, double* out, void* stream) {
  if (!level || !out || !level->vals || !level->is_bc || (level->nd != 3 && level->nd != 4)) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bv)) return DIFFHE_E_BATCHPAD;
  Level L{};
  L.nx = level->nx; L.ny = level->ny; L.W = level->nx + 1; L.n = (level->nx + 1) * (level->ny + 1); L.nd = level->nd;
  L.v = level->vals; L.bc = level->is_bc;
  int rc = diffhe::check(hipMemsetAsync(out, 0, sizeof(double) * Bv, (hipStream_t)stream));
  if (rc) return rc;
  diffhe::account(8.0 * (double)L.n * Bv);
  hipLaunchKernelGGL(dia_maxdiag_free_kernel, node_grid(L.n, Bv, 512), dim3(256), 0, (hipStream_t)stream, L, Bv,
                     (unsigned long long*)out);
  return diffhe::check_launch();
}

extern "C" int diffhe_lattice_restrict_kappa(const double* kappa_fine, double* kappa_coarse, int nx_coarse,
                                             int ny_coarse, int sx, int sy, int Bv, void* stream) {
  if (!kappa_fine || !kappa_coarse || nx_coarse < 1 || ny_coarse < 1) return DIFFHE_E_BADARG;
  if ((sx != 1 && sx != 2) || (sy != 1 && sy != 2) || (sx == 1 && sy == 1)) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bv)) return DIFFHE_E_BATCHPAD;
  diffhe::account(8.0 * Bv * (2.0 * nx_coarse * ny_coarse) * (1.0 + sx * sy));
  hipLaunchKernelGGL(mg_restrict_kappa_kernel, node_grid(2 * nx_coarse * ny_coarse, Bv), dim3(256), 0,
                     (hipStream_t)stream, kappa_fine, kappa_coarse, nx_coarse, ny_coarse, sx, sy, Bv);
  return diffhe::check_launch();
}

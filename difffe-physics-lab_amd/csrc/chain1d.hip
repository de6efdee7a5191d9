// 1D chain path: fused P1 assembly + Dirichlet elimination + exact solve (+ adjoint)
// for elements[e] = (e, e+1).  Replaces reference diffhe/solver.py:73-98 and :153-183.
//
// Algorithm.  On a chain the P1 stiffness is a weighted path-graph Laplacian.  With the
// element flux q_e = k_e (u_{e+1} - u_e), k_e = kappa_e / h_e, row i of K u = F reads
// q_{i-1} - q_i = F_i, so inside a Dirichlet-delimited segment [a, b]
//     q_e     = C - S_e,              S_e = sum_{i=a..e} F_i            (flux scan)
//     u_{j}   = u_a + C R_j - T_j,    R_j = sum_{e<j} 1/k_e,  T_j = sum_{e<j} S_e / k_e
// and C, u_a follow from the two segment ends.  (F,R,T) compose associatively:
//     (F1,R1,T1) o (F2,R2,T2) = (F1+F2, R1+R2, T1+T2+F1*R2)
// so the solve is ONE block scan per (segment, sample) plus two thread-local sweeps over
// an LDS-staged copy of the segment.  HBM traffic: read f, write u (16 n B); the adjoint
// reads gbar and u and writes df (+ dkappa): 24-32 n B.  K is never materialised.
//
// One workgroup per (segment, sample).  Thread t owns the odd-length chunk
// [t*c, (t+1)*c) of the staged arrays: odd c makes the stride-c LDS reads conflict-free.
#include "common.h"

namespace {

using namespace diffhe;

struct Trip {
  double F, R, T;
};

__device__ inline Trip comb(const Trip& a, const Trip& b) {
  return {a.F + b.F, a.R + b.R, a.T + b.T + a.F * b.R};
}

__device__ inline Trip shfl_up_trip(const Trip& v, int d) {
  return {__shfl_up(v.F, d), __shfl_up(v.R, d), __shfl_up(v.T, d)};
}

struct ChainArgs {
  const double* x;
  const double* kappa;
  long long ksb, kse;
  const double* rhs;
  long long rhs_sb;
  const int* seg;
  int n_seg;
  const double* g;       // Dirichlet values (NULL == all zero: adjoint)
  double* out;           // u (forward) / df (adjoint), row stride ldo
  long long ldo;
  const double* u;       // adjoint only: forward solution, row stride ldu
  long long ldu;
  double* dk_e;          // adjoint only, optional
  long long lddk;
  double* dk_part;       // adjoint only
  int n, B;
  double* stage;         // global staging (only when !USE_LDS)
  long long stage_len;   // doubles per (sample, segment) half-buffer
};

__device__ inline double lumped_weight(const double* x, int i, int n) {
  const double xi = x[i];
  const double hl = i > 0 ? xi - x[i - 1] : 0.0;
  const double hr = i < n - 1 ? x[i + 1] - xi : 0.0;
  return 0.5 * (hl + hr);
}

template <int NT, bool ADJ, bool USE_LDS>
__global__ __launch_bounds__(NT) void chain_kernel(ChainArgs A) {
  extern __shared__ double dyn[];
  __shared__ Trip wtot[NT / 64];
  __shared__ double red[NT / 64];
  __shared__ double bc_vals[3];  // u_a, C, load at the right end

  const int s = blockIdx.x, b = blockIdx.y;
  const int a = A.seg[3 * s + 0];
  const int bn = A.seg[3 * s + 1];
  const int flags = A.seg[3 * s + 2];
  const bool left_d = flags & 1, right_d = flags & 2;
  const int L = bn - a;  // elements in the segment
  if (L <= 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

  double* Fs;
  double* Rs;
  if (USE_LDS) {
    Fs = dyn;
    Rs = dyn + L;
  } else {
    Fs = A.stage + ((long long)b * A.n_seg + s) * 2 * A.stage_len;
    Rs = Fs + A.stage_len;
  }
  const double* rhs = A.rhs + (long long)b * A.rhs_sb;
  const double* kap = A.kappa + (long long)b * A.ksb;

  // ---- stage: element integrals + load, coalesced --------------------------------
  for (int q = t; q < L; q += NT) {
    const int e = a + q;
    const double he = A.x[e + 1] - A.x[e];                 // solver.py:84-86
    const double w = ADJ ? 1.0 : lumped_weight(A.x, e, A.n);  // solver.py:95-96
    Fs[q] = rhs[e] * w;
    Rs[q] = he / kap[(long long)e * A.kse];                // 1 / k_e, solver.py:88
  }
  if (t == 0) bc_vals[2] = rhs[bn] * (ADJ ? 1.0 : lumped_weight(A.x, bn, A.n));
  __syncthreads();

  // ---- sweep 1: chunk composites ----------------------------------------------------
  int c = (L + NT - 1) / NT;
  c |= 1;
  const int lo = t * c;
  const int hi = lo + c < L ? lo + c : L;
  Trip acc = {0.0, 0.0, 0.0};
  for (int q = lo; q < hi; ++q) {
    const double Fq = (q == 0 && left_d) ? 0.0 : Fs[q];
    const double r = Rs[q];
    acc.F += Fq;
    acc.R += r;
    acc.T += acc.F * r;
  }

  // ---- block scan with the composite operator ---------------------------------------
  Trip inc = acc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    Trip o = shfl_up_trip(inc, d);
    if (lane >= d) inc = comb(o, inc);
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  Trip pre = {0.0, 0.0, 0.0}, total = {0.0, 0.0, 0.0};
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    if (w == wave) pre = total;
    total = comb(total, wtot[w]);
  }
  Trip ex = shfl_up_trip(inc, 1);
  if (lane == 0) ex = {0.0, 0.0, 0.0};
  Trip p = comb(pre, ex);

  // ---- segment constants (solver.py:165-181 by cases) --------------------------------
  const double ga = (A.g && left_d) ? A.g[a] : 0.0;
  const double gb = (A.g && right_d) ? A.g[bn] : 0.0;
  double ua, C;
  if (left_d && right_d) {
    ua = ga;
    C = (gb - ga + total.T) / total.R;
  } else if (left_d) {
    ua = ga;
    C = total.F + bc_vals[2];
  } else if (right_d) {
    C = 0.0;
    ua = gb + total.T;
  } else {  // pure Neumann: singular (the reference returns garbage, solver.py:174)
    ua = C = __builtin_nan("");
  }

  // ---- sweep 2: nodal values ------------------------------------------------------------
  for (int q = lo; q < hi; ++q) {
    const double Fq = (q == 0 && left_d) ? 0.0 : Fs[q];
    const double r = Rs[q];
    p.F += Fq;
    p.R += r;
    p.T += p.F * r;
    Rs[q] = ua + C * p.R - p.T;          // value at node a+q+1
    if (ADJ) Fs[q] = (C - p.F) * r;      // lambda_{e+1} - lambda_e
  }
  __syncthreads();

  // ---- write-out, coalesced ----------------------------------------------------------------
  double* out = A.out + (long long)b * A.ldo;
  if (!ADJ) {
    for (int q = t; q < L; q += NT) out[a + q + 1] = (q == L - 1 && right_d) ? gb : Rs[q];
    if (t == 0) out[a] = ua;
  } else {
    for (int q = t; q <= L; q += NT) {
      const int i = a + q;
      double lam = q == 0 ? ua : Rs[q - 1];
      if ((q == 0 && left_d) || (q == L && right_d)) lam = 0.0;
      out[i] = lam * lumped_weight(A.x, i, A.n);  // df = M^T lambda
    }
    const double* u = A.u + (long long)b * A.ldu;
    double part = 0.0;
    for (int q = t; q < L; q += NT) {
      const int e = a + q;
      const double he = A.x[e + 1] - A.x[e];
      const double dk = -Fs[q] * (u[e + 1] - u[e]) / he;  // -lam_e^T k0_e u_e
      if (A.dk_e) A.dk_e[(long long)b * A.lddk + e] = dk;
      part += dk;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (t == 0) {
      double sum = 0.0;
      for (int w = 0; w < NT / 64; ++w) sum += red[w];
      A.dk_part[(long long)b * A.n_seg + s] = sum;
    }
  }
}

constexpr int kLdsBudget = 160 * 1024 - 2048;  // dynamic bytes one workgroup may take

template <int NT, bool ADJ>
int launch_nt(const ChainArgs& A, int max_len, hipStream_t st) {
  dim3 grid(A.n_seg, A.B);
  const size_t need = (size_t)2 * max_len * sizeof(double);
  if (need <= (size_t)kLdsBudget) {
    auto k = chain_kernel<NT, ADJ, true>;
    if (need > 48 * 1024) {
      int rc = check(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
      if (rc) return rc;
    }
    hipLaunchKernelGGL(k, grid, dim3(NT), need, st, A);
  } else {
    if (!A.stage) return DIFFHE_E_TOOBIG;
    hipLaunchKernelGGL((chain_kernel<NT, ADJ, false>), grid, dim3(NT), 0, st, A);
  }
  return check_launch();
}

template <bool ADJ>
int launch(const ChainArgs& A, int max_len, hipStream_t st) {
  if (max_len <= 2048) return launch_nt<256, ADJ>(A, max_len, st);
  return launch_nt<1024, ADJ>(A, max_len, st);
}

// Longest segment: the host passes n; segments never exceed n - 1 elements.  We size
// LDS by n - 1 (an upper bound) to avoid a device->host read of `seg`.
}  // namespace

extern "C" int diffhe_chain1d_solve(const double* x, const double* kappa, long long kappa_sb, long long kappa_se,
                                    const double* rhs, long long rhs_sb, const int* seg, int n_seg,
                                    const double* g, double* u, long long ldu, int n, int B, double* stage,
                                    void* stream) {
  if (!x || !kappa || !rhs || !seg || !g || !u || n < 2 || B < 1 || n_seg < 1) return DIFFHE_E_BADARG;
  ChainArgs A{};
  A.x = x; A.kappa = kappa; A.ksb = kappa_sb; A.kse = kappa_se; A.rhs = rhs; A.rhs_sb = rhs_sb;
  A.seg = seg; A.n_seg = n_seg; A.g = g; A.out = u; A.ldo = ldu; A.n = n; A.B = B;
  A.stage = stage; A.stage_len = n - 1;
  return launch<false>(A, n - 1, (hipStream_t)stream);
}

extern "C" int diffhe_chain1d_adjoint(const double* x, const double* kappa, long long kappa_sb,
                                      long long kappa_se, const double* gbar, long long gbar_sb, const double* u,
                                      long long ldu, const int* seg, int n_seg, double* df, long long lddf,
                                      double* dkappa_e, long long lddk, double* dkappa_part, int n, int B,
                                      double* stage, void* stream) {
  if (!x || !kappa || !gbar || !u || !seg || !df || !dkappa_part || n < 2 || B < 1 || n_seg < 1)
    return DIFFHE_E_BADARG;
  ChainArgs A{};
  A.x = x; A.kappa = kappa; A.ksb = kappa_sb; A.kse = kappa_se; A.rhs = gbar; A.rhs_sb = gbar_sb;
  A.seg = seg; A.n_seg = n_seg; A.g = nullptr; A.out = df; A.ldo = lddf; A.u = u; A.ldu = ldu;
  A.dk_e = dkappa_e; A.lddk = lddk; A.dk_part = dkappa_part; A.n = n; A.B = B;
  A.stage = stage; A.stage_len = n - 1;
  return launch<true>(A, n - 1, (hipStream_t)stream);
}

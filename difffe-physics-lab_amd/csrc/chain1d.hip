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
// so the solve is ONE scan per (segment, sample).  HBM traffic: read f, write u (16 n B); the adjoint
// reads gbar and u and writes df (+ dkappa): 24-32 n B.  K is never materialised.
//
// REFERENCE-ORDER mode (flag DIFFHE_CHAIN_REFERENCE_ORDER, the default of the Python boundary).  The scan
// solves the weighted Laplacian EXACTLY (to ~1e-15); the reference solves the matrix it assembled in fp64,
// whose diagonal is the ROUNDED sum fl(k_{i-1} + k_i) of the rounded weights k_e = fl(kappa_e / h_e)
// (solver.py:88-92).  That rounding is a data perturbation delta_i = fl(k_{i-1}+k_i) - (k_{i-1}+k_i) of
// the diagonal which the system amplifies by its condition number (0.4 N^2: 4e-10 in u at N = 10^4,
// measured against the reference itself, tests/golden/g10_*).  delta_i is exactly representable (TwoSum), so
// the reference's system is (K + diag(delta)) u = F with K the exact Laplacian of the weights k_e, and
//     u = u0 - K^{-1} (delta o u0) + O((cond eps)^2),        u0 = K^{-1} F
// i.e. by linearity u = K^{-1} (F - delta o u0): a first scan for u0, then the ordinary scan on the corrected
// load, in the same kernel, on data that is already in registers.  Result: 7e-12 from torch.linalg.solve at N = 10^4 -- which is also how far the exact solution
// of the reference's own rounded matrix is from what its LU returns.  The adjoint gets the same correction
// (autograd solves with the same rounded matrix).  The load is formed in the reference's order as well:
// F_i = fl(fl(h_{i-1}/2) f_i) + fl(fl(h_i/2) f_i) (solver.py:95-96).
//
// One workgroup per (segment, sample), two kernels:
//   chain_reg_kernel  segments up to 10 240 elements: the segment lives in REGISTERS, element order =
//                     thread order, so every global access is coalesced; scan = DPP row shifts across the
//                     wave + one LDS pass over the wave totals.
//   chain_kernel      longer segments: element integrals staged in a global scratch buffer (L2-resident
//                     while the workgroup uses it); thread t owns the chunk [t*c, (t+1)*c) of the staged
//                     arrays, block scan over the chunk composites.
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
  double* stage;         // global staging (chain_kernel only): narr arrays of (B, n - 1) doubles, indexed by element
  long long stage_arr;   // doubles per staged array = B * (n - 1)
};

__device__ inline double lumped_weight(const double* x, int i, int n) {
  const double xi = x[i];
  const double hl = i > 0 ? xi - x[i - 1] : 0.0;
  const double hr = i < n - 1 ? x[i + 1] - xi : 0.0;
  return 0.5 * (hl + hr);
}

// 1/d to within an ulp or so: hardware reciprocal + two Newton steps (a tenth of the IEEE division's cost)
__device__ inline double rcp_newton(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  return fma(r, fma(-d, r, 1.0), r);
}

// Reference-order weight k_e = fl(kappa_e / h_e): a true IEEE division (solver.py:84-88)
__device__ inline double ref_weight(const double* x, const double* kap, long long kse, int e) {
  return kap[(long long)e * kse] / (x[e + 1] - x[e]);   // hipcc expands fp64 '/' to the correctly rounded sequence
}

// delta = fl(a + b) - (a + b), exactly (Knuth TwoSum; additions only, nothing to contract)
__device__ inline double sum_rounding(double a, double b) {
  const double s = a + b;
  const double bb = s - a;
  return -((a - (s - bb)) + (b - bb));
}

// Load at node i (solver.py:95-96).  REF: the reference's operation order, every product and sum rounded on
// its own; else 0.5 (h_l + h_r) f_i.  ADJ right-hand sides are not weighted.
template <bool ADJ, bool REF>
__device__ inline double node_load(const double* x, const double* rhs, int i, int n) {
  if (ADJ) return rhs[i];
  const double xi = x[i], fi = rhs[i];
  const double hl = i > 0 ? xi - x[i - 1] : 0.0;
  const double hr = i < n - 1 ? x[i + 1] - xi : 0.0;
  if (REF) {
#pragma clang fp contract(off)
    const double tl = (hl * 0.5) * fi, tr = (hr * 0.5) * fi;   // h/2 is exact; each product rounded on its own
    return tl + tr;
  }
  return fi * (0.5 * (hl + hr));
}

// Segment constants (solver.py:165-181 by cases) from the scan total; Fend = load at the last node.
__device__ inline void seg_constants(const Trip& total, bool left_d, bool right_d, double ga, double gb, double Fend,
                                     double& ua, double& C) {
  if (left_d && right_d) {
    ua = ga;
    C = (gb - ga + total.T) / total.R;
  } else if (left_d) {
    ua = ga;
    C = total.F + Fend;
  } else if (right_d) {
    C = 0.0;
    ua = gb + total.T;
  } else {  // pure Neumann: singular (the reference returns garbage, solver.py:174)
    ua = C = __builtin_nan("");
  }
}

// ---------------------------------------------------------------------------------------------------
// Long segments: staged in global memory.  Arrays (indexed by element e of sample b): Fs load, then value
// difference across the element; Rs 1/k_e, then the nodal values; in reference-order mode Vs, the nodal values
// of the first solve.
// ---------------------------------------------------------------------------------------------------
template <int NT, bool ADJ, bool REF>
__global__ __launch_bounds__(NT) void chain_kernel(ChainArgs A) {
  __shared__ Trip wtot[NT / 64];
  __shared__ double red[NT / 64];

  const int s = blockIdx.x;
  const int b = blockIdx.z * gridDim.y + blockIdx.y;
  if (b >= A.B) return;
  const int a = A.seg[3 * s + 0];
  const int bn = A.seg[3 * s + 1];
  const int flags = A.seg[3 * s + 2];
  const bool left_d = flags & 1, right_d = flags & 2;
  const int L = bn - a;  // elements in the segment
  if (L <= 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int c = (L + NT - 1) / NT;
  c |= 1;
  const int lo = t * c;
  const int hi = lo + c < L ? lo + c : L;

  double* Fs = A.stage + (long long)b * (A.n - 1) + a;
  double* Rs = Fs + A.stage_arr;
  double* Vs = Rs + A.stage_arr;   // REF only
  const double* rhs = A.rhs + (long long)b * A.rhs_sb;
  const double* kap = A.kappa + (long long)b * A.ksb;

  // ---- stage: element integrals + load, coalesced --------------------------------
  for (int q = t; q < L; q += NT) {
    const int e = a + q;
    Fs[q] = (q == 0 && left_d) ? 0.0 : node_load<ADJ, REF>(A.x, rhs, e, A.n);
    if (REF) {
      Rs[q] = 1.0 / ref_weight(A.x, kap, A.kse, e);
    } else {
      const double he = A.x[e + 1] - A.x[e];               // solver.py:84-86
      Rs[q] = he / kap[(long long)e * A.kse];              // 1 / k_e, solver.py:88
    }
  }
  const double Fend = node_load<ADJ, REF>(A.x, rhs, bn, A.n);
  __syncthreads();

  const double ga = (A.g && left_d) ? A.g[a] : 0.0;
  const double gb = (A.g && right_d) ? A.g[bn] : 0.0;
  double ua = 0.0, C = 0.0;
  // One scan solve over the staged (Fs, Rs).  first (REF only): nodal values -> Vs, (Fs, Rs) stay;
  // else in place: values -> Rs (node a+q+1 at [q]), value differences -> Fs.
  auto solve = [&](bool first) {
    Trip acc = {0.0, 0.0, 0.0};
    for (int q = lo; q < hi; ++q) {
      const double r = Rs[q];
      acc.F += Fs[q];
      acc.R += r;
      acc.T += acc.F * r;
    }
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
    seg_constants(total, left_d, right_d, ga, gb, Fend, ua, C);
    for (int q = lo; q < hi; ++q) {
      const double r = Rs[q];
      p.F += Fs[q];
      p.R += r;
      p.T += p.F * r;
      const double val = ua + C * p.R - p.T;  // value at node a+q+1
      if (first) {
        Vs[q] = val;
      } else {
        Rs[q] = val;
        if (ADJ) Fs[q] = (C - p.F) * r;       // lambda_{e+1} - lambda_e
      }
    }
    __syncthreads();
  };

  if (REF) {
    // first solve u0; then the right-hand side F - delta o u0 of the reference's rounded system (top of the file)
    solve(true);
    for (int q = t + 1; q < L; q += NT) {    // interior nodes i = a+q of the segment (delta = 0 at its ends)
      const int i = a + q;
      Fs[q] -= sum_rounding(ref_weight(A.x, kap, A.kse, i - 1), ref_weight(A.x, kap, A.kse, i)) * Vs[q - 1];
    }
    __syncthreads();
  }
  solve(false);

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


// ---- register-resident variant (segments up to NT * EPT elements) -----------------------------
// Element q = k*NT + t lives in the registers of thread t (round k), so every global access is
// coalesced and nothing is staged: each round is scanned across the wave with DPP row shifts /
// row broadcasts of the (F,R,T) triple, the EPT * NT/64 wave totals are scanned once by wave 0
// through LDS, and each thread finishes its own elements from (block prefix) o (in-wave prefix).
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_f64(double v) {  // lanes without a source lane (or in masked rows) read 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (ROW_MASK == 0xf) {
    // every row is written and bound_ctrl supplies the 0 of lanes without a source: no `old` value to materialise
    // (update_dpp(0, ...) costs a v_mov per half: 281 of the 1576 instructions of the forward kernel)
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  } else {
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  }
  return __hiloint2double(hi, lo);
}

template <int CTRL, int ROW_MASK>
__device__ inline Trip scan_step(const Trip& v) {
  const Trip o = {dpp_f64<CTRL, ROW_MASK>(v.F), dpp_f64<CTRL, ROW_MASK>(v.R), dpp_f64<CTRL, ROW_MASK>(v.T)};
  return comb(o, v);  // (0,0,0) is the identity, so lanes that received nothing are unchanged
}

__device__ inline Trip wave_scan(Trip v) {  // inclusive scan over the 64 lanes, lane order = element order
  v = scan_step<0x111, 0xf>(v);  // row_shr:1
  v = scan_step<0x112, 0xf>(v);  // row_shr:2
  v = scan_step<0x114, 0xf>(v);  // row_shr:4
  v = scan_step<0x118, 0xf>(v);  // row_shr:8
  v = scan_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = scan_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}

__device__ inline Trip wave_excl(const Trip& incl) {  // wave_shr:1
  return {dpp_f64<0x138, 0xf>(incl.F), dpp_f64<0x138, 0xf>(incl.R), dpp_f64<0x138, 0xf>(incl.T)};
}

// Second-level scan by wave 0: blk[0..NB) block composites in element order -> their exclusive prefixes,
// blk[NB] = total.  Callers put a barrier before and after.
template <int NB>
__device__ inline void scan_blocks(Trip* blk, int lane) {
  constexpr int PER = (NB + 63) / 64;
  Trip loc[PER];
  Trip acc = {0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = lane * PER + j;
    loc[j] = acc;
    if (idx < NB) acc = comb(acc, blk[idx]);
  }
  const Trip incl = wave_scan(acc);
  const Trip ex = wave_excl(incl);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = lane * PER + j;
    if (idx < NB) blk[idx] = comb(ex, loc[j]);
  }
  if (lane == 63) blk[NB] = incl;
}

// One element's scan inputs: F (load at its LEFT node, 0 at a left Dirichlet end), r = 1/k_e and h_e.
template <bool ADJ, bool REF>
__device__ inline void chain_elem(const ChainArgs& A, const double* rhs, const double* kap, int a, int q, bool left_d,
                                  double& F, double& r, double& he, double& kt) {
  const int e = a + q;
  const double xe = A.x[e];
  he = A.x[e + 1] - xe;                                                        // solver.py:84-86
  kt = 0.0;
  if (REF) {
    F = (q == 0 && left_d) ? 0.0 : node_load<ADJ, true>(A.x, rhs, e, A.n);
    kt = kap[(long long)e * A.kse] / he;                                       // fl(kappa / h), solver.py:88
    r = rcp_newton(kt);
  } else {
    const double w = ADJ ? 1.0 : 0.5 * ((e > 0 ? xe - A.x[e - 1] : 0.0) + he); // solver.py:95-96
    F = (q == 0 && left_d) ? 0.0 : rhs[e] * w;
    r = he * rcp_newton(kap[(long long)e * A.kse]);                            // 1 / k_e, solver.py:88
  }
}

// Thread t owns the VEC consecutive elements (k*NT + t)*VEC .. +VEC-1 of round k (EPT = rounds * VEC
// elements per thread).  KEEP = 2: the element inputs (F, r) stay in registers between the two phases;
// 1: only F, r is recomputed; 0: both are loaded again for the output phase.  Measured on config 2
// (10^4 elements x 4096 samples): KEEP = 2 with VEC = 2 is fastest -- larger VEC (fewer scan rounds but
// strided loads) and the smaller-register variants that fit two workgroups per CU were all slower.
// REF (reference-order mode, see the top of the file): a first scan gives u0 at every element's left node,
// the load becomes H = F - delta o u0 in the same registers, and the ordinary scan + output phase runs on H
// (u = K^{-1} H by linearity: the first solve's values are never stored).
template <int NT, int EPT, int VEC, bool ADJ, int KEEP, bool REF>
__global__ __launch_bounds__(NT) void chain_reg_kernel(ChainArgs A) {
  constexpr int NW = NT / 64;
  constexpr int RND = EPT / VEC;
  constexpr int NB = RND * NW;          // wave-sized blocks of the segment, in element order
  static_assert(EPT % VEC == 0, "EPT must be a multiple of VEC");
  static_assert(!REF || KEEP >= 1, "reference-order mode rewrites the load in registers");
  __shared__ Trip blk_a[NB + 1];        // block composites, then their exclusive prefixes; [NB] = total
  __shared__ Trip blk_b[REF ? NB + 1 : 1];   // REF: the same for the second scan
  __shared__ double red[NW];
  __shared__ double ktl[REF ? NT * EPT : 1];   // REF: the weights k_e = fl(kappa/h) of the segment (the correction needs
                                               // each one twice more, and its left neighbour's: 16 B of LDS traffic
                                               // instead of 1.5 IEEE divisions per element)
  Trip* blk = blk_a;

  const int s = blockIdx.x;
  const int b = blockIdx.z * gridDim.y + blockIdx.y;
  if (b >= A.B) return;
  const int a = A.seg[3 * s + 0];
  const int bn = A.seg[3 * s + 1];
  const int flags = A.seg[3 * s + 2];
  const bool left_d = flags & 1, right_d = flags & 2;
  const int L = bn - a;
  if (L <= 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double* rhs = A.rhs + (long long)b * A.rhs_sb;
  const double* kap = A.kappa + (long long)b * A.ksb;

  Trip excl[RND];                       // in-wave exclusive prefix of this thread's run, per round
  double Fk[KEEP >= 1 ? EPT : 1], rk[KEEP >= 2 ? EPT : 1];
#pragma unroll
  for (int k = 0; k < RND; ++k) {
    Trip run = {0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int q = (k * NT + t) * VEC + j;
      double F = 0.0, r = 0.0, he, kt = 0.0;
      if (q < L) chain_elem<ADJ, REF>(A, rhs, kap, a, q, left_d, F, r, he, kt);
      if (REF && q < L) ktl[q] = kt;
      if (KEEP >= 1) Fk[k * VEC + j] = F;
      if (KEEP >= 2) rk[k * VEC + j] = r;
      run.F += F;
      run.R += r;
      run.T += run.F * r;
    }
    const Trip incl = wave_scan(run);
    excl[k] = wave_excl(incl);
    if (lane == 63) blk[k * NW + wave] = incl;
  }
  __syncthreads();
  if (wave == 0) scan_blocks<NB>(blk, lane);
  __syncthreads();

  const double ga = (A.g && left_d) ? A.g[a] : 0.0;
  const double gb = (A.g && right_d) ? A.g[bn] : 0.0;
  const double Fend = (left_d && !right_d) ? node_load<ADJ, REF>(A.x, rhs, bn, A.n) : 0.0;
  double ua, C;
  seg_constants(blk[NB], left_d, right_d, ga, gb, Fend, ua, C);

  if (REF) {
    // ---- first solve: u0 at the left node of every element; load <- F - delta o u0; second scan ----
#pragma unroll
    for (int k = 0; k < RND; ++k) {
      Trip p = comb(blk_a[k * NW + wave], excl[k]);
      Trip run = {0.0, 0.0, 0.0};
      double k_prev = 0.0;              // fl(kappa/h) of the element to the left of the current one
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int q = (k * NT + t) * VEC + j;
        double H = 0.0, r = 0.0;
        if (q < L) {
          const double kc = ktl[q];
          r = KEEP >= 2 ? rk[k * VEC + j] : rcp_newton(kc);
          if (j == 0 && q > 0) k_prev = ktl[q - 1];
          const double v_left = ua + C * p.R - p.T;        // u0 at node a+q (exclusive prefix)
          H = Fk[k * VEC + j];
          p.F += H;
          p.R += r;
          p.T += p.F * r;
          if (q > 0) H -= sum_rounding(k_prev, kc) * v_left;   // delta_{a+q} = 0 at q = 0 (segment end)
          k_prev = kc;
        }
        Fk[k * VEC + j] = H;
        run.F += H;
        run.R += r;
        run.T += run.F * r;
      }
      const Trip incl = wave_scan(run);
      excl[k] = wave_excl(incl);
      if (lane == 63) blk_b[k * NW + wave] = incl;
    }
    __syncthreads();
    if (wave == 0) scan_blocks<NB>(blk_b, lane);
    __syncthreads();
    blk = blk_b;
    seg_constants(blk[NB], left_d, right_d, ga, gb, Fend, ua, C);
  }

  double* out = A.out + (long long)b * A.ldo;
  const double* u = ADJ ? A.u + (long long)b * A.ldu : nullptr;
  double part = 0.0;
#pragma unroll
  for (int k = 0; k < RND; ++k) {
    Trip p = comb(blk[k * NW + wave], excl[k]);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int q = (k * NT + t) * VEC + j;
      if (q < L) {
        double F, r, he;
        if (KEEP >= 2) {
          F = Fk[k * VEC + j];
          r = rk[k * VEC + j];
          if (ADJ) he = A.x[a + q + 1] - A.x[a + q];
        } else if (KEEP == 1) {
          F = Fk[k * VEC + j];
          he = A.x[a + q + 1] - A.x[a + q];
          r = REF ? rcp_newton(kap[(long long)(a + q) * A.kse] / he) : he * rcp_newton(kap[(long long)(a + q) * A.kse]);
        } else {
          double kt_;
          chain_elem<ADJ, false>(A, rhs, kap, a, q, left_d, F, r, he, kt_);
        }
        p.F += F;
        p.R += r;
        p.T += p.F * r;
        const double val = ua + C * p.R - p.T;        // value at node a+q+1
        const bool last_d = (q == L - 1) && right_d;
        if (!ADJ) {
          out[a + q + 1] = last_d ? gb : val;
        } else {
          const int e = a + q;
          out[e + 1] = last_d ? 0.0 : val * lumped_weight(A.x, e + 1, A.n);  // df = M^T lambda
          const double dlam = (C - p.F) * r;                                  // lambda_{e+1} - lambda_e
          const double dk = -dlam * (u[e + 1] - u[e]) / he;                   // -lam_e^T k0_e u_e
          if (A.dk_e) A.dk_e[(long long)b * A.lddk + e] = dk;
          part += dk;
        }
      }
    }
  }
  if (t == 0) out[a] = ADJ ? (left_d ? 0.0 : ua * lumped_weight(A.x, a, A.n)) : ua;
  if (ADJ) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (t == 0) {
      double sum = 0.0;
      for (int w = 0; w < NW; ++w) sum += red[w];
      A.dk_part[(long long)b * A.n_seg + s] = sum;
    }
  }
}

constexpr int kMaxGridY = 32768;    // samples beyond it go to grid.z (b = z * gridDim.y + y)
constexpr int kRegLimit = 10240;   // longest segment the register kernel takes

inline dim3 sample_grid(const ChainArgs& A) {
  const int gy = A.B < kMaxGridY ? A.B : kMaxGridY;
  return dim3(A.n_seg, gy, (A.B + gy - 1) / gy);
}

template <int NT, int EPT, int VEC, bool ADJ, bool REF>
int launch_reg(const ChainArgs& A, hipStream_t st) {
  hipLaunchKernelGGL((chain_reg_kernel<NT, EPT, VEC, ADJ, 2, REF>), sample_grid(A), dim3(NT), 0, st, A);
  return check_launch();
}

template <bool ADJ, bool REF>
int launch(const ChainArgs& A, int max_len, hipStream_t st) {
  if (max_len <= 256) return launch_reg<64, 4, 2, ADJ, REF>(A, st);
  if (max_len <= 1024) return launch_reg<256, 4, 2, ADJ, REF>(A, st);
  if (max_len <= 4096) return launch_reg<1024, 4, 2, ADJ, REF>(A, st);
  if (max_len <= kRegLimit) return launch_reg<1024, 10, 2, ADJ, REF>(A, st);
  if (!A.stage) return DIFFHE_E_TOOBIG;
  hipLaunchKernelGGL((chain_kernel<1024, ADJ, REF>), sample_grid(A), dim3(1024), 0, st, A);
  return check_launch();
}

inline bool bad_len(int max_seg_len, int n) { return max_seg_len < 1 || max_seg_len > n - 1; }

}  // namespace

extern "C" long long diffhe_chain1d_stage_doubles(int n, int B, int max_seg_len, int flags) {
  if (n < 2 || B < 1 || bad_len(max_seg_len, n)) return -1;
  if (max_seg_len <= kRegLimit) return 0;
  return (long long)((flags & DIFFHE_CHAIN_REFERENCE_ORDER) ? 3 : 2) * B * (n - 1);
}

extern "C" int diffhe_chain1d_solve(const double* x, const double* kappa, long long kappa_sb, long long kappa_se,
                                    const double* rhs, long long rhs_sb, const int* seg, int n_seg,
                                    const double* g, double* u, long long ldu, int n, int B, int max_seg_len,
                                    int flags, double* stage, void* stream) {
  if (!x || !kappa || !rhs || !seg || !g || !u || n < 2 || B < 1 || n_seg < 1 || bad_len(max_seg_len, n))
    return DIFFHE_E_BADARG;
  ChainArgs A{};
  A.x = x; A.kappa = kappa; A.ksb = kappa_sb; A.kse = kappa_se; A.rhs = rhs; A.rhs_sb = rhs_sb;
  A.seg = seg; A.n_seg = n_seg; A.g = g; A.out = u; A.ldo = ldu; A.n = n; A.B = B;
  A.stage = stage; A.stage_arr = (long long)B * (n - 1);
  if (flags & DIFFHE_CHAIN_REFERENCE_ORDER) return launch<false, true>(A, max_seg_len, (hipStream_t)stream);
  return launch<false, false>(A, max_seg_len, (hipStream_t)stream);
}

extern "C" int diffhe_chain1d_adjoint(const double* x, const double* kappa, long long kappa_sb,
                                      long long kappa_se, const double* gbar, long long gbar_sb, const double* u,
                                      long long ldu, const int* seg, int n_seg, double* df, long long lddf,
                                      double* dkappa_e, long long lddk, double* dkappa_part, int n, int B,
                                      int max_seg_len, int flags, double* stage, void* stream) {
  if (!x || !kappa || !gbar || !u || !seg || !df || !dkappa_part || n < 2 || B < 1 || n_seg < 1 ||
      bad_len(max_seg_len, n))
    return DIFFHE_E_BADARG;
  ChainArgs A{};
  A.x = x; A.kappa = kappa; A.ksb = kappa_sb; A.kse = kappa_se; A.rhs = gbar; A.rhs_sb = gbar_sb;
  A.seg = seg; A.n_seg = n_seg; A.g = nullptr; A.out = df; A.ldo = lddf; A.u = u; A.ldu = ldu;
  A.dk_e = dkappa_e; A.lddk = lddk; A.dk_part = dkappa_part; A.n = n; A.B = B;
  A.stage = stage; A.stage_arr = (long long)B * (n - 1);
  if (flags & DIFFHE_CHAIN_REFERENCE_ORDER) return launch<true, true>(A, max_seg_len, (hipStream_t)stream);
  return launch<true, false>(A, max_seg_len, (hipStream_t)stream);
}

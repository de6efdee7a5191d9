// General P1 path (any 1D/2D mesh): element integrals, assembly into a batch-shared ELL
// pattern, Dirichlet elimination, batched Jacobi-PCG, gradient contraction, layout changes.
//
// Data layout: node-major, batch-innermost (n, Bp): entry (i, b) at i*Bp + b.  A wave's 64
// lanes are 64 samples of one node (Bp >= 64), so every load -- including the ELL "gather"
// p[col] -- is one contiguous 512 B segment, column indices are wave-uniform and amortised
// over the batch, and per-sample dot products are per-lane sums with no cross-lane traffic.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "common.h"

namespace {

using namespace diffhe;

typedef long long i64;

// ---------------------------------------------------------------------------------------
// Element integrals (reference solver.py:84-88 1D, solver.py:119-139 2D)
// ---------------------------------------------------------------------------------------
__device__ inline void tri_integrals(double xi, double yi, double xj, double yj, double xk, double yk, double* k0,
                                     double* area_out) {
  const double area = 0.5 * fabs((xj - xi) * (yk - yi) - (xk - xi) * (yj - yi));  // solver.py:119
  const double bb[3] = {yj - yk, yk - yi, yi - yj};                               // solver.py:125-129
  const double cc[3] = {xk - xj, xi - xk, xj - xi};                               // solver.py:130-134
  const bool keep = !(area < 1e-15);                                              // solver.py:120-121
  const double inv = keep ? 1.0 / (4.0 * area) : 0.0;
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) k0[p * 3 + q] = keep ? (bb[p] * bb[q] + cc[p] * cc[q]) * inv : 0.0;
  *area_out = keep ? area : 0.0;
}

__global__ __launch_bounds__(256) void element_integrals_kernel(const double* __restrict__ coords,
                                                                 const int* __restrict__ elems, int dim, int n, int m,
                                                                 double* __restrict__ k0, double* __restrict__ m0) {
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (i64)gridDim.x * blockDim.x) {
    if (dim == 1) {
      const int i = elems[e], j = elems[(i64)m + e];
      const double h = coords[j] - coords[i];
      const double k = 1.0 / h;
      k0[e] = k; k0[(i64)m + e] = -k; k0[2 * (i64)m + e] = -k; k0[3 * (i64)m + e] = k;
      m0[e] = 0.5 * h; m0[(i64)m + e] = 0.0; m0[2 * (i64)m + e] = 0.0; m0[3 * (i64)m + e] = 0.5 * h;
    } else {
      const int i = elems[e], j = elems[(i64)m + e], k = elems[2 * (i64)m + e];
      double loc[9], area;
      tri_integrals(coords[i], coords[(i64)n + i], coords[j], coords[(i64)n + j], coords[k], coords[(i64)n + k], loc,
                    &area);
#pragma unroll
      for (int pq = 0; pq < 9; ++pq) {
        k0[(i64)pq * m + e] = loc[pq];
        m0[(i64)pq * m + e] = area / 9.0;  // F_p += area/3 * (f_i+f_j+f_k)/3, solver.py:143-145
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Deterministic row-gather assembly + Dirichlet elimination
// ---------------------------------------------------------------------------------------
// REF = true: `local` holds t = b_p b_q + c_p c_q (2D) or 1 / -1 (1D) and `den` holds 4 area (2D) or h (1D); every
// contribution is formed as (kappa * t) / den with each operation rounded on its own and added in element order --
// the operation order of the reference's loops (solver.py:88-92, :139-140), so the stored values (and the lifting
// terms) are bit-identical to the reference's K.  REF = false: kappa * k0 with contracted multiply-adds.
template <bool REF>
__global__ __launch_bounds__(256) void assemble_rows_kernel(
    const double* __restrict__ local, const double* __restrict__ den, const double* __restrict__ kappa, i64 kse,
    i64 ksb, const int* __restrict__ ent_ptr, const int* __restrict__ contrib, const int* __restrict__ cols,
    const int* __restrict__ store_slot, const unsigned char* __restrict__ is_bc, const double* __restrict__ g,
    double* __restrict__ vals, double* __restrict__ lift, int n, int m, int W, int Bv) {
#pragma clang fp contract(off)
  const NodeMap nm = node_map(Bv);
  if (nm.b >= Bv) return;
  for (int i = nm.node0; i < n; i += nm.stride) {
    const bool row_bc = is_bc && is_bc[i];
    double lf = 0.0;
    for (int k = 0; k < W; ++k) {
      const i64 ent = (i64)k * n + i;
      const int store = store_slot ? store_slot[k] : k;
      const int j = cols[ent];
      const bool col_bc = is_bc && j != i && is_bc[j];
      // entries that are not stored (lower triangle of a symmetric format) only matter for the lift
      if (store < 0 && (row_bc || !col_bc)) continue;
      const int c0 = ent_ptr[ent], c1 = ent_ptr[ent + 1];
      double v = 0.0;
      for (int c = c0; c < c1; ++c) {
        const int code = contrib[c];
        const int e = code >> 6, pq = code & 63;   // local entry p * npe + q: < 9 for P1 triangles, < 36 for P2
        const double kap = kappa ? kappa[(i64)e * kse + (i64)nm.b * ksb] : 1.0;
        if (REF) {
          const double num = kap * local[(i64)pq * m + e];
          v = v + num / den[e];  // K[p,q] = K[p,q] + kappa * t / (4 area), solver.py:139-140
        } else {
          v = fma(kap, local[(i64)pq * m + e], v);  // K[p,q] += kappa * k0[p,q], solver.py:89-92/:137-140
        }
      }
      if (row_bc) {
        v = (k == 0) ? 1.0 : 0.0;
      } else if (col_bc) {
        lf += v * g[j];  // F_free -= K[free,bc] g, solver.py:166-169
        v = 0.0;
      }
      if (store >= 0) vals[((i64)store * n + i) * Bv + nm.b] = v;
    }
    if (lift) lift[(i64)i * Bv + nm.b] = lf;
  }
}

// ---------------------------------------------------------------------------------------
// The same gather for FEMesh.rectangle connectivity, lists written into the code (diffhe/plan.py: build_dia_pattern):
// quad (r, c), q = r nx + c, holds T0 = [a, b, d] = element 2q and T1 = [b, c, d] = element 2q + 1; node (r, c) sees the
// six triangles A = T1(r-1,c-1), B = T0(r-1,c), C = T1(r-1,c), D = T0(r,c-1), E = T1(r,c-1), F = T0(r,c).  Same
// contributions, same element order, same fma chain as assemble_rows_kernel<false> -- bitwise the same values -- but
// no index lists to read, each kappa_e loaded once per node (six loads, wave-uniform addresses + lane = sample) instead
// of once per contribution (up to 18), local integrals as scalar loads.  One wave per node, lanes over samples.
// Seven entry kinds per row in the order (0, +1, +W, +nx, -1, -W, -nx); the first nd are stored, the others only feed
// the Dirichlet lift.
// ---------------------------------------------------------------------------------------
// The seven entries of node i's row (and its Dirichlet lift) from the kappa of its six triangles: shared by the
// node-per-wave kernel and the strip kernel below, so that both produce bitwise the same values.
__device__ __forceinline__ void lattice_node_entries(const double* __restrict__ local, i64 lm, i64 emask, double kA,
                                                     double kB, double kC, double kD, double kE, double kF, i64 eA, i64 eB,
                                                     i64 eC, i64 eD, i64 eE, i64 eF, bool up, bool dn, bool lf, bool rt,
                                                     int i, int W, int nx, i64 n, int nd,
                                                     const unsigned char* __restrict__ is_bc,
                                                     const double* __restrict__ g, double* __restrict__ vals,
                                                     double* __restrict__ lift, int Bv, int b) {
#pragma clang fp contract(off)
    const bool hA = dn && lf, hBC = dn && rt, hDE = up && lf, hF = up && rt;
    auto loc = [&](int pq, i64 e) -> double { return local[(i64)pq * lm + (e & emask)]; };
    const bool row_bc = is_bc && is_bc[i];
    double lfv = 0.0;
    // entry kinds: offsets and contribution lists (mask, element, kappa, local entry), increasing element id
#define CONTRIB(mask_, k_, pq_, e_) if (mask_) v = fma((k_), loc((pq_), (e_)), v)
#define ENTRY(kind_, off_, any_, BODY)                                                        \
    {                                                                                         \
      const int store = (kind_) < nd ? (kind_) : -1;                                          \
      const i64 j = (any_) ? (i64)i + (off_) : (i64)i;                                        \
      const bool col_bc = is_bc && j != i && is_bc[j];                                        \
      if (!(store < 0 && (row_bc || !col_bc))) {                                              \
        double v = 0.0;                                                                       \
        BODY                                                                                  \
        if (row_bc) v = ((kind_) == 0) ? 1.0 : 0.0;                                           \
        else if (col_bc) { lfv += v * g[j]; v = 0.0; }                                        \
        if (store >= 0) vals[((i64)store * n + i) * Bv + b] = v;                           \
      }                                                                                       \
    }
    ENTRY(0, 0, true, CONTRIB(hA, kA, 4, eA); CONTRIB(hBC, kB, 8, eB); CONTRIB(hBC, kC, 8, eC); CONTRIB(hDE, kD, 4, eD);
          CONTRIB(hDE, kE, 0, eE); CONTRIB(hF, kF, 0, eF);)
    ENTRY(1, 1, rt, CONTRIB(rt && dn, kC, 7, eC); CONTRIB(rt && up, kF, 1, eF);)
    ENTRY(2, W, up, CONTRIB(up && lf, kE, 1, eE); CONTRIB(up && rt, kF, 2, eF);)
    ENTRY(3, nx, up && lf, CONTRIB(up && lf, kD, 5, eD); CONTRIB(up && lf, kE, 2, eE);)
    ENTRY(4, -1, lf, CONTRIB(lf && dn, kA, 5, eA); CONTRIB(lf && up, kD, 3, eD);)
    ENTRY(5, -W, dn, CONTRIB(dn && lf, kA, 3, eA); CONTRIB(dn && rt, kB, 6, eB);)
    ENTRY(6, -nx, dn && rt, CONTRIB(dn && rt, kB, 7, eB); CONTRIB(dn && rt, kC, 6, eC);)
#undef ENTRY
#undef CONTRIB
    if (lift) lift[(i64)i * Bv + b] = lfv;
}

// `local` is (9, m), or -- compact form, emask = 1, lm = 2 -- (9, 2): one unit matrix per triangle ORIENTATION (element
// parity) of a lattice whose triangles are congruent bit for bit (FEMesh.rectangle with exactly representable spacing:
// the bench mesh).  Same values, same order; the 18 wave-uniform loads per node then hit a 144-byte table instead of a
// (9, m) array (151 MB at 1024^2, which lives in the Infinity Cache at best): 6.7 -> see DESIGN section 6, round 4.
__global__ __launch_bounds__(256) void lattice_assemble_kernel(const double* __restrict__ local, i64 lm, i64 emask,
                                                                const double* __restrict__ kappa, i64 kse, i64 ksb,
                                                                const unsigned char* __restrict__ is_bc,
                                                                const double* __restrict__ g, double* __restrict__ vals,
                                                                double* __restrict__ lift, int nx, int ny, int nd,
                                                                int Bv) {
#pragma clang fp contract(off)
  const NodeMap nm = node_map(Bv);
  if (nm.b >= Bv) return;
  const int W = nx + 1;
  const i64 n = (i64)W * (ny + 1);
  const i64 kb = (i64)nm.b * ksb;
  for (int i = nm.node0; i < n; i += nm.stride) {
    const int r = i / W, c = i - r * W;
    const bool up = r < ny, dn = r >= 1, lf = c >= 1, rt = c < nx;
    // element ids (valid only under their masks)
    const i64 eA = 2 * ((i64)(r - 1) * nx + (c - 1)) + 1, eB = 2 * ((i64)(r - 1) * nx + c), eC = eB + 1;
    const i64 eD = 2 * ((i64)r * nx + (c - 1)), eE = eD + 1, eF = 2 * ((i64)r * nx + c);
    const bool hA = dn && lf, hBC = dn && rt, hDE = up && lf, hF = up && rt;
    const double kA = hA ? (kappa ? kappa[eA * kse + kb] : 1.0) : 0.0;
    const double kB = hBC ? (kappa ? kappa[eB * kse + kb] : 1.0) : 0.0;
    const double kC = hBC ? (kappa ? kappa[eC * kse + kb] : 1.0) : 0.0;
    const double kD = hDE ? (kappa ? kappa[eD * kse + kb] : 1.0) : 0.0;
    const double kE = hDE ? (kappa ? kappa[eE * kse + kb] : 1.0) : 0.0;
    const double kF = hF ? (kappa ? kappa[eF * kse + kb] : 1.0) : 0.0;
    lattice_node_entries(local, lm, emask, kA, kB, kC, kD, kE, kF, eA, eB, eC, eD, eE, eF, up, dn, lf, rt, i, W, nx, n, nd,
                         is_bc, g, vals, lift, Bv, nm.b);
  }
}

// The same assembly as a STRIP pass (per-sample kappa fields on big levels): a wave owns RW node columns x 64 samples
// and marches down the node rows with the kappa of two quad rows in registers -- every kappa_e is loaded once per wave
// and quad row (2 (RW + 1) loads per RW nodes) instead of once per incident node (6 per node).  Same per-node arithmetic
// (lattice_node_entries): bitwise the values of lattice_assemble_kernel.
constexpr int kAsmCols = 4;
__global__ __launch_bounds__(256) void lattice_assemble_strip_kernel(const double* __restrict__ local, i64 lm, i64 emask,
                                                                      const double* __restrict__ kappa, i64 kse, i64 ksb,
                                                                      const unsigned char* __restrict__ is_bc,
                                                                      const double* __restrict__ g,
                                                                      double* __restrict__ vals, double* __restrict__ lift,
                                                                      int nx, int ny, int nd, int Bv, int ncb, int TR) {
  constexpr int RW = kAsmCols;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int b = blockIdx.y * kWave + lane;
  const int rc = blockIdx.x / ncb, cb = blockIdx.x - rc * ncb;
  const int c0 = (cb * 4 + wave) * RW;              // first node column
  const int r0 = rc * TR;
  const int r1 = (r0 + TR < ny + 1) ? r0 + TR : ny + 1;
  if (c0 > nx || r0 >= r1) return;
  const int W = nx + 1;
  const i64 n = (i64)W * (ny + 1);
  const i64 kb = (i64)b * ksb;
  // kappa of quad row qr on the quad columns c0 - 1 + j, both triangles; 0 outside the grid (never used there: masks)
  double lo[RW + 1][2], hi[RW + 1][2];
  auto load_quads = [&](int qr, double (*dst)[2]) {
#pragma unroll
    for (int j = 0; j < RW + 1; ++j) {
      const int qc = c0 - 1 + j;
      const bool ok = qr >= 0 && qr < ny && qc >= 0 && qc < nx;
      const i64 e = 2 * ((i64)qr * nx + qc);
      dst[j][0] = ok ? kappa[e * kse + kb] : 0.0;
      dst[j][1] = ok ? kappa[(e + 1) * kse + kb] : 0.0;
    }
  };
  load_quads(r0 - 1, lo);
  for (int r = r0; r < r1; ++r) {
    load_quads(r, hi);
    const bool up = r < ny, dn = r >= 1;
#pragma unroll
    for (int k = 0; k < RW; ++k) {
      const int c = c0 + k;
      if (c > nx) continue;
      const bool lf = c >= 1, rt = c < nx;
      const i64 eA = 2 * ((i64)(r - 1) * nx + (c - 1)) + 1, eB = 2 * ((i64)(r - 1) * nx + c), eC = eB + 1;
      const i64 eD = 2 * ((i64)r * nx + (c - 1)), eE = eD + 1, eF = 2 * ((i64)r * nx + c);
      // node (r, c): A = T1(r-1, c-1), B = T0(r-1, c), C = T1(r-1, c), D = T0(r, c-1), E = T1(r, c-1), F = T0(r, c)
      lattice_node_entries(local, lm, emask, lo[k][1], lo[k + 1][0], lo[k + 1][1], hi[k][0], hi[k][1], hi[k + 1][0], eA, eB,
                           eC, eD, eE, eF, up, dn, lf, rt, r * W + c, W, nx, n, nd, is_bc, g, vals, lift, Bv, b);
    }
#pragma unroll
    for (int j = 0; j < RW + 1; ++j) { lo[j][0] = hi[j][0]; lo[j][1] = hi[j][1]; }
  }
}

// ---------------------------------------------------------------------------------------
// Element-parallel assembly with fp64 atomics; element integrals staged in LDS
// ---------------------------------------------------------------------------------------
constexpr int kElemTile = 64;

__global__ __launch_bounds__(256) void assemble_atomic_kernel(const double* __restrict__ coords,
                                                               const int* __restrict__ elems, int dim,
                                                               const double* __restrict__ kappa, i64 kse, i64 ksb,
                                                               const int* __restrict__ slot_of,
                                                               double* __restrict__ vals, int n, int m, int Bp) {
  __shared__ double k0s[9][kElemTile];
  __shared__ int rows[3][kElemTile];
  const int npe = dim + 1, nloc = npe * npe;
  const int LB = Bp < kWave ? Bp : kWave;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y * kWave + (lane % LB);
  const int sub = lane / LB, nsub = kWave / LB;
  for (i64 base = (i64)blockIdx.x * kElemTile; base < m; base += (i64)gridDim.x * kElemTile) {
    __syncthreads();
    if (threadIdx.x < kElemTile && base + threadIdx.x < m) {
      const i64 e = base + threadIdx.x;
      const int t = threadIdx.x;
      if (dim == 1) {
        const int i = elems[e], j = elems[(i64)m + e];
        const double k = 1.0 / (coords[j] - coords[i]);
        k0s[0][t] = k; k0s[1][t] = -k; k0s[2][t] = -k; k0s[3][t] = k;
        rows[0][t] = i; rows[1][t] = j;
      } else {
        const int i = elems[e], j = elems[(i64)m + e], k = elems[2 * (i64)m + e];
        double loc[9], area;
        tri_integrals(coords[i], coords[(i64)n + i], coords[j], coords[(i64)n + j], coords[k], coords[(i64)n + k],
                      loc, &area);
#pragma unroll
        for (int pq = 0; pq < 9; ++pq) k0s[pq][t] = loc[pq];
        rows[0][t] = i; rows[1][t] = j; rows[2][t] = k;
      }
    }
    __syncthreads();
    if (b >= Bp) continue;
    for (int el = wave * nsub + sub; el < kElemTile && base + el < m; el += 4 * nsub) {
      const i64 e = base + el;
      const double kap = kappa ? kappa[e * kse + (i64)b * ksb] : 1.0;
      for (int pq = 0; pq < nloc; ++pq) {
        const int slot = slot_of[(i64)pq * m + e];
        const int row = rows[pq / npe][el];
        unsafeAtomicAdd(&vals[((i64)slot * n + row) * Bp + b], kap * k0s[pq][el]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void apply_dirichlet_kernel(const int* __restrict__ cols,
                                                               const unsigned char* __restrict__ is_bc,
                                                               const double* __restrict__ g, double* __restrict__ vals,
                                                               double* __restrict__ F, int n, int W, int Bp) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  for (int i = nm.node0; i < n; i += nm.stride) {
    const bool row_bc = is_bc[i];
    double lf = 0.0;
    for (int k = 0; k < W; ++k) {
      const i64 ent = (i64)k * n + i;
      const int j = cols[ent];
      if (row_bc) {
        vals[ent * Bp + nm.b] = (k == 0) ? 1.0 : 0.0;
      } else if (j != i && is_bc[j]) {
        lf += vals[ent * Bp + nm.b] * g[j];
        vals[ent * Bp + nm.b] = 0.0;
      }
    }
    if (F) F[(i64)i * Bp + nm.b] = row_bc ? 0.0 : F[(i64)i * Bp + nm.b] - lf;
  }
}

// ---------------------------------------------------------------------------------------
// Row i of an ELL matrix against x for this lane's sample: acc -/+= a_k x[col_k], k = 0 .. W-1 in that order (the order
// and the operations of the plain loop: bitwise the same sums).  The loads of NU entries are issued before the first
// product (the plain loop waited for col_k, then for x[col_k], entry by entry: two exposed latencies per entry, 0.2 of
// the HBM rate at 512^2 x 64); entries beyond W repeat entry 0 with the value 0.  With batches of >= 64 a wave works on
// ONE node (FOR_EACH_NODE below) and takes ell_row_uniform instead.
// ---------------------------------------------------------------------------------------
template <bool SUB, int NU, typename TV, typename TM>
__device__ __forceinline__ double ell_row_chunks(double acc, const TM* __restrict__ vals, const int* __restrict__ cols,
                                                 const TV* __restrict__ x, int i, int n, int W, int Bp, int Bv, int b) {
  const int vb = Bv == 1 ? 0 : b;
  for (int k0 = 0; k0 < W; k0 += NU) {
    int c[NU];
    double a[NU];
    TV xv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const bool in = k0 + u < W;
      const i64 ent = (i64)(in ? k0 + u : 0) * n + i;
      c[u] = cols[ent];
      a[u] = in ? (double)vals[ent * Bv + vb] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) xv[u] = x[(i64)c[u] * Bp + b];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (SUB) acc -= a[u] * (double)xv[u];
      else acc += a[u] * (double)xv[u];
    }
  }
  return acc;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// The wave-uniform form (a wave = one node, all 64 lanes active): ONE vector load brings the node's column indices
// (lane k: entry k; scalar loads came out one after the other, each waited for) and, for a batch-shared matrix
// (SHARED), ONE its values; v_readlane hands them out as scalars BEFORE the gathers are issued, so the NU gathers of a
// chunk are in flight together, each through a scalar row base + the lane's offset.  FULL: W == NU, one chunk, no
// padding (P1 triangulations of lattice connectivity: 7 entries per row).  Same entries, same order, same operations.
template <bool SUB, bool SHARED, int NU, bool FULL, typename TV, typename TM>
__device__ __forceinline__ double ell_row_uniform(double acc, const TM* __restrict__ vals, const int* __restrict__ cols,
                                                  const TV* __restrict__ x, int i, int n, int W, int Bp, int b) {
  const int lane = threadIdx.x & 63;
  for (int k0 = 0; k0 < W; k0 += kWave) {
    const int nk = FULL ? NU : (W - k0 < kWave ? W - k0 : kWave);
    const i64 entl = (i64)(k0 + (lane < nk ? lane : 0)) * n + i;
    const int cv = cols[entl];
    double av = 0.0;
    if (SHARED) av = (double)vals[entl];
    for (int u0 = 0; u0 < nk; u0 += NU) {
      int c[NU];
      double a[NU];
      TV xv[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const bool in = FULL || u0 + u < nk;
        const int k = in ? u0 + u : 0;
        c[u] = __builtin_amdgcn_readlane(cv, k);
        if (SHARED) a[u] = in ? readlane_f64(av, k) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const TV* __restrict__ xr = x + (i64)c[u] * Bp;   // scalar row base
        xv[u] = xr[b];
      }
      if (!SHARED) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const bool in = FULL || u0 + u < nk;
          const TM* __restrict__ vr = vals + ((i64)(k0 + (in ? u0 + u : 0)) * n + i) * Bp;
          a[u] = in ? (double)vr[b] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (SUB) acc -= a[u] * (double)xv[u];
        else acc += a[u] * (double)xv[u];
      }
    }
  }
  return acc;
}
template <bool SUB, bool UNI, bool SHARED, typename TV, typename TM>
__device__ __forceinline__ double ell_row(double acc, const TM* __restrict__ vals, const int* __restrict__ cols,
                                          const TV* __restrict__ x, int i, int n, int W, int Bp, int Bv, int b) {
  if (UNI)
    return W == 7 ? ell_row_uniform<SUB, SHARED, 7, true>(acc, vals, cols, x, i, n, W, Bp, b)
                  : ell_row_uniform<SUB, SHARED, 8, false>(acc, vals, cols, x, i, n, W, Bp, b);
  return ell_row_chunks<SUB, 8>(acc, vals, cols, x, i, n, W, Bp, Bv, b);
}
__device__ int g_xcd_ranges = 1;   // DIFFHE_ELL_XCD=0 (read once per process, sync_xcd_switch): the plain grid-stride walk
__device__ __forceinline__ bool xcd_ranges() { return g_xcd_ranges != 0; }
// for (i over this lane's nodes) BODY -- with batches of >= 64 through wave-uniform indices (see ell_row_uniform), and
// with the nodes dealt to the XCDs in CONTIGUOUS ranges: workgroups go round-robin to the 8 XCDs (block b -> XCD b % 8),
// each with its own L2, and a row's neighbours sit close to it in any sensible numbering.  With the plain grid-stride
// walk every XCD saw every 8th group of 4 nodes, so each L2 fetched nearly ALL of x (the gathers of the fine-level Jacobi
// sweep at 512^2 x 64 moved ~8x the vector through the fabric); now the blocks of one XCD sweep one eighth of the nodes
// together and the gathers hit their own L2.  Which nodes a block sums changes, not the fixed order: still reproducible.
// nodes first, first + step, ... < hi of this wave in the wave-per-node walk (XCD-contiguous ranges when the grid allows)
__device__ __forceinline__ void wave_node_range(int n, int& first, int& hi, int& step) {
  const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  first = (int)blockIdx.x * 4 + wave_u;
  hi = n;
  step = (int)gridDim.x * 4;
  if ((gridDim.x & 7) == 0 && xcd_ranges()) {
    const int chunk = (n + 7) >> 3, lo = ((int)blockIdx.x & 7) * chunk;
    hi = lo + chunk < n ? lo + chunk : n;
    first = lo + ((int)blockIdx.x >> 3) * 4 + wave_u;
    step = ((int)gridDim.x >> 3) * 4;
  }
}
#define FOR_EACH_NODE(nm_, n_, Bp_, Bv_, ...)                                                     \
  do {                                                                                            \
    if ((Bp_) >= kWave) {                                                                         \
      constexpr bool kUni = true;                                                                 \
      int first_, hi_, step_;                                                                     \
      wave_node_range((n_), first_, hi_, step_);                                                  \
      if ((Bv_) == 1) {                                                                           \
        constexpr bool kShared = true;                                                            \
        for (int i = first_; i < hi_; i += step_) __VA_ARGS__                                     \
      } else {                                                                                    \
        constexpr bool kShared = false;                                                           \
        for (int i = first_; i < hi_; i += step_) __VA_ARGS__                                     \
      }                                                                                           \
    } else {                                                                                      \
      constexpr bool kUni = false;                                                                \
      constexpr bool kShared = false;                                                             \
      for (int i = (nm_).node0; i < (n_); i += (nm_).stride) __VA_ARGS__                          \
    }                                                                                             \
  } while (0)

// ---------------------------------------------------------------------------------------
// y = (is_bc ? 0 : M x - sub), M batch-shared ELL.  Load vector and df = M^T lambda.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 8) void spmv_shared_kernel(const double* __restrict__ vals,
                                                           const int* __restrict__ cols, const double* __restrict__ x,
                                                           const double* __restrict__ sub, int sub_B,
                                                           const double* __restrict__ sub_scale,
                                                           const unsigned char* __restrict__ is_bc,
                                                           double* __restrict__ y, int n, int W, int Bp) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  FOR_EACH_NODE(nm, n, Bp, 1, {
    double acc = ell_row<false, kUni, kShared>(0.0, vals, cols, x, i, n, W, Bp, 1, nm.b);
    if (sub) acc -= (sub_scale ? sub_scale[nm.b] : 1.0) * sub[(i64)i * sub_B + (sub_B == 1 ? 0 : nm.b)];
    if (is_bc && is_bc[i]) acc = 0.0;
    y[(i64)i * Bp + nm.b] = acc;
  });
}

// ---------------------------------------------------------------------------------------
// Batched Jacobi-PCG
// ---------------------------------------------------------------------------------------
struct CgScalars {  // each (Bp) doubles, in `work` after the vectors and partials
  double *rz, *pAp, *alpha, *beta, *bb, *tol2, *rr;
  double* rs;             // per-sample power of two ~ 1 / |b| applied to the fp32 residual copies (NULL: none)
  double* xx;             // |x|^2 of the current iterate (AMG path: attainable-accuracy floor), may be NULL
  const double* maxdiag;  // per-sample (Bv entries) max diagonal entry, with xx
  int Bv;
  int* active;    // (Bp)
  int* iters;     // (Bp)
  int* n_active;  // (1)
};

__global__ __launch_bounds__(256) void cg_init_kernel(const double* __restrict__ vals, const double* __restrict__ bvec,
                                                       double* __restrict__ x, double* __restrict__ r,
                                                       double* __restrict__ z, double* __restrict__ p,
                                                       double* __restrict__ part_rz, double* __restrict__ part_bb, int n,
                                                       int Bp, int Bv) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const bool ok = nm.b < Bp;
  const int vb = Bv == 1 ? 0 : nm.b;
  double s_rz = 0.0, s_bb = 0.0;
  if (ok)
    for (int i = nm.node0; i < n; i += nm.stride) {
      const i64 o = (i64)i * Bp + nm.b;
      const double bi = bvec[o];
      const double zi = bi / vals[(i64)i * Bv + vb];  // slot 0 = diagonal
      x[o] = 0.0; r[o] = bi; z[o] = zi; p[o] = zi;
      s_rz += bi * zi;
      s_bb += bi * bi;
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double t_rz = block_sum_per_sample(s_rz, Bp, lds);
  const double t_bb = block_sum_per_sample(s_bb, Bp, lds);
  if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) {
    part_rz[(i64)blockIdx.x * Bp + nm.b] = t_rz;
    part_bb[(i64)blockIdx.x * Bp + nm.b] = t_bb;
  }
}

__global__ __launch_bounds__(256, 8) void cg_spmv_kernel(const double* __restrict__ vals, const int* __restrict__ cols,
                                                       const double* __restrict__ p, double* __restrict__ Ap,
                                                       double* __restrict__ part_pAp, int n, int W, int Bp, int Bv) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const bool ok = nm.b < Bp;
  double s = 0.0;
  if (ok)
    FOR_EACH_NODE(nm, n, Bp, Bv, {
      const i64 o = (i64)i * Bp + nm.b;
      const double ps = p[o];    // issued with the row's first loads
      const double acc = ell_row<false, kUni, kShared>(0.0, vals, cols, p, i, n, W, Bp, Bv, nm.b);
      Ap[o] = acc;
      s += acc * ps;
    });
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double t = block_sum_per_sample(s, Bp, lds);
  if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) part_pAp[(i64)blockIdx.x * Bp + nm.b] = t;
}

__global__ __launch_bounds__(256) void cg_update_kernel(const double* __restrict__ vals, const double* __restrict__ p,
                                                         const double* __restrict__ Ap, const double* __restrict__ alpha,
                                                         double* __restrict__ x, double* __restrict__ r,
                                                         double* __restrict__ z, double* __restrict__ part_rz,
                                                         double* __restrict__ part_rr, int n, int Bp, int Bv) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const bool ok = nm.b < Bp;
  const int vb = Bv == 1 ? 0 : nm.b;
  double s_rz = 0.0, s_rr = 0.0;
  if (ok) {
    const double a = alpha[nm.b];
    for (int i = nm.node0; i < n; i += nm.stride) {
      const i64 o = (i64)i * Bp + nm.b;
      x[o] += a * p[o];
      const double ri = r[o] - a * Ap[o];
      const double zi = ri / vals[(i64)i * Bv + vb];
      r[o] = ri; z[o] = zi;
      s_rz += ri * zi;
      s_rr += ri * ri;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double t_rz = block_sum_per_sample(s_rz, Bp, lds);
  const double t_rr = block_sum_per_sample(s_rr, Bp, lds);
  if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) {
    part_rz[(i64)blockIdx.x * Bp + nm.b] = t_rz;
    part_rr[(i64)blockIdx.x * Bp + nm.b] = t_rr;
  }
}

template <typename TZ>
__global__ __launch_bounds__(256) void cg_update_p_kernel(const TZ* __restrict__ z, const double* __restrict__ beta,
                                                           double* __restrict__ p, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  const double be = beta[nm.b];
  for (int i = nm.node0; i < n; i += nm.stride) {
    const i64 o = (i64)i * Bp + nm.b;
    p[o] = (double)z[o] + be * p[o];
  }
}

// true residual |b - A x|^2 partials
__global__ __launch_bounds__(256, 8) void residual_kernel(const double* __restrict__ vals, const int* __restrict__ cols,
                                                        const double* __restrict__ bvec, const double* __restrict__ x,
                                                        double* __restrict__ part, int n, int W, int Bp, int Bv) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const bool ok = nm.b < Bp;
  double s = 0.0;
  if (ok)
    FOR_EACH_NODE(nm, n, Bp, Bv, {
      const double acc = ell_row<true, kUni, kShared>(bvec[(i64)i * Bp + nm.b], vals, cols, x, i, n, W, Bp, Bv, nm.b);
      s += acc * acc;
    });
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double t = block_sum_per_sample(s, Bp, lds);
  if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) part[(i64)blockIdx.x * Bp + nm.b] = t;
}

// Sum the block partials of one quantity for sample b (fixed order: deterministic).
// Block = 4 waves: lanes over samples, waves over quarters of the partial list.
__device__ inline double sum_partials(const double* __restrict__ part, int nblk, int Bp, int b, double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double s = 0.0;
  if (b < Bp) {  // 4 independent chains keep several loads in flight (fixed order: still deterministic)
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = wave;
    for (; k + 12 < nblk; k += 16) {
      s0 += part[(i64)k * Bp + b];
      s1 += part[(i64)(k + 4) * Bp + b];
      s2 += part[(i64)(k + 8) * Bp + b];
      s3 += part[(i64)(k + 12) * Bp + b];
    }
    for (; k < nblk; k += 4) s0 += part[(i64)k * Bp + b];
    s = (s0 + s1) + (s2 + s3);
  }
  lds[wave * kWave + lane] = s;
  __syncthreads();
  const double t = (lds[lane] + lds[kWave + lane]) + (lds[2 * kWave + lane] + lds[3 * kWave + lane]);
  __syncthreads();
  return t;
}

// First stage of a long partial list (2048 rows on big meshes: ONE block of cg_scalar_kernel summing them took 68 us per
// phase at 512^2 x 64, three phases per iteration): block (x, y, z) sums the rows y, y + S, ... of list z for the samples
// of chunk x into row y of that list's slice table (S = kEllSlices rows); cg_scalar_kernel then sums S rows.  Fixed
// assignment and order of additions: bitwise reproducible (the lattice solver's pcg_slice_kernel, for two lists at once).
constexpr int kEllSlices = 16;
__global__ __launch_bounds__(256) void cg_slice_kernel(const double* __restrict__ partA, const double* __restrict__ partB,
                                                        int nblk, int Bp, double* __restrict__ slice) {
  __shared__ double lds[4 * kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * kWave + lane;
  const int S = gridDim.y, y = blockIdx.y;
  const double* __restrict__ part = blockIdx.z ? partB : partA;
  double s0 = 0.0, s1 = 0.0;
  if (b < Bp) {
    int k = y + S * wave;
    for (; k + 4 * S < nblk; k += 8 * S) {
      s0 += part[(i64)k * Bp + b];
      s1 += part[(i64)(k + 4 * S) * Bp + b];
    }
    if (k < nblk) s0 += part[(i64)k * Bp + b];
  }
  lds[wave * kWave + lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && b < Bp)
    slice[((i64)blockIdx.z * S + y) * Bp + b] = (lds[lane] + lds[kWave + lane]) + (lds[2 * kWave + lane] + lds[3 * kWave + lane]);
}

enum { PH_INIT = 0, PH_ALPHA = 1, PH_BETA = 2, PH_RELRES = 3, PH_XX = 4, PH_SCALE = 5 };

__global__ __launch_bounds__(256) void cg_scalar_kernel(int phase, const double* __restrict__ partA,
                                                         const double* __restrict__ partB, int nblk, int Bp,
                                                         double tol, CgScalars S, double* __restrict__ relres) {
  __shared__ double lds[4 * kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * kWave + lane;
  const double a = sum_partials(partA, nblk, Bp, b, lds);
  const double c = partB ? sum_partials(partB, nblk, Bp, b, lds) : 0.0;
  if (wave != 0 || b >= Bp) return;
  if (phase == PH_INIT) {  // a = r.z, c = b.b
    S.rz[b] = a;
    S.bb[b] = c;
    S.tol2[b] = tol * tol * c;
    S.active[b] = c > 0.0 ? 1 : 0;
    S.iters[b] = 0;
    S.alpha[b] = 0.0;
    S.beta[b] = 0.0;
  } else if (phase == PH_ALPHA) {  // a = p.Ap
    // scaled fp32 copies: z, p, Ap carry rs and both dots rs^2; the updates of x and r take alpha / rs
    S.alpha[b] = (S.active[b] && a > 0.0) ? (S.rz[b] / a) / (S.rs ? S.rs[b] : 1.0) : 0.0;
    if (b == 0) *S.n_active = 0;
  } else if (phase == PH_BETA) {  // a = r.z (new), c = r.r
    if (S.active[b]) {
      S.iters[b] += 1;
      S.rr[b] = c;
      double thr = S.tol2[b];
      if (S.xx) {  // fp64 cannot bring |b - A x| below ~ u |A| |x|: stop at half of that level (see diffhe_hip.h)
        const double fl = 0.5 * 1.1102230246251565e-16 * 2.0 * S.maxdiag[S.Bv == 1 ? 0 : b];
        const double floor2 = fl * fl * S.xx[b];
        if (floor2 > thr) thr = floor2;
      }
      if (c <= thr) {
        S.active[b] = 0;
        S.beta[b] = 0.0;
      } else {
        S.beta[b] = a / S.rz[b];
        S.rz[b] = a;
        atomicAdd(S.n_active, 1);
      }
    } else {
      S.beta[b] = 0.0;
    }
  } else if (phase == PH_XX) {  // a = x.x
    S.xx[b] = a;
  } else if (phase == PH_SCALE) {  // a = b.b: power of two rs with rs |b| in [1, 2) (see pcg_cvt_kernel in lattice.hip)
    S.rs[b] = a > 0.0 ? ldexp(1.0, -ilogb(sqrt(a))) : 1.0;
  } else {  // PH_RELRES: a = |b - A x|^2
    relres[b] = S.bb[b] > 0.0 ? sqrt(a / S.bb[b]) : 0.0;
  }
}

// ---------------------------------------------------------------------------------------
// dL/dkappa contraction
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_kappa_kernel(const int* __restrict__ elems, const double* __restrict__ k0,
                                                          const double* __restrict__ lam, const double* __restrict__ u,
                                                          const double* __restrict__ g, int npe, int m, int Bp,
                                                          double* __restrict__ dk_e,
                                                          double* __restrict__ dk_part) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);  // "nodes" are elements here
  const bool ok = nm.b < Bp;
  double s = 0.0;
  if (ok)
    for (int e = nm.node0; e < m; e += nm.stride) {
      double le[6], ue[6];   // npe <= 6 (P2 triangles)
      for (int p = 0; p < npe; ++p) {
        const int node = elems[(i64)p * m + e];
        const i64 o = (i64)node * Bp + nm.b;
        le[p] = lam[o];
        ue[p] = u[o] + (g ? g[node] : 0.0);  // full u: Dirichlet values included (Appendix A step 2)
      }
      double acc = 0.0;
      for (int p = 0; p < npe; ++p)
        for (int q = 0; q < npe; ++q) acc += le[p] * k0[(i64)(p * npe + q) * m + e] * ue[q];
      const double dk = -acc;
      if (dk_e) dk_e[(i64)e * Bp + nm.b] = dk;
      s += dk;
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double t = block_sum_per_sample(s, Bp, lds);
  if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) dk_part[(i64)blockIdx.x * Bp + nm.b] = t;
}

// The same contraction summed over the batch, for a kappa field SHARED by all samples (kappa (m,)):
//   dk[e] = sum_b dk[e, b].  One wave per element at a time; its lanes walk the sample chunks in a fixed order and
// meet in a fixed-order wave reduction, so the result is bitwise reproducible and the (m, Bp) per-sample gradient
// (4.3 GB at 1024^2 x 256) is never written.
__global__ __launch_bounds__(256) void grad_kappa_shared_kernel(const int* __restrict__ elems,
                                                                 const double* __restrict__ k0,
                                                                 const double* __restrict__ lam,
                                                                 const double* __restrict__ u,
                                                                 const double* __restrict__ g, int npe, int m, int B,
                                                                 int Bp, double* __restrict__ dk) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = blockIdx.x * 4 + wave; e < m; e += gridDim.x * 4) {
    int node[6];
    double gq[6], kk[36];
    for (int p = 0; p < npe; ++p) {
      node[p] = elems[(i64)p * m + e];
      gq[p] = g ? g[node[p]] : 0.0;
    }
    for (int pq = 0; pq < npe * npe; ++pq) kk[pq] = k0[(i64)pq * m + e];
    double s = 0.0;
    for (int b = lane; b < B; b += kWave) {   // padding samples (b >= B) carry no gradient
      double le[6], ue[6];
      for (int p = 0; p < npe; ++p) {
        const i64 o = (i64)node[p] * Bp + b;
        le[p] = lam[o];
        ue[p] = u[o] + gq[p];
      }
      double acc = 0.0;
      for (int p = 0; p < npe; ++p)
        for (int q = 0; q < npe; ++q) acc += le[p] * kk[p * npe + q] * ue[q];
      s -= acc;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) dk[e] = s;
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part, int nblk, int Bp,
                                                            double* __restrict__ out) {
  __shared__ double lds[4 * kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * kWave + lane;
  const double a = sum_partials(part, nblk, Bp, b, lds);
  if (wave == 0 && b < Bp) out[b] = a;
}

// ---------------------------------------------------------------------------------------
// (B, n) <-> (n, Bp) through a padded LDS tile
// ---------------------------------------------------------------------------------------
constexpr int kT = 64;

__global__ __launch_bounds__(256) void to_node_major_kernel(const double* __restrict__ src, i64 ld,
                                                             const unsigned char* __restrict__ zero_mask,
                                                             double* __restrict__ dst, int n, int B, int Bp) {
  __shared__ double tile[kT][kT + 1];
  const int i0 = blockIdx.x * kT, b0 = blockIdx.y * kT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the 16 rows of a wave are loaded into registers first (16 loads in flight; a load-store loop had one), then staged
  double v[kT / 4];
#pragma unroll
  for (int k = 0; k < kT / 4; ++k) {  // lanes along i: coalesced reads of a sample row
    const int b = b0 + wave + 4 * k, i = i0 + lane;
    v[k] = (b < B && i < n) ? src[(i64)b * ld + i] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < kT / 4; ++k) tile[wave + 4 * k][lane] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kT / 4; ++k) {  // lanes along b: coalesced writes of a node row
    const int ii = wave + 4 * k;
    const int i = i0 + ii, b = b0 + lane;
    if (i < n && b < Bp) {
      double w = tile[lane][ii];
      if (zero_mask && zero_mask[i]) w = 0.0;
      dst[(i64)i * Bp + b] = w;
    }
  }
}

__global__ __launch_bounds__(256) void to_sample_major_kernel(const double* __restrict__ src,
                                                               const double* __restrict__ add, double* __restrict__ dst,
                                                               i64 ld, int n, int B, int Bp) {
  __shared__ double tile[kT][kT + 1];
  const int i0 = blockIdx.x * kT, b0 = blockIdx.y * kT;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[kT / 4];
#pragma unroll
  for (int k = 0; k < kT / 4; ++k) {   // 16 node rows per wave, all requested before the first is staged
    const int i = i0 + wave + 4 * k, b = b0 + lane;
    v[k] = (i < n && b < Bp) ? src[(i64)i * Bp + b] + (add ? add[i] : 0.0) : 0.0;
  }
#pragma unroll
  for (int k = 0; k < kT / 4; ++k) tile[wave + 4 * k][lane] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kT / 4; ++k) {
    const int bb = wave + 4 * k;
    const int b = b0 + bb, i = i0 + lane;
    if (b < B && i < n) dst[(i64)b * ld + i] = tile[lane][bb];
  }
}

// ---------------------------------------------------------------------------------------
// Aggregation multigrid for the general path (diffhe/amg.py builds the batch-shared hierarchy)
// ---------------------------------------------------------------------------------------
// coarse values = P^T A P (per sample) from gather lists: plain sums of fine entries for piecewise-constant P
// (weights == NULL), weighted sums w_c = P_iI P_jJ for a smoothed P.  The lists and weights are batch-shared
// (wave-uniform loads), the fine values arrive as one contiguous row of samples per contribution.
__global__ __launch_bounds__(256) void ell_galerkin_kernel(const double* __restrict__ vals_f,
                                                            const int* __restrict__ ent_ptr,
                                                            const int* __restrict__ contrib,
                                                            const double* __restrict__ weights,
                                                            double* __restrict__ vals_c, int nc, int Wc, int Bv) {
  const NodeMap nm = node_map(Bv);
  if (nm.b >= Bv) return;
  for (int I = nm.node0; I < nc; I += nm.stride)
    for (int k = 0; k < Wc; ++k) {
      const i64 ent = (i64)k * nc + I;
      double v = 0.0;
      if (weights)
        for (int c = ent_ptr[ent]; c < ent_ptr[ent + 1]; ++c) v = fma(weights[c], vals_f[(i64)contrib[c] * Bv + nm.b], v);
      else
        for (int c = ent_ptr[ent]; c < ent_ptr[ent + 1]; ++c) v += vals_f[(i64)contrib[c] * Bv + nm.b];
      vals_c[ent * Bv + nm.b] = v;
    }
}

// weighted Jacobi: xout = xin + omega (b - A xin) / D (xin == NULL: from zero); optional partial of b.xout.
// TV = storage type of the cycle's vectors, TM = storage type of the matrix values (fp32 copies inside a
// single-precision preconditioner); arithmetic is fp64 in registers.
template <typename TV, typename TM>
__global__ __launch_bounds__(256, 8) void ell_jacobi_kernel(const TM* __restrict__ vals, const int* __restrict__ cols,
                                                          const TV* __restrict__ bvec, const TV* __restrict__ xin,
                                                          TV* __restrict__ xout, double omega,
                                                          double* __restrict__ part, int n, int W, int Bp, int Bv) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const bool ok = nm.b < Bp;
  const int vb = Bv == 1 ? 0 : nm.b;
  double s = 0.0;
  if (ok)
    FOR_EACH_NODE(nm, n, Bp, Bv, {
      const i64 o = (i64)i * Bp + nm.b;
      const double d = (double)vals[(i64)i * Bv + vb];
      const double bi = (double)bvec[o];
      double xo;
      if (xin) {
        const double xs = (double)xin[o];    // issued with the row's first loads, not behind its last product
        const double acc = ell_row<true, kUni, kShared>(bi, vals, cols, xin, i, n, W, Bp, Bv, nm.b);
        xo = xs + omega * acc / d;
      } else {
        xo = omega * bi / d;
      }
      xout[o] = (TV)xo;
      s += bi * xo;
    });
  if (part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double t = block_sum_per_sample(s, Bp, lds);
    if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) part[(i64)blockIdx.x * Bp + nm.b] = t;
  }
}

template <typename TV, typename TM>
__global__ __launch_bounds__(256, 8) void ell_residual_out_kernel(const TM* __restrict__ vals, const int* __restrict__ cols,
                                                                const TV* __restrict__ bvec, const TV* __restrict__ x,
                                                                TV* __restrict__ r, int n, int W, int Bp, int Bv) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  FOR_EACH_NODE(nm, n, Bp, Bv, {
    r[(i64)i * Bp + nm.b] = (TV)ell_row<true, kUni, kShared>((double)bvec[(i64)i * Bp + nm.b], vals, cols, x, i, n, W, Bp, Bv, nm.b);
  });
}

// rc = P^T r: rc[I] = sum over the members c of coarse node I (fixed order) of w_c r[member_c]; w == NULL: 1
// (piecewise-constant aggregation: the plain sum over the aggregate)
template <typename TV>
__global__ __launch_bounds__(256) void agg_restrict_kernel(const TV* __restrict__ r, const int* __restrict__ agg_ptr,
                                                            const int* __restrict__ members,
                                                            const double* __restrict__ w, TV* __restrict__ rc,
                                                            int nc, int Bp) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  if (Bp >= kWave) {
    // a wave = one coarse node: 64 member indices (and weights) per vector load, handed out by v_readlane, the gathers
    // of 8 members in flight together -- the plain loop below waited twice per member (33-40 us on levels of a few
    // hundred coarse nodes, whose rows of P^T have 30-40 entries).  Same members, same order, same operations.
    const int lane = threadIdx.x & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int I = (int)blockIdx.x * 4 + wave_u; I < nc; I += (int)gridDim.x * 4) {
      const int beg = agg_ptr[I], end = agg_ptr[I + 1];
      double s = 0.0;
      for (int c0 = beg; c0 < end; c0 += kWave) {
        const int nk = end - c0 < kWave ? end - c0 : kWave;
        const int cl = c0 + (lane < nk ? lane : 0);
        const int mv = members[cl];
        const double wv = w ? w[cl] : 1.0;
        for (int u0 = 0; u0 < nk; u0 += 8) {
          double ww[8];
          TV rv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int k = u0 + u < nk ? u0 + u : 0;
            const TV* __restrict__ rr = r + (i64)__builtin_amdgcn_readlane(mv, k) * Bp;
            rv[u] = rr[nm.b];
            ww[u] = w ? readlane_f64(wv, k) : 1.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (u0 + u < nk) {
              if (w) s = fma(ww[u], (double)rv[u], s);
              else s += (double)rv[u];
            }
        }
      }
      rc[(i64)I * Bp + nm.b] = (TV)s;
    }
    return;
  }
  for (int I = nm.node0; I < nc; I += nm.stride) {
    double s = 0.0;
    if (w)
      for (int c = agg_ptr[I]; c < agg_ptr[I + 1]; ++c) s = fma(w[c], (double)r[(i64)members[c] * Bp + nm.b], s);
    else
      for (int c = agg_ptr[I]; c < agg_ptr[I + 1]; ++c) s += (double)r[(i64)members[c] * Bp + nm.b];
    rc[(i64)I * Bp + nm.b] = (TV)s;
  }
}

// x += scale * P e for a smoothed prolongation stored as ELL rows: p_cols / p_vals (pw, n), -1 = no entry
template <typename TV>
__global__ __launch_bounds__(256) void sa_prolong_add_kernel(const TV* __restrict__ e, const int* __restrict__ p_cols,
                                                              const double* __restrict__ p_vals, int pw,
                                                              TV* __restrict__ x, double scale, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  if (Bp >= kWave && pw <= 8) {
    // a wave = one fine node: its row of P in one vector load (lane k: entry k), entries handed out by v_readlane, the
    // gathers of e in flight together with the node's own x.  Same entries, same order, same operations.
    const int lane = threadIdx.x & 63;
    int first, hi, step;
    wave_node_range(n, first, hi, step);
    for (int i = first; i < hi; i += step) {
      const i64 entl = (i64)(lane < pw ? lane : 0) * n + i;
      const int cv = p_cols[entl];
      const double pv = p_vals[entl];
      const i64 o = (i64)i * Bp + nm.b;
      const TV xs = x[o];
      TV ev[8];
      double pp[8];
      int II[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        II[u] = u < pw ? __builtin_amdgcn_readlane(cv, u) : -1;
        pp[u] = readlane_f64(pv, u);
        ev[u] = (TV)0;
        if (II[u] >= 0) {    // wave-uniform: no load for an absent entry
          const TV* __restrict__ er = e + (i64)II[u] * Bp;
          ev[u] = er[nm.b];
        }
      }
      double s = 0.0;
      bool any = false;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (II[u] >= 0) {
          s = fma(pp[u], (double)ev[u], s);
          any = true;
        }
      if (any) x[o] = (TV)((double)xs + scale * s);
    }
    return;
  }
  for (int i = nm.node0; i < n; i += nm.stride) {
    double s = 0.0;
    bool any = false;
    for (int k = 0; k < pw; ++k) {
      const int I = p_cols[(i64)k * n + i];
      if (I >= 0) {
        s = fma(p_vals[(i64)k * n + i], (double)e[(i64)I * Bp + nm.b], s);
        any = true;
      }
    }
    if (any) x[(i64)i * Bp + nm.b] = (TV)((double)x[(i64)i * Bp + nm.b] + scale * s);
  }
}

// x[i] += scale * e[agg[i]]
template <typename TV>
__global__ __launch_bounds__(256) void agg_prolong_add_kernel(const TV* __restrict__ e, const int* __restrict__ agg,
                                                               TV* __restrict__ x, double scale, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  for (int i = nm.node0; i < n; i += nm.stride) {
    const int I = agg[i];
    if (I >= 0) x[(i64)i * Bp + nm.b] = (TV)((double)x[(i64)i * Bp + nm.b] + scale * (double)e[(i64)I * Bp + nm.b]);
  }
}

// x += alpha p ; r -= alpha Ap ; partial r.r
__global__ __launch_bounds__(256) void amg_update_kernel(const double* __restrict__ p, const double* __restrict__ Ap,
                                                          const double* __restrict__ alpha, double* __restrict__ x,
                                                          double* __restrict__ r, float* __restrict__ r32,
                                                          const double* __restrict__ rs, double* __restrict__ part_rr,
                                                          double* __restrict__ part_xx, int n, int Bp) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const bool ok = nm.b < Bp;
  double s = 0.0, sx = 0.0;
  if (ok) {
    const double a = alpha[nm.b];
    const double sc = (r32 && rs) ? rs[nm.b] : 1.0;
    for (int i = nm.node0; i < n; i += nm.stride) {
      const i64 o = (i64)i * Bp + nm.b;
      const double xi = x[o] + a * p[o];
      x[o] = xi;
      const double ri = r[o] - a * Ap[o];
      r[o] = ri;
      if (r32) r32[o] = (float)(ri * sc);
      s += ri * ri;
      sx += xi * xi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double t = block_sum_per_sample(s, Bp, lds);
  const double tx = part_xx ? block_sum_per_sample(sx, Bp, lds) : 0.0;
  if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) {
    part_rr[(i64)blockIdx.x * Bp + nm.b] = t;
    if (part_xx) part_xx[(i64)blockIdx.x * Bp + nm.b] = tx;
  }
}

// Per-sample max of the ELL diagonal (slot 0), as the bit pattern of a non-negative double (atomicMax: deterministic)
__global__ __launch_bounds__(256) void ell_maxdiag_kernel(const double* __restrict__ vals, int n, int Bv,
                                                           unsigned long long* __restrict__ out) {
  const NodeMap nm = node_map(Bv);
  double m = 0.0;
  if (nm.b < Bv)
    for (int i = nm.node0; i < n; i += nm.stride) {
      const double d = vals[(i64)i * Bv + nm.b];
      m = d > m ? d : m;
    }
  const int LB = Bv < kWave ? Bv : kWave;
  for (int off = LB; off < kWave; off <<= 1) {
    const double o = __shfl_xor(m, off);
    m = o > m ? o : m;
  }
  if ((int)(threadIdx.x & 63) < LB && nm.b < Bv) atomicMax(out + nm.b, (unsigned long long)__double_as_longlong(m));
}

// x = 0 ; r = b ; partial b.b
__global__ __launch_bounds__(256) void amg_init_kernel(const double* __restrict__ bvec, double* __restrict__ x,
                                                        double* __restrict__ r, float* __restrict__ r32,
                                                        double* __restrict__ p, double* __restrict__ part_bb, int n,
                                                        int Bp) {
  __shared__ double lds[4 * kWave];
  const NodeMap nm = node_map(Bp);
  const bool ok = nm.b < Bp;
  double s = 0.0;
  if (ok)
    for (int i = nm.node0; i < n; i += nm.stride) {
      const i64 o = (i64)i * Bp + nm.b;
      const double bi = bvec[o];
      x[o] = 0.0; r[o] = bi; p[o] = 0.0;
      s += bi * bi;
    }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double t = block_sum_per_sample(s, Bp, lds);
  if (wave == 0 && lane < (Bp < kWave ? Bp : kWave) && ok) part_bb[(i64)blockIdx.x * Bp + nm.b] = t;
}

__global__ __launch_bounds__(256) void amg_cvt_kernel(const double* __restrict__ r, const double* __restrict__ rs,
                                                       float* __restrict__ r32, int n, int Bp) {
  const NodeMap nm = node_map(Bp);
  if (nm.b >= Bp) return;
  const double sc = rs[nm.b];
  for (int i = nm.node0; i < n; i += nm.stride) r32[(i64)i * Bp + nm.b] = (float)(r[(i64)i * Bp + nm.b] * sc);
}

constexpr int kAmgMaxLevels = 16;

struct AmgHier {
  diffhe_amg_level lev[kAmgMaxLevels];
  int nl, Bv, Bp, n_coarse, gamma;
  double w0, w1, scale;
  void *xa[kAmgMaxLevels], *xb[kAmgMaxLevels], *res[kAmgMaxLevels], *rhs[kAmgMaxLevels];  // TV vectors
};

// ---------------------------------------------------------------------------------------
// Wave-per-node sweep kernels of the solvers (batches of >= 64), SOFTWARE-PIPELINED: with ell_row_uniform inside a plain
// node loop a wave still paid two dependent memory latencies per node (row meta data -> gathers) for each of its 32
// nodes, which is what the fine-level sweep's 72-87 us were (32 x 2 x ~1.2 us).  Here the next node's meta data
// (column indices, shared values, b_i, own x_i) are requested right behind the current node's gathers, so a node costs
// ONE exposed latency.  Same entries, order and operations as ell_jacobi_kernel / ell_residual_out_kernel /
// cg_spmv_kernel; same node -> (block, wave) assignment, so the block partials are the same sums.
//   W_JACOBI: out = x + omega (b - A x) / D, partial of b.out;  W_RESID: out = b - A x;  W_SPMV: out = A x, partial x.out
// ---------------------------------------------------------------------------------------
enum { W_JACOBI = 0, W_RESID = 1, W_SPMV = 2 };
template <int OP, typename TV, typename TM, bool SHARED>
__global__ __launch_bounds__(256, 8) void ellw_kernel(const TM* __restrict__ vals, const int* __restrict__ cols,
                                                      const TV* __restrict__ bvec, const TV* __restrict__ xin,
                                                      TV* __restrict__ out, double omega, double* __restrict__ part,
                                                      int n, int W, int Bp) {
  __shared__ double lds[4 * kWave];
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y * kWave + lane;
  int i, hi, step;
  wave_node_range(n, i, hi, step);
  constexpr int NU = SHARED ? 8 : 4;   // per-sample values are vector loads of their own: shorter chunks fit 64 VGPRs
  const int nk0 = W < kWave ? W : kWave;           // entries of the first (normally the only) 64-entry chunk
  const i64 lane_ent = (i64)(lane < nk0 ? lane : 0) * n;
  double s = 0.0;
  int cv = 0;
  double av = 0.0;
  TV bi = (TV)0, xs = (TV)0;
  if (i < hi) {
    cv = cols[lane_ent + i];
    if (SHARED) av = (double)vals[lane_ent + i];
    if (OP != W_SPMV) bi = bvec[(i64)i * Bp + b];
    if (OP != W_RESID) xs = xin[(i64)i * Bp + b];
  }
  while (i < hi) {
    const int inext = i + step;
    double acc = OP == W_SPMV ? 0.0 : (double)bi;
    double d = 1.0;
    int cvn = 0;
    double avn = 0.0;
    TV bin = (TV)0, xsn = (TV)0;
    for (int k0 = 0; k0 < W; k0 += kWave) {
      const int nk = W - k0 < kWave ? W - k0 : kWave;
      if (k0 > 0) {   // rows wider than 64 entries: not pipelined
        const i64 e2 = (i64)(k0 + (lane < nk ? lane : 0)) * n + i;
        cv = cols[e2];
        if (SHARED) av = (double)vals[e2];
      }
      for (int u0 = 0; u0 < nk; u0 += NU) {
        int c[NU];
        double a[NU];
        TV xv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const bool in = u0 + u < nk;
          const int k = in ? u0 + u : 0;
          c[u] = __builtin_amdgcn_readlane(cv, k);
          if (SHARED) a[u] = in ? readlane_f64(av, k) : 0.0;
        }
        // (no branches around the loads: a wave-uniform `if` per entry made the compiler wait after every gather;
        // an absent entry re-reads entry 0 and gets the value 0)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const TV* __restrict__ xr = xin + (i64)c[u] * Bp;
          xv[u] = xr[b];
        }
        if (!SHARED) {
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const bool in = u0 + u < nk;
            const TM* __restrict__ vr = vals + ((i64)(k0 + (in ? u0 + u : 0)) * n + i) * Bp;
            const double t = (double)vr[b];
            a[u] = in ? t : 0.0;
          }
        }
        if (k0 == 0 && u0 == 0 && inext < hi) {   // the next node's meta data, behind this node's gathers
          cvn = cols[lane_ent + inext];
          if (SHARED) avn = (double)vals[lane_ent + inext];
          if (OP != W_SPMV) bin = bvec[(i64)inext * Bp + b];
          if (OP != W_RESID) xsn = xin[(i64)inext * Bp + b];
        }
        if (OP == W_JACOBI && k0 == 0 && u0 == 0) d = a[0];   // entry 0 of a row is its diagonal
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (OP == W_SPMV) acc += a[u] * (double)xv[u];
          else acc -= a[u] * (double)xv[u];
        }
      }
    }
    const i64 o = (i64)i * Bp + b;
    if (OP == W_JACOBI) {
      const double xo = (double)xs + omega * acc / d;
      out[o] = (TV)xo;
      s += (double)bi * xo;
    } else if (OP == W_RESID) {
      out[o] = (TV)acc;
    } else {
      out[o] = (TV)acc;
      s += acc * (double)xs;
    }
    i = inext;
    cv = cvn; av = avn; bi = bin; xs = xsn;
  }
  if (OP != W_RESID && part) {
    const int wave = threadIdx.x >> 6;
    const double t = block_sum_per_sample(s, Bp, lds);
    if (wave == 0) part[(i64)blockIdx.x * Bp + b] = t;
  }
}
// (A two-samples-per-lane form of ellw_kernel for batch-shared matrices and batches of 128 k -- 8 / 16-byte gathers, the
// node's scalar work paid once per 128 samples -- was built and measured: bitwise the same values, 117.0 -> 116.0 ms per
// solve at jittered 512^2 x 256, gpurun_out/r4bn.  The sweeps are not instruction-bound; removed.)
// The first sweep of a cycle starts from zero: x = omega b / D, an elementwise pass -- four nodes per trip (one node per
// trip left a wave with a single load outstanding: 55 us for 134 MB on the fine level of 512^2 x 64).
template <typename TV, typename TM>
__global__ __launch_bounds__(256) void ellw_jacobi0_kernel(const TM* __restrict__ vals, const TV* __restrict__ bvec,
                                                           TV* __restrict__ xout, double omega, int n, int Bp, int Bv) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.y * kWave + lane;
  const int vb = Bv == 1 ? 0 : b;
  int i, hi, step;
  wave_node_range(n, i, hi, step);
  for (; (i64)i + 3LL * step < hi; i += 4 * step) {
    TV bv[4];
    double d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bv[u] = bvec[(i64)(i + u * step) * Bp + b];
      d[u] = (double)vals[(i64)(i + u * step) * Bv + vb];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) xout[(i64)(i + u * step) * Bp + b] = (TV)(omega * (double)bv[u] / d[u]);
  }
  for (; i < hi; i += step)
    xout[(i64)i * Bp + b] = (TV)(omega * (double)bvec[(i64)i * Bp + b] / (double)vals[(i64)i * Bv + vb]);
}
template <typename TV, typename TM>
int launch_ellw_jacobi0(const TM* vals, const TV* bvec, TV* xout, double omega, int n, int Bp, int Bv, hipStream_t st) {
  const char* env = getenv("DIFFHE_ELL_PIPE");   // read per launch: a test compares both forms in one process
  const int on = env ? atoi(env) : 1;
  if (!on || Bp < kWave || Bp % kWave) return 0;
  hipLaunchKernelGGL((ellw_jacobi0_kernel<TV, TM>), diffhe::node_grid(n, Bp), dim3(256), 0, st, vals, bvec, xout, omega, n, Bp, Bv);
  return 1;
}
// launch helper: 1 = launched (batch >= 64, DIFFHE_ELL_PIPE != 0), 0 = caller takes the plain kernel
template <int OP, typename TV, typename TM>
int launch_ellw(const TM* vals, const int* cols, const TV* bvec, const TV* xin, TV* out, double omega, double* part, int n,
                int W, int Bp, int Bv, hipStream_t st) {
  const char* env = getenv("DIFFHE_ELL_PIPE");   // read per launch: a test compares both forms in one process
  const int on = env ? atoi(env) : 1;
  if (!on || Bp < kWave || Bp % kWave) return 0;
  const dim3 grid = diffhe::node_grid(n, Bp);
  if (Bv == 1)
    hipLaunchKernelGGL((ellw_kernel<OP, TV, TM, true>), grid, dim3(256), 0, st, vals, cols, bvec, xin, out, omega, part, n, W, Bp);
  else
    hipLaunchKernelGGL((ellw_kernel<OP, TV, TM, false>), grid, dim3(256), 0, st, vals, cols, bvec, xin, out, omega, part, n, W, Bp);
  return 1;
}

#define ALAUNCH(kernel, n_, ...) \
  hipLaunchKernelGGL(kernel, diffhe::node_grid((n_), H.Bp), dim3(256), 0, st, __VA_ARGS__)

// Last level of a batch-shared hierarchy: x = A^-1 rhs as ONE dense product with the cached inverse (n <= 128; the 16
// Jacobi sweeps it replaces were 16 launch-bound launches per cycle and only an approximate solve).  A block = 64
// samples x 4 rows (one per wave): rhs staged in LDS, a row of the inverse is one vector load.
template <typename TV>
__global__ __launch_bounds__(256) void amg_dense_solve_kernel(const double* __restrict__ inv, const TV* __restrict__ rhs,
                                                               TV* __restrict__ x, int n, int Bp) {
  extern __shared__ double sm[];   // (n, 64)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int b = blockIdx.x * kWave + lane;
  const bool ok = b < Bp;
  for (int j = wave; j < n; j += 4) sm[j * kWave + lane] = ok ? (double)rhs[(i64)j * Bp + b] : 0.0;
  __syncthreads();
  const int i = (int)blockIdx.y * 4 + wave;   // one row per wave, four rows per block: the level spreads over n / 4 CUs
  if (i >= n) return;
  const double* __restrict__ row = inv + (i64)i * n;
  double acc = 0.0;
  for (int j0 = 0; j0 < n; j0 += kWave) {   // 64 entries of the row per vector load, handed out by v_readlane
    const int nj = n - j0 < kWave ? n - j0 : kWave;
    const double rv = row[j0 + (lane < nj ? lane : 0)];
    for (int j = 0; j < nj; j += 8) {   // 8 LDS reads in flight (entries beyond nj: coefficient 0)
      double sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) sv[u] = sm[(j0 + (j + u < nj ? j + u : 0)) * kWave + lane];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fma(j + u < nj ? readlane_f64(rv, j + u) : 0.0, sv[u], acc);
    }
  }
  if (ok) x[(i64)i * Bp + b] = (TV)acc;
}

// x ~= A_l^{-1} rhs from a zero guess: V(2,2) weighted Jacobi, `gamma` coarse corrections per level
// (gamma = 2: W-cycle -- affordable because aggregation coarsens by ~10x -- compensates the weak
// piecewise-constant interpolation).  Returns the buffer holding the result.
template <typename TV>
TV* amg_cycle(const AmgHier& H, int l, const TV* rhs, double* rz_part, hipStream_t st) {
  const diffhe_amg_level& L = H.lev[l];
  TV* a = (TV*)H.xa[l];
  TV* b2 = (TV*)H.xb[l];
  const bool last = (l == H.nl - 1);
  if (last && l > 0 && L.dense_inv && H.Bv == 1 && L.n <= 128) {
    hipLaunchKernelGGL(amg_dense_solve_kernel<TV>, dim3((H.Bp + kWave - 1) / kWave, (L.n + 3) / 4), dim3(256),
                       sizeof(double) * L.n * kWave, st, L.dense_inv, rhs, a, L.n, H.Bp);
    return a;
  }
  const int pre = last ? H.n_coarse : 2;
  // per-sample matrices inside the fp32 cycle read the fp32 copy of the values
  const bool m32 = sizeof(TV) == 4 && H.Bv != 1 && L.vals32 != nullptr;
#define AMG_JACOBI(xin_, xout_, w_, part_)                                                                          \
  do {                                                                                                              \
    if ((xin_) != nullptr &&                                                                                        \
        (m32 ? launch_ellw<W_JACOBI, TV, float>(L.vals32, L.cols, rhs, xin_, xout_, w_, part_, L.n, L.W, H.Bp, H.Bv, st) \
             : launch_ellw<W_JACOBI, TV, double>(L.vals, L.cols, rhs, xin_, xout_, w_, part_, L.n, L.W, H.Bp, H.Bv, st))) \
      break;                                                                                                        \
    if ((xin_) == nullptr && (part_) == nullptr &&                                                                  \
        (m32 ? launch_ellw_jacobi0<TV, float>(L.vals32, rhs, xout_, w_, L.n, H.Bp, H.Bv, st)                         \
             : launch_ellw_jacobi0<TV, double>(L.vals, rhs, xout_, w_, L.n, H.Bp, H.Bv, st)))                        \
      break;                                                                                                        \
    if (m32)                                                                                                        \
      ALAUNCH((ell_jacobi_kernel<TV, float>), L.n, L.vals32, L.cols, rhs, xin_, xout_, w_, part_, L.n, L.W, H.Bp, H.Bv); \
    else                                                                                                            \
      ALAUNCH((ell_jacobi_kernel<TV, double>), L.n, L.vals, L.cols, rhs, xin_, xout_, w_, part_, L.n, L.W, H.Bp, H.Bv);  \
  } while (0)
  for (int s = 0; s < pre; ++s) {
    const double w = (s & 1) ? H.w1 : H.w0;
    const bool fin = last && s == pre - 1;
    if (s == 0) {
      AMG_JACOBI((const TV*)nullptr, a, w, fin ? rz_part : (double*)nullptr);
    } else {
      AMG_JACOBI((const TV*)a, b2, w, fin ? rz_part : (double*)nullptr);
      TV* t = a; a = b2; b2 = t;
    }
  }
  if (last) return a;
  const diffhe_amg_level& C = H.lev[l + 1];
  const int cycles = (l + 1 == H.nl - 1) ? 1 : H.gamma;  // the last level is "solved": one visit is enough
  for (int g = 0; g < cycles; ++g) {
    if (m32 ? launch_ellw<W_RESID, TV, float>(L.vals32, L.cols, rhs, (const TV*)a, (TV*)H.res[l], 0.0, nullptr, L.n, L.W,
                                              H.Bp, H.Bv, st)
            : launch_ellw<W_RESID, TV, double>(L.vals, L.cols, rhs, (const TV*)a, (TV*)H.res[l], 0.0, nullptr, L.n, L.W,
                                               H.Bp, H.Bv, st)) {
    } else if (m32)
      ALAUNCH((ell_residual_out_kernel<TV, float>), L.n, L.vals32, L.cols, rhs, (const TV*)a, (TV*)H.res[l], L.n, L.W,
              H.Bp, H.Bv);
    else
      ALAUNCH((ell_residual_out_kernel<TV, double>), L.n, L.vals, L.cols, rhs, (const TV*)a, (TV*)H.res[l], L.n, L.W,
              H.Bp, H.Bv);
    ALAUNCH(agg_restrict_kernel<TV>, C.n, (const TV*)H.res[l], L.agg_ptr, L.agg_members, L.agg_weights,
            (TV*)H.rhs[l + 1], C.n, H.Bp);
    const TV* ec = amg_cycle<TV>(H, l + 1, (const TV*)H.rhs[l + 1], nullptr, st);
    if (L.p_cols)   // smoothed aggregation: P as ELL rows
      ALAUNCH(sa_prolong_add_kernel<TV>, L.n, ec, L.p_cols, L.p_vals, L.p_width, a, H.scale, L.n, H.Bp);
    else
      ALAUNCH(agg_prolong_add_kernel<TV>, L.n, ec, L.agg, a, H.scale, L.n, H.Bp);
  }
  for (int s = 0; s < 2; ++s) {
    const double w = (s & 1) ? H.w0 : H.w1;  // reverse order: symmetric cycle
    AMG_JACOBI((const TV*)a, b2, w, (l == 0 && s == 1) ? rz_part : (double*)nullptr);
    TV* t = a; a = b2; b2 = t;
  }
#undef AMG_JACOBI
  return a;
}

long long amg_carve(AmgHier& H, double* work) {
  long long off = 0;
  auto take = [&](long long cnt) { double* q = work ? work + off : nullptr; off += (cnt + 7) & ~7LL; return q; };
  for (int l = 0; l < H.nl; ++l) {
    const long long nb = (long long)H.lev[l].n * H.Bp;
    H.xa[l] = take(nb);
    H.xb[l] = take(nb);
    H.res[l] = take(nb);
    H.rhs[l] = take(nb);  // level 0: the fp32 copy of the CG residual (fp32 cycle)
  }
  return off;
}

int amg_fill(AmgHier& H, const diffhe_amg_level* levels, int n_levels, int Bv, int Bp) {
  if (!levels || n_levels < 1 || n_levels > kAmgMaxLevels) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  if (Bv != 1 && Bv != Bp) return DIFFHE_E_BADARG;
  for (int l = 0; l < n_levels; ++l) {
    const diffhe_amg_level& s = levels[l];
    if (s.n < 1 || s.W < 1 || !s.vals || !s.cols) return DIFFHE_E_BADARG;
    if (l < n_levels - 1 && (!s.agg || !s.agg_ptr || !s.agg_members)) return DIFFHE_E_BADARG;
    if (s.p_cols && (!s.p_vals || !s.agg_weights || s.p_width < 1)) return DIFFHE_E_BADARG;
    H.lev[l] = s;
  }
  H.nl = n_levels; H.Bv = Bv; H.Bp = Bp;
  return DIFFHE_OK;
}

inline int cg_blocks(int n, int Bp) { return (int)node_grid(n, Bp).x; }

}  // namespace

// DIFFHE_ELL_XCD=0 switches the XCD-contiguous node ranges off (A/B); copied to the device once per process
static int sync_xcd_switch(hipStream_t st) {
  static int done = 0;
  if (done) return DIFFHE_OK;
  const int on = getenv("DIFFHE_ELL_XCD") ? atoi(getenv("DIFFHE_ELL_XCD")) : 1;
  if (!on) {
    const int rc = diffhe::check(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_xcd_ranges), &on, sizeof(int), 0, hipMemcpyHostToDevice, st));
    if (rc) return rc;
    const int rc2 = diffhe::check(hipStreamSynchronize(st));
    if (rc2) return rc2;
  }
  done = 1;
  return DIFFHE_OK;
}

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" int diffhe_p1_element_integrals(const double* coords, const int* elems, int dim, int n, int m,
                                           double* k0, double* m0, void* stream) {
  if (!coords || !elems || !k0 || !m0 || (dim != 1 && dim != 2) || n < 1 || m < 1) return DIFFHE_E_BADARG;
  int blocks = (m + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(element_integrals_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, coords, elems, dim, n,
                     m, k0, m0);
  return diffhe::check_launch();
}

extern "C" int diffhe_ell_assemble_rows(const double* local, const double* kappa, long long kappa_se,
                                        long long kappa_sb, const int* ent_ptr, const int* contrib, const int* cols,
                                        const int* store_slot, const unsigned char* is_bc, const double* g,
                                        double* vals, double* lift, int n, int m, int W, int Bv, void* stream) {
  if (!local || !ent_ptr || !contrib || !cols || !vals || n < 1 || m < 1 || W < 1) return DIFFHE_E_BADARG;
  if (is_bc && !g) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bv)) return DIFFHE_E_BATCHPAD;
  diffhe::account(8.0 * Bv * ((double)W * n + (lift ? n : 0) + ((kappa && kappa_se) ? m : 0)));  // stored values, lift, kappa field
  hipLaunchKernelGGL(assemble_rows_kernel<false>, diffhe::node_grid(n, Bv), dim3(256), 0, (hipStream_t)stream, local,
                     (const double*)nullptr, kappa, kappa_se, kappa_sb, ent_ptr, contrib, cols, store_slot, is_bc, g,
                     vals, lift, n, m, W, Bv);
  return diffhe::check_launch();
}

extern "C" int diffhe_lattice_assemble_rows(const double* local, int local_compact, const double* kappa,
                                            long long kappa_se, long long kappa_sb, const unsigned char* is_bc,
                                            const double* g, double* vals, double* lift, int nx, int ny, int nd, int Bv,
                                            void* stream) {
  if (!local || !vals || nx < 1 || ny < 1 || nd < 3 || nd > 4) return DIFFHE_E_BADARG;
  if (is_bc && !g) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bv)) return DIFFHE_E_BATCHPAD;
  const long long n = (long long)(nx + 1) * (ny + 1), m = 2LL * nx * ny;
  if (n > 2147483647LL) return DIFFHE_E_BADARG;
  diffhe::account(8.0 * Bv * ((double)nd * n + (lift ? n : 0) + ((kappa && kappa_se) ? m : 0)));
  const int strip_on = getenv("DIFFHE_ASM_STRIP") ? atoi(getenv("DIFFHE_ASM_STRIP")) : 1;
  if (strip_on && kappa && kappa_se && Bv >= kWave && Bv % kWave == 0 && nx >= 128 && ny >= 64) {
    // per-sample kappa fields on a strip-sized level: every kappa_e loaded once per wave and quad row
    const int TR = getenv("DIFFHE_ASM_TR") ? atoi(getenv("DIFFHE_ASM_TR")) : 8;
    const int ncb = (nx + 1 + 4 * kAsmCols - 1) / (4 * kAsmCols), nrc = (ny + 1 + TR - 1) / TR;
    hipLaunchKernelGGL(lattice_assemble_strip_kernel, dim3(ncb * nrc, Bv / kWave), dim3(256), 0, (hipStream_t)stream, local,
                       (i64)(local_compact ? 2 : m), (i64)(local_compact ? 1 : -1), kappa, (i64)kappa_se, (i64)kappa_sb,
                       is_bc, g, vals, lift, nx, ny, nd, Bv, ncb, TR);
    return diffhe::check_launch();
  }
  hipLaunchKernelGGL(lattice_assemble_kernel, diffhe::node_grid((int)n, Bv), dim3(256), 0, (hipStream_t)stream, local,
                     (i64)(local_compact ? 2 : m), (i64)(local_compact ? 1 : -1), kappa, kappa_se, kappa_sb, is_bc, g, vals,
                     lift, nx, ny, nd, Bv);
  return diffhe::check_launch();
}

extern "C" int diffhe_ell_assemble_rows_ref(const double* tnum, const double* den, const double* kappa,
                                            long long kappa_se, long long kappa_sb, const int* ent_ptr,
                                            const int* contrib, const int* cols, const int* store_slot,
                                            const unsigned char* is_bc, const double* g, double* vals, double* lift,
                                            int n, int m, int W, int Bv, void* stream) {
  if (!tnum || !den || !ent_ptr || !contrib || !cols || !vals || n < 1 || m < 1 || W < 1) return DIFFHE_E_BADARG;
  if (is_bc && !g) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bv)) return DIFFHE_E_BATCHPAD;
  diffhe::account(8.0 * Bv * ((double)W * n + (lift ? n : 0) + ((kappa && kappa_se) ? m : 0)));
  hipLaunchKernelGGL(assemble_rows_kernel<true>, diffhe::node_grid(n, Bv), dim3(256), 0, (hipStream_t)stream, tnum, den,
                     kappa, kappa_se, kappa_sb, ent_ptr, contrib, cols, store_slot, is_bc, g, vals, lift, n, m, W, Bv);
  return diffhe::check_launch();
}

extern "C" int diffhe_ell_assemble_atomic(const double* coords, const int* elems, int dim, const double* kappa,
                                          long long kappa_se, long long kappa_sb, const int* slot_of, double* vals,
                                          int n, int m, int W, int Bp, void* stream) {
  (void)W;
  if (!coords || !elems || !slot_of || !vals || (dim != 1 && dim != 2) || n < 1 || m < 1) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  int gx = (m + kElemTile - 1) / kElemTile;
  if (gx > 4096) gx = 4096;
  dim3 grid(gx, (Bp + 63) / 64);
  hipLaunchKernelGGL(assemble_atomic_kernel, grid, dim3(256), 0, (hipStream_t)stream, coords, elems, dim, kappa,
                     kappa_se, kappa_sb, slot_of, vals, n, m, Bp);
  return diffhe::check_launch();
}

extern "C" int diffhe_ell_apply_dirichlet(const int* cols, const unsigned char* is_bc, const double* g, double* vals,
                                          double* F, int n, int W, int Bp, void* stream) {
  if (!cols || !is_bc || !g || !vals || n < 1 || W < 1) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  hipLaunchKernelGGL(apply_dirichlet_kernel, diffhe::node_grid(n, Bp), dim3(256), 0, (hipStream_t)stream, cols, is_bc,
                     g, vals, F, n, W, Bp);
  return diffhe::check_launch();
}

extern "C" int diffhe_ell_spmv_shared(const double* vals, const int* cols, const double* x, const double* sub,
                                      int sub_B, const double* sub_scale, const unsigned char* is_bc, double* y,
                                      int n, int W, int Bp, void* stream) {
  if (!vals || !cols || !x || !y || n < 1 || W < 1) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  if (sub && sub_B != 1 && sub_B != Bp) return DIFFHE_E_BADARG;
  diffhe::account(16.0 * n * Bp);
  hipLaunchKernelGGL(spmv_shared_kernel, diffhe::node_grid(n, Bp), dim3(256), 0, (hipStream_t)stream, vals, cols, x,
                     sub, sub_B, sub_scale, is_bc, y, n, W, Bp);
  return diffhe::check_launch();
}

// per-sample scalar phase of the CG loops below: long partial lists go through cg_slice_kernel first
#define ELL_SCALAR(phase_, pa_, pb_)                                                                                      \
  do {                                                                                                                    \
    const double* pa__ = (pa_);                                                                                           \
    const double* pb__ = (pb_);                                                                                           \
    if (two_stage && nblk >= 256) {                                                                                       \
      hipLaunchKernelGGL(cg_slice_kernel, dim3(sgrid.x, kEllSlices, pb__ ? 2 : 1), dim3(256), 0, st, pa__, pb__, nblk, Bp,  \
                         slices);                                                                                         \
      hipLaunchKernelGGL(cg_scalar_kernel, sgrid, dim3(256), 0, st, (int)(phase_), (const double*)slices,                 \
                         pb__ ? (const double*)(slices + (long long)kEllSlices * Bp) : (const double*)nullptr,            \
                         kEllSlices, Bp, tol, S, relres);                                                                 \
    } else {                                                                                                              \
      hipLaunchKernelGGL(cg_scalar_kernel, sgrid, dim3(256), 0, st, (int)(phase_), pa__, pb__, nblk, Bp, tol, S, relres); \
    }                                                                                                                     \
  } while (0)

extern "C" long long diffhe_cg_workspace_doubles(int n, int Bp) {
  const long long nblk = cg_blocks(n, Bp);
  return 4LL * n * Bp + 3LL * nblk * Bp + (16LL + 2 * kEllSlices) * Bp + 64;
}

extern "C" int diffhe_ell_cg_solve(const double* vals, const int* cols, const double* b, double* x, int n, int W,
                                   int Bp, int Bv, double tol, int max_iter, int check_every, double* work,
                                   double* relres, int* iters, int* status_host, void* stream) {
  if (!vals || !cols || !b || !x || !work || !relres || !iters || !status_host || n < 1 || W < 1 || max_iter < 0)
    return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  if (Bv != 1 && Bv != Bp) return DIFFHE_E_BADARG;
  if (check_every < 1) check_every = 1;
  hipStream_t st = (hipStream_t)stream;
  if (int rcx = sync_xcd_switch(st)) return rcx;
  const dim3 grid = diffhe::node_grid(n, Bp);
  const int nblk = grid.x;
  const long long NB = (long long)n * Bp;
  double* r = work;
  double* z = r + NB;
  double* p = z + NB;
  double* Ap = p + NB;
  double* partA = Ap + NB;
  double* partB = partA + (long long)nblk * Bp;
  double* partC = partB + (long long)nblk * Bp;
  double* sc = partC + (long long)nblk * Bp;
  CgScalars S;
  S.rz = sc; S.pAp = sc + Bp; S.alpha = sc + 2 * Bp; S.beta = sc + 3 * Bp; S.bb = sc + 4 * Bp;
  S.tol2 = sc + 5 * Bp; S.rr = sc + 6 * Bp;
  S.active = (int*)(sc + 7 * Bp);
  S.iters = iters;
  S.n_active = (int*)(sc + 8 * Bp);
  S.rs = nullptr;
  S.xx = nullptr;  // plain `tol` stop on this path
  S.maxdiag = nullptr;
  S.Bv = Bv;
  const dim3 sgrid((Bp + 63) / 64);
  double* const slices = sc + 16LL * Bp;   // 2 x kEllSlices rows
  static const int two_stage = getenv("DIFFHE_ELL_SCALAR2") ? atoi(getenv("DIFFHE_ELL_SCALAR2")) : 1;

  hipLaunchKernelGGL(cg_init_kernel, grid, dim3(256), 0, st, vals, b, x, r, z, p, partA, partB, n, Bp, Bv);
  ELL_SCALAR(PH_INIT, (const double*)partA, (const double*)partB);
  int rc = diffhe::check_launch();
  if (rc) return rc;

  int it = 0, n_active = -1;
  while (it < max_iter) {
    hipLaunchKernelGGL(cg_spmv_kernel, grid, dim3(256), 0, st, vals, cols, p, Ap, partA, n, W, Bp, Bv);
    ELL_SCALAR(PH_ALPHA, (const double*)partA, (const double*)nullptr);
    hipLaunchKernelGGL(cg_update_kernel, grid, dim3(256), 0, st, vals, p, Ap, S.alpha, x, r, z, partB, partC, n, Bp,
                       Bv);
    ELL_SCALAR(PH_BETA, (const double*)partB, (const double*)partC);
    hipLaunchKernelGGL(cg_update_p_kernel<double>, grid, dim3(256), 0, st, (const double*)z, (const double*)S.beta, p, n, Bp);
    ++it;
    if (it % check_every == 0 || it == max_iter) {
      rc = diffhe::check(hipMemcpyAsync(&status_host[2], S.n_active, sizeof(int), hipMemcpyDeviceToHost, st));
      if (rc) return rc;
      rc = diffhe::check(hipStreamSynchronize(st));
      if (rc) return rc;
      n_active = status_host[2];
      if (n_active == 0) break;
    }
  }
  hipLaunchKernelGGL(residual_kernel, grid, dim3(256), 0, st, vals, cols, b, x, partA, n, W, Bp, Bv);
  ELL_SCALAR(PH_RELRES, (const double*)partA, (const double*)nullptr);
  rc = diffhe::check_launch();
  if (rc) return rc;
  status_host[0] = it;
  status_host[1] = n_active < 0 ? 0 : n_active;
  return DIFFHE_OK;
}

extern "C" int diffhe_ell_apply(const double* vals, const int* cols, const double* x, double* y, double* part, int n,
                                int W, int Bp, int Bv, void* stream) {
  if (!vals || !cols || !x || !y || !part || n < 1 || W < 1) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  if (Bv != 1 && Bv != Bp) return DIFFHE_E_BADARG;
  hipLaunchKernelGGL(cg_spmv_kernel, diffhe::node_grid(n, Bp), dim3(256), 0, (hipStream_t)stream, vals, cols, x, y,
                     part, n, W, Bp, Bv);
  return diffhe::check_launch();
}

extern "C" int diffhe_ell_galerkin(const double* vals_fine, const int* ent_ptr, const int* contrib, const double* weights,
                                   double* vals_coarse, int n_coarse, int W_coarse, int Bv, void* stream) {
  if (!vals_fine || !ent_ptr || !contrib || !vals_coarse || n_coarse < 1 || W_coarse < 1) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bv)) return DIFFHE_E_BATCHPAD;
  diffhe::account(8.0 * Bv * (double)n_coarse * W_coarse);   // the coarse values written; fine values re-read from cache
  hipLaunchKernelGGL(ell_galerkin_kernel, diffhe::node_grid(n_coarse, Bv), dim3(256), 0, (hipStream_t)stream, vals_fine,
                     ent_ptr, contrib, weights, vals_coarse, n_coarse, W_coarse, Bv);
  return diffhe::check_launch();
}

extern "C" long long diffhe_ell_amg_workspace_doubles(const diffhe_amg_level* levels, int n_levels, int Bp) {
  AmgHier H;
  if (amg_fill(H, levels, n_levels, 1, Bp)) return -1;
  const long long nb = (long long)H.lev[0].n * Bp;
  const long long nblk = cg_blocks(H.lev[0].n, Bp);
  return amg_carve(H, nullptr) + 3 * nb + 4 * nblk * Bp + (16LL + 2 * kEllSlices) * Bp + 64;
}

extern "C" int diffhe_ell_amg_pcg_solve(const diffhe_amg_level* levels, int n_levels, int Bv, const double* b, double* x,
                                        int Bp, double tol, int max_iter, int n_coarse, int gamma, double scale,
                                        int precond_fp32, double* work, double* relres, int* iters, int* status_host, void* stream) {
  if (!b || !x || !work || !relres || !iters || !status_host || max_iter < 0 || n_coarse < 1 || gamma < 1)
    return DIFFHE_E_BADARG;
  AmgHier H;
  int rc = amg_fill(H, levels, n_levels, Bv, Bp);
  if (rc) return rc;
  H.n_coarse = n_coarse; H.gamma = gamma; H.scale = scale;
  H.w0 = 0.56; H.w1 = 1.39;  // Chebyshev weights for the interval [0.5, 2] of D^-1 A
  hipStream_t st = (hipStream_t)stream;
  if (int rcx = sync_xcd_switch(st)) return rcx;
  const diffhe_amg_level& L0 = H.lev[0];
  const int n = L0.n, W = L0.W;
  const dim3 grid = diffhe::node_grid(n, Bp);
  const int nblk = grid.x;
  const long long NB = (long long)n * Bp;
  double* w = work + amg_carve(H, work);
  double* r = w;
  double* p = r + NB;
  double* Ap = p + NB;
  double* partA = Ap + NB;
  double* partB = partA + (long long)nblk * Bp;
  double* partC = partB + (long long)nblk * Bp;
  double* partD = partC + (long long)nblk * Bp;
  double* sc = partD + (long long)nblk * Bp;
  CgScalars S;
  S.rz = sc; S.pAp = sc + Bp; S.alpha = sc + 2 * Bp; S.beta = sc + 3 * Bp; S.bb = sc + 4 * Bp;
  S.tol2 = sc + 5 * Bp; S.rr = sc + 6 * Bp;
  S.active = (int*)(sc + 7 * Bp);
  S.iters = iters;
  S.n_active = (int*)(sc + 8 * Bp);
  S.rs = (precond_fp32 & 1) ? sc + 11 * Bp : nullptr;
  S.xx = (precond_fp32 & 16) ? nullptr : sc + 9 * Bp;  // bit 4: stop on `tol` alone
  S.maxdiag = sc + 10 * Bp;
  S.Bv = Bv;
  const dim3 sgrid((Bp + 63) / 64);
  double* const slices = sc + 16LL * Bp;   // 2 x kEllSlices rows
  static const int two_stage = getenv("DIFFHE_ELL_SCALAR2") ? atoi(getenv("DIFFHE_ELL_SCALAR2")) : 1;
  rc = diffhe::check(hipMemsetAsync((void*)S.maxdiag, 0, sizeof(double) * Bv, st));
  if (rc) return rc;
  if (S.xx) {
    rc = diffhe::check(hipMemsetAsync((void*)S.xx, 0, sizeof(double) * Bp, st));
    if (rc) return rc;
  }
  hipLaunchKernelGGL(ell_maxdiag_kernel, diffhe::node_grid(n, Bv, 512), dim3(256), 0, st, L0.vals, n, Bv,
                     (unsigned long long*)S.maxdiag);

  // fp32 cycle: the preconditioner STORES its vectors (and, for per-sample matrices, reads copies of the values) in
  // fp32; the CG, its residual, the iterate and every dot product stay fp64 (as in diffhe_lattice_pcg_solve)
  const bool f32 = (precond_fp32 & 1) != 0;
  float* r32 = f32 ? (float*)H.rhs[0] : nullptr;
  const void* z = nullptr;
  auto precondition = [&]() {
    if (f32) z = amg_cycle<float>(H, 0, (const float*)r32, partB, st);
    else z = amg_cycle<double>(H, 0, (const double*)r, partB, st);
  };
  auto update_p = [&]() {
    if (f32) hipLaunchKernelGGL(cg_update_p_kernel<float>, grid, dim3(256), 0, st, (const float*)z, (const double*)S.beta, p, n, Bp);
    else hipLaunchKernelGGL(cg_update_p_kernel<double>, grid, dim3(256), 0, st, (const double*)z, (const double*)S.beta, p, n, Bp);
  };
  hipLaunchKernelGGL(amg_init_kernel, grid, dim3(256), 0, st, b, x, r, r32, p, partC, n, Bp);
  if (f32) {  // fp32 copy of rs * b, rs ~ 1 / |b| a power of two: keeps the cycle inside the fp32 range
    ELL_SCALAR(PH_SCALE, (const double*)partC, (const double*)nullptr);
    hipLaunchKernelGGL(amg_cvt_kernel, grid, dim3(256), 0, st, (const double*)r, (const double*)S.rs, r32, n, Bp);
  }
  precondition();
  ELL_SCALAR(PH_INIT, (const double*)partB, (const double*)partC);
  update_p();  // beta = 0: p = z
  rc = diffhe::check_launch();
  if (rc) return rc;
  int it = 0, n_active = -1;
  while (it < max_iter) {
    if (!launch_ellw<W_SPMV, double, double>(L0.vals, L0.cols, (const double*)nullptr, (const double*)p, Ap, 0.0, partA, n, W,
                                             Bp, Bv, st))
      hipLaunchKernelGGL(cg_spmv_kernel, grid, dim3(256), 0, st, L0.vals, L0.cols, (const double*)p, Ap, partA, n, W, Bp, Bv);
    ELL_SCALAR(PH_ALPHA, (const double*)partA, (const double*)nullptr);
    hipLaunchKernelGGL(amg_update_kernel, grid, dim3(256), 0, st, (const double*)p, (const double*)Ap,
                       (const double*)S.alpha, x, r, r32, (const double*)S.rs, partC, S.xx ? partD : (double*)nullptr, n, Bp);
    if (S.xx)
      ELL_SCALAR(PH_XX, (const double*)partD, (const double*)nullptr);
    precondition();
    ELL_SCALAR(PH_BETA, (const double*)partB, (const double*)partC);
    update_p();
    ++it;
    rc = diffhe::check(hipMemcpyAsync(&status_host[2], S.n_active, sizeof(int), hipMemcpyDeviceToHost, st));
    if (rc) return rc;
    rc = diffhe::check(hipStreamSynchronize(st));
    if (rc) return rc;
    n_active = status_host[2];
    if (n_active == 0) break;
  }
  hipLaunchKernelGGL(residual_kernel, grid, dim3(256), 0, st, L0.vals, L0.cols, b, (const double*)x, partA, n, W, Bp, Bv);
  ELL_SCALAR(PH_RELRES, (const double*)partA, (const double*)nullptr);
  rc = diffhe::check_launch();
  if (rc) return rc;
  status_host[0] = it;
  status_host[1] = n_active < 0 ? 0 : n_active;
  return DIFFHE_OK;
}

extern "C" int diffhe_grad_kappa_blocks(int m, int Bp) { return (int)diffhe::node_grid(m, Bp).x; }

extern "C" int diffhe_p1_grad_kappa(const int* elems, const double* k0, const double* lam, const double* u,
                                    const double* g, int npe, int m, int Bp, double* dk_e, double* dk_part,
                                    double* dk_sum, void* stream) {
  if (!elems || !k0 || !lam || !u || !dk_part || !dk_sum || (npe != 2 && npe != 3 && npe != 6) || m < 1) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  const dim3 grid = diffhe::node_grid(m, Bp);
  diffhe::account(8.0 * Bp * (2.0 * m * (npe == 3 ? 0.5 : 1.0) + (dk_e ? m : 0)));  // lambda and u once per node, dk per element
  hipLaunchKernelGGL(grad_kappa_kernel, grid, dim3(256), 0, (hipStream_t)stream, elems, k0, lam, u, g, npe, m, Bp,
                     dk_e, dk_part);
  hipLaunchKernelGGL(sum_partials_kernel, dim3((Bp + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const double*)dk_part, (int)grid.x, Bp, dk_sum);
  return diffhe::check_launch();
}

extern "C" int diffhe_p1_grad_kappa_shared(const int* elems, const double* k0, const double* lam, const double* u,
                                           const double* g, int npe, int m, int B, int Bp, double* dk, void* stream) {
  if (!elems || !k0 || !lam || !u || !dk || (npe != 2 && npe != 3 && npe != 6) || m < 1 || B < 1 || B > Bp) return DIFFHE_E_BADARG;
  if (!diffhe::valid_batch_pad(Bp)) return DIFFHE_E_BATCHPAD;
  long long blocks = ((long long)m + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  diffhe::account(8.0 * (Bp * 2.0 * m * (npe == 3 ? 0.5 : 1.0) + m));  // lambda and u once per node, dk once per element
  hipLaunchKernelGGL(grad_kappa_shared_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, elems, k0, lam, u,
                     g, npe, m, B, Bp, dk);
  return diffhe::check_launch();
}

extern "C" int diffhe_to_node_major(const double* src, long long ld, const unsigned char* zero_mask, double* dst, int n,
                                    int B, int Bp, void* stream) {
  if (!src || !dst || n < 1 || B < 1 || Bp < B) return DIFFHE_E_BADARG;
  dim3 grid((n + kT - 1) / kT, (Bp + kT - 1) / kT);
  diffhe::account(8.0 * n * ((ld ? (double)B : 1.0) + Bp));
  hipLaunchKernelGGL(to_node_major_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ld, zero_mask, dst, n, B, Bp);
  return diffhe::check_launch();
}

extern "C" int diffhe_to_sample_major(const double* src, const double* add, double* dst, long long ld, int n, int B,
                                      int Bp, void* stream) {
  if (!src || !dst || n < 1 || B < 1 || Bp < B) return DIFFHE_E_BADARG;
  dim3 grid((n + kT - 1) / kT, (Bp + kT - 1) / kT);
  diffhe::account(8.0 * n * ((double)B + Bp));
  hipLaunchKernelGGL(to_sample_major_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, add, dst, ld, n, B, Bp);
  return diffhe::check_launch();
}

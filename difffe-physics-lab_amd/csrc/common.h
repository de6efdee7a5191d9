// Shared device/host helpers for libdiffhe_hip (gfx950 only, wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "diffhe_hip.h"

namespace diffhe {

constexpr int kWave = 64;

void set_last_error(hipError_t e);

// Algorithmic-byte accounting (diffhe_traffic_account): every launch on the solve path adds the unique bytes it
// must read + write once (batch-shared data counts 0).  Process-wide: the adjoint solve runs on
// autograd's thread.  bench.py divides the total of a step by the step's duration for the step-level roofline.
void account(double bytes);

// Post-launch check: records the HIP error text and maps to DIFFHE_E_LAUNCH.
inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_error(e);
    return DIFFHE_E_LAUNCH;
  }
  return DIFFHE_OK;
}

inline int check(hipError_t e) {
  if (e != hipSuccess) {
    set_last_error(e);
    return DIFFHE_E_LAUNCH;
  }
  return DIFFHE_OK;
}

// Bp must be a power of two <= 64 or a multiple of 64 (see diffhe_hip.h).
inline bool valid_batch_pad(int Bp) {
  if (Bp <= 0) return false;
  if (Bp <= 64) return (Bp & (Bp - 1)) == 0;
  return (Bp % 64) == 0;
}

// Node-major thread mapping shared by every (n, Bp) kernel.
//   lanes over samples: LB = min(Bp, 64); nodes per wave: NPW = 64 / LB
//   block = 256 threads = 4 waves; grid.y = sample chunks of 64; grid.x strides nodes
struct NodeMap {
  int b;       // sample index of this lane
  int node0;   // first node of this lane
  int stride;  // node stride of the grid-stride loop
};

__device__ inline NodeMap node_map(int Bp) {
  const int LB = Bp < kWave ? Bp : kWave;
  const int npw = kWave / LB;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  NodeMap m;
  m.b = blockIdx.y * kWave + (lane % LB);
  m.node0 = (blockIdx.x * 4 + wave) * npw + lane / LB;
  m.stride = gridDim.x * 4 * npw;
  return m;
}

// Sum `v` over the lanes that hold the same sample (lane % LB equal), then over
// the block's 4 waves; the result is valid in wave 0, lanes < LB.
__device__ inline double block_sum_per_sample(double v, int Bp, double* lds /* >= 4*64 doubles */) {
  const int LB = Bp < kWave ? Bp : kWave;
  for (int off = LB; off < kWave; off <<= 1) v += __shfl_xor(v, off);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  lds[wave * kWave + lane] = v;
  __syncthreads();
  double s = 0.0;
  if (wave == 0) s = (lds[lane] + lds[kWave + lane]) + (lds[2 * kWave + lane] + lds[3 * kWave + lane]);
  __syncthreads();
  return s;
}

inline dim3 node_grid(int n, int Bp, int max_blocks_x = 2048) {
  const int LB = Bp < kWave ? Bp : kWave;
  const int npw = kWave / LB;
  long long groups = ((long long)n + 4 * npw - 1) / (4 * npw);
  int gx = (int)(groups < max_blocks_x ? groups : max_blocks_x);
  if (gx < 1) gx = 1;
  int gy = (Bp + kWave - 1) / kWave;
  return dim3(gx, gy, 1);
}

}  // namespace diffhe

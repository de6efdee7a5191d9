// Stream-copy ceilings on this GPU for the access shapes used by the solver kernels.
// hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o /tmp/stream_bench && /tmp/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void copy_kernel(const T* __restrict__ a, T* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
  for (; i + (UNROLL - 1) * 256 < n; i += stride) {
    T v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) b[i + u * 256] = v[u];
  }
}

template <typename T, int UNROLL>
__global__ __launch_bounds__(256) void triad_kernel(const T* __restrict__ a, const T* __restrict__ c, T* __restrict__ b,
                                                    size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
  for (; i + (UNROLL - 1) * 256 < n; i += stride) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) b[i + u * 256] = a[i + u * 256] + c[i + u * 256];
  }
}

template <typename F>
double timeit(F f, int reps = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 / reps;
}

int main() {
  const size_t bytes = (size_t)1025 * 1025 * 256 * 8;  // one fp64 vector of the bench workload
  char *a, *b, *c;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes);
  hipMemset(a, 1, bytes); hipMemset(c, 1, bytes);
  for (int blocks : {2048, 4096, 8192, 16384}) {
    double t;
    t = timeit([&] { hipLaunchKernelGGL((copy_kernel<float, 4>), dim3(blocks), dim3(256), 0, 0, (const float*)a, (float*)b, bytes / 4); });
    printf("blocks %5d copy  4B/lane x4: %7.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((copy_kernel<double, 4>), dim3(blocks), dim3(256), 0, 0, (const double*)a, (double*)b, bytes / 8); });
    printf("blocks %5d copy  8B/lane x4: %7.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((copy_kernel<double2, 4>), dim3(blocks), dim3(256), 0, 0, (const double2*)a, (double2*)b, bytes / 16); });
    printf("blocks %5d copy 16B/lane x4: %7.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((copy_kernel<double2, 8>), dim3(blocks), dim3(256), 0, 0, (const double2*)a, (double2*)b, bytes / 16); });
    printf("blocks %5d copy 16B/lane x8: %7.1f GB/s\n", blocks, 2.0 * bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((triad_kernel<double, 4>), dim3(blocks), dim3(256), 0, 0, (const double*)a, (const double*)c, (double*)b, bytes / 8); });
    printf("blocks %5d add   8B/lane x4 (2 reads 1 write): %7.1f GB/s\n", blocks, 3.0 * bytes / t / 1e9);
    t = timeit([&] { hipLaunchKernelGGL((triad_kernel<double2, 4>), dim3(blocks), dim3(256), 0, 0, (const double2*)a, (const double2*)c, (double2*)b, bytes / 16); });
    printf("blocks %5d add  16B/lane x4 (2 reads 1 write): %7.1f GB/s\n", blocks, 3.0 * bytes / t / 1e9);
  }
  double t = timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  printf("hipMemcpy D2D: %7.1f GB/s\n", 2.0 * bytes / t / 1e9);
  return 0;
}

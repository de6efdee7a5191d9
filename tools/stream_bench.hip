// Stream ceilings on this GPU for the access shapes used by the solver kernels: bytes per lane (4 / 8 / 16),
// read : write mixes, plain vs nontemporal.  Sets the "practical ceiling" the kernels are judged against
// (the microarchitecture guide quotes 6.29 TB/s for a float4 copy).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o tools/stream_bench.bin && tools/stream_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <typename T> __device__ inline T ld(const T* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <typename T> __device__ inline void st(T* p, T v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d2 __attribute__((ext_vector_type(2)));

// NR reads, NW writes per element; UNROLL independent elements per thread per trip
template <typename T, int NR, int NW, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_kernel(const T* __restrict__ a, const T* __restrict__ c, T* __restrict__ b,
                                                     T* __restrict__ d, size_t n, T* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  T acc = T(0);
  for (; i + (UNROLL - 1) * 256 < n; i += stride) {
    T v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      v[u] = NR >= 1 ? ld(a + i + u * 256, NT) : T(1);
      if (NR >= 2) v[u] += ld(c + i + u * 256, NT);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NW >= 1) st(b + i + u * 256, v[u], NT);
      if (NW >= 2) st(d + i + u * 256, v[u], NT);
      if (NW == 0) acc += v[u];
    }
  }
  if (NW == 0 && sink && acc[0] == 123.456f) *sink = acc;
}
// scalar element types need acc[0] too: wrap them as 1-vectors
typedef float f1 __attribute__((ext_vector_type(1)));
typedef double d1 __attribute__((ext_vector_type(1)));

template <typename F>
double timeit(F f, int reps = 20) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms * 1e-3 / reps;
}

template <typename T, int NR, int NW, int UNROLL, bool NT>
void run(const char* name, int blocks, char* a, char* c, char* b, char* d, size_t bytes) {
  const size_t n = bytes / sizeof(T);
  const double t = timeit([&] {
    hipLaunchKernelGGL((stream_kernel<T, NR, NW, UNROLL, NT>), dim3(blocks), dim3(256), 0, 0, (const T*)a, (const T*)c,
                       (T*)b, (T*)d, n, (T*)nullptr);
  });
  printf("  %-44s %2zu B/lane x%d %s blocks %5d : %7.1f GB/s\n", name, sizeof(T), UNROLL, NT ? "nt   " : "plain", blocks,
         (double)(NR + NW) * bytes / t / 1e9);
}

int main() {
  const size_t bytes = (size_t)1025 * 1025 * 256 * 8;  // one fp64 vector of the bench workload (2.15 GB)
  char *a, *b, *c, *d;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(c, 1, bytes));
  for (int blocks : {2048, 8192, 32768}) {
    printf("blocks = %d\n", blocks);
    run<f1, 1, 1, 4, false>("copy (1R 1W)", blocks, a, c, b, d, bytes);
    run<f2, 1, 1, 4, false>("copy (1R 1W)", blocks, a, c, b, d, bytes);
    run<d1, 1, 1, 4, false>("copy (1R 1W)", blocks, a, c, b, d, bytes);
    run<f4, 1, 1, 4, false>("copy (1R 1W)", blocks, a, c, b, d, bytes);
    run<f4, 1, 1, 4, true>("copy (1R 1W)", blocks, a, c, b, d, bytes);
    run<d2, 1, 1, 8, false>("copy (1R 1W)", blocks, a, c, b, d, bytes);
    run<f1, 1, 0, 8, false>("read only", blocks, a, c, b, d, bytes);
    run<d1, 1, 0, 8, false>("read only", blocks, a, c, b, d, bytes);
    run<f4, 1, 0, 8, false>("read only", blocks, a, c, b, d, bytes);
    run<f1, 0, 1, 8, false>("write only", blocks, a, c, b, d, bytes);
    run<d1, 0, 1, 8, false>("write only", blocks, a, c, b, d, bytes);
    run<f4, 0, 1, 8, false>("write only", blocks, a, c, b, d, bytes);
    run<f4, 0, 1, 8, true>("write only", blocks, a, c, b, d, bytes);
    run<f1, 2, 1, 4, false>("add (2R 1W)", blocks, a, c, b, d, bytes);
    run<d1, 2, 1, 4, false>("add (2R 1W)", blocks, a, c, b, d, bytes);
    run<d1, 2, 1, 4, true>("add (2R 1W)", blocks, a, c, b, d, bytes);
    run<f4, 2, 1, 4, false>("add (2R 1W)", blocks, a, c, b, d, bytes);
    run<d1, 2, 2, 4, false>("2R 2W (the CG residual update's mix)", blocks, a, c, b, d, bytes);
    run<d1, 2, 2, 4, true>("2R 2W (the CG residual update's mix)", blocks, a, c, b, d, bytes);
    run<f4, 2, 2, 4, false>("2R 2W", blocks, a, c, b, d, bytes);
    run<f4, 2, 2, 4, true>("2R 2W", blocks, a, c, b, d, bytes);
  }
  const double t = timeit([&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  printf("hipMemcpy D2D: %7.1f GB/s\n", 2.0 * bytes / t / 1e9);
  return 0;
}
